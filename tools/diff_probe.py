"""Which gradient arrays differ between one gs_backward and the chain in steps (tests/test_gpu_multiview.py::test_chain_in_two_steps_equals_one_backward)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_multiview as M
from gaussiansplat_amd import renderer as R
r, cams, dCs = M._setup()
out = {}
for mode in ("plain", "two_step"):
    R.resetGrads(r)
    for v in (2, 5):
        tps = R.preprocess(r, cams[v]); R.compactIdxs(r); R.forward(r, tps)
        if mode == "plain":
            R.backward(r, dCs[v])
        else:
            R.backward(r, dCs[v], phase="composite"); R.backward(r, dCs[v], phase="params_sh"); R.backward(r, dCs[v], phase="params_geom")
    torch.cuda.synchronize()
    out[mode] = r.splatGrads.flat.cpu().numpy().copy()
n = (out["plain"].size) // (11 + 3 * 16) if out["plain"].size % 59 == 0 else None
a, b = out["plain"], out["two_step"]
print("n", n, "total differing", int((a != b).sum()), "max abs", float(np.abs(a - b).max()))
if n:
    o = 0
    for name, w in (("means", 3), ("scales", 3), ("quats", 4), ("opac", 1), ("shs", 48)):
        x, y = a[o:o + n * w], b[o:o + n * w]; o += n * w
        print(name, "equal" if np.array_equal(x, y) else "differs: %d entries, max abs %.3e, max rel %.3e" % ((x != y).sum(), np.abs(x - y).max(), (np.abs(x - y) / (np.abs(y) + 1e-30)).max()))
