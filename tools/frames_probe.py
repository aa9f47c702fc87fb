"""Frame time of a fresh renderer over its first 48 frames (groups of four) and the shader clock after each group (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from gaussiansplat_amd import renderer as R, synthetic, distributed as D
n, W, H, deg = synthetic.CONFIGS["C3"]
scene = synthetic.make_scene(n, W, H, deg, seed=1234 + 2)
gx, gy = (W + 15) // 16, (H + 15) // 16
cams = [synthetic.scene_camera(W, view=v) for v in (0, 4)]
dCs = [torch.as_tensor(synthetic.make_dC(W, H, 1236 + v)).cuda() for v in (0, 4)]
for rep in range(2):
    r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, device=0)
    hv = D.HipViewRenderer(r)
    torch.cuda.synchronize()
    ts, mhz = [], []
    for g in range(12):                      # groups of 4 frames
        t0 = time.perf_counter()
        for i in range(4):
            k = 4 * g + i
            D.multi_view_step(hv, [cams[k % 2]], [dCs[k % 2]], sync="allreduce", overlap=False, pipeline=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 4 * 1e3)
        mhz.append(r.ctx.clock_mhz())                # one wave, 20 us, right behind the group's last kernel
    print("rep", rep, "ms/frame", " ".join("%.3f" % t for t in ts), flush=True)
    print("rep", rep, "clock MHz", " ".join("%.0f" % m for m in mhz), flush=True)
    del hv, r
