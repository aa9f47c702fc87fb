#!/usr/bin/env python3
"""Where does the HOST spend a small frame?  (GPU box)   python3 tools/host_profile.py [C1] [steps]
cProfile over the bench's own step (distributed.multi_view_step on a HipViewRenderer, one view per step, two cameras), no
synchronisation inside the loop.  Prints the top of the cumulative and the self-time tables."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gaussiansplat_amd import distributed as D, renderer as R, synthetic  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n, W, H, deg = synthetic.CONFIGS[cfg]
seed = 1234 + ["C1", "C2", "C3", "C4", "C5"].index(cfg)
scene = synthetic.make_scene(n, W, H, deg, seed=seed)
gx, gy = (W + 15) // 16, (H + 15) // 16
r = R.getRenderer("GAUSSIAN_3D", (W, H, 3), (16, 16), (gx, gy), scene, t_min=1e-5)
hv = D.HipViewRenderer(r, 3, False)
cams = {}
for v in (0, 4):
    cam = synthetic.scene_camera(W, view=v); cam.id = v; cams[v] = cam
dCs = {v: torch.as_tensor(synthetic.make_dC(W, H, seed + v)).cuda() for v in (0, 4)}
def step(k):
    v = (0, 4)[k & 1]
    D.multi_view_step(hv, [cams[v]], [dCs[v]])
for k in range(100):
    step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host_ms %.4f frame_ms %.4f" % ((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for k in range(K):
    step(k)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    st = pstats.Stats(pr); st.sort_stats(key).print_stats(22)
