import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from gaussiansplat_amd import backend as B
W,H=1920,1080
ctx=B.Context()
gt=torch.rand((3,H,W),device="cuda"); img=(gt+0.1*torch.randn_like(gt)).clamp(0,1); dC=torch.zeros_like(img)
for _ in range(3): ctx.loss_device(img.data_ptr(),gt.data_ptr(),dC.data_ptr(),W,H,3,0.1,want_loss=False)
torch.cuda.synchronize()
d=dC.flatten()[:8*6120].cpu().numpy().reshape(-1,8)
print("phase cycles (memtime ticks=100MHz?) mean fetch %.0f loop %.0f epilogue %.0f" % (d[:,0].mean(), d[:,1].mean(), d[:,2].mean()))
print("percentiles fetch", np.percentile(d[:,0],[10,50,90]), "loop", np.percentile(d[:,1],[10,50,90]), "epi", np.percentile(d[:,2],[10,50,90]))
span=(d[:,4]-d[:,3]) % (1<<24)
print("wg lifetime mean", span.mean())
