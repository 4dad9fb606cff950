"""Aligner (BrainNetwork at 32 rows) under a hipGraph: time per pass (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import weights as W
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
dev = torch.device("cuda:0")
prior = InstructDiffusionPrior.from_state_dict(W.make_prior_weights(3), device=dev)
vox = torch.randn(32, 768, device=dev)
for _ in range(3):
    prior.voxel2clip(vox, need_projection=False)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        out = prior.voxel2clip(vox, need_projection=False)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"aligner at 32 rows: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per pass")
