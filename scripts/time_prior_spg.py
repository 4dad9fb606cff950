"""Standalone time of the batched DDPM sampler for B=32 versus samples per workgroup (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_prior_weights
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
dev = torch.device("cuda:0")
prior = InstructDiffusionPrior.from_state_dict(make_prior_weights(3), device=dev)
B = 32
te = torch.randn(B, 1, 128, device=dev)
noise = torch.randn(101, B, 1, 128, device=dev)
for spg in (1, 2, 3, 4, 5):
    prior.samples_per_group = spg
    for _ in range(2):
        prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    print(f"spg={spg}: {dt*1e3:.2f} ms per 100-step loop, {(B + spg - 1)//spg} workgroups", flush=True)
