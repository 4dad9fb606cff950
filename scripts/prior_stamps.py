"""Cycles per phase of the batched DDPM sampler, summed over 6 layers x 100 steps (dev tool).  Needs a diagnostic build:
   touch avi-talking_amd/csrc/prior_mfma.inc && AVI_DEFINES=-DAVI_PRIOR_STAMPS python avi-talking_amd/build.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_prior_weights
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
dev = torch.device("cuda:0")
prior = InstructDiffusionPrior.from_state_dict(make_prior_weights(3), device=dev)
B = 32
te = torch.randn(B, 1, 128, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
for _ in range(3):
    out = prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
torch.cuda.synchronize()
c = out.view(B, 128)[0, :12].double()
names = ["A pre-LN", "B qkv run", "C attention", "D out run", "E LN x2", "F ff1 run", "G swiglu", "H ff2 run",
         "final proj", "step setup+update", "exchange 1 (+ prefetch ff1)", "exchange 2 (+ prefetch qkv)"]
tot = c[:12].sum().item()
for n, v in zip(names, c[:12].tolist()):
    print(f"{n:18s} {v/600:8.0f} cycles per layer-step  {100*v/tot:5.1f} %")
print(f"total {tot/100:.0f} cycles per DDPM step")
