"""Paired vs unpaired DDPM sampler, 32 samples, alone on the chip (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_prior_weights
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
te = torch.randn(B, 1, 128, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
for pair in ("0", "1"):
    os.environ["AVI_PRIOR_PAIR"] = pair
    p = InstructDiffusionPrior.from_state_dict(make_prior_weights(3), device=dev)
    p.time_table()
    for _ in range(2):
        out = p.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
    torch.cuda.synchronize()
    evs = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = p.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    if pair == "1":
        p.pair_status()
    print(f"paired={pair} B={B}: {min(a.elapsed_time(b) for a, b in evs):.3f} ms per 100-step launch", flush=True)
