"""Timing of the head-dim-64 MFMA attention at the bench shape (dev tool).  AVI_LIB_AB=<path> times another build of the
library for an A/B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avi_talking_amd.lib as L
if os.environ.get("AVI_LIB_AB"):
    L.LIB_PATH = os.environ["AVI_LIB_AB"]
from avi_talking_amd import ops
dev = torch.device("cuda:0")
H = 12
for B, T in ((32, 250), (32, 499), (4, 250)):
    qkv = torch.randn(B, T, 3 * H * 64, device=dev)
    for fmt in (ops.PLANES_BF16, ops.PLANES_F16):
        for _ in range(3):
            o = ops.attention_d64_planes(qkv, H, 0.125, fmt=fmt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            o = ops.attention_d64_planes(qkv, H, 0.125, fmt=fmt)
        e1.record(); torch.cuda.synchronize()
        print(f"attention_d64_planes B={B} T={T} fmt={fmt}: {e0.elapsed_time(e1)/50*1e3:.1f} us  sum={float(o.hi.float().sum()):.3f}")
