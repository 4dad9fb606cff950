"""Timing of the head-dim-64 MFMA attention at the bench shape (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
B, H, T = 32, 12, 250
qkv = torch.randn(B, T, 3 * H * 64, device=dev)
for _ in range(3):
    o = ops.attention_d64(qkv, H, 0.125)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    o = ops.attention_d64(qkv, H, 0.125)
e1.record(); torch.cuda.synchronize()
print(f"attention_d64 (prep + mfma) B={B} T={T}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
