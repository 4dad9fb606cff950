"""Time the CLIP text encoder (B prompts x 77 tokens) on the GPU: python scripts/time_clip.py [B]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = importlib.import_module("avi_talking_amd.weights")
from avi_talking_amd.host.clip_text import FrozenCLIPEmbedder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
m = FrozenCLIPEmbedder(W.make_clip_text_weights(5), device=dev)
ids = torch.randint(0, 49408, (B, 77), device=dev)
m.capture(ids)              # one hipGraph per forward, as bench.py's clip_text leg times it
for _ in range(3):
    m.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    m.replay()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flops = 2.0 * B * 77 * 12 * (768 * 2304 + 768 * 768 + 2 * 768 * 3072)
print(f"clip_text B={B} (small_rows {m.small_rows}, split_rows {m.split_rows}): {ms:.3f} ms/forward  {B / ms * 1e3:.0f} prompts/s  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)")
