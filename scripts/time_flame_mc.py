"""Time the matrix-core FLAME path alone at config[1] size: python scripts/time_flame_mc.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = importlib.import_module("avi_talking_amd.weights")
from avi_talking_amd.host.flame import FLAME  # noqa: E402

dev = torch.device("cuda:0")
B, T = 32, 250
g = torch.Generator(device=dev).manual_seed(11)
shape = torch.randn(B, 300, device=dev, generator=g)
exp = torch.randn(B, T, 50, device=dev, generator=g) * 0.8
pose = torch.zeros(B, T, 15, device=dev)
pose[..., 6:9] = torch.randn(B, T, 3, device=dev, generator=g) * 0.1
fl = FLAME(W.make_flame_basis(4), device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(3):
    v = fl.vertices(shape, exp, pose)
torch.cuda.synchronize()
best = 1e9
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    v = fl.vertices(shape, exp, pose)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"FLAME pass (matrix cores), {B}x{T} frames: best {best * 1e3:.1f} us  ({v.numel() * 4 / best / 1e6:.0f} GB/s of vertices)", flush=True)
