"""Time FLAME vertices at config[1] size on both kernels: python scripts/time_flame.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = importlib.import_module("avi_talking_amd.weights")
from avi_talking_amd.host.flame import FLAME  # noqa: E402

dev = torch.device("cuda:0")
basis = W.make_flame_basis(4)
B, T = 32, 250
g = torch.Generator(device=dev).manual_seed(11)
shape = torch.randn(B, 300, device=dev, generator=g)
exp = torch.randn(B, T, 50, device=dev, generator=g) * 0.8
jaw = torch.randn(B, T, 3, device=dev, generator=g) * 0.1
outs = {}
for name, mc in (("vector pipe", False), ("matrix cores", True)):
    fl = FLAME(basis, device=dev, matrix_cores=mc)
    for _ in range(3):
        v = fl.from_coefficients(shape, exp, jaw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        v = fl.from_coefficients(shape, exp, jaw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    outs[name] = v
    print(f"{name}: {ms:.3f} ms per {B}x{T} frames  ({v.numel() * 4 / ms / 1e6:.0f} GB/s of vertices)")
print("max |diff| between the kernels:", (outs["vector pipe"] - outs["matrix cores"]).abs().max().item())
