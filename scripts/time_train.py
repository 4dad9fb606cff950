"""Timing of the prior training step at config[2] size (dev tool; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import weights as W
from avi_talking_amd.host.training import PriorTrainer
dev = torch.device("cuda:0")
B = 64
tr = PriorTrainer(W.make_prior_weights(3), device=dev, lr=1e-4)
g = torch.Generator(device=dev).manual_seed(4321)
voxel = torch.randn(B, 768, device=dev, generator=g)
target = torch.randn(B, 1, 128, device=dev, generator=g) * 0.3
rand = tr.draw(B, generator=g)
graph = "--eager" not in sys.argv
if graph:
    tr.capture_step(voxel, target, 0.005, rand); step = tr.replay_step
else:
    step = lambda: tr.train_step(voxel, target, 0.005, rand=rand)
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
n = 20
for _ in range(n): step()
torch.cuda.synchronize()
print(f"train step ({'graph' if graph else 'eager'}): {(time.perf_counter() - t) / n * 1e3:.3f} ms")
