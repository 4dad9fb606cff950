"""Dev tool: the training step alone (B = 64, hipGraph) for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import weights as W
from avi_talking_amd.host.training import PriorTrainer
dev = torch.device("cuda:0")
tr = PriorTrainer(W.make_prior_weights(3), device=dev, lr=1e-4)
g = torch.Generator(device=dev).manual_seed(4321)
B = 64
voxel = torch.randn(B, 768, device=dev, generator=g)
target = torch.randn(B, 1, 128, device=dev, generator=g) * 0.3
rand = tr.draw(B, generator=g)
if "--eager" in sys.argv:
    step = lambda: tr.train_step(voxel, target, 0.005, rand=rand)
else:
    tr.capture_step(voxel, target, 0.005, rand)
    step = tr.replay_step
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
N = 20
for _ in range(N): step()
torch.cuda.synchronize()
print(f"train step: {(time.perf_counter() - t) / N * 1e3:.3f} ms")
