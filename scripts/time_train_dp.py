"""One-rank rehearsal of the data-parallel training step (dev tool): run under
  AVI_DP_FORCE_COLLECTIVES=1 python -m torch.distributed.run --standalone --nproc-per-node 1 scripts/time_train_dp.py
(or plainly: no process group, the same chain without collectives).  Prints ms per step for the eager DP step, the
segment-captured DP step and the single-graph step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if "RANK" in os.environ:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", device_id=dev)
from avi_talking_amd import weights as W
from avi_talking_amd.host.rng import DeviceRng
from avi_talking_amd.host.training import PriorTrainer
B = 64
g = torch.Generator(device=dev).manual_seed(1)
voxel = torch.randn(B, 768, device=dev, generator=g); target = torch.randn(B, 1, 128, device=dev, generator=g) * 0.3
def timed(step, n=50):
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
tr = PriorTrainer(W.make_prior_weights(3), device=dev); rand = tr.draw(B, generator=g)
print(f"eager DP step            {timed(lambda: tr.train_step(voxel, target, 0.005, rand=rand)):.3f} ms", flush=True)
tr = PriorTrainer(W.make_prior_weights(3), device=dev)
tr.capture_step_dp(voxel, target, 0.005, rng=DeviceRng(3, dev))
print(f"segment-captured DP step {timed(tr.replay_step_dp):.3f} ms (collectives: {tr.sync._collectives()})", flush=True)
if "RANK" not in os.environ:
    tr = PriorTrainer(W.make_prior_weights(3), device=dev)
    tr.capture_step(voxel, target, 0.005, rng=DeviceRng(3, dev))
    print(f"single-graph step        {timed(tr.replay_step):.3f} ms", flush=True)
if "RANK" in os.environ:
    dist.barrier(device_ids=[dev.index]); dist.destroy_process_group()
