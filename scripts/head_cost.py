"""Dev tool: what the previous pass's head costs the pass it runs beside - the head graph replayed 0 / 1 / 2 / 4 times per
pass, same device, same process (scripts/ab_pipeline.py conventions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avi_talking_amd as pkg
pkg.request_hw_queues(8)
from avi_talking_amd import weights as W
from avi_talking_amd.host.pipeline import SamplingPipeline
import bench

dev = torch.device("cuda:0")
B = 32
p = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=dev, prec="mixed",
                     rng_seed=1)
pcm = bench.synth_audio(B, 160000, 1234).to(dev)
voxel = torch.randn(B, 768, generator=torch.Generator().manual_seed(1235)).to(dev)
p.capture_pipelined(pcm, voxel, None)
print(p.arrangement, flush=True)
g = p._g_head


class Rep:
    def __init__(self, n):
        self.n = n

    def replay(self):
        for _ in range(self.n):
            g.replay()


for rnd in range(2):
    for n in (1, 0, 2, 4, 1):
        p._g_head = Rep(n)
        for _ in range(3):
            p.replay_pipelined()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            p.replay_pipelined()
        torch.cuda.synchronize()
        print(f"head x{n}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/pass", flush=True)
