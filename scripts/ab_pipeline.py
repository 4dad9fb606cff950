"""A/B of pass arrangements inside ONE process on one device (boxes differ by +-0.3 ms per pass, so settings are only
comparable when they alternate on the same device): encoder chains x paired sampler x precision plan at the bench shape.
usage: python scripts/ab_pipeline.py [rounds] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import avi_talking_amd as pkg
pkg.request_hw_queues(8)
from avi_talking_amd import weights as W
from avi_talking_amd.host.pipeline import SamplingPipeline
import bench

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
print("HW_QUEUES", pkg.HW_QUEUES, flush=True)
wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
B = 32
pcm = bench.synth_audio(B, 160000, 1234).to(dev)
voxel = torch.randn(B, 768, generator=torch.Generator().manual_seed(1235)).to(dev)
configs = [dict(split=s, pair=p) for s, p in ((1, 0), (1, 1), (2, 0), (2, 1), (4, 1))]
if os.environ.get("AB_CONFIGS"):
    configs = [dict(zip(("split", "pair"), map(int, c.split(":")))) for c in os.environ["AB_CONFIGS"].split(",")]
pipes = []
for c in configs:
    p = SamplingPipeline(wa, wh, wp, device=dev, prec=os.environ.get("AB_PREC", "mixed"), rng_seed=4242)
    p.capture_pipelined(pcm, voxel, None, arrangements=[(c["split"], bool(c["pair"]) and p.prior.paired)])
    pipes.append(p)
    torch.cuda.synchronize()
res = {i: [] for i in range(len(configs))}
host = {}
for r in range(rounds):
    for i, p in enumerate(pipes):
        for _ in range(3):
            p.replay_pipelined()
        torch.cuda.synchronize()
        t0 = time.perf_counter()               # the replays run on the pipeline's own streams: wall clock between syncs
        for _ in range(reps):
            p.replay_pipelined()
        host[i] = (time.perf_counter() - t0) / reps * 1e3       # enqueue time alone (the host runs ahead of the device)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / reps * 1e3)
class _Nop:
    def replay(self):
        pass

def body_only(p, side, audio, n=30):
    """ms per pass of the body alone (no head), optionally without the sampler's branch or without the audio branch"""
    b = p._pbody
    keep = (b.g_side, b.g_front, list(b.g_chain))
    if not side:
        b.g_side = _Nop()
    if not audio:
        b.g_front, b.g_chain = _Nop(), [_Nop() for _ in b.g_chain]
    try:
        for k in range(n + 3):
            if k == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            with torch.cuda.stream(p._s_body):
                p._enqueue_body(False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    finally:
        b.g_side, b.g_front, b.g_chain = keep

for i, c in enumerate(configs):
    p = pipes[i]
    print(c, f"body alone {body_only(p, True, True):.3f}  audio branch alone {body_only(p, False, True):.3f}  "
          f"sampler branch alone {body_only(p, True, False):.3f} ms/pass", flush=True)
for i, c in enumerate(configs):
    pipes[i].prior.pair_status()
    print(c, " ".join(f"{t:.3f}" for t in res[i]), "ms/pass; host enqueue", f"{host[i]:.3f} ms/pass", flush=True)
