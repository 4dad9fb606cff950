import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import weights as W
from avi_talking_amd.host.pipeline import SamplingPipeline
dev = torch.device("cuda:0")
pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=dev)
B = 32
pcm = (torch.randn(B, 160000) * 3000).to(torch.int16).to(dev)
voxel = torch.randn(B, 768, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
for spg in (0, 1, 2, 3, 4, 5):
    pipe.prior.samples_per_group = spg
    pipe.capture(pcm, voxel, noise)
    for _ in range(3): pipe.replay()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): pipe.replay()
    torch.cuda.synchronize()
    print(f"samples_per_group={spg}: {(time.time()-t)/10*1e3:.2f} ms per pass", flush=True)
# the sampler alone (no audio branch beside it), per samples_per_group
te = torch.randn(B, 1, 128, device=dev)
for spg in (1, 2, 3, 4, 5):
    f = lambda: pipe.prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise, samples_per_group=spg)
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(f"sampler alone, samples_per_group={spg}: {(time.time()-t)/5*1e3:.2f} ms", flush=True)
