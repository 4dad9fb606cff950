"""How much does the concurrent prior sampler slow the audio path?  (dev tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops, weights as W
from avi_talking_amd.host.pipeline import SamplingPipeline
dev = torch.device("cuda:0")
pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=dev)
B = 32
pcm = (torch.randn(B, 160000) * 3000).to(torch.int16).to(dev)
voxel = torch.randn(B, 768, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
style = torch.randn(B, 1, 128, device=dev)
def timeit(fn, tag, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{tag}: {(time.time()-t)/n*1e3:.2f} ms")
def audio_only():
    s = pipe.talking_head.forward_audio({"raw_audio": pcm.view(B, 250, 640), "samplerate": [16000] * B})
    return pipe.talking_head.head(s["audio_feature"], style)
def prior_only():
    return pipe.voxel2style_emb(voxel, noise)
def serial():
    st = pipe.voxel2style_emb(voxel, noise)
    s = pipe.talking_head.forward_audio({"raw_audio": pcm.view(B, 250, 640), "samplerate": [16000] * B})
    return pipe.talking_head.head(s["audio_feature"], st)
g = torch.cuda.CUDAGraph()
for name, fn in (("audio path only", audio_only), ("prior only", prior_only), ("serial", serial), ("concurrent (pipeline.run)", lambda: pipe.run(pcm, voxel, noise))):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    timeit(g.replay, name + " [graph]")
