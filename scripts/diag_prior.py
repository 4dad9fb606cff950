"""Diagnostic: prior sampler timing with the four streamed weight matrices of every layer aliased onto ONE 512 KB
buffer (L2-hot) versus the real 8.3 MB working set: separates the memory stream from compute + barriers."""
import sys, time, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd.weights import make_prior_weights
from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
import avi_talking_amd.lib as L
dev = torch.device("cuda:0")
prior = InstructDiffusionPrior.from_state_dict(make_prior_weights(3), device=dev)
B = 32
te = torch.randn(B, 1, 128, device=dev); noise = torch.randn(101, B, 1, 128, device=dev)
def run(tag):
    for _ in range(2): prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": te}, noise=noise)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print(f"{tag}: {dt*1e3:.2f} ms ({dt*1e4:.1f} us/step)")
run("real weights (8.3 MB/step)")
buf = torch.randn(128 * 1024, device=dev) * 0.01
cw = prior.net.cw
for l in range(cw.depth):
    ly = cw.layer[l]
    ly.wqkv = ly.wout = ly.w1 = ly.w2 = buf.data_ptr()
run("aliased weights (512 KB, L2-hot)")
