"""Small-grid GEMM shapes (decoder head, training step): time per launch (dev tool).  These are bound by the MFMA rate
of the few workgroups they launch (a 128x128x64 step in 3-term mode is ~0.75 us on one CU), not by load latency: a
three-stage register prefetch changed nothing; K slices (ops.linear_ln_skinny) are what helps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
shapes = [("squash 1024x256x2048", 1024, 256, 2048), ("seq_enc 8000x128x768", 8000, 128, 768),
          ("train ff dX 192x128x1024", 192, 128, 1024), ("train to_q 192x640x128", 192, 640, 128),
          ("time mlp 64x512x512", 64, 512, 512), ("tel ffn 8000x256x256", 8000, 256, 256),
          ("bert qkv 8000x384x128", 8000, 384, 128), ("flint lin 8192x256x256", 8192, 256, 256),
          ("dW 512x128x192", 512, 128, 192), ("tel qkv 8000x768x256", 8000, 768, 256)]
for name, M, N, K in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    pw = ops.PackedWeight(w, torch.zeros(N, device=dev))
    ref = x.double() @ w.double().t()
    out = ops.linear(x, pw)
    err = (out.double() - ref).abs().max().item()
    for _ in range(5):
        ops.linear(x, pw, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(50):
            ops.linear(x, pw, out=out)
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us/launch   max err {err:.1e}")
