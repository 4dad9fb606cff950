"""GEMM micro-benchmark over the shapes of the sampling path (dev tool): TFLOP/s per shape and precision."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
shapes = [  # name, batch, M(per batch), N, K, lda
    ("conv1 k3s2", 32, 15999, 512, 1536, 1024),
    ("conv2 k3s2", 32, 7999, 512, 1536, 1024),
    ("conv3 k3s2", 32, 3999, 512, 1536, 1024),
    ("conv4 k3s2", 32, 1999, 512, 1536, 1024),
    ("conv5 k2s2", 32, 999, 512, 1024, 1024),
    ("conv6 k2s2", 32, 499, 512, 1024, 1024),
    ("featproj", 1, 8000, 768, 512, 512),
    ("qkv", 1, 8000, 2304, 768, 768),
    ("ffn1", 1, 8000, 3072, 768, 768),
    ("ffn2", 1, 8000, 768, 3072, 3072),
    ("outproj", 1, 8000, 768, 768, 768),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name, batch, M, N, K, lda in shapes:
    if only and only not in name:
        continue
    if batch == 1 and os.environ.get("GEMM_M"):     # fill experiment: rows of the encoder projections
        M = int(os.environ["GEMM_M"])
    rows = (M - 1) * lda + K
    A = torch.randn(batch, rows, device=dev)
    W = torch.randn(N, K, device=dev) * K ** -0.5
    pw = ops.PackedWeight(W)
    C = torch.empty(batch, M, N, device=dev)
    for prec in ((3, 1, 3 | 0x100, 3 | 0x200, 1 | 0x100, 1 | 0x200) if os.environ.get("GEMM_DIAG") else (3, 1)):
        def run():
            ops.gemm_raw(A=A.data_ptr(), lda=lda, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=C.data_ptr(), ldc=N,
                         M=M, N=N, K=K, act=ops.ACT_GELU, prec=prec, batch=batch, sA=(rows, 0), sC=(M * N, 0))
        if os.environ.get("GEMM_PLANES"):
            hi = A.to(torch.bfloat16); lo = (A - hi.float()).to(torch.bfloat16)
            Ah, Al = hi.view(torch.int16), lo.view(torch.int16)
            Ch = torch.empty(batch, M, N, dtype=torch.int16, device=dev); Cl = torch.empty_like(Ch)
            def run():
                ops.gemm_raw(Ahi=Ah.data_ptr(), Alo=Al.data_ptr(), lda=lda, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(),
                             Chi=Ch.data_ptr(), Clo=Cl.data_ptr(), ldc=N, M=M, N=N, K=K, act=int(os.environ.get('GEMM_ACT', ops.ACT_GELU)), prec=prec,
                             batch=batch, sA=(rows, 0), sC=(M * N, 0))
        for _ in range(3): run()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(reps): run()
        torch.cuda.synchronize(); dt = (time.time() - t) / reps
        fl = 2.0 * batch * M * N * K
        print(f"{name:12s} prec={prec:#x} {dt*1e6:8.1f} us  {fl/dt/1e12:7.1f} TFLOP/s algorithmic  ({fl*prec/dt/1e12:7.1f} MFMA)", flush=True)
