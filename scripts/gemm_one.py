import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
batch, M, N, K, lda = 32, 15999, 512, 1536, 1024
rows = (M - 1) * lda + K
A = torch.randn(batch, rows, device=dev); W = torch.randn(N, K, device=dev) * K ** -0.5
pw = ops.PackedWeight(W); C = torch.empty(batch, M, N, device=dev)
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(4):
    ops.gemm_raw(A=A.data_ptr(), lda=lda, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=C.data_ptr(), ldc=N, M=M, N=N, K=K,
                 act=ops.ACT_GELU, prec=prec, batch=batch, sA=(rows, 0), sC=(M * N, 0))
torch.cuda.synchronize()
