"""In-kernel phase timing of the ping-pong GEMM (dev tool).  Needs a diagnostic build:
   touch avi-talking_amd/csrc/gemm_pp.hip && AVI_DEFINES=-DAVI_PP_STAMPS python avi-talking_amd/build.py
Prints, per workgroup (median over the grid): prologue / K loop / epilogue cycles and the in-kernel clock."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
name, batch, M, N, K, lda = "conv2", 32, 7999, 512, 1536, 1024
if len(sys.argv) > 1 and sys.argv[1] == "conv1":
    name, M = "conv1", 15999
rows = (M - 1) * lda + K
A = torch.randn(batch, rows, device=dev)
W = torch.randn(N, K, device=dev) * K ** -0.5
pw = ops.PackedWeight(W)
hi = A.to(torch.bfloat16); lo = (A - hi.float()).to(torch.bfloat16)
Ah, Al = hi.view(torch.int16), lo.view(torch.int16)
Ch = torch.empty(batch, M, N, dtype=torch.int16, device=dev); Cl = torch.empty_like(Ch)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
Cf = torch.empty(batch, M, N, device=dev)
for prec, act, outmode in ((3, ops.ACT_GELU, "planes"), (3, ops.ACT_NONE, "planes"), (3, ops.ACT_GELU, "fp32"),
                           (3, ops.ACT_NONE, "fp32"), (1, ops.ACT_GELU, "planes")):
    print(f"--- act={act} out={outmode}")
    stamps = torch.zeros(batch * tiles * 2 * 6, dtype=torch.int64, device=dev)
    def run(flag):
        ops.gemm_raw(Ahi=Ah.data_ptr(), Alo=Al.data_ptr(), lda=lda, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(),
                     Chi=Ch.data_ptr() if outmode == "planes" else 0, Clo=Cl.data_ptr() if outmode == "planes" else 0,
                     C_=Cf.data_ptr() if outmode == "fp32" else 0, ldc=N, M=M, N=N, K=K, act=act, prec=prec | flag,
                     batch=batch, sA=(rows, 0), sC=(M * N, 0),
                     scale=stamps.data_ptr() if flag else 0, shift=stamps.data_ptr() if flag else 0)
    for _ in range(20): run(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(0x800); e1.record(); torch.cuda.synchronize()
    st = stamps.view(-1, 2, 6).cpu().double()
    pro, loop, epi = st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
    clk = (st[:, :, 3] - st[:, :, 0]) / (st[:, :, 5] - st[:, :, 4]) * 100e6
    nk = K // (32 if prec == 3 else 64)
    print(f"{name} prec={prec}: launch {e0.elapsed_time(e1)*1e3:.0f} us, {tiles*batch} tiles; per tile (median, group A / B) "
          f"prologue {pro[:,0].median():.0f}/{pro[:,1].median():.0f}  loop {loop[:,0].median():.0f}/{loop[:,1].median():.0f} "
          f"(= {loop[:,0].median()/nk/8:.0f} cyc per barrier interval, ideal {384 if prec == 3 else 256})  "
          f"epilogue {epi[:,0].median():.0f}/{epi[:,1].median():.0f} cycles; clock {clk.median()/1e9:.2f} GHz")
    t0 = st[:, 0, 4]; t0 = (t0 - t0.min()) / 100.0
    dur = (st[:, 0, 5] - st[:, 0, 4]) / 100.0
    print(f"   tile start spread: {t0.max():.0f} us; tile duration median {dur.median():.1f} us (min {dur.min():.1f}, max {dur.max():.1f})")
