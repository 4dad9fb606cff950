"""Encoder-projection GEMMs (plane operands, plane output) per shape and tile choice: event-timed, stand-alone (dev tool).
usage: python scripts/bench_enc_gemm.py [cus]   (AVI_GEMM_KERNEL=4/5/6 forces 256x256 / 128x192 / 128x256)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from avi_talking_amd import ops
dev = torch.device("cuda:0")
cus = int(sys.argv[1]) if len(sys.argv) > 1 else 224
prec = int(os.environ.get("GEMM_PREC", "3"))
M = int(os.environ.get("GEMM_M", "8000"))
shapes = [("qkv", 2304, 768, ops.ACT_NONE, False), ("outproj", 768, 768, ops.ACT_NONE, True),
          ("ffn1", 3072, 768, ops.ACT_GELU, False), ("ffn2", 768, 3072, ops.ACT_NONE, True)]
fmt = ops.plane_fmt(prec)
# in-pass conditions: the 100-step sampler (32 workgroups, ~10 ms) holds 32 CUs on a high-priority side stream while the
# GEMM launches are timed (BESIDE=0: the GEMM alone on the chip)
beside = os.environ.get("BESIDE", "1") == "1"
if beside:
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
    prior = InstructDiffusionPrior.from_state_dict(W.make_prior_weights(3), device=dev, prec=ops.PREC_BF16X3)
    prior.time_table()
    side = torch.cuda.Stream(device=dev, priority=-1)
    te = torch.randn(32, 1, 128, device=dev)
    noise = torch.randn(101, 32, 1, 128, device=dev)
    def occupy():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            prior.p_sample_loop((32, 1, 128), text_cond={"text_embed": te}, cond_scale=1.0, timesteps=100, noise=noise)
for name, N, K, act, resid in shapes:
    x = torch.randn(M, K, device=dev)
    xp = ops.Planes((M, K), dev, fmt)
    dt = torch.float16 if fmt == ops.PLANES_F16 else torch.bfloat16
    hi = x.to(dt)
    xp.hi.copy_(hi.view(torch.int16)); xp.lo.copy_((x - hi.float()).to(dt).view(torch.int16))
    pw = ops.PackedWeight(torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev))
    R = torch.randn(M, N, device=dev) if resid else None
    ws = ops.stream_k_workspace(M, N, dev) if os.environ.get("SK", "0") == "1" else None
    run = lambda: ops.linear_planes(xp, pw, act=act, residual=R, prec=prec, out_planes=not resid, cus=cus, sk_ws=ws)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    evs = []
    for rep in range(3):
        if beside:
            occupy()
            run()                   # the sampler's workgroups are resident before the timed launches start
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    us = ts[len(ts) // 2] * 1e3
    fl = 2.0 * M * N * K
    print(f"{name:8s} N={N:5d} K={K:5d} cus={cus} kernel={os.environ.get('AVI_GEMM_KERNEL','auto')} sk={os.environ.get('SK','0')}: {us:7.1f} us  "
          f"{fl/us/1e6:6.1f} TFLOP/s algorithmic", flush=True)
