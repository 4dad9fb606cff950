"""GPU parity of every C-ABI op against plain torch fp32 on the CPU (same seeded inputs).

Tolerances: the bf16x3 GEMM mode (parity mode) carries ~1e-5 relative error per product sum;
bf16 mode ~4e-3 relative.  Elementwise/LN/attention kernels are fp32 and must agree to ~1e-5.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K", [(300, 512, 1536), (128, 768, 512), (1000, 48, 6144), (77, 53, 1280),
                                    (515, 2304, 768), (64, 128, 64)])
@pytest.mark.parametrize("prec", [3, 1])
def test_linear(gpu, M, N, K, prec):
    from avi_talking_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    out = ops.linear(x.to(gpu), pw, act=ops.ACT_GELU, residual=r.to(gpu), prec=prec).cpu().double()
    err = (out - ref).abs().max().item()
    tol = 2e-5 * math.sqrt(K / 64) if prec == 3 else 3e-2
    assert err < tol, (err, tol)


def test_linear_affine_noresidual(gpu):
    from avi_talking_amd import ops
    M, N, K = 200, 256, 1280
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    sc, sh = _rand((N,), 5) * 0.1 + 1, _rand((N,), 6)
    ref = F.leaky_relu(F.linear(x, w, b), 0.2) * sc + sh
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    out = ops.linear(x.to(gpu), pw, act=ops.ACT_LRELU02, scale=sc.to(gpu), shift=sh.to(gpu)).cpu()
    assert (out - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("k,s,Tin", [(3, 2, 1001), (2, 2, 400), (5, 1, 70)])
def test_conv1d_cl(gpu, k, s, Tin):
    from avi_talking_amd import ops
    B, C = 3, 512 if k < 5 else 256
    x = _rand((B, Tin, C), 7)
    w = _rand((C, C, k), 8, (C * k) ** -0.5)
    ref = F.gelu(F.conv1d(x.transpose(1, 2), w, stride=s)).transpose(1, 2)
    pw = ops.PackedWeight(w.permute(0, 2, 1).reshape(C, -1).to(gpu))
    out = ops.conv1d_cl(x.to(gpu), pw, k, s, act=ops.ACT_GELU).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 5e-5


def test_conv0_gn_gelu(gpu):
    from avi_talking_amd import ops
    B, N = 3, 16000 + 7
    x = _rand((B, N), 9) + 0.05
    w0 = _rand((512, 1, 10), 10, 0.6)
    g, b = _rand((512,), 11) * 0.1 + 1, _rand((512,), 12) * 0.1
    ref = F.gelu(F.group_norm(F.conv1d(x[:, None], w0, stride=5), 512, g, b, 1e-5)).transpose(1, 2)
    out = ops.conv0_gn_gelu(x.to(gpu), w0.reshape(512, 10).contiguous().to(gpu), g.to(gpu), b.to(gpu)).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("dtype", [torch.int16, torch.float32])
def test_audio_normalize(gpu, joint, dtype):
    from avi_talking_amd import ops
    x = (_rand((4, 20000), 13) * 3000 + 100)
    x = x.to(dtype)
    xf = x.float()
    if joint:
        ref = (xf - xf.mean()) / torch.sqrt(xf.var(unbiased=False) + 1e-7)
    else:
        ref = (xf - xf.mean(-1, keepdim=True)) / torch.sqrt(xf.var(-1, unbiased=False, keepdim=True) + 1e-7)
    out = ops.audio_normalize(x.to(gpu), joint=joint).cpu()
    assert (out - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("Tin,Tout", [(499, 250), (199, 100), (99, 49), (50, 50), (7, 1)])
def test_interp_layernorm(gpu, Tin, Tout):
    from avi_talking_amd import ops
    x = _rand((2, Tin, 512), 14)
    g, b = _rand((512,), 15) * 0.1 + 1, _rand((512,), 16) * 0.1
    it = F.interpolate(x.transpose(1, 2), size=Tout, align_corners=True, mode="linear").transpose(1, 2)
    ref = F.layer_norm(it, (512,), g, b, 1e-5)
    out = ops.interp_layernorm(x.to(gpu), Tout, g.to(gpu), b.to(gpu)).cpu()
    assert (out - ref).abs().max().item() < 2e-5
    out2 = ops.interp_layernorm(x.to(gpu), Tout).cpu()
    assert (out2 - it).abs().max().item() < 1e-6


@pytest.mark.parametrize("C", [64, 128, 768, 1024, 2048, 4096])
def test_layernorm(gpu, C):
    from avi_talking_amd import ops
    x = _rand((37, C), 17) * 2 + 0.5
    g, b = _rand((C,), 18) * 0.1 + 1, _rand((C,), 19) * 0.1
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    out = ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu)).cpu()
    assert (out - ref).abs().max().item() < 2e-5


def _slopes(n):
    def p2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(n).is_integer():
        return p2(n)
    c = 2 ** math.floor(math.log2(n))
    return p2(c) + _slopes(2 * c)[0::2][:n - c]


@pytest.mark.parametrize("H,D,T,mode", [(12, 64, 250, 0), (8, 16, 97, 0), (8, 32, 256, 1), (4, 16, 130, 2),
                                        (4, 256, 70, 2), (2, 128, 65, 0), (8, 16, 250, 1), (4, 32, 500, 1),
                                        (2, 32, 300, 2), (2, 64, 700, 2)])
def test_attention(gpu, H, D, T, mode):
    """avi_attention: head dims 16/32/64 on the matrix cores (bf16x3 operands), the fp32 vector kernel for the rest."""
    from avi_talking_amd import ops
    B = 2
    qkv = _rand((B, T, 3 * H * D), 20)
    q, k, v = [t.reshape(B, T, H, D).transpose(1, 2) for t in qkv.split(H * D, -1)]
    s = torch.matmul(q, k.transpose(2, 3)) * D ** -0.5
    slopes = torch.tensor(_slopes(H))
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    period = 30
    if mode == 1:
        s = s - slopes[None, :, None, None] * (i - j).abs()[None, None]
    elif mode == 2:
        bias = -slopes[None, :, None, None] * ((i - j) // period)[None, None].float()
        s = (s + bias).masked_fill((j > i)[None, None], float("-inf"))
    ref = torch.matmul(torch.softmax(s, -1), v).transpose(1, 2).reshape(B, T, H * D)
    dq = qkv.to(gpu)
    out = ops.attention(dq[..., :H * D], dq[..., H * D:2 * H * D], dq[..., 2 * H * D:], H, D, 3 * H * D, 3 * H * D,
                        T, T, B, D ** -0.5, bias_mode=mode, slopes=slopes.to(gpu), period=period).cpu()
    err = (out - ref).abs().max().item()
    assert err < (2e-5 if D > 64 else 5e-5), err      # MFMA path: operands split into bf16 hi+lo (~1e-5 relative)


def test_pad_repeat_and_pack(gpu):
    from avi_talking_amd import ops
    x = _rand((2, 5, 8), 21)
    out = ops.pad_repeat(x.to(gpu), rep=2, padL=2, padR=2, mode=1).cpu()
    ref = F.pad(x.repeat_interleave(2, 1).transpose(1, 2), (2, 2), mode="replicate").transpose(1, 2)
    assert torch.equal(out, ref)
    out0 = ops.pad_repeat(x.to(gpu), rep=1, padL=1, padR=3, mode=0).cpu()
    assert torch.equal(out0, F.pad(x.transpose(1, 2), (1, 3)).transpose(1, 2))
    h = _rand((2, 9, 96), 22)
    xg = ops.group_pad_pack(h.to(gpu), 2, 3).cpu()
    ref = F.pad(h.view(2, 9, 2, 48).permute(0, 2, 1, 3), (0, 0, 3, 3))
    assert torch.equal(xg, ref)
    a = _rand((2, 8), 23)
    assert torch.allclose(ops.add_rowbcast(x.to(gpu), a.to(gpu)).cpu(), x + a[:, None])


@pytest.mark.parametrize("B,H,T", [(2, 12, 250), (1, 3, 64), (2, 2, 65), (1, 1, 17), (1, 4, 499),
                                   (32, 12, 250), (16, 12, 1011)])   # the last two take the 2-query-tile kernel
def test_attention_d64_mfma(gpu, B, H, T):
    from avi_talking_amd import ops
    D = 64
    qkv = _rand((B, T, 3 * H * D), 30) * 1.5
    q, k, v = [t.reshape(B, T, H, D).transpose(1, 2).double() for t in qkv.split(H * D, -1)]
    ref = torch.matmul(torch.softmax(torch.matmul(q, k.transpose(2, 3)) * D ** -0.5, -1), v)
    ref = ref.transpose(1, 2).reshape(B, T, H * D)
    out = ops.attention_d64(qkv.to(gpu), H, D ** -0.5).cpu().double()
    err = (out - ref).abs().max().item()
    assert err < 1e-4, err          # values up to ~6 in magnitude: ~1e-5 relative


@pytest.mark.parametrize("M,N,K", [(300, 512, 1536), (1000, 2304, 768), (257, 128, 64)])
@pytest.mark.parametrize("prec", [3, 1])
def test_linear_split_planes(gpu, M, N, K, prec):
    """LDS-DMA GEMM on split-plane activations, fp32 and plane outputs."""
    from avi_talking_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    xp = ops.Planes((M, K), gpu)                       # split on the host: x = hi + lo
    hi = x.to(torch.bfloat16)
    xp.hi.copy_(hi.view(torch.int16).to(gpu))
    xp.lo.copy_((x - hi.float()).to(torch.bfloat16).view(torch.int16).to(gpu))
    out = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=r.to(gpu), prec=prec).cpu().double()
    err = (out - ref).abs().max().item()
    assert err < (3e-5 * math.sqrt(K / 64) if prec == 3 else 3e-2), err   # planes add 2^-17 relative on x


def test_conv_chain_split_planes(gpu):
    """conv0 -> conv (planes -> planes) -> conv (planes -> fp32) against torch."""
    from avi_talking_amd import ops
    B, N = 2, 4000 + 3
    x = _rand((B, N), 9)
    w0 = _rand((512, 1, 10), 10, 0.6)
    g, b = _rand((512,), 11) * 0.1 + 1, _rand((512,), 12) * 0.1
    w1, w2 = _rand((512, 512, 3), 13, (512 * 3) ** -0.5 * 2), _rand((512, 512, 2), 14, (512 * 2) ** -0.5 * 2)
    h = F.gelu(F.group_norm(F.conv1d(x[:, None], w0, stride=5), 512, g, b, 1e-5))
    ref1 = F.gelu(F.conv1d(h, w1, stride=2))
    ref2 = F.gelu(F.conv1d(ref1, w2, stride=2)).transpose(1, 2)
    hp = ops.conv0_gn_gelu_planes(x.to(gpu), w0.reshape(512, 10).contiguous().to(gpu), g.to(gpu), b.to(gpu))
    assert (hp.float().cpu() - h.transpose(1, 2)).abs().max().item() < 3e-5
    p1 = ops.PackedWeight(w1.permute(0, 2, 1).reshape(512, -1).to(gpu))
    p2 = ops.PackedWeight(w2.permute(0, 2, 1).reshape(512, -1).to(gpu))
    y1 = ops.conv1d_cl_planes(hp, p1, 3, 2, act=ops.ACT_GELU)
    assert (y1.float().cpu() - ref1.transpose(1, 2)).abs().max().item() < 1e-4
    y2 = ops.conv1d_cl_planes(y1, p2, 2, 2, act=ops.ACT_GELU, out_planes=False).cpu()
    assert y2.shape == ref2.shape and (y2 - ref2).abs().max().item() < 1e-4
    it = ops.interp_layernorm_planes(y1, 100).cpu()
    ref_it = F.interpolate(ref1, size=100, align_corners=True, mode="linear").transpose(1, 2)
    assert (it - ref_it).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,do_ln,res", [(32, 4096, 4096, True, True), (7, 4096, 768, True, False),
                                             (32, 128, 4096, False, False), (1, 2048, 128, True, False)])
def test_linear_ln_skinny(gpu, M, N, K, do_ln, res):
    """Split-K linear for a handful of rows + fused bias/LayerNorm/GELU/residual epilogue."""
    from avi_talking_amd import ops
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    gm, bt, r = 1 + 0.1 * _rand((N,), 4), 0.1 * _rand((N,), 5), _rand((M, N), 6)
    ref = F.linear(x.double(), w.double(), b.double())
    if do_ln:
        ref = F.gelu(F.layer_norm(ref, (N,), gm.double(), bt.double(), 1e-5))
    if res:
        ref = ref + r.double()
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    out = ops.linear_ln_skinny(x.to(gpu), pw, gm.to(gpu) if do_ln else None, bt.to(gpu) if do_ln else None,
                               do_ln=do_ln, act=ops.ACT_GELU if do_ln else ops.ACT_NONE,
                               residual=r.to(gpu) if res else None)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 5e-5, err


@pytest.mark.parametrize("M,N,K,budget", [(8000, 768, 768, 0), (8000, 768, 768, 224), (8000, 768, 3072, 224),
                                          (4000, 512, 1536, 224), (333, 256, 192, 224)])
def test_planes_gemm_tile_shapes(gpu, M, N, K, budget):
    """The plane-input GEMM picks its tile (256x256 / 128x192 / 128x256) from the fill of the CUs it may count on
    (the ``cus`` argument -> AviGemm.cus); every choice must give the same result."""
    from avi_talking_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    xp = ops.Planes((M, K), gpu)
    hi = x.to(torch.bfloat16)
    xp.hi.copy_(hi.view(torch.int16).to(gpu))
    xp.lo.copy_((x - hi.float()).to(torch.bfloat16).view(torch.int16).to(gpu))
    out = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=r.to(gpu), cus=budget).cpu().double()
    outp = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=r.to(gpu), out_planes=True, cus=budget)
    err = (out - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, (K / 64) ** 0.5), err
    assert (outp.float().cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, (K / 64) ** 0.5) + 1e-4


@pytest.mark.parametrize("M,N,K", [(200, 768, 768), (200, 2304, 768), (512, 512, 1024)])
def test_pingpong_gemm_cold_operands(gpu, M, N, K):
    """Regression for the ping-pong kernels' read / wait order: a few workgroups on freshly written (cache-cold)
    operands are where a fragment read that is ordered behind the other wave group's counted wait only by timing goes
    wrong (it did: wav2vec2 at B = 2 when group A issued its LDS-DMA pieces later in the phase).  Every wait now sits a
    barrier ahead of every reader; fresh operands per round, 256 MB written in between to empty L2 / Infinity Cache."""
    from avi_talking_amd import ops
    for rnd in range(6):
        x, w = _rand((M, K), 100 + rnd), _rand((N, K), 200 + rnd, K ** -0.5)
        ref = F.linear(x.double(), w.double())
        pw = ops.PackedWeight(w.to(gpu))
        xp = ops.Planes((M, K), gpu)
        hi = x.to(torch.bfloat16)
        xp.hi.copy_(hi.view(torch.int16).to(gpu))
        xp.lo.copy_((x - hi.float()).to(torch.bfloat16).view(torch.int16).to(gpu))
        torch.empty(64 << 20, dtype=torch.float32, device=gpu).fill_(float(rnd))      # evict
        torch.cuda.synchronize()
        out = ops.linear_planes(xp, pw).cpu().double()
        err = (out - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, (K / 64) ** 0.5), (rnd, err)


@pytest.mark.parametrize("B,T", [(2, 250), (1, 100), (3, 129), (1, 7), (3, 1), (3, 127), (3, 500)])
def test_posconv_gelu_residual(gpu, B, T):
    """wav2vec2's positional conv embedding in one launch (csrc/posconv.hip) against torch's grouped Conv1d:
    x + gelu(conv1d(x, k=128, groups=16, padding=64)[..., :-1] + bias); frames at the clip's ends see the zero padding."""
    from avi_talking_amd import ops
    G, K, Cc = 16, 128, 768
    cg = Cc // G
    x = _rand((B, T, Cc), 31)
    w = _rand((Cc, cg, K), 32, (cg * K) ** -0.5 * 2)
    bias = _rand((Cc,), 33)
    ref = x.double() + F.gelu(F.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), padding=K // 2,
                                       groups=G)[..., :-1]).transpose(1, 2)
    wg = w.view(G, cg, cg, K).permute(0, 1, 3, 2).reshape(G, cg, K * cg)          # [g][n][tap * 48 + ch]
    wpad = torch.zeros((G, 64, K * cg))
    wpad[:, :cg] = wg
    pw = ops.PackedWeight(wpad.reshape(G * 64, K * cg).to(gpu))
    xg_, bg_ = x.to(gpu), bias.to(gpu)
    out = ops.posconv_gelu_residual(xg_, pw, bg_, G, K, 64).cpu().double()
    err = (out - ref).abs().max().item()
    assert err < 5e-5, err
    # A/B against the older path (regrouping launch + overlapping-row GEMM per (clip, group), AVI_W2V_POSCONV=gemm)
    xg = ops.group_pad_pack(xg_, G, K // 2)
    h = torch.empty_like(xg_)
    Tp = T + K
    ops.gemm_raw(A=xg.data_ptr(), lda=cg, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=h.data_ptr(), ldc=Cc, M=T, N=cg,
                 K=K * cg, bias=bg_.data_ptr(), R=xg_.data_ptr(), ldr=Cc, act=ops.ACT_GELU, prec=ops.PREC_BF16X3,
                 batch=B * G, z_inner=G, sA=(G * Tp * cg, Tp * cg), sW=(0, 64 * K * cg), sC=(T * Cc, cg), sB=(0, cg),
                 sR=(T * Cc, cg))
    assert (h.cpu().double() - out).abs().max().item() < 5e-5


def test_posconv_rejects_in_place_and_overlap(gpu):
    """The kernel reads a +/-64-frame halo of x that neighbouring workgroups store as out: aliasing is refused at the ABI."""
    from avi_talking_amd import lib as L, ops
    G, K, Cc, B, T = 16, 128, 768, 2, 200
    pw = ops.PackedWeight(torch.zeros((G * 64, K * (Cc // G)), device=gpu))
    bias = torch.zeros(Cc, device=gpu)
    buf = torch.zeros((2 * B * T * Cc,), device=gpu)
    x = buf[: B * T * Cc]
    call = lambda out_ptr: L.load().avi_posconv_gelu_residual(x.data_ptr(), B, T, Cc, G, K, pw.hi.data_ptr(), pw.lo.data_ptr(),
                                                              64, bias.data_ptr(), out_ptr, L.stream_ptr())
    assert call(x.data_ptr()) == L.AVI_EINVAL                               # in place
    assert call(x.data_ptr() + 4 * (B * T * Cc - Cc)) == L.AVI_EINVAL       # tail of x overlaps the head of out
    assert call(x.data_ptr() + 4 * B * T * Cc) == 0                          # adjacent, disjoint: fine
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.int16, torch.float32])
def test_audio_normalize_unaligned_view(gpu, dtype):
    """A pcm pointer taken from an offset view (not 16-byte aligned) takes the scalar loads: same result as the aligned copy."""
    from avi_talking_amd import lib as L
    B, N = 2, 4096
    g = torch.Generator().manual_seed(17)
    base = (torch.randn(B * N + 8, generator=g) * 3000).to(dtype).to(gpu)
    view = base[1: 1 + B * N]                        # 2 or 4 bytes past a 16-byte boundary
    assert view.data_ptr() % 16 != 0
    outs = []
    for src in (view, view.clone()):
        out = torch.empty((B, N), dtype=torch.float32, device=gpu)
        stats = torch.empty((2 * B,), dtype=torch.float64, device=gpu)
        L.check(L.load().avi_audio_normalize(src.data_ptr(), int(dtype == torch.int16), B, N, 0, 1e-7, out.data_ptr(),
                                             stats.data_ptr(), L.stream_ptr()), "avi_audio_normalize")
        outs.append(out.cpu())
    x = view.cpu().float().view(B, N)
    ref = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-7)
    assert (outs[0] - ref).abs().max().item() < 1e-5 and (outs[0] - outs[1]).abs().max().item() < 1e-6


def test_transpose_jobs_and_table(gpu):
    """avi_transpose_jobs (jobs by value) and avi_transpose_table (device table): fp32 and split-plane outputs, ragged
    shapes (rows / columns not multiples of the 32x32 block), padded plane rows zero."""
    import ctypes as C
    from avi_talking_amd import lib as L
    so = L.load()
    g = torch.Generator().manual_seed(3)
    specs = [(70, 45, 64, True, True), (192, 640, 640, True, False), (33, 130, 256, False, True), (1, 7, 64, False, True)]
    jobs = (L.AviTransposeJob * len(specs))()
    keep, total = [], 0
    for jb, (R, Cc, Cp, want_f32, want_planes) in zip(jobs, specs):
        x = torch.randn(R, Cc, generator=g).to(gpu)
        out = torch.full((Cc, R), 7.0, device=gpu) if want_f32 else None
        hi = torch.full((Cp, R), 0x1234, dtype=torch.int16, device=gpu) if want_planes else None
        lo = torch.full((Cp, R), 0x1234, dtype=torch.int16, device=gpu) if want_planes else None
        jb.in_, jb.out, jb.hi, jb.lo = x.data_ptr(), L.ptr(out) or None, L.ptr(hi) or None, L.ptr(lo) or None
        jb.R, jb.C, jb.C_pad, jb.first_block = R, Cc, Cp, total
        total += jb.blocks()
        keep.append((x, out, hi, lo, Cc))

    def check():
        for x, out, hi, lo, Cc in keep:
            if out is not None:
                assert torch.equal(out, x.t())
            if hi is not None:
                v = hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()
                assert (v[:Cc] - x.t()).abs().max().item() < 2e-5 * x.abs().max().item()
                assert torch.equal(hi[:Cc].view(torch.bfloat16), x.t().contiguous().to(torch.bfloat16))
                assert not v[Cc:].any()

    L.check(so.avi_transpose_jobs(jobs, len(specs), L.stream_ptr()), "avi_transpose_jobs")
    torch.cuda.synchronize()
    check()
    for _, out, hi, lo, _ in keep:          # again through a device table
        for t in (out, hi, lo):
            if t is not None:
                t.fill_(1)
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(gpu)
    L.check(so.avi_transpose_table(table.data_ptr(), len(specs), total, L.stream_ptr()), "avi_transpose_table")
    torch.cuda.synchronize()
    check()
    assert so.avi_transpose_jobs(jobs, 5, L.stream_ptr()) != 0      # at most four jobs by value
    # a column-sum job riding in the same launch (the bias gradient of a backward GEMM pair)
    x = torch.randn(192, 45, generator=g).to(gpu)
    cs = torch.full((45,), 3.0, device=gpu)
    xt = torch.empty((45, 192), device=gpu)
    two = (L.AviTransposeJob * 2)()
    two[0].in_, two[0].out, two[0].R, two[0].C = x.data_ptr(), xt.data_ptr(), 192, 45
    two[1].in_, two[1].colsum, two[1].R, two[1].C = x.data_ptr(), cs.data_ptr(), 192, 45
    L.check(so.avi_transpose_jobs(two, 2, L.stream_ptr()), "avi_transpose_jobs")
    torch.cuda.synchronize()
    assert torch.equal(xt, x.t()) and (cs - x.sum(0)).abs().max().item() < 1e-4
    # device table with plane-only jobs whose R and C_pad are multiples of 64: the 64 x 64-tile path (16-byte stores), a ragged
    # column count inside the padded planes, and a 32 x 32 job between them
    specs = [(128, 190, 256), (192, 640, 640), (70, 45, 64), (64, 64, 64)]
    tj = (L.AviTransposeJob * len(specs))()
    keep, total = [], 0
    for jb, (R, Cc, Cp) in zip(tj, specs):
        x = torch.randn(R, Cc, generator=g).to(gpu)
        hi = torch.full((Cp, R), 0x1234, dtype=torch.int16, device=gpu)
        lo = torch.full((Cp, R), 0x1234, dtype=torch.int16, device=gpu)
        jb.in_, jb.hi, jb.lo, jb.R, jb.C, jb.C_pad, jb.first_block = x.data_ptr(), hi.data_ptr(), lo.data_ptr(), R, Cc, Cp, total
        total += jb.blocks()
        keep.append((x, hi, lo, Cc))
    assert tj[0].blocks() == 2 * 4 and tj[2].blocks() == 2 * 3
    table = torch.frombuffer(bytearray(bytes(tj)), dtype=torch.uint8).to(gpu)
    L.check(so.avi_transpose_table(table.data_ptr(), len(specs), total, L.stream_ptr()), "avi_transpose_table")
    torch.cuda.synchronize()
    for x, hi, lo, Cc in keep:
        v = hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()
        assert (v[:Cc] - x.t()).abs().max().item() < 2e-5 * x.abs().max().item()
        assert torch.equal(hi[:Cc].view(torch.bfloat16), x.t().contiguous().to(torch.bfloat16))
        assert not v[Cc:].any()


@pytest.mark.parametrize("M,N,K,budget,prec", [(8000, 768, 768, 224, 3), (8000, 2304, 768, 224, 3), (8000, 768, 3072, 224, 3),
                                               (8000, 768, 3072, 224, 2), (8000, 768, 3072, 256, 3), (4000, 768, 768, 100, 3),
                                               (1000, 768, 1536, 7, 3), (640, 2304, 768, 24, 1)])
def test_stream_k_gemm(gpu, monkeypatch, M, N, K, budget, prec):
    """Stream-K on the 128-row plane-operand GEMM (AviGemm.sk_ws, csrc/gemm_pp192.hip): the K loops of all tiles cut into
    `cus` equal shares, shared tiles finished by the last contributor.  Against float64 and against the data-parallel launch
    of the same kernel (same tile shape, no workspace); repeated launches reuse the workspace (counters return to zero);
    fp32 and plane outputs, bias, GELU and residual in the epilogue.  (The path is opt-in: at the encoder's sizes it measured
    slower than the data-parallel launch, csrc/gemm_pp192.hip.)"""
    from avi_talking_amd import ops
    monkeypatch.setenv("AVI_GEMM_STREAMK", "2")          # force: also where the plan would not bother
    monkeypatch.setenv("AVI_GEMM_KERNEL", "6")           # 128 x 256 tiles
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + r.double()
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    fmt = ops.plane_fmt(prec)
    xp = ops.Planes((M, K), gpu, fmt)
    dt = torch.float16 if fmt == ops.PLANES_F16 else torch.bfloat16
    hi = x.to(dt)
    xp.hi.copy_(hi.view(torch.int16).to(gpu))
    xp.lo.copy_((x - hi.float()).to(dt).view(torch.int16).to(gpu))
    ws = ops.stream_k_workspace(M, N, gpu)
    rg = r.to(gpu)
    plain = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=rg, prec=prec, cus=budget)
    for rep in range(3):
        out = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=rg, prec=prec, cus=budget, sk_ws=ws)
        torch.cuda.synchronize()
        err = (out.cpu().double() - ref).abs().max().item()
        tol = {3: 3e-5, 2: 2e-3, 1: 5e-2}[prec] * max(1.0, (K / 64) ** 0.5)
        assert err < tol, (rep, err)
        # a shared tile is summed in another order than a tile computed by one workgroup: a few fp32 ulps
        assert (out - plain).abs().max().item() < 2e-5 * max(1.0, (K / 64) ** 0.5)
    # the counters sit behind the slots of whichever tile shape ran (128 x 256 or 128 x 192): all back at zero, and the
    # slots were written (the launch really took the stream-K path)
    for bn in (256, 192):
        tiles = -(-M // 128) * -(-N // bn)
        cnt = ws[tiles * 4 * 128 * bn: tiles * 4 * 128 * bn + tiles].view(torch.int32)
        assert int(cnt.abs().sum()) == 0
    assert float(ws.abs().sum()) > 0
    outp = ops.linear_planes(xp, pw, act=ops.ACT_GELU, residual=rg, prec=prec, out_planes=True, cus=budget, sk_ws=ws)
    assert (outp.float() - out).abs().max().item() < 1e-4 * max(1.0, out.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(600, 512, 1536), (300, 512, 1024), (1000, 768, 768), (257, 3072, 768), (512, 512, 128),
                                   (300, 512, 192), (300, 512, 320)])
@pytest.mark.parametrize("tile", ["4", "6"])
def test_two_term_fp16_gemm_exact_against_rounded_weights(gpu, monkeypatch, tile, M, N, K):
    """AVI_PREC_F16X2 on the plane-operand kernels (256 x 256: tile "4", incl. its weight super-tile staging for K % 128 == 0
    and the older staging otherwise; 128 x 256: tile "6"): y = fp16(w) . (x_hi + x_lo).  Against float64 with the weights
    rounded to fp16 the kernel is fp32-accurate (the format's own error is the weight rounding, pinned at the path level by
    tests/test_gpu_mixed_prec.py); cold operands, ragged M, K tiles in fours / pairs / threes; and the weight super-tiles
    give the SAME result as the older staging (AVI_GEMM_WS=0) bit for bit (same products, same accumulation order)."""
    from avi_talking_amd import ops
    if tile == "6" and K % 96:
        pytest.skip("the 128-row kernel takes K tiles in threes")
    if tile == "4" and K % 64:
        pytest.skip("the 256-row kernel takes K tiles in pairs")
    monkeypatch.setenv("AVI_GEMM_KERNEL", tile)
    x, w, b = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3)
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    w16 = pw.f16_plane().view(torch.float16)[:N].cpu()           # the plane the kernel multiplies (fp16 of bf16 hi + lo)
    assert (w16.float() - w).abs().max().item() <= 2.0 ** -11 * w.abs().max().item()
    ref = F.gelu(F.linear(x.double(), w16.double(), b.double()))
    xp = ops.Planes((M, K), gpu, ops.PLANES_F16)
    hi = x.to(torch.float16)
    xp.hi.copy_(hi.view(torch.int16).to(gpu))
    xp.lo.copy_((x - hi.float()).to(torch.float16).view(torch.int16).to(gpu))
    outs = {}
    for ws in ("1", "0"):
        monkeypatch.setenv("AVI_GEMM_WS", ws)
        torch.empty(64 << 20, dtype=torch.float32, device=gpu).fill_(1.0)          # evict
        torch.cuda.synchronize()
        outs[ws] = ops.linear_planes(xp, pw, act=ops.ACT_GELU, prec=ops.PREC_F16X2).cpu()
    err = (outs["1"].double() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, (K / 64) ** 0.5), err
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("M,N,K", [(8000, 3072, 768), (2000, 768, 3072), (1000, 2304, 768), (333, 768, 1536)])
@pytest.mark.parametrize("prec", [3, 2])
def test_plane_gemm_result_does_not_depend_on_the_tile_shape(gpu, monkeypatch, M, N, K, prec):
    """The dispatcher picks 256 x 256, 128 x 256 or 128 x 192 tiles from M, N and the CU budget (csrc/gemm.hip tile_score);
    a projection must return the same bits whichever it picks - the same rows then come out the same in a full batch, a
    sub-batch or one encoder chain of it (host/wav2vec._encoder_layers_split).  Same products in the same order in all three
    kernels, and since this round the same GELU (the LDS table) in their epilogues: fp32 and plane outputs, GELU and
    residual, 3-term bf16 and 2-term fp16."""
    from avi_talking_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((N, K), 2, K ** -0.5), _rand((N,), 3), _rand((M, N), 4)
    pw = ops.PackedWeight(w.to(gpu), b.to(gpu))
    fmt = ops.plane_fmt(prec)
    xp = ops.Planes((M, K), gpu, fmt)
    dt = torch.float16 if fmt == ops.PLANES_F16 else torch.bfloat16
    hi = x.to(dt)
    xp.hi.copy_(hi.view(torch.int16).to(gpu))
    xp.lo.copy_((x - hi.float()).to(dt).view(torch.int16).to(gpu))
    rg = r.to(gpu)
    got = {}
    for tile in ("4", "5", "6"):
        monkeypatch.setenv("AVI_GEMM_KERNEL", tile)
        a = ops.linear_planes(xp, pw, act=ops.ACT_GELU, prec=prec)
        p = ops.linear_planes(xp, pw, act=ops.ACT_GELU, prec=prec, out_planes=True)
        c = ops.linear_planes(xp, pw, residual=rg, prec=prec)
        torch.cuda.synchronize()
        got[tile] = (a.cpu(), p.hi.cpu(), p.lo.cpu(), c.cpu())
    ref = F.gelu(F.linear(x.double(), w.double(), b.double()))
    assert (got["4"][0].double() - ref).abs().max().item() < {3: 3e-5, 2: 2e-3}[prec] * max(1.0, (K / 64) ** 0.5)
    for tile in ("5", "6"):
        for u, v in zip(got["4"], got[tile]):
            assert torch.equal(u, v), tile


@pytest.mark.parametrize("M,N,K,nsplit", [(2464, 768, 768, 2), (2464, 768, 3072, 4), (300, 192, 576, 3)])
def test_split_k_plane_linear(gpu, M, N, K, nsplit):
    """ops.linear_planes_splitk (K slices as one batched plane-operand launch + the partial-sum epilogue) against the
    data-parallel launch of the same product: equal up to the fp32 summation order of the slices, bias and residual included."""
    from avi_talking_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(gpu)
    w, b = (torch.randn(N, K, generator=g) / K ** 0.5).to(gpu), torch.randn(N, generator=g).to(gpu)
    res = torch.randn(M, N, generator=g).to(gpu)
    pw = ops.PackedWeight(w, b)
    xp = ops.Planes((M, K), gpu)                                          # x = hi + lo in bf16, split on the host side of the test
    hi = x.to(torch.bfloat16)
    xp.hi.copy_(hi.view(torch.int16))
    xp.lo.copy_((x - hi.float()).to(torch.bfloat16).view(torch.int16))
    one = ops.linear_planes(xp, pw, residual=res)
    cut = ops.linear_planes_splitk(xp, pw, nsplit, residual=res)
    ref = xp.float().double() @ w.double().t() + b.double() + res.double()
    e1, e2 = (one.double() - ref).abs().max().item(), (cut.double() - ref).abs().max().item()
    print(f"M={M} N={N} K={K} / {nsplit}: data-parallel {e1:.2e}, split-K {e2:.2e} vs float64")
    assert e2 < 2e-4 and (one - cut).abs().max().item() < 1e-4
    with pytest.raises(ValueError):
        ops.linear_planes_splitk(xp, pw, 5)
