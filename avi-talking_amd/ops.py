"""Tensor-level wrappers over the C ABI (``lib.py``): shape checks on the host, then one launch.

Operand shapes are validated here, before the kernel is launched, so that a mismatch raises in
Python instead of faulting on the GPU.
"""
import ctypes as C

import torch

from . import lib as L
from .lib import (ACT_GELU, ACT_LRELU02, ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_SILU, PLANES_BF16,  # noqa: F401
                  PLANES_F16, PREC_BF16, PREC_BF16X3, PREC_F16X2)


# THE default precision plan of every host class (SamplingPipeline, TalkingHeadWrapper, Wav2Vec2Model, Faceformer's audio
# encoder), of host/cli.py (--prec) and of bench.py: the plan whose speed the headline reports is the plan a user of the
# drop-in entry point gets.  "mixed": conv layers 1-6 on the 2-term fp16 GEMM, everything else 3-term bf16 - 1.9-2.5e-4
# max-abs on the coefficients against the oracle and against the reference's own goldens (gate 3e-4; north_star 1e-3).
# fp16 planes have a finite range: the plane producers report overflow / underflow (host/status.py) and
# SamplingPipeline.run_checked falls back to "bf16x3" (2e-5, fp32's range) by itself.
DEFAULT_PREC = "mixed"
FALLBACK_PREC = "bf16x3"


class PrecPlan:
    """Precision per GEMM group of the audio encoder.  One ``AVI_PREC_*`` value per group: ``conv`` = conv layers 1-6
    (half of the path's FLOPs), ``attn`` = q/k/v and out projections, ``ffn`` = the two feed-forward matrices; ``small`` =
    the fp32-operand GEMMs (feature projection, heads, aligner).  Every producer emits its planes in the format its
    consumer's precision reads (``plane_fmt``), and the groups meet in fp32 (the last conv layer's output, the residual
    stream), so any combination is consistent."""

    def __init__(self, name, conv, attn, ffn, small=None, sampler_all_fp16=False):
        self.name, self.conv, self.attn, self.ffn = name, conv, attn, ffn
        self.small = small if small is not None else PREC_BF16X3
        self.sampler_all_fp16 = sampler_all_fp16

    def __repr__(self):
        return f"PrecPlan({self.name}: conv={self.conv:#x} attn={self.attn:#x} ffn={self.ffn:#x} small={self.small:#x})"


def prec_plan(prec):
    """int ``PREC_*`` -> the uniform plan of that precision; "mixed" -> 2-term fp16 conv layers under 3-term bf16
    transformer projections (measured sensitivities of the coefficients to one fp16 weight plane: conv 2e-4, ffn 2.7e-4,
    q/k/v/out 5e-4); "mixed_ffn" -> conv and ffn 2-term, q/k/v/out 3-term; a PrecPlan passes through."""
    if prec is None:
        prec = DEFAULT_PREC
    if isinstance(prec, PrecPlan):
        return prec
    if isinstance(prec, str):
        named = {"bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "f16x2": PREC_F16X2}
        if prec in named:
            prec = named[prec]
        elif prec == "mixed":
            return PrecPlan("mixed", PREC_F16X2, PREC_BF16X3, PREC_BF16X3)
        elif prec == "mixed_ffn":
            return PrecPlan("mixed_ffn", PREC_F16X2, PREC_BF16X3, PREC_F16X2)
        else:
            raise ValueError(f"unknown precision {prec!r}")
    base = prec & 0xff
    if base == PREC_F16X2:
        return PrecPlan("f16x2", prec, prec, prec, PREC_BF16X3, sampler_all_fp16=True)
    return PrecPlan({PREC_BF16X3: "bf16x3", PREC_BF16: "bf16"}.get(base, hex(prec)), prec, prec, prec, prec)


def plan_uses_fp16_planes(plan):
    """True when some GEMM group of the plan reads fp16 activation planes (whose range is finite: host/status.py)."""
    plan = prec_plan(plan)
    return any((p & 0xff) == PREC_F16X2 for p in (plan.conv, plan.attn, plan.ffn))


def fp32_operand_prec(prec):
    """Precision of the small fp32-operand GEMMs (gemm.hip) under a pipeline precision: the 2-term fp16 mode exists on
    the plane-operand kernels only, everything else stays on the 3-term bf16 split."""
    return prec_plan(prec).small


def plane_fmt(prec):
    """Plane format the plane-operand GEMMs of precision ``prec`` consume (and emit)."""
    return PLANES_F16 if (prec & 0xff) == PREC_F16X2 else PLANES_BF16


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous fp32, got {t.dtype} contiguous={t.is_contiguous()}")
    L.require_gpu(t)
    return t


class PackedWeight:
    """A Linear/Conv weight [N][K] pre-split into bf16 hi/lo parts, rows padded for the GEMM tiles."""

    def __init__(self, w2d, bias=None):
        w2d = _f32c(w2d.contiguous(), "weight")
        self.N, self.K = w2d.shape
        if self.K % 64:
            raise ValueError(f"K={self.K} must be a multiple of 64")
        self.N_pad = 64 if self.N <= 64 else (self.N + 127) // 128 * 128
        self.hi = torch.empty((self.N_pad, self.K), dtype=torch.int16, device=w2d.device)
        self.lo = torch.empty_like(self.hi)
        L.check(L.load().avi_pack_weight_split(w2d.data_ptr(), self.N, self.K, self.N_pad, self.hi.data_ptr(),
                                               self.lo.data_ptr(), L.stream_ptr()), "avi_pack_weight_split")
        self.bias = None if bias is None else _f32c(bias.contiguous(), "bias")

    @property
    def nbytes(self):
        return self.hi.numel() * 4

    def f16_plane(self):
        """[N_pad][K] fp16 (int16 storage): the single weight plane of the opt-in 2-term fp16 GEMM (AVI_PREC_F16X2).
        Built on first use from hi + lo (= the fp32 weight to 2^-16), once per model."""
        if getattr(self, "_f16", None) is None:
            w = self.hi.view(torch.bfloat16).float() + self.lo.view(torch.bfloat16).float()
            self._f16 = w.to(torch.float16).view(torch.int16).contiguous()
        return self._f16

    def planes_for(self, prec):
        """(Whi, Wlo) device pointers for a plane-operand GEMM of precision ``prec``."""
        if (prec & 0xff) == PREC_F16X2:
            p = self.f16_plane().data_ptr()
            return p, p
        return self.hi.data_ptr(), self.lo.data_ptr()


def gemm_raw(*, A=0, lda, Whi, Wlo, C_=0, ldc, M, N, K, bias=None, R=None, ldr=0, scale=None, shift=None, act=ACT_NONE,
             prec=PREC_BF16X3, batch=1, z_inner=1, sA=(0, 0), sW=(0, 0), sC=(0, 0), sB=(0, 0), sR=(0, 0),
             Ahi=0, Alo=0, Chi=0, Clo=0, ldw=0, cus=0, C16=0, sk_ws=None):
    """Direct access to ``avi_gemm``; pointers are ints (tensor.data_ptr() + byte offsets).  ``cus``: compute units the
    launch can count on (0 = all 256): the sampling pipeline passes what the sampler's resident workgroups leave, so that
    the plane-operand kernels pick tile shapes that fill whole rounds of the CUs that are actually free (AviGemm.cus)."""
    g = L.AviGemm()
    g.ldw = ldw
    g.cus = cus
    g.C16 = C16 or None
    if sk_ws is not None:           # stream-K workspace (fp32 tensor, zero-filled once by its owner): AviGemm.sk_ws
        g.sk_ws, g.sk_ws_floats = sk_ws.data_ptr(), sk_ws.numel()
    g.A, g.lda, g.sAo, g.sAi = A or None, lda, sA[0], sA[1]
    g.Ahi, g.Alo, g.Chi, g.Clo = Ahi or None, Alo or None, Chi or None, Clo or None
    g.Whi, g.Wlo, g.sWo, g.sWi = Whi, Wlo, sW[0], sW[1]
    g.C, g.ldc, g.sCo, g.sCi = C_ or None, ldc, sC[0], sC[1]
    g.bias, g.sBo, g.sBi = bias or None, sB[0], sB[1]
    g.R, g.ldr, g.sRo, g.sRi = R or None, ldr, sR[0], sR[1]
    g.scale, g.shift = scale or None, shift or None
    g.M, g.N, g.K = M, N, K
    g.batch, g.z_inner = batch, z_inner
    g.act, g.prec = act, prec
    L.check(L.load().avi_gemm(C.byref(g), L.stream_ptr()), "avi_gemm")


def linear(x, pw, out=None, act=ACT_NONE, residual=None, prec=PREC_BF16X3, scale=None, shift=None):
    """out[..., N] = affine(act(x[..., K] @ W^T + b)) + residual."""
    x = _f32c(x, "x")
    K = x.shape[-1]
    if K != pw.K:
        raise ValueError(f"linear: x has K={K}, weight has K={pw.K}")
    M = x.numel() // K
    if out is None:
        out = torch.empty(x.shape[:-1] + (pw.N,), dtype=torch.float32, device=x.device)
    _f32c(out, "out")
    if out.numel() != M * pw.N:
        raise ValueError("linear: bad out shape")
    if residual is not None:
        _f32c(residual, "residual")
        if residual.numel() != M * pw.N:
            raise ValueError("linear: bad residual shape")
    gemm_raw(A=x.data_ptr(), lda=K, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=out.data_ptr(), ldc=pw.N, M=M,
             N=pw.N, K=K, bias=L.ptr(pw.bias), R=L.ptr(residual), ldr=pw.N, scale=L.ptr(scale), shift=L.ptr(shift),
             act=act, prec=prec)
    return out


def conv1d_cl(x, pw, ksize, stride, out=None, act=ACT_NONE, prec=PREC_BF16X3, scale=None, shift=None,
              out_rows=None, out_row_stride=None, out_offset=0, in_row_offset=0, out_dtype=torch.float32):
    """Channels-last Conv1d without padding as an overlapping-row GEMM.

    x [B][Tin][Cin]; weight packed as [Cout][k*Cin] (tap-major); out [B][Tout][Cout] with
    Tout = (Tin-k)//stride+1.  ``out_rows``/``out_row_stride``/``out_offset`` let the caller
    interleave rows into a larger buffer and ``in_row_offset`` skips leading input rows
    (the two phases of a stride-2 ConvTranspose1d).
    """
    x = _f32c(x, "x")
    B, Tin, Cin = x.shape
    if pw.K != ksize * Cin:
        raise ValueError(f"conv1d_cl: weight K={pw.K} != k*Cin={ksize * Cin}")
    Tout = (Tin - in_row_offset - ksize) // stride + 1
    if Tout <= 0:
        raise ValueError("conv1d_cl: input shorter than the kernel")
    half = out_dtype == torch.float16           # the result stored as IEEE half by the GEMM's own epilogue (AviGemm.C16)
    if out is None:
        out = torch.empty((B, Tout, pw.N), dtype=out_dtype, device=x.device)
    if half:
        if out.dtype != torch.float16 or not out.is_contiguous():
            raise ValueError("conv1d_cl: out_dtype float16 needs a contiguous float16 out")
        L.require_gpu(out)
    else:
        _f32c(out, "out")
    rows = Tout if out_rows is None else out_rows
    if rows > Tout:
        raise ValueError("conv1d_cl: more output rows requested than the input provides")
    ldc = pw.N if out_row_stride is None else out_row_stride
    per_b = out.numel() // B
    if out_offset + (rows - 1) * ldc + pw.N > per_b:
        raise ValueError("conv1d_cl: output rows exceed the out buffer")
    c_args = dict(C16=out.data_ptr() + 2 * out_offset) if half else dict(C_=out.data_ptr() + 4 * out_offset)
    gemm_raw(A=x.data_ptr() + 4 * in_row_offset * Cin, lda=stride * Cin, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(),
             ldc=ldc, M=rows, N=pw.N, K=pw.K, bias=L.ptr(pw.bias), **c_args,
             scale=L.ptr(scale), shift=L.ptr(shift), act=act, prec=prec, batch=B, z_inner=1,
             sA=(Tin * Cin, 0), sC=(per_b, 0))
    return out


def audio_normalize(pcm, joint=False, eps=1e-7):
    """int16 or fp32 [B][N] -> zero-mean/unit-variance fp32 [B][N] (per clip, or jointly)."""
    L.require_gpu(pcm)
    if pcm.dtype not in (torch.int16, torch.float32) or not pcm.is_contiguous() or pcm.dim() != 2:
        raise ValueError("audio_normalize: expected contiguous int16/fp32 [B][N]")
    B, N = pcm.shape
    out = torch.empty((B, N), dtype=torch.float32, device=pcm.device)
    stats = torch.empty((2 * B,), dtype=torch.float64, device=pcm.device)
    L.check(L.load().avi_audio_normalize(pcm.data_ptr(), int(pcm.dtype == torch.int16), B, N, int(joint), eps,
                                         out.data_ptr(), stats.data_ptr(), L.stream_ptr()), "avi_audio_normalize")
    return out


def conv0_gn_gelu(x, w0, gamma, beta, eps=1e-5, out=None):
    x = _f32c(x, "x")
    B, N = x.shape
    if tuple(w0.shape) != (512, 10):
        raise ValueError("conv0: weight must be [512][10]")
    T0 = (N - 10) // 5 + 1
    if out is None:
        out = torch.empty((B, T0, 512), dtype=torch.float32, device=x.device)
    mom = torch.empty((65 * B * ((T0 + 511) // 512),), dtype=torch.float64, device=x.device)
    ss = torch.empty((1024 * B,), dtype=torch.float32, device=x.device)
    L.check(L.load().avi_conv0_gn_gelu(x.data_ptr(), B, N, _f32c(w0, "w0").data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), eps, out.data_ptr(), mom.data_ptr(), ss.data_ptr(),
                                       L.stream_ptr()), "avi_conv0_gn_gelu")
    return out


def interp_layernorm(x, Tout, gamma=None, beta=None, eps=1e-5):
    x = _f32c(x, "x")
    B, Tin, Cc = x.shape
    out = torch.empty((B, Tout, Cc), dtype=torch.float32, device=x.device)
    L.check(L.load().avi_interp_layernorm(x.data_ptr(), B, Tin, Cc, Tout, L.ptr(gamma), L.ptr(beta), eps,
                                          out.data_ptr(), L.stream_ptr()), "avi_interp_layernorm")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None, act=ACT_NONE, residual=None):
    """out = act(LN(x)) + residual."""
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    if residual is not None and _f32c(residual, "residual").numel() != x.numel():
        raise ValueError("layernorm: bad residual shape")
    L.check(L.load().avi_layernorm_act(x.data_ptr(), rows, Cc, L.ptr(gamma), L.ptr(beta), eps, act, L.ptr(residual),
                                       out.data_ptr(), L.stream_ptr()), "avi_layernorm_act")
    return out


def linear_ln_skinny(x, pw, gamma=None, beta=None, eps=1e-5, do_ln=True, act=ACT_NONE, residual=None,
                     prec=PREC_BF16X3, kslice=256):
    """out = act(LN(x @ W^T + b)) + residual for few row tiles and a long K (the aligner MLP at batch-size rows, the
    EMOTE squasher at B*T/8 rows).

    A 128-row GEMM tile would be three quarters padding and a 4096-long K loop on 32 workgroups is pure latency,
    so K is cut into slices that run as one batched launch (hundreds of workgroups stream the weight matrix at
    HBM speed); ``avi_splitk_epilogue`` folds the partial sums, adds the bias and applies LN/act/residual."""
    x = _f32c(x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    if K != pw.K:
        raise ValueError(f"linear_ln_skinny: x has K={K}, weight has K={pw.K}")
    while K % kslice:
        kslice //= 2
    if kslice < 64:
        raise ValueError("linear_ln_skinny: K must be a multiple of 64")
    nz = K // kslice
    N = pw.N
    parts = torch.empty((nz, M, N), dtype=torch.float32, device=x.device)
    gemm_raw(A=x.data_ptr(), lda=K, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=parts.data_ptr(), ldc=N, M=M, N=N,
             K=kslice, prec=prec, batch=nz, sA=(kslice, 0), sW=(kslice, 0), sC=(M * N, 0), ldw=K)
    out = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    if residual is not None and _f32c(residual, "residual").numel() != out.numel():
        raise ValueError("linear_ln_skinny: bad residual shape")
    L.check(L.load().avi_splitk_epilogue(parts.data_ptr(), nz, M * N, M, N, L.ptr(pw.bias), L.ptr(gamma), L.ptr(beta),
                                         eps, int(do_ln), act, L.ptr(residual), out.data_ptr(), L.stream_ptr()),
            "avi_splitk_epilogue")
    return out


def posconv_gelu_residual(x, pw, bias, groups, taps, rows_per_group):
    """x + gelu(grouped Conv1d(x, k=taps, padding=taps/2)[:-1] + bias) for x (B, T, C): wav2vec2's positional embedding.
    pw: PackedWeight of [groups * rows_per_group][taps * C/groups] (tap-major), rows beyond C/groups per group zero."""
    x = _f32c(x, "x")
    B, T, Cc = x.shape
    out = torch.empty_like(x)
    L.check(L.load().avi_posconv_gelu_residual(x.data_ptr(), B, T, Cc, groups, taps, pw.hi.data_ptr(), pw.lo.data_ptr(),
                                               rows_per_group, bias.data_ptr(), out.data_ptr(), L.stream_ptr()),
            "avi_posconv_gelu_residual")
    return out


def group_pad_pack(h, G, pad):
    h = _f32c(h, "h")
    B, T, Cc = h.shape
    Cg = Cc // G
    out = torch.empty((B, G, T + 2 * pad, Cg), dtype=torch.float32, device=h.device)
    L.check(L.load().avi_group_pad_pack(h.data_ptr(), B, T, G, Cg, pad, out.data_ptr(), L.stream_ptr()),
            "avi_group_pad_pack")
    return out


def pad_repeat(x, rep=1, padL=0, padR=0, mode=0):
    x = _f32c(x, "x")
    B, T, Cc = x.shape
    out = torch.empty((B, padL + T * rep + padR, Cc), dtype=torch.float32, device=x.device)
    L.check(L.load().avi_pad_repeat(x.data_ptr(), B, T, Cc, rep, padL, padR, mode, out.data_ptr(), L.stream_ptr()),
            "avi_pad_repeat")
    return out


def add_rowbcast(x, add):
    x = _f32c(x, "x")
    add = _f32c(add, "add")
    B, T, Cc = x.shape
    if add.numel() != B * Cc:
        raise ValueError("add_rowbcast: add must be [B][C]")
    out = torch.empty_like(x)
    L.check(L.load().avi_add_rowbcast(x.data_ptr(), add.data_ptr(), B, T, Cc, out.data_ptr(), L.stream_ptr()),
            "avi_add_rowbcast")
    return out


def embed_tokens(ids, table, pos, bad_ids=None):
    """(B, T) int64 token ids -> table[ids] + pos[:T]  (CLIPTextEmbeddings)."""
    L.require_gpu(ids, table, pos)
    if ids.dtype != torch.int64 or ids.dim() != 2 or not ids.is_contiguous():
        raise ValueError("embed_tokens: ids must be a contiguous (B, T) int64 tensor")
    B, T = ids.shape
    vocab, Cc = table.shape
    if pos.shape[0] < T or pos.shape[1] != Cc:
        raise ValueError("embed_tokens: position table does not cover the sequence")
    out = torch.empty((B, T, Cc), dtype=torch.float32, device=ids.device)
    L.check(L.load().avi_embed_tokens(ids.data_ptr(), _f32c(table, "table").data_ptr(), _f32c(pos, "pos").data_ptr(),
                                      B, T, Cc, vocab, out.data_ptr(), L.ptr(bad_ids), L.stream_ptr()),
            "avi_embed_tokens")
    return out


def mean_tokens(x):
    """(B, T, C) -> (B, C) mean over T."""
    x = _f32c(x, "x")
    B, T, Cc = x.shape
    out = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    L.check(L.load().avi_mean_tokens(x.data_ptr(), B, T, Cc, out.data_ptr(), L.stream_ptr()), "avi_mean_tokens")
    return out


def attention(q, k, v, H, D, ldq, ldk, Tq, Tk, B, scale, bias_mode=0, slopes=None, period=1, out=None):
    """q/k/v are base tensors (possibly views into one packed QKV buffer); ldq/ldk are row strides."""
    L.require_gpu(q, k, v)
    if out is None:
        out = torch.empty((B, Tq, H * D), dtype=torch.float32, device=q.device)
    L.check(L.load().avi_attention(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Tq, Tk, D, ldq,
                                   ldk, H * D, scale, bias_mode, L.ptr(slopes), period, L.stream_ptr()),
            "avi_attention")
    return out


def attention_d64(qkv, H, scale, out=None):
    """Unbiased head-dim-64 attention over a packed (B, T, 3*H*64) projection on the matrix cores."""
    qkv = _f32c(qkv, "qkv")
    B, T, ld = qkv.shape
    if ld != 3 * H * 64:
        raise ValueError(f"attention_d64: last dim {ld} != 3*H*64")
    if out is None:
        out = torch.empty((B, T, H * 64), dtype=torch.float32, device=qkv.device)
    Tp = (T + 63) // 64 * 64
    scratch = torch.empty((6 * B * H * Tp * 64,), dtype=torch.int16, device=qkv.device)
    L.check(L.load().avi_attention_d64(qkv.data_ptr(), B, H, T, ld, scale, out.data_ptr(), H * 64, scratch.data_ptr(),
                                       L.stream_ptr()), "avi_attention_d64")
    return out


def attention_d64_planes(qkv, H, scale, bias_mode=0, slopes=None, period=1, fmt=PLANES_BF16):
    """Head-dim-64 attention whose result is written as split planes (operand of the out_proj GEMM)."""
    qkv = _f32c(qkv, "qkv")
    B, T, ld = qkv.shape
    if ld != 3 * H * 64:
        raise ValueError(f"attention_d64_planes: last dim {ld} != 3*H*64")
    out = Planes((B, T, H * 64), qkv.device, fmt)
    if bias_mode == 0:
        L.check(L.load().avi_attention_d64_planes(qkv.data_ptr(), B, H, T, ld, scale, None, out.hi.data_ptr(),
                                                  out.lo.data_ptr(), H * 64, fmt, L.stream_ptr()),
                "avi_attention_d64_planes")
    else:
        L.require_gpu(slopes)
        L.check(L.load().avi_attention_d64_planes_biased(qkv.data_ptr(), B, H, T, ld, scale, bias_mode, L.ptr(slopes),
                                                         period, None, out.hi.data_ptr(), out.lo.data_ptr(), H * 64,
                                                         fmt, L.stream_ptr()), "avi_attention_d64_planes_biased")
    return out


# ------------------------------------------------------------------ split-plane activations (x = hi + lo, bf16 each)
class Planes:
    """An activation stored as two 16-bit planes (int16 storage), x = hi + lo: bf16 (PLANES_BF16) or fp16 (PLANES_F16),
    the operand format of the LDS-DMA GEMMs."""

    def __init__(self, shape, device, fmt=PLANES_BF16):
        self.shape = tuple(shape)
        self.fmt = fmt
        self.hi = torch.empty(self.shape, dtype=torch.int16, device=device)
        self.lo = torch.empty(self.shape, dtype=torch.int16, device=device)

    def float(self):
        dt = torch.float16 if self.fmt == PLANES_F16 else torch.bfloat16
        return self.hi.view(dt).float() + self.lo.view(dt).float()


def conv0_gn_gelu_planes(x, w0, gamma, beta, eps=1e-5, fmt=PLANES_BF16):
    x = _f32c(x, "x")
    B, N = x.shape
    T0 = (N - 10) // 5 + 1
    out = Planes((B, T0, 512), x.device, fmt)
    mom = torch.empty((65 * B * ((T0 + 511) // 512),), dtype=torch.float64, device=x.device)
    ss = torch.empty((1024 * B,), dtype=torch.float32, device=x.device)
    L.check(L.load().avi_conv0_gn_gelu_planes(x.data_ptr(), B, N, _f32c(w0, "w0").data_ptr(), gamma.data_ptr(),
                                              beta.data_ptr(), eps, out.hi.data_ptr(), out.lo.data_ptr(), mom.data_ptr(),
                                              ss.data_ptr(), fmt, L.stream_ptr()), "avi_conv0_gn_gelu_planes")
    return out


def conv1d_cl_planes(xp, pw, ksize, stride, act=ACT_NONE, prec=PREC_BF16X3, out_planes=True, cus=0):
    """Channels-last Conv1d (no padding) on split-plane input: the overlapping-row GEMM on the LDS-DMA kernel."""
    B, Tin, Cin = xp.shape
    if pw.K != ksize * Cin or pw.N <= 64:
        raise ValueError("conv1d_cl_planes: weight does not match the input / N too narrow for the LDS-DMA kernel")
    Tout = (Tin - ksize) // stride + 1
    dev = xp.hi.device
    if xp.fmt != plane_fmt(prec):
        raise ValueError("conv1d_cl_planes: the input planes are not in the format this precision consumes")
    if out_planes:
        out = Planes((B, Tout, pw.N), dev, plane_fmt(prec))
        c_args = dict(Chi=out.hi.data_ptr(), Clo=out.lo.data_ptr())
    else:
        out = torch.empty((B, Tout, pw.N), dtype=torch.float32, device=dev)
        c_args = dict(C_=out.data_ptr())
    whi, wlo = pw.planes_for(prec)
    gemm_raw(Ahi=xp.hi.data_ptr(), Alo=xp.lo.data_ptr(), lda=stride * Cin, Whi=whi, Wlo=wlo,
             ldc=pw.N, M=Tout, N=pw.N, K=pw.K, bias=L.ptr(pw.bias), act=act, prec=prec, batch=B, sA=(Tin * Cin, 0),
             sC=(Tout * pw.N, 0), cus=cus, **c_args)
    return out


def stream_k_workspace(M, N, device):
    """Zero-filled workspace that lets the 128-row plane-operand GEMM cut an (M x N) problem into equal K shares when its
    tiles do not fill the last round of the CUs it may count on (AviGemm.sk_ws; one launch at a time uses it)."""
    tiles = -(-M // 128) * -(-N // 192)            # 128 x 192 tiles are the smaller shape: more tiles, smaller slots
    return torch.zeros(tiles * 4 * 128 * 256 + tiles + 64, dtype=torch.float32, device=device)


def linear_planes(xp, pw, act=ACT_NONE, residual=None, prec=PREC_BF16X3, out=None, out_planes=False, cus=0, sk_ws=None):
    """out[..., N] = act(x @ W^T + b) + residual with x given as split planes; the result is fp32, or split planes
    too (``out_planes=True``: a ``Planes`` for the next GEMM)."""
    K = xp.shape[-1]
    if K != pw.K or pw.N <= 64:
        raise ValueError("linear_planes: shape mismatch / N too narrow for the LDS-DMA kernel")
    M = 1
    for d in xp.shape[:-1]:
        M *= d
    if residual is not None and _f32c(residual, "residual").numel() != M * pw.N:
        raise ValueError("linear_planes: bad residual shape")
    if xp.fmt != plane_fmt(prec):
        raise ValueError("linear_planes: the input planes are not in the format this precision consumes")
    if out_planes:
        out = Planes(tuple(xp.shape[:-1]) + (pw.N,), xp.hi.device, plane_fmt(prec))
        c_args = dict(Chi=out.hi.data_ptr(), Clo=out.lo.data_ptr())
    else:
        if out is None:
            out = torch.empty(tuple(xp.shape[:-1]) + (pw.N,), dtype=torch.float32, device=xp.hi.device)
        c_args = dict(C_=out.data_ptr())
    whi, wlo = pw.planes_for(prec)
    gemm_raw(Ahi=xp.hi.data_ptr(), Alo=xp.lo.data_ptr(), lda=K, Whi=whi, Wlo=wlo,
             ldc=pw.N, M=M, N=pw.N, K=K, bias=L.ptr(pw.bias), R=L.ptr(residual), ldr=pw.N, act=act, prec=prec, cus=cus,
             sk_ws=sk_ws, **c_args)
    return out


def linear_planes_splitk(xp, pw, nsplit, residual=None, prec=PREC_BF16X3, cus=0):
    """out = x @ W^T + b + residual for plane input with K cut into ``nsplit`` slices: one batched launch of the plane-operand
    GEMM over the slices (partial sums in fp32) + ``avi_splitk_epilogue`` (adds them in slice order, bias, residual).  For
    a few hundred rows and a narrow N the data-parallel launch has too few 128-row tiles for 256 CUs (2464 x 768: 80 tiles);
    slices of K multiply the tile count.  K / nsplit must itself tile for the ping-pong kernels (a multiple of 96)."""
    K = xp.shape[-1]
    if K != pw.K or pw.N <= 64 or nsplit < 2 or K % nsplit or (K // nsplit) % 96:
        raise ValueError("linear_planes_splitk: K / nsplit must be a multiple of 96 (and N > 64)")
    if (prec & 0xff) != PREC_BF16X3 or xp.fmt != PLANES_BF16:
        raise ValueError("linear_planes_splitk: 3-term bf16 planes only")
    M = 1
    for d in xp.shape[:-1]:
        M *= d
    N, ks = pw.N, K // nsplit
    dev = xp.hi.device
    if residual is not None and _f32c(residual, "residual").numel() != M * N:
        raise ValueError("linear_planes_splitk: bad residual shape")
    parts = torch.empty((nsplit, M, N), dtype=torch.float32, device=dev)
    gemm_raw(Ahi=xp.hi.data_ptr(), Alo=xp.lo.data_ptr(), lda=K, Whi=pw.hi.data_ptr(), Wlo=pw.lo.data_ptr(), C_=parts.data_ptr(),
             ldc=N, M=M, N=N, K=ks, prec=prec, batch=nsplit, sA=(ks, 0), sW=(ks, 0), sC=(M * N, 0), ldw=K, cus=cus)
    out = torch.empty(tuple(xp.shape[:-1]) + (N,), dtype=torch.float32, device=dev)
    L.check(L.load().avi_splitk_epilogue(parts.data_ptr(), nsplit, M * N, M, N, L.ptr(pw.bias), None, None, 1e-5, 0, ACT_NONE,
                                         L.ptr(residual), out.data_ptr(), L.stream_ptr()), "avi_splitk_epilogue")
    return out


def layernorm_planes(x, gamma, beta, eps=1e-5, out=None, want_f32=True, fmt=PLANES_BF16):
    """LayerNorm emitting the result as fp32 (optional) and as split planes for the next GEMM."""
    x = _f32c(x, "x")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if want_f32 and out is None:
        out = torch.empty_like(x)
    pl = Planes(x.shape, x.device, fmt)
    L.check(L.load().avi_layernorm_planes(x.data_ptr(), rows, Cc, L.ptr(gamma), L.ptr(beta), eps,
                                          out.data_ptr() if want_f32 else None, pl.hi.data_ptr(), pl.lo.data_ptr(),
                                          fmt, L.stream_ptr()), "avi_layernorm_planes")
    return (out if want_f32 else None), pl


def interp_layernorm_planes(xp, Tout, gamma=None, beta=None, eps=1e-5):
    B, Tin, Cc = xp.shape
    out = torch.empty((B, Tout, Cc), dtype=torch.float32, device=xp.hi.device)
    L.check(L.load().avi_interp_layernorm_planes(xp.hi.data_ptr(), xp.lo.data_ptr(), B, Tin, Cc, Tout, L.ptr(gamma),
                                                 L.ptr(beta), eps, out.data_ptr(), L.stream_ptr()),
            "avi_interp_layernorm_planes")
    return out
