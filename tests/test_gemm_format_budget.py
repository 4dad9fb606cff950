"""Error budget of GEMM operand formats, measured on the CPU oracle (no GPU): what the coefficients pay for every way of
spending fewer matrix-core cycles per product than the 3-term bf16 split.

The audio path's GEMMs (conv layers 1-6, the encoder's q/k/v/out/ffn projections) are re-evaluated inside the fp32 oracle
with their operands decomposed the way a kernel would hold them:

  f16x2        x = x_hi + x_lo (fp16 each), w = ONE fp16 plane:  y = x_hi.w + x_lo.w      2 MFMA per product (shipped: --prec f16x2)
  mixed        f16x2 on the conv layers only                                              (shipped: --prec mixed, the default)
  f16+fp8x     main term x_hi.w_hi on fp16; BOTH cross terms x_hi.w_lo + x_lo.w_hi on block-scaled fp8 (e4m3, one E8M0
               scale per 32 k) - `v_mfma_scale_f32_16x16x128_f8f6f4` runs 4x the K at 2x the cycles: 2 MFMA-equivalents
  f16+fp6x     the same with e2m3 operands: 1.7 MFMA-equivalents (measured issue rate, scripts/probe/mfma_scale_probe.hip)
  f16+fp4x     e2m1: 1.5 MFMA-equivalents

This is the study the round-2 review asked for BEFORE any fp8 cross-term kernel is built: it pins (a) that the
emulation reproduces the errors MEASURED on the GPU for the two shipped formats, and (b) what the block-scaled cross terms
would buy.  Nothing here is product code."""
import pytest
import torch
import torch.nn.functional as TF

from avi_talking_amd import weights as W
from oracle import emote as OE
from oracle import wav2vec2 as OW

FORMATS = {"e4m3": (8, -6, 3), "e2m3": (2, 0, 3), "e2m1": (2, 0, 1)}      # (emax, emin of the normals, mantissa bits)


def quant_block(v, fmt, block=32):
    """Block quantisation along the last axis: one power-of-two (E8M0) scale per `block` elements, the smallest that brings the
    block's largest magnitude inside the format (no saturation of the elements that matter most), elements rounded to the
    minifloat `fmt` (round to nearest, subnormals kept)."""
    emax, emin, mbits = FORMATS[fmt]
    shp = v.shape
    K = shp[-1]
    pad = (-K) % block
    x = TF.pad(v, (0, pad)).reshape(-1, block).double()
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    top = (2.0 - 2.0 ** -mbits) * 2.0 ** emax if fmt != "e4m3" else 448.0      # e4m3fn gives its top code to NaN
    scale = torch.exp2(torch.ceil(torch.log2(amax / top)))
    u = x / scale
    e = torch.floor(torch.log2(u.abs().clamp_min(1e-300))).clamp_min(emin)
    step = torch.exp2(e - mbits)
    q = torch.round(u / step) * step
    q = q.clamp(-top, top)
    return (q * scale).reshape(*shp[:-1], K + pad)[..., :K].to(v.dtype)


def emulated_matmul(A, Wt, scheme):
    """A (M, K) . Wt (N, K)^T with the operands held as `scheme` says; fp64 accumulation (the MFMA's fp32 accumulation is not
    what is being studied)."""
    A, Wt = A.double(), Wt.double()
    if scheme == "exact":
        return A @ Wt.t()
    Ah = A.to(torch.float16).double()
    Wh = Wt.to(torch.float16).double()
    Al, Wl = A - Ah, Wt - Wh
    if scheme == "f16x2":
        return A @ Wh.t()
    fmt = {"f16+fp8x": "e4m3", "f16+fp6x": "e2m3", "f16+fp4x": "e2m1"}[scheme]
    q = lambda t: quant_block(t, fmt)
    return Ah @ Wh.t() + q(Ah) @ q(Wl).t() + q(Al) @ q(Wh).t()


class _Shim:
    """torch.nn.functional for oracle/wav2vec2.py with `linear` (encoder projections) and the un-padded, un-grouped
    `conv1d` (conv layers 1-6) routed through the emulation; everything else untouched."""

    def __init__(self, conv_scheme, lin_scheme):
        self.conv_scheme, self.lin_scheme = conv_scheme, lin_scheme

    def __getattr__(self, name):
        return getattr(TF, name)

    def linear(self, x, w, b=None):
        if self.lin_scheme == "exact" or w.shape[1] not in (768, 3072) or w.shape[0] not in (768, 3072):
            return TF.linear(x, w, b)                     # feature projection (512 -> 768): a small fp32-operand launch
        y = emulated_matmul(x.reshape(-1, x.shape[-1]), w, self.lin_scheme).to(x.dtype).reshape(*x.shape[:-1], w.shape[0])
        return y if b is None else y + b

    def conv1d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if self.conv_scheme == "exact" or groups != 1 or padding != 0 or w.shape[1] != 512:
            return TF.conv1d(x, w, b, stride, padding, dilation, groups)   # conv layer 0, positional conv
        Bn, C, L = x.shape
        k = w.shape[2]
        cols = x.unfold(2, k, stride)                                        # (B, C, Lout, k)
        A = cols.permute(0, 2, 3, 1).reshape(-1, k * C)                      # tap-major rows, as the kernel's operand
        Wt = w.permute(0, 2, 1).reshape(w.shape[0], k * C)
        y = emulated_matmul(A, Wt, self.conv_scheme).to(x.dtype).reshape(Bn, -1, w.shape[0]).transpose(1, 2)
        return y if b is None else y + b[None, :, None]


def _coefficients(wa, wh, x, style, conv_scheme, lin_scheme, monkeypatch):
    monkeypatch.setattr(OW, "F", _Shim(conv_scheme, lin_scheme))
    with torch.no_grad():
        feat = OW.forward(wa, x, frame_num=x.shape[1] // 640)
        out = OE.forward(wh, feat, style)
    monkeypatch.undo()
    return torch.cat([out["predicted_exp"], out["predicted_jaw"]], -1), feat


def test_minifloat_quantiser_matches_torch_e4m3():
    v = torch.randn(64, 96, generator=torch.Generator().manual_seed(1)) * 3
    mine = quant_block(v, "e4m3")
    # the same block scale applied by hand, elements through torch's own e4m3 rounding
    x = v.reshape(-1, 32).double()
    scale = torch.exp2(torch.ceil(torch.log2(x.abs().amax(-1, keepdim=True) / 448.0)))
    ref = ((x / scale).float().to(torch.float8_e4m3fn).double() * scale).reshape(v.shape).float()
    assert torch.equal(mine, ref)


@pytest.mark.timeout(900)
def test_operand_format_budget(monkeypatch):
    torch.set_num_threads(8)
    wa, wh = W.make_wav2vec2_weights(0), W.make_emote_weights(1)
    g = torch.Generator().manual_seed(1234)
    x = OW.normalize_audio((torch.randn(1, 50 * 640, generator=g) * 3000).to(torch.int16))
    style = torch.randn(1, 1, 128, generator=g) * 0.5
    ref, feat0 = _coefficients(wa, wh, x, style, "exact", "exact", monkeypatch)
    res = {}
    for name, cs, ls in (("mixed (conv f16x2)", "f16x2", "exact"), ("f16x2 everywhere", "f16x2", "f16x2"),
                         ("f16+fp8x everywhere", "f16+fp8x", "f16+fp8x"), ("f16+fp6x everywhere", "f16+fp6x", "f16+fp6x"),
                         ("f16+fp4x everywhere", "f16+fp4x", "f16+fp4x")):
        c, feat = _coefficients(wa, wh, x, style, cs, ls, monkeypatch)
        res[name] = ((c - ref).abs().max().item(), (feat - feat0).abs().max().item())
        print(f"{name:22s}: coefficients {res[name][0]:.2e}, wav2vec2 hidden state {res[name][1]:.2e}")
    # (a) the emulation reproduces what the GPU measures for the shipped formats (tests/test_gpu_mixed_prec.py: mixed
    #     1.9-2.5e-4; tests/test_gpu_emote.py: f16x2 6-8e-4) to within the spread between clips
    assert 0.8e-4 < res["mixed (conv f16x2)"][0] < 4e-4
    assert 2.5e-4 < res["f16x2 everywhere"][0] < 1.2e-3
    # (b) block-scaled cross terms: e4m3 and e2m3 (three mantissa bits each) give the 3-term split's accuracy at 2 / 1.5
    #     MFMA-equivalents on EVERY plane-operand GEMM; e2m1 (one mantissa bit) still beats the shipped mixed plan
    assert res["f16+fp8x everywhere"][0] < 6e-5 and res["f16+fp6x everywhere"][0] < 6e-5
    assert res["f16+fp4x everywhere"][0] < res["mixed (conv f16x2)"][0] < res["f16x2 everywhere"][0]
    assert max(res["f16+fp8x everywhere"][0], res["f16+fp6x everywhere"][0]) < res["f16+fp4x everywhere"][0]
