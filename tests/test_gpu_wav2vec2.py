"""End-to-end parity of the HIP wav2vec2 encoder against the CPU oracle (same seeded weights/audio)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,frame_num", [(2, 32000, None), (1, 64000, 100)])
def test_wav2vec2_parity(gpu, B, N, frame_num):
    from avi_talking_amd.weights import make_wav2vec2_weights
    from avi_talking_amd.host.wav2vec import Wav2Vec2Model
    from avi_talking_amd import ops
    from oracle import wav2vec2 as O
    w = make_wav2vec2_weights(0)
    x = torch.randn(B, N, generator=torch.Generator().manual_seed(5))
    ref = O.forward(w, x, frame_num=frame_num, return_intermediates=True)
    model = Wav2Vec2Model(w, device=gpu, prec=ops.PREC_BF16X3)
    out = model(x.to(gpu), "vocaset", frame_num=frame_num)
    feats = out.extract_features.cpu()
    e_conv = (feats - ref["conv"].transpose(1, 2)).abs().max().item()
    e_out = (out.last_hidden_state.cpu() - ref["last_hidden_state"]).abs().max().item()
    print(f"bf16x3: conv err {e_conv:.3e}, last_hidden_state err {e_out:.3e}")
    assert out.last_hidden_state.shape == ref["last_hidden_state"].shape
    assert e_conv < 1e-4
    assert e_out < 1e-3          # north_star tolerance (fp32-grade path)
    fast = Wav2Vec2Model(w, device=gpu, prec=ops.PREC_BF16)(x.to(gpu), "vocaset", frame_num=frame_num)
    e_fast = (fast.last_hidden_state.cpu() - ref["last_hidden_state"]).abs().max().item()
    print(f"bf16  : last_hidden_state err {e_fast:.3e}")
    assert e_fast < 0.15         # single-pass bf16: ~4e-3 relative per GEMM through 20 layers


@pytest.mark.parametrize("conv_planes,tf_planes", [("0", "0"), ("1", "0"), ("0", "1")])
def test_wav2vec2_activation_formats(gpu, monkeypatch, conv_planes, tf_planes):
    """The non-default activation formats (fp32 conv stack / split-plane transformer) give the same result."""
    from avi_talking_amd.weights import make_wav2vec2_weights
    from avi_talking_amd.host.wav2vec import Wav2Vec2Model
    from oracle import wav2vec2 as O
    monkeypatch.setenv("AVI_W2V_PLANES", conv_planes)
    monkeypatch.setenv("AVI_W2V_TF_PLANES", tf_planes)
    w = make_wav2vec2_weights(0)
    x = torch.randn(2, 32000, generator=torch.Generator().manual_seed(5))
    ref = O.forward(w, x)
    model = Wav2Vec2Model(w, device=gpu, prec="bf16x3")       # the kernel paths under test, at the all-3-term plan's accuracy
    assert model.use_planes == (conv_planes == "1") and model.use_planes_tf == (tf_planes == "1")
    out = model(x.to(gpu), "vocaset").last_hidden_state.cpu()
    err = (out - ref).abs().max().item()
    print(f"conv planes={conv_planes} transformer planes={tf_planes}: last_hidden_state err {err:.3e}")
    assert err < 1e-3


def test_wav2vec2_two_term_fp16_mode(gpu):
    """The OPT-IN 2-term fp16 mode (AVI_PREC_F16X2: fp16 hi/lo activation planes x ONE fp16 weight plane, two MFMAs per
    product; the reference itself runs fp16 autocast) against the fp32 oracle: hidden state within 2e-3, an order of
    magnitude better than single-pass bf16 and 50x worse than the default 3-term split (which stays the headline)."""
    from avi_talking_amd.weights import make_wav2vec2_weights
    from avi_talking_amd.host.wav2vec import Wav2Vec2Model
    from avi_talking_amd import ops
    from oracle import wav2vec2 as O
    w = make_wav2vec2_weights(0)
    x = torch.randn(2, 32000, generator=torch.Generator().manual_seed(5))
    ref = O.forward(w, x, return_intermediates=True)
    out = Wav2Vec2Model(w, device=gpu, prec=ops.PREC_F16X2)(x.to(gpu), "vocaset")
    e_conv = (out.extract_features.cpu() - ref["conv"].transpose(1, 2)).abs().max().item()
    e_out = (out.last_hidden_state.cpu() - ref["last_hidden_state"]).abs().max().item()
    print(f"f16x2: conv err {e_conv:.3e}, last_hidden_state err {e_out:.3e}")
    assert e_conv < 2e-3 and e_out < 5e-3
