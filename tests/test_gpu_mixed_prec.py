"""GPU parity of the MIXED precision schedules (ops.PrecPlan) against the CPU oracle, audio -> coefficients.

``mixed``: conv layers 1-6 (half of the path's FLOPs) on the 2-term fp16 GEMM (fp16 hi/lo activation planes x ONE fp16
weight plane), the transformer projections on the 3-term bf16 split, the sampler as in the default mode.  Gate 3e-4 max-abs
on the un-normalised coefficients: three times tighter than north_star's 1e-3 (the uniform f16x2 mode lands at 6-8e-4, the
default 3-term mode at 2e-5).  ``mixed_ffn`` adds the two feed-forward matrices to the 2-term group; it is held to
north_star's gate and its measured error is printed.  The reference itself runs fp16 autocast
(train_diffusion_prior.py:434)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

GATE = {"mixed": 3e-4, "mixed_ffn": 1e-3}


def _coeff_err(out, ref):
    return max((out["predicted_exp"].cpu() - ref["predicted_exp"]).abs().max().item(),
               (out["predicted_jaw"].cpu() - ref["predicted_jaw"]).abs().max().item())


def _bandlimited(B, N, g):
    x = torch.randn(B, N, generator=g)
    X = torch.fft.rfft(x)
    X[:, int(4000 / 8000 * (X.shape[1] - 1)):] = 0                       # band-limited noise, int16 RMS 3000
    x = torch.fft.irfft(X, n=N)
    return (x / x.pow(2).mean(-1, keepdim=True).sqrt() * 3000.0).clamp(-32768, 32767).to(torch.int16)


@pytest.fixture(scope="module")
def weights():
    from avi_talking_amd import weights as W
    return W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)


@pytest.mark.parametrize("plan", ["mixed", "mixed_ffn"])
@pytest.mark.parametrize("case", ["config0_1x4s", "config1_sub_4x10s"])
def test_mixed_pipeline_vs_oracle(gpu, weights, plan, case):
    """BASELINE configs[0] (one 4 s clip) and a configs[1] sub-batch (4 clips x 10 s) through the whole sampling path."""
    from avi_talking_amd.host.pipeline import SamplingPipeline
    from oracle import emote as OE, prior as OP, wav2vec2 as OW
    wa, wh, wp = weights
    B, secs = (1, 4) if case.startswith("config0") else (4, 10)
    T = secs * 25
    g = torch.Generator().manual_seed(1234)
    pcm = _bandlimited(B, T * 640, g)
    voxel, noise = torch.randn(B, 768, generator=g), torch.randn(101, B, 1, 128, generator=g)
    pipe = SamplingPipeline(wa, wh, wp, device=gpu, prec=plan)
    assert pipe.plan.name == plan and pipe.talking_head.audio_model.plan.conv == 2          # AVI_PREC_F16X2
    out = pipe.run(pcm.to(gpu), voxel.to(gpu), noise.to(gpu))
    feat = OW.forward(wa, OW.normalize_audio(pcm, joint=False), frame_num=T)
    te, _ = OP.brain_network(wp, voxel)
    ref = OE.forward(wh, feat, OP.p_sample_loop(wp, te.view(B, 1, 128), noise))
    e = _coeff_err(out, ref)
    hid = (pipe.talking_head.audio_model(OW.normalize_audio(pcm, joint=False).to(gpu), frame_num=T).last_hidden_state.cpu()
           - feat).abs().max().item()
    print(f"{plan} {case}: max-abs coefficient err {e:.2e} (gate {GATE[plan]:.0e}), wav2vec2 hidden state {hid:.2e}")
    assert out["predicted_exp"].shape == (B, T, 50) and e < GATE[plan]


@pytest.mark.parametrize("plan", ["mixed", "mixed_ffn"])
def test_mixed_60s_utterance(gpu, weights, plan):
    """BASELINE configs[4] length: T = 1500 frames, audio -> EMOTE/FLINT coefficients with an external style."""
    from avi_talking_amd.host.talking_head import TalkingHeadWrapper
    from oracle import emote as OE, wav2vec2 as OW
    wa, wh, _ = weights
    T = 1500
    g = torch.Generator().manual_seed(61)
    pcm = (torch.randn(1, T * 640, generator=g) * 3000).to(torch.int16)
    style = torch.randn(1, 1, 128, generator=g) * 0.5
    th = TalkingHeadWrapper(wa, wh, device=gpu, prec=plan)
    out = th({"raw_audio": pcm.view(1, T, 640), "samplerate": [16000]}, style_emb=style.to(gpu), is_external_style_emb=True)
    ref = OE.forward(wh, OW.forward(wa, OW.normalize_audio(pcm, joint=True), frame_num=T), style)
    e = _coeff_err(out, ref)
    print(f"{plan} T=1500: max-abs coefficient err {e:.2e} (gate {GATE[plan]:.0e})")
    assert e < GATE[plan]
