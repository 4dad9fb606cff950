"""GPU: long-form utterance (BASELINE config 5: 60 s, T = 1500 frames).  The reference has no behaviour here for
its mask-limited modules (EMOTE mask 1200 frames, FaceFormer 600: SURVEY.md section 5), but the audio encoder and the
mask-free EMOTE/FLINT head (alibi is computed analytically here) are well defined: compare with the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_60s_utterance_matches_oracle(gpu):
    from avi_talking_amd.weights import make_emote_weights, make_wav2vec2_weights
    from avi_talking_amd.host.talking_head import TalkingHeadWrapper
    from oracle import emote as OE, wav2vec2 as OW
    wa, wh = make_wav2vec2_weights(0), make_emote_weights(1)
    T = 1500
    g = torch.Generator().manual_seed(61)
    pcm = (torch.randn(1, T * 640, generator=g) * 3000).to(torch.int16)
    style = torch.randn(1, 1, 128, generator=g) * 0.5
    th = TalkingHeadWrapper(wa, wh, device=gpu)
    out = th({"raw_audio": pcm.view(1, T, 640), "samplerate": [16000]}, style_emb=style.to(gpu),
             is_external_style_emb=True)
    feat = OW.forward(wa, OW.normalize_audio(pcm, joint=True), frame_num=T)
    ref = OE.forward(wh, feat, style)
    e = max((out["predicted_exp"].cpu() - ref["predicted_exp"]).abs().max().item(),
            (out["predicted_jaw"].cpu() - ref["predicted_jaw"]).abs().max().item())
    print(f"T=1500 coefficient err {e:.2e}")
    assert out["predicted_exp"].shape == (1, T, 50)
    assert e < 1e-3


def test_faceformer_rejects_sequences_beyond_reference_tables(gpu):
    """Without an explicit chunk the reference's limit stands (its predict() fails beyond 600 frames)."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    ff = Faceformer(make_faceformer_weights(2, feature_dim=64), device=gpu)
    with pytest.raises(ValueError):
        ff.decode(torch.zeros(1, 601, 64, device=gpu))
    with pytest.raises(ValueError):
        ff.decode(torch.zeros(1, 700, 64, device=gpu), chunk=100)        # not a multiple of the period (30)


@pytest.mark.parametrize("steps,D", [("0", 64), ("1", 64), ("1", 256)])
def test_faceformer_chunked_causal_60s(gpu, monkeypatch, steps, D):
    """BASELINE config 5: T = 1500 frames through the FaceFormer decoder with the chunked-causal window
    (include/avi_talking.h avi_faceformer_decode_chunked; chunk = 600 = the reference's table length) against the
    oracle's definition of the same window, on both device paths."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF
    monkeypatch.setenv("AVI_FF_STEPS", steps)
    T = 1500
    w = make_faceformer_weights(2, feature_dim=D)
    hs = torch.randn(2, T, D, generator=torch.Generator().manual_seed(71))
    ref = OF.predict_cached(w, hs, 30, chunk=600)
    ff = Faceformer(w, period=30, device=gpu)
    out = ff.decode(hs.to(gpu), chunk=600).cpu()
    err = (out - ref).abs().max().item()
    print(f"steps={steps} D={D} T=1500 chunk=600: err {err:.2e}")
    assert out.shape == (2, T, 53) and err < 1e-3
    # the first chunk is the reference's own decode of the first 600 frames
    ref600 = OF.predict_cached(w, hs[:, :600], 30)
    assert (out[:, :600] - ref600).abs().max().item() < 1e-3


def test_fp16_coefficient_output_head(gpu):
    """BASELINE configs[4] "fp16 coeffs": the EMOTE/FLINT head's LAST kernel stores the coefficients as IEEE half
    (AviGemm.C16).  Every element is the fp32 result rounded once: |half - fp32| <= 2^-11 |fp32| (half has 11 significant
    bits; subnormals below 6e-5 get an absolute 2^-25), and it equals torch's own rounding of the fp32 tensor."""
    from avi_talking_amd.weights import make_emote_weights
    from avi_talking_amd.host.talking_head import EmoteHead
    g = torch.Generator().manual_seed(31)
    B, T = 3, 333                                           # T not a multiple of 8: padded and cropped
    feat, style = torch.randn(B, T, 768, generator=g).to(gpu), (torch.randn(B, 1, 128, generator=g) * 0.5).to(gpu)
    head = EmoteHead(make_emote_weights(1), device=gpu)
    f32 = head(feat, style)
    f16 = head(feat, style, out_dtype=torch.float16)
    for k in ("predicted_exp", "predicted_jaw"):
        a, h = f32[k], f16[k]
        assert h.dtype == torch.float16 and h.shape == a.shape
        assert torch.equal(h, a.to(torch.float16))
        assert ((h.float() - a).abs() <= a.abs() * 2.0 ** -11 + 2.0 ** -25).all()


@pytest.mark.parametrize("D,steps", [(64, "0"), (256, "1")])
def test_fp16_coefficient_output_faceformer(gpu, monkeypatch, D, steps):
    """Both FaceFormer device paths with half output (un-normalised coefficients): element by element the rounded fp32
    decode - the fed-back frame stays fp32, so the recursion itself is unchanged."""
    import os
    import numpy as np
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    monkeypatch.setenv("AVI_FF_STEPS", steps)
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mean, std = np.load(os.path.join(G, "coeff_mean.npy")), np.load(os.path.join(G, "coeff_std.npy"))
    ff = Faceformer(make_faceformer_weights(2, feature_dim=D), period=30, device=gpu, coeff_mean=mean, coeff_std=std)
    hs = torch.randn(2, 700, D, generator=torch.Generator().manual_seed(72)).to(gpu)
    a = ff.decode(hs, chunk=600)
    h = ff.decode(hs, chunk=600, out_dtype=torch.float16)
    assert h.dtype == torch.float16 and torch.equal(h, a.to(torch.float16))
    with pytest.raises(ValueError):
        ff.decode(hs, chunk=600, out_dtype=torch.bfloat16)


def test_batch_whose_activations_exceed_2_31_elements(gpu):
    """BASELINE configs[4] at a size where 32-bit element offsets break: 24 clips x 60 s make conv layer 0's output
    24 x 191 999 x 512 = 2.36 G elements (4.7 GB per 16-bit plane).  Size-independent properties through the whole sampling
    path on the default plan: a batch permutation permutes the outputs BIT-EXACTLY (every kernel's indexing is position
    independent up to the last clip), and two rows of the big batch equal the same clips run as a batch of 2, which the CPU
    oracle pins."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    from oracle import emote as OE, prior as OP, wav2vec2 as OW
    wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
    B, T = 24, 1500
    g = torch.Generator().manual_seed(2424)
    pcm = (torch.randn(B, T * 640, generator=g) * 3000).clamp(-32768, 32767).to(torch.int16)
    voxel, noise = torch.randn(B, 768, generator=g), torch.randn(101, B, 1, 128, generator=g)
    assert B * ((T * 640 - 10) // 5 + 1) * 512 > 2 ** 31
    pipe = SamplingPipeline(wa, wh, wp, device=gpu, out_dtype=torch.float16)
    d = [t.to(gpu) for t in (pcm, voxel, noise)]
    out = pipe.run(*d)
    exp, jaw = out["predicted_exp"].clone(), out["predicted_jaw"].clone()
    assert exp.shape == (B, T, 50) and exp.dtype == torch.float16 and torch.isfinite(exp.float()).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(gpu)
    outp = pipe.run(d[0][perm].contiguous(), d[1][perm].contiguous(), d[2][:, perm].contiguous())
    assert torch.equal(outp["predicted_exp"], exp[perm]) and torch.equal(outp["predicted_jaw"], jaw[perm])
    del outp
    sub = [1, B - 1]                                     # the last clip sits beyond the 2^31-element mark in every big tensor
    p32 = SamplingPipeline(wa, wh, wp, device=gpu)       # fp32 coefficients for the comparison with the oracle
    small = p32.run(d[0][sub].contiguous(), d[1][sub].contiguous(), d[2][:, sub].contiguous())
    e_rows = max((small["predicted_exp"] - exp[sub].float()).abs().max().item(),
                 (small["predicted_jaw"] - jaw[sub].float()).abs().max().item())
    feat = OW.forward(wa, OW.normalize_audio(pcm[sub], joint=False), frame_num=T)
    te, _ = OP.brain_network(wp, voxel[sub])
    ref = OE.forward(wh, feat, OP.p_sample_loop(wp, te.view(2, 1, 128), noise[:, sub]))
    e_or = max((small["predicted_exp"].cpu() - ref["predicted_exp"]).abs().max().item(),
               (small["predicted_jaw"].cpu() - ref["predicted_jaw"]).abs().max().item())
    print(f"B = 24 x 60 s: rows of the big batch vs the batch of 2 {e_rows:.2e} (half-precision storage: 1e-3 x |coeff|), "
          f"batch of 2 vs oracle {e_or:.2e}; peak memory {torch.cuda.max_memory_allocated(gpu) / 1e9:.1f} GB")
    assert e_or < 3e-4                                   # the plan's gate
    assert e_rows < 2e-3                                 # fp16 rounding of coefficients up to ~2 (2^-11 relative) + tile-shape order
    pipe.synchronize()
