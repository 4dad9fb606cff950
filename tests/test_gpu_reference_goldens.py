"""GPU parity, HIP compared DIRECTLY with outputs of the reference's own classes (tests/golden/*.npz from
tests/golden/make_golden.py), with no oracle in between: wav2vec2 (models/lib/wav2vec.py Wav2Vec2Model on the
reference's fixture WAV and a seeded clip) and the EMOTE head + FLINT decoder (inferno BertPriorDecoder /
L2lDecoder / StackLinearSquash / LinearSequenceEncoder / LinearEmotionCondition run unmodified).
BrainNetwork, FaceFormer.predict, lbs() and CLIPTextModel have their direct tests next to their kernels
(test_gpu_prior / test_gpu_faceformer / test_gpu_flame / test_gpu_clip_text).
Every test that runs the audio encoder is parametrised over the PRODUCT DEFAULT plan (ops.DEFAULT_PREC = "mixed", the plan
bench.py's headline is measured on) and the all-3-term plan "bf16x3".
Tolerance: north_star's 1e-3 max-abs on coefficients; the gates below are what each plan is held to (mixed: 3e-4 on the
coefficients, three times tighter than north_star; bf16x3: 5e-5)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLANS = ["mixed", "bf16x3"]
COEFF_GATE = {"mixed": 3e-4, "bf16x3": 5e-5}          # un-normalised expression / jaw coefficients
HIDDEN_GATE = {"mixed": 4e-3, "bf16x3": 2e-4}         # wav2vec2 last_hidden_state (unit variance after the final LayerNorm)
CONV_GATE = {"mixed": 4e-3, "bf16x3": 1e-4}           # conv feature encoder output


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("plan", PLANS)
@pytest.mark.parametrize("tag", ["fixture", "randn", "randn_fn40"])
def test_wav2vec2_hip_matches_reference_golden(gpu, tag, plan):
    from avi_talking_amd import ops
    from avi_talking_amd.weights import make_wav2vec2_weights
    from avi_talking_amd.host.wav2vec import Wav2Vec2Model
    g = _load(f"wav2vec2_{tag}.npz")
    if tag == "fixture":          # the reference's experiments/wav_dir/0001 clip, channel 0, normalised per clip ON DEVICE
        pcm = torch.from_numpy(_load("fixture_wav_ch0.npz")["pcm"].copy())
        assert np.array_equal(pcm[:64].numpy(), g["pcm_head"])
        x = ops.audio_normalize(pcm[None].to(gpu), joint=False)
    else:
        x = torch.randn(1, 32000, generator=torch.Generator().manual_seed(5)).to(gpu)
    assert x.shape[1] == int(g["n_samples"])
    fn = int(g["frame_num"])
    model = Wav2Vec2Model(make_wav2vec2_weights(0), device=gpu, prec=plan)
    assert model.plan.name == plan
    out = model(x, "vocaset", frame_num=None if fn < 0 else fn)
    feats = out.extract_features.transpose(1, 2).cpu()               # reference layout (B, 512, L)
    hid = out.last_hidden_state.cpu()
    assert list(feats.shape) == list(g["conv_shape"]) and list(hid.shape) == list(g["out_shape"])
    e_conv = np.abs(feats[0, ::16, ::9].numpy() - g["conv_slice"]).max()
    e_out = np.abs(hid[0, ::3, ::8].numpy() - g["out_slice"]).max()
    print(f"{tag} [{plan}]: HIP vs reference Wav2Vec2Model: conv {e_conv:.2e} last_hidden_state {e_out:.2e}")
    assert e_conv < CONV_GATE[plan]
    assert e_out < HIDDEN_GATE[plan]
    from avi_talking_amd.host import status
    torch.cuda.synchronize()
    assert status.read() == (False, False, False)          # no fp16-plane range report on the reference's inputs


@pytest.mark.parametrize("plan", PLANS)
@pytest.mark.parametrize("tag", ["fixture", "randn2", "fixture641"])
def test_audio_to_coefficients_hip_matches_reference_chain(gpu, tag, plan):
    """The reference's one real input (experiments/wav_dir/0001, channel 0) and two seeded clips from int16 PCM to FLAME
    coefficients: HIP (device-side normalisation -> wav2vec2 -> EMOTE head + FLINT) against the golden of the reference's own
    Wav2Vec2Model + LinearSequenceEncoder + BertPriorDecoder + L2lDecoder chained in make_golden.py::gen_fixture_chain."""
    from avi_talking_amd.weights import make_emote_weights, make_wav2vec2_weights
    from avi_talking_amd.host import status
    from avi_talking_amd.host.talking_head import TalkingHeadWrapper
    g = _load("fixture_chain.npz")
    pcm = torch.from_numpy(g[f"{tag}_pcm"].copy())
    spf = 641 if tag.endswith("641") else 640      # "fixture641": the reference entry point's own framing (host/sample.py)
    B, T = pcm.shape[0], pcm.shape[1] // spf
    if tag == "fixture":
        assert np.array_equal(pcm[0, :64].numpy(), _load("wav2vec2_fixture.npz")["pcm_head"])
    th = TalkingHeadWrapper(make_wav2vec2_weights(0), make_emote_weights(1), device=gpu, prec=plan, joint_norm=False)
    out = th({"raw_audio": pcm.view(B, T, spf).to(gpu), "samplerate": [16000] * B},
             style_emb=torch.from_numpy(g[f"{tag}_style"]).to(gpu), is_external_style_emb=True)
    e_hid = np.abs(out["audio_feature"][:, ::5, ::16].cpu().numpy() - g[f"{tag}_hidden_slice"]).max()
    e_exp = np.abs(out["predicted_exp"].cpu().numpy() - g[f"{tag}_exp"]).max()
    e_jaw = np.abs(out["predicted_jaw"].cpu().numpy() - g[f"{tag}_jaw"]).max()
    print(f"{tag} [{plan}]: HIP vs the reference's wav2vec2 + EMOTE/FLINT chain: hidden {e_hid:.2e} exp {e_exp:.2e} "
          f"jaw {e_jaw:.2e} (gate {COEFF_GATE[plan]:.0e})")
    assert out["predicted_exp"].shape == (B, T, 50)
    assert max(e_exp, e_jaw) < COEFF_GATE[plan] and e_hid < HIDDEN_GATE[plan]
    assert status.read() == (False, False, False)


@pytest.mark.parametrize("tag,B,T", [("a", 2, 250), ("b", 1, 61), ("c", 3, 8)])
def test_emote_head_hip_matches_reference_golden(gpu, tag, B, T):
    from avi_talking_amd import ops
    from avi_talking_amd.weights import make_emote_weights
    from avi_talking_amd.host.talking_head import EmoteHead
    g = _load("emote.npz")
    gen = torch.Generator().manual_seed(31)
    feat, style = torch.randn(B, T, 768, generator=gen), torch.randn(B, 1, 128, generator=gen) * 0.5
    out = EmoteHead(make_emote_weights(1), device=gpu, prec=ops.PREC_BF16X3)(feat.to(gpu), style.to(gpu))
    e_enc = np.abs(out["seq_encoder_output"][:, ::7, ::5].cpu().numpy() - g[tag + "_seq_encoder_output"]).max()
    e_lat = np.abs(out["latent"].cpu().numpy() - g[tag + "_latent"]).max()
    e_exp = np.abs(out["predicted_exp"].cpu().numpy() - g[tag + "_exp"]).max()
    e_jaw = np.abs(out["predicted_jaw"].cpu().numpy() - g[tag + "_jaw"]).max()
    print(f"{tag}: HIP vs reference BertPriorDecoder+L2lDecoder: enc {e_enc:.2e} latent {e_lat:.2e} "
          f"exp {e_exp:.2e} jaw {e_jaw:.2e}")
    assert max(e_exp, e_jaw) < 1e-3 and max(e_exp, e_jaw) < 2e-4
    assert e_lat < 1e-3


def test_emote_style_paths_hip_match_reference_golden(gpu):
    """Per-frame external style; LinearEmotionCondition via only_style_emb; style from the sample's own one-hots."""
    from avi_talking_amd.weights import make_emote_weights
    from avi_talking_amd.host.talking_head import EmoteHead
    g = _load("emote.npz")
    B, T = 2, 25
    gen = torch.Generator().manual_seed(41)
    feat = torch.randn(B, T, 768, generator=gen)
    style_t = torch.randn(B, T, 128, generator=gen) * 0.5
    oh = torch.nn.functional.one_hot
    expr = oh(torch.tensor([3, 5]), 8)[:, None].expand(B, T, 8)
    inten = oh(torch.tensor([2, 0]), 3)[:, None].expand(B, T, 3)
    ident = oh(torch.tensor([7, 30]), 32)[:, None].expand(B, T, 32)
    shape = torch.randn(B, 300, generator=gen)
    head = EmoteHead(make_emote_weights(1), device=gpu)
    out = head(feat.to(gpu), style_t.to(gpu))
    assert np.abs(out["predicted_exp"].cpu().numpy() - g["t_exp"]).max() < 2e-4
    assert np.abs(out["predicted_jaw"].cpu().numpy() - g["t_jaw"]).max() < 2e-4
    own = head.style_condition(expr.to(gpu), inten.to(gpu), ident.to(gpu), shape.to(gpu))
    assert np.abs(own.cpu().numpy() - g["own_style"]).max() < 1e-4
    out = head(feat.to(gpu), own)
    assert np.abs(out["predicted_exp"].cpu().numpy() - g["own_exp"]).max() < 2e-4
    assert np.abs(out["predicted_jaw"].cpu().numpy() - g["own_jaw"]).max() < 2e-4


def test_flint_decoder_hip_matches_reference_l2ldecoder(gpu):
    from avi_talking_amd.weights import make_emote_weights
    from avi_talking_amd.host.talking_head import EmoteHead
    g = _load("emote.npz")
    z = torch.randn(2, 5, 256, generator=torch.Generator().manual_seed(43))
    out = EmoteHead(make_emote_weights(1), device=gpu).flint_decoder(z.to(gpu))
    e = np.abs(out.cpu().numpy() - g["flint_z_out"]).max()
    print(f"HIP flint_decoder vs reference L2lDecoder: {e:.2e}")
    assert e < 2e-4


@pytest.mark.parametrize("plan", PLANS)
def test_emote_audio_wrapper_hip_matches_reference(gpu, plan):
    """int16 raw_audio -> audio_feature through ``TalkingHeadWrapper.forward_audio`` (device-side JOINT normalisation, the
    processor quirk of AudioEncoders.py:170-178; desired length = T) and the free-length ``ceil`` rule of
    ``Wav2Vec2Model(length_mode="ceil")`` against EMOTE's own ``Wav2Vec2Encoder._forward`` / ``Wav2Vec2ModelResampled`` run
    unmodified (tests/golden/emote_audio.npz)."""
    from avi_talking_amd import ops
    from avi_talking_amd.weights import make_emote_weights, make_wav2vec2_weights
    from avi_talking_amd.host.talking_head import TalkingHeadWrapper
    g = _load("emote_audio.npz")
    raw = torch.from_numpy(g["raw_audio"].copy()).to(gpu)
    th = TalkingHeadWrapper(make_wav2vec2_weights(0), make_emote_weights(1), device=gpu, prec=plan, joint_norm=True)
    s = th.forward_audio({"raw_audio": raw, "samplerate": [16000, 16000]})
    assert list(s["audio_feature"].shape) == list(g["audio_feature_shape"])
    e_in = np.abs(s["processed_audio"][:, ::97].cpu().numpy() - g["processed_audio_slice"]).max()
    e = np.abs(s["audio_feature"][:, :, ::8].cpu().numpy() - g["audio_feature_slice"]).max()
    n = int(g["free_len_input"])
    free = th.audio_model(s["processed_audio"][:, :n].contiguous()).last_hidden_state
    assert list(free.shape) == list(g["free_len_shape"])             # 19 000 samples -> 30 frames (ceil), not 29
    ef = np.abs(free[:, :, ::8].cpu().numpy() - g["free_len_slice"]).max()
    print(f"[{plan}] HIP vs EMOTE's own audio wrapper: normalised audio {e_in:.2e}, audio_feature {e:.2e}, free length {ef:.2e}")
    assert e_in < 5e-6 and e < HIDDEN_GATE[plan] and ef < HIDDEN_GATE[plan]
