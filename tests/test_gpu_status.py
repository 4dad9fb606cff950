"""GPU: the device-side failure reports (include/avi_talking.h "Status words", host/status.py) and the safety net on top of
them (SamplingPipeline.run_checked):

* the words live in pinned host memory and are read without a synchronisation of their own;
* fp16 activation planes (the 2-term fp16 GEMM groups of the default ``mixed`` plan) have a finite range: with the conv
  stack's activations scaled by 1e-2 ... 1e2 the coefficients stay within the plan's gate of the oracle and nothing is
  reported; scaled to 1e6 / 1e-6 the plane producers report overflow / underflow, ``check()`` raises and ``run_checked``
  re-runs the batch on the bf16x3 plan (bf16 planes carry fp32's range) and returns oracle-accurate coefficients;
* a paired-sampler workgroup whose partner never answers (fault injection) returns NaN - loud by itself -, raises
  ``PairTimeout`` at the next entry point, and ``run_checked`` repairs the batch on the unpaired kernel.
The conv stack is scale-free downstream (the feature projection's LayerNorm removes the scale), so the oracle is well
defined at every scale: GroupNorm's gain and bias of conv layer 0 (HF Wav2Vec2GroupNormConvLayer) are multiplied by s."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def weights():
    from avi_talking_amd import weights as W
    return W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)


def _scaled(wa, s, last=1.0):
    """GroupNorm gain / bias of conv layer 0 times ``s`` (every later conv layer is bias-free and GELU is near-homogeneous
    at both ends, so the activations of layers 0-5 scale by ~s); ``last``: conv layer 6's weight times this - the layer's
    output is fp32 - so that the features the projection LayerNorm sees keep an ordinary magnitude."""
    w = dict(wa)
    for k in ("feature_extractor.conv_layers.0.layer_norm.weight", "feature_extractor.conv_layers.0.layer_norm.bias"):
        w[k] = wa[k] * s
    w["feature_extractor.conv_layers.6.conv.weight"] = wa["feature_extractor.conv_layers.6.conv.weight"] * last
    return w


def _inputs(B, T, seed=5):
    g = torch.Generator().manual_seed(seed)
    pcm = (torch.randn(B, T * 640, generator=g) * 3000).to(torch.int16)
    return pcm, torch.randn(B, 768, generator=g), torch.randn(101, B, 1, 128, generator=g)


def _oracle(wa, wh, wp, pcm, voxel, noise):
    from oracle import emote as OE, prior as OP, wav2vec2 as OW
    B, T = pcm.shape[0], pcm.shape[1] // 640
    feat = OW.forward(wa, OW.normalize_audio(pcm, joint=False), frame_num=T)
    te, _ = OP.brain_network(wp, voxel)
    return OE.forward(wh, feat, OP.p_sample_loop(wp, te.view(B, 1, 128), noise))


def _err(out, ref):
    return max((out["predicted_exp"].cpu() - ref["predicted_exp"]).abs().max().item(),
               (out["predicted_jaw"].cpu() - ref["predicted_jaw"]).abs().max().item())


def test_status_words_reach_the_host(gpu):
    from avi_talking_amd import lib as L
    from avi_talking_amd.host import status
    status.clear()
    assert L.load().avi_status_words() == status.words().data_ptr()
    for k in (status.F16_OVERFLOW, status.F16_TINY, status.PAIR_TIMEOUT):
        assert status.read() == (False, False, False)
        L.check(L.load().avi_debug_raise_status(k, L.stream_ptr()), "avi_debug_raise_status")
        torch.cuda.synchronize()           # the launch's completion; the READ below touches host memory only
        assert status.read() == tuple(i == k for i in range(3))
        with pytest.raises(status.PairTimeout if k == status.PAIR_TIMEOUT else status.RangeError):
            status.raise_if_set()
        assert status.read() == (False, False, False)          # raised once, then clear
    assert L.load().avi_debug_raise_status(7, L.stream_ptr()) == L.AVI_EINVAL


@pytest.mark.parametrize("scale", [1e-2, 1.0, 1e2])
def test_fp16_planes_hold_their_accuracy_over_four_decades(gpu, weights, scale):
    from avi_talking_amd.host import status
    from avi_talking_amd.host.pipeline import SamplingPipeline
    wa, wh, wp = weights
    was = _scaled(wa, scale)
    pcm, voxel, noise = _inputs(2, 50)
    status.clear()
    pipe = SamplingPipeline(was, wh, wp, device=gpu)           # the default plan
    assert pipe.plan.name == "mixed"
    out = pipe.run(pcm.to(gpu), voxel.to(gpu), noise.to(gpu))
    pipe.synchronize()                                          # raises on any report
    e = _err(out, _oracle(was, wh, wp, pcm, voxel, noise))
    print(f"conv-stack activations x {scale:g}: mixed vs oracle {e:.2e} (gate 3e-4), no range report")
    assert e < 3e-4


@pytest.mark.parametrize("scale,last,which", [(1e6, 1.0, "overflow"), (1e-5, 1e5, "tiny"), (1e-6, 1.0, "tiny")])
def test_fp16_plane_range_guard_and_fallback(gpu, weights, scale, last, which):
    """(1e-5, x 1e5 in the last layer): tiny planes in layers 0-5 but ordinary features - the lost bits show in the
    coefficients; (1e-6, 1): tiny all the way - the projection LayerNorm's epsilon then wipes the features out in the
    reference too, so the unguarded result happens to be right (printed), the report is raised all the same."""
    from avi_talking_amd.host import status
    from avi_talking_amd.host.pipeline import SamplingPipeline
    wa, wh, wp = weights
    was = _scaled(wa, scale, last)
    pcm, voxel, noise = _inputs(2, 50)
    ref = _oracle(was, wh, wp, pcm, voxel, noise)
    status.clear()
    pipe = SamplingPipeline(was, wh, wp, device=gpu)
    d = [t.to(gpu) for t in (pcm, voxel, noise)]
    out = pipe.run(*d)
    torch.cuda.synchronize()
    ovf, tiny, pair = status.read()
    e_bad = _err(out, ref)
    print(f"conv-stack activations x {scale:g} (last layer x {last:g}): overflow={ovf} tiny={tiny}, unguarded error vs oracle "
          f"{e_bad:.2e}")
    assert (ovf if which == "overflow" else tiny) and not pair
    if last != 1.0:
        assert e_bad > 3e-4                 # without the guard this batch would have been returned beyond the plan's gate
    with pytest.raises(status.RangeError):
        pipe.check()
    # a caller that never calls check(): the NEXT entry point raises by itself (no synchronisation, no device read)
    pipe.run(*d)
    torch.cuda.synchronize()
    with pytest.raises(status.RangeError):
        pipe.run(*d)
    # the safety net: the batch comes back from the bf16x3 plan, oracle-accurate
    good = pipe.run_checked(*d)
    e = _err(good, ref)
    print(f"run_checked: {pipe.last_fallback}; error vs oracle {e:.2e}")
    assert "bf16x3" in pipe.last_fallback and e < 1e-4
    assert status.read() == (False, False, False)
    # and an all-3-term pipeline never reports (its planes are bf16)
    p3 = SamplingPipeline(was, wh, wp, device=gpu, prec="bf16x3")
    o3 = p3.run(*d)
    p3.synchronize()
    assert _err(o3, ref) < 1e-4


def test_paired_sampler_timeout_is_loud_and_repaired(gpu, weights):
    from avi_talking_amd import lib as L
    from avi_talking_amd.host import status
    from avi_talking_amd.host.pipeline import SamplingPipeline
    wa, wh, wp = weights
    pcm, voxel, noise = _inputs(3, 25, seed=9)
    d = [t.to(gpu) for t in (pcm, voxel, noise)]
    status.clear()
    pipe = SamplingPipeline(wa, wh, wp, device=gpu)
    assert pipe.prior.uses_pairs(3) and pipe.prior.cus_held(3) == 4
    ok = pipe.run(*d)
    pipe.synchronize()
    style_ok = ok["style_emb"].clone()
    L.load().avi_debug_fault_inject(status.FAULT_PAIR_PARTNER_ABSENT)
    try:
        bad = pipe.run(*d)
        torch.cuda.synchronize()
        # loud by itself: the half that gave up took NaN for what it never received
        assert torch.isnan(bad["style_emb"]).all() and torch.isnan(bad["predicted_exp"]).any()
        assert status.read() == (False, False, True)
        with pytest.raises(status.PairTimeout):
            pipe.run(*d)                                   # the next entry point raises; nothing was launched
        assert status.read() == (False, False, False)
        fixed = pipe.run_checked(*d)                       # times out again, then re-runs on the unpaired kernel
        assert "unpaired" in pipe.last_fallback
    finally:
        L.load().avi_debug_fault_inject(0)
    assert torch.isfinite(fixed["predicted_exp"]).all()
    assert (fixed["style_emb"] - style_ok).abs().max().item() < 1e-5          # paired vs unpaired: 1-3e-6
    assert _err(fixed, _oracle(wa, wh, wp, pcm, voxel, noise)) < 3e-4
    again = pipe.run_checked(*d)                           # fault gone: the paired kernel works and nothing is repaired
    assert pipe.last_fallback is None and torch.equal(again["style_emb"], style_ok)


def test_paired_sampler_refuses_more_exchanges_than_its_tags_hold(gpu, weights):
    """2 * depth * timesteps must fit the 16-bit exchange number of a granule tag (the reference samples with up to 1000
    steps at train_diffusion_prior.py:842: 12 000 exchanges fit; 6 000 steps would not)."""
    import ctypes as C
    from avi_talking_amd import lib as L
    from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
    prior = InstructDiffusionPrior.from_state_dict(weights[2], device=gpu)
    B = 2
    te = torch.zeros(B, 128, device=gpu)
    noise = torch.zeros(101, B, 128, device=gpu)
    out = torch.empty(B, 128, device=gpu)
    ws = torch.zeros(L.load().avi_prior_pair_workspace_bytes(B) // 8, dtype=torch.int64, device=gpu)
    table = prior.time_table()             # built with the real step count, BEFORE the struct below is altered
    torch.cuda.synchronize()
    cw = prior.net.cw
    saved = cw.timesteps
    try:
        cw.timesteps = 6000                # only the argument check may see this value: nothing is launched
        rc = L.load().avi_prior_sample_paired(C.byref(cw), C.byref(prior.net.planes), te.data_ptr(), noise.data_ptr(), B,
                                              1.0, out.data_ptr(), table.data_ptr(), ws.data_ptr(), L.stream_ptr())
        assert rc == L.AVI_EINVAL
    finally:
        cw.timesteps = saved
