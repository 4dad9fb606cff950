"""GPU: the in-pass random draws (csrc/rng.hip) against the numpy oracle (oracle/rng.py, pinned by the Random123
known-answer vectors): raw Philox words, masks and integers BIT-EXACT, normals to float rounding; determinism and
freshness under hipGraph replay; the sampling pass and the training step with their draws inside the graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(gpu, seed, offset, subseq, kind, n, param=0.0, pad=0):
    from avi_talking_amd.host import rng as R
    r = R.DeviceRng(seed, gpu, offset)
    out = torch.zeros(n + pad, dtype=R._DTYPE[kind], device=gpu)
    view = out[:n]
    r.fill(view, kind, subseq, param)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("seed,offset,subseq", [(0, 0, 0), (1234, 7, 3), ((0xDEADBEEF << 32) | 0x12345678, (5 << 32) | 9, 65535)])
@pytest.mark.parametrize("n", [1, 3, 4, 1021, 4096 * 25 + 2])
def test_raw_words_bit_exact(gpu, seed, offset, subseq, n):
    from oracle import rng as OR
    got = _fill(gpu, seed, offset, subseq, OR.KIND_RAW, n, pad=8).view(np.uint32)
    ref = OR.fill(seed, offset, subseq, OR.KIND_RAW, n)
    assert np.array_equal(got[:n], ref)
    assert not got[n:].any()                           # nothing written past n (ragged tail of the last block)


def test_masks_and_integers_bit_exact(gpu):
    from oracle import rng as OR
    n = 64 * 4096 + 3
    for kind, param in ((OR.KIND_KEEP_SCALED, 0.5), (OR.KIND_KEEP_SCALED, 0.15), (OR.KIND_BERNOULLI_U8, 0.8),
                        (OR.KIND_RANDINT_I32, 100.0), (OR.KIND_UNIFORM, 0.0)):
        got = _fill(gpu, 99, 12, 5, kind, n, param)
        ref = OR.fill(99, 12, 5, kind, n, param)
        assert np.array_equal(got, ref), kind
    keep = _fill(gpu, 99, 12, 5, OR.KIND_KEEP_SCALED, n, 0.15)
    assert abs((keep > 0).mean() - 0.85) < 4e-3 and np.allclose(keep[keep > 0], 1 / 0.85)
    assert abs(keep.mean() - 1.0) < 5e-3                                        # E[mask] = 1: dropout is unbiased
    b = _fill(gpu, 99, 12, 5, OR.KIND_BERNOULLI_U8, n, 0.8)
    assert abs(b.mean() - 0.8) < 4e-3
    t = _fill(gpu, 99, 12, 5, OR.KIND_RANDINT_I32, n, 100.0)
    assert t.min() == 0 and t.max() == 99 and np.abs(np.bincount(t, minlength=100) / n - 0.01).max() < 1.5e-3


def test_normals_match_oracle_and_moments(gpu):
    from oracle import rng as OR
    n = 101 * 32 * 128                                  # the DDPM noise of configs[1]
    got = _fill(gpu, 7, 3, 0, OR.KIND_NORMAL, n)
    ref = OR.fill(7, 3, 0, OR.KIND_NORMAL, n)
    assert np.abs(got - ref).max() < 5e-6               # fp32 log / sincos on the device vs float64 rounded once
    assert abs(got.mean()) < 6e-3 and abs(got.var() - 1.0) < 8e-3
    assert abs((np.abs(got) < 1.0).mean() - 0.6827) < 4e-3 and np.abs(got).max() < 6.0
    x = got.reshape(101, -1)
    assert abs(np.corrcoef(x[0], x[1])[0, 1]) < 0.02    # consecutive DDPM steps are uncorrelated
    other = _fill(gpu, 7, 4, 0, OR.KIND_NORMAL, n)      # next offset / another subsequence: different numbers
    sub = _fill(gpu, 7, 3, 1, OR.KIND_NORMAL, n)
    assert abs(np.corrcoef(got, other)[0, 1]) < 0.01 and abs(np.corrcoef(got, sub)[0, 1]) < 0.01


def test_fill_rejects_bad_arguments(gpu):
    from avi_talking_amd import lib as L
    from avi_talking_amd.host import rng as R
    r = R.DeviceRng(1, gpu)
    out = torch.zeros(64, device=gpu)
    with pytest.raises(ValueError):
        r.fill(out.to(torch.float64), R.NORMAL, 0)
    so = L.load()
    a = (r.state.data_ptr(), 0, R.KEEP_SCALED, 1.0, 64, out.data_ptr(), L.stream_ptr())
    assert so.avi_rng_fill(*a) == L.AVI_EINVAL                                   # drop probability 1
    assert so.avi_rng_fill(r.state.data_ptr(), 70000, R.NORMAL, 0.0, 64, out.data_ptr(), L.stream_ptr()) == L.AVI_EINVAL
    assert so.avi_rng_fill(r.state.data_ptr(), 0, R.NORMAL, 0.0, 60, out.data_ptr() + 4, L.stream_ptr()) == L.AVI_EINVAL


def test_sampling_pass_draws_its_noise_inside_the_graph(gpu):
    """noise=None + rng_seed: the (101,B,1,128) noise is drawn by graph nodes.  Replays draw fresh noise (offset advanced
    by the graph itself), a reset of the stream state reproduces the sequence bit for bit, and a pass equals the pass with
    the SAME noise injected as a tensor (the parity path)."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    from oracle import rng as OR
    wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
    B, T = 4, 50
    g = torch.Generator().manual_seed(5)
    pcm = (torch.randn(B, T * 640, generator=g) * 3000).to(torch.int16).to(gpu)
    voxel = torch.randn(B, 768, generator=g).to(gpu)
    pipe = SamplingPipeline(wa, wh, wp, device=gpu, rng_seed=2024)
    with pytest.raises(ValueError):
        SamplingPipeline(wa, wh, wp, device=gpu).run(pcm, voxel, None)
    pipe.capture_pipelined(pcm, voxel, None)
    torch.cuda.synchronize()
    pipe.rng.set_state(offset=100)
    seq = []
    for _ in range(3):
        o = pipe.replay_pipelined()
        torch.cuda.synchronize()
        seq.append((pipe._noise_buf.clone(), o["predicted_exp"].clone(), o["style_emb"].clone()))
    assert pipe.rng.get_state() == (2024, 103)
    for k in range(3):                                   # the graph drew exactly the oracle's numbers for offsets 100, 101, 102
        ref = OR.fill(2024, 100 + k, 0, OR.KIND_NORMAL, seq[k][0].numel())
        assert np.abs(seq[k][0].cpu().numpy().reshape(-1) - ref).max() < 5e-6
    assert not torch.equal(seq[0][2], seq[1][2]) and not torch.equal(seq[1][2], seq[2][2])
    pipe.rng.set_state(offset=100)
    for k in range(3):
        o = pipe.replay_pipelined()
        torch.cuda.synchronize()
        assert torch.equal(o["predicted_exp"], seq[k][1]) and torch.equal(pipe._noise_buf, seq[k][0])
    with pytest.raises(ValueError):
        pipe.replay_pipelined(noise=seq[0][0])
    inj = SamplingPipeline(wa, wh, wp, device=gpu).run(pcm, voxel, seq[1][0])
    torch.cuda.synchronize()
    assert torch.equal(inj["predicted_exp"], seq[1][1]) and torch.equal(inj["style_emb"], seq[1][2])


def test_training_step_draws_inside_the_graph(gpu):
    """capture_step(rng=...): times / noise / cond-drop / dropout masks are drawn by graph nodes.  The tensors a replay
    trained on equal the oracle's draws for its offset; replays differ; a reset reproduces the losses."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.rng import DeviceRng
    from avi_talking_amd.host.training import PriorTrainer
    from oracle import rng as OR
    B = 64
    g = torch.Generator().manual_seed(3)
    voxel, target = torch.randn(B, 768, generator=g).to(gpu), (torch.randn(B, 1, 128, generator=g) * 0.3).to(gpu)

    def losses(offset, n):
        tr = PriorTrainer(W.make_prior_weights(3), device=gpu, lr=1e-4)
        rng = DeviceRng(77, gpu)
        tr.capture_step(voxel, target, 0.005, rng=rng)
        torch.cuda.synchronize()
        rng.set_state(offset=offset)
        out, drawn = [], []
        for _ in range(n):
            o = tr.replay_step()
            torch.cuda.synchronize()
            out.append((float(o["loss_prior"]), float(o["loss_nce"])))
            r = tr._static["rand"]
            drawn.append((r["times"].clone(), r["noise"].clone(), r["brain_keep"].clone(), r["dropout_masks"][1].clone()))
        return out, drawn, rng.get_state()

    out, drawn, state = losses(50, 3)
    assert state == (77, 53)
    for k, (times, noise, bk, m1) in enumerate(drawn):
        assert np.array_equal(times.cpu().numpy(), OR.fill(77, 50 + k, 0, OR.KIND_RANDINT_I32, B, 100.0))
        assert np.abs(noise.cpu().numpy().reshape(-1) - OR.fill(77, 50 + k, 1, OR.KIND_NORMAL, B * 128)).max() < 5e-6
        assert np.array_equal(bk.cpu().numpy(), OR.fill(77, 50 + k, 2, OR.KIND_BERNOULLI_U8, B, 0.8))
        assert np.array_equal(m1.cpu().numpy().reshape(-1), OR.fill(77, 50 + k, 5, OR.KIND_KEEP_SCALED, B * 4096, 0.15))
    assert not torch.equal(drawn[0][0], drawn[1][0])
    assert all(np.isfinite(v) for pair in out for v in pair)
