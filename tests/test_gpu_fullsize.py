"""Size-independent properties of the sampling path at BASELINE.json's full config[1] size (32 clips x 10 s,
100-step DDPM), where the CPU oracle is too slow to be the checker:

* utterances are independent: permuting the batch permutes the outputs BIT-EXACTLY (every output element is reduced
  in a fixed order that does not depend on which rows share its GEMM tile, attention workgroup or split-K slice);
* a sub-batch reproduces the rows of the full batch (the oracle pins those rows at small sizes elsewhere);
* hipGraph replays equal the eager pass, run to run, and BOTH equal a fully serial pass (sampler and audio encoder on
  one stream): the two-stream overlap must not change a single bit.  This also holds with the aligner's dozen short
  matrix-core launches next to conv layer 0 (AVI_ALIGNER_SIDE=1), the arrangement that corrupted rows of conv layer 0
  while the library still contained packed-FP32 instructions (build.py, scripts/diag_concurrency.py).
Per-clip audio normalisation is used so that clips do not interact through the batch statistics (the joint mode of
AudioEncoders.py:170-178 couples them by design and is covered at small sizes by tests/test_gpu_emote.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, seed):
    g = torch.Generator().manual_seed(seed)
    pcm = (torch.randn(B, 160000, generator=g) * 3000).clamp(-32768, 32767).to(torch.int16)
    voxel = torch.randn(B, 768, generator=g)
    noise = torch.randn(101, B, 1, 128, generator=g)
    return pcm, voxel, noise


@pytest.mark.parametrize("plan", ["mixed", "bf16x3"])
def test_config1_batch_permutation_and_subbatch(gpu, plan):
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu,
                            joint_norm=False, prec=plan)
    assert pipe.plan.name == plan
    B = 32
    pcm, voxel, noise = (t.to(gpu) for t in _inputs(B, 4242))
    out = pipe.run(pcm, voxel, noise)
    exp, jaw = out["predicted_exp"].clone(), out["predicted_jaw"].clone()
    assert exp.shape == (B, 250, 50) and jaw.shape == (B, 250, 3)
    assert torch.isfinite(exp).all() and torch.isfinite(jaw).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(7)).to(gpu)
    outp = pipe.run(pcm[perm].contiguous(), voxel[perm].contiguous(), noise[:, perm].contiguous())
    assert torch.equal(outp["predicted_exp"], exp[perm]) and torch.equal(outp["predicted_jaw"], jaw[perm])
    # a sub-batch (different tile fill, different number of sampler workgroups) reproduces its rows.  Not bit for bit:
    # the small conv layers of a 3-clip batch run on the 128-row tile kernels (other summation order; their epilogue
    # evaluates GELU by the rational erf, the 256 x 256 kernel by its LDS table - both within 5e-7 of the exact GELU)
    sub = [3, 17, 30]
    outs = pipe.run(pcm[sub].contiguous(), voxel[sub].contiguous(), noise[:, sub].contiguous())
    e_sub = max((outs["predicted_exp"] - exp[sub]).abs().max().item(), (outs["predicted_jaw"] - jaw[sub]).abs().max().item())
    print(f"[{plan}] sub-batch vs its rows of the full batch: {e_sub:.2e}")
    assert e_sub < {"mixed": 1e-4, "bf16x3": 3e-5}[plan]
    # fully serial reference: everything on the current stream
    side, pipe.side = pipe.side, torch.cuda.current_stream(gpu)
    try:
        ser = pipe.run(pcm, voxel, noise)
        torch.cuda.synchronize()
    finally:
        pipe.side = side
    assert torch.equal(ser["predicted_exp"], exp) and torch.equal(ser["predicted_jaw"], jaw)
    # graph replays == eager == serial, several times
    pipe.capture(pcm, voxel, noise)
    for _ in range(4):
        rep = pipe.replay()
        torch.cuda.synchronize()
        assert torch.equal(rep["predicted_exp"], exp) and torch.equal(rep["predicted_jaw"], jaw)
    pipe.check()                                         # full size: no fp16-plane range report, no sampler timeout


def test_config1_aligner_overlapping_the_audio_branch(gpu, monkeypatch):
    """The aligner's short matrix-core launches on the second stream, beside conv layer 0 and the conv GEMMs: eager
    passes and graph replays must equal the serial pass bit for bit (they did not before packed-FP32 instructions
    were removed from the build)."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu,
                            joint_norm=False)
    pcm, voxel, noise = (t.to(gpu) for t in _inputs(32, 99))
    side, pipe.side = pipe.side, torch.cuda.current_stream(gpu)
    try:
        ser = pipe.run(pcm, voxel, noise)
        exp, jaw = ser["predicted_exp"].clone(), ser["predicted_jaw"].clone()
        torch.cuda.synchronize()
    finally:
        pipe.side = side
    monkeypatch.setenv("AVI_ALIGNER_SIDE", "1")
    for _ in range(3):
        out = pipe.run(pcm, voxel, noise)
        torch.cuda.synchronize()
        assert torch.equal(out["predicted_exp"], exp) and torch.equal(out["predicted_jaw"], jaw)
    pipe.capture(pcm, voxel, noise)
    for _ in range(3):
        rep = pipe.replay()
        torch.cuda.synchronize()
        assert torch.equal(rep["predicted_exp"], exp) and torch.equal(rep["predicted_jaw"], jaw)



def test_two_stream_reproduction_is_clean(gpu):
    """scripts/diag_concurrency.py with the aggressor shape that corrupted conv layer 0 when the library contained
    packed-FP32 instructions (128-row GEMM tiles: AVI_GEMM_ROWS64=0; the variable is read once per process, hence
    the subprocess): every iteration must report zero differing values.  With `AVI_PACKED_FP32=1 build.py` this fails
    in every iteration."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AVI_GEMM_ROWS64="0", SIDE="gemm")
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_concurrency.py"), "overlap"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    counts = [int(m) for m in re.findall(r"iter \d+ \[overlap\].* n (\d+)", out.stdout)]
    assert len(counts) == 4 and all(c == 0 for c in counts), out.stdout[-2000:]


def test_pipelined_replay_equals_plain_replay(gpu):
    """capture_pipelined / replay_pipelined (head of batch k beside the start of batch k+1, two graphs on two streams)
    returns bit for bit what the single-graph replay returns, pass after pass, at config[1] size."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu)
    pcm, voxel, noise = (t.to(gpu) for t in _inputs(32, 777))
    pipe.capture(pcm, voxel, noise)
    ref = pipe.replay()
    torch.cuda.synchronize()
    exp, jaw, style = ref["predicted_exp"].clone(), ref["predicted_jaw"].clone(), ref["style_emb"].clone()
    pipe.capture_pipelined(pcm, voxel, noise)
    for _ in range(5):
        out = pipe.replay_pipelined()
    torch.cuda.synchronize()
    assert torch.equal(out["predicted_exp"], exp) and torch.equal(out["predicted_jaw"], jaw)
    assert torch.equal(out["style_emb"], style)


def test_replay_pipelined_takes_the_next_batch(gpu):
    """replay_pipelined(pcm, voxel, noise) has replay()'s input-update contract: batches A, B, A handed over back to back
    (no synchronisation in between) each return what a fresh eager pass on that batch returns, bit for bit."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu)
    A = [t.to(gpu) for t in _inputs(32, 801)]
    Bt = [t.to(gpu) for t in _inputs(32, 802)]
    ref = []
    for batch in (A, Bt):
        o = pipe.run(*batch)
        torch.cuda.synchronize()
        ref.append({k: o[k].clone() for k in ("predicted_exp", "predicted_jaw", "style_emb")})
    pipe.capture_pipelined(*A)
    got = []
    for batch in (A, Bt, A):
        o = pipe.replay_pipelined(*batch)
        pipe._s_head.synchronize()                   # the pass's own results (the next pass overwrites the static outputs)
        got.append({k: o[k].clone() for k in ref[0]})
    for g, r in zip(got, (ref[0], ref[1], ref[0])):
        for k in r:
            assert torch.equal(g[k], r[k]), k
    with pytest.raises(ValueError):
        pipe.replay_pipelined(pcm=A[0][:16])


def test_second_pipeline_replays_as_fast(gpu):
    """Two pipeline objects captured in ONE process (two precisions of the same batch) replay within 3 % of what each
    does when it is the only pipeline the process has captured so far.  Round 2 measured the second object 10-30 % slower;
    the cause was per-object streams (a new side stream and five new pool streams per object changed which hardware queue
    the body, head and sampler of the second object were multiplexed onto): streams are now shared per device
    (host/pipeline.device_streams)."""
    import time
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline, device_streams
    wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
    pcm, voxel, noise = (t.to(gpu) for t in _inputs(32, 778))

    def timed(pipe, n=30):
        for _ in range(5):
            pipe.replay_pipelined()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            pipe.replay_pipelined()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    first = SamplingPipeline(wa, wh, wp, device=gpu, prec="bf16x3").capture_pipelined(pcm, voxel, noise)
    t_first = min(timed(first) for _ in range(2))
    second = SamplingPipeline(wa, wh, wp, device=gpu, prec="bf16x3").capture_pipelined(pcm, voxel, noise)
    t_second = min(timed(second) for _ in range(2))
    t_first_again = min(timed(first) for _ in range(2))
    st = device_streams(gpu)
    assert second.side is first.side and second._s_body is first._s_body
    if first.arrangement["encoder_chains"] == second.arrangement["encoder_chains"]:     # picks are kept per chain count
        pick = st["picks"][first.arrangement["encoder_chains"]]
        assert second._s_head is first._s_head
        assert first.stream_choice["head"] == second.stream_choice["head"] == pick["head"]
        assert first.stream_choice["chains"] == second.stream_choice["chains"] == pick["chains"]
    print(f"first {t_first:.3f} ms, second object {t_second:.3f} ms, first again {t_first_again:.3f} ms per pass; "
          f"arrangements {first.arrangement} / {second.arrangement}; stream picks {st['picks']}")
    assert t_second < 1.03 * t_first and t_first_again < 1.03 * t_first


def test_encoder_chains_are_bit_identical(gpu):
    """The 12 encoder layers as 2 or 4 chains of clips on separate streams (host/wav2vec._encoder_layers_split) return
    bit for bit what one chain returns - eagerly, call after call with the results consumed and dropped on the launch
    stream in between (the gathered result used to be a block of a side stream's pool: freed while its readers on the
    launch stream were still queued, it was handed to the next call's chain and the first 16 clips came out wrong), and
    under graph capture."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.wav2vec import Wav2Vec2Model
    m = Wav2Vec2Model(W.make_wav2vec2_weights(0), device=gpu, prec="mixed", length_mode="ceil")
    x = (torch.randn(32, 160000, generator=torch.Generator().manual_seed(3)) * 0.1).to(gpu)
    for cus in (0, 192):
        m.split_streams = 1
        ref = m(x, cus=cus).last_hidden_state.clone()
        for n in (2, 4):
            m.split_streams = n
            for _ in range(4):
                got = m(x, cus=cus).last_hidden_state.clone()      # clone + drop: the pattern that exposed the lifetime bug
                torch.cuda.synchronize()
                assert torch.equal(got, ref), (cus, n)
    # a 2-term plan's fp16 weight planes exist before the first pass: the very first call of a fresh model, chains first
    # (the planes used to be derived on first use, i.e. on one chain's stream while the other chain already read them)
    m2 = Wav2Vec2Model(W.make_wav2vec2_weights(0), device=gpu, prec="f16x2", length_mode="ceil")
    m2.split_streams = 2
    first = m2(x, cus=192).last_hidden_state.clone()
    m2.split_streams = 1
    assert torch.isfinite(first).all() and torch.equal(first, m2(x, cus=192).last_hidden_state)
    del m2
    m.split_streams = 2
    xs = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(xs, cus=192).last_hidden_state
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_pipelined_arrangements_are_bit_identical(gpu):
    """capture_pipelined records the body as one graph per branch (sampler's branch, audio front, one graph per encoder
    chain) on streams of the device's pool; whichever chain count is used - forced, or chosen by timing - a replay returns
    bit for bit what the eager pass returns, and the choice is reported."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu,
                            prec="mixed")
    pcm, voxel, noise = (t.to(gpu) for t in _inputs(32, 555))
    ref = pipe.run(pcm, voxel, noise)
    torch.cuda.synchronize()
    ref = {k: ref[k].clone() for k in ("predicted_exp", "predicted_jaw", "style_emb")}
    paired = pipe.prior.paired
    for chains in (1, 2, 4, None):
        pipe.capture_pipelined(pcm, voxel, noise, arrangements=None if chains is None else [(chains, paired)])
        for _ in range(3):
            out = pipe.replay_pipelined()
        torch.cuda.synchronize()
        for k, r in ref.items():
            assert torch.equal(out[k], r), (chains, k)
        a = pipe.arrangement
        assert a["paired_sampler"] == bool(paired) and (chains is None or a["encoder_chains"] == chains)
        assert len(pipe.stream_choice["chains"]) == a["encoder_chains"] - 1
        print(a, pipe.stream_choice["head"], pipe.stream_choice["chains"])
    pipe.prior.pair_status()
