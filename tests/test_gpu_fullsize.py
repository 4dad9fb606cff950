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


def test_config1_batch_permutation_and_subbatch(gpu):
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline
    pipe = SamplingPipeline(W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3), device=gpu,
                            joint_norm=False)
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
    assert (outs["predicted_exp"] - exp[sub]).abs().max().item() < 3e-5
    assert (outs["predicted_jaw"] - jaw[sub]).abs().max().item() < 3e-5
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
