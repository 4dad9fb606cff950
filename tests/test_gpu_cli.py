"""GPU: the reference's command line end to end (host/cli.py behind train_diffusion_prior.py): test mode writes the
reference's output layout for an utterance, train mode writes train_logs/<jobname>/last.pth that resumes."""
import os
import pickle
import subprocess
import sys
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_test_mode_writes_reference_layout(gpu, tmp_path):
    wav_dir = tmp_path / "Mead_X" / "X001_front_happy_level1_001"
    wav_dir.mkdir(parents=True)
    pcm = np.load(os.path.join(ROOT, "tests", "golden", "fixture_wav_ch0.npz"))["pcm"][:32000]
    with wave.open(str(wav_dir / "X001_front_happy_level1_001.wav"), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(16000)
        f.writeframes(pcm.astype("<i2").tobytes())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_diffusion_prior.py"), "--is_test", "1",
                        "--is_talking_instruct", "1", "--test_audio_path", str(wav_dir / "X001_front_happy_level1_001.wav"),
                        "--save_subdir", "res", "--run_dir", str(tmp_path / "run")],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    folder = tmp_path / "run" / "test_videos_res" / "Mead_X" / "X001_front_happy_level1_001"
    assert (folder / "instruction.txt").exists()
    with open(folder / "flame" / "flame_X001_front_happy_level1_001.pkl", "rb") as fh:
        d = pickle.load(fh)                                        # written by this test run: plain numpy arrays
    assert set(d) == {"shape", "expression", "jaw_pose", "global_pose"}
    # 32 000 samples = 50 frames; the reference's create_base_sample (evaluation_functions.py:141-161, mirrored by
    # host/sample.py and pinned on it) pads one zero frame and one zero sample column: its pass, and ours, returns 51 frames
    assert d["expression"].shape == (51, 50) and d["jaw_pose"].shape == (51, 3) and np.isfinite(d["expression"]).all()


def test_cli_train_mode_writes_and_resumes_checkpoint(gpu, tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "train_diffusion_prior.py"), "--jobname", "job", "--max_epoch", "3",
           "--batch_size", "64", "--synthetic_steps", "2", "--log_loss_steps", "1", "--max_lr", "0.001"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    path = tmp_path / "train_logs" / "job" / "last.pth"
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["epoch"] == 2 and len(ck["train_losses"]) == 6 and len(ck["lrs"]) == 6
    assert ck["lrs"][0] == pytest.approx(0.001 / 25) and max(ck["lrs"]) <= 0.001 + 1e-12
    r = subprocess.run(cmd + ["--resume_from_ckpt", "1", "--ckpt_path", str(path), "--max_epoch", "4"],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert torch.load(path, map_location="cpu", weights_only=True)["epoch"] == 3


def test_bench_two_ranks_rehearsal_on_one_gpu(gpu):
    """`python bench.py --gpus 2` end to end with two REAL rank processes on the one card (AVI_BENCH_ONE_GPU_REHEARSAL=1: gloo
    as the transport, RCCL refuses two ranks on one device): the launcher, the rendezvous, the barriers and the max-reduce
    around the timed region, and the data-parallel training leg (hipGraph segments + one all-reduce per gradient bucket between
    two processes).  One JSON line from rank 0 with n_gpus = 2; the numbers are not measurements and the line says so."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AVI_BENCH_ONE_GPU_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--train-steps", "3", "--legs", "train"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and "rehearsal" in rec and rec["value"] > 0
    tr = rec["train"]
    assert tr["global_batch"] == 128 and "segments" in str(tr["hipgraph"]) and tr["loss_prior"] == tr["loss_prior"]
    assert len(tr["gradient_buckets"]) == 7
