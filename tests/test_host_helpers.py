"""CPU: host-side helpers around the hot path - audio framing (row A0), lr / temperature schedules (row C6), the
command-line surface and the noise draw order of the sampler."""
import os
import wave

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> (type, default) of every flag at train_diffusion_prior.py:909-947 (data copied from the reference's parser)
REFERENCE_FLAGS = {
    "max_epoch": (int, 5000), "epoch": (int, 0), "local_rank": (int, 0), "clip_size": (int, 128),
    "model_name": (str, "EMOTE"), "use_projector": (bool, True), "jobname": (str, "text2emo"), "save_subdir": (str, ""),
    "is_tensorboard_log": (int, 1), "is_test": (int, 0), "is_talking_instruct": (int, 0), "log_loss_steps": (int, 5),
    "resume_from_ckpt": (int, 0), "ckpt_path": (str, ""), "test_json_path": (str, ""), "is_output_gt": (int, 0),
    "is_use_rvd": (int, 0), "is_cal_diversity": (int, 0), "is_vis_diversity": (int, 0), "is_no_diffusion": (int, 0),
    "unset_prior": (int, 0), "unset_v2c": (int, 0), "load_talkclip_dataset": (int, 1), "wo_dataset_aug": (int, 0),
    "dataset_names": (str, ""), "seq_length": (int, 25), "vertice_dim": (int, 53), "batch_size": (int, 1),
    "workers": (int, 8), "only_load_caption": (int, 1), "max_lr": (float, 3e-4),
}


def test_cli_has_every_reference_flag_with_its_default():
    from avi_talking_amd.host.cli import build_parser
    args = build_parser().parse_args([])
    for name, (typ, default) in REFERENCE_FLAGS.items():
        assert hasattr(args, name), name
        assert getattr(args, name) == default and isinstance(getattr(args, name), typ), name
    assert args.test_audio_path.endswith("W019_front_angry_level2_007.wav") and hasattr(args, "path_to_models")
    # the argument list experiments/diffusion_test.sh passes
    a = build_parser().parse_args(
        "--dataset_names Mead_M,Mead_W --jobname align_emote_x --vertice_dim 15069 --batch_size 256 --resume_from_ckpt 1 "
        "--ckpt_path train_logs/a/last.pth --only_load_caption 1 --is_tensorboard_log 0 --max_lr 0.001 "
        "--test_audio_path a.wav --test_json_path caps/ --is_test 1 --is_talking_instruct 1 --is_output_gt 0 "
        "--save_subdir res --is_no_diffusion 0 --is_cal_diversity 0 --is_vis_diversity 0 --is_use_rvd 0".split())
    assert a.is_test == 1 and a.max_lr == 0.001 and a.batch_size == 256 and a.vertice_dim == 15069
    assert build_parser().parse_args(["--no-use_projector"]).use_projector is False


def test_cli_output_folder_layout():
    from avi_talking_amd.host.cli import output_folder
    f = output_folder("/runs/emote", "res", "/data/Mead_W/W019_front_angry_level2_007/W019_front_angry_level2_007.wav")
    assert f == "/runs/emote/test_videos_res/Mead_W/W019_front_angry_level2_007"


def _write_wav(path, x, rate=16000, nch=1):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(nch)
        f.setsampwidth(2)
        f.setframerate(rate)
        f.writeframes(x.astype("<i2").tobytes())


def test_read_and_process_audio(tmp_path):
    """evaluation_functions.py:680-714: int16 mono, cut at 22 s, whole (T, 640) frames."""
    from avi_talking_amd.host.audio_io import process_audio, read_audio
    rng = np.random.default_rng(0)
    x = rng.integers(-20000, 20000, 16000 * 23 + 123).astype(np.int16)
    _write_wav(tmp_path / "a.wav", x)
    wav, sr = read_audio(tmp_path / "a.wav")
    assert sr == 16000 and wav.dtype == np.int16 and wav.shape[0] == 22 * 16000
    assert np.array_equal(wav, x[:22 * 16000])
    s = process_audio(wav, sr, 25)
    assert s["samplerate"] == 16000 and s["raw_audio"].shape == (550, 640)
    assert np.array_equal(s["raw_audio"].reshape(-1), x[:550 * 640])
    short = process_audio(x[:1000], 16000, 25)                      # 1 whole frame, remainder dropped
    assert short["raw_audio"].shape == (1, 640) and np.array_equal(short["raw_audio"][0], x[:640])
    assert process_audio(x[:100], 16000, 25)["raw_audio"].shape == (0, 640)
    st = np.stack([x[:3200], -x[:3200]], 1).reshape(-1)             # stereo -> mono mean = 0
    _write_wav(tmp_path / "st.wav", st, nch=2)
    assert np.abs(read_audio(tmp_path / "st.wav")[0]).max() == 0
    _write_wav(tmp_path / "r.wav", x[:100], rate=44100)
    with pytest.raises(ValueError):
        read_audio(tmp_path / "r.wav")
    with pytest.raises(AssertionError):
        process_audio(x[:100], 16000, 30)


def test_fixture_wav_through_the_helpers():
    """The reference's fixture clip (tests/golden/fixture_wav_ch0.npz, 79 872 samples) -> 124 frames of 640."""
    from avi_talking_amd.host.audio_io import process_audio
    pcm = np.load(os.path.join(ROOT, "tests", "golden", "fixture_wav_ch0.npz"))["pcm"]
    s = process_audio(pcm, 16000, 25)
    assert s["raw_audio"].shape == (79872 // 640, 640) and s["raw_audio"].dtype == np.int16


@pytest.mark.parametrize("epochs,n", [(10, 7), (40, 3), (5, 11)])
def test_one_cycle_lr_matches_torch(epochs, n):
    """train_diffusion_prior.py:343-357: OneCycleLR(max_lr, epochs*len*5, final_div_factor 1000, pct_start 2/epochs)."""
    from avi_talking_amd.host.schedule import reference_schedule
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    total = int(epochs * n) * 5
    ref = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=total, final_div_factor=1000, last_epoch=-1,
                                              pct_start=2 / epochs)
    mine = reference_schedule(1e-3, epochs, n)
    for s in range(total - 1):
        assert abs(opt.param_groups[0]["lr"] - mine.lr_at(s)) <= 1e-12 + 1e-9 * mine.lr_at(s)
        assert abs(mine.get_last_lr()[0] - mine.lr_at(s)) == 0
        # cycle_momentum (torch's default): AdamW's beta1 runs inversely to the rate, beta2 stays
        assert abs(opt.param_groups[0]["betas"][0] - mine.momentum_at(s)) <= 1e-12
        assert opt.param_groups[0]["betas"][1] == 0.999
        opt.step()
        ref.step()
        mine.step()


def test_cosine_anneal_matches_oracle():
    from avi_talking_amd.host.schedule import cosine_anneal
    from oracle import prior as OP
    a = torch.tensor(cosine_anneal(0.004, 0.0075, 37), dtype=torch.float64)
    assert torch.allclose(a.float(), OP.cosine_anneal(0.004, 0.0075, 37), atol=1e-9)


def test_draw_noise_call_order_matches_reference_loop():
    """T+1 separate (B,1,128) draws from the generator: x_T first, then one per step (models/diffusion_prior.py:337,
    347-351) - checked on the CPU generator against the reference's call sequence."""
    from types import SimpleNamespace
    from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior
    obj = InstructDiffusionPrior.__new__(InstructDiffusionPrior)
    obj.device = torch.device("cpu")
    obj.noise_scheduler = SimpleNamespace(num_timesteps=100)
    out = InstructDiffusionPrior.draw_noise(obj, 3, torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(0)
    ref = [torch.randn((3, 1, 128), generator=g)]                 # image_embed = torch.randn(shape, generator)
    for _ in range(100):
        ref.append(torch.randn(torch.Size((3, 1, 128)), dtype=torch.float32, generator=g))   # p_sample noise
    assert out.shape == (101, 3, 1, 128) and torch.equal(out, torch.stack(ref))


def test_precision_plans_resolve():
    import pytest
    from avi_talking_amd import ops
    for name, conv, attn, ffn in (("bf16x3", 3, 3, 3), ("mixed", 2, 3, 3), ("mixed_ffn", 2, 3, 2), ("f16x2", 2, 2, 2)):
        p = ops.prec_plan(name)
        assert (p.conv, p.attn, p.ffn, p.small) == (conv, attn, ffn, 3) and ops.prec_plan(p) is p
    assert ops.prec_plan(ops.PREC_F16X2).sampler_all_fp16 and not ops.prec_plan("mixed").sampler_all_fp16
    with pytest.raises(ValueError):
        ops.prec_plan("fp8")


def test_pass_arrangement_rules(monkeypatch):
    """Host logic of the sampling pass's arrangement (host/pipeline._arrangements, host/diffusion_prior.uses_pairs /
    cus_held, the package's hardware-queue default): which candidates a capture times, which sampler kernel a batch gets
    (a fixed rule: the two kernels differ by 1-3e-6, so timing must not choose), how many CUs the audio GEMMs tile for."""
    from types import SimpleNamespace
    import avi_talking_amd as pkg
    from avi_talking_amd.host import pipeline as P
    from avi_talking_amd.host.diffusion_prior import InstructDiffusionPrior as Prior

    def prior(paired=True, spg=1, max_batch=32):
        p = SimpleNamespace(paired=paired, samples_per_group=spg, pair_max_batch=max_batch)
        p.uses_pairs = lambda B: Prior.uses_pairs(p, B)
        p.cus_held = lambda B: Prior.cus_held(p, B)
        return p

    p = prior()
    assert p.uses_pairs(1) and p.uses_pairs(32) and not p.uses_pairs(33)
    # a PAIR of samples shares two workgroups (csrc/prior_pair.hip launch_pair: 2 * ceil(B / 2) of them do work)
    assert p.cus_held(32) == 32 and p.cus_held(33) == 33 and p.cus_held(5) == 6 and p.cus_held(1) == 2
    assert prior(paired=False).cus_held(32) == 32 and prior(paired=False, spg=5).cus_held(32) == 7
    assert prior(spg=0).cus_held(32) == 32 and not prior(spg=0).uses_pairs(4)          # fp32 vector kernel: one CU per sample

    def candidates(B, T, split_streams=2, hwq=8, env=None, paired=True):
        monkeypatch.setattr(pkg, "HW_QUEUES", hwq)
        if env is None:
            monkeypatch.delenv("AVI_W2V_SPLIT", raising=False)
        else:
            monkeypatch.setenv("AVI_W2V_SPLIT", env)
        me = SimpleNamespace(talking_head=SimpleNamespace(audio_model=SimpleNamespace(split_streams=split_streams)),
                             prior=prior(paired))
        return P.SamplingPipeline._arrangements(me, B, T)

    assert candidates(32, 250) == [(2, True), (1, True)]               # timed: two chains against one, same sampler kernel
    assert candidates(32, 250, paired=False) == [(2, False), (1, False)]
    assert candidates(32, 250, hwq=4) == [(1, True)]                   # four queues: a second chain would share one
    assert candidates(31, 250) == [(1, True)] and candidates(2, 25) == [(1, True)]     # odd batch / too few rows per chain
    assert candidates(32, 250, split_streams=4, env="4") == [(4, True)]                # pinned by the environment
    assert candidates(30, 250, split_streams=4, env="4") == [(1, True)]                # ... when the batch divides
    assert candidates(32, 250, split_streams=1, env="1") == [(1, True)]

    # the package's queue default: respects an explicit setting, does not touch the environment otherwise on import here
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    assert pkg._hw_queue_default() == 6
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "x")
    assert pkg._hw_queue_default() == 4
    # the opt-in: an explicit setting wins; importing the package never writes the variable
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    assert pkg.request_hw_queues(8) == 6 and os.environ["GPU_MAX_HW_QUEUES"] == "6"
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import avi_talking_amd as p; "
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ and p.HW_QUEUES == 4; "
            "assert p.request_hw_queues(8) == 8 and os.environ['GPU_MAX_HW_QUEUES'] == '8'" % ROOT)
    assert subprocess.run([sys.executable, "-c", code], capture_output=True).returncode == 0


def test_default_plan_is_the_benchmarked_plan():
    """One default everywhere: the host classes, the CLI and bench.py resolve `None` to ops.DEFAULT_PREC."""
    import inspect
    from avi_talking_amd import ops
    from avi_talking_amd.host import cli, faceformer, pipeline, talking_head, wav2vec
    assert ops.DEFAULT_PREC == "mixed" and ops.prec_plan(None).name == "mixed"
    for fn in (pipeline.SamplingPipeline.__init__, talking_head.TalkingHeadWrapper.__init__, wav2vec.Wav2Vec2Model.__init__,
               faceformer.Faceformer.__init__):
        assert inspect.signature(fn).parameters["prec"].default is None
    assert cli.build_parser().parse_args([]).prec is None


def test_sample_dict_builders_match_the_reference():
    """host/sample.py against the reference's own ``process_audio`` / ``create_base_sample`` / ``create_condition`` /
    ``create_high_intensity_emotions`` run on the fixture WAV (tests/golden/sample_dict.npz from make_golden.py::
    gen_sample_dict): same arrays, same dtypes, same names - including what the reference's padding line really does (one extra
    zero FRAME and one extra zero SAMPLE COLUMN at smallest_unit = 1: raw_audio (125, 641))."""
    from avi_talking_amd.host import sample as S
    from avi_talking_amd.host.audio_io import process_audio
    g = np.load(os.path.join(ROOT, "tests", "golden", "sample_dict.npz"))
    wav = g["read_audio"]
    assert wav.dtype == np.int16 and int(g["read_audio_sr"]) == 16000
    fr = process_audio(wav, 16000, 25)
    assert fr["raw_audio"].dtype == g["process_audio_raw"].dtype and np.array_equal(fr["raw_audio"], g["process_audio_raw"])
    base = S.create_base_sample(wav)
    assert base["raw_audio"].shape == (125, 641) and np.array_equal(base["raw_audio"], g["base_raw_audio"])
    rec = base["reconstruction"]["EMICA-MEAD_flame2020"]
    assert [rec["gt_exp"].shape[0], rec["gt_exp"].shape[1], rec["gt_shape"].shape[0], rec["gt_jaw"].shape[1],
            rec["gt_tex"].shape[0]] == list(g["base_rec_shapes"])
    for k in S.CONDITION_KEYS:
        assert base[k].dtype == g["base_" + k].dtype and np.array_equal(base[k], g["base_" + k]), k
    b8 = S.create_base_sample(wav, smallest_unit=8, silent_frames_start=3, silent_frames_end=2)
    assert list(b8["raw_audio"].shape) == list(g["base8_raw_audio_shape"])
    assert float(b8["raw_audio"].astype(np.float64).sum()) == float(g["base8_raw_audio_sum"])
    subjects = [f"M{i:03d}" for i in range(32)]
    hs = S.create_high_intensity_emotions(base, identity_list=[5, 30], emotion_index_list=[3, 6], intensity_list=[2, 0],
                                          training_subjects=subjects)
    for i, h in enumerate(hs):
        for k in S.CONDITION_KEYS:
            assert np.array_equal(h[k], g[f"hi{i}_{k}"]), (i, k)
        assert h["output_name"] == str(g[f"hi{i}_name"])
    hq = S.create_high_intensity_emotions(base, identity_list=[1], emotion_index_list=[4], intensity_list=[1],
                                          silent_frames_start=4, silent_emotion_start=0)
    assert np.array_equal(hq[0]["gt_expression_label_condition"], g["hiq_gt_expression_label_condition"])
    col = S.recursive_collate(hs, device="cpu")
    assert col["raw_audio"].shape == (2, 125, 641) and col["gt_expression_label_condition"].shape == (2, 125, 8)
    assert col["reconstruction"]["EMICA-MEAD_flame2020"]["gt_exp"].shape == (2, 125, 50) and col["output_name"] == ["_M005_Surprise_2", "_M030_Anger_0"]
    with pytest.raises(ValueError):
        S.create_condition({}, emotions=[8])
