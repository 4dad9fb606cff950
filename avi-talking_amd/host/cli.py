"""Command-line surface of the reference entry point (SURVEY.md 8b "CLI"): every flag of
``train_diffusion_prior.py:909-947`` with the same name, type and default, the positional mapping of
``experiments/diffusion_test.sh:1-9`` unchanged (the script only passes flags), and the reference's output layout:
``train_logs/<jobname>/{best,last}.pth`` (:155-168) and
``<run_dir>/test_videos_<save_subdir>/<a>/<b>/{instruction.txt, flame/flame_*.pkl}`` (:758-766,
evaluation_functions.py:624-638).

What the reference pulls from outside the hot path is supplied explicitly here (additional flags, all optional):
  --run_dir          the EMOTE run directory (``talking_head.cfg.inout.full_run_dir``; default ./run)
  --caption_tokens   .npy of CLIP token ids (n, 77) for the captions of --test_json_path (the CLIP tokenizer is host
                     string work and its vocabulary files are not available offline); without it captions get seeded
                     synthetic text features
  --audio_ckpt / --head_ckpt / --clip_ckpt   state_dict files (``torch.load(weights_only=True)``) of wav2vec2, the EMOTE
                     head + FLINT decoder and the CLIP text tower under the reference's key names; absent = seeded
                     random-init weights of the same architectures
  --synthetic_steps  training runs on seeded synthetic (caption feature, style target) batches: the MEAD dataset and
                     its loaders are out of scope
Datasets, video rendering and tensorboard logging are outside the hot path and are not reproduced.
"""
import argparse
import glob
import json
import os
import sys
import time

REFERENCE_TEST_AUDIO = ("/data/yashengsun/local_storage/Mead_emoca/Mead_W/W019_front_angry_level2_007/"
                        "W019_front_angry_level2_007.wav")


def build_parser():
    """train_diffusion_prior.py:909-947, flag for flag."""
    p = argparse.ArgumentParser(description="FaceFormer: Speech-Driven 3D Facial Animation with Transformers")
    p.add_argument("--max_epoch", type=int, default=5000)
    p.add_argument("--epoch", type=int, default=0)
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--clip_size", type=int, default=128)
    p.add_argument("--model_name", type=str, default="EMOTE")
    p.add_argument("--path_to_models", type=str, default=os.path.join("assets", "TalkingHead", "models"))
    p.add_argument("--use_projector", action=argparse.BooleanOptionalAction, default=True)
    p.add_argument("--jobname", type=str, default="text2emo")
    p.add_argument("--save_subdir", type=str, default="")
    p.add_argument("--is_tensorboard_log", type=int, default=1)
    p.add_argument("--is_test", type=int, default=0)
    p.add_argument("--is_talking_instruct", type=int, default=0)
    p.add_argument("--log_loss_steps", type=int, default=5)
    p.add_argument("--resume_from_ckpt", type=int, default=0)
    p.add_argument("--ckpt_path", type=str, default="")
    p.add_argument("--test_audio_path", type=str, default=REFERENCE_TEST_AUDIO)
    p.add_argument("--test_json_path", type=str, default="")
    p.add_argument("--is_output_gt", type=int, default=0)
    p.add_argument("--is_use_rvd", type=int, default=0)
    p.add_argument("--is_cal_diversity", type=int, default=0)
    p.add_argument("--is_vis_diversity", type=int, default=0)
    p.add_argument("--is_no_diffusion", type=int, default=0)
    p.add_argument("--unset_prior", type=int, default=0)
    p.add_argument("--unset_v2c", type=int, default=0)
    p.add_argument("--load_talkclip_dataset", type=int, default=1)
    p.add_argument("--wo_dataset_aug", type=int, default=0)
    p.add_argument("--dataset_names", type=str, default="")
    p.add_argument("--seq_length", type=int, default=25)
    p.add_argument("--vertice_dim", type=int, default=53)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--workers", type=int, default=8)
    p.add_argument("--only_load_caption", type=int, default=1)
    p.add_argument("--max_lr", type=float, default=3e-4)
    x = p.add_argument_group("supplied explicitly here (not reference flags)")
    x.add_argument("--run_dir", type=str, default="run")
    x.add_argument("--caption_tokens", type=str, default="")
    x.add_argument("--audio_ckpt", type=str, default="")
    x.add_argument("--head_ckpt", type=str, default="")
    x.add_argument("--clip_ckpt", type=str, default="")
    x.add_argument("--synthetic_steps", type=int, default=20, help="optimizer steps per epoch on synthetic batches")
    x.add_argument("--device", type=str, default="cuda:0")
    x.add_argument("--prec", type=str, default=None, choices=["mixed", "bf16x3", "mixed_ffn", "f16x2"],
                   help="precision plan of the audio path (default: ops.DEFAULT_PREC = mixed, the plan bench.py's headline is "
                        "measured on: conv layers 2-term fp16, everything else 3-term bf16, 2.5e-4 on the coefficients; "
                        "bf16x3 = 3-term everywhere, 2e-5).  A pass whose fp16 planes leave their range is re-run on bf16x3 "
                        "by itself (SamplingPipeline.run_checked)")
    return p


def load_captions(test_json_path):
    """The caption files of the reference's test loop: a .json file or a directory of them; each holds a dict (or a
    list of dicts) with the caption under "text" / "caption" and the utterance under "audio_path" / "wav"."""
    files = sorted(glob.glob(os.path.join(test_json_path, "*.json"))) if os.path.isdir(test_json_path) else [test_json_path]
    out = []
    for f in files:
        with open(f) as fh:
            d = json.load(fh)
        for e in (d if isinstance(d, list) else [d]):
            out.append((e.get("text", e.get("caption", "")), e.get("audio_path", e.get("wav", ""))))
    return out


def output_folder(run_dir, save_subdir, audio_path):
    """train_diffusion_prior.py:758-760: test_videos_<save_subdir>/<parent of parent>/<parent>."""
    parts = os.path.abspath(audio_path).split("/")
    return os.path.join(run_dir, f"test_videos_{save_subdir}", parts[-3], parts[-2])


def _state(path, make):
    import torch
    return torch.load(path, map_location="cpu", weights_only=True) if path else make()


def run_test(args):
    import numpy as np
    import torch
    from .. import weights as W
    from .audio_io import read_audio
    from .checkpoint import save_flame_pkl
    from .sample import create_base_sample
    from .clip_text import FrozenCLIPEmbedder
    from .pipeline import SamplingPipeline
    dev = torch.device(args.device)
    prior_sd = W.make_prior_weights(3)
    if args.ckpt_path:
        ck = torch.load(args.ckpt_path, map_location="cpu", weights_only=True)
        prior_sd = dict(prior_sd, **ck["model_state_dict"])
    pipe = SamplingPipeline(_state(args.audio_ckpt, lambda: W.make_wav2vec2_weights(0)),
                            _state(args.head_ckpt, lambda: W.make_emote_weights(1)), prior_sd, device=dev, prec=args.prec)
    captions = load_captions(args.test_json_path) if args.test_json_path else [("", args.test_audio_path)]
    tokens = np.load(args.caption_tokens) if args.caption_tokens else None
    clip = FrozenCLIPEmbedder(_state(args.clip_ckpt, lambda: W.make_clip_text_weights(5)), device=dev) \
        if tokens is not None else None
    t0 = time.time()
    for i, (text, audio_path) in enumerate(captions):
        audio_path = audio_path or args.test_audio_path
        wav, sr = read_audio(audio_path)
        # the reference's own framing (trainer(): create_base_sample, train_diffusion_prior.py:695 -> evaluation_functions.py:
        # 141-161), padding quirk included: T + 1 frames of 641 samples, the last frame and the last column zero
        sample = create_base_sample(wav)
        pcm = torch.from_numpy(sample["raw_audio"][None].copy()).to(dev)
        if clip is not None:                     # voxel = mean over the 77 token states (:710-711)
            voxel = clip(torch.from_numpy(tokens[i:i + 1]).to(dev)).mean(1)
        else:
            voxel = torch.randn(1, 768, generator=torch.Generator().manual_seed(1000 + i)).to(dev)
        gen = torch.Generator(device=dev).manual_seed(0)                   # voxel2style_emb seed (:803-804)
        # synchronous, with the safety net: a pass the device reports as untrustworthy (fp16-plane range, paired-sampler
        # timeout) is re-run on the fallback plan / kernel, so what is written below is always a valid result
        out = pipe.run_checked(pcm, voxel, pipe.prior.draw_noise(1, gen))
        if pipe.last_fallback:
            print(f"note: utterance {i} re-run on {pipe.last_fallback}")
        folder = output_folder(args.run_dir, args.save_subdir, audio_path)
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "instruction.txt"), "w") as fh:     # save_text (:773-776)
            fh.write(text)
        name = os.path.basename(audio_path).split(".")[0]
        save_flame_pkl(os.path.join(folder, "flame", f"flame_{name}.pkl"), torch.zeros(300), out["predicted_exp"][0],
                       out["predicted_jaw"][0])
        print(i, text, audio_path)
        print("{:04d} cost {:.3f} s, ave {:.3f} s".format(i, time.time() - t0, (time.time() - t0) / (i + 1)))
    return 0


def run_train(args):
    import torch
    from .. import weights as W
    from . import checkpoint as CK
    from .schedule import cosine_anneal, reference_schedule
    from .training import PriorTrainer, _layout
    dev = torch.device(args.device)
    sd = W.make_prior_weights(3)
    tr = PriorTrainer(sd, device=dev, lr=args.max_lr)
    names = [n for n in _layout() if n in sd]
    epoch0 = CK.resume_ckpt(args.ckpt_path, tr, names) if (args.resume_from_ckpt and args.ckpt_path) else 0
    # the fused trainer's GEMM tiles take batches that are multiples of 64 rows (PriorTrainer: "training batch rows must be a
    # multiple of 64"); the reference's --batch_size default is 1 and its script passes 256
    B = max(args.batch_size, 64) // 64 * 64
    if B != args.batch_size:
        print(f"note: --batch_size {args.batch_size} -> effective batch {B} (the HIP trainer needs multiples of 64 rows); "
              f"the schedule length follows --synthetic_steps ({args.synthetic_steps} steps per epoch) as given")
    steps = args.synthetic_steps
    sched = reference_schedule(args.max_lr, args.max_epoch, steps)
    temps = cosine_anneal(0.004, 0.0075, max(args.max_epoch, 1))        # soft_loss_temps (:375)
    outdir = os.path.abspath(os.path.join("train_logs", args.jobname))
    g = torch.Generator(device=dev).manual_seed(1234 + args.local_rank)
    losses, lrs = [], []
    for epoch in range(epoch0, args.max_epoch):
        for it in range(steps):
            voxel = torch.randn(B, 768, device=dev, generator=g)
            target = torch.randn(B, 1, 128, device=dev, generator=g) * 0.3
            k = min(sched.last_step, sched.total_steps - 1)
            lr = sched.lr_at(k)                # OneCycleLR moves the rate AND AdamW's beta1 (cycle_momentum, :351-357)
            out = tr.train_step(voxel, target, temps[min(epoch, len(temps) - 1)], rand=tr.draw(B, generator=g), lr=lr,
                                beta1=sched.momentum_at(k))
            sched.step()
            if it % args.log_loss_steps == 0:
                lp, ln = float(out["loss_prior"].item()), float(out["loss_nce"].item())
                if not (lp == lp and ln == ln):
                    raise ValueError("NaN loss")                               # check_loss (:135-137)
                losses.append(ln + 30.0 * lp)
                lrs.append(lr)
                print(f"epoch {epoch} it {it} loss {losses[-1]:.4f} lr {lr:.3e}")
        if args.local_rank == 0:
            CK.save_ckpt("last", outdir, epoch, tr, names, sched.state_dict(), losses, [], lrs)
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    from .. import request_hw_queues
    request_hw_queues(8)                 # this program owns its process: before its first CUDA call
    return run_test(args) if args.is_test else run_train(args)


if __name__ == "__main__":
    sys.exit(main())
