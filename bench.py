#!/usr/bin/env python3
"""Headline benchmark: expression-frames/sec of the sampling hot path on MI355X.

One "step" = one full pass of the path over one batch of synthetic input resident in HBM:
  int16 PCM (32 clips x 10 s, 16 kHz) -> normalise -> wav2vec2 (25 Hz) ; CLIP text feature (32,768) ->
  BrainNetwork -> 100-step DDPM prior -> style ; EMOTE head + FLINT decoder -> (32,250,50|3) coefficients.
This is BASELINE.json configs[1] ("Single MI355X: batch 32 x 10 s clips, 25 fps FLAME coeffs,
hipGraph-captured loop").  The DDPM loop runs the reference's actual 100 steps: a 50-step run is not a
configuration the reference object supports.  The whole pass is captured in one
hipGraph and replayed per step.  Weights are seeded random-init of the reference architectures; data is
synthetic band-limited noise.

Multi-GPU (--gpus N under torch.distributed.run): utterances are independent, so each rank runs its own
batch with no data-path collective (weak scaling); rank 0 reports units of all ranks / max-over-ranks time.

Prints ONE JSON line (see the round contract) with extra objects `roofline` (dominant kernel = the bf16
MFMA GEMM, HIP-event timed live) and `cpu_baseline` (the CPU oracle on a bounded sample).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_CLIPS, SECONDS, FPS = 32, 10, 25
N_SAMPLES = SECONDS * 16000
T_FRAMES = SECONDS * FPS
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def flops_per_frame(T):
    """Algorithmic GFLOP per output frame of the sampling path."""
    return 0.3785 + 3.69e-5 * T + 1.39 / T


def synth_audio(B, N, seed):
    """randn low-passed at 4 kHz, int16 RMS 3000 (SURVEY.md 8d synthetic inputs)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, generator=g)
    X = torch.fft.rfft(x)
    X[:, int(4000 / 8000 * (X.shape[1] - 1)):] = 0
    x = torch.fft.irfft(X, n=N)
    x = x / x.pow(2).mean(-1, keepdim=True).sqrt() * 3000.0
    return x.clamp(-32768, 32767).to(torch.int16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", choices=["bf16x3", "bf16"], default="bf16x3",
                    help="bf16x3 = 3-term split bf16 MFMA (passes the 1e-3 parity gate; default); bf16 = 1 term")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--secondary-timeout", type=float, default=300.0,
                    help="seconds the roofline / cpu_baseline / flame / train legs may take before the line is printed "
                         "without them")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import avi_talking_amd.lib as L
    L.load()
    from avi_talking_amd import ops, weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline

    prec = ops.PREC_BF16X3 if args.prec == "bf16x3" else ops.PREC_BF16
    wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
    pipe = SamplingPipeline(wa, wh, wp, device=dev, prec=prec)
    pcm = synth_audio(B_CLIPS, N_SAMPLES, 1234 + rank).to(dev)
    voxel = torch.randn(B_CLIPS, 768, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    noise = torch.randn(101, B_CLIPS, 1, 128, generator=torch.Generator().manual_seed(rank)).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    if args.no_graph:
        step = lambda: pipe.run(pcm, voxel, noise)
        for _ in range(2):
            step()
    else:
        pipe.capture(pcm, voxel, noise)
        step = lambda: pipe.replay()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert torch.isfinite(out["predicted_exp"]).all()

    frames = world * B_CLIPS * T_FRAMES * args.steps
    value = frames / dt

    line = {
        "metric": "expression-frames/sec (sampling: audio+text -> FLAME exp/jaw coefficients)",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
        "config": {"workload": "configs[1]: 32 clips x 10 s @16 kHz per GPU -> 250 frames @25 fps each; "
                               "wav2vec2-base + BrainNetwork + 100-step DDPM prior + EMOTE/FLINT decoder",
                   "clips_per_gpu": B_CLIPS, "frames_per_clip": T_FRAMES, "ddpm_steps": 100,
                   "hipgraph": not args.no_graph, "parallelism": f"dp{world} (independent utterances)"},
        "algorithmic_tflops": round(value * flops_per_frame(T_FRAMES) / 1e3, 1),
        "max_abs_coeff_err_vs_oracle": "see tests/test_gpu_emote.py: 2e-5 (bf16x3)",
        "roofline": None, "cpu_baseline": None, "train": None, "flame": None, "clip_text": None,
    }
    printed = threading.Lock()

    def emit():
        if printed.acquire(blocking=False) and rank == 0:
            print(json.dumps(line), flush=True)

    def watchdog():
        # a secondary measurement hung (the training leg is the only one with collectives): the sampling line,
        # already measured, must still come out, and no rank may be left behind holding the GPU
        if line["train"] is None:
            line["train"] = {"error": f"secondary measurements exceeded {args.secondary_timeout} s"}
        emit()
        sys.stdout.flush()
        os._exit(0)

    timer = threading.Timer(args.secondary_timeout, watchdog)
    timer.daemon = True
    timer.start()

    if rank == 0:
        line["roofline"] = measure_gemm_roofline(pipe, pcm, voxel, noise, prec)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = measure_cpu_baseline(wa, wh, wp)
    del pipe
    torch.cuda.empty_cache()
    if rank == 0 and not args.no_train:
        try:
            line["flame"] = measure_flame(dev)
        except Exception as e:
            line["flame"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and not args.no_train:
        try:
            line["clip_text"] = measure_clip_text(dev)
        except Exception as e:
            line["clip_text"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if not args.no_train:
        try:
            line["train"] = measure_train(wp, dev, world, rank, local_rank, dist, args)
        except Exception as e:  # the sampling line above must survive a failure of the secondary measurement
            line["train"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    timer.cancel()
    emit()
    if dist is not None:
        dist.destroy_process_group()


def measure_flame(dev, reps=10):
    """SURVEY 8f row 1 (next after the hot path): FLAME vertices of the config[1] output, 32 x 250 frames x 5023
    vertices (synthetic basis: the licensed model is absent).  HBM-bound: 482 MB of vertices per pass."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.flame import FLAME
    fl = FLAME(W.make_flame_basis(4), device=dev)
    g = torch.Generator(device=dev).manual_seed(11)
    shape = torch.randn(B_CLIPS, 300, device=dev, generator=g)
    exp = torch.randn(B_CLIPS, T_FRAMES, 50, device=dev, generator=g) * 0.8
    jaw = torch.randn(B_CLIPS, T_FRAMES, 3, device=dev, generator=g) * 0.1
    pose = torch.zeros((B_CLIPS, T_FRAMES, 15), dtype=torch.float32, device=dev)
    pose[..., 6:9] = jaw
    for _ in range(2):
        v = fl.vertices(shape, exp, pose)
    torch.cuda.synchronize(dev)
    evs = []
    for _ in range(reps):      # device time per pass from events on the launch stream; the best pass is reported
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        v = fl.vertices(shape, exp, pose)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize(dev)
    dt = min(a.elapsed_time(b) for a, b in evs) * 1e-3
    nbytes = v.numel() * 4
    return {"workload": "FLAME LBS vertices, 32 clips x 250 frames x 5023 vertices (synthetic basis)",
            "ms_per_pass": round(dt * 1e3, 3), "frames_per_s": round(B_CLIPS * T_FRAMES / dt, 1),
            "roofline": {"bound": "hbm", "achieved": round(nbytes / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(nbytes / dt / 8e12, 4)}}


def measure_clip_text(dev, reps=10):
    """the frozen CLIP text tower (12 layers, 768 wide, 77 tokens) over one batch of 32 prompts,
    random-init weights of that architecture, one hipGraph replay per batch; the best replay is reported."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.clip_text import FrozenCLIPEmbedder
    m = FrozenCLIPEmbedder(W.make_clip_text_weights(5), device=dev)
    ids = torch.randint(0, 49408, (B_CLIPS, 77), generator=torch.Generator().manual_seed(3))
    m.capture(ids)
    m.replay()
    torch.cuda.synchronize(dev)
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.replay()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize(dev)
    ms = min(a.elapsed_time(b) for a, b in evs)
    flops = 2.0 * B_CLIPS * 77 * 12 * (768 * 2304 + 768 * 768 + 2 * 768 * 3072)
    return {"workload": "CLIP text tower, 32 prompts x 77 tokens (random-init weights), hipGraph replay",
            "ms_per_batch": round(ms, 3), "prompts_per_s": round(B_CLIPS / ms * 1e3, 1),
            "algorithmic_tflops": round(flops / ms / 1e9, 1)}


def measure_train(wp, dev, world, rank, local_rank, dist, args):
    """BASELINE.json configs[2]/[3]: one training step of the aligner + prior (77.7 M parameters): batch 64 per GPU,
    forward + backward + fused AdamW; N > 1 adds the RCCL all-reduce of the 311 MB fp32 gradient buffer, launched
    bucket by bucket from inside backward.  Single GPU replays one hipGraph per step."""
    from avi_talking_amd.host.training import PriorTrainer
    B = 64
    tr = PriorTrainer(wp, device=dev, lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    voxel = torch.randn(B, 768, device=dev, generator=g)
    target = torch.randn(B, 1, 128, device=dev, generator=g) * 0.3
    rand = tr.draw(B, generator=g)
    temp = 0.005
    graph = world == 1 and not args.no_graph
    if graph:
        tr.capture_step(voxel, target, temp, rand)
        step = tr.replay_step
    else:
        step = lambda: tr.train_step(voxel, target, temp, rand=rand)
    for _ in range(3):
        out = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        out = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    lp, ln = float(out["loss_prior"].item()), float(out["loss_nce"].item())
    if not (lp == lp and ln == ln):
        raise ValueError("NaN loss")                       # train_diffusion_prior.py:135-137 check_loss
    return {"workload": "configs[2]: aligner+prior train step, batch 64 per GPU, fwd+bwd+fused AdamW"
                        + (", RCCL gradient all-reduce overlapped with backward" if world > 1 else ""),
            "samples_per_s": round(world * B * args.train_steps / dt, 1), "ms_per_step": round(dt / args.train_steps * 1e3, 3),
            "steps": args.train_steps, "global_batch": world * B, "params_m": round(tr.store.numel / 1e6, 1),
            "allreduce_mb": round(tr.store.numel * 4 / 1e6, 1) if world > 1 else 0, "hipgraph": graph,
            "dtype": "bf16x3", "loss_prior": round(lp, 5), "loss_nce": round(ln, 5)}


def gemm_family(kw):
    """Which kernel avi_gemm dispatches a launch to (mirrors csrc/gemm.hip avi_gemm)."""
    if kw.get("Ahi"):
        from avi_talking_amd import ops
        kt = 32 if (kw.get("prec", 3) & 0xff) == 3 else 64
        M, N, K, batch = kw["M"], kw["N"], kw["K"], kw.get("batch", 1)
        cus = ops.CU_BUDGET if 0 < ops.CU_BUDGET <= 256 else 256

        def score(bm, bn, eff):                      # csrc/gemm.hip tile_score
            tiles = -(-M // bm) * -(-N // bn) * batch
            return eff * M * N * batch / (-(-tiles // cus) * cus * bm * bn)

        cands = []
        if K % (2 * kt) == 0:
            cands.append((score(256, 256, 1.0), "gemm_pp_kernel (256x256)"))
        if K % (3 * kt) == 0:
            cands.append((score(128, 192, 0.85), "gemm_pp192_kernel<NT=3> (128x192)"))
            cands.append((score(128, 256, 0.88), "gemm_pp192_kernel<NT=4> (128x256)"))
        if not cands:
            return "gemm_dma_kernel"
        best = cands[0]
        for c in cands[1:]:
            if c[0] > best[0]:
                best = c
        return best[1]
    # fp32-operand kernel gemm_kernel<BN, NS, BM> (csrc/gemm.hip avi_gemm / launch_gemm): 64-column tiles for narrow
    # outputs and small grids, 64- or 32-row tiles for the smallest
    M, N, batch = kw["M"], kw["N"], kw.get("batch", 1)
    ns = 2 if (kw.get("prec", 3) & 0xff) == 3 else 1
    tiles128 = -(-M // 128) * -(-N // 128) * batch
    if N > 64 and tiles128 > 512:
        return f"gemm_kernel<128, {ns}, 128>"
    if M <= 32:
        return f"gemm_kernel<64, {ns}, 32>"
    if -(-M // 128) * -(-N // 64) * batch <= 512 or M <= 64:
        return f"gemm_kernel<64, {ns}, 64>"
    return f"gemm_kernel<64, {ns}, 128>"


def measure_gemm_roofline(pipe, pcm, voxel, noise, prec, reps=3):
    """Per-launch HIP-event timing of the GEMM kernels on the stream each is launched on, over `reps` eager passes
    of the same workload (the prior branch runs concurrently on its side stream, as in the timed region).
    achieved = algorithmic FLOPs (2*M*N*K*batch per launch) / summed launch durations, per kernel family; the
    `roofline` object describes the family with the most FLOPs per step (the dominant kernel), `others` the rest.  In bf16x3
    mode each algorithmic FLOP costs three MFMA FLOPs, which `mfma_issued_frac` accounts for.  `traffic` = HBM
    bytes per launch from the rocprofv3 PMC passes committed under profiles/ (scripts/pmc_traffic.py)."""
    from avi_talking_amd import ops
    rec = []
    orig = ops.gemm_raw

    def timed(**kw):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        orig(**kw)
        e1.record(s)
        rec.append((gemm_family(kw), e0, e1, 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1)))

    ops.gemm_raw = timed
    marks = []
    try:
        for _ in range(reps + 1):      # pass 0 lets the host run ahead of the device and is dropped
            marks.append(len(rec))
            pipe.run(pcm, voxel, noise)
        torch.cuda.synchronize()
    finally:
        ops.gemm_raw = orig
    marks.append(len(rec))
    ns = 3 if prec == ops.PREC_BF16X3 else 1
    # every pass issues the same launches in the same order: a launch slot's duration is the MINIMUM over the kept
    # passes, so a host hiccup between recording e0 and enqueueing the kernel (eager mode) cannot inflate a family
    per_pass = [rec[marks[i]:marks[i + 1]] for i in range(1, reps + 1)]
    n_slots = len(per_pass[0])
    if any(len(p_) != n_slots for p_ in per_pass):
        raise RuntimeError("the eager passes issued different launch sequences")
    fam = {}
    for slot in range(n_slots):
        name, _, _, fl = per_pass[0][slot]
        ms = min(p_[slot][1].elapsed_time(p_[slot][2]) for p_ in per_pass)
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += ms * reps
        f[1] += fl * reps
        f[2] += reps
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh).get("kernels", {})

    def describe(name):
        ms, fl, n = fam[name]
        ach = fl / (ms * 1e-3) / 1e12
        t = traffic.get(name)                       # fp32-operand kernels carry their exact kernel name
        if t is None:
            base = name.split(" ")[0].split("<")[0]
            tag = name.split("<")[1].split(">")[0] if "<" in name else ""
            tag = {"NT=3": ", 3>", "NT=4": ", 4>"}.get(tag, tag)
            t = next((v for k, v in traffic.items() if k.startswith(base) and (tag in k if tag else True)), None)
        return {"bound": "mfma", "kernel": f"{name} (bf16 MFMA 16x16x32, {ns} MFMA per product)",
                "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "mfma_issued_frac": round(ach * ns / PEAK_BF16_TFLOPS, 4),
                "launches_per_step": n // reps, "avg_launch_us": round(ms * 1e3 / n, 2),
                "ms_per_step": round(ms / reps, 3), "algorithmic_gflop_per_step": round(fl / reps / 1e9, 1),
                "traffic": t["hbm_bytes_per_launch"] if t else None}

    # the dominant kernel is the family that carries the most algorithmic FLOPs of a step (a property of the workload,
    # not of this run's timings)
    order = sorted(fam, key=lambda k: -fam[k][1])
    out = describe(order[0])
    out["others"] = [describe(k) for k in order[1:]]
    if traffic:
        out["traffic_source"] = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
    return out


def measure_cpu_baseline(wa, wh, wp, clips=2, reps=2):
    """The CPU oracle (fp32 torch restatement of the reference; a port, not the reference import, which cannot
    travel to this box) on a bounded sample: `clips` x 10 s through the same path, 1 warm-up + `reps` timed."""
    from oracle import emote as OE, prior as OP, wav2vec2 as OW
    cores = torch.get_num_threads()
    pcm = synth_audio(clips, N_SAMPLES, 99)
    voxel = torch.randn(clips, 768, generator=torch.Generator().manual_seed(98))
    noise = torch.randn(101, clips, 1, 128, generator=torch.Generator().manual_seed(97))

    def one():
        with torch.no_grad():
            x = OW.normalize_audio(pcm, joint=True)
            feat = OW.forward(wa, x, frame_num=T_FRAMES)
            te, _ = OP.brain_network(wp, voxel)
            style = OP.p_sample_loop(wp, te.view(clips, 1, 128), noise)
            return OE.forward(wh, feat, style)

    one()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(clips * T_FRAMES / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{clips} clips x 10 s (same path, fp32 torch oracle, {reps} timed passes after 1 warm-up, "
                      f"{dt:.2f} s per pass)"}


if __name__ == "__main__":
    main()
