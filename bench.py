#!/usr/bin/env python3
"""Headline benchmark: expression-frames/sec of the sampling hot path on MI355X.

One "step" = one full pass of the path over one batch of synthetic input resident in HBM:
  int16 PCM (32 clips x 10 s, 16 kHz) -> normalise -> wav2vec2 (25 Hz) ; CLIP text feature (32,768) ->
  BrainNetwork -> 100-step DDPM prior -> style ; EMOTE head + FLINT decoder -> (32,250,50|3) coefficients.
This is BASELINE.json configs[1] ("Single MI355X: batch 32 x 10 s clips, 25 fps FLAME coeffs,
hipGraph-captured loop").  The DDPM loop runs the reference's actual 100 steps: a 50-step run is not a
configuration the reference object supports.  The whole pass - the draw of the DDPM noise
included (device-resident Philox stream, csrc/rng.hip) - is captured in hipGraphs and replayed per step: one graph per
branch (sampler's branch; audio front; the 12 encoder layers as two chains of 16 clips; the EMOTE/FLINT head beside the next
pass), on streams picked by timing on the device (`config.arrangement`, `config.replay_streams` in the line).  Default
precision plan: `mixed` (conv layers on 2 fp16 MFMAs per product, everything else on 3 bf16 MFMAs: 2.5e-4 on the
coefficients, gate 3e-4, north_star 1e-3); `precision_modes` times the other plans, all-3-term `bf16x3` (2e-5)
among them, in the same process.  Weights are seeded random-init of the reference architectures; data is
synthetic band-limited noise.

Multi-GPU: utterances are independent, so each rank runs its own batch with no data-path collective (weak
scaling); rank 0 reports units of all ranks / max-over-ranks time.  Ranks come from torch.distributed.run (the driver's
command line: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when `--gpus N` is given without WORLD_SIZE,
from this script itself: the parent spawns N children (one per GPU, 127.0.0.1 rendezvous) BEFORE any GPU call and
exits with the worst child's code.  `--dry-run` runs the same launcher, rendezvous, barrier and max-over-ranks timing
on the CPU over gloo with a stand-in step (tests/test_dist_gloo.py): no GPU, no HIP library.

Prints ONE JSON line (see the round contract) with extra objects `roofline` (the matrix-core GEMM family that does most of
the pass's arithmetic, every launch HIP-event timed live on its own stream; the other families, the HBM-bound launches and
the sampler under `others`) and `cpu_baseline` (the CPU oracle on a bounded sample), plus secondary legs: precision_modes,
train, faceformer, longform, flame, clip_text.
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
import avi_talking_amd  # noqa: E402

# before the first CUDA call: this program owns its process and asks the runtime for 8 hardware queues (an explicit opt-in;
# the line records what it got under config.hw_queues)
avi_talking_amd.request_hw_queues(8)

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


def launch_ranks(n, argv, timeout, script=None):
    """`python bench.py --gpus N` without a launcher: spawn N rank processes of this script (`script`: another program to
    run as the ranks - the launcher's own tests).  The parent makes no GPU
    call (a process that has initialised the GPU must not be replaced or forked), waits for all children, kills the
    others by PID when one fails or the timeout expires, and returns the worst exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool supports dmabuf IPC only; with the legacy mode RCCL's
        # (and torch's) cross-process buffer sharing fails in hipIpcGetMemHandle ("invalid argument").  The image exports
        # it already; it is set here so that ranks spawned from a scrubbed environment still get it.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env))
    deadline = time.monotonic() + timeout
    rc = 0
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is not None:
                procs.remove(p)
                if code != 0:
                    rc = rc or code
        if procs and (rc != 0 or time.monotonic() > deadline):
            rc = rc or 124
            for p in procs:
                p.kill()
            for p in procs:
                p.wait()
            break
        time.sleep(0.05)
    return rc


def dry_run(args, world, rank):
    """The N-rank scaffolding on the CPU (gloo): rendezvous, barrier, EXACTLY `steps` timed stand-in steps, max over
    ranks, the gradient-span all-reduce of the training leg, one JSON line from rank 0.  The stand-in step is the CPU
    oracle's EMOTE head on a tiny batch (the checker used as a placeholder workload; nothing here is a measurement)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.sharding import max_over_ranks
    from avi_talking_amd.host.training import FlatLayout, GradSync, _layout, grad_spans
    from oracle import emote as OE
    torch.set_num_threads(1)
    wh = W.make_emote_weights(1)
    g = torch.Generator().manual_seed(1234 + rank)
    feat, style = torch.randn(2, 16, 768, generator=g), torch.randn(2, 1, 128, generator=g)
    step = lambda: OE.forward(wh, feat, style)
    bar = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    for _ in range(args.warmup):
        step()
    bar()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    bar()
    dt = max_over_ranks(time.perf_counter() - t0, torch.device("cpu"), dist if world > 1 else None)
    # training leg: every gradient span announced in backward order and reduced once over the ranks
    lay = FlatLayout.of_state_dict(W.make_prior_weights(3), _layout())
    sync = GradSync(lay)
    G = torch.full((lay.numel,), float(rank + 1))
    for a, b in grad_spans():
        sync.ready(G, a, b)
    wsz = sync.finish(G)
    ok = bool((G == sum(range(1, world + 1))).all()) and wsz == world
    if rank == 0:
        print(json.dumps({"metric": "dry-run (CPU/gloo stand-in, not a measurement)", "dry_run": True,
                          "value": round(world * 2 * 16 * args.steps / dt, 1), "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "scaling": "weak", "grad_spans_reduced_once": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # 0.6 s of timed region: the clock settles within the first passes
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prec", choices=["bf16x3", "bf16", "f16x2", "mixed", "mixed_ffn"], default=None,
                    help="default: ops.DEFAULT_PREC, the product's default plan (SamplingPipeline, host/cli.py).  mixed = conv layers 1-6 (half of the FLOPs) in 2-term fp16 (fp16 hi/lo activation "
                         "planes x one fp16 weight plane), transformer projections, heads and sampler in 3-term bf16: 2.5e-4 max-abs "
                         "on the coefficients vs the oracle on every tested config, gated at 3e-4 (north_star: 1e-3; the reference "
                         "itself runs fp16 autocast); bf16x3 = 3-term split bf16 everywhere (2e-5); mixed_ffn = conv and ffn 2-term "
                         "(3.5-4.7e-4); f16x2 = every plane-operand GEMM and the sampler 2-term fp16 (6-8e-4); bf16 = 1 term (fails "
                         "the gate).  Every mode is timed by the precision_modes leg")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="replay the pass as ONE graph, passes strictly one after the other (default: two graphs on two "
                         "streams, the head of a pass runs beside the start of the next; the aligner opens the sampler's branch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-kernel event timing (roofline object)")
    ap.add_argument("--train-steps", type=int, default=20)
    ap.add_argument("--legs", default="all", help="comma-separated secondary legs to run (roofline, cpu_baseline, precision_modes, "
                                                    "faceformer, longform, flame, clip_text, train); default: all")
    ap.add_argument("--secondary-timeout", type=float, default=300.0,
                    help="seconds the roofline / cpu_baseline / flame / train legs may take before the line is printed "
                         "without them (the process then exits with code 3)")
    ap.add_argument("--joint-norm", action="store_true",
                    help="audio statistics over the whole batch (the HF processor quirk of AudioEncoders.py:170-178) "
                         "instead of per clip (what the reference's batch-1 loop computes; default)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the N-rank scaffolding (no GPU)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the spawned ranks may take")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing above or in this branch touches the GPU.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # AVI_BENCH_ONE_GPU_REHEARSAL=1: every rank on GPU 0 with gloo as the transport (RCCL refuses two ranks on one device):
    # the whole N-rank flow of this script - spawn, barriers, the max-reduce, rank-0-only legs while the other ranks wait in
    # the training leg's collectives, the DP step with real inter-process all-reduces - on a one-GPU box.  The numbers of such
    # a run mean nothing (the ranks share the card); the line says so.
    rehearsal = os.environ.get("AVI_BENCH_ONE_GPU_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.dry_run:
        raise SystemExit(dry_run(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # AVI_BENCH_FORCE_DIST=1: take the RCCL code path (init, barriers, all-reduces, the trainer's gradient buckets) with
    # a single rank too - the rehearsal a one-GPU box allows
    if world > 1 or os.environ.get("AVI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    import avi_talking_amd.lib as L
    L.load()
    from avi_talking_amd import ops, weights as W
    from avi_talking_amd.host.pipeline import SamplingPipeline

    if args.prec is None:
        args.prec = ops.DEFAULT_PREC         # one default everywhere: the plan the product's classes and CLI use
    prec = ops.prec_plan(args.prec)
    wa, wh, wp = W.make_wav2vec2_weights(0), W.make_emote_weights(1), W.make_prior_weights(3)
    # the DDPM noise of every timed pass is drawn INSIDE the pass (graph nodes, device-resident Philox stream), as the
    # reference draws it inside its loop (models/diffusion_prior.py:337,349-351); `noise` below is the recorded tensor the
    # parity and per-kernel legs inject
    pipe = SamplingPipeline(wa, wh, wp, device=dev, prec=prec, joint_norm=args.joint_norm, rng_seed=4242 + rank)
    pcm = synth_audio(B_CLIPS, N_SAMPLES, 1234 + rank).to(dev)
    voxel = torch.randn(B_CLIPS, 768, generator=torch.Generator().manual_seed(1235 + rank)).to(dev)
    noise = torch.randn(101, B_CLIPS, 1, 128, generator=torch.Generator().manual_seed(rank)).to(dev)

    def barrier():
        if dist is not None:
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])

    if args.no_graph:
        step = lambda: pipe.run(pcm, voxel, None)
        for _ in range(2):
            step()
    elif args.no_pipeline:
        pipe.capture(pcm, voxel, None)
        step = lambda: pipe.replay()
    else:
        pipe.capture_pipelined(pcm, voxel, None)
        step = lambda: pipe.replay_pipelined()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    def marker():
        """AVI_BENCH_MARKERS=1 (profiling runs): one `where_kernel` launch in front of and one behind the timed region, so
        that scripts/trace_by_shape.py keeps exactly the timed passes of a kernel trace (outside the clock: each is followed
        by a synchronisation)."""
        if os.environ.get("AVI_BENCH_MARKERS") == "1":
            mk = torch.zeros(1, dtype=torch.int32, device=dev)
            L.check(L.load().avi_debug_where(mk.data_ptr(), 1, 64, 0, 0, L.stream_ptr()), "avi_debug_where")
            torch.cuda.synchronize(dev)

    marker()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    marker()
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
        **({"rehearsal": "AVI_BENCH_ONE_GPU_REHEARSAL: all ranks share ONE GPU over gloo - a test of the N-rank flow, NOT a measurement"}
           if rehearsal else {}),
        "config": {"workload": "configs[1]: 32 clips x 10 s @16 kHz per GPU -> 250 frames @25 fps each; "
                               "wav2vec2-base + BrainNetwork + 100-step DDPM prior + EMOTE/FLINT decoder",
                   "clips_per_gpu": B_CLIPS, "frames_per_clip": T_FRAMES, "ddpm_steps": 100,
                   "audio_normalisation": "joint over the batch" if args.joint_norm else "per clip",
                   "hipgraph": not args.no_graph, "precision_plan": repr(prec),
                   "precision_plan_is_product_default": args.prec == ops.DEFAULT_PREC,
                   "hw_queues": avi_talking_amd.HW_QUEUES,
                   "random_draws": "in the pass: (101,B,1,128) DDPM noise from the library's Philox-4x32-10 stream, drawn by "
                                   "nodes of the captured graph (fresh at every replay)",
                   "replay": "eager" if args.no_graph else "one graph per pass" if args.no_pipeline else
                             "one graph per branch (sampler's branch, audio front, one per encoder chain, head) on streams of their "
                             "own, tied by events: the head of pass k runs beside the start of pass k+1; every pass does all of "
                             "its work, results bit-identical to the one-graph replay (roofline.branches has the timeline)",
                   "parallelism": f"dp{world} (independent utterances)"},
        "algorithmic_tflops": round(value * flops_per_frame(T_FRAMES) / 1e3, 1),
        # MEASURED in this run (measure_parity below): the same pipeline object against the CPU oracle on one clip
        "max_abs_coeff_err_vs_oracle": None, "parity": None,
        "roofline": None, "cpu_baseline": None, "precision_modes": None, "train": None, "faceformer": None, "longform": None, "flame": None,
        "clip_text": None,
    }
    printed = threading.Lock()
    leg = {"name": "roofline"}

    def emit():
        if printed.acquire(blocking=False) and rank == 0:
            print(json.dumps(line), flush=True)

    def watchdog():
        # a secondary measurement hung: the sampling line, already measured, still comes out, with the name of the leg
        # that was running, and the process ends with a NON-ZERO code so that the hang is on record (no rank may be
        # left behind holding the GPU; os._exit because a hung HIP call cannot be interrupted)
        line["secondary_error"] = f"leg '{leg['name']}' exceeded --secondary-timeout {args.secondary_timeout} s"
        emit()
        sys.stdout.flush()
        os._exit(3)

    timer = threading.Timer(args.secondary_timeout, watchdog)
    timer.daemon = True
    timer.start()

    legs = None if args.legs == "all" else set(args.legs.split(","))

    def run_leg(name, fn, only_rank0=True):
        if (only_rank0 and rank != 0) or (legs is not None and name not in legs):
            return
        leg["name"] = name
        try:
            line[name] = fn()
        except Exception as e:  # the sampling line above must survive a failure of a secondary measurement
            line[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if not args.no_roofline:
        run_leg("roofline", lambda: measure_roofline(pipe, pcm, voxel, noise, line["ms_per_step"],
                                                     pipelined=not (args.no_graph or args.no_pipeline)))
    oracle_case = {}
    if world == 1 and not args.no_cpu_baseline:
        run_leg("cpu_baseline", lambda: measure_cpu_baseline(wa, wh, wp, args.joint_norm, keep=oracle_case))
    run_leg("parity", lambda: measure_parity(pipe, wa, wh, wp, dev, args.joint_norm, oracle_case))
    if isinstance(line.get("parity"), dict) and "max_abs_coeff_err" in line["parity"]:
        line["max_abs_coeff_err_vs_oracle"] = line["parity"]["max_abs_coeff_err"]
    o_ref = pipe.run(pcm, voxel, noise)                      # recorded noise: the reference point of the parity diffs
    torch.cuda.synchronize(dev)
    ref_out = {k: o_ref[k].clone() for k in ("predicted_exp", "predicted_jaw")}
    del o_ref
    try:                         # after the synchronisations above: no paired-sampler launch gave up on its partner
        pipe.check()
        line["config"]["paired_sampler_status"] = "ok" if pipe.prior.uses_pairs(B_CLIPS) else "not used"
    except RuntimeError as e:    # the line stays printable, but says that its passes are not valid
        line["config"]["paired_sampler_status"] = f"FAILED: {e}"
        line["error"] = str(e)
    line["config"]["replay_streams"] = getattr(pipe, "stream_choice", None)
    line["config"]["arrangement"] = getattr(pipe, "arrangement", None)      # encoder chains x sampler kernel, timed at capture
    del pipe, out
    torch.cuda.empty_cache()
    if not args.no_train and world == 1:      # N = 1 only, like the CPU baseline
        run_leg("precision_modes", lambda: measure_precision_modes(wa, wh, wp, dev, pcm, voxel, noise, ref_out, args.prec, args))
        torch.cuda.empty_cache()
    if not args.no_train:
        run_leg("faceformer", lambda: measure_faceformer(dev))
        run_leg("longform", lambda: measure_longform(wa, wh, wp, dev, prec))
        run_leg("flame", lambda: measure_flame(dev))
        run_leg("clip_text", lambda: measure_clip_text(dev))
        run_leg("train", lambda: measure_train(wp, dev, world, rank, local_rank, dist, args), only_rank0=False)
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
    traffic = None
    tp = latest_profile("pmc_flame_traffic.json")       # rocprofv3 FETCH_SIZE / WRITE_SIZE passes over scripts/time_flame_mc.py
    if tp:
        with open(tp) as fh:
            ks = json.load(fh).get("kernels", {})
        parts = [x["hbm_bytes_per_launch"] for k, x in ks.items() if k.startswith(("flame_vertices", "flame_prep"))]
        traffic = int(sum(parts)) if parts else None
    return {"workload": "FLAME LBS vertices, 32 clips x 250 frames x 5023 vertices (synthetic basis)",
            "ms_per_pass": round(dt * 1e3, 3), "frames_per_s": round(B_CLIPS * T_FRAMES / dt, 1),
            "roofline": {"bound": "hbm", "achieved": round(nbytes / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(nbytes / dt / 8e12, 4), "algorithmic_bytes": nbytes, "traffic": traffic,
                         "traffic_note": "HBM-side bytes of the pass's two launches from the PMC passes (profiles/): the output "
                                         "plus the basis fragments fetched again for every clip (they do not fit an XCD's L2)"}}


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
    from avi_talking_amd.host.rng import DeviceRng
    # times / noise / cond-drop masks / dropout masks are drawn inside the captured step (train_diffusion_prior.py:449)
    if args.no_graph:
        graph = False
        step = lambda: tr.train_step(voxel, target, temp, rand=rand)
    elif dist is None:
        graph = "one graph"
        tr.capture_step(voxel, target, temp, rng=DeviceRng(4321 + rank, dev))
        step = tr.replay_step
    else:
        # data parallel: hipGraph segments cut at the gradient-bucket announcements, the RCCL all-reduces issued eagerly
        # between them (collectives stay outside the graphs), one fused-AdamW launch per bucket as its sum arrives
        graph = "segments (7 graphs) + eager RCCL all-reduce per bucket + per-bucket AdamW"
        try:
            tr.capture_step_dp(voxel, target, temp, rng=DeviceRng(4321 + rank, dev))
            step = tr.replay_step_dp
        except Exception as e:      # every rank runs the same code on the same shapes: a capture error is the same on all of
            graph = f"eager (segment capture failed: {type(e).__name__}: {str(e)[:120]})"      # them, and so is the fallback
            tr = PriorTrainer(wp, device=dev, lr=1e-4)
            step = lambda: tr.train_step(voxel, target, temp, rand=rand)
    for _ in range(3):
        out = step()
    torch.cuda.synchronize(dev)
    bar = lambda: (dist.barrier() if os.environ.get("AVI_BENCH_ONE_GPU_REHEARSAL") == "1" else dist.barrier(device_ids=[local_rank]))
    if dist is not None:
        bar()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        out = step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        bar()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    lp, ln = float(out["loss_prior"].item()), float(out["loss_nce"].item())
    if not (lp == lp and ln == ln):
        raise ValueError("NaN loss")                       # train_diffusion_prior.py:135-137 check_loss
    ms_step = dt / args.train_steps * 1e3
    roof = None
    if rank == 0:
        try:
            roof = train_roofline(PriorTrainer(wp, device=dev, lr=1e-4) if graph and dist is not None else tr, voxel, target,
                                  temp, rand, ms_step)
        except Exception as e:
            roof = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"workload": "configs[2]: aligner+prior train step, batch 64 per GPU, fwd+bwd+fused AdamW"
                        + (", RCCL gradient all-reduce overlapped with backward" if world > 1 else ""),
            "samples_per_s": round(world * B * args.train_steps / dt, 1), "ms_per_step": round(ms_step, 3),
            "steps": args.train_steps, "global_batch": world * B, "params_m": round(tr.store.numel / 1e6, 1),
            "allreduce_mb": round(tr.store.numel * 4 / 1e6, 1) if world > 1 else 0, "hipgraph": graph,
            "random_draws": "inside the captured step (Philox stream: times, noise, cond-drop and dropout masks)" if graph
                            else "recorded tensors",
            "gradient_buckets": [f"{a}..{b}" for a, b in __import__("avi_talking_amd.host.training", fromlist=["x"]).grad_spans()] if world > 1 else None,
            "dtype": "bf16x3", "loss_prior": round(lp, 5), "loss_nce": round(ln, 5),
            "optimizer_schedule": (("sharded over the ranks (reduce-scatter -> AdamW on the rank's 1/world slice -> all-gather)"
                                    if tr.sync.shard else "all-reduce, every rank updates everything") if dist is not None
                                   else "single GPU: one fused AdamW over the flat buffers"),
            "roofline": roof}


def train_roofline(tr, voxel, target, temp, rand, step_ms, reps=3):
    """Roofline of the TRAINING half of the metric (~31 GFLOP of GEMMs + 28 B per parameter of optimizer
    traffic per step at B = 64).  Event-timed on the launch stream, kernel by kernel, in eager passes of the same step:
      * fused AdamW (`adamw_kernel`, HBM bound): algorithmic bytes = 28 B x parameters (fp32 p, m, v read + written, g read;
        the kernel also writes the 4 B of bf16 planes per parameter, not counted) / its launch time (lr = 0 so the timed
        launches leave the model alone);
      * every GEMM launch of forward + backward (MFMA bound): 2 M N K batch / time, all launches together and the aligner's
        4096 x 4096 layers (97 % of the FLOPs) on their own;
      * the step against its HBM floor: optimizer traffic + the bf16 planes of every matrix read for the forward, the dX and
        (as transposed planes rebuilt each step: read fp32, write planes) pass + the gradient written once."""
    from avi_talking_amd import ops
    S = tr.store
    n = S.numel
    evs = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tr._adamw_span(0, S.n_decay, 1, lr=0.0)
        tr._adamw_span(S.n_decay, n, 1, lr=0.0)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    adam_ms = min(a.elapsed_time(b) for a, b in evs[1:])
    adam_bytes = 28.0 * n
    rec, marks = [], []
    orig = ops.gemm_raw

    def timed(**kw):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        orig(**kw)
        e1.record(s)
        rec.append((e0, e1, kw["M"], kw["N"], kw["K"], kw.get("batch", 1)))
    ops.gemm_raw = timed
    tr.sync._collectives = lambda: False          # rank 0 alone runs this leg: bookkeeping only, no collective is issued
    try:
        for _ in range(reps + 1):
            marks.append(len(rec))
            tr.forward_backward(voxel, target, rand["times"], rand["noise"], temp, rand["brain_keep"], rand["image_keep"],
                                rand["dropout_masks"])
            tr.sync.finish(S.G, launch=False)
        torch.cuda.synchronize()
    finally:
        ops.gemm_raw = orig
        del tr.sync._collectives
    marks.append(len(rec))
    passes = [rec[marks[i]:marks[i + 1]] for i in range(1, reps + 1)]
    slots = len(passes[0])
    if any(len(p) != slots for p in passes):
        raise RuntimeError("the eager training passes issued different launch sequences")
    tot_ms = tot_fl = big_ms = big_fl = 0.0
    for k in range(slots):
        _, _, M, N, K, bt = passes[0][k]
        ms = min(p[k][0].elapsed_time(p[k][1]) for p in passes)
        fl = 2.0 * M * N * K * bt
        tot_ms, tot_fl = tot_ms + ms, tot_fl + fl
        if max(M, N, K * bt) >= 4096 and min(max(M, N), max(N, K * bt), max(M, K * bt)) >= 4096:      # a 4096 x 4096 layer
            big_ms, big_fl = big_ms + ms, big_fl + fl
    adam_traffic = None
    tp = latest_profile("pmc_train_traffic.json")      # rocprofv3 FETCH_SIZE / WRITE_SIZE passes over `bench.py --legs train`
    if tp:
        with open(tp) as fh:
            shapes_ = json.load(fh).get("by_shape", {})
        big = [v for k, v in shapes_.items() if k.startswith("adamw_kernel")]
        if big:                                         # the decay region's launch (75 M of the 77.8 M parameters)
            adam_traffic = max(v["hbm_bytes_per_launch"] for v in big)
    floor_bytes = adam_bytes + n * (4.0 + 4.0 + 8.0) + n * 4.0
    floor_ms = floor_bytes / 8e12 * 1e3
    gemm = lambda name, fl, ms, cnt: {
        "bound": "mfma", "kernel": name, "launches_per_step": cnt, "ms_per_step": round(ms, 3),
        "algorithmic_gflop_per_step": round(fl / 1e9, 1), "achieved": round(fl / ms / 1e9, 1), "peak": PEAK_BF16_TFLOPS,
        "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4), "mfma_per_product": 3,
        "mfma_issued_frac": round(3 * fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4),
        "note": "batch-64 GEMMs: M or K is 64, so a launch streams its whole weight matrix for 64 rows of work - bound by "
                "the bytes of the bf16 planes, not by the matrix cores"}
    return {"bound": "hbm", "kernel": "adamw_kernel (fused AdamW over the flat fp32 buffers, emits the bf16 hi/lo planes)",
            "launches_per_step": 2, "avg_launch_us": round(adam_ms * 1e3 / 2, 1), "ms_per_step": round(adam_ms, 3),
            "frac_of_step": round(adam_ms / step_ms, 3), "algorithmic_bytes_per_launch": int(adam_bytes / 2),
            "achieved": round(adam_bytes / adam_ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(adam_bytes / adam_ms / 1e6 / 8000.0, 4), "traffic": adam_traffic,
            "traffic_note": "HBM bytes of the decay region's launch from the PMC passes (profiles/), which also writes the 4 B of "
                            "bf16 planes per parameter that the algorithmic 28 B do not count",
            "others": [gemm("all GEMM launches of forward + backward (gemm_kernel family, 3-term bf16)", tot_fl, tot_ms, slots),
                       gemm("of these: the aligner's 4096 x 4096 layers (forward split-K, dX, dW)", big_fl, big_ms, None)],
            "step_hbm_floor": {"bytes": int(floor_bytes), "ms_at_8_TB_s": round(floor_ms, 3),
                               "step_ms": round(step_ms, 3), "floor_over_step": round(floor_ms / step_ms, 3),
                               "what": "28 B/param optimizer + bf16 planes read for forward and dX (4 + 4 B/param) + transposed "
                                       "planes rebuilt (4 B read + 4 B written per param) + gradient written once (4 B/param)"}}


def gemm_family(kw):
    """Which kernel avi_gemm dispatches a launch to (mirrors csrc/gemm.hip avi_gemm)."""
    if kw.get("Ahi"):
        from avi_talking_amd import ops
        kt = 64 if (kw.get("prec", 3) & 0xff) == 1 else 32
        M, N, K, batch = kw["M"], kw["N"], kw["K"], kw.get("batch", 1)
        cus = kw.get("cus", 0) if 0 < kw.get("cus", 0) <= 256 else 256

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


NS_OF_PREC = {1: 1, 2: 2, 3: 3}                     # MFMAs per product: AVI_PREC_BF16 / F16X2 / BF16X3
PREC_NAME = {1: "bf16", 2: "f16x2", 3: "bf16x3"}


def latest_profile(stem):
    """profiles/rNN_<stem>: the newest round's file."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{stem}")))
    return c[-1] if c else None


def measure_parity(pipe, wa, wh, wp, dev, joint_norm, case):
    """The second half of BASELINE's metric, measured in this run: max-abs difference of the un-normalised coefficients
    between THIS pipeline object (the plan and kernels that were just timed) and the CPU oracle on the same (audio, text
    feature, DDPM noise).  The clip is the one the cpu_baseline leg ran through the oracle (1 x 10 s; `case` holds its
    inputs and the oracle's outputs); without that leg a 2 s clip goes through the oracle here."""
    from oracle import emote as OE, prior as OP, wav2vec2 as OW
    if not case:
        T = 50
        case["pcm"] = synth_audio(1, T * 640, 99)
        case["voxel"] = torch.randn(1, 768, generator=torch.Generator().manual_seed(98))
        case["noise"] = torch.randn(101, 1, 1, 128, generator=torch.Generator().manual_seed(97))
        with torch.no_grad():
            feat = OW.forward(wa, OW.normalize_audio(case["pcm"], joint=joint_norm), frame_num=T)
            te, _ = OP.brain_network(wp, case["voxel"])
            case["style"] = OP.p_sample_loop(wp, te.view(1, 1, 128), case["noise"])
            case["ref"] = OE.forward(wh, feat, case["style"])
    pcm = case["pcm"]
    out = pipe.run(pcm.to(dev), case["voxel"].to(dev), case["noise"].to(dev))
    torch.cuda.synchronize(dev)
    pipe.check()
    ref = case["ref"]
    e_exp = (out["predicted_exp"].cpu() - ref["predicted_exp"]).abs().max().item()
    e_jaw = (out["predicted_jaw"].cpu() - ref["predicted_jaw"]).abs().max().item()
    e_style = (out["style_emb"].cpu() - case["style"]).abs().max().item()
    gate = {"mixed": 3e-4, "bf16x3": 5e-5}.get(pipe.plan.name, 1e-3)
    return {"max_abs_coeff_err": float("%.3e" % max(e_exp, e_jaw)), "exp": float("%.3e" % e_exp), "jaw": float("%.3e" % e_jaw),
            "style_emb": float("%.3e" % e_style), "plan": pipe.plan.name, "plan_gate": gate, "north_star_gate": 1e-3,
            "within_plan_gate": max(e_exp, e_jaw) < gate,
            "case": f"1 clip x {pcm.shape[1] / 16000:g} s (seed 99), 100-step DDPM with recorded noise, HIP pipeline vs the fp32 CPU "
                    f"oracle, un-normalised expression(50) + jaw(3) coefficients"}


def measure_roofline(pipe, pcm, voxel, noise, step_ms, reps=3, pipelined=False):
    """Per-launch HIP-event timing, on the stream each kernel is launched on, of the kernels that make up the step:
    every GEMM launch, the other launches of the audio branch that take >= 2 % of the step (conv layer 0, positional conv,
    encoder attention, LayerNorm; launch stream) and the one-launch DDPM sampler (side stream, concurrent with the audio
    branch as in the timed region), over `reps` eager passes of the same workload.
      GEMM families (kernel x precision): achieved = algorithmic FLOPs (2*M*N*K*batch per launch) / summed launch
        durations; each algorithmic FLOP costs `mfma_per_product` MFMA FLOPs (`mfma_issued_frac`).
      Attention / positional conv: algorithmic FLOPs of the launch / duration (MFMA bound; bf16x3 = 3 MFMA per product).
      conv layer 0, LayerNorm: algorithmic bytes (input read + every output written once) / duration (HBM bound).
      Sampler: bytes of weight planes every workgroup streams through its CU (planes x DDPM steps x workgroups; served
        by L2 / Infinity Cache, hence above the HBM-side `traffic`) / launch duration.
    The `roofline` object is the family with the largest time per step ON THE BRANCH THAT BOUNDS THE STEP (`frac_of_step` =
    its time / the measured step; the sampler runs BESIDE the audio branch and leads only when its launch fills the
    step), `others` the rest; `audio_branch_ms_accounted` = the sum over the audio branch's entries.  `traffic` = HBM
    bytes per launch and `mfma_busy_pmc` from the newest rocprofv3 PMC passes under profiles/."""
    from avi_talking_amd import ops
    rec, srec = [], []
    prior = pipe.prior
    orig_sample = prior.p_sample_loop
    saved = {}

    def ev_pair():
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        return s, e0, e1

    def wrap(name, work):
        """work(args, kwargs, result) -> (family, bound, units: FLOPs or bytes, mfma per product or None)"""
        fn = saved[name] = getattr(ops, name)

        def timed(*a, **kw):
            s, e0, e1 = ev_pair()
            out = fn(*a, **kw)
            e1.record(s)
            rec.append((e0, e1) + work(a, kw, out))
            return out
        setattr(ops, name, timed)

    shape_of = {}

    def gemm_work(a, kw, out):
        p = kw.get("prec", 3) & 0xff
        shape_of[len(rec)] = (kw["M"], kw["N"], kw["K"], kw.get("batch", 1))      # this launch's slot in `rec`
        return (f"{gemm_family(kw)} [{PREC_NAME[p]}]", "mfma", 2.0 * kw["M"] * kw["N"] * kw["K"] * kw.get("batch", 1),
                NS_OF_PREC[p])

    def attn_work(a, kw, out):
        qkv, H = a[0], a[1]
        B, T, _ = qkv.shape
        return ("attn_fused_kernel<64> (encoder attention, softmax on the vector pipe)", "mfma", 4.0 * B * H * T * T * 64, 3)

    def posconv_work(a, kw, out):
        x, groups, taps = a[0], a[3], a[4]
        B, T, C_ = x.shape
        return ("posconv_kernel (grouped k=128 conv + bias + GELU + residual)", "mfma",
                2.0 * B * T * C_ * taps * (C_ // groups), 3)

    def conv0_work(a, kw, out):
        x = a[0]
        return ("conv0 (stats + moments + conv layer 0 / GroupNorm / GELU -> planes)", "hbm",
                float(x.numel() * 4 + out.hi.numel() * 4), None)

    def ln_work(a, kw, out):
        x = a[0]
        n = x.numel()
        return ("layernorm_kernel (fp32 in; fp32 + planes out)", "hbm", float(n * (4 + 4 + (4 if kw.get("want_f32", True) else 0))), None)

    def interp_work(a, kw, out):
        x = a[0]
        nin = x.numel() if torch.is_tensor(x) else x.hi.numel()
        return ("interp_ln_kernel (50 -> 25 Hz + LayerNorm)", "hbm", float(nin * 4 + out.numel() * 4), None)

    def timed_sample(*a, **kw):
        s, e0, e1 = ev_pair()                        # the pipeline's side stream
        out = orig_sample(*a, **kw)
        e1.record(s)
        srec.append((e0, e1))
        return out

    arec = []
    orig_aligner = prior.voxel2clip

    def timed_aligner(*a, **kw):                     # the aligner opens the sampler's branch (text feature -> text_embed)
        s, e0, e1 = ev_pair()
        out = orig_aligner(*a, **kw)
        e1.record(s)
        arec.append((e0, e1))
        return out

    marks = []
    # The pass runs the 12 encoder layers as chains of clips on separate streams (host/wav2vec._encoder_layers_split): here
    # the chains keep their launch shapes (M = clips per chain x T) but run one after the other on the launch stream, so
    # that an event pair brackets ONE kernel beside the sampler and not two overlapping ones
    am = pipe.talking_head.audio_model
    chain_streams, am._split_streams = am._split_streams, [torch.cuda.current_stream(pipe.device)] * 8
    try:
        wrap("gemm_raw", gemm_work)
        wrap("attention_d64_planes", attn_work)
        wrap("posconv_gelu_residual", posconv_work)
        wrap("conv0_gn_gelu_planes", conv0_work)
        wrap("layernorm_planes", ln_work)
        wrap("interp_layernorm", interp_work)
        prior.p_sample_loop = timed_sample
        prior.voxel2clip = timed_aligner
        for _ in range(reps + 1):      # pass 0 lets the host run ahead of the device and is dropped
            marks.append(len(rec))
            pipe.run(pcm, voxel, noise)
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
        prior.p_sample_loop = orig_sample
        prior.voxel2clip = orig_aligner
        am._split_streams = chain_streams
    marks.append(len(rec))
    # every pass issues the same launches in the same order: a launch slot's duration is the MINIMUM over the kept
    # passes, so a host hiccup between recording e0 and enqueueing the kernel (eager mode) cannot inflate a family
    per_pass = [rec[marks[i]:marks[i + 1]] for i in range(1, reps + 1)]
    n_slots = len(per_pass[0])
    if any(len(p_) != n_slots for p_ in per_pass):
        raise RuntimeError("the eager passes issued different launch sequences")
    fam, shapes = {}, {}
    for slot in range(n_slots):
        _, _, name, bound, units, ns = per_pass[0][slot]
        ms = min(p_[slot][0].elapsed_time(p_[slot][1]) for p_ in per_pass)
        f = fam.setdefault(name, [0.0, 0.0, 0, bound, ns])
        f[0] += ms
        f[1] += units
        f[2] += 1
        shp = shape_of.get(marks[1] + slot)
        if shp is not None:           # per problem shape of a GEMM family: the rows a kernel trace's per-shape summary holds
            s_ = shapes.setdefault(name, {}).setdefault(shp, [0.0, 0])
            s_[0] += ms
            s_[1] += 1

    def load_kernels(stem):
        path = latest_profile(stem)
        if not path:
            return {}, {}, None
        with open(path) as fh:
            doc = json.load(fh)
        return doc.get("kernels", {}), doc.get("by_shape", {}), "profiles/" + os.path.basename(path)

    traffic, traffic_shapes, tsrc = load_kernels("pmc_traffic.json")
    busy, busy_shapes, bsrc = load_kernels("pmc_mfma.json")

    def shape_lookup(table, name, wgs, field):
        """profiles' per-shape tables are keyed "<kernel> @ <workgroups> workgroups": the launch's tile count tells the conv
        layers and the M = 4000 / 8000 projections of one kernel apart."""
        base = name.split(" ")[0].split("<")[0]
        f16 = "[f16x2]" in name
        for k, v in table.items():
            if not k.startswith(base) or not k.endswith(f"@ {wgs} workgroups"):
                continue
            if "NT=3" in name and ", 3," not in k or "NT=4" in name and ", 4," not in k:
                continue
            if base.startswith("gemm_pp") and (("true" in k.split("@")[0]) != f16):
                continue
            return v.get(field)
        return None

    def workgroups(name, m_, n_, b_):
        bm, bn = (256, 256) if "(256x256)" in name else (128, 192) if "(128x192)" in name else (128, 256) if "(128x256)" in name else (0, 0)
        return -(-m_ // bm) * -(-n_ // bn) * b_ if bm else None

    def pmc_lookup(table, name, field):
        """profiles' tables are keyed by the demangled kernel name: match on the kernel's base name and, for the
        templated GEMMs, on the tile / precision arguments bench.py knows the launch by."""
        base = name.split(" ")[0].split("<")[0]
        want = []
        if "NT=3" in name:
            want.append(", 3,")
        if "NT=4" in name:
            want.append(", 4,")
        if name.startswith("gemm_kernel<"):
            want.append(name.split(" ")[0].split("<")[1].rstrip(">"))
        f16 = "[f16x2]" in name
        for k, v in table.items():
            if not k.startswith(base):
                continue
            if any(w_ not in k + "," for w_ in want):
                continue
            if base.startswith("gemm_pp") and (("true" in k) != f16):
                continue
            return v.get(field)
        return None

    sampler_cus = prior.cus_held(voxel.shape[0])

    def describe(name):
        ms, units, n, bound, ns = fam[name]
        e = {"bound": bound, "kernel": name, "launches_per_step": n, "avg_launch_us": round(ms * 1e3 / n, 2),
             "ms_per_step": round(ms, 3), "frac_of_step": round(ms / step_ms, 3),
             "traffic": pmc_lookup(traffic, name, "hbm_bytes_per_launch")}
        if bound == "mfma":
            ach = units / (ms * 1e-3) / 1e12
            e.update({"kernel": f"{name} (MFMA 16x16x32, {ns} MFMA per product)", "achieved": round(ach, 1),
                      "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                      "mfma_per_product": ns, "mfma_issued_frac": round(ach * ns / PEAK_BF16_TFLOPS, 4),
                      # inside the pass the sampler's workgroups hold `sampler_cus` of the 256 CUs: the same rate against
                      # the matrix-core peak of the CUs the audio branch can actually run on
                      "mfma_issued_frac_of_free_cus": round(ach * ns / (PEAK_BF16_TFLOPS * (256 - sampler_cus) / 256.0), 4),
                      "algorithmic_gflop_per_step": round(units / 1e9, 1),
                      # matrix-pipe busy cycles / (kernel cycles x 1024 SIMDs), rocprofv3 PMC pass (kernels serialised by
                      # the profiler, i.e. the kernel alone on the chip)
                      "mfma_busy_pmc": pmc_lookup(busy, name, "mfma_busy_frac"),
                      "by_shape": [{"M": m_, "N": n_, "K": k_, "batch": b_, "launches_per_step": c_, "avg_launch_us": round(t_ / c_ * 1e3, 1),
                                    "achieved_tflops": round(2.0 * m_ * n_ * k_ * b_ * c_ / t_ / 1e9, 1),
                                    "frac": round(2.0 * m_ * n_ * k_ * b_ * c_ / t_ / 1e9 / PEAK_BF16_TFLOPS, 4),
                                    "workgroups": workgroups(name, m_, n_, b_),
                                    "traffic": shape_lookup(traffic_shapes, name, workgroups(name, m_, n_, b_), "hbm_bytes_per_launch"),
                                    "mfma_busy_pmc": shape_lookup(busy_shapes, name, workgroups(name, m_, n_, b_), "mfma_busy_frac")}
                                   for (m_, n_, k_, b_), (t_, c_) in sorted(shapes.get(name, {}).items(), key=lambda kv: -kv[1][0])]})
        else:
            gbps = units / (ms * 1e-3) / 1e9
            e.update({"achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
                      "algorithmic_bytes_per_launch": int(units / n)})
            if name.startswith("layernorm"):
                e["note"] = "operands were just written by the producing GEMM: served by L2 / Infinity Cache"
        return e

    entries = [describe(k) for k in fam]
    audio_ms = round(sum(e["ms_per_step"] for e in entries), 3)
    if srec:
        B = voxel.shape[0]
        sms = min(a.elapsed_time(b) for a, b in srec[1:])
        spg = max(1, min(prior.samples_per_group, B))
        paired = prior.uses_pairs(B)
        groups = B if paired else -(-B // spg)
        cus = prior.cus_held(B)
        plane_bytes = sum(t.numel() * t.element_size() for t in prior.net._packs)
        T = prior.noise_scheduler.num_timesteps
        # paired (csrc/prior_pair.hip): q heads and feed-forward halves split over the two CUs of a sample, each CU streams
        # 524 KB of the 983 KB a layer's planes hold (k / v and the norms on both)
        nbytes = int(plane_bytes * (524.0 / 983.0 if paired else 1.0)) * T * cus
        ams = min(a.elapsed_time(b) for a, b in arec[1:]) if len(arec) > 1 else 0.0
        entries.append({
            "bound": "hbm", "bound_detail": "weights re-streamed L2 -> CU every DDPM step (per-CU ingest, served by L2 / "
                                            "Infinity Cache: neither the HBM nor the MFMA roof); latency-bound chain of "
                                            f"{T} dependent steps",
            "kernel": f"prior sampler ({T}-step DDPM in one launch, {B} samples on {cus} CUs"
                      f"{', a PAIR of samples on two CUs, each streaming half of every matrix' if paired else ''}, side stream)",
            "achieved": round(nbytes / sms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(nbytes / sms / 1e6 / 8000.0, 4), "per_cu_gbps": round(nbytes / cus / sms / 1e6, 1),
            # what actually bounds it: a CU takes in 66-73 GB/s from its XCD's L2 (MI355X_MICROARCH.md, gather rates)
            "per_cu_ingest_ceiling_gbps": 70.0, "frac_of_ingest_ceiling": round(nbytes / cus / sms / 1e6 / 70.0, 3),
            "aligner_ms_in_front_of_it": round(ams, 3), "branch_ms": round(sms + ams, 3),
            "launches_per_step": 1, "avg_launch_us": round(sms * 1e3, 1), "ms_per_step": round(sms, 3),
            "frac_of_step": round(sms / step_ms, 3), "algorithmic_bytes_per_launch": nbytes,
            "algorithmic_gflop_per_step": round(B * T * 12.4e-3, 1),
            "traffic": pmc_lookup(traffic, "prior_sample", "hbm_bytes_per_launch"),
            "mfma_busy_pmc": pmc_lookup(busy, "prior_sample", "mfma_busy_frac")})
    # Order.  `roofline` = the kernel family that does most of the pass's ARITHMETIC (largest algorithmic FLOPs per step: the
    # 256 x 256 GEMM of the conv layers, half of the path's FLOPs) - stable across devices of the pool and precision plans;
    # `others` by time per step.  The sampler runs BESIDE the audio branch on its own 32 CUs: with the conv layers on two MFMAs
    # per product the two branches are within 2-5 % of each other, so which one ends a pass last differs from device to
    # device; its entry says so (`role`) and leads `others` when its branch (aligner + launch) is >= 95 % of the step.
    smp = [e for e in entries if e["kernel"].startswith("prior sampler")]
    rest = [e for e in entries if e not in smp]
    lead = max((e for e in rest if e["bound"] == "mfma"), key=lambda e: e["algorithmic_gflop_per_step"])
    entries = [lead] + sorted((e for e in rest if e is not lead), key=lambda e: -e["ms_per_step"])
    if smp:
        bounds = smp[0]["branch_ms"] >= 0.95 * step_ms
        smp[0]["role"] = (f"its branch (aligner + sampler, {smp[0]['branch_ms']:.2f} ms) ends the pass together with the audio "
                          f"branch ({audio_ms:.2f} ms accounted in the eager instrumented pass): the step is bound by both" if bounds
                          else f"beside the audio branch with {step_ms - smp[0]['branch_ms']:.1f} ms of slack (eager, instrumented pass)")
        entries.insert(1 if bounds else 2, smp[0])
    entries[0]["chosen_by"] = "largest algorithmic FLOPs per step among the matrix-core families"
    out = entries[0]
    out["others"] = entries[1:]
    out["audio_branch_ms_accounted"] = audio_ms
    if am.split_streams > 1:
        out["audio_branch_ms_accounted_note"] = (
            f"sum of launches timed ONE AT A TIME beside the sampler; in the pass the 12 encoder layers run as "
            f"{am.split_streams} chains of clips side by side (their launches - 128-row GEMMs, attention, LayerNorm - overlap), "
            f"so the branch is shorter than this sum")
    out["pmc_sources"] = [x for x in (tsrc, bsrc) if x]
    if pipelined:
        # the arrangement that was TIMED: start / end of every branch graph inside back-to-back replayed passes
        out["branches"] = pipe.pass_timeline()
    return out


def measure_cpu_baseline(wa, wh, wp, joint_norm=False, reps=5, threads=16, keep=None):
    """the CPU oracle (fp32 torch restatement of the reference - a port; the reference import
    cannot travel to this box) on a BOUNDED sample, units B1-B5 timed separately, 1 warm-up + `reps` timed repetitions,
    median.  `value` = frames/s of the same path as the headline (B1 audio encoder + B3 prior + B2 head on one
    10 s clip).  Threads: the GPU box gives one GPU's job a 16-core share (torch would otherwise start one thread per
    visible core and oversubscribe it)."""
    import statistics
    from avi_talking_amd import weights as W
    from oracle import emote as OE, faceformer as OF, prior as OP, wav2vec2 as OW
    prev = torch.get_num_threads()
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    try:
        pcm = synth_audio(1, N_SAMPLES, 99)
        voxel = torch.randn(1, 768, generator=torch.Generator().manual_seed(98))
        noise = torch.randn(101, 1, 1, 128, generator=torch.Generator().manual_seed(97))
        st = {}

        def med(fn, n=reps):
            with torch.no_grad():
                fn()
                ts = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    fn()
                    ts.append(time.perf_counter() - t0)
            return statistics.median(ts)

        def b1():
            st["feat"] = OW.forward(wa, OW.normalize_audio(pcm, joint=joint_norm), frame_num=T_FRAMES)

        def b3():
            te, _ = OP.brain_network(wp, voxel)
            st["style"] = OP.p_sample_loop(wp, te.view(1, 1, 128), noise)

        t1, t3 = med(b1), med(b3)
        t2 = med(lambda: OE.forward(wh, st["feat"], st["style"]))
        if keep is not None:      # the clip and the oracle's results: measure_parity runs the GPU pipeline on the same inputs
            with torch.no_grad():
                keep.update(pcm=pcm, voxel=voxel, noise=noise, style=st["style"], ref=OE.forward(wh, st["feat"], st["style"]))
        units = {"B1_wav2vec2_1x10s_s": round(t1, 4), "B2_emote_flint_head_1x250_s": round(t2, 4),
                 "B3_aligner_plus_100step_ddpm_1_sample_s": round(t3, 4)}
        # B4: FaceFormer AR loop, hidden (1,T,D), as written (whole prefix re-decoded every frame) and KV-cached
        for D, T in ((64, 100), (1024, 100)):
            wf = W.make_faceformer_weights(2, feature_dim=D)
            hs = torch.randn(1, T, D, generator=torch.Generator().manual_seed(7))
            units[f"B4_faceformer_D{D}_T{T}_as_written_frames_per_s"] = round(
                T / med(lambda: OF.predict_as_written(wf, hs, 30), 3), 1)
            units[f"B4_faceformer_D{D}_T{T}_cached_frames_per_s"] = round(T / med(lambda: OF.predict_cached(wf, hs, 30), 3), 1)
        # B5: one training step (fwd + autograd bwd + AdamW over every parameter), B = 64
        g = torch.Generator().manual_seed(5)
        Bt = 64
        params = {k: v.clone().requires_grad_(True) for k, v in wp.items()
                  if v.is_floating_point() and not k.startswith("noise_scheduler")}
        mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
        vx, tg = torch.randn(Bt, 768, generator=g), torch.randn(Bt, 1, 128, generator=g) * 0.3
        times, nz = torch.randint(0, 100, (Bt,), generator=g), torch.randn(Bt, 1, 128, generator=g)

        def b5():
            with torch.enable_grad():
                loss = OP.train_loss(params, vx, tg, times, nz, 0.005)[0]
                grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
            for (k, p_), g_ in zip(params.items(), grads):
                if g_ is None:
                    continue
                OP.adamw_step(p_.data, g_, mom[k][0], mom[k][1], 1, 1e-4, weight_decay=0.0 if OP.no_decay(k) else 1e-2)

        t5 = med(b5, 3)
        units["B5_train_step_B64_s"] = round(t5, 4)
        units["B5_train_samples_per_s"] = round(Bt / t5, 1)
    finally:
        torch.set_num_threads(prev)
    total = t1 + t2 + t3
    return {"value": round(T_FRAMES / total, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 clip x 10 s through the same path (B1 + B3 + B2 = {total:.2f} s), fp32 torch oracle, 1 warm-up + "
                      f"{reps} timed repetitions each, median; audio normalised "
                      + ("jointly" if joint_norm else "per clip"),
            "units": units}


def measure_precision_modes(wa, wh, wp, dev, pcm, voxel, noise, ref_out, ref_name, args):
    """Secondary: the same config[1] pass under the other precision plans (ops.PrecPlan), each with its own parity line:
    max-abs difference of its coefficients from the bf16x3 (3-term) pass on the same inputs (that mode is 2e-5 from the
    oracle; tests/test_gpu_mixed_prec.py and tests/test_gpu_emote.py pin every plan against the oracle itself).  Timed IN
    THIS PROCESS with the headline's own timed region (pipelined two-graph replay, `--steps` passes after `--warmup`):
    pipelines of one device share their streams (host/pipeline.device_streams), so a second pipeline object replays as
    fast as the first (tests/test_gpu_fullsize.py::test_second_pipeline_replays_as_fast)."""
    from avi_talking_amd.host.pipeline import SamplingPipeline
    out = {"reference_mode_for_diff": "bf16x3", "modes": []}
    base = ref_out if ref_name == "bf16x3" else None
    order = [m for m in ("bf16x3", "mixed", "mixed_ffn", "f16x2") if m != ref_name]
    for mode in order:
        pipe = SamplingPipeline(wa, wh, wp, device=dev, prec=mode, joint_norm=args.joint_norm, rng_seed=777)
        o = pipe.run(pcm, voxel, noise)                      # recorded noise for the diff, in-pass draws for the timing
        torch.cuda.synchronize(dev)
        cur = {k: o[k].clone() for k in ("predicted_exp", "predicted_jaw")}
        if mode == "bf16x3":
            base = cur
        pipe.capture_pipelined(pcm, voxel, None)
        for _ in range(args.warmup):
            pipe.replay_pipelined()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.replay_pipelined()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        rec = {"dtype": mode, "plan": repr(pipe.plan), "ms_per_step": round(dt / args.steps * 1e3, 3),
               "frames_per_s": round(B_CLIPS * T_FRAMES * args.steps / dt, 1), "steps": args.steps,
               "_out": cur}
        out["modes"].append(rec)
        del pipe, o
        torch.cuda.empty_cache()
    if base is None:
        base = ref_out
    for rec in out["modes"]:
        cur = rec.pop("_out")
        rec["max_abs_coeff_diff_vs_bf16x3"] = float("%.3e" % max((cur[k] - base[k]).abs().max().item() for k in cur))
    if ref_name != "bf16x3":
        out["headline_max_abs_coeff_diff_vs_bf16x3"] = float(
            "%.3e" % max((ref_out[k] - base[k]).abs().max().item() for k in ref_out))
    out["gates"] = {"north_star": 1e-3, "mixed (tests/test_gpu_mixed_prec.py, vs oracle)": 3e-4}
    return out


def measure_longform(wa, wh, wp, dev, prec, reps=5):
    """BASELINE.json configs[4]: long-form utterances, 60 s = 1500 frames at 25 fps, fp16 coefficients.
      * the sampling path (audio -> wav2vec2 -> EMOTE/FLINT, style from the 100-step prior) on B x 60 s clips through
        SamplingPipeline as ONE hipGraph, coefficients stored as IEEE half by the head's last kernel, DDPM noise drawn in
        the pass; B = 8 and B = 32 (what stays resident in HBM: torch.cuda.max_memory_allocated);
      * the FaceFormer decoder, chunked-causal window of 600 frames, D = 64 and 1024, 8 utterances, half output.
    The reference has NO behaviour at this length: its FaceFormer mask / PPE tables stop at 600 frames
    (models/faceformer.py:88,147) and EMOTE's mask at 1200 (FaceFormerDecoder.py:1010); the chunked-causal window is defined
    by this build (include/avi_talking.h avi_faceformer_decode_chunked, oracle/faceformer.py predict_cached(chunk=)), the
    EMOTE/FLINT path has no mask-length limit here (its ALiBi bias is evaluated analytically)."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.faceformer import Faceformer
    from avi_talking_amd.host.pipeline import SamplingPipeline
    T, N = 1500, 1500 * 640
    out = {"workload": "configs[4]: 60 s utterances (1500 frames), fp16 coefficients",
           "window_semantics": "builder-defined: the reference cannot decode beyond 600 (FaceFormer, models/faceformer.py:88,147) "
                               "/ 1200 (EMOTE mask) frames; FaceFormer here = chunked-causal, chunk 600 (avi_talking.h)",
           "sampling": [], "faceformer": []}

    def best_ms(fn):
        fn()
        torch.cuda.synchronize(dev)
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize(dev)
        return min(a.elapsed_time(b) for a, b in evs)

    def synth_audio_big(B):
        """synth_audio in chunks of 32 clips (its FFT of 256 x 960 000 samples at once would take 4 GB of host memory)."""
        return torch.cat([synth_audio(min(32, B - b0), N, 4321 + b0) for b0 in range(0, B, 32)], 0)

    # B = 256 x 60 s is the "288 GB HBM residency" point of configs[4]: conv layer 0's output alone is 25 G elements (100 GB
    # of fp16 hi/lo planes), beside conv layer 1's 50 GB; AVI_BENCH_LONGFORM_MAX_CLIPS trims it (0 = skip the big point)
    big = int(os.environ.get("AVI_BENCH_LONGFORM_MAX_CLIPS", "256"))
    for B in (8, 32) + ((big,) if big > 32 else ()):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(dev)
        pipe = SamplingPipeline(wa, wh, wp, device=dev, prec=prec, rng_seed=99, out_dtype=torch.float16)
        pcm = synth_audio_big(B).to(dev)
        voxel = torch.randn(B, 768, generator=torch.Generator().manual_seed(4322)).to(dev)
        pipe.capture(pcm, voxel, None, warmup=1)
        ms = best_ms(pipe.replay)
        o = pipe.replay()
        pipe.synchronize()                # + no device-side failure report (fp16 plane range, sampler timeout)
        assert o["predicted_exp"].dtype == torch.float16 and o["predicted_exp"].shape == (B, T, 50)
        assert all(bool(torch.isfinite(o["predicted_exp"][b0:b0 + 32].float()).all()) for b0 in range(0, B, 32))
        out["sampling"].append({"clips": B, "frames_per_clip": T, "ms_per_pass": round(ms, 3),
                                "frames_per_s": round(B * T / ms * 1e3, 1), "dtype": repr(pipe.plan), "coeff_dtype": "fp16",
                                "max_memory_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
                                "largest_activation_elements": B * ((N - 10) // 5 + 1) * 512})
        del pipe, o, pcm
    torch.cuda.empty_cache()
    for D in (64, 1024):
        m = Faceformer(W.make_faceformer_weights(2, feature_dim=D), period=30, device=dev)
        hs = torch.randn(8, T, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        ms = best_ms(lambda: m.decode(hs, chunk=600, out_dtype=torch.float16))
        out["faceformer"].append({"D": D, "utterances": 8, "frames": T, "chunk": 600, "coeff_dtype": "fp16",
                                  "ms": round(ms, 3), "frames_per_s": round(8 * T / ms * 1e3, 1),
                                  "us_per_frame_step": round(ms * 1e3 / T, 2)})
        del m
    return out


def measure_faceformer(dev, reps=5):
    """North_star's second decoder (row E, models/faceformer.py:710-729): the autoregressive FaceFormer decode, all T
    frames in one launch, on hidden states resident in HBM.  T = 250 (10 s), D = 64 and 1024 (config/vocaset/demo.yaml),
    one utterance and a batch of 32; best of `reps` event-timed calls."""
    from avi_talking_amd import weights as W
    from avi_talking_amd.host.faceformer import Faceformer
    out = {"workload": "FaceFormer AR decode, T = 250 frames per utterance, hidden states in HBM, period 30",
           "cases": []}
    for D in (64, 1024):
        m = Faceformer(W.make_faceformer_weights(2, feature_dim=D), period=30, device=dev)
        for B in (1, 32):
            hs = torch.randn(B, T_FRAMES, D, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            m.decode(hs)
            torch.cuda.synchronize(dev)
            evs = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                m.decode(hs)
                e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize(dev)
            ms = min(a.elapsed_time(b) for a, b in evs)
            wbytes = (8 * D * D + 2 * 53 * D) * 4                        # fp32 weights of one frame step (qkv, out, ff1, ff2, maps)
            path = {"single": "one workgroup per utterance", "steps": "launch chain", "persist": "persistent launch"}[m._last_path]
            out["cases"].append({"D": D, "utterances": B, "path": path, "ms": round(ms, 3),
                                 "frames_per_s": round(B * T_FRAMES / ms * 1e3, 1),
                                 "us_per_frame_step": round(ms * 1e3 / T_FRAMES, 2),
                                 "weight_bytes_per_step": wbytes,
                                 "weight_stream_gbps_per_utterance": round(wbytes * T_FRAMES / ms / 1e6, 1)})
    from avi_talking_amd.host import status
    status.raise_if_set()       # a persistent decode that timed out on an exchange would have returned NaN: not a measurement
    return out


if __name__ == "__main__":
    main()
