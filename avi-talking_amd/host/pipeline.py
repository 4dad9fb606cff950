"""Sampling pipeline of the reference entry point (``train_diffusion_prior.py --is_test 1``, the per-utterance
loop at :689-771) as one batched, graph-capturable pass:

    int16 PCM (B, T*640) --normalise--> wav2vec2 (25 Hz) --------------------.
    CLIP text feature (B,768) --BrainNetwork--> text_embed --100-step DDPM--> style (B,1,128)
                                                                              |
    audio features + style --EMOTE head + FLINT decoder--> expression (B,T,50) | jaw (B,T,3)

Differences from the reference loop that do not change results: utterances are batched (the
reference runs batch 1; audio statistics are therefore taken PER CLIP, which is what the HF processor inside
AudioEncoders.py:170-178 computes when it only ever sees one utterance - ``joint_norm=True`` selects the processor's
joint-over-the-batch quirk instead, under which a clip's coefficients depend on its batch mates and on how the batch
is sharded over ranks; tests/test_gpu_fullsize.py pins batched == utterance by utterance), wav2vec2 runs once per utterance (the reference runs it twice,
train_diffusion_prior.py:696 vs :764), no host syncs (NaN sweeps, .item()) and no per-step Python in
the DDPM loop; the prior is sampled on a second HIP stream concurrently with the audio encoder.
"""
import os

import torch

from .. import ops
from . import status
from .diffusion_prior import InstructDiffusionPrior
from .talking_head import TalkingHeadWrapper


# Streams are a per-DEVICE resource here, not a per-object one.  HIP multiplexes streams onto a few in-order hardware queues
# and which queue a stream gets depends on how many streams the process has created before it: a second pipeline object
# that made its own side stream and its own pool of candidate streams got a different (worse) queue assignment than the
# first and replayed 10-30 % slower in the same process (round 2: 13.0-14.9 vs 11.0 ms for the same graphs in a fresh
# process).  Every pipeline of a device therefore shares ONE high-priority side stream, ONE pool of replay streams and the
# (body, head) pair chosen from it by the first capture_pipelined() on that device.
_DEVICE_STREAMS = {}


def device_streams(device):
    """{'side': high-priority stream of the sampler's branch, 'pool': candidate replay streams (pool[0] replays the body's
    audio branch), 'picks': {encoder chains: {'head': index of the head's stream in the pool, 'chains': indices of the
    further chains' streams, 'timings_ms': ...}} (filled by the first capture_pipelined on the device that needs it)}."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _DEVICE_STREAMS.get(key)
    if st is None:
        st = _DEVICE_STREAMS[key] = {
            # high priority: the sampler's 32 workgroups must get their CUs at once, not behind a round of GEMM tiles
            "side": torch.cuda.Stream(device=device, priority=-1),
            "pool": [torch.cuda.Stream(device=device) for _ in range(8)], "picks": {}}
    return st


class SamplingPipeline:
    def __init__(self, audio_sd, head_sd, prior_sd, device="cuda", prec=None, joint_norm=False, side_stream=None,
                 rng_seed=None, out_dtype=torch.float32):
        """``prec``: None = ``ops.DEFAULT_PREC`` ("mixed": the plan bench.py's headline is measured on), a plan name
        ("bf16x3", "mixed", "mixed_ffn", "f16x2"), an AVI_PREC_* value or an ops.PrecPlan.
        ``rng_seed``: draw the DDPM noise INSIDE the pass from the library's device-resident Philox stream (host/rng.py)
        whenever a call passes ``noise=None`` - the reference draws it inside its loop (models/diffusion_prior.py:337,
        349-351); a captured pass then draws fresh noise at every replay.  None = the caller supplies the noise tensor."""
        self.device = torch.device(device)
        self.out_dtype = out_dtype         # float16: the head's last kernel stores the coefficients as IEEE half (configs[4])
        from .rng import DeviceRng
        self.rng = None if rng_seed is None else DeviceRng(rng_seed, self.device)
        self._noise_bufs = {}          # per batch size, never freed: captured graphs keep writing into theirs
        self.plan = plan = ops.prec_plan(prec)
        self._sds, self._joint_norm = (audio_sd, head_sd, prior_sd), joint_norm      # run_checked's fallback pipeline
        self._fallback_pipe = None
        self.last_fallback = None      # what run_checked had to do for its most recent batch (None: nothing)
        status.words()                 # the device-side failure reports this object polls (host/status.py)
        self.talking_head = TalkingHeadWrapper(audio_sd, head_sd, device=device, prec=plan, joint_norm=joint_norm)
        # the uniform fp16 mode also stores the sampler's attention matrices as one fp16 plane (the mixed plans leave the
        # sampler as it is in the default mode)
        self.prior = InstructDiffusionPrior.from_state_dict(
            prior_sd, device=device, prec=plan.small, attn_fp16=True if plan.sampler_all_fp16 else None)
        self.side = side_stream if side_stream is not None else device_streams(self.device)["side"]
        self.prior.time_table()          # built once, outside any capture
        self._graph = None
        self._static = None

    def voxel2style_emb(self, voxel, noise):
        """train_diffusion_prior.py:783-853: voxel (B,768) -> sampled style embedding (B,1,128)."""
        clip_voxels, _ = self.prior.voxel2clip(voxel, need_projection=False)
        B = voxel.shape[0]
        return self.prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": clip_voxels.view(B, 1, 128)},
                                        cond_scale=1.0, timesteps=self.prior.noise_scheduler.num_timesteps,
                                        noise=noise)

    def _body(self, pcm, voxel, noise, clip_voxels=None, aligner_on_side=None):
        """Everything up to the join of the two branches: -> (audio features (B,T,768), sampled style (B,1,128)).
        ``clip_voxels``: the aligner's output when it was computed ahead (pipelined replay); None = compute it here."""
        # pcm (B, N): 640 samples per frame (process_audio's framing); pcm (B, T, S): the caller's own frames, e.g. the
        # (T + 1, 641) rows of the reference's create_base_sample (host/sample.py) - T frames come out either way
        if pcm.dim() == 3:
            B, T, S_ = pcm.shape
            raw = pcm
        else:
            B, N = pcm.shape
            T = N // 640
            raw = pcm.view(B, T, 640)
        cur = torch.cuda.current_stream(self.device)
        # 1. Aligner MLP on the launch stream, before anything else (0.17 ms: split-K launches that stream the 300 MB
        #    of weights at HBM speed).  On the second stream its ten launches each queue behind resident GEMM
        #    workgroups of the audio branch and the sampler starts late (+0.4 ms per step, measured);
        #    AVI_ALIGNER_SIDE=1 selects that arrangement for tests/test_gpu_fullsize.py, which pins that overlapping
        #    short matrix-core launches with conv layer 0 no longer corrupts it (build.py: no packed-FP32 instructions).
        if aligner_on_side is None:
            aligner_on_side = os.environ.get("AVI_ALIGNER_SIDE", "0") == "1"
        aligner_on_side = aligner_on_side and clip_voxels is None
        if not aligner_on_side and clip_voxels is None:
            clip_voxels, _ = self.prior.voxel2clip(voxel, need_projection=False)
        # 2. fork: the sampler (32 workgroups for ~12 ms) on the side stream, the audio encoder on this one
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            if noise is None:            # the draws of the reference's loop (:337,349-351), inside the pass
                noise = self._draw_noise(B)
            if aligner_on_side:
                clip_voxels, _ = self.prior.voxel2clip(voxel, need_projection=False)
            style = self.prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": clip_voxels.view(B, 1, 128)},
                                             cond_scale=1.0, timesteps=self.prior.noise_scheduler.num_timesteps,
                                             noise=noise)
        # the sampler's workgroups (one per samples_per_group samples) hold a CU each until the join: GEMM tile shapes of the
        # audio branch are chosen for the CUs that remain (avi_talking.h AviGemm.cus)
        free_cus = max(32, 256 - self.prior.cus_held(B))
        sample = self.talking_head.forward_audio({"raw_audio": raw, "samplerate": [16000] * B}, cus=free_cus)
        # 3. join
        cur.wait_stream(self.side)
        return sample["audio_feature"], style

    def _draw_noise(self, B):
        """(T_d+1, B, 1, 128) standard normals from the device stream into a buffer this object keeps, then the offset moves
        on (both launches on the current stream: inside a capture they are nodes of the graph)."""
        from . import rng as R
        if self.rng is None:
            raise ValueError("noise=None needs a pipeline built with rng_seed=... (in-pass draws)")
        T = self.prior.noise_scheduler.num_timesteps
        buf = self._noise_bufs.get(B)
        if buf is None:
            buf = self._noise_bufs[B] = torch.empty((T + 1, B, 1, 128), dtype=torch.float32, device=self.device)
        self._noise_buf = buf          # the noise of the most recent pass (tests read it back)
        self.rng.fill(buf, R.NORMAL, subsequence=0)
        self.rng.advance(1)
        return buf

    def run(self, pcm, voxel, noise=None):
        """pcm int16/fp32 (B, T*640) - or (B, T, samples per frame) - resident on the device, voxel (B,768), noise (T_d+1,B,1,128) or None (drawn inside
        the pass, ``rng_seed``) -> dict(predicted_exp (B,T,50), predicted_jaw (B,T,3), style_emb (B,1,128)).
        Asynchronous.  A device-side failure of an EARLIER pass (fp16 plane range, paired-sampler timeout: host/status.py)
        raises here, at the start of the next one, without any synchronisation; ``run_checked`` is the synchronous form
        that detects and repairs a failure of its own pass."""
        self._poll()
        feat, style = self._body(pcm, voxel, noise)
        out = self.talking_head.head(feat, style, out_dtype=self.out_dtype)
        out["style_emb"] = style
        return out

    def _poll(self):
        """Raise for whatever the device has reported so far (two reads of pinned host memory, no synchronisation)."""
        self.prior.pair_status()
        status.raise_if_set()

    def check(self):
        """After the caller has synchronised with the pass(es): raise ``status.PairTimeout`` if a paired-sampler launch gave
        up on its partner (that pass's style and coefficients are NaN) or ``status.RangeError`` if an fp16 activation plane
        left its range.  No device read: the kernels write into pinned host memory.  Every entry point of this object polls
        the same words on its way in, so a caller that never calls check() still gets the exception at its next call."""
        self._poll()

    def synchronize(self):
        """Wait for every pass enqueued so far, then check()."""
        torch.cuda.synchronize(self.device)
        self.check()

    def run_checked(self, pcm, voxel, noise=None):
        """``run`` + a synchronisation + the safety net: the batch is RE-RUN when the device reported that its pass cannot be
        trusted - on the unpaired sampler kernel after a paired-sampler timeout, on the ``ops.FALLBACK_PREC`` plan (bf16
        planes: fp32's range) after an fp16-plane range report - with the same DDPM noise as the failed pass.  Returns valid
        results or raises; ``self.last_fallback`` says what was done.  This is what host/cli.py calls per utterance."""
        self.last_fallback = None
        self._poll()
        out = self.run(pcm, voxel, noise)
        torch.cuda.synchronize(self.device)
        ovf, tiny, pair = status.read()
        if not (ovf or tiny or pair):
            return out
        if noise is None:                      # the failed pass drew its noise inside: the re-run takes the same numbers
            noise = self._noise_buf.clone()
        status.clear()
        for ws in self.prior._pair_ws.values():
            ws[1].zero_()
        if ovf or tiny:
            if self.plan.name == ops.FALLBACK_PREC:
                raise status.RangeError("fp16-plane range report under the fallback plan itself (no fp16 planes there): "
                                        "another object of this process is running a 2-term fp16 plan")
            if self._fallback_pipe is None:
                self._fallback_pipe = SamplingPipeline(*self._sds, device=self.device, prec=ops.FALLBACK_PREC,
                                                       joint_norm=self._joint_norm, out_dtype=self.out_dtype)
            self.last_fallback = (f"precision plan {ops.FALLBACK_PREC} (fp16 planes "
                                  + ("overflowed" if ovf else "underflowed") + f" under {self.plan.name})")
            out = self._fallback_pipe.run_checked(pcm, voxel, noise)
            if self._fallback_pipe.last_fallback:
                self.last_fallback += " + " + self._fallback_pipe.last_fallback
            return out
        was, self.prior.paired = self.prior.paired, False
        try:
            self.last_fallback = "unpaired sampler kernel (paired-sampler timeout)"
            out = self.run(pcm, voxel, noise)
            torch.cuda.synchronize(self.device)
        finally:
            self.prior.paired = was
        self._poll()                           # a failure of the re-run itself raises
        return out

    # ---- hipGraph capture of the whole pass (static input buffers, replayed per batch)
    def capture(self, pcm, voxel, noise=None, warmup=2):
        self._static = (pcm.clone(), voxel.clone(), None if noise is None else noise.clone())
        for _ in range(warmup):
            self.run(*self._static)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self.run(*self._static)
        return self

    def replay(self, pcm=None, voxel=None, noise=None):
        if self._graph is None:
            raise RuntimeError("capture() first")
        self._poll()
        for dst, src in zip(self._static, (pcm, voxel, noise)):
            if src is not None:
                if dst is None:
                    raise ValueError("the pass was captured with in-pass noise draws: it takes no noise tensor")
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._out

    # ---- software-pipelined replay: the serial end of a pass runs beside its successor
    def capture_pipelined(self, pcm, voxel, noise=None, warmup=2, arrangements=None):
        """Body and head instead of one graph.  A pass ends with the EMOTE/FLINT head (~30 small dependent launches, 0.37 ms,
        the chip almost idle); captured separately and replayed on a stream of its own it runs beside the start of pass
        k+1.  The BODY is itself one graph per branch (`_capture_body`): the sampler's branch (noise draw, aligner, DDPM
        loop) on the high-priority side stream, the audio front on the body's stream, the 12 encoder layers as one graph
        per chain of clips on streams of the device's pool, joined by events (`_enqueue_body`).  Back-to-back passes cost
        max(aligner + sampler, audio branch) each; the hand-off goes through private copies of the body's two results,
        guarded by events; every pass still does all of its work and the results are bit-identical to `replay()`
        (tests/test_gpu_fullsize.py).
        ``arrangements``: [(encoder chains, paired sampler)] candidates; None = `_arrangements` (the chain count 2 vs 1 is
        timed here, on this device, and the faster kept: `self.arrangement`).  Stream roles are timed once per device
        (`_pick_streams`, `self.stream_choice`).
        History: a separate aligner graph one pass ahead on a stream of its own was the first design - with the runtime's
        default four hardware queues its stream shared the body's queue and its launches ran BETWEEN two bodies (0.19 ms
        per pass on the critical path, scripts/pass_timeline.py --gap); one multi-branch body graph was the second (the
        runtime chooses the branches' queues: see `_capture_body`)."""
        from .. import lib as L
        self._static = (pcm.clone(), voxel.clone(), None if noise is None else noise.clone())
        dev = self.device
        self._e_body, self._e_taken = (torch.cuda.Event() for _ in range(2))

        def copy(src, dst):
            C_ = src.shape[-1]
            L.check(L.load().avi_copy_rows(src.data_ptr(), C_, None, dst.data_ptr(), C_, src.numel() // C_, C_,
                                           L.stream_ptr()), "avi_copy_rows")
        self._copy = copy
        cands = self._arrangements(pcm.shape[0], pcm.shape[1] // 640) if arrangements is None else list(arrangements)
        for _ in range(warmup):
            self.run(*self._static)
        torch.cuda.synchronize(dev)
        timed, best, self._g_head = {}, None, None
        for chains, paired in cands:
            body = self._capture_body(chains, paired)
            if self._g_head is None:   # the head reads private copies of the body's results: one capture serves every candidate
                self._h_feat, self._h_style = torch.empty_like(body.feat), torch.empty_like(body.style)
                self._g_head = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._g_head):
                    self._pout = self.talking_head.head(self._h_feat, self._h_style, out_dtype=self.out_dtype)
                    self._pout["style_emb"] = self._h_style
            self._pbody = body
            self._pick_streams()
            ms = self._time_replays() if len(cands) > 1 else 0.0
            timed[f"chains={chains},paired={int(paired)}"] = round(ms, 3)
            if best is None or ms < best[0]:
                best = (ms, body)
        self._pbody = body = best[1]
        self._pick_streams()              # the kept arrangement's streams (cached per chain count: nothing is timed again)
        self.talking_head.audio_model.split_streams, self.prior.paired = body.chains, body.paired
        self._feat, self._style = body.feat, body.style
        self.arrangement = {"encoder_chains": body.chains, "paired_sampler": bool(body.paired),
                            "candidate_ms_per_pass": timed}
        torch.cuda.synchronize(dev)
        return self

    def _capture_body(self, chains, paired):
        """The body of a pass as one graph PER BRANCH - the sampler's branch (noise draw, aligner, DDPM loop), the audio front
        (normalisation ... positional conv) and the 12 layers of each group of clips - replayed on a stream each and tied
        together with events (`_enqueue_body`).  One multi-branch graph was the first design and is what `capture()` still
        records: there the runtime maps the branches onto internal streams of its own, and with a third branch the audio
        branch of pass k+1 started 0.2-0.5 ms late, behind the aligner of the sampler's branch (kernel trace,
        scripts/pass_timeline.py) - which queue a branch lands on is not ours to choose inside a graph, on streams it is."""
        from types import SimpleNamespace
        pcm, voxel, noise = self._static
        B, T = pcm.shape[0], pcm.shape[1] // 640
        if B % chains:
            raise ValueError(f"{B} clips do not split into {chains} encoder chains")
        th, am = self.talking_head, self.talking_head.audio_model
        self.prior.paired = paired
        cus = max(32, 256 - self.prior.cus_held(B))
        per = B // chains

        def side():
            nz = self._draw_noise(B) if noise is None else noise
            clip_voxels, _ = self.prior.voxel2clip(voxel, need_projection=False)
            return self.prior.p_sample_loop((B, 1, 128), text_cond={"text_embed": clip_voxels.view(B, 1, 128)},
                                            cond_scale=1.0, timesteps=self.prior.noise_scheduler.num_timesteps, noise=nz)

        def front():      # common to all chains: the CNN as two chains of clips measured slower (9.2 vs 8.7 ms, audio alone)
            return th.forward_audio({"raw_audio": pcm.view(B, T, 640), "samplerate": [16000] * B}, cus=cus, front_only=True)

        # every launch shape once outside a capture (kernel attributes such as LDS grants are set on first use)
        side()
        h = front()
        for i in range(chains):
            am.encoder_layers(h[i * per:(i + 1) * per], cus)
        torch.cuda.synchronize(self.device)
        body = SimpleNamespace(chains=chains, paired=paired, g_side=torch.cuda.CUDAGraph(), g_front=torch.cuda.CUDAGraph(),
                               g_chain=[torch.cuda.CUDAGraph() for _ in range(chains)],
                               e_in=torch.cuda.Event(), e_side=torch.cuda.Event(), e_front=torch.cuda.Event(),
                               e_chain=[torch.cuda.Event() for _ in range(chains)], streams=None)    # streams: _pick_streams
        with torch.cuda.graph(body.g_side):
            body.style = side()
        with torch.cuda.graph(body.g_front):
            body.h = front()
        body.feat = torch.empty_like(body.h)
        for i, g in enumerate(body.g_chain):
            with torch.cuda.graph(g):
                body.feat[i * per:(i + 1) * per].copy_(am.encoder_layers(body.h[i * per:(i + 1) * per], cus))
        return body

    def _enqueue_body(self, new_inputs):
        """On the body's stream A (current).  What a pass must wait for, and no more:
          * the audio front reads the static PCM and writes only its own buffers: it follows the previous pass's chains on A
            with NO cross-stream dependency - the audio branch, which bounds the pass, runs back to back;
          * the chains' last launch overwrites `feat` and the sampler's last step `style`, which the previous head copies
            when it starts: the chains and the sampler's branch wait for `_e_taken` (long past when they get there);
          * the sampler's branch reads the static text feature / noise: it waits for this call's copies only if there
            were any.  It is launched first (its workgroups must get their CUs before a round of GEMM tiles does).
        Leaves `_e_body` recorded on A."""
        b, A, S = self._pbody, self._s_body, self.side
        mark = self._mark
        if new_inputs:
            b.e_in.record(A)
        with torch.cuda.stream(S):
            if new_inputs:
                S.wait_event(b.e_in)
            S.wait_event(self._e_taken)
            mark("sampler_branch", 0, S)
            b.g_side.replay()
            mark("sampler_branch", 1, S)
            b.e_side.record(S)
        mark("audio_front", 0, A)
        b.g_front.replay()
        mark("audio_front", 1, A)
        b.e_front.record(A)
        for i, C in enumerate(b.streams, 1):
            with torch.cuda.stream(C):
                C.wait_event(b.e_front)
                C.wait_event(self._e_taken)
                mark(f"encoder_chain_{i}", 0, C)
                b.g_chain[i].replay()
                mark(f"encoder_chain_{i}", 1, C)
                b.e_chain[i].record(C)
        A.wait_event(self._e_taken)
        mark("encoder_chain_0", 0, A)
        b.g_chain[0].replay()
        mark("encoder_chain_0", 1, A)
        for i in range(1, b.chains):
            A.wait_event(b.e_chain[i])
        A.wait_event(b.e_side)
        self._e_body.record(A)

    def _arrangements(self, B, T):
        """Candidate (encoder chains, paired sampler) arrangements of a pass, best guess first.  Only the CHAIN COUNT is left
        to timing - results are bit-identical across chain counts, so a choice made with a stopwatch cannot change what a
        pass returns - and it is timed because which count is fastest depends on how the runtime maps the pass's
        concurrent branches (sampler, encoder chains, previous head) onto its in-order hardware queues: two branches on one
        queue serialise.  The sampler kernel is NOT timed: paired and plain differ by 1-3e-6 in the style vector, so that
        choice follows a fixed rule (`prior.uses_pairs`: paired up to 32 samples).  AVI_W2V_SPLIT pins the chain count."""
        from .. import HW_QUEUES
        from .wav2vec import SPLIT_MIN_ROWS
        am = self.talking_head.audio_model
        fits = lambda n: n >= 1 and B % n == 0 and (n == 1 or B * T >= n * SPLIT_MIN_ROWS)
        if os.environ.get("AVI_W2V_SPLIT") is not None:
            chains = [am.split_streams if fits(am.split_streams) else 1]
        elif HW_QUEUES >= 8 and fits(2):
            chains = [2, 1]
        else:
            chains = [1]
        return [(c, self.prior.paired) for c in chains]

    def _time_replays(self, replays=6):
        import time
        for _ in range(2):
            self.replay_pipelined()
        torch.cuda.synchronize(self.device)
        t = time.perf_counter()
        for _ in range(replays):
            self.replay_pipelined()
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t) / replays * 1e3

    def _pick_streams(self, replays=4):
        """HIP multiplexes streams onto a few IN-ORDER hardware queues, and which queue a stream gets is not ours to choose:
        with the head's stream on the body's queue the head runs between two bodies instead of beside the next one
        (+0.5 ms per pass in the kernel trace), and an encoder chain on the sampler's queue waits for the whole DDPM loop
        (+3 ms).  The graphs do not care which stream replays them, so the roles - head, further encoder chains - are
        given streams of the device's pool by TIMING a few replays per candidate, one role at a time, ONCE PER DEVICE and
        chain count; the result is kept in ``device_streams(device)`` and every later capture on the device - this
        object's or another's - reuses it without measuring again: the choice is deterministic for the life of the
        process and visible as ``self.stream_choice`` (scripts/pass_timeline.py --gap shows the queue of every launch)."""
        import time
        dev = self.device
        st = device_streams(dev)
        pool, body = st["pool"], self._pbody
        n_extra = body.chains - 1
        if n_extra + 2 > len(pool):
            raise ValueError(f"{body.chains} encoder chains need {n_extra + 2} replay streams, the pool has {len(pool)}")
        pick = st["picks"].get(body.chains)
        if pick is None:
            roles = list(range(1, n_extra + 2))          # pool indices of [chain 1, ..., chain n, head]: first guess
            log = []

            def timed():
                self._s_body, self._s_head = pool[0], pool[roles[-1]]
                body.streams = [pool[i] for i in roles[:-1]]
                self.replay_pipelined()
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                for _ in range(replays):
                    self.replay_pipelined()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t) / replays * 1e3

            for r in range(len(roles)):                  # one role at a time, the others fixed
                best = None
                for k in range(1, len(pool)):
                    if k in roles[:r] + roles[r + 1:]:
                        continue
                    roles[r] = k
                    ms = timed()
                    log.append((list(roles), round(ms, 3)))
                    if best is None or ms < best[0]:
                        best = (ms, k)
                roles[r] = best[1]
            pick = st["picks"][body.chains] = {"head": roles[-1], "chains": roles[:-1], "timings_ms": log}
        self._s_body, self._s_head = pool[0], pool[pick["head"]]
        body.streams = [pool[i] for i in pick["chains"]]
        self.stream_choice = {"body": 0, "head": pick["head"], "chains": pick["chains"],
                              "candidate_ms_per_pass": pick["timings_ms"]}

    def replay_pipelined(self, pcm=None, voxel=None, noise=None):
        """Enqueue one pass; returns its output dict (valid once the device, or `self._s_head`, has been synchronised).
        ``pcm`` / ``voxel`` / ``noise``: the next batch, copied into the static input buffers on the body's stream in front
        of the replay (same contract as ``replay``; None = keep what the buffer holds).  The copy is ordered behind the
        previous body by the stream, so a caller may hand over batch k+1 while pass k is still running."""
        self._poll()
        new_inputs = pcm is not None or voxel is not None or noise is not None
        if new_inputs:
            self._s_body.wait_stream(torch.cuda.current_stream(self.device))   # the caller's stream produced the new inputs
        with torch.cuda.stream(self._s_body):
            for dst, src in zip(self._static, (pcm, voxel, noise)):
                if src is not None:
                    if dst is None:
                        raise ValueError("the pass was captured with in-pass noise draws: it takes no noise tensor")
                    if src.shape != dst.shape or src.dtype != dst.dtype:
                        raise ValueError(f"replay_pipelined: input {tuple(src.shape)} {src.dtype} does not match the "
                                         f"captured {tuple(dst.shape)} {dst.dtype}")
                    dst.copy_(src, non_blocking=True)
            self._enqueue_body(new_inputs)
        with torch.cuda.stream(self._s_head):
            self._s_head.wait_event(self._e_body)
            self._mark("head", 0, self._s_head)
            self._copy(self._pbody.feat, self._h_feat)
            self._copy(self._pbody.style, self._h_style)
            self._e_taken.record(self._s_head)
            self._g_head.replay()
            self._mark("head", 1, self._s_head)
        return self._pout

    # ---- where a replayed pass spends its time: timing events between the branch graphs of the ARRANGEMENT THAT RUNS
    _marks = None

    def _mark(self, name, end, stream):
        if self._marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            self._marks[-1].setdefault(name, [None, None])[end] = e

    def pass_timeline(self, passes=8, drop=2):
        """Start / end of every branch graph of `replay_pipelined`, in ms from the start of the pass's audio front, measured
        with timing events recorded on the branches' own streams between the graph replays of back-to-back passes - i.e.
        in the arrangement the benchmark times, not launch by launch.  Returns {"branches": {name: {"start_ms", "end_ms",
        "ms"}} (medians over the kept passes), "pass_ms": front start -> next front start, "bound_by": the branch that
        ends last among those the next pass waits for}."""
        import statistics
        self._marks = []
        try:
            for _ in range(passes + 1):
                self._marks.append({})
                self.replay_pipelined()
            torch.cuda.synchronize(self.device)
            marks = self._marks
        finally:
            self._marks = None
        rows, period = {}, []
        for k in range(drop, passes):
            t0 = marks[k]["audio_front"][0]
            period.append(t0.elapsed_time(marks[k + 1]["audio_front"][0]))
            for name, (a, b) in marks[k].items():
                rows.setdefault(name, []).append((t0.elapsed_time(a), t0.elapsed_time(b)))
        med = statistics.median
        out = {name: {"start_ms": round(med(v[0] for v in vs), 3), "end_ms": round(med(v[1] for v in vs), 3),
                      "ms": round(med(v[1] - v[0] for v in vs), 3)} for name, vs in sorted(rows.items())}
        body = {n: r for n, r in out.items() if n != "head"}
        return {"branches": out, "pass_ms": round(med(period), 3),
                "bound_by": max(body, key=lambda n: body[n]["end_ms"]),
                "note": "events on the branches' own streams between the graph replays of back-to-back pipelined passes; the "
                        "head of pass k runs beside the front of pass k+1"}

    # ---- utterances of different lengths (the reference's loop takes them one at a time: train_diffusion_prior.py:689-771)
    def run_many(self, pcm_list, voxels, noises=None, generator=None, max_batch=32):
        """Ragged input: ``pcm_list`` holds one 1-D int16/fp32 tensor per utterance (any lengths), ``voxels`` (n, 768) the
        caption features, ``noises`` (T_d+1, n, 1, 128) the DDPM noise per utterance (drawn from ``generator`` in the
        reference's call order when None).  Utterances of equal frame count are batched (per-clip audio statistics make
        the result independent of the grouping); audio beyond the last whole 640-sample frame is dropped as in
        ``process_audio`` (evaluation_functions.py:699-714).  Returns a list of dicts in input order; an utterance shorter
        than one frame yields empty (0, 50) / (0, 3) coefficient arrays (the reference's sample would hold zero frames)."""
        n = len(pcm_list)
        if voxels.shape[0] != n:
            raise ValueError("one caption feature per utterance")
        if n == 0:
            return []
        if self.talking_head.joint_norm:
            raise ValueError("run_many needs per-clip audio statistics (joint_norm=False)")
        if noises is None:
            noises = torch.stack([self.prior.draw_noise(1, generator)[:, 0] for _ in range(n)], 1)
        frames = [int(p.numel()) // 640 for p in pcm_list]
        out = [None] * n
        groups = {}
        for i, t in enumerate(frames):
            groups.setdefault(t, []).append(i)
        for t, idx in sorted(groups.items()):
            if t == 0:
                for i in idx:
                    z = lambda c: torch.zeros((0, c), dtype=torch.float32, device=self.device)
                    out[i] = {"predicted_exp": z(50), "predicted_jaw": z(3), "style_emb": None}
                continue
            for k in range(0, len(idx), max_batch):
                sel = idx[k:k + max_batch]
                pcm = torch.stack([pcm_list[i].reshape(-1)[:t * 640].to(self.device) for i in sel], 0).contiguous()
                sel_t = torch.as_tensor(sel, device=noises.device)
                # synchronous per group, with the safety net (a failed pass is re-run, never returned)
                res = self.run_checked(pcm, voxels[sel].to(self.device).contiguous(),
                                       noises.index_select(1, sel_t).to(self.device).contiguous())
                for j, i in enumerate(sel):
                    out[i] = {"predicted_exp": res["predicted_exp"][j], "predicted_jaw": res["predicted_jaw"][j],
                              "style_emb": res["style_emb"][j]}
        return out
