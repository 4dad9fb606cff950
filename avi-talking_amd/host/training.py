"""Training step of the reference entry point (``train_diffusion_prior.py:434-499``) on the HIP kernels:

    voxel (B,768) --BrainNetwork (train: dropout 0.5/0.15)--> clip_voxels (B,128), proj (B,1,128)
    loss_prior, pred = diffusion_prior(text_embed=clip_voxels, image_embed=clip_target)      (:449, p_losses)
    loss_nce = soft_clip_loss(normalize(proj), normalize(clip_target), temp)                 (:454-469)
    loss = loss_nce + 30 * loss_prior ; backward ; AdamW (4 groups: wd 1e-2 / 0 by name)     (:474-486,997-1004)

Everything numeric runs behind the C ABI: forward/backward GEMMs on the bf16x3 MFMA kernel (dX = dY.W with a
packed W^T, dW = dY^T.X with transposed activations), fused LayerNorm(+GELU+dropout) backward, the 3-token
attention backward, the two losses and a fused flat AdamW that also emits the bf16 hi/lo weight planes.
Parameters, gradients and Adam moments live in flat fp32 buffers (one all-reduce-able gradient buffer; RCCL
buckets are slices of it in backward-completion order).

The random draws of the reference step (timesteps, noise, cond-drop masks, dropout masks) are inputs, so the
step is reproducible against the oracle; ``draw()`` produces them with torch's device RNG.
"""
import math

import torch

from .. import lib as L
from .. import ops
from .diffusion_prior import _rel_pos_bias_table, _rotary_tables, _time_table, cosine_schedule

DIM = 128


def no_decay(name):
    """train_diffusion_prior.py:997-1003 (substring match on parameter names)."""
    return any(nd in name for nd in ("bias", "LayerNorm.bias", "LayerNorm.weight"))


def bucketed_allreduce(flat, spans, group=None, async_op=True):
    """Sum ``flat[a:b]`` over the ranks for every (a, b) in ``spans`` (one collective per span).  With the
    "nccl" backend (= RCCL on ROCm) an async collective is enqueued behind the work already on the current
    stream and runs on RCCL's own stream, so later backward kernels overlap with it; returns the work handles."""
    import torch.distributed as dist
    works = []
    for a, b in spans:
        if b > a:
            works.append(dist.all_reduce(flat[a:b], group=group, async_op=async_op))
    return works


class FlatLayout:
    """Offsets of every parameter in the flat buffers: decay region first, every view 16-byte aligned.
    Pure bookkeeping (no device memory), so the data-parallel span logic can be checked on the CPU."""

    def __init__(self, shapes, order):
        names = [n for n in order if n in shapes]
        missing = [n for n in shapes if n not in names]
        if missing:
            raise ValueError(f"parameters without a slot in the flat layout: {missing[:4]}...")
        dec = [n for n in names if not no_decay(n)]
        nod = [n for n in names if no_decay(n)]
        self.offset, self.shape = {}, {}
        off = 0
        for group in (dec, nod):
            for n in group:
                self.offset[n], self.shape[n] = off, tuple(shapes[n])
                off += (math.prod(shapes[n]) + 3) // 4 * 4          # keep every view 16-byte aligned
            if group is dec:
                off = (off + 7) // 8 * 8                  # the second region starts on a 32-byte boundary (16-byte planes)
                self.n_decay = off                        # [0, n_decay): weight decay; [n_decay, numel): none
        self.numel = off
        self.names = dec + nod

    @classmethod
    def of_state_dict(cls, state_dict, order):
        return cls({n: tuple(t.shape) for n, t in state_dict.items()
                    if t.is_floating_point() and not n.startswith("noise_scheduler")}, order)

    def span(self, first, last):
        """[start, end) of the contiguous run of parameters ``first`` .. ``last`` (padding included)."""
        end = self.offset[last] + (math.prod(self.shape[last]) + 3) // 4 * 4
        return (self.offset[first], end)


def grad_spans(depth=6, n_blocks=4):
    """(first, last) parameter names of the decay-region gradient spans in the order backward completes them
    (``PriorTrainer.forward_backward`` calls ``GradSync.ready`` with exactly these, in this order); together with the
    no-decay tail [n_decay, numel) they cover the flat gradient buffer exactly once
    (tests/test_dist_gloo.py::test_grad_spans_cover_flat_buffer_once)."""
    v, c = "voxel2clip.", "net.causal_transformer."
    # the four 67 MB aligner blocks first (97 % of the volume): backward reaches them as soon as the dX chain of the prior
    # and of the projector has produced their incoming gradient - the parameter gradients of the prior, the projector and
    # lin1 (leaves of backward) are formed AFTER them, under the blocks' all-reduces
    spans = [(v + f"mlp.{b}.0.weight", v + f"mlp.{b}.1.weight") for b in reversed(range(n_blocks))]
    spans.append((v + "lin0.0.weight", v + "lin0.1.weight"))
    spans.append((v + "lin1.weight", v + "projector.8.weight"))
    spans.append(("net.to_time_embeds.0.1.net.0.0.weight", c + "project_out.weight"))   # the whole prior network
    return spans


SHARD_ALIGN = 8          # elements: a rank's slice of a bucket starts on a 32-byte boundary (16-byte aligned 16-bit planes)


def shard_of(a, b, world, rank):
    """How bucket [a, b) of the flat buffers is divided among ``world`` ranks for the SHARDED optimizer:
    -> (lo, hi, main_end): rank r owns [a + r c, a + (r + 1) c) with c = the largest multiple of SHARD_ALIGN such that
    world * c <= b - a; [main_end, b) - fewer than world * SHARD_ALIGN elements - is the bucket's tail, which every rank
    updates redundantly from an all-reduced gradient."""
    c = (b - a) // (world * SHARD_ALIGN) * SHARD_ALIGN
    return a + rank * c, a + (rank + 1) * c, a + world * c


class GradSync:
    """Data-parallel gradient exchange over the flat gradient buffer, bucket by bucket, launched from inside backward the
    moment a bucket is final (``ready``), the no-decay tail and the waits at the end (``finish``).  Device-agnostic: "nccl"
    (= RCCL) on the GPU, "gloo" in the CPU tests.  Replaces the reference's dead ``distributed`` branches
    (train_diffusion_prior.py:338,442,450).

    Two schedules:
      * ``shard=False``: one all-reduce per bucket; every rank then runs the whole AdamW (28 B of HBM traffic per parameter
        on every rank, on identical data).
      * ``shard=True`` (default, AVI_DP_SHARD=0 switches it off): the optimizer state is SHARDED over the ranks, ZeRO-1
        style.  Per bucket: reduce-scatter of the gradient (each rank receives the sum of ITS 1/world slice, in place),
        fused AdamW on that slice only (``on_span``), all-gather of the updated fp32 parameters (in place; ``on_gathered``
        then rebuilds the bf16 planes of the slices this rank did not own).  Same bytes on the wire as the all-reduce
        (reduce-scatter + all-gather IS how a bandwidth-optimal all-reduce moves them), 1/world of the optimizer's HBM
        traffic and time per rank, and the all-gather of bucket i runs beside the AdamW of bucket i + 1.  Fewer than
        world * SHARD_ALIGN elements per bucket (its tail) are all-reduced and updated by every rank.
        gloo has no reduce-scatter: there the slice's sum is taken from an all-reduce of the bucket (same values)."""

    def __init__(self, layout, depth=6, n_blocks=4, process_group=None, shard=None):
        import os
        self.layout, self.pg = layout, process_group
        self.expected = [layout.span(a, b) for a, b in grad_spans(depth, n_blocks)]
        self.works, self.done = [], []
        # AVI_DP_FORCE_COLLECTIVES=1: issue the collectives with a single rank too (rehearsal of the RCCL path on a
        # one-GPU box; an all-reduce over one rank leaves the buffer unchanged)
        self.force = os.environ.get("AVI_DP_FORCE_COLLECTIVES") == "1"
        self.shard = (os.environ.get("AVI_DP_SHARD", "1") == "1") if shard is None else bool(shard)

    def rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.pg) if (dist.is_available() and dist.is_initialized()) else 0

    def _exchange(self, G, a, b):
        """Start the gradient exchange of bucket [a, b): -> list of work handles."""
        import torch.distributed as dist
        if not self.shard:
            return bucketed_allreduce(G, [(a, b)], self.pg)
        world = self.world()
        lo, hi, main = shard_of(a, b, world, self.rank())
        works = []
        if main > a:
            if dist.get_backend(self.pg) == "nccl":
                # in place: the receive slice is send buffer + rank * count (NCCL / RCCL's in-place reduce-scatter)
                works.append(dist.reduce_scatter_tensor(G[lo:hi], G[a:main], group=self.pg, async_op=True))
            else:
                works.append(dist.all_reduce(G[a:main], group=self.pg, async_op=True))
        if b > main:
            works.append(dist.all_reduce(G[main:b], group=self.pg, async_op=True))
        return works

    def _gather(self, P, a, b):
        """Start the all-gather of the updated parameters of bucket [a, b) (every rank contributes its slice, in place)."""
        import torch.distributed as dist
        world = self.world()
        lo, hi, main = shard_of(a, b, world, self.rank())
        if main <= a:
            return None
        if dist.get_backend(self.pg) == "nccl":
            return dist.all_gather_into_tensor(P[a:main], P[lo:hi], group=self.pg, async_op=True)
        c = hi - lo                 # gloo: the list form, receive buffers = views of the parameter buffer
        return dist.all_gather([P[a + r * c:a + (r + 1) * c] for r in range(world)], P[lo:hi].clone(), group=self.pg,
                               async_op=True)

    def world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.pg)
        return 1

    def _collectives(self):
        import torch.distributed as dist
        return self.world() > 1 or (self.force and dist.is_available() and dist.is_initialized())

    def ready(self, G, first, last, launch=True):
        """``launch=False``: bookkeeping only (the segment capture of the DP step walks the announcements without
        issuing collectives)."""
        span = self.layout.span(first, last)
        i = len(self.done)
        if i >= len(self.expected) or span != self.expected[i]:
            raise RuntimeError(f"gradient span {first}..{last} announced out of order (position {i})")
        self.done.append(span)
        if launch and self._collectives():
            self.works.append(self._exchange(G, *span))

    def finish(self, G, launch=True, on_span=None, P=None, on_gathered=None):
        """Exchange the no-decay tail, wait for every bucket; returns the world size (AdamW scales by 1/world).
        ``on_span(a, b)``: the optimizer's update of [a, b), called as soon as the gradient sum of [a, b) has been waited
        for on the current stream (later buckets are still on the wire).  Unsharded: once per bucket - the announced ones
        in announcement order, then the no-decay tail.  Sharded: per bucket for THIS RANK's slice and for the bucket's
        few-element tail; then, with ``P`` (the flat parameter buffer), the all-gather of the bucket's updated parameters
        is started at once, and when all buckets are through ``on_gathered(a, b)`` is called for every range whose
        parameters arrived from another rank (the caller rebuilds what it derives from them)."""
        if len(self.done) != len(self.expected):
            raise RuntimeError(f"{len(self.expected) - len(self.done)} gradient spans were never announced")
        spans = self.done + [(self.layout.n_decay, self.layout.numel)]
        self.done = []
        world = self.world()
        active = launch and self._collectives()
        if active:
            self.works.append(self._exchange(G, self.layout.n_decay, self.layout.numel))
            if len(self.works) != len(spans):
                raise RuntimeError("one gradient exchange per span expected")
        sharded = active and self.shard
        if sharded and on_span is not None and P is None:
            raise ValueError("the sharded schedule updates one slice per rank: finish() needs P to gather the others")
        rank = self.rank() if sharded else 0
        gathers = []
        for i, (a, b) in enumerate(spans):
            if active:
                for w in self.works[i]:
                    w.wait()
            if not sharded:
                if on_span is not None:
                    on_span(a, b)
                continue
            lo, hi, main = shard_of(a, b, world, rank)
            if on_span is not None:
                if hi > lo:
                    on_span(lo, hi)
                if b > main:
                    on_span(main, b)
            if P is not None:
                gathers.append((self._gather(P, a, b), a, lo, hi, main))
        for work, a, lo, hi, main in gathers:
            if work is not None:
                work.wait()
            if on_gathered is not None:
                if lo > a:
                    on_gathered(a, lo)
                if main > hi:
                    on_gathered(hi, main)
        self.works = []
        return world


class _Lin:
    """One Linear layer y = x W^T (+ b) with weights in the flat store."""

    def __init__(self, store, wname, bname=None, need_dx=True, rows=None):
        self.s = store
        self.w = wname
        self.b = bname
        self.N, self.K = rows or store.shape[wname]
        self.need_dx = need_dx
        if need_dx:
            dev = store.P.device
            self.Kp = 64 if self.K <= 64 else (self.K + 127) // 128 * 128        # row padding of a GEMM weight operand
            self.hiT = torch.empty((self.Kp, self.N), dtype=torch.int16, device=dev)
            self.loT = torch.empty_like(self.hiT)

    def refresh_transposed(self):
        if not self.need_dx:
            return
        L.check(L.load().avi_transpose_pack_split(self.s.ptr(self.w), self.N, self.K, self.Kp, self.hiT.data_ptr(),
                                                  self.loT.data_ptr(), L.stream_ptr()), "avi_transpose_pack_split")

    @staticmethod
    def _skinny(A, lda, Whi, Wlo, M, N, K, bias=0, act=ops.ACT_NONE, residual=None, kslice=256):
        """y = act(A W^T + bias) + residual for a few rows and a long K (the aligner at batch-size rows): K slices as
        one batched launch (hundreds of workgroups stream the weights) + avi_splitk_epilogue, instead of a 128-row tile
        GEMM on N/128 workgroups (125 us per 4096x4096 layer)."""
        nz = K // kslice
        dev = A.device
        parts = torch.empty((nz, M, N), dtype=torch.float32, device=dev)
        ops.gemm_raw(A=A.data_ptr(), lda=lda, Whi=Whi, Wlo=Wlo, C_=parts.data_ptr(), ldc=N, M=M, N=N, K=kslice,
                     batch=nz, sA=(kslice, 0), sW=(kslice, 0), sC=(M * N, 0), ldw=K)
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        L.check(L.load().avi_splitk_epilogue(parts.data_ptr(), nz, M * N, M, N, bias or None, None, None, 1e-5, 0, act,
                                             L.ptr(residual), y.data_ptr(), L.stream_ptr()), "avi_splitk_epilogue")
        return y

    @staticmethod
    def _use_skinny(M, N, K):
        """A few row tiles and a long K: one or two workgroups per column tile walking 8..64 K tiles are pure latency
        (40 us for the 256 x 128 x 1024 dX of the prior's feed-forward); K slices spread the walk over the chip."""
        return M <= 256 and K >= 512 and K % 128 == 0 and N % 4 == 0 and N <= 4096

    @staticmethod
    def _kslice(K):
        return 256 if K >= 2048 and K % 256 == 0 else 128

    def fwd(self, x, act=ops.ACT_NONE):
        M = x.numel() // self.K
        if self._use_skinny(M, self.N, self.K):
            return self._skinny(x, self.K, self.s.hi_ptr(self.w), self.s.lo_ptr(self.w), M, self.N, self.K,
                                bias=self.s.ptr(self.b) if self.b else 0, act=act, kslice=self._kslice(self.K))
        y = torch.empty((M, self.N), dtype=torch.float32, device=x.device)
        ops.gemm_raw(A=x.data_ptr(), lda=self.K, Whi=self.s.hi_ptr(self.w), Wlo=self.s.lo_ptr(self.w), C_=y.data_ptr(),
                     ldc=self.N, M=M, N=self.N, K=self.K, bias=self.s.ptr(self.b) if self.b else 0, act=act)
        return y

    def bwd(self, x, dy, dx_residual=None):
        """Accumulate nothing: writes dW (and db) into the flat gradient buffer, returns dx (+ residual)."""
        self.bwd_dw(x, dy)
        if not self.need_dx:
            return None
        return self.bwd_dx(dy, dx_residual)

    def bwd_dw(self, x, dy):
        """dW = dy^T x (and db = column sums of dy) into the flat gradient buffer.  A leaf of the backward pass: callers
        may run it any time after ``dy`` exists (the data-parallel step orders these launches by bucket size)."""
        so = L.load()
        M = dy.numel() // self.N
        if M % 64:
            raise ValueError("training batch rows must be a multiple of 64")
        dev = dy.device
        dyT = torch.empty((self.N, M), dtype=torch.float32, device=dev)
        Kp = 64 if self.K <= 64 else (self.K + 127) // 128 * 128
        xhi = torch.empty((Kp, M), dtype=torch.int16, device=dev)      # x^T as [K][M] hi/lo: the "weight" operand of
        xlo = torch.empty_like(xhi)                                    # the dW GEMM, transposed and split in one pass
        jobs = (L.AviTransposeJob * 3)()              # dY^T (fp32), x^T (planes) and the bias gradient in ONE launch
        jobs[0].in_, jobs[0].out, jobs[0].R, jobs[0].C = dy.data_ptr(), dyT.data_ptr(), M, self.N
        jobs[1].in_, jobs[1].hi, jobs[1].lo = x.data_ptr(), xhi.data_ptr(), xlo.data_ptr()
        jobs[1].R, jobs[1].C, jobs[1].C_pad = M, self.K, Kp
        if self.b:
            jobs[2].in_, jobs[2].colsum, jobs[2].R, jobs[2].C = dy.data_ptr(), self.s.gptr(self.b), M, self.N
        L.check(so.avi_transpose_jobs(jobs, 3 if self.b else 2, L.stream_ptr()), "avi_transpose_jobs")
        ops.gemm_raw(A=dyT.data_ptr(), lda=M, Whi=xhi.data_ptr(), Wlo=xlo.data_ptr(),
                     C_=self.s.gptr(self.w), ldc=self.K, M=self.N, N=self.K, K=M)

    def bwd_dx(self, dy, dx_residual=None):
        """dx = dy . W (+ residual) alone: the parameter gradient of this layer is formed elsewhere (the prior's layers
        defer theirs to one batched launch per matrix, PriorTrainer._deferred_dw)."""
        M = dy.numel() // self.N
        if self._use_skinny(M, self.K, self.N):
            return self._skinny(dy, self.N, self.hiT.data_ptr(), self.loT.data_ptr(), M, self.K, self.N,
                                residual=dx_residual, kslice=self._kslice(self.N))
        dx = torch.empty((M, self.K), dtype=torch.float32, device=dy.device)
        ops.gemm_raw(A=dy.data_ptr(), lda=self.N, Whi=self.hiT.data_ptr(), Wlo=self.loT.data_ptr(), C_=dx.data_ptr(),
                     ldc=self.K, M=M, N=self.K, K=self.N, R=L.ptr(dx_residual), ldr=self.K)
        return dx


class ParamStore:
    """Flat fp32 parameter / gradient / Adam-moment buffers with named views; decay region first."""

    def __init__(self, state_dict, device, order):
        self.layout = lay = FlatLayout.of_state_dict(state_dict, order)
        self.offset, self.shape, self.n_decay, self.numel, self.names = (lay.offset, lay.shape, lay.n_decay, lay.numel,
                                                                         lay.names)
        off = lay.numel
        self.P = torch.zeros(off, dtype=torch.float32, device=device)
        for n in self.names:
            self.view(n).copy_(state_dict[n].to(device, torch.float32))
        self.G = torch.zeros_like(self.P)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.HI = torch.zeros(off, dtype=torch.int16, device=device)
        self.LO = torch.zeros_like(self.HI)
        L.check(L.load().avi_pack_weight_split(self.P.data_ptr(), 1, off, 1, self.HI.data_ptr(), self.LO.data_ptr(),
                                               L.stream_ptr()), "pack")

    def view(self, n, buf=None):
        buf = self.P if buf is None else buf
        o = self.offset[n]
        return buf[o:o + math.prod(self.shape[n])].view(self.shape[n])

    def grad(self, n):
        return self.view(n, self.G)

    def ptr(self, n):
        return self.P.data_ptr() + 4 * self.offset[n]

    def gptr(self, n):
        return self.G.data_ptr() + 4 * self.offset[n]

    def hi_ptr(self, n):
        return self.HI.data_ptr() + 2 * self.offset[n]

    def lo_ptr(self, n):
        return self.LO.data_ptr() + 2 * self.offset[n]

    def state_dict(self):
        return {n: self.view(n).detach().clone() for n in self.names}


def _layout(depth=6, n_blocks=4):
    """Flat order = reverse of backward completion, so gradient buckets finish front-to-back... kept simple:
    aligner first, prior after; fused matrices (to_q | to_kv) adjacent."""
    v, n = "voxel2clip.", "net."
    order = [v + "lin0.0.weight", v + "lin0.0.bias", v + "lin0.1.weight", v + "lin0.1.bias"]
    for b in range(n_blocks):
        order += [v + f"mlp.{b}.0.weight", v + f"mlp.{b}.0.bias", v + f"mlp.{b}.1.weight", v + f"mlp.{b}.1.bias"]
    order += [v + "lin1.weight", v + "lin1.bias"]
    for i in (0, 2, 3, 5, 6, 8):
        order += [v + f"projector.{i}.weight", v + f"projector.{i}.bias"]
    t = n + "to_time_embeds.0.1.net."
    order += [t + "0.0.weight", t + "0.0.bias", t + "1.0.weight", t + "1.0.bias", t + "2.weight", t + "2.bias",
              n + "learned_query", n + "null_brain_embeds", n + "null_image_embed"]
    c = n + "causal_transformer."
    order += [c + "rel_pos_bias.relative_attention_bias.weight"]
    for l in range(depth):
        a, f = c + f"layers.{l}.0.", c + f"layers.{l}.1."
        order += [a + "norm.g", a + "to_q.weight", a + "to_kv.weight", a + "null_kv", a + "to_out.0.weight",
                  a + "to_out.1.g", f + "0.g", f + "1.weight", f + "5.weight"]
    order += [c + "norm.g", c + "project_out.weight"]
    return order


class PriorTrainer:
    def __init__(self, state_dict, device="cuda", lr=1e-4, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8,
                 timesteps=100, prior_mult=30.0, depth=6, n_blocks=4, process_group=None):
        self.device = torch.device(device)
        self.depth, self.n_blocks = depth, n_blocks
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.prior_mult = prior_mult
        self.step_count = 0
        self.pg = process_group
        self.store = S = ParamStore(state_dict, self.device, _layout(depth, n_blocks))
        v, n = "voxel2clip.", "net."
        self.lin0 = _Lin(S, v + "lin0.0.weight", v + "lin0.0.bias", need_dx=False)
        self.mlp = [_Lin(S, v + f"mlp.{b}.0.weight", v + f"mlp.{b}.0.bias") for b in range(n_blocks)]
        self.lin1 = _Lin(S, v + "lin1.weight", v + "lin1.bias")
        self.proj = [_Lin(S, v + f"projector.{i}.weight", v + f"projector.{i}.bias") for i in (2, 5, 8)]
        t = n + "to_time_embeds.0.1.net."
        self.tm = [_Lin(S, t + "0.0.weight", t + "0.0.bias", need_dx=False), _Lin(S, t + "1.0.weight", t + "1.0.bias"),
                   _Lin(S, t + "2.weight", t + "2.bias")]
        c = n + "causal_transformer."
        self.layers = []
        for l in range(depth):
            a, f = c + f"layers.{l}.0.", c + f"layers.{l}.1."
            self.layers.append(dict(
                a=a, f=f,
                qkv=_Lin(S, a + "to_q.weight", rows=(640, DIM)),            # to_q | to_kv are adjacent in the store
                out=_Lin(S, a + "to_out.0.weight"), w1=_Lin(S, f + "1.weight"), w2=_Lin(S, f + "5.weight")))
            assert S.offset[a + "to_kv.weight"] == S.offset[a + "to_q.weight"] + 512 * DIM
        self.cproj = _Lin(S, c + "project_out.weight")
        self.c = c
        self.lins = ([self.lin0, self.lin1] + self.mlp + self.proj + self.tm + [self.cproj] +
                     [ly[k] for ly in self.layers for k in ("qkv", "out", "w1", "w2")])
        sched = self.sched = cosine_schedule(timesteps)
        dv = lambda x: x.to(self.device).contiguous()
        self.sqrt_ac, self.sqrt_1mac = dv(sched["sqrt_alphas_cumprod"]), dv(sched["sqrt_one_minus_alphas_cumprod"])
        self.time_table = dv(_time_table(timesteps))
        rc, rs = _rotary_tables(3)
        self.rot_cos, self.rot_sin = dv(rc), dv(rs)
        q = torch.arange(3)[:, None]
        k = torch.arange(4)[None, :]
        self.rel_index = dv(torch.clamp(q - k, min=0))                        # (3,4) bucket of each (i,j)
        self.dyn = torch.zeros(4, dtype=torch.float32, device=self.device)   # lr, bc1, rsqrt(bc2) for the graph
        self.sync = GradSync(S.layout, depth, n_blocks, process_group)
        self._cut = None
        self._ttable = None
        self._ws = {}
        import os
        # AVI_TRAIN_FUSED_FWD=0: the forward of the denoiser as the chain of ~55 launches it used to be (A/B switch, tests)
        self.fused_forward = os.environ.get("AVI_TRAIN_FUSED_FWD", "1") == "1"
        self.fwd_samples_per_group = int(os.environ.get("AVI_TRAIN_FWD_SPG", "1"))   # one sample per workgroup: 64 CUs at B = 64
        # AVI_TRAIN_FUSED_BWD=0: the dX chain of the denoiser's backward as ~90 launches (needs the fused forward's dumps)
        self.fused_backward = self.fused_forward and os.environ.get("AVI_TRAIN_FUSED_BWD", "1") == "1"
        self.bwd_samples_per_group = int(os.environ.get("AVI_TRAIN_BWD_SPG", "1"))
        self.atomics = os.environ.get("AVI_TRAIN_ATOMICS", "0") == "1"
        self.refresh()

    # ------------------------------------------------------------------ data parallel (C1 of SURVEY.md section 2)
    def _grads_ready(self, first, last):
        """Called as soon as the backward pass has finished writing the decay-region gradients from parameter
        ``first`` through ``last`` (contiguous in the flat buffer): start their all-reduce now, so it overlaps
        with the rest of backward (the 67 M-parameter aligner blocks dominate the 311 MB volume)."""
        if self._cut is not None:                 # segment capture (capture_step_dp): end a graph here, start the next
            self._cut(first, last)
        else:
            self.sync.ready(self.store.G, first, last)

    # ------------------------------------------------------------------ deferred parameter gradients of the prior layers
    _DW = (("qkv", 640, DIM), ("out", DIM, 512), ("w1", 1024, DIM), ("w2", DIM, 512))     # (lin, N out, K in)

    def _dw_ws(self, B):
        """Persistent per-batch-size workspace: the inputs x and output gradients dy of the four matrices of every
        prior layer, stacked over the layers, written in place by the kernels that produce them; their transposes; and
        a device job table per matrix (addresses never change, so it is built once).  With these the 24 weight-gradient
        GEMMs of the 6 layers and their 24 transpose launches collapse into 4 + 4 launches at the end of the prior's
        backward pass: dW is a leaf of the graph, nothing in backward waits for it."""
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        R, dev, Ld = 3 * B, self.device, self.depth
        ws = {}
        for name, N, K in self._DW:
            Kp = 64 if K <= 64 else (K + 127) // 128 * 128
            x = torch.empty((Ld, R, K), dtype=torch.float32, device=dev)
            dy = torch.empty((Ld, R, N), dtype=torch.float32, device=dev)
            dyT = torch.empty((Ld, N, R), dtype=torch.float32, device=dev)
            xhi = torch.empty((Ld, Kp, R), dtype=torch.int16, device=dev)
            xlo = torch.empty_like(xhi)
            jobs = (L.AviTransposeJob * (2 * Ld))()
            total = 0
            for l in range(Ld):
                for jb, (src, rows_, cols, kw) in zip((jobs[2 * l], jobs[2 * l + 1]),
                                                      ((dy[l], R, N, dict(out=dyT[l].data_ptr())),
                                                       (x[l], R, K, dict(hi=xhi[l].data_ptr(), lo=xlo[l].data_ptr(), C_pad=Kp)))):
                    jb.in_, jb.R, jb.C = src.data_ptr(), rows_, cols
                    for k_, v_ in kw.items():
                        setattr(jb, k_, v_)
                    jb.first_block = total
                    total += jb.blocks()
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
            ws[name] = dict(x=x, dy=dy, dyT=dyT, xhi=xhi, xlo=xlo, Kp=Kp, table=(raw, 2 * Ld, total))
        # what the one-launch forward (avi_prior_train_forward) stores besides the four x arrays above
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        ws["fwd"] = dict(tok_in=f(Ld, R, DIM), qkv=f(Ld, R, 640), o1=f(Ld, R, DIM), tokm=f(Ld, R, DIM), hff=f(Ld, R, 1024),
                         tok_out=f(R, DIM), fin=f(R, DIM), po=f(R, DIM), rel_bias=f(8, 3, 4))
        self._ws[B] = ws
        return ws

    def _fused_forward_setup(self):
        """Constants of the one-launch forward, built once: fragment-major plane buffers of the denoiser's matrices with
        the device job table that re-packs them from the flat bf16 hi/lo parameter planes every step, and the struct of
        small-vector pointers into the flat parameter buffer (addresses never change)."""
        S, dev = self.store, self.device
        c = self.c
        mats = []
        for l in range(self.depth):
            a, f = c + f"layers.{l}.0.", c + f"layers.{l}.1."
            mats += [(a + "to_q.weight", 640, DIM), (a + "to_out.0.weight", DIM, 512), (f + "1.weight", 1024, DIM),
                     (f + "5.weight", DIM, 512)]
        mats.append((c + "project_out.weight", DIM, DIM))
        jobs = (L.AviPlaneJob * len(mats))()
        self._fplanes = []
        total = 0
        for jb, (name, N, K) in zip(jobs, mats):
            hi = torch.empty(N * K, dtype=torch.int16, device=dev)
            lo = torch.empty_like(hi)
            self._fplanes.append((hi, lo))
            jb.src_hi, jb.src_lo, jb.dst_hi, jb.dst_lo = S.hi_ptr(name), S.lo_ptr(name), hi.data_ptr(), lo.data_ptr()
            jb.N, jb.K, jb.first_block, jb.transpose = N, K, total, 0
            total += jb.blocks()
        self._fjobs = (torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev), len(mats), total)
        pl = L.AviPriorPlanes()
        for l in range(self.depth):
            lp = pl.layer[l]
            (lp.qkv_hi, lp.qkv_lo), (lp.out_hi, lp.out_lo), (lp.w1_hi, lp.w1_lo), (lp.w2_hi, lp.w2_lo) = [
                (h.data_ptr(), o.data_ptr()) for h, o in self._fplanes[4 * l:4 * l + 4]]
        pl.proj_hi, pl.proj_lo = (t.data_ptr() for t in self._fplanes[-1])
        cw = L.AviPriorWeights()
        cw.depth, cw.timesteps = self.depth, 100
        cw.rot_cos, cw.rot_sin = self.rot_cos.data_ptr(), self.rot_sin.data_ptr()
        for l in range(self.depth):
            a, f = c + f"layers.{l}.0.", c + f"layers.{l}.1."
            ly = cw.layer[l]
            ly.norm_g, ly.null_kv, ly.out_g, ly.ff_g = S.ptr(a + "norm.g"), S.ptr(a + "null_kv"), S.ptr(a + "to_out.1.g"), S.ptr(f + "0.g")
        cw.final_g = S.ptr(c + "norm.g")
        self._fplanes_struct, self._fweights = pl, cw
        # the same matrices TRANSPOSED (dX = dY . W runs the linear over W^T), for the one-launch backward
        tjobs = (L.AviPlaneJob * (4 * self.depth))()
        self._tplanes = []
        ttotal = 0
        for jb, (name, N, K) in zip(tjobs, mats[:-1]):
            hi = torch.empty(N * K, dtype=torch.int16, device=dev)
            lo = torch.empty_like(hi)
            self._tplanes.append((hi, lo))
            jb.src_hi, jb.src_lo, jb.dst_hi, jb.dst_lo = S.hi_ptr(name), S.lo_ptr(name), hi.data_ptr(), lo.data_ptr()
            jb.N, jb.K, jb.first_block, jb.transpose = K, N, ttotal, 1        # the packed matrix is W^T: [K_w][N_w]
            ttotal += jb.blocks()
        self._tjobs = (torch.frombuffer(bytearray(bytes(tjobs)), dtype=torch.uint8).to(dev), 4 * self.depth, ttotal)
        pt = L.AviPriorPlanes()
        for l in range(self.depth):
            lp = pt.layer[l]
            (lp.qkv_hi, lp.qkv_lo), (lp.out_hi, lp.out_lo), (lp.w1_hi, lp.w1_lo), (lp.w2_hi, lp.w2_lo) = [
                (h.data_ptr(), o.data_ptr()) for h, o in self._tplanes[4 * l:4 * l + 4]]
        self._tplanes_struct = pt
        gg = L.AviPriorGainGrads()
        for l in range(self.depth):
            a, f = c + f"layers.{l}.0.", c + f"layers.{l}.1."
            gg.g[l][0], gg.g[l][1], gg.g[l][2] = S.gptr(a + "norm.g"), S.gptr(a + "to_out.1.g"), S.gptr(f + "0.g")
        self._gain_grads = gg

    def _fused_backward(self, ws, dtok, B, drel):
        """The dX chain of the six layers in one launch (+ the gain-gradient reduction): fills the stacked dy buffers the
        deferred dW launches read, returns the gradient of the token rows."""
        import ctypes as C
        so = L.load()
        S, R = self.store, 3 * B
        raw, n, total = self._tjobs
        L.check(so.avi_pack_fragment_planes(raw.data_ptr(), n, total, L.stream_ptr()), "avi_pack_fragment_planes")
        fw = ws["fwd"]
        groups = -(-B // self.bwd_samples_per_group)
        part = fw.get("dgamma_part")
        if part is None or part.numel() < groups * self.depth * 3 * DIM:
            part = fw["dgamma_part"] = torch.empty(groups * self.depth * 3 * DIM, dtype=torch.float32, device=self.device)
            fw["dtok0"] = torch.empty((R, DIM), dtype=torch.float32, device=self.device)
            # per (layer, sample) contributions to the null-kv and relative-bias gradients, summed in sample order by the
            # launch's last kernel: no float atomics, the step is run-to-run deterministic (AVI_TRAIN_ATOMICS=1: the old way)
            fw["attn_part"] = torch.empty(self.depth * B * 224, dtype=torch.float32, device=self.device)
        d = L.AviPriorTrainBwd()
        d.dtok_top = dtok.data_ptr()
        d.tok_in, d.qkv, d.o1, d.tokm, d.hff = (fw[k].data_ptr() for k in ("tok_in", "qkv", "o1", "tokm", "hff"))
        d.dy_w2, d.dy_w1, d.dy_out, d.dy_qkv = (ws[k]["dy"].data_ptr() for k in ("w2", "w1", "out", "qkv"))
        d.dtok0, d.dgamma_part, d.drel = fw["dtok0"].data_ptr(), part.data_ptr(), drel.data_ptr()
        d.attn_part = None if self.atomics else fw["attn_part"].data_ptr()
        for l in range(self.depth):
            d.dnull_kv[l] = S.gptr(self.c + f"layers.{l}.0.null_kv")
        L.check(so.avi_prior_train_backward(C.byref(self._fweights), C.byref(self._tplanes_struct), C.byref(d),
                                            C.byref(self._gain_grads), B, self.bwd_samples_per_group, L.stream_ptr()),
                "avi_prior_train_backward")
        return fw["dtok0"]

    def _fused_forward(self, ws, tok, B, rel_bias):
        """tokens -> all six layers -> project_out in two launches (plane re-pack + avi_prior_train_forward).  Returns
        (saved, tok_out, fin, po) in the form the backward pass below expects."""
        import ctypes as C
        so = L.load()
        if getattr(self, "_fjobs", None) is None:
            self._fused_forward_setup()
        raw, n, total = self._fjobs
        L.check(so.avi_pack_fragment_planes(raw.data_ptr(), n, total, L.stream_ptr()), "avi_pack_fragment_planes")
        fw = ws["fwd"]
        d = L.AviPriorTrainDump()
        d.tok0 = tok.data_ptr()
        d.tok_in, d.qkv, d.o1, d.tokm, d.hff = (fw[k].data_ptr() for k in ("tok_in", "qkv", "o1", "tokm", "hff"))
        d.n1, d.ao, d.n2, d.sw = (ws[k]["x"].data_ptr() for k in ("qkv", "out", "w1", "w2"))
        d.tok_out, d.fin, d.po = fw["tok_out"].data_ptr(), fw["fin"].data_ptr(), fw["po"].data_ptr()
        self._fweights.rel_bias = rel_bias.data_ptr()
        L.check(so.avi_prior_train_forward(C.byref(self._fweights), C.byref(self._fplanes_struct), C.byref(d), B,
                                           self.fwd_samples_per_group, L.stream_ptr()), "avi_prior_train_forward")
        saved = [(fw["tok_in"][l], ws["qkv"]["x"][l], fw["qkv"][l], ws["out"]["x"][l], fw["o1"][l], fw["tokm"][l],
                  ws["w1"]["x"][l], fw["hff"][l], ws["w2"]["x"][l]) for l in range(self.depth)]
        return saved, fw["tok_out"], fw["fin"], fw["po"]

    def _deferred_dw(self, ws, R):
        """dW[l] = dy[l]^T . x[l] for the four matrices of all layers: one transpose launch + one batched GEMM each."""
        S, so = self.store, L.load()
        c0, c1 = self.c + "layers.0.", self.c + "layers.1."
        wname = {"qkv": "0.to_q.weight", "out": "0.to_out.0.weight", "w1": "1.1.weight", "w2": "1.5.weight"}
        for name, N, K in self._DW:
            w = ws[name]
            raw, n, total = w["table"]
            L.check(so.avi_transpose_table(raw.data_ptr(), n, total, L.stream_ptr()), "avi_transpose_table")
            stride = (S.offset[c1 + wname[name]] - S.offset[c0 + wname[name]]) if self.depth > 1 else 0
            ops.gemm_raw(A=w["dyT"].data_ptr(), lda=R, Whi=w["xhi"].data_ptr(), Wlo=w["xlo"].data_ptr(),
                         C_=S.gptr(c0 + wname[name]), ldc=K, M=N, N=K, K=R, batch=self.depth, sA=(N * R, 0),
                         sW=(w["Kp"] * R, 0), sC=(stride, 0))

    # ------------------------------------------------------------------ helpers
    def refresh(self):
        """Transposed split planes of every weight that needs a dX GEMM, rebuilt after each optimizer step: one
        launch over a device table of jobs (built once: parameter and plane addresses never change)."""
        if self._ttable is None:
            lins = [l for l in self.lins if l.need_dx]
            jobs = (L.AviTransposeJob * len(lins))()
            total = 0
            for jb, l in zip(jobs, lins):
                jb.in_, jb.hi, jb.lo = l.s.ptr(l.w), l.hiT.data_ptr(), l.loT.data_ptr()
                jb.R, jb.C, jb.C_pad, jb.first_block = l.N, l.K, l.Kp, total
                total += jb.blocks()
            raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.device)
            self._ttable = (raw, len(lins), total)
        raw, n, total = self._ttable
        L.check(L.load().avi_transpose_table(raw.data_ptr(), n, total, L.stream_ptr()), "avi_transpose_table")

    def reload_planes(self):
        """Rebuild every derived copy of the parameters (bf16 hi/lo planes, transposed planes) after the flat fp32
        buffer was written from outside (checkpoint resume)."""
        S = self.store
        L.check(L.load().avi_pack_weight_split(S.P.data_ptr(), 1, S.numel, 1, S.HI.data_ptr(), S.LO.data_ptr(),
                                               L.stream_ptr()), "pack")
        self.refresh()

    def _ln(self, x, g, b=None, act=ops.ACT_NONE, mask=None, residual=None, stable=0, out=None):
        S = self.store
        out = torch.empty_like(x) if out is None else out
        C = x.shape[-1]
        L.check(L.load().avi_layernorm_ex(x.data_ptr(), x.numel() // C, C, S.ptr(g), S.ptr(b) if b else 0, 1e-5, act,
                                          L.ptr(mask), L.ptr(residual), stable, out.data_ptr(), L.stream_ptr()), "ln")
        return out

    def _ln_bwd(self, x, dy, g, b=None, act=ops.ACT_NONE, mask=None, stable=0, dx_add=None, out=None):
        S = self.store
        C = x.shape[-1]
        rows = x.numel() // C
        dx = torch.empty_like(x) if out is None else out
        stats = torch.empty(3 * rows, dtype=torch.float32, device=x.device)
        L.check(L.load().avi_layernorm_bwd(x.data_ptr(), dy.data_ptr(), S.ptr(g), S.ptr(b) if b else 0, L.ptr(mask),
                                           rows, C, 1e-5, act, stable, L.ptr(dx_add), dx.data_ptr(), S.gptr(g),
                                           S.gptr(b) if b else 0, 0, stats.data_ptr(), L.stream_ptr()), "ln_bwd")
        return dx

    def draw(self, B, generator=None):
        """The random inputs of one reference step: times (sample_random_times), noise, cond-drop keep masks
        (prob_mask_like(1 - 0.2)), dropout keep masks scaled by 1/(1-p)."""
        dev = self.device
        r = lambda *s: torch.rand(*s, device=dev, generator=generator)
        masks = [(r(B, 4096) >= 0.5).float() / 0.5] + [(r(B, 4096) >= 0.15).float() / 0.85 for _ in range(self.n_blocks)]
        return dict(times=torch.randint(0, 100, (B,), device=dev, generator=generator).to(torch.int32),
                    noise=torch.randn(B, DIM, device=dev, generator=generator),
                    brain_keep=(r(B) < 0.8).to(torch.uint8), image_keep=(r(B) < 0.8).to(torch.uint8),
                    dropout_masks=masks)

    def draw_device(self, rng, rand):
        """The same draws from the library's device-resident Philox stream (host/rng.py), written INTO the tensors of
        ``rand`` (a dict shaped like ``draw``'s) by launches on the current stream, then the stream's offset moves on: inside
        a captured step these are nodes of the graph, so every replay trains on fresh times / noise / masks, as the
        reference's step does (train_diffusion_prior.py:449 -> models/diffusion_prior.py:445,453,255-259; Dropout 0.5 /
        0.15 of BrainNetwork :62-75).  One subsequence per tensor."""
        from . import rng as R
        rng.fill(rand["times"], R.RANDINT_I32, 0, 100.0)
        rng.fill(rand["noise"], R.NORMAL, 1)
        rng.fill(rand["brain_keep"], R.BERNOULLI_U8, 2, 0.8)
        rng.fill(rand["image_keep"], R.BERNOULLI_U8, 3, 0.8)
        for i, m in enumerate(rand["dropout_masks"]):
            rng.fill(m, R.KEEP_SCALED, 4 + i, 0.5 if i == 0 else 0.15)
        rng.advance(1)
        return rand

    # ------------------------------------------------------------------ forward + backward
    def forward_backward(self, voxel, clip_target, times, noise, temp, brain_keep=None, image_keep=None,
                         dropout_masks=None):
        so = L.load()
        S, dev = self.store, self.device
        B = voxel.shape[0]
        R = 3 * B
        st = L.stream_ptr
        v = "voxel2clip."
        x = voxel.to(dev, torch.float32).contiguous()
        target = clip_target.reshape(B, DIM).to(dev, torch.float32).contiguous()
        dm = dropout_masks or [None] * (self.n_blocks + 1)
        if not self.fused_backward:
            # the launch-chain backward adds the null-kv gradients into the buffer with atomics; in the fused path every
            # gradient element is STORED by exactly one launch (GEMMs, column sums, LayerNorm / gain reductions, token and
            # relative-bias backward; the null-kv accumulators are cleared by avi_prior_train_backward itself), so the
            # 311 MB clear is skipped (tests/test_gpu_training.py runs two steps and compares)
            L.check(so.avi_zero(S.G.data_ptr(), S.numel, st()), "zero")

        # ---- BrainNetwork forward (models/diffusion_prior.py:95-117, train mode)
        h0p = self.lin0.fwd(x)
        h = [self._ln(h0p, v + "lin0.1.weight", v + "lin0.1.bias", ops.ACT_GELU, dm[0])]
        yp = []
        for b in range(self.n_blocks):
            yp.append(self.mlp[b].fwd(h[b]))
            h.append(self._ln(yp[b], v + f"mlp.{b}.1.weight", v + f"mlp.{b}.1.bias", ops.ACT_GELU, dm[b + 1], h[b]))
        out = self.lin1.fwd(h[-1])                                            # clip_voxels (B,128)
        z0 = self._ln(out, v + "projector.0.weight", v + "projector.0.bias", ops.ACT_GELU)
        z1p = self.proj[0].fwd(z0)
        z1 = self._ln(z1p, v + "projector.3.weight", v + "projector.3.bias", ops.ACT_GELU)
        z2p = self.proj[1].fwd(z1)
        z2 = self._ln(z2p, v + "projector.6.weight", v + "projector.6.bias", ops.ACT_GELU)
        proj = self.proj[2].fwd(z2)

        # ---- prior forward (p_losses, models/diffusion_prior.py:369-400)
        t32 = times if (times.dtype == torch.int32 and times.device == dev) else times.to(dev, torch.int32)
        te0 = torch.empty((B, DIM), dtype=torch.float32, device=dev)
        L.check(so.avi_copy_rows(self.time_table.data_ptr(), DIM, t32.data_ptr(), te0.data_ptr(), DIM, B, DIM, st()), "temb")
        a1p = self.tm[0].fwd(te0)
        a1 = torch.empty_like(a1p)
        L.check(so.avi_act_fwd(a1p.data_ptr(), a1p.numel(), ops.ACT_SILU, a1.data_ptr(), st()), "act")
        a2p = self.tm[1].fwd(a1)
        a2 = torch.empty_like(a2p)
        L.check(so.avi_act_fwd(a2p.data_ptr(), a2p.numel(), ops.ACT_SILU, a2.data_ptr(), st()), "act")
        temb = self.tm[2].fwd(a2)
        nz = noise.reshape(B, DIM).to(dev, torch.float32).contiguous()
        bk = None if brain_keep is None else brain_keep.to(dev, torch.uint8).contiguous()
        ik = None if image_keep is None else image_keep.to(dev, torch.uint8).contiguous()
        x0 = torch.empty((B, DIM), dtype=torch.float32, device=dev)
        tok = torch.empty((R, DIM), dtype=torch.float32, device=dev)
        L.check(so.avi_prior_tokens_fwd(target.data_ptr(), nz.data_ptr(), t32.data_ptr(), self.sqrt_ac.data_ptr(),
                                        self.sqrt_1mac.data_ptr(), DIM ** 0.5, out.data_ptr(), temb.data_ptr(),
                                        L.ptr(bk), L.ptr(ik), S.ptr("net.null_brain_embeds"),
                                        S.ptr("net.null_image_embed"), S.ptr("net.learned_query"), B, x0.data_ptr(),
                                        tok.data_ptr(), st()), "tokens_fwd")
        rel_name = self.c + "rel_pos_bias.relative_attention_bias.weight"
        ws = self._dw_ws(B)
        rel_bias = ws["fwd"]["rel_bias"]
        L.check(so.avi_prior_rel_bias(S.ptr(rel_name), rel_bias.data_ptr(), None, None, 8, 3, st()), "rel_bias")
        if self.fused_forward:
            saved, tok, fin, po = self._fused_forward(ws, tok, B, rel_bias)
        else:
            saved = []
            for li, ly in enumerate(self.layers):
                a, f = ly["a"], ly["f"]
                n1 = self._ln(tok, a + "norm.g", out=ws["qkv"]["x"][li])
                qkv = ly["qkv"].fwd(n1)
                ao = ws["out"]["x"][li]
                L.check(so.avi_prior_attn_fwd(qkv.data_ptr(), S.ptr(a + "null_kv"), rel_bias.data_ptr(),
                                              self.rot_cos.data_ptr(), self.rot_sin.data_ptr(), B, ao.data_ptr(), st()), "attn")
                o1 = ly["out"].fwd(ao)
                tokm = self._ln(o1, a + "to_out.1.g", residual=tok)
                n2 = self._ln(tokm, f + "0.g", out=ws["w1"]["x"][li])
                hff = ly["w1"].fwd(n2)
                sw = ws["w2"]["x"][li]
                L.check(so.avi_swiglu_fwd(hff.data_ptr(), R, 512, sw.data_ptr(), st()), "swiglu")
                tok_next = torch.empty_like(tok)
                ops.gemm_raw(A=sw.data_ptr(), lda=512, Whi=S.hi_ptr(f + "5.weight"), Wlo=S.lo_ptr(f + "5.weight"),
                             C_=tok_next.data_ptr(), ldc=DIM, M=R, N=DIM, K=512, R=tokm.data_ptr(), ldr=DIM)
                saved.append((tok, n1, qkv, ao, o1, tokm, n2, hff, sw))
                tok = tok_next
            fin = self._ln(tok, self.c + "norm.g", stable=1)
            po = self.cproj.fwd(fin)
        pred = torch.empty((B, DIM), dtype=torch.float32, device=dev)                 # tokens[:, -1] (:311)
        L.check(so.avi_copy_rows(po.data_ptr() + 2 * DIM * 4, 3 * DIM, None, pred.data_ptr(), DIM, B, DIM, st()), "pred")

        # ---- losses (+ their gradients)
        zbuf = torch.empty(4 + R * DIM + 96, dtype=torch.float32, device=dev)         # losses | d(project_out) | d(rel bias)
        L.check(so.avi_zero(zbuf.data_ptr(), zbuf.numel(), st()), "zero")
        losses = zbuf[0:2]
        dpred = torch.empty_like(pred)
        L.check(so.avi_mse_loss(pred.data_ptr(), x0.data_ptr(), B * DIM, self.prior_mult, losses.data_ptr(),
                                dpred.data_ptr(), st()), "mse")
        dproj = torch.empty_like(proj)
        scratch = torch.empty(2 * B * DIM + 2 * B + 3 * B * B, dtype=torch.float32, device=dev)
        L.check(so.avi_soft_clip_loss(proj.data_ptr(), target.data_ptr(), B, DIM, float(temp), 1.0,
                                      losses.data_ptr() + 4, dproj.data_ptr(), scratch.data_ptr(), st()), "clip")

        # ---- prior backward
        dpo = zbuf[4:4 + R * DIM].view(B, 3, DIM)
        L.check(so.avi_copy_rows(dpred.data_ptr(), DIM, None, dpo.data_ptr() + 2 * DIM * 4, 3 * DIM, B, DIM, st()), "dpo")
        dfin = self.cproj.bwd(fin, dpo.view(R, DIM))
        drel = zbuf[4 + R * DIM:4 + R * DIM + 96].view(8, 3, 4)
        if self.fused_backward:
            dtok = self._ln_bwd(tok, dfin, self.c + "norm.g", stable=1)
            dtok = self._fused_backward(ws, dtok, B, drel)
        else:
            # every layer's output gradients land in the stacked workspace (dy of w2 = the gradient arriving at the
            # layer); only the dX chain runs here, the parameter gradients follow in _deferred_dw
            dtok = self._ln_bwd(tok, dfin, self.c + "norm.g", stable=1, out=ws["w2"]["dy"][self.depth - 1])
            for li, ly, (tk, n1, qkv, ao, o1, tokm, n2, hff, sw) in zip(reversed(range(self.depth)), reversed(self.layers),
                                                                       reversed(saved)):
                a, f = ly["a"], ly["f"]
                dsw = ly["w2"].bwd_dx(dtok)
                dhff = ws["w1"]["dy"][li]
                L.check(so.avi_swiglu_bwd(hff.data_ptr(), dsw.data_ptr(), R, 512, dhff.data_ptr(), st()), "swiglu_bwd")
                dn2 = ly["w1"].bwd_dx(dhff)
                dtokm = self._ln_bwd(tokm, dn2, f + "0.g", dx_add=dtok)
                do1 = self._ln_bwd(o1, dtokm, a + "to_out.1.g", out=ws["out"]["dy"][li])
                dao = ly["out"].bwd_dx(do1)
                dqkv = ws["qkv"]["dy"][li]
                L.check(so.avi_prior_attn_bwd(qkv.data_ptr(), S.ptr(a + "null_kv"), rel_bias.data_ptr(),
                                              self.rot_cos.data_ptr(), self.rot_sin.data_ptr(), dao.data_ptr(), B,
                                              dqkv.data_ptr(), S.gptr(a + "null_kv"), drel.data_ptr(), st()), "attn_bwd")
                dn1 = ly["qkv"].bwd_dx(dqkv)
                dtok = self._ln_bwd(tk, dn1, a + "norm.g", dx_add=dtokm, out=ws["w2"]["dy"][li - 1] if li > 0 else None)
        dtext = torch.empty((B, DIM), dtype=torch.float32, device=dev)
        dtemb = torch.empty((B, DIM), dtype=torch.float32, device=dev)
        L.check(so.avi_prior_tokens_bwd(dtok.data_ptr(), L.ptr(bk), L.ptr(ik), B, dtext.data_ptr(), dtemb.data_ptr(),
                                        S.gptr("net.null_brain_embeds"), S.gptr("net.null_image_embed"),
                                        S.gptr("net.learned_query"), st()), "tokens_bwd")

        # ---- BrainNetwork backward.  Order = what the data-parallel step wants (results do not depend on it: every
        # gradient element is still produced once, by the same launch on the same operands): first the dX chain down to the
        # four residual blocks - their 4096 x 4096 matrices are 97 % of the gradient bytes, each block's bucket starts its
        # all-reduce the moment its dW is written - and only then the parameter gradients that are leaves of backward
        # (projector, lin1, the prior's layers, the time MLP), which run under those all-reduces.
        dz2 = self.proj[2].bwd_dx(dproj)
        dz2p = self._ln_bwd(z2p, dz2, v + "projector.6.weight", v + "projector.6.bias", ops.ACT_GELU)
        dz1 = self.proj[1].bwd_dx(dz2p)
        dz1p = self._ln_bwd(z1p, dz1, v + "projector.3.weight", v + "projector.3.bias", ops.ACT_GELU)
        dz0 = self.proj[0].bwd_dx(dz1p)
        dout = self._ln_bwd(out, dz0, v + "projector.0.weight", v + "projector.0.bias", ops.ACT_GELU, dx_add=dtext)
        dh = self.lin1.bwd_dx(dout)
        for b in reversed(range(self.n_blocks)):
            dyp = self._ln_bwd(yp[b], dh, v + f"mlp.{b}.1.weight", v + f"mlp.{b}.1.bias", ops.ACT_GELU, dm[b + 1])
            dh = self.mlp[b].bwd(h[b], dyp, dx_residual=dh)
            self._grads_ready(v + f"mlp.{b}.0.weight", v + f"mlp.{b}.1.weight")
        dh0p = self._ln_bwd(h0p, dh, v + "lin0.1.weight", v + "lin0.1.bias", ops.ACT_GELU, dm[0])
        self.lin0.bwd(x, dh0p)
        self._grads_ready(v + "lin0.0.weight", v + "lin0.1.weight")
        self.proj[2].bwd_dw(z2, dproj)
        self.proj[1].bwd_dw(z1, dz2p)
        self.proj[0].bwd_dw(z0, dz1p)
        self.lin1.bwd_dw(h[-1], dout)
        self._grads_ready(v + "lin1.weight", v + "projector.8.weight")

        # ---- leaves of the prior's backward: the 24 layer matrices (batched launches), the T5 bias table, the time MLP
        self._deferred_dw(ws, R)
        # scatter the (8,3,4) bias gradient back onto the (32,8) T5 bucket table
        L.check(so.avi_prior_rel_bias(None, None, drel.data_ptr(), S.gptr(rel_name), 8, 3, st()), "rel_bias_bwd")
        da2 = self.tm[2].bwd(a2, dtemb)
        da2p = torch.empty_like(a2p)
        L.check(so.avi_act_bwd(a2p.data_ptr(), da2.data_ptr(), a2p.numel(), ops.ACT_SILU, da2p.data_ptr(), st()), "act_bwd")
        da1 = self.tm[1].bwd(a1, da2p)
        da1p = torch.empty_like(a1p)
        L.check(so.avi_act_bwd(a1p.data_ptr(), da1.data_ptr(), a1p.numel(), ops.ACT_SILU, da1p.data_ptr(), st()), "act_bwd")
        self.tm[0].bwd(te0, da1p)
        self._grads_ready("net.to_time_embeds.0.1.net.0.0.weight", self.c + "project_out.weight")
        return {"loss_prior": losses[0:1], "loss_nce": losses[1:2], "pred": pred, "proj": proj, "clip_voxels": out}

    # ------------------------------------------------------------------ optimizer
    def allreduce_grads(self):
        """Finish the gradient exchange of a step whose optimizer runs over the WHOLE buffer on this rank afterwards
        (``optimizer_step``): the single-GPU path and the unsharded data-parallel schedule (AVI_DP_SHARD=0).  The weight
        buckets were launched from inside backward (``_grads_ready``); the small no-decay tail (biases, T5 table) goes
        last.  AdamW then scales by 1/world.  With ranks and the sharded schedule use ``dp_update`` instead: there a rank
        holds the gradient sum of its own slices only."""
        if self.sync.shard and self.sync._collectives():
            raise RuntimeError("sharded data-parallel schedule: call dp_update(), not allreduce_grads() + optimizer_step()")
        return self.sync.finish(self.store.G)

    def _repack_span(self, a, b):
        """bf16 hi / lo planes of parameters [a, b) that another rank updated and the all-gather just delivered."""
        S = self.store
        L.check(L.load().avi_pack_weight_split(S.P.data_ptr() + 4 * a, 1, b - a, 1, S.HI.data_ptr() + 2 * a,
                                               S.LO.data_ptr() + 2 * a, L.stream_ptr()), "avi_pack_weight_split")

    def dp_update(self, lr=None, beta1=None, use_dyn=False, count=True):
        """The data-parallel optimizer step after ``forward_backward``: bucket by bucket - as its gradient exchange completes
        - fused AdamW on what this rank owns (GradSync: the whole bucket when unsharded, its 1/world slice + the bucket's
        few-element tail when sharded), the all-gather of the updated parameters started at once, then the planes of the
        slices other ranks own and the transposed planes.  Works without ranks too (every bucket owned here)."""
        S = self.store
        if count:
            self.step_count += 1
            if use_dyn:
                self._set_dyn(self.lr if lr is None else lr, beta1)
        world = self.sync.world()
        self.sync.finish(S.G, on_span=lambda a, b: self._adamw_span(a, b, world, lr=lr, beta1=beta1, use_dyn=use_dyn),
                         P=S.P, on_gathered=self._repack_span)
        self.refresh()
        return world

    def _set_dyn(self, lr, beta1=None):
        """Step-dependent AdamW scalars go through device memory so a captured graph can be replayed: the rate, the two
        bias corrections and this step's beta1 (OneCycleLR cycles it, host/schedule.py; torch forms the first bias
        correction from the CURRENT beta1: 1 - beta1 ** step)."""
        b1, b2 = self.betas
        b1 = b1 if beta1 is None else float(beta1)
        # the bias corrections exactly as avi_adamw forms them from its arguments (csrc/train.hip): in double, from the
        # betas ROUNDED TO fp32 (what the kernel multiplies with) - a replayed graph and an argument-driven step then agree
        # bit for bit
        f32 = lambda x: torch.tensor(x, dtype=torch.float32).item()
        b1, b2 = f32(b1), f32(b2)
        host = torch.tensor([lr, 1 - b1 ** self.step_count, 1 / math.sqrt(1 - b2 ** self.step_count), b1])
        self.dyn.copy_(host, non_blocking=True)

    def optimizer_step(self, lr=None, world=1, use_dyn=False, _in_graph=False, beta1=None):
        S = self.store
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        b1 = b1 if beta1 is None else float(beta1)
        if not _in_graph:
            self.step_count += 1
            if use_dyn:
                self._set_dyn(lr, b1)
        dyn = self.dyn.data_ptr() if use_dyn else 0
        so = L.load()
        for lo, hi_, wd in ((0, S.n_decay, self.wd), (S.n_decay, S.numel, 0.0)):
            if hi_ > lo:
                L.check(so.avi_adamw(S.P.data_ptr() + 4 * lo, S.G.data_ptr() + 4 * lo, S.M.data_ptr() + 4 * lo,
                                     S.V.data_ptr() + 4 * lo, hi_ - lo, lr, b1, b2, self.eps, wd, max(self.step_count, 1),
                                     1.0 / world, dyn, S.HI.data_ptr() + 2 * lo, S.LO.data_ptr() + 2 * lo,
                                     L.stream_ptr()), "adamw")
        self.refresh()

    def _adamw_span(self, a, b, world, lr=None, beta1=None, use_dyn=False):
        """Fused AdamW over [a, b) of the flat buffers (one gradient bucket): decay iff the span lies in the decay region."""
        S = self.store
        b1, b2 = self.betas
        b1 = b1 if beta1 is None else float(beta1)
        wd = self.wd if b <= S.n_decay else 0.0
        if a < S.n_decay < b:
            raise ValueError("a gradient span must not straddle the decay / no-decay boundary")
        L.check(L.load().avi_adamw(S.P.data_ptr() + 4 * a, S.G.data_ptr() + 4 * a, S.M.data_ptr() + 4 * a,
                                   S.V.data_ptr() + 4 * a, b - a, self.lr if lr is None else lr, b1, b2, self.eps, wd,
                                   max(self.step_count, 1), 1.0 / world, self.dyn.data_ptr() if use_dyn else 0,
                                   S.HI.data_ptr() + 2 * a, S.LO.data_ptr() + 2 * a, L.stream_ptr()), "adamw")

    # ------------------------------------------------------------------ data-parallel step as hipGraph segments
    def capture_step_dp(self, voxel, clip_target, temp, rand=None, warmup=2, rng=None):
        """The data-parallel step (train_diffusion_prior.py:338,442,450 are the reference's dead ``distributed`` branches)
        as a chain of hipGraph SEGMENTS: forward + backward are cut at the gradient-bucket announcements (``_grads_ready``:
        four aligner blocks, lin0, lin1 + projector, the prior network), the bucket's RCCL all-reduce is issued eagerly
        between two segments - collectives stay outside the graphs - and runs on RCCL's stream beside the next segment;
        after the last segment every bucket is waited for in turn and updated by its own fused-AdamW launch, so the
        optimizer works on the first buckets while the last ones are still on the wire.  ~140 launches per step become 7
        graph replays + 8 all-reduces + 9 launches.  Works with any world size (world 1 = the same chain without
        collectives; AVI_DP_FORCE_COLLECTIVES=1 issues them over one rank: the rehearsal a one-GPU box allows)."""
        if rand is None:
            if rng is None:
                raise ValueError("capture_step_dp needs rand (recorded draws) or rng (in-graph draws)")
            rand = self.draw(voxel.shape[0])
        if warmup < 1:
            raise ValueError("capture_step_dp needs at least one eager warm-up step (workspaces and job tables are built on "
                             "first use, which a stream capture does not allow)")
        self._static = dict(voxel=voxel.clone(), target=clip_target.clone(),
                            rand={k: ([m.clone() for m in v] if isinstance(v, list) else v.clone())
                                  for k, v in rand.items()})
        st = self._static

        def fb():
            r = st["rand"]
            if rng is not None:
                self.draw_device(rng, r)
            return self.forward_backward(st["voxel"], st["target"], r["times"], r["noise"], temp, r["brain_keep"],
                                         r["image_keep"], r["dropout_masks"])

        for _ in range(warmup):                              # eager steps (with collectives when there are ranks)
            fb()
            self.dp_update(use_dyn=True)
        torch.cuda.synchronize(self.device)
        self._segs = []
        pool = torch.cuda.graph_pool_handle()
        stream = torch.cuda.Stream(device=self.device)
        cur = {"g": None}
        n_spans = len(self.sync.expected)

        def begin():
            cur["g"] = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread may query events while a segment is being captured
            cur["g"].capture_begin(pool=pool, capture_error_mode="thread_local")

        def cut(first, last):
            cur["g"].capture_end()
            self._segs.append((cur["g"], (first, last)))
            self.sync.ready(self.store.G, first, last, launch=False)        # order / coverage checks still run
            if len(self._segs) < n_spans:
                begin()

        stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream):
            self._cut = cut
            try:
                begin()
                self._gout = fb()
            except BaseException:
                # leave nothing half open: end the capture this thread is in (its graph is thrown away), forget the segments
                # and the announcements made so far - the caller may fall back to the eager step on this object
                try:
                    cur["g"].capture_end()
                except Exception:
                    pass
                self._segs, self.sync.done, self.sync.works = [], [], []
                raise
            finally:
                self._cut = None
        torch.cuda.current_stream(self.device).wait_stream(stream)
        if len(self._segs) != n_spans:
            raise RuntimeError("the backward pass announced fewer gradient spans than the layout expects")
        self.sync.finish(self.store.G, launch=False)
        self._dp_stream = stream                  # the chain replays on the stream it was captured on (replay_step_dp)
        torch.cuda.synchronize(self.device)
        return self

    def replay_step_dp(self, lr=None, beta1=None):
        self.step_count += 1
        cur = torch.cuda.current_stream(self.device)
        s = self._dp_stream
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            # An explicit stream for the whole chain: the collectives order themselves behind "the work already on the current
            # stream" through an event recorded when they are issued, and a hipGraph launched on the NULL stream is not
            # reliably ordered in front of such an event (world-2 run on one GPU: the buckets announced between two segments
            # were reduced before their segment had finished writing them - wrong sums, sometimes NaN; the bucket behind the
            # last segment was always right.  tests/test_gpu_dp_world2.py).
            self._set_dyn(self.lr if lr is None else lr, beta1)
            G = self.store.G
            for graph, (first, last) in self._segs:
                graph.replay()
                self.sync.ready(G, first, last)                   # eager all-reduce of the bucket this segment completed
            self.dp_update(use_dyn=True, count=False)
        cur.wait_stream(s)
        return self._gout

    def train_step(self, voxel, clip_target, temp, rand=None, lr=None, beta1=None):
        """``lr`` / ``beta1``: this step's rate and AdamW beta1 (``OneCycleLR.lr_at`` / ``momentum_at``); None = the
        constructor's."""
        rand = rand or self.draw(voxel.shape[0])
        out = self.forward_backward(voxel, clip_target, rand["times"], rand["noise"], temp, rand["brain_keep"],
                                    rand["image_keep"], rand["dropout_masks"])
        if self.sync._collectives():          # ranks: per-bucket exchange -> update (sharded by default) -> gather
            self.dp_update(lr, beta1)
        else:
            self.optimizer_step(lr, self.allreduce_grads(), beta1=beta1)
        return out

    # ------------------------------------------------------------------ hipGraph capture (single GPU)
    def capture_step(self, voxel, clip_target, temp, rand=None, warmup=2, rng=None):
        """Capture forward + backward + AdamW (no collective) into one hipGraph over static input buffers.  ``rng`` (a
        host/rng.DeviceRng): the step's random inputs are drawn INSIDE the graph (fresh at every replay); otherwise the
        tensors of ``rand`` are frozen into it (parity runs)."""
        if rand is None:
            if rng is None:
                raise ValueError("capture_step needs rand (recorded draws) or rng (in-graph draws)")
            rand = self.draw(voxel.shape[0])                  # shapes and dtypes of the static buffers
        self._static = dict(voxel=voxel.clone(), target=clip_target.clone(),
                            rand={k: ([m.clone() for m in v] if isinstance(v, list) else v.clone())
                                  for k, v in rand.items()})
        st = self._static

        def body(in_graph):
            r = st["rand"]
            if rng is not None:
                self.draw_device(rng, r)
            out = self.forward_backward(st["voxel"], st["target"], r["times"], r["noise"], temp, r["brain_keep"],
                                        r["image_keep"], r["dropout_masks"])
            if self.allreduce_grads() != 1:
                raise RuntimeError("capture_step is the single-GPU path: collectives stay outside the graph")
            self.optimizer_step(use_dyn=True, _in_graph=in_graph)
            return out

        for _ in range(warmup):
            body(False)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._gout = body(True)
        return self

    def replay_step(self, lr=None, beta1=None):
        self.step_count += 1
        self._set_dyn(self.lr if lr is None else lr, beta1)
        self._graph.replay()
        return self._gout
