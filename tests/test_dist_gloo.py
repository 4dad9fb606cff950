"""CPU, world_size 2, gloo: the N>1 sampling path shards utterances with no data-path collective and gathers
outputs; bench.py's max-over-ranks timing rule.  The per-rank compute stand-in is the CPU oracle head."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd.host.sharding import gather_outputs, max_over_ranks, partition_by_length
    from avi_talking_amd.weights import make_emote_weights
    from oracle import emote as OE
    torch.set_num_threads(1)
    w = make_emote_weights(1)
    mine = partition_by_length(lengths, world)[rank]
    local = {}
    for i in mine:
        g = torch.Generator().manual_seed(100 + i)
        feat = torch.randn(1, lengths[i], 768, generator=g)
        style = torch.randn(1, 1, 128, generator=g)
        o = OE.forward(w, feat, style)
        local[i] = torch.cat([o["predicted_exp"], o["predicted_jaw"]], -1)[0]
    outs = gather_outputs(local, len(lengths), dist)
    tmax = max_over_ranks(1.0 + rank, torch.device("cpu"), dist)
    if rank == 0:
        ret["outs"] = [o.clone() for o in outs]
        ret["tmax"] = tmax
        ret["shards"] = partition_by_length(lengths, world)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampling_matches_single_process():
    lengths = [24, 9, 40, 16, 8]
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), lengths, ret), nprocs=world, join=True)
    from avi_talking_amd.weights import make_emote_weights
    from oracle import emote as OE
    w = make_emote_weights(1)
    assert ret["tmax"] == 2.0                       # slowest rank
    shards = ret["shards"]
    assert sorted(i for s in shards for i in s) == list(range(len(lengths))) and all(shards)
    for i, T in enumerate(lengths):
        g = torch.Generator().manual_seed(100 + i)
        feat = torch.randn(1, T, 768, generator=g)
        style = torch.randn(1, 1, 128, generator=g)
        o = OE.forward(w, feat, style)
        ref = torch.cat([o["predicted_exp"], o["predicted_jaw"]], -1)[0]
        assert ret["outs"][i].shape == (T, 53)
        assert torch.allclose(ret["outs"][i], ref, atol=1e-6)


def _ar_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd.host.training import bucketed_allreduce
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    spans = [(600, 1000), (100, 600), (0, 100)]            # backward-completion order, covering [0, n) once
    works = bucketed_allreduce(flat, spans, async_op=True)
    for w in works:
        w.wait()
    if rank == 0:
        ret["flat"] = flat.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce():
    """The DP gradient sum of the training step (C1): every span of the flat gradient buffer is reduced once."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ar_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    expect = torch.arange(1000, dtype=torch.float32) * 3     # rank 0: x1, rank 1: x2
    assert torch.equal(ret["flat"], expect)


# ----------------------------------------------------------------------------- training step: gradient spans
def test_grad_spans_cover_flat_buffer_once():
    """The spans PriorTrainer announces from inside backward (training.grad_spans, checked call by call by GradSync)
    plus the no-decay tail cover [0, numel) of the shipped flat layout exactly once, without overlap."""
    from avi_talking_amd.host.training import FlatLayout, _layout, grad_spans, no_decay
    from avi_talking_amd.weights import make_prior_weights
    w = make_prior_weights(3)
    lay = FlatLayout.of_state_dict(w, _layout())
    spans = sorted(lay.span(a, b) for a, b in grad_spans()) + [(lay.n_decay, lay.numel)]
    pos = 0
    for a, b in spans:
        assert a == pos and b > a, (a, b, pos)
        pos = b
    assert pos == lay.numel
    # the decay region holds exactly the names the reference's substring rule decays (train_diffusion_prior.py:997-1003)
    for n in lay.names:
        assert (lay.offset[n] < lay.n_decay) == (not no_decay(n)), n
    assert lay.numel >= sum(v.numel() for k, v in w.items() if v.is_floating_point() and not k.startswith("noise_scheduler"))


def test_gradsync_rejects_unannounced_and_misordered_spans():
    import pytest
    from avi_talking_amd.host.training import FlatLayout, GradSync, _layout, grad_spans
    from avi_talking_amd.weights import make_prior_weights
    lay = FlatLayout.of_state_dict(make_prior_weights(3), _layout())
    G = torch.zeros(lay.numel)
    sync = GradSync(lay)
    spans = grad_spans()
    with pytest.raises(RuntimeError, match="out of order"):
        sync.ready(G, *spans[1])
    sync = GradSync(lay)
    sync.ready(G, *spans[0])
    with pytest.raises(RuntimeError, match="never announced"):
        sync.finish(G)


def _flat_grads(lay, w, half, seed=77, B=8):
    """Oracle autograd gradients of the reference training loss (oracle/prior.py train_loss) on one half of a seeded
    batch, packed into the trainer's flat layout."""
    from oracle import prior as OP
    g = torch.Generator().manual_seed(seed)
    voxel, target = torch.randn(2 * B, 768, generator=g), torch.randn(2 * B, 1, 128, generator=g) * 0.3
    times, noise = torch.randint(0, 100, (2 * B,), generator=g), torch.randn(2 * B, 1, 128, generator=g)
    sl = slice(half * B, (half + 1) * B)
    params = {k: v.clone().requires_grad_(True) for k, v in w.items() if k in lay.offset}
    loss = OP.train_loss(params, voxel[sl], target[sl], times[sl], noise[sl], 0.005)[0]
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    G = torch.zeros(lay.numel)
    for (k, p), gr in zip(params.items(), grads):
        if gr is not None:
            G[lay.offset[k]:lay.offset[k] + p.numel()] = gr.reshape(-1)
    return G


def _trainer_sync_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd.host.training import FlatLayout, GradSync, _layout, grad_spans
    from avi_talking_amd.weights import make_prior_weights
    torch.set_num_threads(2)
    w = make_prior_weights(3)
    lay = FlatLayout.of_state_dict(w, _layout())
    G = _flat_grads(lay, w, rank)
    sync = GradSync(lay)
    for a, b in grad_spans():               # the order PriorTrainer.forward_backward announces them in
        sync.ready(G, a, b)
    world_out = sync.finish(G)
    G /= world_out                           # avi_adamw's 1/world gradient scale
    if rank == 0:
        ret["G"] = G.clone()
        ret["world"] = world_out
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_gradient_sync_world2_matches_single_process():
    """World size 2 over gloo: each rank's oracle gradients on its half batch, reduced through the trainer's own span
    bookkeeping (GradSync + the shipped layout), equal the single-process mean of the two half-batch gradients -
    every element of the flat buffer (a span reduced twice or never would show)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    from avi_talking_amd.host.training import FlatLayout, _layout
    from avi_talking_amd.weights import make_prior_weights
    w = make_prior_weights(3)
    lay = FlatLayout.of_state_dict(w, _layout())
    expect = (_flat_grads(lay, w, 0) + _flat_grads(lay, w, 1)) / 2
    assert ret["world"] == 2
    assert expect.abs().max() > 0
    # the workers run 2 threads, this process more: fp32 summation order differs in the last bits
    err = (ret["G"] - expect).abs().max().item()
    print(f"max |sync - single process| = {err:.2e} (gradient scale {expect.abs().max().item():.2e})")
    assert err < 2e-5 * expect.abs().max().item()


def _span_update_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd.host.training import FlatLayout, GradSync, _layout, grad_spans
    from avi_talking_amd.weights import make_prior_weights
    lay = FlatLayout.of_state_dict(make_prior_weights(3), _layout())
    G = torch.full((lay.numel,), float(rank + 1))
    P = torch.zeros(lay.numel)
    sync = GradSync(lay, shard=False)        # the unsharded schedule: every rank updates every bucket
    for a, b in grad_spans():
        sync.ready(G, a, b)
    seen = []

    def update(a, b):                        # what PriorTrainer.dp_update does per bucket: its own optimizer launch
        assert bool((G[a:b] == 3.0).all()), "a bucket was handed to the optimizer before its sum had arrived"
        P[a:b] += G[a:b] / world
        seen.append((a, b))
    sync.finish(G, on_span=update)
    if rank == 0:
        ret["seen"], ret["P"] = seen, P.clone()
        ret["expected"] = [lay.span(a, b) for a, b in grad_spans()] + [(lay.n_decay, lay.numel)]
    dist.barrier()
    dist.destroy_process_group()


def test_per_bucket_optimizer_callback_world2():
    """GradSync.finish(on_span=...): every span of the flat buffer - the announced buckets in announcement order, then the
    no-decay tail - is handed to the optimizer exactly once and only after ITS all-reduce has completed (world 2, gloo)."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_span_update_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["seen"] == ret["expected"]
    covered = torch.zeros_like(ret["P"])
    for a, b in ret["seen"]:
        covered[a:b] += 1
    assert bool((covered == 1).all())
    assert bool((ret["P"] == 1.5).all())


def test_shard_of_partitions_every_bucket():
    """The sharded optimizer's slices: for every bucket of the shipped layout and several world sizes the ranks' slices
    and the bucket's tail tile [a, b) exactly, slices are equal and 32-byte aligned, the tail is shorter than world * 8."""
    from avi_talking_amd.host.training import FlatLayout, SHARD_ALIGN, _layout, grad_spans, shard_of
    from avi_talking_amd.weights import make_prior_weights
    lay = FlatLayout.of_state_dict(make_prior_weights(3), _layout())
    buckets = [lay.span(a, b) for a, b in grad_spans()] + [(lay.n_decay, lay.numel)]
    for world in (1, 2, 3, 8):
        for a, b in buckets:
            pos = a
            for r in range(world):
                lo, hi, main = shard_of(a, b, world, r)
                assert lo == pos and (hi - lo) % SHARD_ALIGN == 0 and lo % SHARD_ALIGN == 0
                assert hi - lo == (main - a) // world
                pos = hi
            assert pos == main and 0 <= b - main < world * SHARD_ALIGN
    assert shard_of(0, 100, 1, 0) == (0, 96, 96)


def _sharded_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from avi_talking_amd.host.training import FlatLayout, GradSync, _layout, grad_spans
    from avi_talking_amd.weights import make_prior_weights
    torch.set_num_threads(2)
    w = make_prior_weights(3)
    lay = FlatLayout.of_state_dict(w, _layout())
    G = _flat_grads(lay, w, rank)
    P = torch.zeros(lay.numel)
    for n in lay.names:
        P[lay.offset[n]:lay.offset[n] + w[n].numel()] = w[n].reshape(-1).float()
    sync = GradSync(lay)                     # default: sharded
    assert sync.shard
    for a, b in grad_spans():
        sync.ready(G, a, b)
    updated, arrived = torch.zeros(lay.numel), torch.zeros(lay.numel)

    def update(a, b):                        # stand-in for the fused AdamW launch on [a, b): plain SGD on the mean gradient
        P[a:b] -= 0.1 * G[a:b] / world
        updated[a:b] += 1

    def gathered(a, b):                      # ranges another rank updated: the trainer rebuilds their bf16 planes
        arrived[a:b] += 1
    assert sync.finish(G, on_span=update, P=P, on_gathered=gathered) == world
    ret[f"P{rank}"], ret[f"updated{rank}"], ret[f"arrived{rank}"] = P.clone(), updated, arrived
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_world2_matches_single_process():
    """ZeRO-1 schedule over gloo, world 2: reduce-scatter per bucket (gloo: the slice of an all-reduce), every rank updates
    ITS slice of every bucket (+ the bucket's few-element tail), all-gather of the updated parameters.  Both ranks end with
    the parameters a single process gets from the mean gradient - every element -; every element is updated by exactly
    one rank (the tails by both), and what a rank did not update itself is reported as arrived."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_sharded_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    from avi_talking_amd.host.training import FlatLayout, _layout
    from avi_talking_amd.weights import make_prior_weights
    w = make_prior_weights(3)
    lay = FlatLayout.of_state_dict(w, _layout())
    P0 = torch.zeros(lay.numel)
    for n in lay.names:
        P0[lay.offset[n]:lay.offset[n] + w[n].numel()] = w[n].reshape(-1).float()
    expect = P0 - 0.1 * (_flat_grads(lay, w, 0) + _flat_grads(lay, w, 1)) / 2
    assert torch.equal(ret["P0"], ret["P1"]), "the ranks ended the step with different parameters"
    scale = (expect - P0).abs().max().item()
    err = (ret["P0"] - expect).abs().max().item()
    print(f"sharded world-2 step vs single process: {err:.2e} (the step moved the parameters by {scale:.2e})")
    assert scale > 0 and err < 2e-5 * scale
    total = ret["updated0"] + ret["updated1"]
    assert bool(((total == 1) | (total == 2)).all()) and int((total == 2).sum()) < 8 * 2 * 8     # tails only
    for r in range(world):
        assert bool(((ret[f"updated{r}"] + ret[f"arrived{r}"]) == 1).all())      # own or arrived, never both, never neither


# ----------------------------------------------------------------------------- bench.py launcher
def test_bench_gpus_flag_spawns_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher in the environment starts 2 ranks itself (CPU/gloo rehearsal): one
    JSON line from rank 0 with n_gpus = 2, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True and rec["grad_spans_reduced_once"] is True


def test_bench_launcher_reports_failing_rank(tmp_path):
    """A rank that dies makes the launcher kill the rest and exit non-zero (never a silent success)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    helper = tmp_path / "failing_rank.py"
    helper.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(120)\n")
    code = ("import sys; sys.path.insert(0, %r); sys.argv=['bench.py']; import bench; "
            "sys.exit(bench.launch_ranks(2, [], 60, script=%r))" % (root, str(helper)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr[-1000:])
