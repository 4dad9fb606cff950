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
