"""Data-parallel sharding of the sampling path: utterances are independent, so ranks take
disjoint subsets with NO data-path collective; only the small (T,53) outputs are gathered.

One process per GPU under ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).
"""
import torch


def partition_by_length(lengths, world_size):
    """Longest-first greedy bin packing of utterance indices onto ranks, balancing total frames.
    Deterministic (ties broken by index), identical on every rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    bins = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(b) for b in bins]


def pad_to_multiple(t, mult=8):
    """EMOTE pads T up to a multiple of the FLINT latent frame size (FaceFormerDecoder.py:1112-1125)."""
    return (int(t) + mult - 1) // mult * mult


def gather_outputs(local, n_total, dist=None):
    """``local``: {utterance index: (T_i, C) tensor}.  Returns the list of all n_total outputs on every rank.
    Uses all_gather_object on CPU copies: outputs are <= 53*T*4 bytes each, off the hot path."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local[i] for i in range(n_total)]
    payload = {i: t.detach().cpu() for i, t in local.items()}
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, payload)
    merged = {}
    for p in parts:
        merged.update(p)
    if sorted(merged) != list(range(n_total)):
        raise RuntimeError("utterance shards do not cover the batch exactly once")
    return [merged[i] for i in range(n_total)]


def max_over_ranks(seconds, device, dist=None):
    """bench.py timing rule: the step time of the job is the slowest rank's."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
