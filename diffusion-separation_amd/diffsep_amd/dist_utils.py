"""Utterance sharding across ranks (one process per GPU) and the result gather.

Partitioning follows the reference's own (evaluate_mp.py:495-503): contiguous index ranges of
floor(n / workers) items, the last range takes the remainder.  There is no exchange step inside the
sampler, so the only collective is the gather of per-utterance results to rank 0 — RCCL over xGMI on
the GPU box (backend "nccl"), gloo in the CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n_items, world, rank):
    per = n_items // world
    start = rank * per
    end = n_items if rank == world - 1 else start + per
    return start, end


def rank_indices(n_items, world, rank, lengths=None, balance=False):
    """Utterance indices of one rank, ascending: the reference's contiguous range (evaluate_mp.py:495-503), or with
    balance=True and known lengths the length-balanced deal."""
    if balance and lengths and world > 1:
        return sorted(length_balanced_order(lengths, world)[rank])
    lo, hi = shard_range(n_items, world, rank)
    return list(range(lo, hi))


def length_balanced_order(lengths, world):
    """Optional better balance for variable-length sets: sort by length, deal round-robin.
    Returns one index list per rank."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    return [order[r::world] for r in range(world)]


def gather_waveforms(local, lengths, device=None):
    """local: list of [S, T_i] float32 tensors of this rank; returns on rank 0 the list over all ranks
    (rank-major order), None elsewhere.  Tensors are padded to the global max length and moved with one
    all_gather of lengths + one gather of the padded block."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return list(local)
    rank = dist.get_rank()
    dev = device if device is not None else (local[0].device if local else torch.device("cpu"))
    n_loc = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_loc) for _ in range(world)]
    dist.all_gather(counts, n_loc)
    counts = [int(c.item()) for c in counts]
    S = local[0].shape[0] if local else 1
    tmax_loc = torch.tensor([max([int(x.shape[-1]) for x in local], default=0), S], dtype=torch.int64, device=dev)
    tm = [torch.zeros_like(tmax_loc) for _ in range(world)]
    dist.all_gather(tm, tmax_loc)
    tmax, S = max(int(t[0]) for t in tm), max(int(t[1]) for t in tm)
    nmax = max(counts)
    block = torch.zeros((nmax, S, tmax), dtype=torch.float32, device=dev)
    lens = torch.zeros((nmax,), dtype=torch.int64, device=dev)
    for i, x in enumerate(local):
        block[i, :, : x.shape[-1]] = x.to(dev)
        lens[i] = x.shape[-1]
    blocks = [torch.empty_like(block) for _ in range(world)] if rank == 0 else None
    lens_all = [torch.empty_like(lens) for _ in range(world)] if rank == 0 else None
    dist.gather(block, blocks, dst=0)
    dist.gather(lens, lens_all, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        for i in range(counts[r]):
            out.append(blocks[r][i, :, : int(lens_all[r][i])].clone())
    return out


def gather_objects(obj):
    """Small python results (metric dicts) to rank 0."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [obj]
    out = [None] * world if dist.get_rank() == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out
