"""Multi-GPU sharding of the receive path: one process per GPU, receivers partitioned across ranks, every rank
reads the same I/Q stream.  The path has no exchange step; the only collective is the gather of the outputs
(audio blocks, graph row) to one rank.  On GPUs that gather lives inside libqkengine (qk_comm_init / qk_gather_audio /
qk_gather_graph: RCCL from C, include/qk_engine.h); this module keeps the partition arithmetic, the harness's max-over-ranks
and a torch.distributed form of the same gather layout that the CPU tests run under gloo (tests/test_shard_gloo.py)."""
import torch
import torch.distributed as dist


def receiver_slice(total_rx, world, rank):
    """Contiguous block of receivers owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(total_rx, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_audio(local, counts, dst=0):
    """local: [n_local, stride, 2] float64 tensor (interleaved stereo), counts: [n_local] int64 tensor.
    Returns (audio [total_rx, stride, 2], counts [total_rx]) on rank dst, (None, None) elsewhere.
    Ranks may own different numbers of receivers: rows are padded to the largest shard for the collective."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local, counts
    rank = dist.get_rank()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    cpad = torch.zeros(nmax, dtype=torch.int64, device=local.device)
    cpad[:local.shape[0]] = counts
    recv = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    crecv = [torch.empty_like(cpad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, recv, dst=dst)
    dist.gather(cpad, crecv, dst=dst)
    if rank != dst:
        return None, None
    return (torch.cat([recv[r][:sizes[r]] for r in range(world)], 0),
            torch.cat([crecv[r][:sizes[r]] for r in range(world)], 0))


def max_over_ranks(seconds, device):
    """Elapsed time of the slowest rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
