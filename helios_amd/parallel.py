"""Multi-GPU layer: independent atmosphere columns are partitioned over the ranks (one process per GPU) -- in
contiguous blocks (`shard_columns`) or dealt out in turn (`column_list(..., "cyclic")`: neighbours in a parameter sweep
need similar numbers of iterations, so dealing them out evens the load without any exchange); the iteration path has
NO collective.  The only exchange is one gather of the output spectra at the
end of a run, done with torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm,
"gloo" is used by the CPU tests.  The reference has no counterpart (single process, single device).
"""
import numpy as np


def shard_columns(ncol_total, rank, world):
    """block partition: columns [start, stop) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(int(ncol_total), int(world))
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return start, stop


def column_list(ncol_total, rank, world, mode="block"):
    """the columns of rank `rank`, ascending: a contiguous block, or every world-th column starting at `rank`"""
    if mode == "cyclic":
        return list(range(int(rank), int(ncol_total), int(world)))
    if mode != "block":
        raise ValueError("partition mode must be 'block' or 'cyclic'")
    a, b = shard_columns(ncol_total, rank, world)
    return list(range(a, b))


def gather_spectra(local, dist=None, device=None, columns=None):
    """all-gather per-column output vectors: local [ncol_local, n] -> [ncol_total, n] on every rank.
    Column counts may differ between ranks (padding to the maximum, then trimming).  `columns`: the global indices of
    the local rows (default: ranks hold consecutive blocks in rank order); the result is in global column order."""
    local = np.ascontiguousarray(local, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev))
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    pad = np.zeros((nmax, local.shape[1] + 1))          # last entry of a row: its global column index
    pad[:local.shape[0], :-1] = local
    pad[:local.shape[0], -1] = -1.0 if columns is None else np.asarray(columns, np.float64)
    mine = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    rows = np.concatenate([o.cpu().numpy()[:n] for o, n in zip(out, counts)], axis=0)
    if columns is None:
        return rows[:, :-1]
    order = np.argsort(rows[:, -1].astype(np.int64), kind="stable")
    return rows[order, :-1]
