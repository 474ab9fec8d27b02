"""Multi-GPU layer: independent atmosphere columns are block-partitioned over the ranks (one process per
GPU); the iteration path has NO collective.  The only exchange is one gather of the output spectra at the
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


def gather_spectra(local, dist=None, device=None):
    """all-gather per-column output vectors: local [ncol_local, n] -> [ncol_total, n] on every rank.
    Column counts may differ between ranks (padding to the maximum, then trimming)."""
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
    pad = np.zeros((nmax, local.shape[1]))
    pad[:local.shape[0]] = local
    mine = torch.from_numpy(pad).to(dev)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return np.concatenate([o.cpu().numpy()[:n] for o, n in zip(out, counts)], axis=0)
