"""Multi-GPU layer: independent atmosphere columns are partitioned over the ranks (one process per GPU) -- in
contiguous blocks (`shard_columns`) or dealt out in turn (`column_list(..., "cyclic")`: neighbours in a parameter sweep
need similar numbers of iterations, so dealing them out evens the load without any exchange) or claimed chunk by chunk
from a shared work list as a rank retires the columns it holds (`WorkList`); the iteration path has NO collective.
The only exchange is one gather of the output spectra at the end of a run, done with torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm,
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


class WorkList:
    """The columns of a sweep as a shared work list: ranks claim the next `chunk` columns when they have retired the
    ones they hold, so a rank whose columns converge early takes over work from the others.  A claim
    is one atomic add on the process group's key-value store (the TCP store torch.distributed.run already provides) --
    a host-side counter, not a collective: no rank ever waits for another.  Without a process group the list is local."""

    KEY = "helios_amd/worklist/next"

    def __init__(self, ncol_total, chunk, dist=None, store=None, key=None):
        self.ncol = int(ncol_total)
        self.chunk = max(1, int(chunk))
        self.key = key or self.KEY
        self._local_next = 0
        self.store = store
        if self.store is None and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            from torch.distributed import distributed_c10d as c10d
            try:
                self.store = c10d._get_default_store()
            except Exception as e:      # a torch without that accessor: pass `store=` (e.g. a TCPStore) explicitly
                raise RuntimeError("WorkList needs the process group's key-value store; pass store=...") from e
        self.claimed = []

    def claim(self):
        """the next columns of this rank (ascending, at most `chunk`), or [] when the list is exhausted"""
        if self.store is not None:
            start = int(self.store.add(self.key, self.chunk)) - self.chunk
        else:
            start = self._local_next
            self._local_next += self.chunk
        cols = list(range(start, min(start + self.chunk, self.ncol)))
        self.claimed += cols
        return cols


class _DeviceView(object):
    """a strided view of device memory owned by the library, for torch.as_tensor (the CUDA array interface)"""

    def __init__(self, ptr, shape, strides_bytes):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 3,
                                         "strides": tuple(int(v) for v in strides_bytes)}


def emission_spectra_on_device(rt, ncol=None):
    """the top-of-atmosphere upward band fluxes of the batch's columns as ONE torch tensor on the GPU, [ncol, nbin],
    copied out of the library's band array on the device (internal layout [column][bin][interface]: the spectrum is the
    last interface of every bin) -- no host round trip between the iteration and the gather.  The caller has synchronised
    the library's stream."""
    import torch
    ncol = rt.ncol if ncol is None else int(ncol)
    X, I = rt.nbin, rt.ninterface
    base = rt.device_ptr("F_up_band_n", 0).value
    view = _DeviceView(base + 8 * (I - 1), (ncol, X), (8 * X * I, 8 * I))
    return torch.as_tensor(view, device="cuda").contiguous()


def gather_spectra(local, dist=None, device=None, columns=None):
    """all-gather per-column output vectors: local [ncol_local, n] -> [ncol_total, n] on every rank (a numpy array).
    `local` is a numpy array or a torch tensor; a tensor that already lives on the collective's device (`device="cuda"`
    for RCCL) is gathered from there -- padded, exchanged and trimmed on the GPU, one copy to the host at the end.
    Column counts may differ between ranks (padding to the maximum, then trimming).  `columns`: the global indices of
    the local rows (default: ranks hold consecutive blocks in rank order); the result is in global column order."""
    is_tensor = not isinstance(local, np.ndarray) and hasattr(local, "detach")
    if dist is None or not dist.is_initialized():
        return local.detach().cpu().numpy() if is_tensor else np.ascontiguousarray(local, dtype=np.float64)
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    t = local.detach() if is_tensor else torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    t = t.to(dev, torch.float64)
    shapes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(shapes, torch.tensor([t.shape[0], t.shape[1] if t.dim() == 2 else 0], dtype=torch.int64, device=dev))
    counts = [int(c[0].item()) for c in shapes]
    width = max(int(c[1].item()) for c in shapes)       # a rank of a dynamic sweep may have claimed nothing
    n = t.shape[0]
    nmax = max(counts)
    pad = torch.zeros((nmax, width + 1), dtype=torch.float64, device=dev)   # last entry of a row: its global column index
    if n:
        pad[:n, :-1] = t
        pad[:n, -1] = -1.0 if columns is None else torch.as_tensor(np.asarray(columns, np.float64), device=dev)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    rows = torch.cat([o[:k] for o, k in zip(out, counts)], dim=0).cpu().numpy()
    if columns is None:
        return rows[:, :-1]
    order = np.argsort(rows[:, -1].astype(np.int64), kind="stable")
    return rows[order, :-1]
