"""Multi-GPU layer: independent atmosphere columns are partitioned over the ranks (one process per GPU) -- in
contiguous blocks (`shard_columns`) or dealt out in turn (`column_list(..., "cyclic")`: neighbours in a parameter sweep
need similar numbers of iterations, so dealing them out evens the load without any exchange) or claimed chunk by chunk
from a shared work list as a rank retires the columns it holds (`WorkList`); the iteration path has NO collective.
The only exchange is one gather of the output spectra at the end of a run, done with torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm,
"gloo" is used by the CPU tests.  The reference has no counterpart (single process, single device).
"""
import os

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


def _cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in the order the HIP runtime enumerates them: the KFD topology's nodes that have SIMDs
    (`/sys/class/kfd/kfd/topology/nodes/<n>/properties`: simd_count > 0), in node order; their PCI address (domain,
    location_id = bus << 8 | devfn) leads to `/sys/bus/pci/devices/<bdf>/numa_node`.  Where the KFD topology cannot be read
    the DRM cards (`/sys/class/drm/card<k>/device/numa_node`, AMD vendor id, ascending PCI address) stand in.  [] when
    neither is there (containers without sysfs); -1 = the platform does not say."""
    nodes = []
    top = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        for n in sorted(os.listdir(top), key=int):
            props = {}
            with open(os.path.join(top, n, "properties")) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) == 0:
                continue
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
            try:
                with open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")) as f:
                    nodes.append(int(f.read()))
            except (OSError, ValueError):
                nodes.append(-1)
        if nodes:
            return nodes
    except (OSError, ValueError):
        nodes = []
    drm = os.path.join(sysfs, "class/drm")
    try:
        cards = []
        for c in os.listdir(drm):
            if not (c.startswith("card") and c[4:].isdigit()):
                continue
            dev = os.path.join(drm, c, "device")
            try:
                with open(os.path.join(dev, "vendor")) as f:
                    if int(f.read(), 16) != 0x1002:
                        continue
                with open(os.path.join(dev, "numa_node")) as f:
                    numa = int(f.read())
            except (OSError, ValueError):
                continue
            cards.append((os.path.basename(os.path.realpath(dev)), numa))
        return [numa for _, numa in sorted(cards)]
    except OSError:
        return []


def visible_device_index(local_index, env=None):
    """the physical GPU behind HIP device `local_index` when ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES hold plain index
    lists (ROCR's is applied first, HIP's indexes into what is left); None when a list is given by UUID"""
    env = os.environ if env is None else env
    idx = int(local_index)
    hip_var = "HIP_VISIBLE_DEVICES" if env.get("HIP_VISIBLE_DEVICES") else "CUDA_VISIBLE_DEVICES"   # (HIP reads either)
    for var in (hip_var, "ROCR_VISIBLE_DEVICES"):
        v = env.get(var)
        if not v:
            continue
        items = [x.strip() for x in v.split(",") if x.strip()]
        if not all(x.isdigit() for x in items):
            return None
        if idx >= len(items):
            return None
        idx = int(items[idx])
    return idx


def bind_to_gpu_numa_node(local_index, sysfs="/sys", env=None, apply=True):
    """Bind this process to the host cores of the NUMA node its GPU hangs on -- `os.sched_setaffinity`, before the first
    GPU call, so that the runtime's helper threads and the pinned staging buffers start there; no numactl, no re-exec.
    The mask is the node's cpulist intersected with the cores the process may use already (cgroups, an outer taskset); an
    empty intersection, an unknown node or a sysfs that is not there leave the process as it is.  Returns what happened,
    for the bench line: {"gpu", "numa_node", "cpus", "bound", "why"}."""
    out = {"gpu": None, "numa_node": None, "cpus": None, "bound": False, "why": None}
    if not hasattr(os, "sched_getaffinity"):
        out["why"] = "no sched_getaffinity on this platform"
        return out
    allowed = set(os.sched_getaffinity(0))
    out["cpus"] = len(allowed)
    gpu = visible_device_index(local_index, env)
    if gpu is None:
        out["why"] = "visible-devices list is not a plain index list"
        return out
    out["gpu"] = gpu
    nodes = gpu_numa_nodes(sysfs)
    if gpu >= len(nodes):
        out["why"] = "no sysfs entry for GPU %d (%d found)" % (gpu, len(nodes))
        return out
    numa = nodes[gpu]
    out["numa_node"] = numa
    if numa < 0:
        out["why"] = "platform reports no NUMA node for the GPU"
        return out
    try:
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % numa)) as f:
            cpus = _cpulist(f.read())
    except (OSError, ValueError):
        out["why"] = "node%d has no cpulist" % numa
        return out
    mask = cpus & allowed
    if not mask:
        out["why"] = "none of node%d's cores is usable by this process" % numa
        return out
    if apply and mask != allowed:
        os.sched_setaffinity(0, mask)
    out["cpus"], out["bound"] = len(mask), bool(apply)
    if not apply:
        out["why"] = "not applied (single rank: the CPU baseline keeps every core)"
    return out


class _stdout_to_stderr(object):
    """file descriptor 1 points at file descriptor 2 inside the block (what C++ libraries print goes along), if `active`"""

    def __init__(self, active=True):
        self.active, self.saved = active, None

    def __enter__(self):
        if self.active:
            import sys
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            import sys
            sys.stdout.flush()
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def init_process_group_checked(dist, backend, rank, world, device=None, timeout_s=180.0, log=None):
    """`init_process_group` with a deadline, and a roll call before the first real collective: a gloo side group's
    `monitored_barrier` names the ranks that did not arrive (RCCL itself only hangs or aborts), then one all-reduce on
    the real backend confirms that every rank's communicator works.  A failure is reported with this rank's number and
    device on stderr before it is raised again.  Returns {"init_s", "roll_call_s", "first_collective_s"}."""
    import datetime
    import sys
    import time
    import torch
    log = log or (lambda m: (sys.stderr.write(m + "\n"), sys.stderr.flush()))
    who = "rank %d/%d (device %s, pid %d)" % (rank, world, device, os.getpid())
    to = datetime.timedelta(seconds=float(timeout_s))
    t0 = time.perf_counter()
    try:
        if backend == "nccl":
            dist.init_process_group("nccl", timeout=to, device_id=device)
        else:
            dist.init_process_group(backend, timeout=to)
    except Exception as e:
        log("helios_amd.parallel: %s: init_process_group(%r) failed within %.0f s: %s: %s"
            % (who, backend, timeout_s, type(e).__name__, e))
        raise
    t1 = time.perf_counter()
    # The roll call is a diagnostic, never a reason to fail: its own (shorter) deadline, the loopback interface when the
    # rendezvous is local (a host whose name does not resolve may refuse gloo any other device), and every failure in it is
    # reported and passed over -- the first collective on the real backend below is the test that counts.
    side, roll_call = None, True
    roll_to = datetime.timedelta(seconds=min(float(timeout_s), 60.0))
    with _stdout_to_stderr(backend != "gloo"):     # (gloo announces its connections on stdout: not in front of a bench line)
        if backend != "gloo":
            if os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                side = dist.new_group(backend="gloo", timeout=roll_to)
            except Exception as e:
                log("helios_amd.parallel: %s: no gloo side group for the roll call (%s: %s); going on without"
                    % (who, type(e).__name__, e))
                roll_call = False
        if roll_call:
            try:
                dist.monitored_barrier(group=side, timeout=roll_to, wait_all_ranks=True)
            except Exception as e:
                log("helios_amd.parallel: %s: roll call incomplete -- %s: %s" % (who, type(e).__name__, e))
    t2 = time.perf_counter()
    try:
        t = torch.ones(1, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        if backend == "nccl":
            torch.cuda.synchronize()
        seen = int(t.item())
        if seen != world:
            raise RuntimeError("first all-reduce counted %d ranks, expected %d" % (seen, world))
    except Exception as e:
        log("helios_amd.parallel: %s: first %s collective failed -- %s: %s" % (who, backend, type(e).__name__, e))
        raise
    t3 = time.perf_counter()
    return {"init_s": t1 - t0, "roll_call_s": t2 - t1, "first_collective_s": t3 - t2}
