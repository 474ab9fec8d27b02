"""Multi-process path on CPU: world_size-2 gloo run of the column sharding + spectra gather."""
import os
import socket
import sys

import numpy as np
import pytest

from helios_amd.parallel import shard_columns


def test_shard_columns_covers_everything():
    for ncol, world in ((512, 8), (10, 4), (3, 8), (7, 2)):
        seen = []
        for r in range(world):
            a, b = shard_columns(ncol, r, world)
            seen += list(range(a, b))
        assert seen == list(range(ncol))
    assert shard_columns(512, 3, 8) == (192, 256)


def test_column_lists():
    from helios_amd.parallel import column_list
    for mode in ("block", "cyclic"):
        for ncol, world in ((512, 8), (10, 4), (7, 2)):
            seen = sorted(c for r in range(world) for c in column_list(ncol, r, world, mode))
            assert seen == list(range(ncol))
    assert column_list(10, 1, 4, "cyclic") == [1, 5, 9] and column_list(10, 1, 4, "block") == [3, 4, 5]


def _worker(rank, world, port, ncol, q, mode="block"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from helios_amd.parallel import column_list, gather_spectra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cols = column_list(ncol, rank, world, mode)
    local = np.array([[c * 100.0 + k for k in range(5)] for c in cols]).reshape(len(cols), 5)
    full = gather_spectra(local, dist, columns=cols if mode == "cyclic" else None)
    q.put((rank, full))
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol,mode", [(6, "block"), (7, "block"), (7, "cyclic"), (5, "cyclic")])
def test_gather_spectra_gloo_world2(ncol, mode):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ncol, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    want = np.array([[c * 100.0 + k for k in range(5)] for c in range(ncol)])
    for _rank, full in res:
        np.testing.assert_array_equal(full, want)


def test_expand_sweep():
    from helios_amd.sweep import expand_sweep
    cols = expand_sweep("internal_temperature=100,300; f_factor=0.25,0.5,1")
    assert len(cols) == 6 and cols[0] == {"internal_temperature": "100", "f_factor": "0.25"}
    assert cols[-1] == {"internal_temperature": "300", "f_factor": "1"}
    assert expand_sweep("") == [{}]
    import pytest
    with pytest.raises(ValueError):
        expand_sweep("number_of_layers=10,20")        # changes the batch itself, cannot vary inside it


def test_work_list_without_a_process_group():
    from helios_amd.parallel import WorkList
    w = WorkList(7, 3)
    got = list(iter(w.claim, []))
    assert got == [[0, 1, 2], [3, 4, 5], [6]] and w.claimed == list(range(7))


def _worklist_worker(rank, world, port, ncol, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    import torch.distributed as dist
    from helios_amd.parallel import WorkList, gather_spectra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    work = WorkList(ncol, chunk, dist)
    for cols in iter(work.claim, []):
        time.sleep(0.02 * (1 + 4 * rank) * len(cols))       # rank 1 is five times slower per column
    cols = work.claimed
    local = np.array([[c * 100.0 + k for k in range(5)] for c in cols]).reshape(len(cols), 5)
    full = gather_spectra(local, dist, columns=cols)
    # a second sweep in the same process group draws from its own counter (its own key), not from the exhausted one
    again = WorkList(ncol, chunk, dist, key=WorkList.KEY + "/sweep2")
    second = [c for cols2 in iter(again.claim, []) for c in cols2]
    stale = WorkList(ncol, chunk, dist)          # the first key once more: nothing left to claim
    every = [None, None]
    dist.all_gather_object(every, (second, stale.claim()))
    assert sorted(every[0][0] + every[1][0]) == list(range(ncol)) and every[0][1] == every[1][1] == []
    q.put((rank, cols, full))
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol,chunk", [(23, 2), (3, 8)])
def test_work_list_gloo_world2(ncol, chunk):
    """two ranks claim chunks from the shared list until it is empty: every column exactly once, the faster rank retires
    more of them, and the spectra still come back in sweep order (also when one rank got nothing)"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worklist_worker, args=(r, 2, port, ncol, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in procs:
        rank, cols, full = q.get(timeout=120)
        res[rank] = (cols, full)
    for p in procs:
        p.join(timeout=60)
    assert sorted(res[0][0] + res[1][0]) == list(range(ncol))
    if ncol > chunk:
        assert len(res[0][0]) > len(res[1][0]) > 0
    want = np.array([[c * 100.0 + k for k in range(5)] for c in range(ncol)])
    for rank in res:
        np.testing.assert_array_equal(res[rank][1], want)


def test_sweep_batches_are_keyed_on_the_chemistry():
    """columns of one device batch share the species list and the SOURCE of every species' mixing ratio (the device keeps one
    (T, P) table per FastChem species and column: which species those are must agree); the FastChem directory itself may vary
    from column to column (`-sweep "directory_with_fastchem_files=..."`)"""
    from helios_amd.sweep import _batch_signature, expand_sweep

    class Sp(object):
        def __init__(self, name, source):
            self.name, self.source_for_vmr = name, source

    class Q(object):
        nbin, ny, nlayer, scat, dir_beam, clouds, scat_corr, smooth, convection = 30, 20, 10, 1, 0, 0, 0, 0, 0
        opacity_mixing, g_0, epsi, planet_type, iso, singlewalk, flux_calc_method = "on-the-fly", 0.0, 0.5, "gas", 0, 0, "iteration"

    a, b, c = Q(), Q(), Q()
    a.species_list = [Sp("H2O", "FastChem"), Sp("CO2", "file")]
    b.species_list = [Sp("H2O", "FastChem"), Sp("CO2", "file")]
    c.species_list = [Sp("H2O", "1e-3"), Sp("CO2", "file")]
    assert _batch_signature(a) == _batch_signature(b) != _batch_signature(c)
    cols = expand_sweep("directory_with_fastchem_files=chem/m0/,chem/m1/;internal_temperature=100,300")
    assert len(cols) == 4 and cols[1] == {"directory_with_fastchem_files": "chem/m0/", "internal_temperature": "300"}


def _fake_sysfs(root, gpus, nodes):
    """a sysfs tree with the KFD topology of `gpus` = [(pci bus, numa node)] behind two CPU nodes, and `nodes` = {n: cpulist}"""
    top = os.path.join(root, "class/kfd/kfd/topology/nodes")
    k = 0
    for n in sorted(nodes):           # the CPU agents come first in the KFD topology, without SIMDs
        os.makedirs(os.path.join(top, str(k)))
        with open(os.path.join(top, str(k), "properties"), "w") as f:
            f.write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        os.makedirs(os.path.join(root, "devices/system/node/node%d" % n))
        with open(os.path.join(root, "devices/system/node/node%d/cpulist" % n), "w") as f:
            f.write(nodes[n] + "\n")
        k += 1
    for bus, numa in gpus:
        os.makedirs(os.path.join(top, str(k)))
        with open(os.path.join(top, str(k), "properties"), "w") as f:
            f.write("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        dev = os.path.join(root, "bus/pci/devices/0000:%02x:00.0" % bus)
        os.makedirs(dev)
        with open(os.path.join(dev, "numa_node"), "w") as f:
            f.write("%d\n" % numa)
        k += 1


def test_rank_binds_to_the_cores_of_its_gpus_numa_node(tmp_path):
    """bench.py binds every rank of a multi-GPU run to the host cores of its GPU's NUMA node before the first GPU call:
    KFD topology order = HIP device order, PCI address -> numa_node -> cpulist, intersected with the cores the process may
    use; a sysfs that is not there (this container), an unknown node or a UUID-style visible-devices list leave the process
    alone and say why"""
    from helios_amd import parallel as par
    before = set(os.sched_getaffinity(0))
    try:        # (an OpenMP runtime loaded by an earlier test may have pinned this thread to one core: ask for all again)
        os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        os.sched_setaffinity(0, before)
        pytest.skip("one usable core: nothing to bind to")
    lo, hi = allowed[: len(allowed) // 2], allowed[len(allowed) // 2:]
    as_list = lambda cpus: ",".join(str(c) for c in cpus) + ",4090-4095"      # (cores this process may not use are ignored)
    root = str(tmp_path)
    _fake_sysfs(root, [(0x05, 0), (0x15, 0), (0x65, 1), (0x75, 1), (0x85, -1)], {0: as_list(lo), 1: as_list(hi)})
    assert par.gpu_numa_nodes(root) == [0, 0, 1, 1, -1]
    assert par._cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    try:
        r = par.bind_to_gpu_numa_node(2, sysfs=root, env={})
        assert r["bound"] and r["gpu"] == 2 and r["numa_node"] == 1 and r["cpus"] == len(hi)
        assert sorted(os.sched_getaffinity(0)) == hi
        os.sched_setaffinity(0, allowed)
        # HIP's visible-devices list indexes into ROCR's: local device 0 -> HIP list [3, 0] -> 3 -> ROCR list [4, 2, 1, 3] -> GPU 3
        r = par.bind_to_gpu_numa_node(0, sysfs=root, env={"HIP_VISIBLE_DEVICES": "3,0", "ROCR_VISIBLE_DEVICES": "4,2,1,3"}, apply=False)
        assert r["gpu"] == 3 and r["numa_node"] == 1 and not r["bound"] and sorted(os.sched_getaffinity(0)) == allowed
        r = par.bind_to_gpu_numa_node(4, sysfs=root, env={})
        assert not r["bound"] and r["numa_node"] == -1 and "no NUMA node" in r["why"]
        r = par.bind_to_gpu_numa_node(0, sysfs=root, env={"ROCR_VISIBLE_DEVICES": "GPU-deadbeef"})
        assert not r["bound"] and "index list" in r["why"]
        r = par.bind_to_gpu_numa_node(7, sysfs=root, env={})
        assert not r["bound"] and "no sysfs entry" in r["why"]
        r = par.bind_to_gpu_numa_node(0, sysfs=os.path.join(root, "nothing_here"), env={})
        assert not r["bound"] and r["gpu"] == 0 and sorted(os.sched_getaffinity(0)) == allowed
    finally:
        os.sched_setaffinity(0, before)


def _init_worker(rank, world, port, q, late):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time
    import torch.distributed as dist
    from helios_amd.parallel import init_process_group_checked
    if late and rank == 1:
        time.sleep(1.0)
    msgs = []
    try:
        t = init_process_group_checked(dist, "gloo", rank, world, device="cpu", timeout_s=60.0, log=msgs.append)
        q.put((rank, "ok", t))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error", (type(e).__name__, msgs)))


def test_checked_process_group_init_gloo_world2():
    """the set-up bench.py uses for its ranks: init with a deadline, a roll call (monitored barrier), one all-reduce that must
    count every rank -- here with gloo and one rank arriving a second late"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_init_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == ["ok", "ok"], res
    for _rank, _ok, t in res:
        assert set(t) == {"init_s", "roll_call_s", "first_collective_s"} and all(v >= 0 for v in t.values())
