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
