"""The HDF5 branch of the opacity / star readers (reference source/read.py:1041-1103, :1598-1645).  h5py is not part of the
build image, so the branch is exercised twice: with a stand-in `h5py` module that serves the datasets of an .npz through
h5py's File / dataset interface (`f.keys()`, `f[name][()]`, context manager) -- which runs every line of the branch --
and, where the real library is installed, with a real HDF5 file written by h5py."""
import sys
import types

import numpy as np
import pytest

from helios_amd import phys_const as pc
from helios_amd import synthetic as syn
from helios_amd.read import Read


class Q(object):
    pass


def _table(seed=3, nbin=7, ny=20, ntemp=4, npress=3):
    rng = np.random.default_rng(seed)
    inter, wave, dwave = syn.wavelength_grid(nbin)
    gy, _ = syn.gauss_points(ny)
    ktemp, kpress = syn.tp_grid(ntemp, npress)
    return {"kpoints": syn.ktable(rng, nbin, ny, ktemp, kpress, gy), "ypoints": gy, "center wavelengths": wave,
            "interface wavelengths": inter, "wavelength width of bins": dwave, "temperatures": ktemp, "pressures": kpress,
            "meanmolmass": np.full(ntemp * npress, 2.3), "weighted Rayleigh cross-sections": syn.rayleigh_table(wave, ntemp, npress)}


def _read(path):
    r, q = Read(), Q()
    k = r.read_opac_file(q, str(path), type="premixed")
    return k, q


def _same(a, b):
    (ka, qa), (kb, qb) = a, b
    np.testing.assert_array_equal(ka, kb)
    for n in ("opac_scat_cross", "opac_meanmass", "opac_wave", "opac_interwave", "opac_deltawave", "gauss_y", "ktemp", "kpress"):
        np.testing.assert_array_equal(getattr(qa, n), getattr(qb, n), err_msg=n)
    assert (qa.nbin, qa.ny, qa.ntemp, qa.npress) == (qb.nbin, qb.ny, qb.ntemp, qb.npress)


def test_hdf5_branch_with_a_stand_in_h5py(tmp_path, monkeypatch):
    d = _table()
    np.savez(tmp_path / "t.npz", **d)

    class _Dataset(object):
        def __init__(self, a):
            self.a = a

        def __getitem__(self, key):
            assert key == ()                       # the reader takes whole datasets: f[name][()]
            return self.a

    class _File(object):
        def __init__(self, path, mode="r"):
            assert mode == "r" and str(path).endswith(".h5")
            self.d = dict(np.load(str(path)[:-3] + ".npz"))

        def keys(self):
            return self.d.keys()

        def __getitem__(self, k):
            return _Dataset(self.d[k])

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    fake = types.ModuleType("h5py")
    fake.File = _File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    got = _read(tmp_path / "t.h5")
    _same(got, _read(tmp_path / "t.npz"))
    assert got[1].opac_meanmass[0] == 2.3 * pc.AMU and got[1].ny == 20


def test_hdf5_file_without_h5py_says_what_to_do(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "h5py", None)          # import h5py -> ImportError
    with pytest.raises(IOError, match="h5py is not installed"):
        _read(tmp_path / "t.h5")


def test_hdf5_branch_with_the_real_library(tmp_path):
    h5py = pytest.importorskip("h5py")
    d = _table(seed=5)
    np.savez(tmp_path / "t.npz", **d)
    with h5py.File(tmp_path / "t.h5", "w") as f:
        for k, v in d.items():
            f.create_dataset(k, data=v)
    _same(_read(tmp_path / "t.h5"), _read(tmp_path / "t.npz"))
