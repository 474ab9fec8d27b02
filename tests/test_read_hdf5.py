"""The HDF5 side of the opacity / scattering / star readers (reference source/read.py:1041-1103, :1195-1236, :1598-1645).

The system interpreter has no h5py; the readers then go through the HDF5 C library itself (helios_amd/hdf5_lite.py,
ctypes).  Pinned here against REAL files: tests/golden/reader/hdf5/*.h5 were written by h5py 3.3.0 / HDF5 1.10.6 and
read by the REFERENCE's own `read.py` (tests/golden/make_hdf5_golden.py, run under the image's conda Python 3.9 which
has h5py and astropy); `expected.npz` is what the reference's methods left in `quant`.  The product's methods read the
same files here and have to leave the same arrays, bit for bit.  Also: the h5py branch through a stand-in module (so
that its lines run where h5py is absent), the error text when neither is there, and hdf5_lite's writer against
`h5dump` of the same HDF5 distribution."""
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
    (tmp_path / "t.h5").write_bytes(b"")          # the stand-in serves t.npz's datasets under this name
    got = _read(tmp_path / "t.h5")
    _same(got, _read(tmp_path / "t.npz"))
    assert got[1].opac_meanmass[0] == 2.3 * pc.AMU and got[1].ny == 20


def test_hdf5_file_without_h5py_and_without_libhdf5_says_what_to_do(tmp_path, monkeypatch):
    from helios_amd import hdf5_lite
    monkeypatch.setitem(sys.modules, "h5py", None)          # import h5py -> ImportError
    monkeypatch.setattr(hdf5_lite, "_lib", None)
    monkeypatch.setattr(hdf5_lite, "_lib_error", "none here")
    (tmp_path / "t.h5").write_bytes(b"")
    with pytest.raises(IOError, match="neither h5py nor an HDF5 library"):
        _read(tmp_path / "t.h5")
    with pytest.raises(IOError, match="no such file"):
        _read(tmp_path / "absent.h5")


def test_hdf5_branch_with_the_real_library(tmp_path):
    h5py = pytest.importorskip("h5py")
    d = _table(seed=5)
    np.savez(tmp_path / "t.npz", **d)
    with h5py.File(tmp_path / "t.h5", "w") as f:
        for k, v in d.items():
            f.create_dataset(k, data=v)
    _same(_read(tmp_path / "t.h5"), _read(tmp_path / "t.npz"))


# ---- real files, the reference's reader as the oracle ---------------------------------------------------------------
import json
import os
import shutil
import subprocess

from helios_amd import hdf5_lite
from helios_amd.read import Species

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reader", "hdf5")
GRID = ("opac_wave", "nbin", "gauss_y", "ny", "opac_interwave", "opac_deltawave", "ktemp", "ntemp", "kpress", "npress")
needs_hdf5 = pytest.mark.skipif(not hdf5_lite.available(), reason="no libhdf5 on this host")


def _expected(tag):
    with np.load(os.path.join(H5, "expected.npz")) as z:
        return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(tag + "/")}


def _check(tag, got):
    exp = _expected(tag)
    assert exp
    for k, v in exp.items():
        g = np.asarray(got[k])
        assert g.shape == v.shape, (tag, k, g.shape, v.shape)
        np.testing.assert_array_equal(g, v, err_msg="%s/%s" % (tag, k))


def _grid(q):
    return {n: getattr(q, n) for n in GRID if hasattr(q, n)}


@pytest.fixture(params=["libhdf5", "h5py"])
def no_h5py(request, monkeypatch):
    """the readers' two ways to an HDF5 file: the C library through ctypes (what the system interpreter has), and h5py
    where it is installed (`/opt/conda/bin/python3.9 -m pytest tests/test_read_hdf5.py` in the build image)"""
    if request.param == "h5py":
        pytest.importorskip("h5py")
    else:
        monkeypatch.setitem(sys.modules, "h5py", None)
    return request.param


def test_constants_are_astropys():
    """source/phys_const.py:27-44 under astropy 4.3.1, as the generator recorded them"""
    meta = json.load(open(os.path.join(H5, "expected_constants.json")))
    assert meta["astropy"] == "4.3.1" and meta["hdf5"].startswith("1.10")
    for name in ("C", "K_B", "H", "R_UNIV", "N_A", "SIGMA_SB", "AU", "AMU", "R_SUN", "R_JUP", "R_EARTH", "G"):
        assert getattr(pc, name) == meta["constants"][name], name


@needs_hdf5
def test_fixture_files_are_hdf5_and_list_their_datasets(no_h5py):
    for name in os.listdir(H5):
        if name.endswith(".h5"):
            assert open(os.path.join(H5, name), "rb").read(8) == b"\x89HDF\r\n\x1a\n", name
    with hdf5_lite.File(os.path.join(H5, "premixed_opac_kdistr.h5")) as f:
        assert sorted(f.keys()) == sorted(["pressures", "temperatures", "meanmolmass", "kpoints", "weighted Rayleigh cross-sections",
                                           "included molecules", "wavelengths", "FastChem path", "units", "center wavelengths",
                                           "interface wavelengths", "wavelength width of bins", "ypoints"])
        assert [s.decode() for s in f["included molecules"][()]] == ["H2O", "CO2", "CIA_H2-H2"]     # variable-length strings
        assert f["units"][()] == b"CGS" and f["FastChem path"][()] == b"/some/where/fastchem/"     # scalar strings
        assert "nothing" not in f and "kpoints" in f and "kpoints/deeper" not in f
        with pytest.raises(KeyError):
            f["nothing"]
    with hdf5_lite.File(os.path.join(H5, "star.h5")) as f:
        assert sorted(f.keys()) == ["original", "r20_kdistr", "r50_kdistr"] and sorted(f.keys("/r50_kdistr")) == ["blackbody", "lambda", "phoenix"]
        assert "/r50_kdistr/phoenix/gj1214" in f and "r50_kdistr/phoenix" in f and "r50_kdistr/phoenix/gj9999" not in f
        assert f.is_dataset("r50_kdistr/lambda") and not f.is_dataset("r50_kdistr/phoenix")
        with pytest.raises(KeyError, match="group"):
            f["r50_kdistr"]


@needs_hdf5
def test_premixed_table_as_the_reference_reads_it(no_h5py):
    r, q = Read(), Q()
    k = r.read_opac_file(q, os.path.join(H5, "premixed_opac_kdistr.h5"), type="premixed")
    _check("premixed", dict(_grid(q), opac_scat_cross=q.opac_scat_cross, opac_meanmass=q.opac_meanmass, opac_k=k))


@needs_hdf5
def test_sampling_layout_two_dimensional_table_and_a_compressed_dataset(no_h5py):
    r, q = Read(), Q()
    k = r.read_opac_file(q, os.path.join(H5, "DDD_opac_ip_sampling.h5"), type="species", read_grid_parameters=True)
    _check("sampling", dict(_grid(q), opac_k=k))        # `opacities`, `wavelengths`, no y-points: interfaces and widths derived
    r, q = Read(), Q()
    k = r.read_opac_file(q, os.path.join(H5, "EEE_opac_ip_kdistr.h5"), type="species", read_grid_parameters=True)
    _check("twodim", dict(_grid(q), opac_k=k))
    r, q = Read(), Q()
    k = r.read_opac_file(q, os.path.join(H5, "BBB_opac_ip.h5"), type="species", read_grid_parameters=False)   # gzip + shuffle, chunked
    assert not hasattr(q, "nbin")
    _check("species_no_grid", dict(opac_k=k))


@needs_hdf5
def test_species_loop_file_name_fallbacks_and_scattering_cross_sections(no_h5py):
    r, q = Read(), Q()
    r.opacity_path = H5 + os.sep
    q.fl_prec = np.float64
    q.species_list = []
    for name, absorbing, scattering in (("AAA", "yes", "yes"), ("BBB", "yes", "no"), ("CCC", "yes", "yes"), ("H2O", "no", "yes")):
        q.species_list.append(Species(name=name, absorbing=absorbing, scattering=scattering))
    r.read_species_opacities(q)
    q.nlayer, q.ninterface, q.iso = np.int32(3), np.int32(4), np.int32(0)
    r.read_species_scat_cross_sections(q)
    got = _grid(q)
    for s in q.species_list:
        if s.absorbing == "yes":
            got["opacity_pretab_" + s.name] = s.opacity_pretab
        if s.scattering == "yes" and s.name != "H2O":
            got["scat_cross_sect_pretab_" + s.name] = s.scat_cross_sect_pretab
            got["scat_cross_sect_layer_" + s.name] = s.scat_cross_sect_layer
            got["scat_cross_sect_interface_" + s.name] = s.scat_cross_sect_interface
    _check("species", got)
    q.species_list.append(Species(name="ZZZ", absorbing="yes", scattering="no"))
    with pytest.raises(IOError):
        r.read_species_opacities(q)


@needs_hdf5
def test_stellar_spectrum_by_nested_path(no_h5py):
    for tag, data_set in (("star", "/r50_kdistr/phoenix/gj1214"), ("star_noslash", "r50_kdistr/blackbody/gj1214")):
        r, q = Read(), Q()
        r.stellar_model, r.stellar_path, r.stellar_data_set = "file", os.path.join(H5, "star.h5"), data_set
        q.nbin, q.fl_prec = np.int32(6), np.float64
        r.read_star(q)
        _check(tag, dict(starflux=q.starflux, real_star=q.real_star))
    r, q = Read(), Q()
    r.stellar_model, r.stellar_path, r.stellar_data_set = "file", os.path.join(H5, "star.h5"), "/r20_kdistr/phoenix/gj1214"
    q.nbin, q.fl_prec = np.int32(6), np.float64
    assert str(_expected("star_wrong_length")["raised"]) == "OverflowError"
    with pytest.raises(OverflowError):
        r.read_star(q)
    r.stellar_data_set = "/r50_kdistr/phoenix/nosuchstar"      # the reference asks on the terminal; a batch run refuses
    with pytest.raises(IOError, match="no such stellar spectrum"):
        r.read_star(q)
    r.stellar_data_set = "/r50_kdistr/phoenix"                 # a group
    with pytest.raises(IOError, match="no such stellar spectrum"):
        r.read_star(q)


@needs_hdf5
def test_storage_types_other_tools_choose(no_h5py):
    """single precision, big-endian doubles and integers, 32-bit and unsigned 8-bit integers, a chunked + compressed 2-D
    single-precision table, all written by h5py: the readers hand out fp64 / int64 with the values h5py itself returns"""
    from helios_amd.read import _Table
    t = _Table(os.path.join(H5, "dtypes.h5"))
    exp = _expected("dtypes")
    assert sorted(exp) == ["f4", "f8_be", "i4", "i8_be", "table_f4", "u1"] and sorted(t.keys()) == sorted(exp)
    for k, v in exp.items():
        got = np.asarray(t[k])
        assert got.shape == v.shape, k
        np.testing.assert_array_equal(got.astype(v.dtype), v, err_msg=k)
    if no_h5py == "libhdf5":
        with hdf5_lite.File(os.path.join(H5, "dtypes.h5")) as f:
            assert f["f4"][()].dtype == np.float64 and f["table_f4"][()].dtype == np.float64 and f["table_f4"].shape == (4, 6)
            assert f["i4"][()].dtype == np.int64 and f["u1"][()].dtype == np.int64 and f["i8_be"][()].dtype == np.int64
            assert abs(int(f["i8_be"][()][0])) > 2 ** 31 or abs(int(f["i8_be"][()][1])) > 2 ** 31 or abs(int(f["i8_be"][()][2])) > 2 ** 31


@needs_hdf5
def test_writer_round_trip_and_h5dump(tmp_path, no_h5py):
    rng = np.random.default_rng(8)
    d = {"kpoints": rng.random(240), "grid/deep/er": rng.random((3, 4)), "count": np.arange(5), "scalar": np.float64(2.5),
         "names": np.array(["H2O", "CO2"]), "empty": np.zeros(0)}
    for level in (None, 4):
        path = tmp_path / ("w%s.h5" % level)
        hdf5_lite.write(path, d, compression=level)
        with hdf5_lite.File(path) as f:
            assert sorted(f.keys()) == sorted(["kpoints", "grid", "count", "scalar", "names", "empty"])
            np.testing.assert_array_equal(f["kpoints"][()], d["kpoints"])
            np.testing.assert_array_equal(f["grid/deep/er"][:], d["grid/deep/er"])
            assert f["count"][()].dtype == np.int64 and list(f["count"][()]) == [0, 1, 2, 3, 4]
            assert f["scalar"][()] == 2.5 and f["scalar"][()].shape == ()
            assert list(f["names"][()]) == [b"H2O", b"CO2"] and f["empty"][()].shape == (0,)
    # an .npz table turned into .h5 reads the same through the product's reader
    t = _table(seed=11)
    np.savez(tmp_path / "t.npz", **t)
    hdf5_lite.write(tmp_path / "t.h5", t, compression=6)
    _same(_read(tmp_path / "t.h5"), _read(tmp_path / "t.npz"))
    # the distribution's own tool reads what the writer wrote
    tool = shutil.which("h5dump") or os.path.join(os.path.dirname(os.path.dirname(hdf5_lite.library()[0])), "bin", "h5dump")
    if not os.path.exists(tool):
        pytest.skip("no h5dump next to the library")
    out = subprocess.run([tool, "-d", "/grid/deep/er", "-m", "%.17g", str(tmp_path / "w4.h5")], capture_output=True, text=True)
    assert out.returncode == 0 and "H5T_IEEE_F64LE" in out.stdout and "( 3, 4 )" in out.stdout
    vals = [float(x) for line in out.stdout.splitlines() if line.strip().startswith("(") for x in line.split(":")[1].split(",") if x.strip()]
    np.testing.assert_array_equal(np.array(vals).reshape(3, 4), d["grid/deep/er"])
