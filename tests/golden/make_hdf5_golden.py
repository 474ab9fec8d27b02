"""Real HDF5 fixtures for the readers, made by the real `h5py`, and what the REFERENCE's own reader parses from them.

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py     # build container only; never runs on the GPU box

The image's system interpreter has no h5py and no astropy, but /opt/conda ships a Python 3.9 with h5py 3.3.0
(HDF5 1.10.6) and astropy 4.3.1.  Under that interpreter this script

  1. writes small opacity / scattering / stellar containers with h5py, dataset by dataset as the reference's k-table
     tool writes them (ktable/source_ktable/combination.py:294-307, :482-512; nested groups for the stellar file as
     the reference's `dataset in stellar spectrum file = /r50_kdistr/phoenix/...` addresses them), one of them gzip-
     compressed and chunked, into tests/golden/reader/hdf5/*.h5;
  2. imports the reference's `source/read.py` -- with the real h5py and the real astropy; only `pycuda.*`, which
     read.py reaches through host_functions.py and never calls here, is an empty module -- and lets ITS methods read
     the files: `read_opac_file` (premixed; species with and without grid parameters; sampling layout),
     `read_species_opacities` (the three file-name fall-backs), `read_species_scat_cross_sections`, `read_star`;
  3. stores what they left in `quant` as tests/golden/reader/hdf5/expected.npz, and astropy's constants as the
     reference's phys_const.py derives them in expected_constants.json.

tests/test_read_hdf5.py reads the same files with the product's reader on the system interpreter (no h5py: through
libhdf5 and ctypes) and compares.  Outputs are data only.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "reader", "hdf5")
REF = "/root/reference"
sys.dont_write_bytecode = True


def tables(rng):
    nbin, ny, ntemp, npress = 6, 20, 3, 4
    inter = 0.3e-4 * (500 / 0.3) ** (np.arange(nbin + 1) / nbin)
    wave = 0.5 * (inter[1:] + inter[:-1])
    dwave = np.diff(inter)
    gy = 0.5 * np.polynomial.legendre.leggauss(ny)[0] + 0.5
    ktemp = np.linspace(300.0, 1500.0, ntemp)
    kpress = 10.0 ** np.linspace(1.0, 7.0, npress)
    n = ny * nbin * npress * ntemp
    grid = dict(pressures=kpress, temperatures=ktemp)
    kd = {"interface wavelengths": inter, "center wavelengths": wave, "wavelength width of bins": dwave, "ypoints": gy}
    files = {}
    premixed = dict(grid)
    premixed["meanmolmass"] = 2.2 + 0.1 * rng.random(npress * ntemp)
    premixed["kpoints"] = 10.0 ** rng.uniform(-8, 2, n)
    premixed["weighted Rayleigh cross-sections"] = 10.0 ** rng.uniform(-28, -24, nbin * npress * ntemp)
    premixed["included molecules"] = ["H2O", "CO2", "CIA_H2-H2"]
    premixed["wavelengths"] = wave
    premixed["FastChem path"] = "/some/where/fastchem/"
    premixed["units"] = "CGS"
    premixed.update(kd)
    files["premixed_opac_kdistr.h5"] = premixed
    for name, stem in (("AAA", "_opac_ip_kdistr"), ("BBB", "_opac_ip"), ("CCC", "_opac_ip_sampling")):
        d = dict(grid)
        d.update(kd)
        d["kpoints"] = 10.0 ** rng.uniform(-10, 3, n)
        files[name + stem + ".h5"] = d
    nsamp = 9
    swave = np.sort(rng.uniform(1e-4, 2e-3, nsamp))
    files["DDD_opac_ip_sampling.h5"] = dict(grid, wavelengths=swave, opacities=10.0 ** rng.uniform(-10, 3, nsamp * npress * ntemp))
    # a 2-D k-table (some tools keep the axes): the readers flatten it
    files["EEE_opac_ip_kdistr.h5"] = dict(grid, **kd, kpoints=(10.0 ** rng.uniform(-10, 3, n)).reshape(ntemp * npress, nbin * ny))
    files["scat_cross_sections.h5"] = {"wavelengths": wave, "rayleigh_AAA": 10.0 ** rng.uniform(-28, -24, nbin),
                                       "rayleigh_CCC": 10.0 ** rng.uniform(-28, -24, nbin)}
    files["star.h5"] = {"r50_kdistr/phoenix/gj1214": 10.0 ** rng.uniform(2, 8, nbin),
                        "r50_kdistr/blackbody/gj1214": 10.0 ** rng.uniform(2, 8, nbin),
                        "r50_kdistr/lambda": wave,
                        "original/phoenix/gj1214/flux": 10.0 ** rng.uniform(2, 8, 31),
                        "original/phoenix/gj1214/lambda": np.linspace(1e-5, 1e-2, 31),
                        "r20_kdistr/phoenix/gj1214": 10.0 ** rng.uniform(2, 8, nbin + 2)}
    # storage types other tools choose: single precision, big-endian, 32-bit and unsigned 8-bit integers, a chunked 2-D
    # single-precision table -- h5py hands every one of them to the reference as numbers; the readers take fp64 / int64
    files["dtypes.h5"] = {"f4": rng.random(11).astype("<f4"), "f8_be": rng.random(5).astype(">f8"),
                          "i4": rng.integers(-9, 9, 7).astype("<i4"), "u1": rng.integers(0, 255, 6).astype("u1"),
                          "table_f4": rng.random((4, 6)).astype("<f4"), "i8_be": rng.integers(-2 ** 40, 2 ** 40, 3).astype(">i8")}
    return files


def write_files(files):
    import h5py
    os.makedirs(OUT, exist_ok=True)
    for fname, d in files.items():
        with h5py.File(os.path.join(OUT, fname), "w") as f:
            for k, v in d.items():
                if fname == "dtypes.h5" and k == "table_f4":
                    f.create_dataset(k, data=v, chunks=(2, 3), compression="gzip")
                elif fname == "BBB_opac_ip.h5" and k == "kpoints":
                    f.create_dataset(k, data=v, chunks=(97,), compression="gzip", compression_opts=4, shuffle=True)
                else:
                    f.create_dataset(k, data=v)      # nested names make the intermediate groups
    return h5py.version.version, h5py.version.hdf5_version


def import_reference_reader():
    for name in ("pycuda", "pycuda.driver", "pycuda.autoinit", "pycuda.gpuarray", "pycuda.compiler"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pycuda.compiler"].SourceModule = object
    # astropy 4.3.1 (the conda env's) lists two numpy functions by name at import time that the env's numpy 1.26 no longer
    # has; it never calls them here
    for gone, fn in (("asscalar", lambda a: a.item()), ("alen", len)):
        if not hasattr(np, gone):
            setattr(np, gone, fn)
    sys.path.insert(0, REF)
    from source import read as ref_read
    from source import phys_const as ref_pc
    return ref_read, ref_pc


class Q(object):
    pass


def snapshot(q, names):
    return {n: np.asarray(getattr(q, n)) for n in names if hasattr(q, n)}


GRID = ("opac_wave", "nbin", "gauss_y", "ny", "opac_interwave", "opac_deltawave", "ktemp", "ntemp", "kpress", "npress")


def main():
    files = tables(np.random.default_rng(20251))
    h5py_version, hdf5_version = write_files(files)
    ref_read, ref_pc = import_reference_reader()
    exp = {}

    def keep(tag, d):
        for k, v in d.items():
            exp[tag + "/" + k] = v

    q = Q()
    k = ref_read.Read.read_opac_file(q, os.path.join(OUT, "premixed_opac_kdistr.h5"), type="premixed")
    keep("premixed", dict(snapshot(q, GRID + ("opac_scat_cross", "opac_meanmass")), opac_k=np.asarray(k)))

    q = Q()
    k = ref_read.Read.read_opac_file(q, os.path.join(OUT, "DDD_opac_ip_sampling.h5"), type="species", read_grid_parameters=True)
    keep("sampling", dict(snapshot(q, GRID), opac_k=np.asarray(k)))

    q = Q()
    k = ref_read.Read.read_opac_file(q, os.path.join(OUT, "EEE_opac_ip_kdistr.h5"), type="species", read_grid_parameters=True)
    keep("twodim", dict(snapshot(q, GRID), opac_k=np.asarray(k, float).reshape(-1)))

    q = Q()
    k = ref_read.Read.read_opac_file(q, os.path.join(OUT, "BBB_opac_ip.h5"), type="species", read_grid_parameters=False)
    assert not hasattr(q, "nbin")
    keep("species_no_grid", dict(opac_k=np.asarray(k)))

    # the species loop with its file-name fall-backs, then the scattering cross-sections
    r, q = ref_read.Read(), Q()
    r.opacity_path = OUT + os.sep
    q.fl_prec = np.float64
    q.species_list = []
    for name, absorbing, scattering in (("AAA", "yes", "yes"), ("BBB", "yes", "no"), ("CCC", "yes", "yes"), ("H2O", "no", "yes")):
        s = ref_read.Species()
        s.name, s.absorbing, s.scattering = name, absorbing, scattering
        q.species_list.append(s)
    r.read_species_opacities(q)
    q.nlayer, q.ninterface, q.iso = np.int32(3), np.int32(4), np.int32(0)
    r.read_species_scat_cross_sections(q)
    d = snapshot(q, GRID)
    for s in q.species_list:
        if s.absorbing == "yes":
            d["opacity_pretab_" + s.name] = np.asarray(s.opacity_pretab)
        if s.scattering == "yes" and s.name != "H2O":
            d["scat_cross_sect_pretab_" + s.name] = np.asarray(s.scat_cross_sect_pretab)
            d["scat_cross_sect_layer_" + s.name] = np.asarray(s.scat_cross_sect_layer)
            d["scat_cross_sect_interface_" + s.name] = np.asarray(s.scat_cross_sect_interface)
    keep("species", d)

    # the stellar spectrum, addressed by a nested path with and without the leading slash
    for tag, data_set in (("star", "/r50_kdistr/phoenix/gj1214"), ("star_noslash", "r50_kdistr/blackbody/gj1214")):
        r, q = ref_read.Read(), Q()
        r.stellar_model, r.stellar_path, r.stellar_data_set = "file", os.path.join(OUT, "star.h5"), data_set
        q.nbin, q.fl_prec = np.int32(6), np.float64
        r.read_star(q)
        keep(tag, dict(starflux=np.asarray(q.starflux), real_star=np.asarray(q.real_star)))
    # wrong length: the reference raises OverflowError
    r, q = ref_read.Read(), Q()
    r.stellar_model, r.stellar_path, r.stellar_data_set = "file", os.path.join(OUT, "star.h5"), "/r20_kdistr/phoenix/gj1214"
    q.nbin, q.fl_prec = np.int32(6), np.float64
    try:
        r.read_star(q)
        raised = "none"
    except OverflowError:
        raised = "OverflowError"
    exp["star_wrong_length/raised"] = np.array(raised)

    # what h5py itself returns for the storage-type file, as float64 / int64
    import h5py
    with h5py.File(os.path.join(OUT, "dtypes.h5"), "r") as f:
        for k in f.keys():
            v = f[k][()]
            exp["dtypes/" + k] = v.astype(np.float64 if v.dtype.kind == "f" else np.int64)

    np.savez(os.path.join(OUT, "expected.npz"), **exp)

    import astropy
    consts = {n: float(getattr(ref_pc, n)) for n in dir(ref_pc)
              if n.isupper() and isinstance(getattr(ref_pc, n), (float, np.floating))}
    meta = {"astropy": astropy.__version__, "h5py": h5py_version, "hdf5": hdf5_version, "numpy": np.__version__,
            "python": sys.version.split()[0], "constants": consts}
    with open(os.path.join(OUT, "expected_constants.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
