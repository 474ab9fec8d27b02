"""HDF5 containers through the HDF5 C library itself (ctypes), for hosts where `h5py` is not installed.

The reference opens its opacity tables, scattering cross-sections and stellar spectra with
`h5py.File(path, "r")` and takes whole datasets (`source/read.py:1044-1101`, `:1202-1203`, `:1639-1640`).  This
module serves exactly that use -- open read-only, ask whether a (possibly nested) dataset exists, read it whole
as a numpy array -- on `libhdf5.so` (1.10 / 1.12 / 1.14: only calls whose signatures are the same in all of
them), plus a small writer (`write`) so that .npz tables can be turned into .h5 and tests can make files.

    with hdf5_lite.File("table.h5") as f:
        f.keys()                                  # names of the root group's links
        "r50_kdistr/phoenix/gj1214" in f          # nested paths as in h5py
        k = f["kpoints"][()]                      # whole dataset, fp64 / int64 / bytes; also f[name][:]

The library is looked for in $HELIOS_HDF5_LIB, on the loader's path and in the usual prefixes; `available()`
tells whether one was found.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_hid = C.c_int64
_hsize = C.c_uint64
_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT, _H5S_ALL = 0, 0
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0

_lib = None
_lib_error = None


def _candidates():
    env = os.environ.get("HELIOS_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("libhdf5.so", "libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*",
                os.path.join(os.environ.get("CONDA_PREFIX", "/opt/conda"), "lib", "libhdf5.so*"), "/opt/conda/lib/libhdf5.so*"):
        hits = sorted(glob.glob(pat)) if "*" in pat or os.path.isabs(pat) else [pat]
        for h in hits:
            yield h


def _load():
    global _lib, _lib_error
    if _lib is not None or _lib_error is not None:
        return _lib
    errors = []
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                raise OSError("H5open failed")
        except (OSError, AttributeError) as e:
            errors.append("%s: %s" % (cand, e))
            continue
        maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
        lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
        if (maj.value, mnr.value) < (1, 10):     # hid_t is a 32-bit int before 1.10
            errors.append("%s: HDF5 %d.%d is older than 1.10" % (cand, maj.value, mnr.value))
            continue
        lib._version = (maj.value, mnr.value, rel.value)
        lib._path = cand
        try:
            _declare(lib)
        except (AttributeError, ValueError) as e:    # a build that lacks a symbol this module needs: try the next candidate
            errors.append("%s: %s" % (cand, e))
            continue
        lib.H5Eset_auto2(_hid(0), None, None)    # errors are reported through return codes, not printed
        _lib = lib
        return _lib
    _lib_error = "; ".join(errors) or "no libhdf5 found"
    return None


def _declare(lib):
    sig = {
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]), "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fclose": (C.c_int, [_hid]), "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]), "H5Dclose": (C.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]), "H5Dget_type": (_hid, [_hid]), "H5Sclose": (C.c_int, [_hid]),
        "H5Tclose": (C.c_int, [_hid]), "H5Tget_class": (C.c_int, [_hid]), "H5Tget_size": (C.c_size_t, [_hid]),
        "H5Tis_variable_str": (C.c_int, [_hid]), "H5Tcopy": (_hid, [_hid]), "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Sget_simple_extent_npoints": (C.c_int64, [_hid]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]), "H5Screate": (_hid, [C.c_int]),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]), "H5Gclose": (C.c_int, [_hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [_hid, C.c_char_p, C.c_int, C.c_int, _hsize, C.c_char_p, C.c_size_t, _hid]),
        "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        "H5Pcreate": (_hid, [_hid]), "H5Pclose": (C.c_int, [_hid]),
        "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]), "H5Pset_deflate": (C.c_int, [_hid, C.c_uint]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # variable-length strings are handed back through H5Treclaim (1.12 on) or H5Dvlen_reclaim (deprecated there, absent from
    # builds without the deprecated API): whichever the library has; with neither the few bytes stay with the library
    lib._reclaim = None
    for name in ("H5Treclaim", "H5Dvlen_reclaim"):
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = C.c_int, [_hid, _hid, _hid, C.c_void_p]
            lib._reclaim = fn
            break
    lib.H5Tget_sign.restype, lib.H5Tget_sign.argtypes = C.c_int, [_hid]
    for g in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_INT64_g", "H5T_NATIVE_UINT64_g", "H5T_C_S1_g", "H5P_CLS_DATASET_CREATE_ID_g"):
        setattr(lib, "_" + g, _hid.in_dll(lib, g).value)


def available():
    """True when an HDF5 library (>= 1.10) can be loaded"""
    return _load() is not None


def library():
    """(path, (major, minor, release)) of the HDF5 library in use"""
    lib = _need()
    return lib._path, lib._version


def _need():
    lib = _load()
    if lib is None:
        raise IOError("no HDF5 library: %s (set HELIOS_HDF5_LIB to a libhdf5.so, or install h5py)" % _lib_error)
    return lib


class _Dataset(object):
    """what `File.__getitem__` returns: the whole dataset through `[()]` or `[:]`, as h5py's datasets give it"""

    def __init__(self, owner, name):
        self._owner, self._name = owner, name

    def __getitem__(self, key):
        a = self._owner.read(self._name)
        return a if (isinstance(key, tuple) and key == ()) else a[key]

    @property
    def shape(self):
        return self._owner.read(self._name).shape


class File(object):
    """read-only HDF5 file"""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("hdf5_lite.File opens read-only; hdf5_lite.write makes files")
        self._lib = _need()
        self._path = str(path)
        if not os.path.exists(self._path):
            raise IOError("Unable to open file (no such file: %s)" % self._path)
        self._id = self._lib.H5Fopen(self._path.encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
        if self._id < 0:
            raise IOError("Unable to open file (%s is not an HDF5 file or cannot be read)" % self._path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_id", -1) >= 0:
            self._lib.H5Fclose(self._id)
            self._id = -1

    def _exists(self, name):
        """every link of a path has to be there (H5Lexists fails on a path with a missing intermediate group)"""
        parts = [p for p in str(name).split("/") if p]
        if not parts:
            return False
        for k in range(1, len(parts) + 1):
            if self._lib.H5Lexists(self._id, "/".join(parts[:k]).encode(), _H5P_DEFAULT) <= 0:
                return False
        return True

    def __contains__(self, name):
        return self._exists(name)

    def keys(self, group="/"):
        out = []
        buf = C.create_string_buffer(1024)
        n = 0
        while True:
            got = self._lib.H5Lget_name_by_idx(self._id, str(group).encode(), _H5_INDEX_NAME, _H5_ITER_INC, n, buf, 1024,
                                               _H5P_DEFAULT)
            if got < 0:
                break
            out.append(buf.value.decode())
            n += 1
        return out

    def is_dataset(self, name):
        if not self._exists(name):
            return False
        d = self._lib.H5Dopen2(self._id, str(name).encode(), _H5P_DEFAULT)
        if d < 0:
            return False
        self._lib.H5Dclose(d)
        return True

    def __getitem__(self, name):
        if not self._exists(name):
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % name)
        if not self.is_dataset(name):
            raise KeyError("'%s' is a group, not a dataset" % name)
        return _Dataset(self, name)

    def read(self, name):
        """the whole dataset: floats as fp64, integers as int64, strings as bytes objects"""
        lib = self._lib
        d = lib.H5Dopen2(self._id, str(name).encode(), _H5P_DEFAULT)
        if d < 0:
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % name)
        space = lib.H5Dget_space(d)
        ftype = lib.H5Dget_type(d)
        try:
            nd = lib.H5Sget_simple_extent_ndims(space)
            dims = (_hsize * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(space, dims, None)
            shape = tuple(int(dims[k]) for k in range(max(nd, 0)))
            npoints = int(lib.H5Sget_simple_extent_npoints(space))
            cls = lib.H5Tget_class(ftype)
            if cls == _H5T_FLOAT or cls == _H5T_INTEGER:
                # unsigned 64-bit storage is read as such (through int64 its upper half would saturate silently) and handed
                # over as int64 only where every value fits
                wide_unsigned = cls == _H5T_INTEGER and lib.H5Tget_sign(ftype) == 0 and lib.H5Tget_size(ftype) >= 8
                out = np.empty(shape, np.float64 if cls == _H5T_FLOAT else (np.uint64 if wide_unsigned else np.int64))
                mem = (lib._H5T_NATIVE_DOUBLE_g if cls == _H5T_FLOAT else
                       lib._H5T_NATIVE_UINT64_g if wide_unsigned else lib._H5T_NATIVE_INT64_g)
                if npoints and lib.H5Dread(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                    raise IOError("cannot read dataset '%s' of %s" % (name, self._path))
                if wide_unsigned and (not npoints or int(out.max()) <= np.iinfo(np.int64).max):
                    out = out.astype(np.int64)
                return out
            if cls == _H5T_STRING:
                mem = lib.H5Tcopy(ftype)       # the file's own string type (character set, padding) serves as memory type
                try:
                    if lib.H5Tis_variable_str(ftype) > 0:
                        ptrs = (C.c_char_p * max(npoints, 1))()
                        if npoints and lib.H5Dread(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, ptrs) < 0:
                            raise IOError("cannot read dataset '%s' of %s" % (name, self._path))
                        vals = [ptrs[k] or b"" for k in range(npoints)]
                        if npoints and lib._reclaim is not None:
                            lib._reclaim(mem, space, _H5P_DEFAULT, ptrs)
                    else:
                        size = int(lib.H5Tget_size(ftype))
                        raw = C.create_string_buffer(max(npoints, 1) * size)
                        if npoints and lib.H5Dread(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, raw) < 0:
                            raise IOError("cannot read dataset '%s' of %s" % (name, self._path))
                        vals = [raw.raw[k * size:(k + 1) * size].split(b"\0")[0] for k in range(npoints)]
                finally:
                    lib.H5Tclose(mem)
                out = np.empty(npoints, object)
                out[:] = vals
                return out.reshape(shape) if shape else out[0]
            raise IOError("dataset '%s' of %s has a type this reader does not take (class %d)" % (name, self._path, cls))
        finally:
            lib.H5Tclose(ftype)
            lib.H5Sclose(space)
            lib.H5Dclose(d)


def write(path, datasets, compression=None):
    """writes {name: array} -- names may be nested paths, intermediate groups are made -- as fp64 / int64 / fixed-length
    string datasets; `compression` = gzip level (chunked, one chunk per dataset) or None (contiguous)"""
    lib = _need()
    f = lib.H5Fcreate(str(path).encode(), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
    if f < 0:
        raise IOError("cannot create %s" % path)
    try:
        made = set()
        for name, value in datasets.items():
            parts = [p for p in str(name).split("/") if p]
            for k in range(1, len(parts)):
                g = "/".join(parts[:k])
                if g not in made:
                    made.add(g)
                    if lib.H5Lexists(f, g.encode(), _H5P_DEFAULT) <= 0:
                        gid = lib.H5Gcreate2(f, g.encode(), _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
                        if gid < 0:
                            raise IOError("cannot create group %s in %s" % (g, path))
                        lib.H5Gclose(gid)
            a = np.asarray(value)
            tclose = None
            if a.dtype.kind in "SU" or a.dtype == object:
                items = [s if isinstance(s, bytes) else str(s).encode() for s in np.atleast_1d(a).ravel()]
                size = max([len(s) for s in items] + [1])
                mem = tclose = lib.H5Tcopy(lib._H5T_C_S1_g)
                lib.H5Tset_size(mem, size)
                buf = C.create_string_buffer(b"".join(s.ljust(size, b"\0") for s in items), max(len(items), 1) * size)
                ptr = C.cast(buf, C.c_void_p)
            elif a.dtype.kind in "iub":
                a = np.array(a, np.int64, order="C")
                mem, ptr = lib._H5T_NATIVE_INT64_g, a.ctypes.data_as(C.c_void_p)
            else:
                a = np.array(a, np.float64, order="C")
                mem, ptr = lib._H5T_NATIVE_DOUBLE_g, a.ctypes.data_as(C.c_void_p)
            nd = a.ndim
            dims = (_hsize * max(nd, 1))(*[int(n) for n in a.shape])
            space = lib.H5Screate_simple(nd, dims, None) if nd > 0 else lib.H5Screate(0)     # 0 = H5S_SCALAR
            dcpl = _H5P_DEFAULT
            if compression is not None and nd > 0 and a.size > 0 and tclose is None:
                dcpl = lib.H5Pcreate(lib._H5P_CLS_DATASET_CREATE_ID_g)
                lib.H5Pset_chunk(dcpl, nd, dims)
                lib.H5Pset_deflate(dcpl, int(compression))
            d = lib.H5Dcreate2(f, "/".join(parts).encode(), mem, space, _H5P_DEFAULT, dcpl, _H5P_DEFAULT)
            ok = d >= 0 and (a.size == 0 or lib.H5Dwrite(d, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, ptr) >= 0)
            if d >= 0:
                lib.H5Dclose(d)
            if dcpl != _H5P_DEFAULT:
                lib.H5Pclose(dcpl)
            lib.H5Sclose(space)
            if tclose is not None:
                lib.H5Tclose(tclose)
            if not ok:
                raise IOError("cannot write dataset '%s' to %s" % (name, path))
    finally:
        lib.H5Fclose(f)
