"""TEST INFRASTRUCTURE ONLY: ctypes front-ends of the two CPU checkers.

    oracle.port  -- libhelios_oracle.so, this repo's C restatement (oracle/helios_oracle.c)
    oracle.ref   -- _ref/libhelios_ref.so, the reference's own kernels.cu compiled for the host
                    through oracle/ref_shim.h (None when that file has not been built / is not
                    present).  A CPU-side convenience; it is NOT the pin.
    oracle.refgpu -- _ref/libhelios_ref_gfx950.so, THE PIN: the reference's kernels.cu compiled
                    unmodified by hipcc for gfx950 (no shim) and run on the GPU with the
                    reference's launch geometry (None without the file or without a GPU).
                    Same function names and argument lists as oracle.ref.

`ref` and `refgpu` are loaded on first use, so a process that never touches them does not map them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package `helios_amd` never does (tests/test_abi.py::test_product_does_not_import_oracle).

Functions are exposed under their C names without the prefix, e.g. oracle.port.fband_noniso(...)
and oracle.ref.fband_noniso(...).  Array arguments are C-contiguous numpy arrays (float64 or
int32) that the C code reads/writes in place; scalars are Python numbers.
"""
import ctypes
import os
import subprocess

import numpy as np

from helios_amd._cproto import parse_prototypes, bind

_HERE = os.path.dirname(os.path.abspath(__file__))
_PD = ctypes.POINTER(ctypes.c_double)
_PI = ctypes.POINTER(ctypes.c_int32)


class _CLib(object):
    def __init__(self, path, proto_text, prefix):
        self._lib = ctypes.CDLL(path)
        self._protos = parse_prototypes(proto_text, prefix)
        bind(self._lib, self._protos)
        self.path = path
        for name, (_r, argtypes, argnames) in self._protos.items():
            setattr(self, name[len(prefix):], self._make(name, argtypes, argnames))

    def names(self):
        return sorted(self._protos)

    def _make(self, name, argtypes, argnames):
        fn = getattr(self._lib, name)

        def call(*args):
            if len(args) != len(argtypes):
                raise TypeError("%s expects %d arguments (%s), got %d"
                                % (name, len(argtypes), ", ".join(argnames), len(args)))
            conv = []
            for a, t, n in zip(args, argtypes, argnames):
                if t is _PD or t is _PI:
                    want = np.float64 if t is _PD else np.int32
                    if a is None:
                        conv.append(None)
                        continue
                    if not isinstance(a, np.ndarray) or a.dtype != want or not a.flags["C_CONTIGUOUS"]:
                        raise TypeError("%s: argument %s must be a C-contiguous %s array"
                                        % (name, n, np.dtype(want).name))
                    conv.append(a.ctypes.data_as(t))
                else:
                    conv.append(a)
            return fn(*conv)

        call.__name__ = name
        return call


def build(verbose=False):
    """(Re)build both checkers with oracle/Makefile."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout + out.stderr)
    if out.returncode != 0:
        raise RuntimeError("building the oracle failed")


def _load_port():
    path = os.path.join(_HERE, "libhelios_oracle.so")
    if not os.path.exists(path):
        build()
    with open(os.path.join(_HERE, "helios_oracle.h")) as f:
        return _CLib(path, f.read(), "orc_")


def _load_ref():
    path = os.path.join(_HERE, "_ref", "libhelios_ref.so")
    if not os.path.exists(path):
        return None
    with open(os.path.join(_HERE, "ref_driver.cpp")) as f:
        return _CLib(path, f.read(), "ref_")


class RefBuf(object):
    """a device-resident array of the GPU reference build (stays on the GPU between calls)"""

    def __init__(self, owner, arr):
        arr = np.ascontiguousarray(arr)
        self._o = owner
        self.dtype, self.shape, self.nbytes = arr.dtype, arr.shape, arr.nbytes
        self.ptr = owner._alloc(arr.nbytes)
        owner._h2d(self.ptr, arr.ctypes.data, arr.nbytes)

    def get(self):
        out = np.empty(self.shape, self.dtype)
        self._o._d2h(out.ctypes.data, self.ptr, self.nbytes)
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        self._o._h2d(self.ptr, arr.ctypes.data, arr.nbytes)

    def free(self):
        if self.ptr:
            self._o._free(self.ptr)
            self.ptr = None


class _GpuRefLib(object):
    """the reference's kernels on the GPU.  Array arguments are numpy arrays (uploaded before the
    launch, every one downloaded again after it -- the kernels take non-const pointers throughout)
    or RefBuf objects (left on the device)."""

    def __init__(self, path, proto_text):
        self._lib = ctypes.CDLL(path)
        L = self._lib
        L.refgpu_device_count.restype = ctypes.c_int
        L.refgpu_alloc.restype = ctypes.c_void_p
        L.refgpu_alloc.argtypes = [ctypes.c_size_t]
        L.refgpu_free.argtypes = [ctypes.c_void_p]
        L.refgpu_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.refgpu_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.refgpu_status.restype = ctypes.c_int
        self._alloc, self._free, self._h2d, self._d2h = L.refgpu_alloc, L.refgpu_free, L.refgpu_h2d, L.refgpu_d2h
        self._protos = parse_prototypes(proto_text, "ref_")
        bind(L, self._protos)
        self.path = path
        for name, (_r, argtypes, argnames) in self._protos.items():
            setattr(self, name[4:], self._make(name, argtypes, argnames))

    def device_count(self):
        return self._lib.refgpu_device_count()

    def names(self):
        return sorted(self._protos)

    def buf(self, arr):
        return RefBuf(self, arr)

    def _make(self, name, argtypes, argnames):
        fn = getattr(self._lib, name)

        def call(*args):
            if len(args) != len(argtypes):
                raise TypeError("%s expects %d arguments (%s), got %d"
                                % (name, len(argtypes), ", ".join(argnames), len(args)))
            conv, temps = [], []
            for a, t, n in zip(args, argtypes, argnames):
                if t is _PD or t is _PI:
                    want = np.float64 if t is _PD else np.int32
                    if isinstance(a, RefBuf):
                        if a.dtype != want:
                            raise TypeError("%s: device argument %s must be %s" % (name, n, np.dtype(want).name))
                        conv.append(ctypes.cast(a.ptr, t))
                        continue
                    if not isinstance(a, np.ndarray) or a.dtype != want or not a.flags["C_CONTIGUOUS"]:
                        raise TypeError("%s: argument %s must be a C-contiguous %s array"
                                        % (name, n, np.dtype(want).name))
                    d = RefBuf(self, a)
                    temps.append((a, d))
                    conv.append(ctypes.cast(d.ptr, t))
                else:
                    conv.append(a)
            fn(*conv)
            for a, d in temps:
                if a.flags.writeable:
                    a[...] = d.get()
                d.free()
            st = self._lib.refgpu_status()
            if st != 0:
                raise RuntimeError("%s on the GPU: HIP error %d" % (name, st))

        call.__name__ = name
        return call


def _load_refgpu():
    path = os.path.join(_HERE, "_ref", "libhelios_ref_gfx950.so")
    if not os.path.exists(path):
        return None
    with open(os.path.join(_HERE, "ref_driver_gfx950.hip")) as f:
        text = f.read()
    lib = _GpuRefLib(path, text)
    if lib.device_count() < 1:
        return None
    return lib


port = _load_port()
_lazy = {}


def __getattr__(name):
    # PEP 562: oracle.ref / oracle.refgpu are mapped on first use only
    if name in ("ref", "refgpu"):
        if name not in _lazy:
            _lazy[name] = _load_ref() if name == "ref" else _load_refgpu()
        return _lazy[name]
    raise AttributeError(name)


# tiny test problems + hundreds of OpenMP threads (or a CPU quota below the visible core count) make
# every parallel region crawl: default to at most 8 threads; bench.py's cpu_baseline sets its own.
try:
    _ncpu = len(os.sched_getaffinity(0))
except AttributeError:
    _ncpu = os.cpu_count() or 1
port.set_num_threads(max(1, min(8, _ncpu)))
