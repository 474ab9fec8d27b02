"""TEST INFRASTRUCTURE ONLY: ctypes front-ends of the two CPU checkers.

    oracle.port  -- libhelios_oracle.so, this repo's C restatement (oracle/helios_oracle.c)
    oracle.ref   -- _ref/libhelios_ref.so, the reference's own kernels.cu compiled for the host
                    (None when that file has not been built / is not present)

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


port = _load_port()
ref = _load_ref()
# tiny test problems + hundreds of OpenMP threads (or a CPU quota below the visible core count) make
# every parallel region crawl: default to at most 8 threads; bench.py's cpu_baseline sets its own.
try:
    _ncpu = len(os.sched_getaffinity(0))
except AttributeError:
    _ncpu = os.cpu_count() or 1
port.set_num_threads(max(1, min(8, _ncpu)))
