"""Loads libhelios_hip.so (the gfx950 HIP kernels + C-ABI of include/helios_hip.h) with ctypes.

There is no CPU fallback: if the shared library is missing or cannot be loaded this module raises,
and so does every product entry point that needs it.
"""
import ctypes
import os

from ._cproto import parse_prototypes, bind

_HERE = os.path.dirname(os.path.abspath(__file__))
# HELIOS_HIP_LIB: another build of the same library (same-box A/B measurements select their variant with it instead of
# copying over the in-tree file); the product loads the in-tree library
LIB_PATH = os.environ.get("HELIOS_HIP_LIB") or os.path.join(_HERE, "libhelios_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "helios_hip.h")

_lib = None
_protos = None


class HeliosHipError(RuntimeError):
    pass


def prototypes():
    global _protos
    if _protos is None:
        with open(HEADER_PATH) as f:
            _protos = parse_prototypes(f.read(), "hx_")
    return _protos


def lib():
    """the bound ctypes library; raises HeliosHipError if it cannot be loaded"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HeliosHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). helios_amd has no CPU fallback." % LIB_PATH)
        try:
            # If PyTorch is part of the process its bundled HIP runtime must be the one in use
            # (same SONAME as /opt/rocm's): import it first when it is importable and already wanted.
            import sys
            if "torch" in sys.modules:
                import torch  # noqa: F401
            _lib = bind(ctypes.CDLL(LIB_PATH), prototypes())
        except OSError as e:
            raise HeliosHipError("cannot load %s: %s" % (LIB_PATH, e))
    return _lib


def check(ctx_handle, rc, what=""):
    if rc != 0:
        msg = lib().hx_last_error(ctx_handle) if ctx_handle else b""
        raise HeliosHipError("%s failed with status %d: %s" % (what or "libhelios_hip call", rc,
                                                             (msg or b"").decode("utf-8", "replace")))
