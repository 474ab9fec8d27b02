"""Device context and device arrays: the thin replacement for what the reference takes from PyCUDA
(`pycuda.autoinit`, `gpuarray.to_gpu(a)`, `dev.get()`, `cuda.mem_alloc(nbytes)`;
source/quantities.py:463-665).  Device memory is owned by libhelios_hip.so; Python only holds
opaque pointers."""
import ctypes

import numpy as np

from . import _lib


class HxDiag(ctypes.Structure):
    """hx_diag of include/helios_hip.h"""
    _fields_ = [("negative_down_flux", ctypes.c_uint64), ("negative_up_flux", ctypes.c_uint64),
                ("g_limited", ctypes.c_uint64), ("ro_rebin_skipped", ctypes.c_uint64),
                ("energy_correction", ctypes.c_double), ("ro_fixup_passes", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64 * 2)]


class Context(object):
    """one HIP device + one stream (include/helios_hip.h section 1)"""

    def __init__(self, device_id=0):
        self._l = _lib.lib()
        h = ctypes.c_void_p()
        rc = self._l.hx_create(int(device_id), ctypes.byref(h))
        if rc != 0:
            raise _lib.HeliosHipError("hx_create(device %d) failed with status %d (is a gfx950 GPU "
                                      "visible?)" % (device_id, rc))
        self.handle = h
        self.device_id = device_id

    def check(self, rc, what=""):
        _lib.check(self.handle, rc, what)

    def synchronize(self):
        self.check(self._l.hx_sync(self.handle), "hx_sync")

    def name(self):
        buf = ctypes.create_string_buffer(256)
        self.check(self._l.hx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def mem_info(self):
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        self.check(self._l.hx_mem_info(self.handle, ctypes.byref(free), ctypes.byref(total)))
        return free.value, total.value

    def diag(self):
        """the conditions the reference reports through device-side printf, as counts (hx_diag in helios_hip.h);
        blocks until the stream has drained"""
        d = HxDiag()
        self.check(self._l.hx_diag_read(self.handle, ctypes.byref(d)), "hx_diag_read")
        return {k: getattr(d, k) for k, _ in HxDiag._fields_ if k != "reserved"}

    def diag_reset(self):
        self.check(self._l.hx_diag_reset(self.handle), "hx_diag_reset")

    def timer_start(self):
        self.check(self._l.hx_timer_start(self.handle))

    def timer_stop_ms(self):
        ms = ctypes.c_double()
        self.check(self._l.hx_timer_stop_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    # --- arrays ---------------------------------------------------------------------------------
    def to_gpu(self, array):
        """counterpart of pycuda.gpuarray.to_gpu"""
        a = np.ascontiguousarray(array)
        d = DeviceArray(self, a.shape, a.dtype)
        d.set(a)
        return d

    def zeros(self, shape, dtype=np.float64):
        d = DeviceArray(self, shape, dtype)
        d.fill_zero()
        return d

    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def close(self):
        if self.handle:
            self._l.hx_destroy(self.handle)
            self.handle = None


class DeviceArray(object):
    """a typed device buffer with the two PyCUDA methods the reference relies on: get() / set()"""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        p = ctypes.c_void_p()
        ctx.check(ctx._l.hx_alloc(ctx.handle, self.nbytes, ctypes.byref(p)), "hx_alloc")
        self.ptr = p

    def _as(self, ctype):
        return ctypes.cast(self.ptr, ctypes.POINTER(ctype))

    @property
    def d(self):
        """pointer typed for a `double*` parameter"""
        return self._as(ctypes.c_double)

    @property
    def i(self):
        """pointer typed for an `int*` parameter"""
        return self._as(ctypes.c_int32)

    def get(self):
        out = np.empty(self.shape, self.dtype)
        self.ctx.check(self.ctx._l.hx_d2h(self.ctx.handle, out.ctypes.data_as(ctypes.c_void_p), self.ptr,
                                          self.nbytes), "hx_d2h")
        return out

    def set(self, array):
        a = np.ascontiguousarray(array, dtype=self.dtype)
        if a.size != self.size:
            raise ValueError("size mismatch: device array has %d elements, host array %d" % (self.size, a.size))
        self.ctx.check(self.ctx._l.hx_h2d(self.ctx.handle, self.ptr, a.ctypes.data_as(ctypes.c_void_p),
                                          self.nbytes), "hx_h2d")
        return self

    def copy_from_device(self, src_ptr, nbytes):
        """device-to-device copy of `nbytes` from a raw device pointer into the start of this array"""
        if nbytes > self.nbytes:
            raise ValueError("copy_from_device: %d bytes into an array of %d" % (nbytes, self.nbytes))
        self.ctx.check(self.ctx._l.hx_d2d(self.ctx.handle, self.ptr, src_ptr, ctypes.c_size_t(nbytes)), "hx_d2d")
        return self

    def fill_zero(self):
        self.ctx.check(self.ctx._l.hx_memset0(self.ctx.handle, self.ptr, self.nbytes), "hx_memset0")
        return self

    def free(self):
        if self.ptr:
            self.ctx._l.hx_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.ctx._l.hx_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
