"""Device memory for the GPU tests, straight from the HIP runtime the engine itself is linked against (ctypes on
libamdhip64): the tests need buffers in HBM to hand to the C ABI, nothing more."""
import ctypes as C

import numpy as np

_rt = None


def rt():
    global _rt
    if _rt is None:
        from quisk_amd import lib
        lib()                                   # the engine pulls in the runtime it was linked with; bind to that copy
        L = C.CDLL("libamdhip64.so.7")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _rt = L
    return _rt


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class DevBuf:
    """A device copy of a numpy array (or `nbytes` of zeros); .ptr is the device address."""

    def __init__(self, array=None, nbytes=0):
        L = rt()
        self.nbytes = int(array.nbytes if array is not None else nbytes)
        p = C.c_void_p()
        _ok(L.hipMalloc(C.byref(p), max(self.nbytes, 16)), "hipMalloc")
        self.ptr = p.value
        if array is not None:
            a = np.ascontiguousarray(array)
            _ok(L.hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, 1), "hipMemcpy H2D")
        elif self.nbytes:
            z = np.zeros(self.nbytes, dtype=np.uint8)
            _ok(L.hipMemcpy(self.ptr, z.ctypes.data, self.nbytes, 1), "hipMemcpy H2D")

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def to_host(self, dtype=np.float64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _ok(rt().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def stream_create():
    L = rt()
    L.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    st = C.c_void_p()
    _ok(L.hipStreamCreate(C.byref(st)), "hipStreamCreate")
    return st


def stream_destroy(st):
    L = rt()
    L.hipStreamDestroy.argtypes = [C.c_void_p]
    _ok(L.hipStreamDestroy(st), "hipStreamDestroy")


def sync():
    _ok(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")


def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    _ok(rt().hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
    return f.value
