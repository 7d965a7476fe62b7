"""Host-COHERENT pinned memory for the buffers the GPU and the host exchange while a kernel is still running
(``rank_forward``: the kernel reads the pid list from, and writes the top-k and its completion word to, host memory
that the host polls).  torch's ``pin_memory()`` gives no coherence guarantee before a synchronisation point, so these
few KB are allocated with ``hipHostMalloc(hipHostMallocCoherent)`` directly."""
import ctypes

import numpy as np

_HIP_HOST_MALLOC_COHERENT = 0x40000000
_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")      # the runtime torch already loaded
        _hip.hipHostMalloc.restype = ctypes.c_int
        _hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        _hip.hipHostFree.restype = ctypes.c_int
        _hip.hipHostFree.argtypes = [ctypes.c_void_p]
    return _hip


class PinnedBuffer:
    """``nbytes`` of coherent pinned host memory; ``view(dtype, offset, count)`` gives numpy arrays over it; ``ptr`` is
    valid on the host and on the device (unified addressing)."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        rc = _lib().hipHostMalloc(ctypes.byref(p), nbytes, _HIP_HOST_MALLOC_COHERENT)
        if rc != 0 or not p.value:
            raise MemoryError(f"hipHostMalloc({nbytes}, coherent) failed with {rc}")
        self.ptr, self.nbytes = p.value, nbytes
        self._raw = (ctypes.c_char * nbytes).from_address(self.ptr)

    def view(self, dtype, offset, count):
        return np.frombuffer(self._raw, dtype=dtype, count=count, offset=offset)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib().hipHostFree(self.ptr)
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass
