"""Host-COHERENT pinned memory for the buffers the GPU and the host exchange while a kernel is still running
(``rank_forward``: the kernel reads the pid list from, and writes the top-k and its completion word to, host memory
that the host polls).  torch's ``pin_memory()`` gives no coherence guarantee before a synchronisation point, so these
few KB are allocated with ``hipHostMalloc(hipHostMallocCoherent)`` -- through ``maxsim_host_alloc_coherent``, i.e. by the
HIP runtime libmaxsim itself is linked against (resolving ``libamdhip64.so`` by name here could load a second runtime
whose allocations the first one does not know)."""
import ctypes

import numpy as np

from . import _lib


class PinnedBuffer:
    """``nbytes`` of coherent pinned host memory; ``view(dtype, offset, count)`` gives numpy arrays over it; ``ptr`` is
    valid on the host and on the device (unified addressing)."""

    def __init__(self, nbytes):
        p = _lib.lib.maxsim_host_alloc_coherent(int(nbytes))
        if not p:
            raise MemoryError(f"maxsim_host_alloc_coherent({nbytes}) failed")
        self.ptr, self.nbytes = int(p), nbytes
        self._raw = (ctypes.c_char * nbytes).from_address(self.ptr)

    def view(self, dtype, offset, count):
        return np.frombuffer(self._raw, dtype=dtype, count=count, offset=offset)

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                _lib.lib.maxsim_host_free(self.ptr)
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass
