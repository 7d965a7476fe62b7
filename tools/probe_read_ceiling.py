"""On the GPU box: the read ceiling of the kernels' fetch path (nt LDS-DMA into per-wave rings, nothing consumed) for
contiguous streams and for scattered granules of 1 KiB .. 64 KiB (a short doc), by ring shape."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colbert_amd import _lib
dev = torch.device("cuda", 0)
nbytes = 16 << 30
buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
buf.view(torch.int32)[:] = 0x3d800000
st = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def T(f, n=3):
    f(); e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n
names = {0: "1x16KiB 8w/CU", 1: "2x8KiB 8w/CU", 2: "1x8KiB 16w/CU", 3: "1x8KiB 16w/CU, 2x bytes/wave"}
got = ctypes.c_int64(0)
for v in (0, 1, 2):
    _lib.lib.maxsim_hbm_read_probe(buf.data_ptr(), nbytes, v, ctypes.addressof(got), st)
    ms = T(lambda: _lib.lib.maxsim_hbm_read_probe(buf.data_ptr(), nbytes, v, None, st))
    print("contiguous          ring %-28s %.0f GB/s" % ("2x16KiB 4w/CU" if v == 2 else names[v], got.value / ms / 1e6))
for rd in (1 << 30, 16 << 30):
  for gran in (1024, 4096, 16384, 65536):
    for v in (0, 2, 3):
        ms = T(lambda: _lib.lib.maxsim_hbm_read_probe_scattered(buf.data_ptr(), nbytes, gran, v, rd, st))
        print("granule %6d B  launch of %5d MiB  ring %-28s %.0f GB/s" % (gran, rd >> 20, names[v], rd / ms / 1e6))
