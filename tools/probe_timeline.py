"""On the GPU box, with a timing build (MAXSIM_OUT=tools/ab/stamp.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG -DMAXSIM_STAMP,
MAXSIM_LIB=tools/ab/stamp.so): where the waves of ONE small rerank launch (1 query x 1000 candidates: the reference's
online call) spend their time.  Every wave stamps (100 MHz s_memrealtime) entry, descriptors there, first fetch issued + query
loaded, first tile arrived, last tile reduced, exit."""
import ctypes, os, sys
import numpy as np
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd import _lib
dev = "cuda"
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
nd, nq, ncand = int(os.environ.get("NDOCS", 1000000)), int(os.environ.get("NQ", 1)), int(os.environ.get("NCAND", 1000))
g = torch.Generator(device=dev).manual_seed(0)
idx = torch.empty(nd * 180, 128, device=dev, dtype=dt)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
_lib.lib.maxsim_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(1 << 16, 8, dtype=torch.int64, device=dev)
_lib.lib.maxsim_diag_set_stamp_buffer(stamps.data_ptr())
res = []
for it in range(12):
    cand = torch.randint(0, nd, (nq, ncand), generator=g, device=dev)
    torch.cuda.synchronize()
    stamps.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.score_candidates(Q, cand); e1.record(); torch.cuda.synchronize()
    st = stamps.cpu().numpy()
    st = st[st[:, 0] != 0]
    if it < 4:
        continue
    t0 = st[:, 0].min()
    rel = (st[:, [0, 1, 2, 3, 4, 6]] - t0) * 0.01        # us
    res.append((e0.elapsed_time(e1) * 1e3, rel, st[:, 5]))
ev = np.median([x[0] for x in res])
rel = np.concatenate([x[1] for x in res]); tiles = np.concatenate([x[2] for x in res])
def q(a): return "p10 %5.2f  p50 %5.2f  p90 %5.2f  max %5.2f" % tuple(np.percentile(a, [10, 50, 90, 100]))
print("%s index, %d x %d: %d waves per launch with work, %.1f tiles per wave; events around the launch: median %.1f us" % (os.environ.get("DT", "fp32"), nq, ncand, len(res[0][1]), tiles.mean(), ev))
print("  wave entry after the first wave's entry      ", q(rel[:, 0]))
print("  entry -> descriptor lanes there              ", q(rel[:, 1] - rel[:, 0]))
print("  descriptors -> first fetch issued + Q loaded ", q(rel[:, 2] - rel[:, 1]))
print("  -> first tile arrived                        ", q(rel[:, 3] - rel[:, 2]))
print("  first tile arrived -> last tile reduced      ", q(rel[:, 4] - rel[:, 3]), "  per tile %.2f us" % np.median((rel[:, 4] - rel[:, 3]) / np.maximum(tiles, 1)))
print("  last tile reduced -> exit                    ", q(rel[:, 5] - rel[:, 4]))
print("  wave exit after the first wave's entry       ", q(rel[:, 5]))
per_launch_end = [x[1][:, 5].max() for x in res]
print("  last wave's exit per launch: median %.2f us" % np.median(per_launch_end))
