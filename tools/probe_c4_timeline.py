"""On the GPU box with a timing build (MAXSIM_LIB=tools/ab/stamp.so): when do the waves of a C4 launch (256 queries x 1000
eight-token docs, k_maxsim_stream_f32h) start and end?  The launch is one round of workgroups that all do the same work."""
import ctypes, os, sys
import numpy as np
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
from colbert_amd import _lib
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c4"]
doclens = [wl["ld"]] * wl["ndocs"]
idx = bench.build_index(sum(doclens), 128, dev, 1234, torch.float32)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
nq = int(os.environ.get("NQ", 256))
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(nq, wl["lq"], 128, generator=g, device=dev), dim=-1)
_lib.lib.maxsim_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(1 << 17, 8, dtype=torch.int64, device=dev)
_lib.lib.maxsim_diag_set_stamp_buffer(stamps.data_ptr())
for it in range(8):
    cand = torch.randint(0, len(doclens), (nq, 1000), generator=g, device=dev)
    torch.cuda.synchronize(); stamps.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.score_candidates(Q, cand); e1.record(); torch.cuda.synchronize()
    if it < 5:
        continue
    st = stamps.cpu().numpy(); st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    beg, end = (st[:, 0] - t0) * 0.01, (st[:, 6] - t0) * 0.01
    dur = end - beg
    xcc = (st[:, 7] >> 32) & 0xf
    hw = st[:, 7] & 0xffffffff
    cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    print("launch %d: events %.1f us; %d waves; start p50 %.1f max %.1f | end p10 %.1f p50 %.1f p90 %.1f max %.1f | duration p10 %.1f p50 %.1f p90 %.1f max %.1f"
          % (it, e0.elapsed_time(e1) * 1e3, len(st), np.median(beg), beg.max(), *np.percentile(end, [10, 50, 90, 100]), *np.percentile(dur, [10, 50, 90, 100])))
    print("   bytes/us while k waves still run: ", end="")
    order = np.sort(end)
    for frac in (0.5, 0.75, 0.9, 0.97, 1.0):
        print("%.0f%% done at %.1f us; " % (frac * 100, order[int(frac * len(order)) - 1]), end="")
    print()
    print("   mean end by XCD:", " ".join("%d:%.1f" % (x, end[xcc == x].mean()) for x in range(8)))
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    ends_by_cu = {k: end[key == k].max() for k in np.unique(key)}
    v = np.array(list(ends_by_cu.values()))
    print("   per-CU last end: %d CUs, p10 %.1f p50 %.1f p90 %.1f max %.1f" % (len(v), *np.percentile(v, [10, 50, 90, 100])))
