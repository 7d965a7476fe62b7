"""On the GPU box: where the time of one rank_forward(Q, 1000 pids, depth=100) call goes: host conversions, the one
library call (two launches + stream sync), the kernels themselves (HIP events), result lists."""
import array, ctypes, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd import _lib
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
nd = int(os.environ.get("NDOCS", 1000000))
idx = F.normalize(torch.randn(nd * 180, 128, generator=gen, device=dev), dim=-1)
if os.environ.get("DTYPE", "fp32") == "fp16":        # the reference's storage dtype (colbert_ranker.py:62)
    idx = idx.half()
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(1, 32, 128, generator=gen, device=dev), dim=-1).permute(0, 2, 1)    # [1,h,Lq] view
lists = [torch.randperm(nd)[:1000].tolist() for _ in range(64)]
it = {"i": 0}
def nxt():
    it["i"] += 1
    return lists[it["i"] % len(lists)]
def T(f, n=300):
    for _ in range(30): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
print("rank_forward total (fresh docs each call)   %.1f us" % T(lambda: r.rank_forward(Q, nxt(), depth=100)))
p0 = lists[0]
print("rank_forward total (same 1000 docs: cached) %.1f us" % T(lambda: r.rank_forward(Q, p0, depth=100)))
print("  array('q', pids) + memmove to pinned      %.1f us" % T(lambda: ctypes.memmove(r._tls.ws.in_ptr, array.array("q", p0).buffer_info()[0], 8000)))
print("  Q.permute/.to/.contiguous (no-op path)    %.1f us" % T(lambda: Q.permute(0, 2, 1).to(device=r.device, dtype=torch.float32).contiguous()))
print("  torch.cuda.current_stream().cuda_stream   %.1f us" % T(lambda: torch.cuda.current_stream(r.device).cuda_stream))
ws = r._tls.ws
Qt = Q.permute(0, 2, 1).contiguous()
st = torch.cuda.current_stream().cuda_stream
def call(sync):
    ctypes.memmove(ws.in_ptr, array.array("q", nxt()).buffer_info()[0], 8000)
    return _lib.lib.maxsim_rank_forward(ctypes.byref(r._iv), Qt.data_ptr(), 0, 32, ws.in_ptr, 1000, 100, ws.scratch_ptr, ws.out_p_ptr, ws.out_s_ptr, ws.flag_ptr, sync, st)
print("  maxsim_rank_forward(sync=1) incl. memmove  %.1f us" % T(lambda: call(1)))
print("  maxsim_rank_forward(sync=0) launch cost    %.1f us" % T(lambda: call(0)))
print("  2x tolist of 100                          %.1f us" % T(lambda: (ws.pin_out_p[:100].tolist(), ws.pin_out_s[:100].tolist())))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
spans = []
for _ in range(100):
    e0.record(); call(0); e1.record(); e1.synchronize(); spans.append(e0.elapsed_time(e1) * 1e3)
spans.sort()
print("  GPU span of both kernels (events)         median %.1f us, min %.1f us" % (spans[50], spans[0]))
cand = torch.tensor(lists[1], device=dev).view(1, -1)
ks = []
for i in range(100):
    c = torch.tensor(lists[i % 64], device=dev).view(1, -1)
    e0.record(); r.score_candidates(Qt, c); e1.record(); e1.synchronize(); ks.append(e0.elapsed_time(e1) * 1e3)
ks.sort()
print("  rerank kernel alone (events, fresh docs)  median %.1f us, min %.1f us  -> %.2f TB/s" % (ks[50], ks[0], 1000 * 180 * idx.element_size() * 128 / ks[50] / 1e6))
