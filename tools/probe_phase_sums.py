"""On the GPU box, with a timing build (MAXSIM_OUT=tools/ab/stamp.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG -DMAXSIM_STAMP
-DMAXSIM_STAMP2; MAXSIM_LIB=tools/ab/stamp.so): where the waves of a FULL rerank launch (256 queries x 1000 candidates) spend
their time, summed over each wave's tiles: waiting for the tile to arrive | reading it into registers | requesting the next
one (fill_tile + the LDS-DMA instructions) | contraction + reduce.  Warm clocks: WARM launches first, the last one stamped.
env: DT=fp32|fp16, LD (180), RAGGED=1 (N(120,40)), WARM (40)."""
import ctypes, os, sys
import numpy as np
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd import _lib
dev = "cuda"
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
nd, nq, ncand = 1000000, 256, 1000
g = torch.Generator(device=dev).manual_seed(0)
if os.environ.get("RAGGED"):
    dl = torch.randn(nd, generator=g, device=dev).mul(40).add(120).round().clamp(20, 180).long().tolist()
else:
    dl = [int(os.environ.get("LD", 180))] * nd
ntok = sum(dl)
idx = torch.empty(ntok, 128, device=dev, dtype=dt)
for s in range(0, ntok, 1 << 22):
    e = min(s + (1 << 22), ntok)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, dl)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
_lib.lib.maxsim_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(1 << 17, 8, dtype=torch.int64, device=dev)
cands = [torch.randint(0, nd, (nq, ncand), generator=g, device=dev) for _ in range(8)]
for i in range(int(os.environ.get("WARM", 40))):
    r.score_candidates(Q, cands[i % 8])
_lib.lib.maxsim_diag_set_stamp_buffer(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); r.score_candidates(Q, cands[0]); e1.record(); torch.cuda.synchronize()
_lib.lib.maxsim_diag_set_stamp_buffer(None)
st = stamps.cpu().numpy()
st = st[st[:, 0] != 0]
tiles = st[:, 5].astype(np.float64)
ph = st[:, 7].astype(np.uint64)
parts = np.stack([(ph >> np.uint64(16 * k)) & np.uint64(0xffff) for k in range(4)], 1).astype(np.float64) * 0.01   # us
span = (st[:, 6] - st[:, 0]) * 0.01
if os.environ.get("RAW"):
    print([hex(int(x) & 0xffffffffffffffff) for x in st[:6, 7]])
print(f"launch {e0.elapsed_time(e1) * 1e3:.0f} us, {len(st)} waves, {tiles.mean():.1f} tiles per wave, wave span {span.mean():.1f} us (p10 {np.percentile(span, 10):.1f}, p90 {np.percentile(span, 90):.1f})")
d = np.diff(st[:, [0, 1, 2, 3, 4, 6]].astype(np.float64), axis=1) * 0.01
print("  wave phases, us (mean): entry -> descriptors %.2f | -> first fetches issued + query in registers %.2f | -> first tile arrived %.2f | -> last tile reduced %.2f | -> exit %.2f" % tuple(d.mean(0)))
names = ["arrival wait", "operand reads", "next fetch issue", "contraction + reduce"]
tot = parts.sum(1)
for k, n in enumerate(names):
    print(f"  {n:22s} {parts[:, k].mean():7.2f} us per wave = {parts[:, k].sum() / tiles.sum():.3f} us per tile  ({100 * parts[:, k].sum() / tot.sum():.1f} % of the loop)")
print(f"  loop total {tot.mean():.2f} us per wave = {tot.sum() / tiles.sum():.3f} us per tile; saturated counters: {(parts >= 655.3).sum()}")
