"""On the GPU box, diagnostic build only (MAXSIM_LIB=tools/ab/diag.so): where one workgroup of the all-pairs kernel
spends a K slice.  The kernel stamps s_memtime at 8 points of 24 consecutive slices (one tile at dim 768) for its first
and last wave; this prints cycles between consecutive points, median over the slices, and the whole tile.
  points: 0 loop top | 1 after the counted vmcnt wait | 2 after the barrier | 3 set-0 fragments in | 4 k-step 0 issued
          5 set-1 fragments in | 6 k-step 1 issued | 7 epilogue done (last slice of a tile only) | 8 next loop top
  (the stamps of a slice are stored right after the next top: that cost sits in the "top+vmcnt" segment)"""
import ctypes, os, sys
import numpy as np
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAXSIM_AP_STAMP_WG", "100")
os.environ.setdefault("MAXSIM_AP_STAMP_G0", "48")
from colbert_amd import _lib
from colbert_amd.scoring import _DT, _MDT
nq, nd, lq, ld, h = [int(x) for x in os.environ.get("SHAPE", "272,544,32,384,768").split(",")]
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(0)
Q = F.normalize(torch.randn(nq, lq, h, generator=g, device="cuda"), dim=-1).to(dt)
D = F.normalize(torch.randn(nd, ld, h, generator=g, device="cuda"), dim=-1).to(dt)
qm = torch.ones(nq, lq, dtype=torch.float32, device="cuda")
dm = torch.ones(nd, ld, dtype=torch.float32, device="cuda")
out = torch.empty(nq, nd, device="cuda")
arg = torch.empty(nq, nd, lq, dtype=torch.int32, device="cuda")
lib = ctypes.CDLL(os.environ["MAXSIM_LIB"])
for _ in range(int(os.environ.get("N", "200"))):   # warm: the clock settles under load
    rc = _lib.lib.maxsim_score_dense_fwd(Q.data_ptr(), D.data_ptr(), qm.data_ptr(), dm.data_ptr(), nq, nd, lq, ld, h, _DT[dt], _MDT[torch.float32], out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
buf = np.zeros(2 * 24 * 9, dtype=np.uint64)
assert lib.maxsim_diag_allpairs_stamps(ctypes.c_void_p(buf.ctypes.data)) == 0
t = buf.reshape(2, 24, 9).astype(np.int64)
names = ["top+vmcnt", "barrier", "frags0 wait", "k-step 0", "frags1 wait", "k-step 1", "loop back"]
for w in range(2):
    print(f"wave {'first' if w == 0 else 'last'}:")
    tw = t[w]
    ok = tw[:, 0] > 0
    if not ok.any():
        print("  (no stamps: is the workgroup / slice window inside the launch?)")
        continue
    seg = np.diff(tw[:, :7], axis=1)             # [24, 6]
    back = np.where(tw[:, 7] > 0, tw[:, 8] - tw[:, 7], tw[:, 8] - tw[:, 6])   # after the epilogue, when there was one
    seg = np.concatenate([seg, back[:, None]], axis=1)
    for k in range(7):
        print(f"  {names[k]:12s} median {int(np.median(seg[ok, k])):6d}  mean {seg[ok, k].mean():8.1f}  max {seg[ok, k].max():6d}")
    period = np.diff(tw[ok, 0])
    print(f"  slice period median {int(np.median(period))}  mean {period.mean():.1f}  (24 slices: {tw[ok, 0][-1] - tw[ok, 0][0]} + last)")
    epi = tw[:, 7] - tw[:, 6]
    print("  epilogue:", [int(e) for e, s7, o in zip(epi, tw[:, 7], ok) if s7 > 0 and o])
    print("  per slice (top->top):", period.tolist())
