"""On the GPU box, timing build (build.sh -DMAXSIM_DIAG -DMAXSIM_STAMP -> MAXSIM_LIB, MAXSIM_KCHAIN=1): where a step of
the register-query chain kernel goes, per compute wave: shader-clock ticks summed over the steps of a launch, divided by
the steps.  Phases: 0 hop read + barrier A | 1 map head + wait for the sub-tile | 2 A-operand reads | 3 MFMAs + requests |
4 hand-over write / reduce | 5 barrier B | 6 loop edge."""
import ctypes, os, sys
import numpy as np
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
from colbert_amd import _lib
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["dep768"])
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"])
idx = bench.build_index(sum(doclens), wl["h"], dev, 1234, torch.float16)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(256, 32, 768, generator=g, device=dev), dim=-1)
_lib.lib.maxsim_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(1 << 17, 8, dtype=torch.int64, device=dev)
_lib.lib.maxsim_diag_set_stamp_buffer(stamps.data_ptr())
for nq in (256, 1):
    for it in range(4):
        cand = torch.randint(0, len(doclens), (nq, 1000), generator=g, device=dev)
        stamps.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.score_candidates(Q[:nq], cand); e1.record(); torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8, 8)            # [workgroup][wave][8]
    st = st[st[:, 0, 7] > 0]
    steps = st[:, :, 7].astype(np.float64)
    print(f"nq={nq}: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(st)} workgroups, {steps[:, 0].mean():.1f} steps each")
    for kb in range(6):
        per = st[:, kb, :7].sum(0) / steps[:, kb].sum()
        print(f"  wave {kb}: ticks per step " + " ".join(f"{x:7.0f}" for x in per) + f"   total {per.sum():7.0f}")
