"""On the GPU box: the all-pairs (training-form) forward alone, for rocprofv3: `score` with arg-max at the reference's
training step shape (Q 272x32x768, D 544x384x768 bf16), N launches back to back."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd import _lib
from colbert_amd.scoring import _DT, _MDT
nq, nd, lq, ld, h = [int(x) for x in os.environ.get("SHAPE", "272,544,32,384,768").split(",")]
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("DTYPE", "bf16")]
g = torch.Generator(device="cuda").manual_seed(0)
Q = F.normalize(torch.randn(nq, lq, h, generator=g, device="cuda"), dim=-1).to(dt)
D = F.normalize(torch.randn(nd, ld, h, generator=g, device="cuda"), dim=-1).to(dt)
qm = torch.ones(nq, lq, dtype=torch.float32, device="cuda")     # float32 masks: what colbert_amd.score hands over
dl = torch.randint(ld // 4, ld + 1, (nd, 1), generator=g, device="cuda")
dm = (torch.arange(ld, device="cuda").unsqueeze(0) < dl).float()
out = torch.empty(nq, nd, device="cuda")
arg = torch.empty(nq, nd, lq, dtype=torch.int32, device="cuda")
am = int(os.environ.get("ARGMAX", "1"))
def launch():
    st = torch.cuda.current_stream().cuda_stream
    if am:
        rc = _lib.lib.maxsim_score_dense_fwd(Q.data_ptr(), D.data_ptr(), qm.data_ptr(), dm.data_ptr(), nq, nd, lq, ld, h, _DT[dt], _MDT[torch.float32], out.data_ptr(), arg.data_ptr(), st)
    else:
        rc = _lib.lib.maxsim_score_dense(Q.data_ptr(), D.data_ptr(), qm.data_ptr(), dm.data_ptr(), nq, nd, lq, ld, h, _DT[dt], _MDT[torch.float32], out.data_ptr(), st)
    assert rc == 0, rc
# WARM_MS of the same launches first: the clocks need ~45 ms of load to come back from idle (tools/probe_step_timeline.py) --
# round 5's earlier A/B runs of this file (3 warm-ups, 20 launches) sat on that ramp
for _ in range(3): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(os.environ.get("N", "40"))
fl = 2.0 * nq * nd * lq * ld * h
e0.record()
for _ in range(10): launch()
e1.record(); e1.synchronize()
cold = e0.elapsed_time(e1) / 10
for _ in range(int(float(os.environ.get("WARM_MS", "150")) / cold)): launch()
e0.record()
for _ in range(n): launch()
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"all-pairs fwd argmax={am} {nq}x{nd} {lq}x{ld} dim {h}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  = {fl / ms / 1e9 / 2500 * 100:.1f} % of 2.5 PFLOP/s"
      f"   (launches 4-13 after the set-up: {cold:.3f} ms)")
