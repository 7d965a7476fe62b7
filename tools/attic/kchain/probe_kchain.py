"""On the GPU box, diagnostic library (MAXSIM_LIB=tools/ab/diag.so): the register-query chain kernel (MAXSIM_KCHAIN=1)
against the LDS-query kernel (MAXSIM_KCHAIN=0) on the default deployment's shape -- dim 768, fp16 index, ragged docs --
scores saved per run (OUT=...) for a bit-for-bit comparison, rerank kernel timed with HIP events: 256 x 1000 batch,
1 x 1000 online call, and a small adversarial case (padding slots, empty docs, q_len, q_mask, 16-bit query)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS[os.environ.get("WL", "dep768")])
dt = os.environ.get("DT", wl["dtype"])
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"])
doclens[5] = 0; doclens[77] = 0
idx = bench.build_index(sum(doclens), wl["h"], dev, 1234, bench.TDT[dt])
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator(device=dev).manual_seed(1)
qdt = bench.TDT[os.environ.get("QDT", wl.get("qdtype", "fp32"))]
Q = F.normalize(torch.randn(256, wl["lq"], wl["h"], generator=g, device=dev), dim=-1).to(qdt)
NB = 6
cands = torch.randint(0, len(doclens), (NB, 256, 1000), generator=g, device=dev)
out = {}
def run(nq, n=12, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); s = r.score_candidates(Q[:nq], cands[i % NB, :nq]); ev[i][1].record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n, s
for nq in (256, 64, 16, 4, 2, 1):
    ms, s = run(nq, n=12 if nq > 1 else 100, w=3 if nq > 1 else 20)
    tok = float(r.d_doclens[cands[:, :nq].flatten()].sum()) / NB
    print(f"KCHAIN={os.environ.get('MAXSIM_KCHAIN', '0')} {dt} q{qdt} nq={nq}: {ms * 1e3:.1f} us = {tok * wl['h'] * idx.element_size() / ms / 1e6:.0f} GB/s", flush=True)
    out[f"b{nq}"] = r.score_candidates(Q[:nq], cands[0, :nq]).cpu()
# adversarial small case
c = cands[1, :7, :333].clone()
c[0, 3] = -1; c[1, 0] = 5; c[2, 10:20] = 77; c[3, 100:] = -1; c[6, :] = -1
ql = torch.tensor([32, 1, 17, 32, 5, 31, 32], dtype=torch.int32, device=dev)
qm = (torch.rand(7, 32, generator=g, device=dev) < 0.7).long(); qm[:, 0] = 1
out["adv_len"] = r.score_candidates(Q[:7], c, q_len=ql).cpu()
out["adv_mask"] = r.score_candidates(Q[:7], c, q_mask=qm).cpu()
out["adv_q8"] = r.score_candidates(Q[:7, :8], c).cpu()
torch.save(out, os.environ["OUT"])
