"""On the GPU box: one workload of bench.py's table (WL=c2|ragged|c4|c5|dep768, DT= index dtype override, UNIFORM=1: the
workload's mean doc length for every doc), 256 queries x 1000 candidates, rerank kernel only (HIP events), (a) static grid,
(b) counted rows with every row full (the work-list form), for the library selected by MAXSIM_LIB / its knobs."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
name = os.environ.get("WL", "ragged")
wl = dict(bench.WORKLOADS[name])
if os.environ.get("RAGGED"):          # mean,sd,lo,hi of the clipped normal
    wl["ragged"] = tuple(int(x) for x in os.environ["RAGGED"].split(","))
if os.environ.get("UNIFORM"):
    wl["ld"], wl["ragged"] = wl["ragged"][0], None
dt = os.environ.get("DT", wl["dtype"])
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"])
idx = bench.build_index(sum(doclens), wl["h"], dev, 1234, bench.TDT[dt])
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens, fp32_mode=os.environ.get("MODE", "exact"))
nq = int(os.environ.get("NQ", "256"))
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(nq, wl["lq"], wl["h"], generator=g, device=dev), dim=-1).to(bench.TDT[os.environ.get("QDT", wl.get("qdtype", "fp32"))])
NB = 6
cands = torch.randint(0, len(doclens), (NB, nq, 1000), generator=g, device=dev)
full = torch.full((nq,), 1000, dtype=torch.int32, device=dev)
def run(f, n=12, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); s = f(i); ev[i][1].record(); r.topk(s, cands[i % NB], 100)
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n
tok = sum(int(r.d_doclens[cands[i % NB].flatten()].sum()) for i in range(3, 15)) / 12
byts = tok * wl["h"] * idx.element_size()
a = run(lambda i: r.score_candidates(Q, cands[i % NB]))
out = "%s %s%s knobs[%s]: static %.4f ms = %.0f GB/s (%.3f of 8 TB/s)" % (
    name, dt, " uniform" if os.environ.get("UNIFORM") else (" N" + os.environ["RAGGED"] if os.environ.get("RAGGED") else ""), " ".join(f"{k[7:]}={v}" for k, v in os.environ.items() if k.startswith("MAXSIM_") and k != "MAXSIM_LIB"),
    a, byts / a / 1e6, byts / a / 1e6 / 8000)
if wl["h"] == 128 and not os.environ.get("NOLIST"):
    b = run(lambda i: r.score_candidates(Q, cands[i % NB], cand_count=full))
    a2 = run(lambda i: r.score_candidates(Q, cands[i % NB]))
    out += " | list %.4f ms (%.3f) | static again %.4f" % (b, byts / b / 1e6 / 8000, a2)
print(out)
