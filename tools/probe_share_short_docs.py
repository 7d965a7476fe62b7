"""On the GPU box (diagnostic library; MAXSIM_LIST_SHORT=1 lets the list form serve short-doc indexes): counted rows (125
live of 1000 slots, 2048 rows) against the static grid for RAGGED short docs (1..24 tokens), dim 128, DT=fp32|fp16."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = torch.device("cuda", 0)
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
g = torch.Generator(device=dev).manual_seed(1)
gc = torch.Generator().manual_seed(1)
nd = 1000000
doclens = torch.randint(1, 25, (nd,), generator=gc).tolist()
idx = F.normalize(torch.randn(sum(doclens), 128, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
nq, ncand, live = 2048, 1000, 125
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
cand = torch.full((nq, ncand), -1, dtype=torch.int64, device=dev)
cand[:, :live] = torch.randint(0, nd, (nq, live), generator=g, device=dev)
cnt = torch.full((nq,), live, dtype=torch.int32, device=dev)
def run(**kw):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(13)]
    for a, b in ev:
        a.record(); s = r.score_candidates(Q, cand, **kw); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[3:]) / 10, s
a, sa = run()
b, sb = run(cand_count=cnt)
dense = cand[:256 * 0 + 256, :live].repeat(1, 8)          # the same docs per launch as 256 dense rows of 1000
c, _ = run.__call__() if False else (0, 0)
print(f"{os.environ.get('DT', 'fp32')} LIST_SHORT={os.environ.get('MAXSIM_LIST_SHORT', '0')}: static grid {a:.3f} ms | counted rows {b:.3f} ms | scores {'bit-identical' if torch.equal(sa, sb) else 'DIFFER'}")
