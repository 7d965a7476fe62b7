"""On the GPU box, diagnostic library: the default deployment's online call (dim 768, fp16 index, 1 query x 1000 ragged
docs) with the candidates (a) in the caller's order, (b) dealt so that every workgroup's docs hold about the same number
of tiles (sorted by length, dealt boustrophedon over the workgroups) -- for the split forms selected by MAXSIM_SPLIT /
MAXSIM_DPW.  Rerank kernel only, HIP events, 100 launches of 8 different lists."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
wl = dict(bench.WORKLOADS["dep768"])
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"])
idx = bench.build_index(sum(doclens), 768, dev, 1234, torch.float16)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator().manual_seed(1)
Q = F.normalize(torch.randn(1, 32, 768, generator=g), dim=-1).cuda()
dl = torch.tensor(doclens)
per_wg = int(os.environ.get("PER_WG", "4"))
lists = []
for i in range(8):
    c = torch.randint(0, len(doclens), (1000,), generator=g)
    tiles = (dl[c] + 31) // 32
    order = torch.argsort(tiles, descending=True)
    nwg = (1000 + per_wg - 1) // per_wg
    slots = torch.full((nwg, per_wg), -1, dtype=torch.long)
    for k, o in enumerate(order.tolist()):                       # boustrophedon deal: round k % per_wg, direction alternates
        rnd, pos = divmod(k, nwg)
        wg = pos if rnd % 2 == 0 else nwg - 1 - pos
        slots[wg, rnd] = o
    perm = slots.flatten()
    perm = perm[perm >= 0]
    lists.append((c.cuda()[None], c[perm].cuda()[None], perm))
def run(which):
    n, w = 100, 20
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); s = r.score_candidates(Q, lists[i % 8][which]); ev[i][1].record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n * 1e3
a, b = run(0), run(1)
s0 = r.score_candidates(Q, lists[0][0]).cpu()[0]
s1 = r.score_candidates(Q, lists[0][1]).cpu()[0]
same = torch.equal(s0[lists[0][2]], s1)
print(f"SPLIT={os.environ.get('MAXSIM_SPLIT', '-')} DPW={os.environ.get('MAXSIM_DPW', '-')}: caller's order {a:.1f} us | balanced {b:.1f} us | scores {'bit-identical' if same else 'DIFFER'}")
