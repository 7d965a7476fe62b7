"""On the GPU box: token-row indices past 2^31 -- an fp16 index of 2.3 G rows x 16 dims (74 GB), docs of 50..150 rows, candidates
from its LAST 300 docs (first rows > 2^31), scores against the float64 closed form on the gathered rows; ids -> pids for
token rows > 2^31 against searchsorted.  One-off check of the 32-bit row arithmetic (n_tokens <= 2^32 - 1 is the contract)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from oracle.maxsim_oracle import ragged_scores_f64
dev = "cuda"
h = 16
g = torch.Generator().manual_seed(5)
ndocs = 23_000_000
doclens_t = torch.randint(50, 151, (ndocs,), generator=g)
ntok = int(doclens_t.sum())
print("rows", ntok, "> 2^31:", ntok > 2 ** 31, "< 2^32:", ntok < 2 ** 32, flush=True)
idx = torch.empty(ntok, h, dtype=torch.float16, device=dev)
gd = torch.Generator(device=dev).manual_seed(7)
step = 1 << 26
for s in range(0, ntok, step):
    e = min(s + step, ntok)
    idx[s:e] = F.normalize(torch.randn(e - s, h, generator=gd, device=dev), dim=-1).half()
doclens = doclens_t.tolist()
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
nq, ncand, Lq = 3, 40, 32
Q = F.normalize(torch.randn(nq, Lq, h, generator=g), dim=-1)
cand = torch.randint(ndocs - 300, ndocs, (nq, ncand), generator=g)
rows_lo = int(r.doclens_pfxsum[ndocs - 300])
assert rows_lo > 2 ** 31
got = r.score_candidates(Q.cuda(), cand.cuda()).cpu()
tail = idx[rows_lo:].cpu()
offs, pad = r.doclens_pfxsum, r.d_pad_len.cpu()
offs_tail = {p: int(offs[p]) - rows_lo for p in range(ndocs - 300, ndocs)}
class _O:                      # (offsets / doclens / pad_len looked up by pid without materialising 23 M-entry python lists twice)
    def __init__(s, f): s.f = f
    def __getitem__(s, p): return s.f(p)
ok = True
for q in range(nq):
    exp = ragged_scores_f64(tail, _O(lambda p: doclens[p]), _O(lambda p: offs_tail[p]), _O(lambda p: int(pad[p])), Q[q], cand[q].tolist())
    d = np.abs(got[q].numpy() - exp).max()
    print("query", q, "max |d| =", d, flush=True)
    ok = ok and d <= 1e-3
# ids -> pids for rows past 2^31
ids = torch.randint(rows_lo, ntok, (2, 4096), generator=g)
c, n = r.embedding_ids_to_pids(ids.cuda(), trim=False)
for q in range(2):
    exp = sorted(set((torch.searchsorted(offs, ids[q], right=True) - 1).tolist()))
    ok = ok and c[q, :int(n[q])].tolist() == exp
print("ALL OK" if ok else "MISMATCH")
