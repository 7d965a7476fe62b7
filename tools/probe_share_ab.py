"""On the GPU box: A/B of library builds (argv: .so paths) on one rank's share of an N-way step, against the dense rows:
each build runs in its own child process, rounds interleaved.  HIP events around the rerank only, fresh candidates."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    libs = sys.argv[1:]
    for rnd in range(int(os.environ.get("ROUNDS", "2"))):
        for lib in libs:
            env = dict(os.environ, MAXSIM_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
    sys.exit(0)
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd.sharded import shard_candidates
dev = "cuda"
of = int(os.environ.get("OF", "8"))
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
g = torch.Generator(device=dev).manual_seed(0)
nd, nq = 1000000, 256 * of
idx = torch.empty(nd * 180, 128, device=dev, dtype=dt)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
NB = 6
glob = torch.randint(0, of * nd, (NB, nq, 1000), generator=g, device=dev)
dense = torch.randint(0, nd, (NB, 256, 1000), generator=g, device=dev)
def run(step, n=12, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        step(i, ev[i])
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n
def dense_topk(i, e):
    e[0].record(); s = r.score_candidates(Q[:256], dense[i % NB]); e[1].record(); r.topk(s, dense[i % NB], 100)
def filt_cnt_topk(i, e):
    loc, gp, cnt = shard_candidates(glob[i % NB], 3 * nd, 4 * nd, with_counts=True)
    e[0].record(); s = r.score_candidates(Q, loc, cand_count=cnt); e[1].record(); r.topk(s, gp, 100, cnt)
a, b, a2, b2 = run(dense_topk), run(filt_cnt_topk), run(dense_topk), run(filt_cnt_topk)
print("of=%d dense %.3f %.3f  share %.3f %.3f  ratio %.4f" % (of, a, a2, b, b2, (b + b2) / (a + a2)))
