"""On the GPU box: where the time of one rank_forward(Q, 1000 pids, depth=100) call goes (host conversions, launches, syncs)."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev="cuda"
gen=torch.Generator(device=dev).manual_seed(0)
nd=200000
idx=F.normalize(torch.randn(nd*180,128,generator=gen,device=dev),dim=-1)
r=colbert_amd.ColbertRanker.from_device_tensor(idx,[180]*nd)
Q=F.normalize(torch.randn(1,32,128,generator=gen,device=dev),dim=-1).permute(0,2,1).contiguous()
pids=torch.randperm(nd)[:1000].tolist()
def T(f,n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
print("rank_forward total        %.1f us" % T(lambda: r.rank_forward(Q,pids,depth=100)))
print("torch.tensor(pids)        %.1f us" % T(lambda: torch.tensor(pids)))
pt=torch.tensor(pids)
print("pids.to(device) view      %.1f us" % T(lambda: pt.to(dev,torch.int64).view(1,-1)))
cand=pt.to(dev).view(1,-1)
Qt=Q.permute(0,2,1)
print("score_candidates          %.1f us" % T(lambda: r.score_candidates(Qt,cand)))
sc=r.score_candidates(Qt,cand)
print("topk                      %.1f us" % T(lambda: r.topk(sc,cand,100)))
tp,ts=r.topk(sc,cand,100)
print("2x tolist                 %.1f us" % T(lambda: (tp[0].tolist(), ts[0].tolist())))
print("Q.to+permute+contig       %.1f us" % T(lambda: Q.to(dev).permute(0,2,1).contiguous()))
