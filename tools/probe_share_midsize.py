"""On the GPU box (diagnostic library: MAXSIM_LIST_SLOTS=0 switches the one-round rule of k_worklist_scan off): counted
rows of a doc shard's share (125 live of 1000 slots) for mid-size batches, rows = 96 .. 512, DT=fp32|fp16, dim 128,
180-token docs: the work list a little longer than one round of the resident wave slots."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = torch.device("cuda", 0)
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
h = int(os.environ.get("H", "128"))
g = torch.Generator(device=dev).manual_seed(1)
nd = 300000 if h == 128 else 60000
idx = F.normalize(torch.randn(nd * 180, h, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
out = []
for rows in (96, 128, 144, 160, 192, 224, 256, 320, 384, 512):
    Q = F.normalize(torch.randn(rows, 32, h, generator=g, device=dev), dim=-1)
    cand = torch.full((rows, 1000), -1, dtype=torch.int64, device=dev)
    cand[:, :125] = torch.randint(0, nd, (rows, 125), generator=g, device=dev)
    cnt = torch.full((rows,), 125, dtype=torch.int32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(25)]
    for a, b in ev:
        a.record(); r.score_candidates(Q, cand, cand_count=cnt); b.record()
    torch.cuda.synchronize()
    out.append(f"{rows}:{sum(a.elapsed_time(b) for a, b in ev[5:]) / 20 * 1e3:.0f}")
print(f"h={h} {os.environ.get('DT', 'fp32')} LIST_SLOTS={os.environ.get('MAXSIM_LIST_SLOTS', 'default')} us per launch  " + "  ".join(out))
