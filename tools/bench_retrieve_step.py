#!/usr/bin/env python3
"""The batched driver's step after the ANN search, stage by stage, for `rocprofv3` (kernel statistics / PMC passes):
256 queries x (32 tokens x faiss_depth 512) synthetic ANN ids over ~HOT docs per query -> distinct pids
(k_unique_pids) -> counted rerank (k_worklist_scan / k_worklist_fill / k_maxsim_stream list form) -> counted top-100
(k_topk / k_topk_count); then one rank's leg of an 8-way doc-sharded step on global candidate lists
(k_shard_candidates -> counted rerank -> counted top-k), and the one-query driver call.  Prints the per-stage times
(HIP events) as one JSON line.    env: NDOCS (1000000), DTYPE (fp16), HOT (1500), ITERS (10)
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (build_index, TDT)
import colbert_amd  # noqa: E402
from colbert_amd.sharded import ShardedRanker  # noqa: E402

dev = torch.device("cuda", 0)
nd = int(os.environ.get("NDOCS", 1000000))
dt = os.environ.get("DTYPE", "fp16")
hot = int(os.environ.get("HOT", 1500))
iters = int(os.environ.get("ITERS", 10))
L, H, LQ, NQ, DEPTH, TOPK = 180, 128, 32, 256, 512, 100
idx = bench.build_index(nd * L, H, dev, 1234, bench.TDT[dt])
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [L] * nd)
g = torch.Generator(device=dev).manual_seed(7)
Q = F.normalize(torch.randn(NQ, LQ, H, generator=g, device=dev), dim=-1)
n = LQ * DEPTH
docs = torch.randint(0, nd, (NQ, hot), generator=g, device=dev)
ids = (docs.gather(1, torch.randint(0, hot, (NQ, n), generator=g, device=dev)) * L
       + torch.randint(0, L, (NQ, n), generator=g, device=dev)).view(NQ, LQ, DEPTH)
keep = torch.rand(NQ, LQ, generator=g, device=dev) > 0.1


def t(f, k=iters):       # after 60 ms of the same calls: the clocks need ~45 ms of load to come back from idle
    return round(bench.warm_then_time(f, k), 4)


cand, cnt = r.embedding_ids_to_pids(ids, trim=False, keep=keep)
sc = r.score_candidates(Q, cand, q_mask=keep, cand_count=cnt)
out = {"shape": f"{NQ} queries x {n} ANN ids, {float(cnt.float().mean()):.0f} distinct candidates per query, {dt} index of {nd} docs"}
out["ids_to_pids_ms"] = t(lambda: r.embedding_ids_to_pids(ids, trim=False, keep=keep))
out["counted_rerank_ms"] = t(lambda: r.score_candidates(Q, cand, q_mask=keep, cand_count=cnt))
out["counted_topk_ms"] = t(lambda: r.topk(sc, cand, TOPK, cnt))
tok = int(r.d_doclens[cand[cand >= 0]].sum().item())
out["rerank_algorithmic_bytes"] = tok * H * idx.element_size()
# one rank's leg of an 8-way doc-sharded step: 2048 queries x 1000 global candidates, ~125 of them local
sh = ShardedRanker(r, 3 * nd, 4 * nd)
Q8 = F.normalize(torch.randn(8 * NQ, LQ, H, generator=g, device=dev), dim=-1)
c8 = torch.randint(0, 8 * nd, (8 * NQ, 1000), generator=g, device=dev)
out["shard_leg_ms"] = t(lambda: sh.local_topk(Q8, c8, TOPK))
from colbert_amd.sharded import shard_candidates  # noqa: E402
out["shard_candidates_ms"] = t(lambda: shard_candidates(c8, sh.lo, sh.hi, with_counts=True))
# the driver serving one query at a time (dense_server_client.py:56-63)
ids1, keep1 = ids[:1], torch.ones(1, LQ, dtype=torch.bool, device=dev)
for _ in range(5):
    colbert_amd.retrieve_batch(r, Q[:1], keep1, topk=TOPK, embedding_ids=ids1)
lat = []
for _ in range(30):
    t0 = time.perf_counter()
    colbert_amd.retrieve_batch(r, Q[:1], keep1, topk=TOPK, embedding_ids=ids1)
    lat.append(time.perf_counter() - t0)
out["one_query_end_to_end_ms"] = round(sorted(lat)[15] * 1e3, 4)
out["one_query_ids_to_pids_ms"] = t(lambda: r.embedding_ids_to_pids(ids1, trim=False, keep=keep1), 30)
print(json.dumps(out), flush=True)
