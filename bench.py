#!/usr/bin/env python3
"""bench.py -- MaxSim rerank throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Workload (config.workload): BASELINE.json configs[1] -- a step is one batch of 256 queries x 1000 candidate docs,
32 x 180 tokens, dim 128, fp32 token index resident in HBM, fused gather+MaxSim+top-100.  The synthetic index is
1,000,000 docs (92 GB >> 256 MB Infinity Cache) and every step draws fresh random candidates, so document reads
are real HBM reads.  N > 1 (doc-sharded, weak scaling): each rank holds its own 1M-doc shard, the batch is
256*N queries, each query's 1000 candidates are stratified 1000/N per shard, local top-100 -> one RCCL
all_gather -> per-query merge.  value = queries of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LQ, LD, H, NQ, NCAND, TOPK = 32, 180, 128, 256, 1000, 100

# extra workloads (reported next to the headline one, never as `value` of the default run):
#   ragged : doclens ~ clipped N(120, 40) in [8, 180]     -- exercises packed tiles + 0-floor buckets
#   c4     : multi-view, 8 viewer tokens per doc, Lq = 8 (dense.yaml q_view = d_view)
#   c5     : bf16, dim 768, 256 tokens per doc, 200k docs
WORKLOADS = {
    "c2": dict(lq=32, ld=180, h=128, ndocs=1_000_000, ragged=False, dtype="fp32"),
    "ragged": dict(lq=32, ld=180, h=128, ndocs=1_000_000, ragged=True, dtype="fp32"),
    "c4": dict(lq=8, ld=8, h=128, ndocs=4_000_000, ragged=False, dtype="fp32"),
    "c5": dict(lq=32, ld=256, h=768, ndocs=200_000, ragged=False, dtype="bf16", qdtype="bf16"),
}
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def build_index(ntok, h, dev, seed, dtype):
    """F.normalize(randn) token embeddings, generated on-device in chunks (encoder output contract, BaseModel.py:26)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    idx = torch.empty(ntok, h, dtype=dtype, device=dev)
    chunk = max(1, (1 << 28) // h)
    for s in range(0, ntok, chunk):
        e = min(s + chunk, ntok)
        idx[s:e] = F.normalize(torch.randn(e - s, h, generator=gen, device=dev), dim=-1).to(dtype)
    return idx


def host_cores():
    """Cores this process may actually use: the cgroup CPU quota (the GPU box gives 16 of 256), else affinity."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds=12.0):
    """The oracle restatement of the reference's score() (BaseModel.py:39-46) on the host cores: the reference's
    unit of work, 1 query x 1000 docs per call (colbert_ranker.py:111-112), fp32."""
    from oracle.maxsim_oracle import ref_score
    cores = host_cores()
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(0)
    Q = F.normalize(torch.randn(1, LQ, H, generator=gen), dim=-1)
    D = F.normalize(torch.randn(NCAND, LD, H, generator=gen), dim=-1)
    qm, dm = torch.ones(1, LQ, dtype=torch.long), torch.ones(NCAND, LD, dtype=torch.long)
    for _ in range(2):
        ref_score(Q, D, qm, dm)
    n, t0 = 0, time.perf_counter()
    while True:
        ref_score(Q, D, qm, dm)
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= 10) or el >= 3 * seconds:
            break
    out = {"value": round(n / el, 3), "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": f"{n} calls of 1 query x {NCAND} docs x ({LQ}x{LD}) tokens dim {H} fp32, torch CPU, {cores} threads"}
    # the whole reference-shaped rank_forward (colbert_ranker.py:75-137: CPU gather from the fp16 strided view, cast,
    # mask, score, sort) on a small host-resident index -- what one query costs the reference before PCIe
    from oracle.maxsim_oracle import RefRanker
    nd = 4000
    part = F.normalize(torch.randn(nd * LD, H, generator=gen), dim=-1).half()
    rr = RefRanker([part], [[LD] * nd], dim=H)
    Qr = Q.permute(0, 2, 1).contiguous()
    pids = torch.randperm(nd, generator=gen)[:NCAND].tolist()
    rr.rank_forward(Qr, pids, depth=TOPK)
    m, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 4.0 or m < 5:
        rr.rank_forward(Qr, pids, depth=TOPK)
        m += 1
    out["rank_forward"] = {"value": round(m / (time.perf_counter() - t1), 3), "unit": "queries/s",
                           "sample": f"{m} calls of the restated rank_forward, 1 query x {NCAND} of {nd} docs, fp16 CPU index"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--ndocs", type=int, default=0, help="docs per GPU shard (0 = the workload's default)")
    ap.add_argument("--index-dtype", default="", choices=["", "fp32", "fp16", "bf16"])
    ap.add_argument("--lq", type=int, default=0, help="query tokens (0 = the workload's default)")
    ap.add_argument("--ld", type=int, default=0, help="tokens per doc (0 = the workload's default; diagnostic)")
    ap.add_argument("--nq", type=int, default=0, help="queries per GPU per step (0 = 256, the metric's batch; diagnostic)")
    ap.add_argument("--q-dtype", default="", choices=["", "fp32", "fp16", "bf16"],
                    help="element type the queries are handed over in (default: fp32; c5: bf16)")
    ap.add_argument("--fp32-mode", default="exact", choices=["exact", "fast", "bf16x3"],
                    help="fp32 index only: exact f32 MFMA (default) or the split-fp16 fast mode")
    ap.add_argument("--ncand", type=int, default=0,
                    help="candidates per query on each GPU (0 = 1000 / N; diagnostic: --nq 2048 --ncand 125 is one rank's share of N = 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the host-side legs (cpu_baseline and the single-query latency probe): profiling runs")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still initialise RCCL (world 1) and run the all_gather + merge leg")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): anything native libraries print there (RCCL's start-up banner) is
    # routed to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    # MAXSIM_BENCH_ONE_GPU=1: rehearsal of the N > 1 path with every rank on cuda:0 (a one-GPU box)
    dev = torch.device("cuda", 0 if os.environ.get("MAXSIM_BENCH_ONE_GPU") else local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if os.environ.get("MAXSIM_BENCH_ONE_GPU"):
            dist.init_process_group("gloo", rank=rank, world_size=world)     # rehearsal only (see above)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import colbert_amd
    from colbert_amd.sharded import ShardedRanker

    wl = WORKLOADS[args.workload]
    LQ, LD, H = (args.lq or wl["lq"]), (args.ld or wl["ld"]), wl["h"]
    args.index_dtype = args.index_dtype or wl["dtype"]
    dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[args.index_dtype]
    esize = torch.empty(0, dtype=dtype).element_size()
    ndocs = args.ndocs or wl["ndocs"]
    if wl["ragged"]:
        g = torch.Generator().manual_seed(99 + rank)
        doclens = (torch.randn(ndocs, generator=g) * 40 + 120).round().clamp(8, LD).long().tolist()
    else:
        # uniform docs: strides = [LD], one bucket, no padding floor
        doclens = [LD] * ndocs
    ntok = sum(doclens)
    idx = build_index(ntok, H, dev, 1234 + rank, dtype)
    ranker = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens, fp32_mode=args.fp32_mode)
    lo, hi = rank * ndocs, (rank + 1) * ndocs
    sharded = ShardedRanker(ranker, lo, hi)
    sharded.force_exchange = args.force_dist

    nq = (args.nq or NQ) * world
    per = args.ncand or NCAND // world
    assert args.ncand or per * world == NCAND
    total = args.warmup + args.steps
    gq = torch.Generator(device=dev).manual_seed(1)            # same queries on every rank
    Q = F.normalize(torch.randn(nq, LQ, H, generator=gq, device=dev), dim=-1)
    q_dtype = args.q_dtype or wl.get("qdtype", "fp32")
    Q = Q.to({"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[q_dtype])
    gc = torch.Generator(device=dev).manual_seed(2 + rank)     # this shard's candidates, fresh per step
    cands = torch.randint(lo, hi, (total, nq, per), generator=gc, device=dev, dtype=torch.int64)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(total)]

    def step(i):
        cand_global = cands[i]
        cand_local, inr = cand_global - lo, None
        ev[i][0].record()
        scores = ranker.score_candidates(Q, cand_local)
        ev[i][1].record()
        top_p, top_s = ranker.topk(scores, cand_global, TOPK if per >= TOPK else per)
        if not use_dist:
            return top_p, top_s
        # the ONE exchange step (all_gather over xGMI) + the per-query merge run on the side stream: batch i's exchange
        # overlaps batch i+1's rerank kernel; every batch is complete before the timed region ends (result() + sync)
        h = sharded.exchange_async(top_p, top_s, TOPK)
        if os.environ.get("MAXSIM_BENCH_NO_PIPELINE"):   # diagnostic: resolve the exchange before the next batch is issued
            h.result()
        return h

    def finish(h):
        return h.result() if use_dist else h

    for i in range(args.warmup):
        finish(step(i))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for i in range(args.warmup, total):
        h = step(i)
        if pending is not None:
            out = finish(pending)
        pending = h
    out = finish(pending)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    kern_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(args.warmup, total)) / args.steps
    # algorithmic bytes of ONE rerank launch on this rank: doc tokens read once + Q + pid/offset/len + score
    docs = nq * per
    cand_tokens = int(ranker.d_doclens[(cands[args.warmup:] - lo).reshape(-1)].sum().item()) / args.steps
    alg_bytes = int(cand_tokens * H * esize + nq * LQ * H * Q.element_size() + docs * (8 + 12 + 4))
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs of this same command, corrected as
    # MI355X_MICROARCH.md prescribes; summary committed under profiles/ by tools/summarize_profile.py)
    traffic = mfma_busy = None
    pmc = os.path.join(ROOT, "profiles", f"r01_{args.workload}_{'f32' if args.index_dtype == 'fp32' else args.index_dtype}_pmc.json")
    if world == 1 and ndocs == wl["ndocs"] and not (args.lq or args.nq or args.ncand or args.ld) and os.path.exists(pmc):
        try:
            for k, v in json.load(open(pmc)).items():
                if "k_maxsim" in k and "hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)" in v:
                    traffic = int(v["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] + v.get("hbm_write_bytes_per_launch(WRITE_SIZE*1024)", 0))
                    mfma_busy = v.get("mfma_util(SQ_VALU_MFMA_BUSY_CYCLES/1024 / (GRBM_GUI_ACTIVE/8))")
        except (OSError, ValueError):
            traffic = None

    if rank == 0:
        res = {
            "metric": "queries/sec MaxSim rerank, 32q x 180d tokens, dim=128, 1000 docs/query" if args.workload == "c2"
                      else f"queries/sec MaxSim rerank, workload {args.workload}",
            "value": round(nq * args.steps / el, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {args.nq or NQ} queries/GPU x {NCAND} candidates/query, {LQ}x"
                                   f"{'~120 (8..180 ragged)' if wl['ragged'] else LD} tokens, dim {H}, "
                                   f"{args.index_dtype} index of {ndocs} docs/GPU in HBM, fused rerank + top-{TOPK}",
                       "queries_per_step": nq, "candidates_per_query": per * world, "docs_per_gpu": ndocs,
                       "index_dtype": args.index_dtype, "q_dtype": q_dtype, "fp32_mode": args.fp32_mode, "parallelism": f"doc-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": ("k_maxsim_stream" if H == 128 else "k_maxsim_stream_bigh" if H % 128 == 0 and H <= 1024 else "k_maxsim_generic") if LQ <= 32 else "k_maxsim_generic", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # matrix-pipe view of the same launch (PMC pass, profiles/): busy fraction of the MFMA pipe
                         "mfma_busy_frac": None if mfma_busy is None else round(mfma_busy, 3),
                         "mfma_tflops": round(2.0 * LQ * H * cand_tokens / (kern_ms * 1e-3) / 1e12, 1)},
        }
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            # the reference's online call: ONE query x 1000 candidates through rank_forward (faiss_indexers.py:234),
            # python lists in and out, host-synchronous -- latency, not throughput
            Q1 = Q[:1].permute(0, 2, 1).contiguous()               # [1, h, Lq] as ColbertRetriever.search hands it over
            pids1 = (cands[0, 0] - lo).tolist()
            lat = []
            for _ in range(60):
                t1 = time.perf_counter()
                ranker.rank_forward(Q1, pids1, depth=TOPK)
                lat.append(time.perf_counter() - t1)
            lat = sorted(lat[10:])
            res["single_query"] = {"call": "rank_forward(Q[1,h,Lq], 1000 pids, depth=100) -> python lists",
                                   "median_ms": round(lat[len(lat) // 2] * 1e3, 4), "min_ms": round(lat[0] * 1e3, 4)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            res["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
