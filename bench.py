#!/usr/bin/env python3
"""bench.py -- MaxSim rerank throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]

N > 1 needs one process per GPU: launched under torch.distributed.run (RANK / WORLD_SIZE in the environment) this file
is a rank; launched as plain `python bench.py --gpus N` it starts the N ranks itself (a child
`python -m torch.distributed.run ...`, before this process has made any GPU call) and exits with the child's code.

Workload (config.workload): BASELINE.json configs[1] -- a step is one batch of 256 queries x 1000 candidate docs,
32 x 180 tokens, dim 128, fp32 token index resident in HBM, fused gather+MaxSim+top-100.  The synthetic index is
1,000,000 docs (92 GB >> 256 MB Infinity Cache) and every step draws fresh random candidates, so document reads
are real HBM reads.

N > 1 (configs[2]: doc-sharded, weak scaling): every rank holds its own 1M-doc shard (pid range [rank*1M, (rank+1)*1M)),
the batch is 256*N queries, every query has 1000 candidates drawn uniformly over ALL N*1M pids (about 1000/N +- sqrt
per shard) and the same global lists are handed to every rank.  A step is the shipped sharded path:
ShardedRanker.local_topk (maxsim_shard_candidates -> fused rerank -> local top-100 with global pids) ->
exchange_async (ONE RCCL all_gather + per-query merge, on a side stream, overlapping the next batch's rerank).
value = queries of all ranks / max-over-ranks time.  A second, labelled measurement ("stratified") runs the same path
on lists with exactly 1000/N candidates per shard.
"""
import argparse
import json
import os
import socket
import subprocess
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
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
HBM_ACHIEVABLE_GBS = 6300.0  # what a read-only stream reaches on this part (same guide; the kernel's DMA-only ablation agrees)


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
    ap.add_argument("--as-rank", type=int, default=-1,
                    help="diagnostic, with --gpus 1: run ONE rank's share of an --of N job on this GPU (256*N queries, candidates "
                         "drawn over all N shards, shard filter + rerank + local top-k; no exchange unless --force-dist)")
    ap.add_argument("--of", type=int, default=8, help="see --as-rank")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still initialise RCCL (world 1) and run the all_gather + merge leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as a child job.  Nothing in this process has touched the GPU
        # (importing torch does not), and it never will: it only waits and passes the exit code on.
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    # stdout carries exactly ONE line (the JSON): anything native libraries print there (RCCL's start-up banner) is
    # routed to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # MAXSIM_BENCH_ONE_GPU=1: rehearsal of the N > 1 path with every rank on cuda:0 (a one-GPU box)
    one_gpu = bool(os.environ.get("MAXSIM_BENCH_ONE_GPU"))
    dev = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if one_gpu:
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
    sim = args.as_rank >= 0 and world == 1
    job_world, job_rank = (args.of, args.as_rank) if sim else (world, rank)
    lo, hi = job_rank * ndocs, (job_rank + 1) * ndocs
    sharded = ShardedRanker(ranker, lo, hi)        # N > 1: re-buckets the shard by the strides of the whole index
    sharded.force_exchange = args.force_dist

    nq = (args.nq or NQ) * job_world
    ncand_q = args.ncand or NCAND                   # candidates per query, over all shards
    total = args.warmup + args.steps
    gq = torch.Generator(device=dev).manual_seed(1)            # same queries on every rank
    Q = F.normalize(torch.randn(nq, LQ, H, generator=gq, device=dev), dim=-1)
    q_dtype = args.q_dtype or wl.get("qdtype", "fp32")
    Q = Q.to({"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[q_dtype])
    # candidate lists: GLOBAL pids, the same on every rank (same seed); a ring of NB distinct batches (one batch of docs
    # is >= 23 GB of tokens >> the 256 MB Infinity Cache, so re-using a batch NB steps later still reads HBM)
    NB = total if job_world == 1 else min(total, 8)
    gc = torch.Generator(device=dev).manual_seed(2)
    cands = torch.randint(0, job_world * ndocs, (NB, nq, ncand_q), generator=gc, device=dev, dtype=torch.int64)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(total)]
    timed = {"i": 0}
    score_inner = ranker.score_candidates

    def timed_score(Qb, cand_local, **kw):          # HIP events around the rerank kernel only, on the launch stream
        e0, e1 = ev[timed["i"]]
        e0.record()
        out = score_inner(Qb, cand_local, **kw)
        e1.record()
        return out
    sharded.score_fn = timed_score

    def step(i, batches):
        timed["i"] = i
        cand_global = batches[i % batches.size(0)]
        if world == 1 and not use_dist and not sim:
            scores = timed_score(Q, cand_global)
            return ranker.topk(scores, cand_global, min(TOPK, ncand_q))
        # the shipped sharded path: shard filter -> rerank -> local top-k (global pids) ...
        top_p, top_s = sharded.local_topk(Q, cand_global, TOPK)
        # ... then the ONE exchange step (all_gather over xGMI) + the per-query merge on the side stream: batch i's
        # exchange overlaps batch i+1's rerank kernel; every batch is complete before the timed region ends
        if sim and not use_dist:
            return top_p, top_s
        h = sharded.exchange_async(top_p, top_s, TOPK)
        if os.environ.get("MAXSIM_BENCH_NO_PIPELINE"):   # diagnostic: resolve the exchange before the next batch is issued
            h.result()
        return h

    def finish(h):
        return h.result() if hasattr(h, "result") else h

    def run(batches):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; max over ranks."""
        for i in range(args.warmup):
            finish(step(i, batches))
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        sharded.exchange_events = [] if use_dist else None
        t0 = time.perf_counter()
        pending = None
        for i in range(args.warmup, total):
            h = step(i, batches)
            if pending is not None:
                finish(pending)
            pending = h
        finish(pending)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            if one_gpu:
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        kern_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(args.warmup, total)) / args.steps
        xch = sharded.exchange_events or []
        xch_ms = sum(a.elapsed_time(b) for a, b in xch) / len(xch) if xch else 0.0
        return el, kern_ms, xch_ms

    el, kern_ms, xch_ms = run(cands)

    # algorithmic bytes of ONE rerank launch on this rank: doc tokens read once + Q + pid/offset/len + score
    def local_tokens(batches):
        tot, n = 0, 0
        for i in range(args.warmup, total):
            c = batches[i % batches.size(0)]
            loc = c[(c >= lo) & (c < hi)] - lo
            tot += int(ranker.d_doclens[loc].sum().item())
            n += loc.numel()
        return tot / args.steps, n / args.steps
    cand_tokens, docs = local_tokens(cands)
    alg_bytes = int(cand_tokens * H * esize + nq * LQ * H * Q.element_size() + docs * (8 + 12 + 4))
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # second, labelled measurement for N > 1: the same path on stratified lists (exactly 1000/N candidates per shard)
    strat = None
    if world > 1 and ncand_q % world == 0:
        per = ncand_q // world
        gs = torch.Generator(device=dev).manual_seed(3)
        sc = torch.cat([torch.randint(r * ndocs, (r + 1) * ndocs, (NB, nq, per), generator=gs, device=dev, dtype=torch.int64)
                        for r in range(world)], dim=2)
        sc = sc[:, :, torch.randperm(ncand_q, generator=gs, device=dev)]        # shards interleaved within a list
        s_el, s_kern, s_xch = run(sc)
        strat = {"value": round(nq * args.steps / s_el, 2), "ms_per_step": round(s_el / args.steps * 1e3, 4),
                 "kernel_ms_rank0": round(s_kern, 4), "candidates": f"exactly {per} per shard per query"}

    # per-rank figures (every rank contributes one row)
    per_rank = None
    if use_dist:
        row = torch.tensor([kern_ms, xch_ms, docs / nq], dtype=torch.float64, device="cpu" if one_gpu else dev)
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        per_rank = {"rerank_kernel_ms": [round(float(r[0]), 4) for r in rows],
                    "exchange_merge_ms": [round(float(r[1]), 4) for r in rows],
                    "local_candidates_per_query": [round(float(r[2]), 2) for r in rows]}

    # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs of this same command, corrected as
    # MI355X_MICROARCH.md prescribes; summaries committed under profiles/ by tools/summarize_profile.py).  These two
    # fields are REPLAYED from that file (named in pmc_source), not measured in this run.
    traffic = mfma_busy = pmc_source = None
    mode_tag = "" if (args.index_dtype != "fp32" or args.fp32_mode == "exact") else args.fp32_mode
    dt_tag = "f32" if args.index_dtype == "fp32" else args.index_dtype
    default_shape = world == 1 and ndocs == wl["ndocs"] and not (args.lq or args.nq or args.ncand or args.ld or args.q_dtype)
    for tag in ("r02", "r01"):
        pmc = os.path.join(ROOT, "profiles", f"{tag}_{args.workload}_{dt_tag}{mode_tag}_pmc.json")
        if not (default_shape and os.path.exists(pmc)):
            continue
        try:
            for k, v in json.load(open(pmc)).items():
                if "k_maxsim" in k and "hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)" in v:
                    traffic = int(v["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] + v.get("hbm_write_bytes_per_launch(WRITE_SIZE*1024)", 0))
                    mfma_busy = v.get("mfma_util(SQ_VALU_MFMA_BUSY_CYCLES/1024 / (GRBM_GUI_ACTIVE/8))")
                    pmc_source = os.path.relpath(pmc, ROOT)
        except (OSError, ValueError):
            traffic = None
        if traffic is not None:
            break

    if rank == 0:
        res = {
            "metric": "queries/sec MaxSim rerank, 32q x 180d tokens, dim=128, 1000 docs/query" if args.workload == "c2"
                      else f"queries/sec MaxSim rerank, workload {args.workload}",
            "value": round(nq * args.steps / el, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {args.nq or NQ} queries/GPU x {ncand_q} candidates/query, {LQ}x"
                                   f"{'~120 (8..180 ragged)' if wl['ragged'] else LD} tokens, dim {H}, "
                                   f"{args.index_dtype} index of {ndocs} docs/GPU in HBM, fused rerank + top-{TOPK}"
                                   + (f", doc-sharded x{world}: candidates uniform over all {world * ndocs} pids, shard filter + "
                                      f"local top-{TOPK} + RCCL all_gather + merge" if world > 1 else "")
                                   + (f", SIMULATED rank {job_rank} of {job_world} (one rank's share of the job, no exchange)" if sim else ""),
                       "queries_per_step": nq, "candidates_per_query": ncand_q, "docs_per_gpu": ndocs,
                       "index_dtype": args.index_dtype, "q_dtype": q_dtype, "fp32_mode": args.fp32_mode, "parallelism": f"doc-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4), "achievable_peak": HBM_ACHIEVABLE_GBS,
                         "traffic": traffic, "pmc_source": pmc_source,
                         "kernel": ("k_maxsim_stream" if H == 128 else "k_maxsim_stream_bigh" if H % 128 == 0 and H <= 1024 else "k_maxsim_generic") if LQ <= 32 else "k_maxsim_generic", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # matrix-pipe view of the same launch (PMC pass, profiles/): busy fraction of the MFMA pipe
                         "mfma_busy_frac": None if mfma_busy is None else round(mfma_busy, 3),
                         "mfma_tflops": round(2.0 * LQ * H * cand_tokens / (kern_ms * 1e-3) / 1e12, 1)},
        }
        if use_dist:
            res["n_ranks_seen"] = dist.get_world_size()
            res["backend"] = dist.get_backend()
            res["per_rank"] = per_rank
        if strat is not None:
            res["stratified"] = strat
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            res["single_query"] = single_query_probe(ranker, Q, cands, H, LQ, esize)
        if world == 1 and args.workload == "c2" and not args.no_cpu_baseline:
            res["training_form"] = training_form_probe(dev)
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            res["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def single_query_probe(ranker, Q, cands, H, LQ, esize):
    """The reference's online call: ONE query x 1000 candidates through rank_forward (faiss_indexers.py:234), python
    list in, python lists out, host-synchronous -- latency, not throughput.  `gpu_span_ms` is the time between two HIP
    events recorded on the launch stream right before and after the call (both kernels + the gap between them);
    `host_ms` = end-to-end minus that span (the events themselves add a few us to the span: the kernel's own duration is
    in profiles/r02_single_query_*)."""
    Q1 = Q[:1].float().permute(0, 2, 1)     # [1, h, Lq]: the permuted VIEW of a [1, Lq, h] tensor, as faiss_indexers.py:232-233 hands it over
    out = {"call": "rank_forward(Q[1,h,Lq], 1000 pids, depth=100) -> python lists"}
    lat, span = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes
    from colbert_amd import _lib
    Qt = Q1.permute(0, 2, 1).contiguous()   # (what rank_forward makes of it: the original layout, no copy)
    for it in range(160):
        pids1 = cands[it % cands.size(0), it % cands.size(1)].tolist()        # fresh docs every call: HBM, not cache
        if it % 2 == 0:
            t1 = time.perf_counter()
            ranker.rank_forward(Q1, pids1, depth=TOPK)
            lat.append(time.perf_counter() - t1)
        else:
            # the same library call without its wait, between two HIP events on the launch stream: GPU time of the call
            ws = ranker._tls.ws
            ws.pin_in[:len(pids1)] = pids1
            st = torch.cuda.current_stream().cuda_stream
            e0.record()
            _lib.lib.maxsim_rank_forward(ctypes.byref(ranker._iv), Qt.data_ptr(), 0, LQ, ws.in_ptr, len(pids1), TOPK,
                                         ws.scratch_ptr, ws.out_p_ptr, ws.out_s_ptr, None, 0, st)
            e1.record()
            e1.synchronize()
            span.append(e0.elapsed_time(e1))
    lat, span = sorted(lat[10:]), sorted(span[10:])
    med, gspan = lat[len(lat) // 2] * 1e3, span[len(span) // 2]
    ntok = len(pids1) * int(ranker.d_doclens[0].item())
    out.update({"median_ms": round(med, 4), "min_ms": round(lat[0] * 1e3, 4), "gpu_span_ms": round(gspan, 4),
                "host_ms": round(max(med - gspan, 0.0), 4), "queries_per_s_sequential": round(1e3 / med, 1),
                "algorithmic_GBps_over_gpu_span": round((ntok * H * esize + LQ * H * 4) / (gspan * 1e-3) / 1e9, 1),
                "kernel_profile": "profiles/r02_single_query_summary.json"})
    # 16 queries per launch (a small server batch): the rerank kernel alone, 20 launches back to back between two events
    ks = []
    for rep in range(5):
        e0.record()
        for it in range(20):
            ranker.score_candidates(Q[:16], cands[(rep * 20 + it) % cands.size(0), 16 * (it % 8):16 * (it % 8) + 16])
        e1.record()
        e1.synchronize()
        ks.append(e0.elapsed_time(e1) / 20)
    ks = sorted(ks[1:])
    b16 = 16 * cands.size(2) * int(ranker.d_doclens[0].item()) * H * esize
    out["batch16"] = {"kernel_ms": round(ks[len(ks) // 2], 4), "algorithmic_GBps": round(b16 / (ks[len(ks) // 2] * 1e-3) / 1e9, 1),
                      "how": "20 launches of 16 queries x 1000 candidates back to back, HIP events around the 20"}
    return out


def training_form_probe(dev):
    """The operator's second caller: BaseModel.score on the gathered training batch (colbert_model.py:87-90),
    every query against every doc, at the reference's step -- Q 272 x 32 x 768, D 544 x 384 x 768 bf16 (dense.yaml:6-8) --
    through maxsim_score_dense_fwd (scores + arg-max for the backward).  Matrix-bound, unlike the rerank path: priced
    against the dense bf16 MFMA peak.  Not the headline metric; one line so that the number is in the bench record."""
    import torch.nn.functional as F
    from colbert_amd import _lib
    from colbert_amd.scoring import _DT, _MDT
    nq, nd, lq, ld, h = 272, 544, 32, 384, 768
    g = torch.Generator(device=dev).manual_seed(3)
    Qt = F.normalize(torch.randn(nq, lq, h, generator=g, device=dev), dim=-1).bfloat16()
    Dt = F.normalize(torch.randn(nd, ld, h, generator=g, device=dev), dim=-1).bfloat16()
    qm = torch.ones(nq, lq, dtype=torch.float32, device=dev)
    dm = (torch.arange(ld, device=dev)[None, :] < torch.randint(ld // 4, ld + 1, (nd, 1), generator=g, device=dev)).float()
    out = torch.empty(nq, nd, device=dev)
    arg = torch.empty(nq, nd, lq, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = _lib.lib.maxsim_score_dense_fwd(Qt.data_ptr(), Dt.data_ptr(), qm.data_ptr(), dm.data_ptr(), nq, nd, lq, ld, h,
                                             _DT[torch.bfloat16], _MDT[torch.float32], out.data_ptr(), arg.data_ptr(), st)
        assert rc == 0, rc
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    flop = 2.0 * nq * nd * lq * ld * h
    return {"op": "maxsim_score_dense_fwd (scores + arg-max), Q 272x32x768 x D 544x384x768, bf16, prefix d_mask",
            "kernel": "k_maxsim_allpairs" if _lib.lib.maxsim_score_dense_kernel(nq, nd, lq, ld, h, _DT[torch.bfloat16], _MDT[torch.float32]) == 1 else "k_maxsim_stream_bigh",
            "forward_ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1), "peak_tflops_dense_bf16": 2500.0,
            "frac": round(flop / ms / 1e9 / 2500.0, 4), "how": f"{n} launches back to back between two HIP events",
            "profile": "profiles/r02_allpairs_kernel_stats.csv, profiles/r02_allpairs_pmc.json"}


if __name__ == "__main__":
    main()
