#!/usr/bin/env python3
"""GPU box: how long does each rerank launch of a workload take, step by step, from an IDLE GPU?

The driver's bench run is 5 warm-up + 20 timed steps (105 ms of C2); a 2500-step loop of the same launches runs 4 % faster
per launch with the package-power limiter active 99 % of the time (tools/run_throttle_watch.sh).  This probe builds the
workload's index, lets the GPU idle for --idle seconds, then issues --steps launches back to back with a HIP event pair around
each and prints the durations: where in the burst the launches are slow, and how long the ramp lasts.

    python tools/probe_step_timeline.py [--workload c2] [--steps 600] [--idle 3] [--repeat 2]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import torch.nn.functional as F                 # noqa: E402
import bench                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--idle", type=float, default=3.0)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--fp32-mode", default="exact")
    args = ap.parse_args()
    import colbert_amd
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS[args.workload]
    LQ, LD, H = wl["lq"], wl["ld"], wl["h"]
    dtype = bench.TDT[wl["dtype"]]
    ndocs = wl["ndocs"]
    doclens = bench.make_doclens(wl, ndocs, LD, 0)
    idx = bench.build_index(sum(doclens), H, dev, 1234, dtype)
    ranker = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens, fp32_mode=args.fp32_mode)
    gq = torch.Generator(device=dev).manual_seed(1)
    Q = F.normalize(torch.randn(bench.NQ, LQ, H, generator=gq, device=dev), dim=-1).to(bench.TDT[wl.get("qdtype", "fp32")])
    NB = 24
    cands = bench.draw_candidates(ndocs, (NB, bench.NQ, bench.NCAND), torch.Generator(device=dev).manual_seed(2), dev)
    tok = float(sum(int(ranker.d_doclens[cands[b].reshape(-1)].sum().item()) for b in range(NB))) / NB
    alg = bench.algorithmic_bytes(tok, bench.NQ * bench.NCAND, bench.NQ, LQ, H, idx.element_size(), Q.element_size())
    out = {"workload": args.workload, "algorithmic_bytes_per_launch": alg, "runs": []}
    for r in range(args.repeat):
        torch.cuda.synchronize()
        time.sleep(args.idle)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(args.steps)]
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()
            ranker.score_candidates(Q, cands[i % NB])
            ev[i][1].record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms = [ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps)]
        start = [ev[0][0].elapsed_time(ev[i][0]) for i in range(args.steps)]

        def frac(a, b):
            seg = ms[a:b]
            return round(alg / (sum(seg) / len(seg) * 1e-3) / 8e12, 4) if seg else None
        marks = [i for i in (0, 1, 2, 3, 4, 5, 7, 10, 15, 20, 25, 30, 40, 50, 75, 100, 150, 200, 300, 400, 500, 750, 1000, 1500, 2000) if i < args.steps]
        run = {"idle_s": args.idle, "wall_s": round(wall, 3),
               "ms_at_step": {str(i): round(ms[i], 4) for i in marks},
               "t_ms_at_step": {str(i): round(start[i], 1) for i in marks},
               "frac_steps_5_25": frac(5, 25), "frac_steps_25_100": frac(25, 100), "frac_steps_100_300": frac(100, 300),
               "frac_last_100": frac(args.steps - 100, args.steps)}
        out["runs"].append(run)
        print(json.dumps(run), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"step_timeline_{args.workload}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
