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
ShardedRanker.local_topk (maxsim_shard_candidates -> counted rerank from a device-built work list -> local top-100 with
global pids) -> exchange_async (ONE RCCL all_gather + per-query merge, on a side stream, overlapping the next batch's
rerank).  value = queries of all ranks / max-over-ranks time.  Two more, labelled measurements of the same path: "stratified"
(lists with exactly 1000/N candidates per shard) and "per_shard_1000" (SURVEY 8d's other weak-scaling variant: lists of
1000 x N pids, 1000 on every shard).  Candidates are drawn WITHOUT replacement within a list.
`sharded_self_check_ok` says whether the from-files self-check (load_shard + rank_forward + retrieve_batch over the job's own
process group == the unsharded HIP ranker) held on every rank.

The default N = 1 run also reports, in the same JSON line and each driver-timed in this process:
  sharded_share   one rank's share of an N = 2 / 4 / 8 step on this GPU (256*N queries, ~1000/N live candidates per row):
                  what every rank of that job does before the exchange -- the expected weak-scaling curve's compute side
  other_workloads BASELINE configs[3], [4], the reference's fp16 storage dtype, ragged doclens, the reference's default
                  deployment shape (dim 768 fp16 ragged), its multi-view deployments on the fp16 index (mv128: 8 x 8 tokens
                  dim 128; mv768: 16 x 16 tokens dim 768) and the opt-in bf16x3 contraction of the fp32 index
  single_query    the reference's online call (one rank_forward), training_form (the operator's second caller),
  cpu_baseline    the oracle on the host cores (never the product path), roofline.read_ceiling (measured on this box).
and measures roofline.traffic / mfma_busy_frac itself: three `rocprofv3 --pmc` child runs come first, before this process
touches the GPU (live_pmc: FETCH_SIZE over the headline and every other_workloads entry, WRITE_SIZE and the MFMA-busy
counters on the headline -- the same pass gives roofline.effective_clock_GHz; --no-pmc skips them and replays profiles/).
roofline.power is a best-effort sysfs sample of board power / cap / shader clock over the headline's steps.

stdout carries ONE compact JSON line (the contract's fields + one numeric row per extra measurement, roofline last); the
full record of the run -- every entry with its shape, method and sources -- is written to gpurun_out/bench_details.json
(`details_file`).
"""
import argparse
import gc
import ctypes
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

NQ, NCAND, TOPK = 256, 1000, 100

# ragged = (mean, sd, lo, hi) of a clipped normal; None = every doc has `ld` tokens
#   ragged : doclens ~ clipped N(120, 40) in [8, 180]     -- exercises packed tiles + 0-floor buckets
#   c4     : multi-view, 8 viewer tokens per doc, Lq = 8 (dense.yaml q_view = d_view)
#   c5     : bf16, dim 768, 256 tokens per doc, 200k docs
#   mv128 / mv768 : multi-view on the fp16 index: 8 x 8 tokens dim 128 (2 KiB per doc) / 16 x 16 tokens dim 768 (24 KiB per doc)
#   dep768 : the reference's default deployment (proj_conf/dense.yaml:6-8 dim 768, doc_maxlen 384; encoder.py:175 fp16 index):
#            ragged doclens ~ clipped N(200, 80) in [8, 384]
WORKLOADS = {
    "c2": dict(lq=32, ld=180, h=128, ndocs=1_000_000, ragged=None, dtype="fp32"),
    "ragged": dict(lq=32, ld=180, h=128, ndocs=1_000_000, ragged=(120, 40, 8, 180), dtype="fp32"),
    "c4": dict(lq=8, ld=8, h=128, ndocs=4_000_000, ragged=None, dtype="fp32"),
    "c5": dict(lq=32, ld=256, h=768, ndocs=200_000, ragged=None, dtype="bf16", qdtype="bf16"),
    "dep768": dict(lq=32, ld=384, h=768, ndocs=200_000, ragged=(200, 80, 8, 384), dtype="fp16"),
    # the reference's multi-view deployments in its own storage dtype (colbert_ranker.py:62: fp16; BaseModel.py:21-24 keeps the
    # first `view` tokens): C4's shape on the fp16 index, and the yaml's defaults (dense.yaml:8,29-32: dim 768, q_view = d_view = 16)
    "mv128": dict(lq=8, ld=8, h=128, ndocs=4_000_000, ragged=None, dtype="fp16"),
    "mv768": dict(lq=16, ld=16, h=768, ndocs=1_000_000, ragged=None, dtype="fp16"),
}
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
TDT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def build_index(ntok, h, dev, seed, dtype):
    """F.normalize(randn) token embeddings, generated on-device in chunks (encoder output contract, BaseModel.py:26)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    idx = torch.empty(ntok, h, dtype=dtype, device=dev)
    chunk = max(1, (1 << 28) // h)
    for s in range(0, ntok, chunk):
        e = min(s + chunk, ntok)
        idx[s:e] = F.normalize(torch.randn(e - s, h, generator=gen, device=dev), dim=-1).to(dtype)
    return idx


def draw_candidates(n_docs, shape, gen, dev, lo=0):
    """Candidate pids drawn uniformly WITHOUT replacement within each list, `shape` = [..., ncand], in
    [lo, lo + n_docs): a uniform draw whose repeated entries (about ncand^2 / 2 n_docs per list) are re-drawn until none is left."""
    assert shape[-1] <= n_docs
    ncand = shape[-1]
    nrows = 1
    for d in shape[:-1]:
        nrows *= d
    if 4 * ncand > n_docs:      # a large share of the range per list (tests, tiny indexes): the first ncand of a random permutation
        flat = torch.empty(nrows, ncand, dtype=torch.int64, device=dev)
        step = max(1, (1 << 24) // n_docs)
        for a in range(0, nrows, step):
            b = min(a + step, nrows)
            flat[a:b] = torch.rand(b - a, n_docs, generator=gen, device=dev).argsort(dim=-1)[:, :ncand]
        c = flat.view(shape)
        return c + lo if lo else c
    c = torch.randint(0, n_docs, shape, generator=gen, device=dev, dtype=torch.int64)
    flat = c.view(-1, ncand)
    for _ in range(1000):       # (each pass re-draws the repeated entries: a handful after the first pass, none after two or three)
        srt, idx = flat.sort(dim=-1)
        dup = torch.zeros_like(srt, dtype=torch.bool)
        dup[:, 1:] = srt[:, 1:] == srt[:, :-1]
        n = int(dup.sum().item())
        if n == 0:
            break
        rows = dup.nonzero(as_tuple=True)
        flat[rows[0], idx[rows]] = torch.randint(0, n_docs, (n,), generator=gen, device=dev, dtype=torch.int64)
    else:
        raise RuntimeError("draw_candidates: repeated entries left after 1000 passes")
    return c + lo if lo else c


def make_doclens(wl, ndocs, ld, rank=0):
    if wl["ragged"] is None:
        return [ld] * ndocs                         # uniform docs: strides = [LD], one bucket, no padding floor
    mean, sd, lo, hi = wl["ragged"]
    g = torch.Generator().manual_seed(99 + rank)
    return (torch.randn(ndocs, generator=g) * sd + mean).round().clamp(lo, min(hi, ld)).long().tolist()


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


def kernel_name(h, lq):
    if lq > 32 and h != 128:
        return "k_maxsim_generic"
    return "k_maxsim_stream" if h == 128 else "k_maxsim_stream_bigh" if 16 <= h <= 1024 else "k_maxsim_generic"


def pmc_lookup(workload, index_dtype, fp32_mode, suffix=""):
    """HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs of this same command, corrected as
    MI355X_MICROARCH.md prescribes; summaries committed under profiles/ by tools/summarize_profile.py).  These fields are
    REPLAYED from that file (named in pmc_source), not measured in this run."""
    mode_tag = "" if (index_dtype != "fp32" or fp32_mode == "exact") else fp32_mode
    dt_tag = "f32" if index_dtype == "fp32" else index_dtype
    for tag in ("r05", "r04", "r03", "r02", "r01"):
        pmc = os.path.join(ROOT, "profiles", f"{tag}_{workload}{suffix}_{dt_tag}{mode_tag}_pmc.json")
        if not os.path.exists(pmc):
            continue
        try:
            for k, v in json.load(open(pmc)).items():
                if "k_maxsim" in k and "hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)" in v:
                    traffic = int(v["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] + v.get("hbm_write_bytes_per_launch(WRITE_SIZE*1024)", 0))
                    return traffic, v.get("mfma_util(SQ_VALU_MFMA_BUSY_CYCLES/1024 / (GRBM_GUI_ACTIVE/8))"), os.path.relpath(pmc, ROOT)
        except (OSError, ValueError):
            pass
    return None, None, None


PMC_PASSES = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]))
# the sweep child's workloads, in launch order: (key, workload, extra_workload keywords).  "headline" is the run's own workload.
PMC_SWEEP = (("headline", "c2", {}), ("c2_bf16x3", "c2", dict(fp32_mode="bf16x3", reuse_prev=True)),
             ("c2_fp16", "c2", dict(index_dtype="fp16")), ("ragged", "ragged", {}), ("ragged_bf16x3", "ragged", dict(fp32_mode="bf16x3", reuse_prev=True)),
             ("ragged_fp16", "ragged", dict(index_dtype="fp16")),
             ("c4", "c4", {}), ("c5", "c5", {}),
             ("dep768", "dep768", {}), ("mv128", "mv128", {}), ("mv768", "mv768", {}))
PMC_SWEEP_LAUNCHES = 4


def pmc_sweep_child(manifest_path):
    """Child of live_pmc's FETCH_SIZE pass: 1 warm-up + 3 rerank launches of the headline workload and of every
    other_workloads entry, in PMC_SWEEP's order, nothing else from this library on the GPU; what each group of launches
    should have read (its own algorithmic bytes) goes to the manifest."""
    import colbert_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    manifest, prev = [], None
    for key, name, kw in PMC_SWEEP:
        wl = WORKLOADS[name]
        lq, ld, h = wl["lq"], wl["ld"], wl["h"]
        index_dtype = kw.get("index_dtype") or wl["dtype"]
        dtype = TDT[index_dtype]
        if kw.get("reuse_prev") and prev is not None:
            idx, doclens = prev
        else:
            prev = idx = None
            torch.cuda.empty_cache()
            doclens = make_doclens(wl, wl["ndocs"], ld)
            idx = build_index(sum(doclens), h, dev, 1234, dtype)
        ranker = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens, fp32_mode=kw.get("fp32_mode", "exact"))
        gq = torch.Generator(device=dev).manual_seed(1)
        Q = F.normalize(torch.randn(NQ, lq, h, generator=gq, device=dev), dim=-1).to(TDT[wl.get("qdtype", "fp32")])
        gen_c = torch.Generator(device=dev).manual_seed(2)
        cands = draw_candidates(len(doclens), (PMC_SWEEP_LAUNCHES, NQ, NCAND), gen_c, dev)
        for i in range(PMC_SWEEP_LAUNCHES):
            ranker.score_candidates(Q, cands[i])
        torch.cuda.synchronize()
        cand_tokens, docs = live_tokens(ranker, cands, 0, len(doclens), 1, PMC_SWEEP_LAUNCHES - 1)
        manifest.append({"key": key, "launches": PMC_SWEEP_LAUNCHES,
                         "algorithmic_bytes_per_launch": algorithmic_bytes(cand_tokens, docs, NQ, lq, h, idx.element_size(), Q.element_size())})
        prev = (idx, doclens)
        ranker = None
    with open(manifest_path, "w") as f:
        json.dump(manifest, f)


def live_pmc(budget_s=180.0):
    """HBM counters measured IN this run: before this process touches the GPU, three short child runs of this same file
    under `rocprofv3 --pmc` -- one counter set per pass, no trace domains, as MI355X_MICROARCH.md's HBM section prescribes:
      fetch  FETCH_SIZE over 1 warm-up + 3 rerank launches of the headline workload AND of every other_workloads entry
             (pmc_sweep_child); read bytes = 2 * FETCH_SIZE * 1024 (the guide's gfx950 correction for wide coalesced streams)
      write  WRITE_SIZE, headline workload (12 warm-up + 3 counted launches): write bytes = WRITE_SIZE * 1024
      sq     SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, headline workload (12 + 3 launches)
    Per-launch means over the 3 launches after the warm-up, from the passes' counter_collection.csv.
    Returns (fields, None) or (None, reason): the caller then replays profiles/."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any("rocprof" in v.lower() for k, v in os.environ.items() if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")):
        return None, "this process is itself running under a profiler: no nested passes"
    tmp = tempfile.mkdtemp(prefix="maxsim_pmc_", dir="/tmp")
    manifest_path = os.path.join(tmp, "sweep.json")
    env = dict(os.environ, TMPDIR="/tmp", MAXSIM_BENCH_PMC_CHILD="1", MAXSIM_PMC_MANIFEST=manifest_path)
    t_end = time.time() + budget_s
    means, sweep = {}, None
    try:
        for name, counters in PMC_PASSES:
            left = t_end - time.time()
            if left < 20.0:
                return None, f"time budget of {budget_s:.0f} s spent before the {name} pass"
            out = os.path.join(tmp, name)
            # (write / sq passes: 12 warm-up launches, so that the 3 counted ones run at the clock a loop of them holds, not on the ramp
            # out of idle -- the first launches after a pause take 8.4, 5.1, 4.6, 4.4, 4.3 ms where the tenth takes 4.1)
            child = ["--pmc-sweep"] if name == "fetch" else ["--steps", "3", "--warmup", "12", "--no-cpu-baseline"]
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__)] + child
            try:
                with open(os.path.join(tmp, name + ".log"), "w") as log:
                    rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=left).returncode
            except subprocess.TimeoutExpired:
                return None, f"{name} pass exceeded the time budget"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, f"{name} pass: rc {rc}, {len(files)} counter files"
            rows = [r for r in csv.DictReader(open(max(files, key=os.path.getmtime))) if "k_maxsim_stream" in r["Kernel_Name"]]
            for c in counters:
                seq = [float(r["Counter_Value"]) for r in sorted((r for r in rows if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))]
                if name == "fetch":
                    manifest = json.load(open(manifest_path))
                    if len(seq) != sum(m["launches"] for m in manifest):
                        return None, f"fetch pass: {len(seq)} rerank launches counted, {sum(m['launches'] for m in manifest)} expected"
                    sweep, at = {}, 0
                    for m in manifest:
                        v = seq[at + 1:at + m["launches"]]          # (the first launch of a group is its warm-up)
                        at += m["launches"]
                        rd = 2.0 * (sum(v) / len(v)) * 1024.0
                        sweep[m["key"]] = {"hbm_read_bytes": int(rd), "launches_sampled": len(v),
                                           "read_over_algorithmic": round(rd / m["algorithmic_bytes_per_launch"], 4)}
                else:
                    if len(seq) < 2:
                        return None, f"{name} pass: no {c} rows for the rerank kernel"
                    means[c] = sum(seq[-3:]) / len(seq[-3:])          # the 3 launches after the child's warm-ups
                    if c == "GRBM_GUI_ACTIVE":
                        # clock the chip held DURING those launches: cycles (summed over the 8 XCDs) / 8 / the same dispatches'
                        # own duration (MI355X_MICROARCH.md, DVFS give-back; a profiled pass runs a few % below an un-profiled one)
                        rr = sorted((r for r in rows if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))[-3:]
                        try:
                            ghz = [float(r["Counter_Value"]) / 8.0 / (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rr]
                            means["effective_clock_GHz"] = sum(ghz) / len(ghz)
                        except (KeyError, ValueError, ZeroDivisionError):
                            pass
    except (OSError, ValueError, KeyError) as e:
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd, wr = float(sweep["headline"]["hbm_read_bytes"]), means["WRITE_SIZE"] * 1024.0
    src = ("live: `rocprofv3 --pmc` child runs of this file, started by this bench.py before its own timed region (FETCH_SIZE over "
           "every workload's launches | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on the headline workload)")
    return {"traffic": int(rd + wr), "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
            "mfma_busy_frac": round((means["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (means["GRBM_GUI_ACTIVE"] / 8.0), 3),
            "effective_clock_GHz": None if "effective_clock_GHz" not in means else round(means["effective_clock_GHz"], 3),
            "launches_sampled": sweep["headline"]["launches_sampled"], "pmc_source": src, "sweep": sweep}, None


class PowerSampler:
    """Best-effort board power and shader clock of one GPU while a measurement runs: a thread that reads the amdgpu hwmon
    files of the device (`power1_input` / `power1_average` in microwatts, `power1_cap`, `freq1_input` in Hz) every few
    milliseconds.  Nothing is installed, nothing is written; where the files are missing or unreadable the sample is
    {"available": False}.  It lets a reader of the bench line tell a slow BOX (power cap reached at a lower clock) from a
    regression.  sysfs clocks are context, not the test: roofline.effective_clock_GHz (PMC) is the in-kernel figure."""

    def __init__(self, dev_index, period_s=0.004, firmware=True):
        import glob
        import threading
        self.period, self.hw, self.samples, self._stop, self._th = period_s, None, [], threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for c in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(c)).startswith(want):
                    hw = glob.glob(os.path.join(c, "hwmon", "hwmon*"))
                    if hw:
                        self.hw = hw[0]
                        break
        except Exception:       # best effort by contract
            self.hw = None
        self.pfile = None
        if self.hw:
            for f in ("power1_input", "power1_average"):
                if os.access(os.path.join(self.hw, f), os.R_OK):
                    self.pfile = os.path.join(self.hw, f)
                    break
        # the firmware's own counters (tools/smi_sample.c, built by __graft_entry__.build(); absent -> skipped): one sample on
        # either side of the region -> energy taken, share of the region with the package-power / thermal limiters active
        self.smi, self.smi_dev, self.smi_a, self.smi_b = None, -1, None, None
        try:
            if not firmware:      # (N > 1: one process per GPU -- the SMI library's cross-process mutex stays out of a multi-rank run)
                raise RuntimeError("firmware counters not requested")
            import ctypes
            lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libsmi_sample.so"))
            lib.smi_open.argtypes, lib.smi_open.restype = [ctypes.c_uint32, ctypes.c_uint32], ctypes.c_int
            lib.smi_read.argtypes, lib.smi_read.restype = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)], ctypes.c_int
            pr = torch.cuda.get_device_properties(dev_index)
            got = []
            # (the library's start-up takes a cross-process lock: in a daemon thread with a deadline, so that a stale lock on
            # the box costs this run its firmware sample, not its bench line)
            th = threading.Thread(target=lambda: got.append(lib.smi_open(pr.pci_domain_id, pr.pci_bus_id)), daemon=True)
            th.start()
            th.join(timeout=8.0)
            self.smi_dev = got[0] if got else -1
            if self.smi_dev >= 0:
                self.smi = lib
        except Exception:       # best effort by contract
            self.smi = None

    def _smi_read(self):
        if self.smi is None:
            return None
        import ctypes
        buf = (ctypes.c_double * 16)()
        if self.smi.smi_read(self.smi_dev, buf) != 0:
            return None
        return time.perf_counter(), list(buf)

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _loop(self):
        ffile = os.path.join(self.hw, "freq1_input")
        while not self._stop.is_set():
            self.samples.append((self._rd(self.pfile), self._rd(ffile)))
            self._stop.wait(self.period)

    def __enter__(self):
        self.smi_a = self._smi_read()
        if self.pfile:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self.smi_b = self._smi_read()
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=1.0)
        return False

    def _smi_summary(self):
        """Differences of the firmware's counters over the region (None where the table does not carry a field)."""
        if not self.smi_a or not self.smi_b:
            return None
        (ta, a), (tb, b) = self.smi_a, self.smi_b
        out = {"interval_s": round(tb - ta, 4), "source": "SMI gpu_metrics table (tools/smi_sample.c), one sample before and one after the region"}
        cyc = b[0] - a[0] if a[0] >= 0 and b[0] >= 0 else 0
        out["accumulation_cycles"] = int(cyc)
        if cyc > 0:
            def share(i):
                return None if a[i] < 0 or b[i] < 0 else round((b[i] - a[i]) / cyc, 4)
            out["ppt_limiter_active_share"] = share(1)          # PVIOL: the package-power limiter held the clock down
            th = [share(i) for i in (2, 3, 4, 5)]
            out["thermal_limiters_active_share"] = None if all(x is None for x in th) else max(x for x in th if x is not None)
        if a[6] >= 0 and b[6] >= 0 and b[11] > 0 and tb > ta:
            out["energy_J"] = round((b[6] - a[6]) * b[11] * 1e-6, 2)
            out["energy_avg_W"] = round(out["energy_J"] / (tb - ta), 1)
        out["socket_power_W_after"] = None if b[7] < 0 else b[7]
        out["sclk_target_MHz_after"] = None if b[8] < 0 else round(b[8])
        out["hotspot_C_after"], out["mem_C_after"] = (None if b[9] < 0 else b[9]), (None if b[10] < 0 else b[10])
        return out

    def summary(self):
        pw = [p for p, _ in self.samples if p is not None]
        fq = [f for _, f in self.samples if f is not None]
        smi = self._smi_summary()
        if not pw:
            return {"available": smi is not None, "firmware": smi} if smi else {"available": False}
        cap = self._rd(os.path.join(self.hw, "power1_cap"))
        # (power1_input is the firmware's filtered average: over a region of ~0.1 s it is still climbing towards what a long
        # loop of the same launches shows -- 1400 W after about a second; `firmware` below is exact over the region)
        out = {"available": True, "avg_W": round(sum(pw) / len(pw) / 1e6, 1), "max_W": round(max(pw) / 1e6, 1),
               "cap_W": None if cap is None else round(cap / 1e6, 1), "sclk_MHz_avg": round(sum(fq) / len(fq) / 1e6) if fq else None,
               "samples": len(pw), "source": "sysfs hwmon power1_input (a filtered average) / freq1_input, sampled over the warm-up + timed steps"}
        if smi:
            out["firmware"] = smi
        return out


PREWARM_MS = 60.0     # the labelled extra measurements (never the headline): GPU time of untimed launches in front of a region


def timed_steps(step, warmup, steps, barrier=None, prewarm_ms=0.0):
    """W untimed warm-up steps, then EXACTLY K timed steps between (barrier +) synchronize on both sides.  `step(i)`
    issues step i and returns a handle (or None); a step's handle is resolved while the next step is in flight.
    prewarm_ms > 0 (other_workloads / sharded_share entries only; the headline runs exactly its W warm-ups): step 0 is
    repeated in front of the warm-ups until the GPU has been busy that long -- the clocks need ~45 ms of load to come back
    from idle (tools/probe_step_timeline.py), and W = 5 launches of a 0.1-2 ms kernel are over long before that."""
    def finish(h):
        return h.result() if hasattr(h, "result") else h
    # the interpreter's cyclic garbage collector stays out of the timed region: a generation-2 pass walks the workloads'
    # million-element doclens lists (30-45 ms -- 4 ms per step of a 10-step region: the "host gap" of round 3's builder
    # record, reproduced in round 4 as 4.6 vs 1.95 ms and 4.8 vs 0.23 ms between two regions of the same workload).
    # The collection runs BEFORE the warm-up steps, not between them and the timed ones: while the host collects, the GPU
    # idles, its clocks fall within milliseconds, and the next ~8 launches climb back (round 5, kernel ms of consecutive C2
    # steps: warm-up 8.4 5.1 4.6 4.4 4.3 | collect | timed 5.1 5.1 4.6 4.4 4.3 4.2 4.2 4.1 4.07 ... -- the warm-up undone,
    # 4 % on a 20-step region); with nothing but the synchronize between them the timed steps continue where the warm-up ended
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        if prewarm_ms > 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            finish(step(0))
            e1.record()
            e1.synchronize()
            for _ in range(min(3000, int(prewarm_ms / max(e0.elapsed_time(e1), 0.02)))):
                finish(step(0))
        for i in range(warmup):
            finish(step(i))
        if barrier:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = None
        for i in range(warmup, warmup + steps):
            h = step(i)
            if pending is not None:
                finish(pending)
            pending = h
        finish(pending)
        torch.cuda.synchronize()
        if barrier:
            barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    return el


def warm_then_time(f, n, warm_ms=PREWARM_MS):
    """ms per call of `f` over n calls back to back between ONE pair of HIP events, after warm_ms of GPU time of the same
    calls (3 calls are timed first to size the warm-up): a measurement that starts on an idle GPU runs on the clock ramp."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        f()
    e1.record()
    e1.synchronize()
    for _ in range(min(3000, int(warm_ms / max(e0.elapsed_time(e1) / 3, 0.02)))):
        f()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def live_tokens(ranker, batches, lo, hi, warmup, steps):
    """Mean (candidate tokens, candidate docs) of the timed steps that fall in this rank's pid range."""
    tot, n = 0, 0
    for i in range(warmup, warmup + steps):
        c = batches[i % batches.size(0)]
        loc = c[(c >= lo) & (c < hi)] - lo
        tot += int(ranker.d_doclens[loc].sum().item())
        n += loc.numel()
    return tot / steps, n / steps


def algorithmic_bytes(cand_tokens, docs, nq, lq, h, esize, qsize):
    """doc tokens read once + Q + pid (8) + offset/len (12) + score (4) per candidate."""
    return int(cand_tokens * h * esize + nq * lq * h * qsize + docs * (8 + 12 + 4))


def bench_rows(ranker, Q, cands, warmup, steps, k, sharded=None):
    """One GPU, no exchange: K steps of rerank + top-k over `cands[i]`.  sharded = a ShardedRanker: the step is its
    local_topk (shard filter + counted rerank + counted top-k) on GLOBAL candidate lists.  Returns (wall s, kernel ms):
    kernel ms = HIP events around the rerank launch(es) on the launch stream (torch's current stream)."""
    total = warmup + steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(total)]
    cur = {"i": 0}
    inner = ranker.score_candidates

    def timed_score(Qb, cand_local, **kw):
        e0, e1 = ev[cur["i"]]
        e0.record()
        out = inner(Qb, cand_local, **kw)
        e1.record()
        return out

    if sharded is not None:
        sharded.score_fn = timed_score

    def step(i):
        cur["i"] = i
        c = cands[i % cands.size(0)]
        if sharded is not None:
            return sharded.local_topk(Q, c, k)
        return ranker.topk(timed_score(Q, c), c, min(k, c.size(1)))

    el = timed_steps(step, warmup, steps, prewarm_ms=PREWARM_MS)
    kern_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(warmup, total)) / steps
    return el, kern_ms


def read_ceiling(buf):
    """The read rate this box's memory system delivers to the kernels' own fetch pattern (non-temporal LDS-DMA into
    per-wave rings, nothing consumed): maxsim_hbm_read_probe over the first 16 GiB of `buf`, best of its three ring
    shapes, HIP events around 3 launches each (SURVEY 8d asks for a measured ceiling next to the spec peak)."""
    from colbert_amd import _lib
    nbytes = min(buf.numel() * buf.element_size(), 16 << 30)
    st = torch.cuda.current_stream().cuda_stream
    got = ctypes.c_int64(0)
    best, per_variant = 0.0, {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for variant, name in ((0, "1x16KiB"), (1, "2x8KiB"), (2, "2x16KiB")):
        rc = _lib.lib.maxsim_hbm_read_probe(buf.data_ptr(), nbytes, variant, ctypes.addressof(got), st)
        assert rc == 0, rc
        e0.record()
        for _ in range(3):
            _lib.lib.maxsim_hbm_read_probe(buf.data_ptr(), nbytes, variant, None, st)
        e1.record()
        e1.synchronize()
        gbs = got.value * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        per_variant[name] = round(gbs, 1)
        best = max(best, gbs)
    return {"GBps": round(best, 1), "by_ring_shape": per_variant, "bytes_per_launch": got.value,
            "how": "maxsim_hbm_read_probe: every wave streams 512 KiB through nt LDS-DMA into its LDS ring, nothing consumed; "
                   "3 launches between two HIP events per shape"}


def cpu_baseline(seconds=12.0):
    """The oracle restatement of the reference's score() (BaseModel.py:39-46) on the host cores: the reference's
    unit of work, 1 query x 1000 docs per call (colbert_ranker.py:111-112), fp32; plus C1 (1 x 10, BASELINE configs[0])
    and the whole reference-shaped rank_forward."""
    from oracle.maxsim_oracle import RefRanker, ref_score
    LQ, LD, H = 32, 180, 128
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
    # C1 (BASELINE.json configs[0]): 1 query x 10 docs, the reference's own CPU-runnable case
    D10, dm10 = D[:10].contiguous(), dm[:10].contiguous()
    for _ in range(5):
        ref_score(Q, D10, qm, dm10)
    m, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0 or m < 20:
        ref_score(Q, D10, qm, dm10)
        m += 1
    el1 = time.perf_counter() - t1
    out["c1"] = {"value": round(m / el1, 2), "unit": "queries/s", "ms_per_call": round(el1 / m * 1e3, 4),
                 "sample": f"{m} calls of 1 query x 10 docs x ({LQ}x{LD}) tokens dim {H} fp32 (C1), torch CPU, {cores} threads"}
    # the whole reference-shaped rank_forward (colbert_ranker.py:75-137: CPU gather from the fp16 strided view, cast,
    # mask, score, sort) on a small host-resident index -- what one query costs the reference before PCIe
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


def roofline_entry(kern_ms, alg_bytes, cand_tokens, lq, h, workload, index_dtype, fp32_mode, default_shape, suffix=""):
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = mfma_busy = pmc_source = None
    if default_shape:
        traffic, mfma_busy, pmc_source = pmc_lookup(workload, index_dtype, fp32_mode, suffix)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "pmc_source": pmc_source,
            "kernel": kernel_name(h, lq), "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
            # matrix-pipe view of the same launch (PMC pass, profiles/): busy fraction of the MFMA pipe
            "mfma_busy_frac": None if mfma_busy is None else round(mfma_busy, 3),
            "mfma_tflops": round(2.0 * lq * h * cand_tokens / (kern_ms * 1e-3) / 1e12, 1)}


def extra_workload(colbert_amd, name, dev, steps, warmup, index_dtype=None, fp32_mode="exact", reuse=None, label=None,
                   online_call=False):
    """One more BASELINE / deployment workload on this GPU, 256 queries x 1000 candidates per step, the index built once
    (or `reuse` = (idx, doclens) of a workload that is still resident).  Not `value`: a labelled entry of other_workloads."""
    wl = WORKLOADS[name]
    lq, ld, h = wl["lq"], wl["ld"], wl["h"]
    index_dtype = index_dtype or wl["dtype"]
    dtype = TDT[index_dtype]
    esize = torch.empty(0, dtype=dtype).element_size()
    if reuse is None:
        doclens = make_doclens(wl, wl["ndocs"], ld)
        idx = build_index(sum(doclens), h, dev, 1234, dtype)
    else:
        idx, doclens = reuse
    ranker = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens, fp32_mode=fp32_mode)
    gq = torch.Generator(device=dev).manual_seed(1)
    Q = F.normalize(torch.randn(NQ, lq, h, generator=gq, device=dev), dim=-1).to(TDT[wl.get("qdtype", "fp32")])
    total = warmup + steps
    gen_c = torch.Generator(device=dev).manual_seed(2)
    cands = draw_candidates(len(doclens), (total, NQ, NCAND), gen_c, dev)
    # two timed regions; the SECOND is reported (the first doubles as a long warm-up), both are recorded: round 3's builder
    # record showed 6.99 ms wall against 2.91 ms of kernel once -- a garbage-collector pass inside the region (timed_steps
    # now keeps it out)
    torch.cuda.synchronize()
    runs = [bench_rows(ranker, Q, cands, warmup, steps, TOPK) for _ in range(2)]
    el, kern_ms = runs[1]
    cand_tokens, docs = live_tokens(ranker, cands, 0, len(doclens), warmup, steps)
    alg = algorithmic_bytes(cand_tokens, docs, NQ, lq, h, esize, Q.element_size())
    rf = roofline_entry(kern_ms, alg, cand_tokens, lq, h, name, index_dtype, fp32_mode, True)
    shape = f"{lq}x{ld}" if wl["ragged"] is None else f"{lq}x~{wl['ragged'][0]} ({wl['ragged'][2]}..{wl['ragged'][3]} ragged)"
    out = {"workload": label or name, "shape": f"{NQ} queries x {NCAND} candidates, {shape} tokens, dim {h}, {index_dtype} index of "
                                               f"{len(doclens)} docs" + (f", fp32_mode {fp32_mode}" if fp32_mode != "exact" else ""),
           "steps": steps, "warmup": warmup, "prewarm_ms": PREWARM_MS,      # (untimed launches of step 0 in front of the warm-ups: timed_steps)
           "queries_per_s": round(NQ * steps / el, 1), "ms_per_step": round(el / steps * 1e3, 4),
           "ms_per_step_both_regions": [round(r[0] / steps * 1e3, 4) for r in runs]}
    out.update({k: rf[k] for k in ("kernel", "kernel_ms", "algorithmic_bytes_per_launch", "achieved", "frac", "traffic",
                                   "pmc_source", "mfma_busy_frac", "mfma_tflops")})
    if kern_ms < 0.5:
        # a launch this short: the pair of events around ONE launch also times its dispatch and the event packets (~10 us of
        # 180); the same launches back to back between ONE pair of events is what rocprofv3's kernel duration agrees with
        it = {"i": 0}

        def one():
            it["i"] += 1
            ranker.score_candidates(Q, cands[it["i"] % total])
        b2b = warm_then_time(one, 20)
        out["kernel_ms_back_to_back"] = round(b2b, 4)
        out["frac_back_to_back"] = round(alg / (b2b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        # ... and THAT is the launch duration the entry reports (kernel_ms / achieved / frac): it is the one rocprofv3's
        # kernel-trace average of the same kernel agrees with (profiles/r05_mv128_fp16_*: 0.0881 ms against 0.0879 here, 0.0974
        # through the event pair); the event-pair figure stays beside it
        out["kernel_ms_event_pair"], out["frac_event_pair"] = out["kernel_ms"], out["frac"]
        out["kernel_ms"], out["frac"] = out["kernel_ms_back_to_back"], out["frac_back_to_back"]
        out["achieved"] = round(alg / (b2b * 1e-3) / 1e9, 1)
        out["mfma_tflops"] = round(2.0 * lq * h * cand_tokens / (b2b * 1e-3) / 1e12, 1)
        # ... and what a bigger batch per launch reads (a launch this short pays its ramp and its one-round tail in full):
        # 8 x the queries, same docs per query
        nq8 = 8 * NQ
        Q8 = F.normalize(torch.randn(nq8, lq, h, generator=gq, device=dev), dim=-1).to(Q.dtype)
        c8 = draw_candidates(len(doclens), (4, nq8, NCAND), gen_c, dev)
        def one8():
            it["i"] += 1
            ranker.score_candidates(Q8, c8[it["i"] % 4])
        ms8 = warm_then_time(one8, 8)
        tok8, docs8 = live_tokens(ranker, c8, 0, len(doclens), 0, 4)
        alg8 = algorithmic_bytes(tok8, docs8, nq8, lq, h, esize, Q.element_size())
        out["batch_x8"] = {"queries_per_launch": nq8, "kernel_ms_back_to_back": round(ms8, 4),
                           "frac": round(alg8 / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del Q8, c8
    if online_call:     # the reference's online call on this index (its storage dtype): one rank_forward per query
        out["single_query"] = single_query_probe(ranker, Q, cands, h, lq, esize)
        if wl["ragged"] is None and h == 128 and online_call != "only":
            out["batched_retrieve_step"] = retrieve_step_probe(ranker, Q, dev)
    return out, (idx, doclens)


def sharded_share(colbert_amd, ranker, ndocs, dev, lq, h, esize, steps, warmup, Q1, cands1):
    """One rank's share of an N-way doc-sharded step, N = 2, 4, 8, on THIS GPU: 256*N queries whose 1000 candidates are
    drawn uniformly over all N shards, through the shipped per-rank path (ShardedRanker.local_topk: shard filter ->
    counted rerank from the device-built work list -> counted top-100 with global pids).  Everything a rank does per step
    except the all_gather + merge; the docs it reads are ~256 x 1000, as at N = 1.  The chip's clock drifts by a few
    percent with what ran before, so every share is bracketed by two short runs of the N = 1 step (256 dense rows, the
    headline workload) and compared with their mean."""
    from colbert_amd.sharded import ShardedRanker
    out = {}

    def n1():
        return bench_rows(ranker, Q1, cands1, warmup, steps, TOPK)[1]
    before = n1()
    for of in (2, 4, 8):
        job_rank = min(3, of - 1)
        lo, hi = job_rank * ndocs, (job_rank + 1) * ndocs
        sh = ShardedRanker(ranker, lo, hi)
        nq = NQ * of
        gq = torch.Generator(device=dev).manual_seed(1)
        Q = F.normalize(torch.randn(nq, lq, h, generator=gq, device=dev), dim=-1)
        nb = min(warmup + steps, 6)
        gen_c = torch.Generator(device=dev).manual_seed(2)
        cands = draw_candidates(of * ndocs, (nb, nq, NCAND), gen_c, dev)
        el, kern_ms = bench_rows(ranker, Q, cands, warmup, steps, TOPK, sharded=sh)
        after = n1()
        n1_kernel_ms, before = 0.5 * (before + after), after
        cand_tokens, docs = live_tokens(ranker, cands, lo, hi, warmup, steps)
        alg = algorithmic_bytes(cand_tokens, docs, nq, lq, h, esize, 4)
        out[str(of)] = {"kernel_ms": round(kern_ms, 4), "step_ms": round(el / steps * 1e3, 4),
                        "frac": round(alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": alg, "local_candidates_per_query": round(docs / nq, 2),
                        "queries_per_step": nq, "simulated_rank": job_rank,
                        # same docs per step as N = 1: the ratio is what sharding costs a rank before the exchange
                        "n1_kernel_ms_bracketing": round(n1_kernel_ms, 4), "kernel_ms_vs_n1": round(kern_ms / n1_kernel_ms, 4)}
        del cands, Q, sh
    out["how"] = ("ShardedRanker.local_topk on one GPU, rank 'simulated_rank' of N; kernel_ms = HIP events around the counted rerank "
                  "(work-list scan + fill + stream kernel); step_ms = wall per step incl. the shard filter and the top-k; no "
                  f"exchange; {steps} steps after {warmup} warm-up")
    return out


def sharded_from_files_check(colbert_amd, dev, rank, world, one_gpu):
    """N > 1 only: the shipped sharded path from the reference's index FILES to python lists, over the job's own process
    group (RCCL).  Rank 0 writes a small ragged index in the reference's format ({i}.pt + doclens.{i}.json, 4 parts) to
    the node's /tmp; every rank ``load_shard``s its pid range from it and all ranks call ``ShardedRanker.rank_forward``
    (1000 global pids) and ``ShardedRanker.retrieve_batch`` (16 queries x 32 x 64 GLOBAL token rows, masked query tokens);
    every rank also loads the whole index as one ``ColbertRanker`` and compares: the sharded results must equal the
    unsharded HIP path's.  A self-check of shipped code on the multi-GPU run, not a timed leg and not the oracle."""
    from colbert_amd.index_io import save_index
    from colbert_amd.sharded import load_shard
    path = os.path.join("/tmp", f"maxsim_bench_index_{os.environ.get('MASTER_PORT', '0')}")
    g = torch.Generator().manual_seed(21)
    ndocs, nparts, h, lq = 4096, 4, 128, 32
    doclens = (torch.randn(ndocs, generator=g) * 40 + 100).round().clamp(4, 180).long().tolist()
    cut = [0, 1000, 2048, 2500, ndocs]                      # part boundaries that are not shard boundaries
    if rank == 0:
        parts_dl = [doclens[cut[i]:cut[i + 1]] for i in range(nparts)]
        parts = [F.normalize(torch.randn(sum(dl), h, generator=g), dim=-1).half() for dl in parts_dl]
        save_index(path, parts, parts_dl)
        del parts
    dist.barrier()
    sh = load_shard(path, device=dev)                        # rank / world from the process group
    whole = colbert_amd.ColbertRanker(index_path=path, device=dev)
    gq = torch.Generator().manual_seed(22)                   # same inputs on every rank
    Q1 = F.normalize(torch.randn(1, lq, h, generator=gq), dim=-1).to(dev).permute(0, 2, 1)
    pids = torch.randperm(ndocs, generator=gq)[:1000].tolist()
    got = sh.rank_forward(Q1, pids, depth=TOPK)
    exp = whole.rank_forward(Q1, pids, depth=TOPK)
    rf_ok = got[1] == exp[1] and sorted(got[0]) == sorted(exp[0])
    nq, depth = 16, 64
    Qb = F.normalize(torch.randn(nq, lq, h, generator=gq), dim=-1).to(dev)
    keep = (torch.rand(nq, lq, generator=gq) > 0.25).long().to(dev)
    hot = torch.randint(0, ndocs, (nq, 300), generator=gq)
    offs = torch.tensor([0] + doclens).cumsum(0)
    docs = hot.gather(1, torch.randint(0, 300, (nq, lq * depth), generator=gq))
    ids = (offs[docs] + (torch.rand(nq, lq * depth, generator=gq) * torch.tensor(doclens)[docs]).long()).view(nq, lq, depth).to(dev)
    t0 = time.perf_counter()
    got_b = sh.retrieve_batch(Qb, keep, TOPK, embedding_ids=ids)
    rb_ms = (time.perf_counter() - t0) * 1e3
    exp_b = colbert_amd.retrieve_batch(whole, Qb, keep, TOPK, embedding_ids=ids)
    rb_ok = all(gs == es and sorted(gp) == sorted(ep) and dict(zip(gp, gs)) == dict(zip(ep, es))
                for (gp, gs), (ep, es) in zip(got_b, exp_b))
    flags = torch.tensor([int(rf_ok), int(rb_ok)], dtype=torch.int64, device="cpu" if one_gpu else dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        for f in os.listdir(path):
            os.remove(os.path.join(path, f))
        os.rmdir(path)
    return {"what": "load_shard from {i}.pt/doclens.{i}.json on every rank; ShardedRanker.rank_forward + retrieve_batch over the job's "
                    "process group vs the unsharded ColbertRanker over the same files (ties aside: equal score lists, equal pid sets)",
            "docs": ndocs, "parts": nparts, "rank_forward_equal_on_all_ranks": bool(flags[0].item()),
            "retrieve_batch_equal_on_all_ranks": bool(flags[1].item()), "retrieve_batch_first_call_ms": round(rb_ms, 3),
            "shard_of_rank0": [sh.lo, sh.hi], "strides": list(sh.local.strides)}


def sharded_retrieve_step(sharded, ranker, Q, dev, world, doclens, steps=6, faiss_depth=512, hot=1500):
    """N > 1 only, labelled extra (not `value`): the doc-sharded batched driver's step after the ANN search over the job's
    process group -- every rank gets the same 256 x world queries and the same GLOBAL token rows (32 tokens x faiss_depth
    synthetic ANN ids per query that cluster on ~1500 docs drawn over ALL shards; uniform docs, so row = pid * L + t), keeps
    the rows inside its token range (`id_base`), reranks its distinct docs (counted rows), takes a counted local top-100, the
    ONE all_gather, merge: ShardedRanker.local_retrieve_topk + all_gather_topk + merge_gathered, as
    ShardedRanker.retrieve_batch runs them, minus the final copy to python lists.  Wall time per step, max over ranks."""
    from colbert_amd.sharded import all_gather_topk, merge_gathered
    L = int(doclens[0])
    if any(int(x) != L for x in doclens[:1000]):
        return {"skipped": "needs a uniform index (row = pid * L + t)"}
    nq, lq = Q.size(0), Q.size(1)
    n = lq * faiss_depth
    nd_total = world * len(doclens)
    g = torch.Generator(device=dev).manual_seed(7)
    docs = torch.randint(0, nd_total, (nq, hot), generator=g, device=dev)
    ids = docs.gather(1, torch.randint(0, hot, (nq, n), generator=g, device=dev)) * L + torch.randint(0, L, (nq, n), generator=g, device=dev)
    keep = torch.ones(nq, lq, dtype=torch.bool, device=dev)

    def step():
        top_p, top_s = sharded.local_retrieve_topk(Q, keep, ids, TOPK)
        gs, gp = all_gather_topk(top_s, top_p, world, sharded.group)
        return merge_gathered(gs, gp, TOPK, sharded.topk_fn)
    for _ in range(2):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=out[0].device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) / steps * 1e3
    found = int((out[0] >= 0).sum().item())
    return {"shape": f"{nq} queries x {n} GLOBAL ANN token rows over {nd_total} docs, ~{hot} distinct docs per query over all {world} shards",
            "ms_per_step": round(ms, 4), "queries_per_s": round(nq / (ms * 1e-3), 1), "steps": steps,
            "merged_top_entries_found": found, "expected": nq * TOPK}


def main():
    t_main = time.time()
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
                    help="skip everything but the headline measurement (cpu_baseline, single_query, training_form, sharded_share, "
                         "other_workloads, read ceiling): profiling runs")
    ap.add_argument("--no-extras", action="store_true", help="skip sharded_share and other_workloads only")
    ap.add_argument("--pmc-sweep", action="store_true", help=argparse.SUPPRESS)     # live_pmc's FETCH_SIZE child (pmc_sweep_child)
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes first (roofline.traffic is then replayed from profiles/)")
    ap.add_argument("--extra-steps", type=int, default=10, help="timed steps of every sharded_share / other_workloads entry")
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

    if args.pmc_sweep:
        pmc_sweep_child(os.environ["MAXSIM_PMC_MANIFEST"])
        return
    # the default run measures its own PMC counters (children first: nothing in this process has touched the GPU yet)
    pmc_live = pmc_err = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("MAXSIM_BENCH_PMC_CHILD") and not args.no_pmc
            and args.workload == "c2" and not args.no_cpu_baseline and args.as_rank < 0 and not args.force_dist
            and not (args.ndocs or args.lq or args.nq or args.ncand or args.ld or args.q_dtype or args.index_dtype)
            and args.fp32_mode == "exact"):
        try:
            pmc_live, pmc_err = live_pmc()
        except Exception as e:      # whatever happens in the profiler passes, the bench line itself must still be produced
            pmc_live, pmc_err = None, f"{type(e).__name__}: {e}"

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
    dtype = TDT[args.index_dtype]
    esize = torch.empty(0, dtype=dtype).element_size()
    ndocs = args.ndocs or wl["ndocs"]
    doclens = make_doclens(wl, ndocs, LD, rank)
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
    Q = Q.to(TDT[q_dtype])
    # candidate lists: GLOBAL pids, the same on every rank (same seed); a ring of NB distinct batches (one batch of docs
    # is >= 23 GB of tokens >> the 256 MB Infinity Cache, so re-using a batch NB steps later still reads HBM)
    NB = min(total, 64) if job_world == 1 else min(total, 8)      # (long profile loops: 64 batches = 1.5 TB of docs between re-uses)
    gen_c = torch.Generator(device=dev).manual_seed(2)
    cands = draw_candidates(job_world * ndocs, (NB, nq, ncand_q), gen_c, dev)

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

    def make_step(batches):
        def step(i):
            timed["i"] = i
            cand_global = batches[i % batches.size(0)]
            if world == 1 and not use_dist and not sim:
                scores = timed_score(Q, cand_global)
                return ranker.topk(scores, cand_global, min(TOPK, ncand_q))
            # the shipped sharded path: shard filter -> counted rerank -> local top-k (global pids) ...
            top_p, top_s = sharded.local_topk(Q, cand_global, TOPK)
            # ... then the ONE exchange step (all_gather over xGMI) + the per-query merge on the side stream: batch i's
            # exchange overlaps batch i+1's rerank kernel; every batch is complete before the timed region ends
            if sim and not use_dist:
                return top_p, top_s
            h = sharded.exchange_async(top_p, top_s, TOPK)
            if os.environ.get("MAXSIM_BENCH_NO_PIPELINE"):   # diagnostic: resolve the exchange before the next batch is issued
                h.result()
            return h
        return step

    def run(batches):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; max over ranks."""
        sharded.exchange_events = None

        def barrier():
            if use_dist:
                dist.barrier()
            # the exchange timing list starts with the timed region (first call = before, second = after)
            if use_dist and sharded.exchange_events is None:
                sharded.exchange_events = []
        el = timed_steps(make_step(batches), args.warmup, args.steps, barrier)
        if use_dist:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            if one_gpu:
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        kern_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(args.warmup, total)) / args.steps
        per_step_ms[:] = [round(ev[i][0].elapsed_time(ev[i][1]), 4) for i in range(total)][:130]
        xch = sharded.exchange_events or []
        xch_ms = sum(a.elapsed_time(b) for a, b in xch) / len(xch) if xch else 0.0
        return el, kern_ms, xch_ms

    per_step_ms = []        # rerank kernel of every step, warm-ups first (side file only: does the region sit on a clock ramp?)
    with PowerSampler(dev.index if dev.index is not None else 0, firmware=(world == 1)) as power:
        el, kern_ms, xch_ms = run(cands)
    power = power.summary()
    headline_steps = list(per_step_ms)

    # algorithmic bytes of ONE rerank launch on this rank: doc tokens read once + Q + pid/offset/len + score
    cand_tokens, docs = live_tokens(ranker, cands, lo, hi, args.warmup, args.steps)
    alg_bytes = algorithmic_bytes(cand_tokens, docs, nq, LQ, H, esize, Q.element_size())

    # second, labelled measurement for N > 1: the same path on stratified lists (exactly 1000/N candidates per shard)
    strat = None
    if world > 1 and ncand_q % world == 0:
        per = ncand_q // world
        gs = torch.Generator(device=dev).manual_seed(3)
        sc = torch.cat([draw_candidates(ndocs, (NB, nq, per), gs, dev, lo=r * ndocs) for r in range(world)], dim=2)
        sc = sc[:, :, torch.randperm(ncand_q, generator=gs, device=dev)]        # shards interleaved within a list
        s_el, s_kern, s_xch = run(sc)
        strat = {"value": round(nq * args.steps / s_el, 2), "ms_per_step": round(s_el / args.steps * 1e3, 4),
                 "kernel_ms_rank0": round(s_kern, 4), "candidates": f"exactly {per} per shard per query"}
        del sc
    # third, labelled: SURVEY 8d's weak-scaling variant with 1000 candidates PER SHARD -- lists of 1000 x N global pids, every
    # rank scores as many docs per query as the N = 1 step does (the per-rank rerank then costs N x the N = 1 kernel: the job
    # does N x the queries AND N x the docs per query)
    per_shard = None
    if world > 1 and ncand_q * world <= 16384:
        gs = torch.Generator(device=dev).manual_seed(4)
        nb2 = min(NB, 4)
        sc = torch.cat([draw_candidates(ndocs, (nb2, nq, ncand_q), gs, dev, lo=r * ndocs) for r in range(world)], dim=2)
        sc = sc[:, :, torch.randperm(ncand_q * world, generator=gs, device=dev)]
        p_el, p_kern, p_xch = run(sc)
        per_shard = {"value": round(nq * args.steps / p_el, 2), "ms_per_step": round(p_el / args.steps * 1e3, 4),
                     "kernel_ms_rank0": round(p_kern, 4), "candidates": f"{ncand_q} per shard per query ({ncand_q * world} per list)",
                     "docs_scored_per_s_all_ranks": round(nq * ncand_q * world * args.steps / p_el, 1)}
        del sc

    # per-rank figures (every rank contributes one row)
    per_rank = None
    if use_dist:
        row = torch.tensor([kern_ms, xch_ms, docs / nq], dtype=torch.float64, device="cpu" if one_gpu else dev)
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        per_rank = {"rerank_kernel_ms": [round(float(r[0]), 4) for r in rows],
                    "exchange_merge_ms": [round(float(r[1]), 4) for r in rows],
                    "local_candidates_per_query": [round(float(r[2]), 2) for r in rows]}

    files_check = None
    if use_dist:
        try:
            files_check = sharded_from_files_check(colbert_amd, dev, rank, world, one_gpu)
        except Exception as e:        # a failed self-check is reported in the line, it must not lose the measurement
            files_check = {"error": f"{type(e).__name__}: {e}"}
    shard_retrieve = None
    if use_dist and world > 1 and not sim:
        try:
            shard_retrieve = sharded_retrieve_step(sharded, ranker, Q, dev, world, doclens)
        except Exception as e:
            shard_retrieve = {"error": f"{type(e).__name__}: {e}"}

    default_shape = ndocs == wl["ndocs"] and not (args.lq or args.nq or args.ncand or args.ld or args.q_dtype) and (world == 1)
    suffix = f"_shard{job_world}" if sim else ""
    rf = roofline_entry(kern_ms, alg_bytes, cand_tokens, LQ, H, args.workload, args.index_dtype, args.fp32_mode,
                        default_shape, suffix)

    if pmc_live is not None and default_shape:
        rf["traffic_replayed"], rf["pmc_source_replayed"] = rf["traffic"], rf["pmc_source"]
        rf.update({k: pmc_live[k] for k in ("traffic", "hbm_read_bytes", "hbm_write_bytes", "mfma_busy_frac", "pmc_source", "effective_clock_GHz")})
        rf["pmc_launches_sampled"] = pmc_live["launches_sampled"]
    elif pmc_err is not None:
        rf["pmc_live_error"] = pmc_err
    rf["kernel_ms_per_step"] = headline_steps
    rf["power"] = power        # board power / cap / sysfs shader clock over the headline's warm-up + timed steps (best effort)

    if rank == 0:
        rag = wl["ragged"]
        res = {
            "metric": "queries/sec MaxSim rerank, 32q x 180d tokens, dim=128, 1000 docs/query" if args.workload == "c2"
                      else f"queries/sec MaxSim rerank, workload {args.workload}",
            "value": round(nq * args.steps / el, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload.upper()}: {args.nq or NQ} queries/GPU x {ncand_q} candidates/query, {LQ}x"
                                   f"{f'~{rag[0]} ({rag[2]}..{rag[3]} ragged)' if rag else LD} tokens, dim {H}, "
                                   f"{args.index_dtype} index of {ndocs} docs/GPU in HBM, fused rerank + top-{TOPK}"
                                   + (f", doc-sharded x{world}: candidates uniform over all {world * ndocs} pids, shard filter + "
                                      f"local top-{TOPK} + RCCL all_gather + merge" if world > 1 else "")
                                   + (f", SIMULATED rank {job_rank} of {job_world} (one rank's share of the job, no exchange)" if sim else ""),
                       "queries_per_step": nq, "candidates_per_query": ncand_q, "docs_per_gpu": ndocs,
                       "index_dtype": args.index_dtype, "q_dtype": q_dtype, "fp32_mode": args.fp32_mode, "parallelism": f"doc-shard x{world}"},
            "roofline": rf,
        }
        if use_dist:
            res["n_ranks_seen"] = dist.get_world_size()
            res["backend"] = dist.get_backend()
            res["per_rank"] = per_rank
            res["sharded_from_files"] = files_check
            # the self-check gates the line: false = the sharded path did not reproduce the unsharded ranker on this job's
            # own process group (or the check itself failed) -- read `value` with that in mind
            res["sharded_self_check_ok"] = bool(files_check and files_check.get("rank_forward_equal_on_all_ranks")
                                                and files_check.get("retrieve_batch_equal_on_all_ranks"))
            res["sharded_retrieve_step"] = shard_retrieve
        if strat is not None:
            res["stratified"] = strat
        if per_shard is not None:
            res["per_shard_1000"] = per_shard
        full = world == 1 and args.workload == "c2" and not args.no_cpu_baseline and not sim and default_shape
        if full:
            ceil = read_ceiling(idx)
            rf["read_ceiling"] = ceil
            rf["read_ceiling_GBps"] = ceil["GBps"]
            rf["frac_of_read_ceiling"] = round(rf["achieved"] / ceil["GBps"], 4)
            res["single_query"] = single_query_probe(ranker, Q, cands, H, LQ, esize)
        if full and not args.no_extras:
            xs, xw = args.extra_steps, 3
            res["sharded_share"] = sharded_share(colbert_amd, ranker, ndocs, dev, LQ, H, esize, xs, xw, Q, cands)
            others = []
            # the opt-in 3 x bf16 contraction of the SAME fp32 index (fp32-class accuracy, not an exact fmaf chain: labelled extra)
            o, keep = extra_workload(colbert_amd, "c2", dev, xs, xw, fp32_mode="bf16x3", reuse=(idx, doclens), online_call="only", label="c2 fp32 index, fp32_mode=bf16x3 (opt-in)")
            o["accuracy"] = "fp32-class (tests/test_gpu_parity.py::test_fp32_bf16x3_mode_is_fp32_accurate); default stays the exact fmaf chain"
            o["pmc_key"] = "c2_bf16x3"
            others.append(o)
            # the 92 GB headline index leaves HBM before the next ones are built (every name that reaches it is cleared:
            # the closures above share these cells)
            ranker = sharded = idx = cands = score_inner = keep = None
            torch.cuda.empty_cache()
            for key, name, kw, label in (("c2_fp16", "c2", dict(index_dtype="fp16", online_call=True), "c2 with the reference's fp16 index (colbert_ranker.py:62)"),
                                         ("ragged", "ragged", {}, "ragged fp32 (doclens N(120,40) in 8..180)"),
                                         ("ragged_bf16x3", "ragged", dict(fp32_mode="bf16x3", reuse_prev=True), "ragged fp32 index, fp32_mode=bf16x3 (opt-in; fp32-class accuracy, "
                                                              "tests/test_gpu_parity.py::test_fp32_bf16x3_mode_is_fp32_accurate)"),
                                         ("ragged_fp16", "ragged", dict(index_dtype="fp16"), "ragged docs on the reference's fp16 index (doclens N(120,40) in 8..180, colbert_ranker.py:62)"),
                                         ("c4", "c4", {}, "C4 multi-view: 8 x 8 tokens (BASELINE configs[3])"),
                                         ("c5", "c5", {}, "C5 bf16 dim 768, 32 x 256 tokens (BASELINE configs[4])"),
                                         ("dep768", "dep768", dict(online_call=True), "reference default deployment: dim 768, fp16 index, doclens N(200,80) in 8..384 "
                                                              "(proj_conf/dense.yaml:6-8, encoder.py:175)"),
                                         ("mv128", "mv128", {}, "multi-view on the reference's fp16 index: 8 x 8 tokens, dim 128 (C4's shape in the storage "
                                                                "dtype of colbert_ranker.py:62; k_maxsim_stream_uni16)"),
                                         ("mv768", "mv768", {}, "the reference's DEFAULT multi-view deployment: q_view = d_view = 16, dim 768, fp16 index "
                                                                "(proj_conf/dense.yaml:8,29-32, BaseModel.py:21-24, colbert_ranker.py:62)")):
                kw = dict(kw)
                reuse = keep if kw.pop("reuse_prev", False) else None      # (the same fp32 tokens, another contraction)
                keep = None
                if reuse is None:
                    torch.cuda.empty_cache()
                o, keep = extra_workload(colbert_amd, name, dev, xs, xw, label=label, reuse=reuse, **kw)
                o["pmc_key"] = key
                others.append(o)
            keep = None
            torch.cuda.empty_cache()
            if "read_ceiling" in rf:
                for o in others:
                    o["frac_of_read_ceiling"] = round(o["achieved"] / rf["read_ceiling"]["GBps"], 4)
            for o in others:       # this run's own FETCH_SIZE pass over the workload's launches (live_pmc), where there is one
                o["key"] = o.pop("pmc_key")
                live = (pmc_live or {}).get("sweep", {}).get(o["key"])
                if live is not None:
                    o["traffic_replayed"], o["pmc_source_replayed"] = o["traffic"], o["pmc_source"]
                    o["traffic"] = live["hbm_read_bytes"]
                    o["read_over_algorithmic"] = live["read_over_algorithmic"]
                    o["pmc_source"] = ("live: FETCH_SIZE pass of this run over 3 launches of this workload (reads only; the writes are "
                                       "its score matrix, 1 MB; mfma_busy_frac stays the replayed value)")
            res["other_workloads"] = others
        if full:
            res["training_form"] = training_form_probe(dev)
            res["cpu_baseline"] = cpu_baseline()
        res["wall_s"] = round(time.time() - t_main, 1)       # the whole run (index build, extra workloads, PMC children, CPU baseline)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(compact_line(res)) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def compact_line(res):
    """The ONE stdout line.  Everything measured goes, in full, to a side file (`details_file`: gpurun_out/bench_details.json
    next to this script); the line keeps the contract's fields plus one short numeric row per extra measurement, so that the
    part of it a log tail keeps still shows every workload's roofline fraction.  Long prose (how / what / notes / sources)
    lives in the side file only."""
    path = os.path.join(ROOT, "gpurun_out", "bench_details.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
        details = os.path.relpath(path, ROOT)
    except OSError as e:
        details = f"not written: {type(e).__name__}"
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data", "config") if k in res}
    for k in ("wall_s", "n_ranks_seen", "backend", "per_rank", "sharded_self_check_ok", "stratified", "per_shard_1000"):
        if k in res:
            out[k] = res[k]
    if res.get("sharded_retrieve_step"):
        out["sharded_retrieve_step"] = {k: v for k, v in res["sharded_retrieve_step"].items() if k in ("ms_per_step", "queries_per_s", "error", "skipped")}
    if res.get("sharded_from_files"):
        out["sharded_from_files"] = {k: v for k, v in res["sharded_from_files"].items()
                                     if k in ("rank_forward_equal_on_all_ranks", "retrieve_batch_equal_on_all_ranks", "error")}
    if "cpu_baseline" in res:
        cb = res["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample")}
        out["cpu_baseline"]["c1_queries_per_s"] = cb.get("c1", {}).get("value")
        out["cpu_baseline"]["rank_forward_queries_per_s"] = cb.get("rank_forward", {}).get("value")
    if "single_query" in res:
        sq = res["single_query"]
        out["single_query"] = {k: sq.get(k) for k in ("median_ms", "gpu_span_ms", "host_ms")}
    if "sharded_share" in res:
        out["sharded_share"] = {n: {"kernel_ms": v["kernel_ms"], "vs_n1": v["kernel_ms_vs_n1"], "frac": v["frac"]}
                                for n, v in res["sharded_share"].items() if isinstance(v, dict)}
    if "training_form" in res:
        tf = res["training_form"]
        out["training_form"] = {"forward_ms": tf["forward_ms"], "frac_of_2.5PF": tf["frac"], "tflops": tf["tflops"],
                                "backward_ms": tf["backward_ms"], "vendor_gemm_step_shape_tflops": tf["vendor_gemm"]["step_shape_8704x52224x768_tflops"]}
    rows = []
    for o in res.get("other_workloads", []):
        row = {"workload": o.get("key", o["workload"]), "frac": o["frac"], "kernel_ms": o["kernel_ms"], "qps": o["queries_per_s"],
               "read_over_alg": o.get("read_over_algorithmic")}
        if "frac_back_to_back" in o:
            row["frac_b2b"] = o["frac_back_to_back"]
        if "batch_x8" in o:
            row["frac_x8"] = o["batch_x8"]["frac"]
        if "single_query" in o:
            row["online_ms"] = o["single_query"]["median_ms"]
        if "batched_retrieve_step" in o:
            b = o["batched_retrieve_step"]
            row["retrieve"] = {"ids_to_pids_ms": b["ids_to_pids_ms"], "rerank_ms": b["counted_rerank_ms"], "topk_ms": b["counted_topk_ms"],
                               "one_query_ms": b["one_query_end_to_end_ms"]}
        rows.append(row)
    if rows:
        out["other_workloads"] = rows
    out["details_file"] = details
    # roofline LAST (a truncated tail keeps it): numbers and short strings only
    rf = dict(res["roofline"])
    if isinstance(rf.get("pmc_source"), str):
        rf["pmc_source"] = "live rocprofv3 --pmc child passes of this run" if rf["pmc_source"].startswith("live") else rf["pmc_source"]
    rf.pop("kernel_ms_per_step", None)
    rc = rf.pop("read_ceiling", None)
    if rc:
        rf["read_ceiling_by_ring_shape"] = rc.get("by_ring_shape")
    if rows:
        rf["other_workloads_frac"] = {r["workload"]: r["frac"] for r in rows}
    out["roofline"] = rf
    return out


def retrieve_step_probe(ranker, Q, dev, faiss_depth=512, hot=1500):
    """The batched driver's step after the ANN search (the reference: per query `emb2pid` + `set()` in a
    Pool(16), then rank_forward -- colbert_ranker.py:212-229, dense_server_client.py:44-48): 256 queries x (32 tokens x
    faiss_depth) synthetic ANN ids that cluster on ~1500 docs per query -> distinct pids (counted rows, nothing read
    back) -> counted rerank -> counted top-100.  HIP events around each of the three launches' groups, 8 repetitions after 60 ms of the same calls."""
    nq, nd = Q.size(0), ranker.n_docs
    n = Q.size(1) * faiss_depth
    g = torch.Generator(device=dev).manual_seed(7)
    docs = torch.randint(0, nd, (nq, hot), generator=g, device=dev)
    L = int(ranker.d_doclens[0].item())
    ids = docs.gather(1, torch.randint(0, hot, (nq, n), generator=g, device=dev)) * L + torch.randint(0, L, (nq, n), generator=g, device=dev)

    def t(f, k=8):      # (the batched step is a serving loop: timed on warm clocks, after 60 ms of the same calls)
        return warm_then_time(f, k)
    ids = ids.view(nq, Q.size(1), faiss_depth)
    keep = torch.ones(nq, Q.size(1), dtype=torch.bool, device=dev)       # (the driver's keep-mask, applied in the kernels)
    cand, cnt = ranker.embedding_ids_to_pids(ids, trim=False, keep=keep)
    sc = ranker.score_candidates(Q, cand, q_mask=keep, cand_count=cnt)
    a = t(lambda: ranker.embedding_ids_to_pids(ids, trim=False, keep=keep))
    b = t(lambda: ranker.score_candidates(Q, cand, q_mask=keep, cand_count=cnt))
    c = t(lambda: ranker.topk(sc, cand, TOPK, cnt))
    live = cand[cand >= 0]
    rerank_bytes = int(ranker.d_doclens[live].sum().item()) * Q.size(2) * ranker.tensor.element_size() + live.numel() * 24
    ids_bytes = 2 * nq * n * 8                                           # ids read + pid rows written (-1 tail included)
    # the same driver serving ONE query at a time (the reference's server loop, dense_server_client.py:56-63): the whole
    # colbert_amd.retrieve_batch call, ids on the device in, python lists on the host out -- wall clock, median of 30
    import colbert_amd
    ids1 = ids[:1]
    keep1 = torch.ones(1, Q.size(1), dtype=torch.bool, device=dev)
    for _ in range(5):
        colbert_amd.retrieve_batch(ranker, Q[:1], keep1, topk=TOPK, embedding_ids=ids1)
    lat = []
    for _ in range(30):
        t0 = time.perf_counter()
        colbert_amd.retrieve_batch(ranker, Q[:1], keep1, topk=TOPK, embedding_ids=ids1)
        lat.append(time.perf_counter() - t0)
    lat.sort()
    return {"shape": f"{nq} queries x {n} ANN ids (faiss_depth {faiss_depth}), {float(cnt.float().mean()):.0f} distinct candidates per query",
            "ids_to_pids_ms": round(a, 4), "counted_rerank_ms": round(b, 4), "counted_topk_ms": round(c, 4),
            "step_ms": round(a + b + c, 4), "queries_per_s": round(nq / ((a + b + c) * 1e-3), 1),
            "one_query_end_to_end_ms": round(lat[len(lat) // 2] * 1e3, 4),
            # per-stage rooflines (HBM): the rerank streams the distinct candidates' tokens; ids -> pids reads the ids and
            # writes the pid rows (its table lookups are cache-resident); the top-k moves ~3 MB and is latency-bound
            "counted_rerank_roofline": {"bound": "hbm", "achieved": round(rerank_bytes / b / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(rerank_bytes / b / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": rerank_bytes},
            "ids_to_pids_roofline": {"bound": "hbm", "achieved": round(ids_bytes / a / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(ids_bytes / a / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes": ids_bytes,
                                     "note": "one 1024-thread workgroup per query: 16 384 divergent 8-byte table lookups through one CU's "
                                             "vector-memory path, hash-set dedupe in LDS, a 2048-key register sort; bound by that "
                                             "lookup rate and the workgroup's serial chain, not by bytes (stamped phases: "
                                             "docs/experiments.md round 5)"},
            "profile": "profiles/r05_retrieve_step_kernel_stats.csv, profiles/r05_retrieve_step_pmc.json (tools/bench_retrieve_step.py)"}


def single_query_probe(ranker, Q, cands, H, LQ, esize):
    """The reference's online call: ONE query x 1000 candidates through rank_forward (faiss_indexers.py:234), python
    list in, python lists out, host-synchronous -- latency, not throughput.  `gpu_span_ms` is the time between two HIP
    events recorded on the launch stream right before and after the call (both kernels + the gap between them);
    `host_ms` = end-to-end minus that span (the events themselves add a few us to the span: the kernel's own duration is
    in profiles/r05_single_query_*)."""
    Q1 = Q[:1].float().permute(0, 2, 1)     # [1, h, Lq]: the permuted VIEW of a [1, Lq, h] tensor, as faiss_indexers.py:232-233 hands it over
    out = {"call": "rank_forward(Q[1,h,Lq], 1000 pids, depth=100) -> python lists"}
    lat, span = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    ntok = int(ranker.d_doclens[torch.tensor(pids1, device=ranker.device)].sum().item())     # tokens of the last call's candidates
    out.update({"median_ms": round(med, 4), "min_ms": round(lat[0] * 1e3, 4), "gpu_span_ms": round(gspan, 4),
                "host_ms": round(max(med - gspan, 0.0), 4), "queries_per_s_sequential": round(1e3 / med, 1),
                "algorithmic_GBps_over_gpu_span": round((ntok * H * esize + LQ * H * 4) / (gspan * 1e-3) / 1e9, 1),
                "kernel_profile": "profiles/r05_single_query_summary.json"})
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
    b16 = int(ranker.d_doclens[cands[0, :16].flatten()].sum().item()) * H * esize
    out["batch16"] = {"kernel_ms": round(ks[len(ks) // 2], 4), "algorithmic_GBps": round(b16 / (ks[len(ks) // 2] * 1e-3) / 1e9, 1),
                      "how": "20 launches of 16 queries x 1000 candidates back to back, HIP events around the 20"}
    return out


def training_form_probe(dev):
    """The operator's second caller: BaseModel.score on the gathered training batch (colbert_model.py:87-90),
    every query against every doc, at the reference's step -- Q 272 x 32 x 768, D 544 x 384 x 768 bf16 (dense.yaml:6-8) --
    through maxsim_score_dense_fwd (scores + arg-max for the backward) and maxsim_score_dense_bwd (dQ and dD through the
    saved arg-max).  Matrix-bound, unlike the rerank path: the forward is priced against the dense bf16 MFMA peak.  Not the
    headline metric; one entry so that the numbers are in the bench record."""
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
    gout = torch.randn(nq, nd, generator=g, device=dev)
    dQ = torch.empty(nq, lq, h, device=dev)
    dD = torch.empty(nd, ld, h, device=dev)
    wsb = int(_lib.lib.maxsim_score_dense_bwd_workspace(nq, nd, lq, ld))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    bf, mf = _DT[torch.bfloat16], _MDT[torch.float32]

    def fwd():
        rc = _lib.lib.maxsim_score_dense_fwd(Qt.data_ptr(), Dt.data_ptr(), qm.data_ptr(), dm.data_ptr(), nq, nd, lq, ld, h,
                                             bf, mf, out.data_ptr(), arg.data_ptr(), st)
        assert rc == 0, rc

    def bwd():
        rc = _lib.lib.maxsim_score_dense_bwd(Qt.data_ptr(), Dt.data_ptr(), qm.data_ptr(), dm.data_ptr(), arg.data_ptr(),
                                             gout.data_ptr(), nq, nd, lq, ld, h, bf, mf, dQ.data_ptr(), dD.data_ptr(),
                                             ws.data_ptr(), wsb, st)
        assert rc == 0, rc

    def t(f, n=20):
        return warm_then_time(f, n)
    ms, bms = t(fwd), t(bwd)
    flop = 2.0 * nq * nd * lq * ld * h
    # yardstick, measured in this run like roofline.read_ceiling: what the vendor GEMM (torch.matmul -> hipBLASLt) reaches
    # on this box under its power cap -- a large square bf16 GEMM, and the step's own shape (K = 768) on a quarter of the
    # docs WITH the similarity matrix written out, which the fused kernel never does.  Not on the product path.
    ga, gb = torch.randn(8192, 8192, device=dev).bfloat16(), torch.randn(8192, 8192, device=dev).bfloat16()
    gout_sq = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    sq_ms = t(lambda: torch.matmul(ga, gb.t(), out=gout_sq), 10)
    ga = gb = gout_sq = None
    Q2, D2 = Qt.view(nq * lq, h), Dt.view(nd * ld, h)[: nd * ld // 4]
    sim = torch.empty(nq * lq, nd * ld // 4, device=dev, dtype=torch.bfloat16)
    st_ms = t(lambda: torch.matmul(Q2, D2.t(), out=sim), 10)
    sim = None
    vendor = {"square_8192_bf16_tflops": round(2.0 * 8192 ** 3 / sq_ms / 1e9, 1),
              "step_shape_8704x52224x768_tflops": round(flop / 4 / st_ms / 1e9, 1),
              "what": "torch.matmul (hipBLASLt) in this run, 10 launches between two HIP events; plain GEMMs that write their result"}
    # backward: two gather-reduce passes through the saved arg-max (maxsim_backward.h).  Algorithmic bytes of one launch
    # group: every (q, m, d) triple selects ONE row of D for dQ and adds ONE row of Q into dD (rows of h elements), the
    # arg-max tensor is read by both passes, dQ and dD are written once in fp32.  The rows come out of tables that are
    # cache-sized (Q 13 MB: L2 / Infinity Cache; D 321 MB: Infinity Cache + HBM), so the yardstick is the guide's measured
    # row-gather rate (MI355X_MICROARCH.md "Indexed rows": 7.4-7.9 TB/s from a 151 MB table, 8.6 TB/s from 38 MB), next to
    # the HBM spec peak the bench line's other fractions use
    triples = nq * nd * lq
    esz = Qt.element_size()
    bwd_bytes = {"dQ_gathered_D_rows": triples * h * esz, "dD_gathered_Q_rows": triples * h * esz, "argmax_reads": 2 * triples * 4,
                 "grad_reads": 2 * nq * nd * 4, "dQ_write": nq * lq * h * 4, "dD_write": nd * ld * h * 4}
    bwd_total = sum(bwd_bytes.values())
    bwd = {"ms": round(bms, 4), "algorithmic_bytes": bwd_bytes, "algorithmic_bytes_total": bwd_total,
           "roofline": {"bound": "row gather from cache-resident tables", "achieved": round(bwd_total / bms / 1e6, 1),
                        # NOT a ceiling: the guide's measured gather rate from a 38 MB table -- one kernel of this pair (dD, rows
                        # partly L2-resident) runs above it; a reference rate to read `achieved` against
                        "reference_rate": 8600.0, "unit": "GB/s", "frac_of_reference_rate": round(bwd_total / bms / 1e6 / 8600.0, 4),
                        "frac_of_hbm_peak": round(bwd_total / bms / 1e6 / HBM_PEAK_GBS, 4),
                        "note": "the gathered rows are served by L2 / Infinity Cache (tables of 13 MB and 321 MB), not streamed from "
                                "HBM: reference_rate = the guide's measured row-gather rate from a 38 MB table (MI355X_MICROARCH.md 'Indexed "
                                "rows': 8.6 TB/s; 7.4-7.9 from 151 MB); per kernel (profiles/r05_train_*): dQ 7.6 TB/s, dD 9.8 TB/s "
                                "(partly L2), index pass 40 MB in 0.05 ms"},
           "kernels": "profiles/r05_train_kernel_stats.csv: k_maxsim_bwd_dq_v8 0.953 ms (7.27 GB of D rows: 7.6 TB/s), k_maxsim_bwd_dd_rows "
                      "0.811 ms (7.27 GB of Q rows + 0.64 GB written: 9.8 TB/s), k_maxsim_bwd_index 0.049 ms; PMC: profiles/r05_train_pmc.json"}
    return {"op": "maxsim_score_dense_fwd (scores + arg-max) / maxsim_score_dense_bwd (dQ, dD), Q 272x32x768 x D 544x384x768, bf16, prefix d_mask",
            "kernel": "k_maxsim_allpairs" if _lib.lib.maxsim_score_dense_kernel(nq, nd, lq, ld, h, bf, mf) == 1 else "k_maxsim_stream_bigh",
            "forward_ms": round(ms, 4), "backward_ms": round(bms, 4), "forward_backward_ms": round(ms + bms, 4),
            "tflops": round(flop / ms / 1e9, 1), "peak_tflops_dense_bf16": 2500.0,
            "frac": round(flop / ms / 1e9 / 2500.0, 4), "vendor_gemm": vendor,
            "frac_of_vendor_square_gemm": round(flop / ms / 1e9 / vendor["square_8192_bf16_tflops"], 4), "how": "20 launches back to back between two HIP events after 60 ms of the same launches, forward and backward separately",
            "backward": bwd,
            "profile": "profiles/r05_train_kernel_stats.csv, profiles/r05_train_pmc.json (forward with arg-max 2.49 ms, without 2.33 ms; "
                       "r02_allpairs_* hold the forward's SQ counter passes)"}


if __name__ == "__main__":
    main()
