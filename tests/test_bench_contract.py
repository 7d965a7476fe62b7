"""CPU: the parts of bench.py the driver's contract hangs on that need no GPU -- the ONE compact stdout line (contract fields
present, `roofline` and `cpu_baseline` objects intact, short enough for a log tail, every extra workload's fraction visible),
candidate lists drawn without replacement, and the power sampler's "nothing readable" behaviour."""
import importlib.util
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_compact_line_keeps_the_contract(tmp_path, monkeypatch):
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_builder_run_details.json")))   # a full record of a real run
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))                   # the side file goes to a scratch directory
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)                                  # fits a log tail whole
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert list(line)[-1] == "roofline"                                 # a truncated tail still shows it
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "effective_clock_GHz", "read_ceiling_GBps", "power"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb)
    names = [o["workload"] for o in line["other_workloads"]]
    assert {"c2_fp16", "ragged", "c4", "c5", "dep768", "mv128", "mv768"} <= set(names)
    assert set(rf["other_workloads_frac"]) == set(names)
    for o in line["other_workloads"]:
        assert 0.3 < o["frac"] < 1.0 and o["kernel_ms"] > 0
    assert line["config"]["workload"].startswith("C2: 256 queries/GPU x 1000 candidates/query")
    # the full record went to the side file
    side = json.load(open(os.path.join(str(tmp_path), line["details_file"])))
    assert side["roofline"]["algorithmic_bytes_per_launch"] == full["roofline"]["algorithmic_bytes_per_launch"]


def test_candidates_are_drawn_without_replacement():
    g = torch.Generator().manual_seed(3)
    c = bench.draw_candidates(1200, (3, 40, 1000), g, "cpu")           # 1000 of 1200: a plain randint would repeat ~340 per list
    assert c.shape == (3, 40, 1000) and int(c.min()) >= 0 and int(c.max()) < 1200
    s = c.sort(dim=-1).values
    assert not bool((s[..., 1:] == s[..., :-1]).any())
    d = bench.draw_candidates(50, (7, 50), g, "cpu", lo=1000)           # a whole shard's pid range: a permutation of it
    assert torch.equal(d.sort(dim=-1).values, torch.arange(1000, 1050).expand(7, 50))


def test_power_sampler_without_a_gpu_reports_unavailable():
    with bench.PowerSampler(0) as p:
        pass
    assert p.summary() == {"available": False}


def test_timed_steps_keeps_the_gpu_busy_between_warm_up_and_timed_steps(monkeypatch):
    """Round 5 found -4 % on every 20-step figure: the garbage collection (30-45 ms of host time) ran between the warm-up and
    the timed steps, the GPU's clocks fell and the first timed launches climbed back.  The order that must hold: collect,
    W warm-ups, barrier, synchronize, K steps, synchronize, barrier -- nothing else between the last warm-up and the first
    timed step, the collector off for the whole region and restored afterwards."""
    import gc
    log = []
    monkeypatch.setattr(bench.gc, "collect", lambda *a: log.append("collect") or 0)
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a: log.append("sync"))
    was = gc.isenabled()
    el = bench.timed_steps(lambda i: log.append(("step", i, gc.isenabled())), 3, 4, barrier=lambda: log.append("barrier"))
    assert el >= 0 and gc.isenabled() == was
    assert log == ["collect", ("step", 0, False), ("step", 1, False), ("step", 2, False), "barrier", "sync",
                   ("step", 3, False), ("step", 4, False), ("step", 5, False), ("step", 6, False), "sync", "barrier", "sync"]
