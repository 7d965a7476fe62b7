"""Register / LDS / occupancy guard of the hot kernels (CPU: hipcc cross-compiles gfx950 without a GPU).

The rerank kernels are latency-hiding streams whose speed depends on how many waves a SIMD holds, i.e. on hipcc's
register assignment: docs/experiments.md records -16 % twice when an edit outside the hot loop moved the fp32 kernel from
186 to 214-244 VGPRs.  This test compiles tu_stream / tu_bigh_rerank / tu_allpairs with
-Rpass-analysis=kernel-resource-usage (what tools/resource_usage.py prints) and compares every guarded kernel with the
committed table tests/golden/kernel_resources.json:
  * no scratch, no VGPR spills -- ever; SGPRs parked in VGPR lanes (the list forms do that) must not grow;
  * waves per SIMD (the occupancy step the launch heuristics were tuned for) must not drop;
  * VGPR + AGPR count must stay within the kernel's 8-register allocation granule budget: moving to another granule row of
    MI355X_MICROARCH.md's register table (or by more than 8 registers inside the 2-waves row) fails;
  * static LDS unchanged (the rings are dynamic LDS sized by the launchers).
After a DELIBERATE kernel change: `python tools/resource_usage.py --write-table`, look at the diff, commit it with the
measurement that justifies it."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import resource_usage  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


def waves_by_registers(n):
    """MI355X_MICROARCH.md 'Register files': allocation granule 8, waves per SIMD = min(8, 512 // allocated)."""
    alloc = (n + 7) // 8 * 8
    return min(8, 512 // max(alloc, 8))


@pytest.fixture(scope="module")
def tables():
    want = json.load(open(resource_usage.TABLE))["kernels"]
    got = resource_usage.guarded_table()
    return want, got


def problems(want, got):
    out = []
    for tu, kernels in want.items():
        for name, w in kernels.items():
            g = got.get(tu, {}).get(name)
            if g is None:
                out.append(f"{tu}: {name} is no longer built (dispatch changed? regenerate the table)")
                continue
            if g["scratch"] != 0 or g["vgpr_spill"] != 0:
                out.append(f"{tu}: {name} spills: scratch {g['scratch']} B/lane, vgpr_spill {g['vgpr_spill']}")
            if g["sgpr_spill"] > w["sgpr_spill"]:        # (SGPRs parked in VGPR lanes: no memory traffic, but instructions in the loop)
                out.append(f"{tu}: {name} sgpr_spill {w['sgpr_spill']} -> {g['sgpr_spill']}")
            if g["waves_per_simd"] < w["waves_per_simd"]:
                out.append(f"{tu}: {name} waves/SIMD {w['waves_per_simd']} -> {g['waves_per_simd']}")
            wr, gr = w["vgpr"] + w["agpr"], g["vgpr"] + g["agpr"]
            if waves_by_registers(gr) < waves_by_registers(wr) or gr > wr + 8:
                out.append(f"{tu}: {name} registers {w['vgpr']}+{w['agpr']} -> {g['vgpr']}+{g['agpr']}")
            if g["lds_static"] != w["lds_static"]:
                out.append(f"{tu}: {name} static LDS {w['lds_static']} -> {g['lds_static']}")
    for tu, kernels in got.items():
        for name in kernels:
            if name not in want.get(tu, {}):
                out.append(f"{tu}: {name} is built but not in the table (regenerate it: tools/resource_usage.py --write-table)")
    return out


def test_hot_kernels_keep_their_registers_and_occupancy(tables):
    want, got = tables
    assert sum(len(v) for v in want.values()) >= 40
    bad = problems(want, got)
    assert not bad, "\n".join(bad)


def test_headline_kernels_are_in_the_table(tables):
    want, _ = tables
    for tu, name in (("tu_stream", "k_maxsim_stream<0, 0, 4, 1, 0, 48, false, false, false>"),        # C2 fp32 exact (headline)
                     ("tu_stream", "k_maxsim_stream<0, 1, 4, 2, 0, 48, false, false, false>"),        # fp16 index
                     ("tu_stream", "k_maxsim_stream<0, 2, 4, 2, 0, 48, false, false, false>"),        # bf16 index
                     ("tu_stream", "k_maxsim_stream<0, 1, 4, 2, 0, 48, false, false, true>"),   # ragged fp16 index (token-balanced cut)
                     ("tu_stream", "k_maxsim_stream_uni<8, 1, 8, 0, false>"),                  # C4
                     ("tu_stream", "k_maxsim_stream_uni16<1, 8, 1, 8, 1, false>"),             # mv128
                     ("tu_bigh_rerank", "k_maxsim_stream_bigh<0, 2, 1, 4, 2, false, 1, false, false, false, false, false>"),   # C5
                     ("tu_bigh_rerank", "k_maxsim_stream_bigh<0, 1, 2, 8, 1, false, 1, false, false, false, true, false>"),    # dep768 (token-balanced cut)
                     ("tu_bigh_rerank", "k_maxsim_stream_bigh<0, 1, 2, 12, 1, false, 1, false, false, false, false, true>"),   # mv768 (16-row query image)
                     ("tu_allpairs", "k_maxsim_allpairs<2, 3, 3, true>")):                     # training forward
        assert name in want[tu], name
        assert want[tu][name]["scratch"] == 0


def test_the_guard_trips_on_a_moved_kernel(tables):
    """The comparison itself: a kernel that loses a wave per SIMD, gains a granule row, or spills must be reported."""
    want, got = tables
    name = "k_maxsim_stream<0, 0, 4, 1, 0, 48, false, false, false>"
    for field, delta in (("vgpr", 64), ("scratch", 16), ("waves_per_simd", -1), ("lds_static", 1024)):
        moved = json.loads(json.dumps(got))
        moved["tu_stream"][name][field] += delta
        assert any(name in p for p in problems(want, moved)), field
