"""The line bench.py prints, as the driver's contract reads it.  Without a GPU: the command line and the byte model.  On
the GPU box (`-m gpu`): a line PRODUCED there by this build — `bench.py --no-extras` on 1 M amplicons — checked for the
contract's schema: metric / unit are BASELINE.json's, the workload is named in config (no model keys), vs_baseline is
null (BASELINE.md holds no published number for this metric), `roofline` carries bound / achieved / peak / unit / frac /
traffic with frac = achieved / peak.  (Round 4 asserted thresholds on a committed JSON file here: a ledger, not a test.)"""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def check_headline(line: dict, n: int, steps: int, warmup: int) -> None:
    base = json.loads((ROOT / "BASELINE.json").read_text())
    # (BASELINE.json names two metrics in one string: "amplicons/sec clustered (d=1, 10M×150bp); edit-dist comparisons/sec (d=2)" —
    # the line carries the first, its shape in config.workload; the second is the top-level configs3 object of a full run)
    assert line["metric"] == "amplicons/sec clustered (d=1)" and base["metric"].startswith("amplicons/sec clustered (d=1")
    assert line["unit"] == "amplicons/s"
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "u64"
    assert line["steps"] == steps and line["warmup"] == warmup
    assert line["config"]["per_gpu_queries"] == n
    assert abs(line["value"] - n / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6   # whole-job throughput over the timed steps
    assert "workload" in line["config"] and "model" not in line["config"]
    assert str(n) in line["config"]["workload"]


def check_roofline(r: dict) -> None:
    assert r["bound"] in ("hbm", "valu")
    assert (r["unit"], r["peak"]) in (("GB/s", 8000.0), ("wave-instr/s", 256 * 4 * 0.5 * 2.4e9))
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert "traffic" in r
    single = {g: v["ms"] for g, v in r["kernels"].items() if not g.startswith("partition")}
    dom = max(single, key=single.get)
    assert r["dominant_group"] == dom and abs(single[dom] - r["avg_kernel_ms"]) < 1e-9
    assert r["largest_group_including_multi_launch"] in r["kernels"]
    assert 0.0 < r["step"]["frac_of_hbm_peak"] < 1.0
    if r["bound"] == "hbm" and r["traffic"] is None:           # no counter passes: algorithmic bytes / launch duration
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    if r["bound"] == "valu":                                    # the HBM view of the same kernel stays beside it, on counter bytes
        assert r["hbm"]["traffic"] > 0 and abs(r["hbm"]["frac"] - r["hbm"]["achieved"] / 8000.0) < 1e-9


def check_compact(line: dict) -> None:
    """The printed line: the contract's roofline keys as scalars (the driver's record keeps scalars and short strings)."""
    r = line["roofline"]
    assert r["bound"] in ("hbm", "valu")
    assert (r["unit"], r["peak"]) in (("GB/s", 8000.0), ("wave-instr/s", 256 * 4 * 0.5 * 2.4e9))
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    assert "traffic" in r and r["avg_kernel_ms"] > 0
    for k, v in line.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                assert not isinstance(vv, str) or len(vv) <= 128, (k, kk)


def test_a_full_shape_line_stays_below_eight_kilobytes():
    """the line of a FULL default run (every extra, counters, ceilings — round 5's closing line, 21.7 KB as it
    was printed then) through the trimming the bench now prints with."""
    sys.path.insert(0, str(ROOT))
    import bench
    full = json.loads((ROOT / "profiles" / "r05" / "bench_default_closing.json").read_text())
    assert len(json.dumps(full)) > 20000
    # (what this round adds to the full shape: baselines of configs[2] / configs[3], percentiles of the whole run)
    base = {"value": 1.0e5, "unit": "amplicons/s", "cores": 16, "kind": "reference", "seconds": 12.3, "sample": "x" * 300}
    full["configs2"]["cpu_baseline"] = dict(base)
    full["configs3"]["cpu_baseline"] = dict(base)
    full["whole_run"]["n10000000"].update({"median_s": 0.21, "p95_s": 0.24, "max_s": 0.3, "runs": 40})
    line = bench.compact_line(full, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT == 8192 and len(text) < 6000
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k]
    check_headline(line, 10_000_000, 10, 2)
    check_compact(line)
    assert line["roofline"]["traffic"] == full["roofline"]["traffic"] and line["roofline"]["hbm"]["frac"] > 0
    assert line["roofline"]["step"]["minimum_bytes"] > 0 and line["roofline"]["step_traffic_over_minimum"] > 1
    assert line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == 16 and line["cpu_baseline"]["value"] > 0
    assert line["configs2"]["cpu_baseline"]["kind"] == "reference" and line["configs3"]["qgram_comparisons_per_s"] > 0
    assert line["whole_run"]["n10000000"]["p95_s"] == 0.24 and line["detail"] == "bench_detail.json"
    assert set(line["config"]) <= {"workload", "per_gpu_queries", "db_amplicons", "step", "route", "sharding", "build", "neighbour_links"}


@pytest.mark.gpu
def test_the_line_this_build_prints_on_the_gpu():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--per-gpu", "1000000"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    assert len(lines[0]) < 8192, "the driver must be able to parse the line (r05's 21.7 KB line left BENCH_r05.parsed null)"
    line = json.loads(lines[0])
    check_headline(line, 1_000_000, 3, 1)
    check_compact(line)
    detail = json.loads((ROOT / line["detail"]).read_text())    # everything the run measured: the side file the line names
    assert detail["value"] == line["value"] and detail["ms_per_step"] == line["ms_per_step"]
    check_roofline(detail["roofline"])
    assert "cpu_baseline" not in line                           # (--no-extras: nothing beside the headline)


def test_bench_defaults_without_a_gpu():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--per-gpu", "--extras"):
        assert flag in r.stdout
    sys.path.insert(0, str(ROOT))
    import bench
    m = bench.step_byte_model(10_000_000, 17_000_000, 1, 2)
    assert set(m) == {"keys", "partition_keys", "partition_links", "groups", "pairs0", "pairs1", "csr_rows"}
    assert m["keys"] == 80 * 10_000_000 and m["groups"] == 24 * 10_000_000
    assert all(v > 0 for v in m.values())
    assert bench.VALU_PEAK_WAVE_INSTR_S == 256 * 4 * 0.5 * 2.4e9 and bench.HBM_PEAK_GBS == 8000.0
