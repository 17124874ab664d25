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


@pytest.mark.gpu
def test_the_line_this_build_prints_on_the_gpu():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--per-gpu", "1000000"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line"
    line = json.loads(lines[0])
    check_headline(line, 1_000_000, 3, 1)
    check_roofline(line["roofline"])
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
    assert m["keys"] == 84 * 10_000_000 and m["groups"] == 28 * 10_000_000
    assert all(v > 0 for v in m.values())
    assert bench.VALU_PEAK_WAVE_INSTR_S == 256 * 4 * 0.5 * 2.4e9 and bench.HBM_PEAK_GBS == 8000.0
