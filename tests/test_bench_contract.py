"""The line bench.py prints, as the driver's contract reads it — checked on the line of the round's last build
(profiles/r04/bench_default_final_lease_v.json: `python bench.py`, N = 1, on an MI355X) and on bench.py itself without a GPU:
metric / unit are BASELINE.json's, the workload is named in config (no model keys), vs_baseline is null (BASELINE.md holds
no published number for this metric), `roofline` is the dominant kernel's with frac = achieved / peak and the peak the
guide's 8 TB/s, `cpu_baseline` names the reference, its cores and its sample."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LINE = ROOT / "profiles" / "r04" / "bench_default_final_lease_v.json"


@pytest.fixture(scope="module")
def line():
    return json.loads(LINE.read_text().strip().splitlines()[-1])


def test_headline_fields(line):
    base = json.loads((ROOT / "BASELINE.json").read_text())
    # (BASELINE.json names two metrics in one string: "amplicons/sec clustered (d=1, 10M×150bp); edit-dist comparisons/sec (d=2)" —
    # the line carries the first, its shape in config.workload; the second is config.configs3's)
    assert line["metric"] == "amplicons/sec clustered (d=1)" and base["metric"].startswith("amplicons/sec clustered (d=1")
    assert line["unit"] == "amplicons/s"
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["dtype"] == "u64"
    assert line["steps"] > 0 and line["warmup"] >= 0
    # value = whole-job throughput over the timed steps
    n = line["config"]["per_gpu_queries"]
    assert abs(line["value"] - n / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "10000000" in line["config"]["workload"]


def test_roofline_object(line):
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is not None and r["traffic"] > 0
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    # the dominant kernel is the slowest of the groups that are one kernel
    single = {g: v["ms"] for g, v in r["kernels"].items() if not g.startswith("partition")}
    dom = max(single, key=single.get)
    assert abs(single[dom] - r["avg_kernel_ms"]) < 1e-9
    assert 0.0 < r["step"]["frac_of_hbm_peak"] < 1.0
    assert r["ceilings"]["stream_copy"]["rate"] > 5500.0          # (the copy ceiling the guide names, reached by the microbenchmark)


def test_cpu_baseline_object(line):
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "amplicons/s"
    assert "reference swarm" in c["sample"] and "1000000" in c["sample"]
    assert line["value"] / c["value"] > 100.0                    # (reported beside, never as the target)


def test_other_configs_are_reported_beside(line):
    cfg = line["config"]
    for name in ("configs1", "heavy_tail", "d1_x400", "d1_x460", "mixed_lengths", "configs2", "configs3", "whole_run"):
        assert name in cfg and "error" not in cfg[name], name
    assert cfg["whole_run"]["n1000000"]["output_md5_equals_reference"] is True
    assert cfg["configs1"]["ms_per_step"] < 0.55                  # 
    assert cfg["heavy_tail"]["ms_per_step"] < 2.0 * line["ms_per_step"]   # (item 3)
    assert cfg["mixed_lengths"]["ms_per_step"] < 1.5 * line["ms_per_step"]   # (item 2)
    assert cfg["configs3"]["clustering_seconds"] <= 0.065        # (item 7: <= 60 ms; 56-64 across boxes)


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
