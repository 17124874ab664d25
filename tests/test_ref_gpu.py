"""The boundary, proven by construction: oracle/_ref/swarm_gpu is the REFERENCE program (its own main, option parsing,
FASTA reader, greedy clustering, writers — compiled from /root/reference by `make -C oracle ref-gpu`) with the four hot
seams of INTEGRATION.md bound to libswarm_amd.so: B1 (network), B2 (fastidious), B3 (q-gram), B4 (alignment scan).
Run over the golden cases it must reproduce the unmodified reference's output files byte for byte."""
import filecmp
import subprocess

import pytest

import support as S

pytestmark = pytest.mark.gpu
G = S.GOLDEN
BIN = S.ROOT / "oracle" / "_ref" / "swarm_gpu"
FLAG = {"o": "-o", "s": "-s", "i": "-i", "w": "-w", "j": "-j", "u": "-u"}
CASES = ["d1_1k", "d1_nobreak", "d1_mothur", "d1_short", "d1_usearch", "d1_fastidious", "d1_fastidious_b10_y8", "d1_uclust",
         "d2_small", "d3_400", "d5_ties", "d8_16bit"]


@pytest.mark.skipif(not BIN.exists(), reason="oracle/_ref/swarm_gpu not built (make -C oracle ref-gpu, needs /root/reference)")
@pytest.mark.parametrize("name", CASES)
def test_reference_with_bound_seams_matches_reference(tmp_path, name):
    args = (G / f"{name}.args").read_text().split()
    kept = [k for k in FLAG if (G / f"{name}.{k}").exists()]
    cmd = [str(BIN)] + args
    for k in kept:
        cmd += [FLAG[k], str(tmp_path / k)]
    cmd += ["-l", str(tmp_path / "log"), str(G / f"{name}.fasta")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for k in kept:
        assert filecmp.cmp(tmp_path / k, G / f"{name}.{k}", shallow=False), k
    if (G / f"{name}.log").exists():
        log = (tmp_path / "log").read_text()
        for line in (G / f"{name}.log").read_text().splitlines():
            assert line in log, line


@pytest.mark.skipif(not BIN.exists(), reason="oracle/_ref/swarm_gpu not built")
def test_reference_with_bound_seams_on_a_larger_set(tmp_path):
    """60 k amplicons with light swarms, -f: both binaries, every file."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 60000, 150, 311, 1, 0.3)
    outs = "osiwj"
    ref_cmd, gpu_cmd = ["-d", "1", "-f"], [str(BIN), "-d", "1", "-f"]
    for k in outs:
        ref_cmd += [FLAG[k], str(tmp_path / f"r{k}")]
        gpu_cmd += [FLAG[k], str(tmp_path / f"g{k}")]
    r = S.run_ref_swarm(ref_cmd + ["-l", "/dev/null", str(fa)])
    assert r.returncode == 0, r.stderr
    g = subprocess.run(gpu_cmd + ["-l", "/dev/null", str(fa)], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr
    for k in outs:
        assert filecmp.cmp(tmp_path / f"r{k}", tmp_path / f"g{k}", shallow=False), k
