"""The drop-in command line (swarm_amd/bin/swarm) end to end on the GPU: same arguments as the
reference run that produced each golden fixture, every output file byte-compared, and the
deterministic log lines compared."""
import filecmp
import subprocess

import pytest

import support as S

pytestmark = pytest.mark.gpu
G = S.GOLDEN
BIN = S.ROOT / "swarm_amd" / "bin" / "swarm"
FLAG = {"o": "-o", "s": "-s", "i": "-i", "w": "-w", "j": "-j", "u": "-u"}
CASES = ["d1_1k", "d1_nobreak", "d1_mothur", "d1_short", "d1_usearch", "d1_fastidious", "d1_fastidious_b10_y8",
         "d1_uclust", "d2_small", "d3_400", "d5_ties", "d8_16bit", "d0_derep", "d0_mothur"]
CASES += [f"tiny_{t}_d{d}" for t in ("one", "mix", "empty") for d in (0, 1, 2)]


@pytest.mark.parametrize("name", CASES)
def test_cli_matches_reference_files(tmp_path, name):
    args = (G / f"{name}.args").read_text().split()
    kept = [k for k in FLAG if (G / f"{name}.{k}").exists()]
    cmd = [str(BIN)] + args
    for k in kept:
        cmd += [FLAG[k], str(tmp_path / k)]
    fasta = G / ("d0_derep.fasta" if name.startswith("d0_") else f"{name}.fasta")
    if name.startswith("tiny_"):
        fasta = G / (name.rsplit("_d", 1)[0] + ".fasta")
    cmd += ["-l", str(tmp_path / "log"), str(fasta)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for k in kept:
        assert filecmp.cmp(tmp_path / k, G / f"{name}.{k}", shallow=False), k
    if (G / f"{name}.log").exists():
        log = (tmp_path / "log").read_text()
        for line in (G / f"{name}.log").read_text().splitlines():
            assert line in log, line


def test_cli_stdout_and_duplicates(tmp_path):
    r = subprocess.run([str(BIN), "-d", "1", str(G / "d1_short.fasta")], capture_output=True)
    assert r.returncode == 0 and r.stdout == (G / "d1_short.o").read_bytes()
    dup = tmp_path / "dup.fa"
    dup.write_text(">a_3\nACGTACGTACGTAAAC\n>b_2\nACGTACGTACGTAAAC\n")
    r = subprocess.run([str(BIN), "-d", "1", str(dup)], capture_output=True, text=True)
    assert r.returncode == 1 and "some fasta entries have identical sequences" in r.stderr
    r = subprocess.run([str(BIN), "-d", "2", str(dup)], capture_output=True, text=True)
    assert r.returncode == 1 and "some fasta entries have identical sequences" in r.stderr
