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


@pytest.mark.skipif(not S.have_reference(), reason="compiled reference not available on this box")
@pytest.mark.parametrize("n,length,seed,light,args", [
    (60000, 150, 301, 0.0, ["-d", "1"]),
    (60000, 150, 302, 0.3, ["-d", "1", "-f"]),
    (40000, 80, 303, 0.3, ["-d", "1", "-f", "-b", "5", "-y", "10"]),
    (30000, 250, 304, 0.0, ["-d", "1", "-n"]),
    (20000, 100, 305, 0.0, ["-d", "1", "-r"]),
])
def test_cli_against_reference_binary(tmp_path, n, length, seed, light, args):
    """Random sets, both binaries, every output file byte-compared (the reference binary is the
    one oracle/Makefile builds into oracle/_ref; it travels with the repository snapshot)."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, length, seed, 1, light)
    outs = "osiwj" if "-r" not in args else "o"
    ref_cmd, our_cmd = list(args), [str(BIN)] + list(args)
    for k in outs:
        ref_cmd += [FLAG[k], str(tmp_path / f"r{k}")]
        our_cmd += [FLAG[k], str(tmp_path / f"g{k}")]
    r = S.run_ref_swarm(ref_cmd + ["-l", "/dev/null", str(fa)])
    assert r.returncode == 0, r.stderr
    g = subprocess.run(our_cmd + ["-l", "/dev/null", str(fa)], capture_output=True, text=True)
    assert g.returncode == 0, g.stderr
    for k in outs:
        assert filecmp.cmp(tmp_path / f"r{k}", tmp_path / f"g{k}", shallow=False), k
