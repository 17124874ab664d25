"""Command-line validation of the drop-in front end (no GPU needed: every case fails before
any device call).  Messages are the reference's (src/swarm.cc:486-630)."""
import subprocess

import pytest

import support as S

BIN = S.ROOT / "swarm_amd" / "bin" / "swarm"

CASES = [
    (["-d", "1", "-b", "3"], "Option -b or --boundary specified without -f or --fastidious."),
    (["-d", "3", "-f"], "Fastidious mode (specified with -f or --fastidious) only works"),
    (["-d", "1", "-m", "3"], "Option -m or --match-reward specified when d < 2."),
    (["-d", "256"], "Illegal number of differences specified with -d or --differences, must be in the range 0 to"),
    (["-t", "0"], "must be in the range 1 to 512."),
    (["-d", "2", "-j", "x"], "A network file can only written when d = 1."),
    (["-d", "1", "-d", "2"], "Option -d or --differences specified more than once."),
    (["-d", "1x"], "Invalid numeric argument for option -d or --differences."),
    (["-f", "-y", "1"], "must be in the range 2 to 64."),
    (["-a", "0"], "must be at least 1."),
    (["-d", "1", "-x"], "Option --disable-sse3 or -x has no effect when d < 2"),
]


@pytest.mark.parametrize("args,msg", CASES)
def test_rejected_like_the_reference(args, msg):
    if not BIN.exists():
        subprocess.run(["make", "-C", str(S.ROOT / "swarm_amd" / "csrc"), "-j4"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([str(BIN)] + args + ["/nonexistent.fa"], capture_output=True, text=True)
    assert r.returncode == 1
    assert msg in r.stderr
    if S.have_reference():
        ref = subprocess.run([str(S.ref_swarm_bin())] + args + ["/nonexistent.fa"], capture_output=True)
        assert ref.returncode == 1 and msg.encode() in ref.stderr


def test_help_and_version_exit_zero():
    for flag in ("-h", "-v"):
        r = subprocess.run([str(BIN), flag], capture_output=True, text=True)
        assert r.returncode == 0 and "Swarm" in r.stderr


def test_fasta_errors_reach_stderr(tmp_path):
    fa = tmp_path / "bad.fa"
    fa.write_text(">a_1\nACGN\n")
    r = subprocess.run([str(BIN), "-d", "1", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "Illegal character 'N' in sequence on line 2." in r.stderr
    r = subprocess.run([str(BIN), "-d", "1", str(tmp_path / "missing.fa")], capture_output=True, text=True)
    assert r.returncode == 1 and "Unable to open input data file" in r.stderr
