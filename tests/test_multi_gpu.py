"""d = 1 on several GPUs from C++ (swa_multi_*, swarm_amd/csrc/multi.hip): one context + host thread per
rank inside the library, ownership sharding, link lists all-gathered, fastidious shards combined with MIN.  A
one-GPU box runs it with several ranks on device 0 (the exchange then uses device-to-device copies) and with ONE
rank through RCCL (communicator of size 1: the ncclBroadcast / ncclAllReduce calls themselves).  Whatever the
device list, the result must be the single-GPU result, byte for byte."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import support as S
from swarm_amd import Context, D1Clusters, HostDb, MultiContext

pytestmark = pytest.mark.gpu
BIN = S.ROOT / "swarm_amd" / "bin" / "swarm"


@pytest.fixture(scope="module")
def light_set(tmp_path_factory):
    fa = tmp_path_factory.mktemp("multi") / "in.fa"
    S.gen_fasta(fa, 60000, 150, 61, 1, 0.3)
    hdb = HostDb(fa)
    ctx = Context(0)
    ctx.upload_hostdb(hdb)
    assert ctx.d1_index_build() is False
    off, nb = ctx.d1_network()
    cl = D1Clusters(hdb, off, nb)
    flags, stats = cl.light_flags(3)
    graft, counters = ctx.d1_fastidious(flags, stats[2], 16)
    ctx.close()
    return fa, hdb, off, nb, flags, stats, graft, counters


@pytest.mark.parametrize("devices,rccl,build", [([0, 0], False, "routed"), ([0, 0, 0], False, "routed"), ([0, 0, 0], False, "streamed"),
                                                ([0], True, "routed")])
def test_multi_matches_single(light_set, monkeypatch, devices, rccl, build):
    """build: how the ranks get the members of their anchor groups — routed: every rank keys its slice and the ids travel
    to the owners (swa_d1_route_slice / swa_d1_index_build_routed); streamed: every rank walks the whole database."""
    fa, hdb, off, nb, flags, stats, graft, counters = light_set
    if rccl:
        monkeypatch.setenv("SWARM_AMD_FORCE_RCCL", "1")
    monkeypatch.setenv("SWARM_AMD_MULTI_BUILD", build)
    m = MultiContext(devices)
    assert m.uses_rccl() is rccl
    m.upload_hostdb(hdb)
    for ncb in (False, True):
        moff, mnb = m.d1_network(ncb)
        if not ncb:
            assert np.array_equal(moff, off) and np.array_equal(mnb, nb)
    moff, mnb = m.d1_network(False)
    assert np.array_equal(moff, off) and np.array_equal(mnb, nb)
    mgraft, mcounters = m.d1_fastidious(flags, stats[2], 16)
    assert np.array_equal(mgraft, graft)
    assert [int(x) for x in mcounters[:5]] == [int(x) for x in counters[:5]]
    m.close()


@pytest.mark.parametrize("args", [["-d", "1"], ["-d", "1", "-f"], ["-d", "1", "-n"]])
def test_cli_with_several_ranks_is_byte_identical(light_set, tmp_path, args):
    fa = light_set[0]
    outs = {}
    for tag, env in (("one", {}), ("two", {"SWARM_AMD_DEVICES": "0,0"}), ("three", {"SWARM_AMD_DEVICES": "0,0,0"})):
        cmd = [str(BIN)] + args + ["-o", str(tmp_path / f"{tag}.o"), "-s", str(tmp_path / f"{tag}.s"), "-i", str(tmp_path / f"{tag}.i"),
                                   "-j", str(tmp_path / f"{tag}.j"), "-l", str(tmp_path / f"{tag}.log"), str(fa)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs[tag] = tmp_path
    for tag in ("two", "three"):
        for suffix in "osij":
            assert filecmp.cmp(tmp_path / f"one.{suffix}", tmp_path / f"{tag}.{suffix}", shallow=False), (tag, suffix)
        keep = lambda p: [ln for ln in p.read_text().splitlines() if ln.startswith(("Number", "Largest", "Max", "Got", "Made", "Heavy var", "Generated"))]
        assert keep(tmp_path / "one.log") == keep(tmp_path / f"{tag}.log")


def test_multi_reports_duplicates(tmp_path):
    fa = tmp_path / "dup.fa"
    seq = "ACGTTGCAAGCTTAGCGATCGGATCCATGCAAGTCTAGCTAGGCTAACGTACGATCGATCGTAGCTAGCTAGCATCGATCAGCTACGACTAGCATCAGCTAC"
    fa.write_text(f">a_3\n{seq}\n>b_2\n{seq}\n>c_1\n{seq[:-1]}G\n")
    r = subprocess.run([str(BIN), "-d", "1", str(fa)], capture_output=True, text=True, env=dict(os.environ, SWARM_AMD_DEVICES="0,0"))
    assert r.returncode == 1 and "some fasta entries have identical sequences" in r.stderr


# ---- d >= 2 on several ranks (swa_multi_dn_graph: the bulk graph divided by ownership of window groups) ----------

G = S.GOLDEN
FLAG = {"o": "-o", "s": "-s", "i": "-i", "w": "-w", "u": "-u"}


@pytest.mark.parametrize("name", ["d2_small", "d3_400", "d5_ties", "d8_16bit"])
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_cli_dn_with_several_ranks_matches_reference_files(tmp_path, name, devices):
    """The golden d >= 2 cases (files written by the unmodified reference) through 2 and 3 ranks on device 0 (graph route
    where every sequence has room for d + 1 windows; otherwise rank 0 clusters alone with the fused scan)."""
    args = (G / f"{name}.args").read_text().split()
    kept = [k for k in FLAG if (G / f"{name}.{k}").exists()]
    cmd = [str(BIN)] + args
    for k in kept:
        cmd += [FLAG[k], str(tmp_path / k)]
    cmd += ["-l", str(tmp_path / "log"), str(G / f"{name}.fasta")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, SWARM_AMD_DEVICES=devices))
    assert r.returncode == 0, r.stderr
    for k in kept:
        assert filecmp.cmp(tmp_path / k, G / f"{name}.{k}", shallow=False), k


@pytest.mark.parametrize("ncb", [False, True])
def test_multi_dn_graph_equals_single(tmp_path, ncb):
    """The whole graph of a 25 k random set (d = 3, up to 3 edits per amplicon, tied abundances among them) from 2 and
    3 ranks, entry for entry the single-GPU graph; and RCCL with one rank (the send / receive branch is then empty,
    the communicator and the sort on rank 0 are exercised)."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 25000, 120, 77, 3)
    hdb = HostDb(fa)
    want = None
    for devices in ([0], [0, 0], [0, 0, 0]):
        m = MultiContext(devices)
        m.upload_hostdb(hdb)
        got = m.dn_graph(3, ncb)
        assert got is not None
        if want is None:
            want = got
            assert len(got[1]) > 1000
        else:
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
        m.close()
    # ... and through the command line: one GPU against three ranks
    one = subprocess.run([str(BIN), "-d", "3"] + (["-n"] if ncb else []) + ["-o", str(tmp_path / "one.o"), "-i", str(tmp_path / "one.i"), str(fa)],
                         capture_output=True, text=True)
    two = subprocess.run([str(BIN), "-d", "3"] + (["-n"] if ncb else []) + ["-o", str(tmp_path / "two.o"), "-i", str(tmp_path / "two.i"), str(fa)],
                         capture_output=True, text=True, env=dict(os.environ, SWARM_AMD_DEVICES="0,0,0"))
    assert one.returncode == 0 and two.returncode == 0, one.stderr + two.stderr
    assert filecmp.cmp(tmp_path / "one.o", tmp_path / "two.o", shallow=False)
    assert filecmp.cmp(tmp_path / "one.i", tmp_path / "two.i", shallow=False)


def _fasta_of(db, path):
    """a database built in memory (tests/test_d1_gpu.py's hard sets) as a FASTA file for the production reader"""
    out = []
    for i in range(db.n):
        w = db.words(i)
        s = "".join("ACGT"[int((w[p >> 5] >> np.uint64((p & 31) * 2)) & np.uint64(3))] for p in range(int(db.seqlen[i])))
        out.append(f">h{i}_{int(db.abundance[i])}\n{s}\n")
    path.write_text("".join(out))


@pytest.mark.parametrize("which", ["flanks", "giant_groups", "length_mix"])
@pytest.mark.parametrize("build", ["routed", "streamed"])
def test_multi_on_sets_that_leave_the_friendly_route(tmp_path, monkeypatch, which, build):
    """Separate contexts per rank (MultiContext: per-context state cannot leak between ranks as it can when one context
    plays them in turn) on the sets whose index build takes decisions: conserved flanks (the anchor windows move — the same
    way on every rank, or the ranks would divide the pairs differently: ADVICE r02), groups beyond the pair kernels' limit,
    sequences around the anchoring thresholds.  Three ranks on device 0 against one GPU, entry for entry."""
    import test_d1_gpu as T
    fa = tmp_path / "in.fa"
    if which == "flanks":
        T._conserved_flank_set(fa, 20000, 77)
    else:
        _fasta_of(T._giant_group_db() if which == "giant_groups" else T._length_mix_db(), fa)
    hdb = HostDb(fa)
    ctx = Context(0)
    ctx.upload_hostdb(hdb)
    assert ctx.d1_index_build() is False
    off, nb = ctx.d1_network()
    ctx.close()
    assert len(nb) > 1000
    monkeypatch.setenv("SWARM_AMD_MULTI_BUILD", build)
    m = MultiContext([0, 0, 0])
    m.upload_hostdb(hdb)
    for ncb in (False, True, False):
        moff, mnb = m.d1_network(ncb)
        if not ncb:
            assert np.array_equal(moff, off) and np.array_equal(mnb, nb)
    m.close()
