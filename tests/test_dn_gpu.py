"""d >= 2 end to end on the GPU: FASTA -> HostDb -> fused scan steps (B3 + B4 in HBM) driven by
the host greedy loop -> writers, byte-compared with the reference's own output files."""
import filecmp

import pytest

import support as S
from swarm_amd import DnClusters, HostDb

pytestmark = pytest.mark.gpu
G = S.GOLDEN


@pytest.mark.parametrize("name", ["d2_small", "d3_400", "d5_ties", "d8_16bit"])
def test_dn_outputs_byte_identical(gpu_ctx, tmp_path, name):
    args = (G / f"{name}.args").read_text().split()
    d = int(args[args.index("-d") + 1])
    hdb = HostDb(G / f"{name}.fasta", check_duplicate_sequences=True)
    gpu_ctx.upload_hostdb(hdb)
    cl = DnClusters(gpu_ctx, hdb, d)
    cl.write_swarms(tmp_path / "o")
    assert filecmp.cmp(tmp_path / "o", G / f"{name}.o", shallow=False)
    for suffix, writer in (("s", cl.write_stats), ("i", cl.write_structure), ("w", cl.write_seeds), ("u", cl.write_uclust)):
        if (G / f"{name}.{suffix}").exists():
            writer(tmp_path / suffix)
            assert filecmp.cmp(tmp_path / suffix, G / f"{name}.{suffix}", shallow=False), suffix
    log = (G / f"{name}.log").read_text()
    s = cl.summary()
    assert f"Number of swarms:  {s['swarms']}\n" in log
    assert f"Largest swarm:     {s['largest']}\n" in log
    assert f"Max generations:   {s['maxgen']}\n" in log
    t = cl.scan_totals()
    assert t["qgram_comparisons"] > 0 and t["aligned_pairs"] > 0


@pytest.mark.skipif(not S.have_reference(), reason="compiled reference not available on this box")
@pytest.mark.parametrize("n,length,d,edits,extra", [(3000, 150, 2, 2, []), (2000, 400, 3, 3, []), (1500, 100, 3, 3, ["-n"]),
                                                     # bigger sets: long generation chains (radius outgrows the candidate
                                                     # list's bound), many swarms, the wavefront kernel at d = 2 and 3, the
                                                     # banded one at d = 4
                                                     (25000, 200, 2, 2, []), (15000, 300, 3, 3, []), (8000, 120, 4, 4, []),
                                                     (6000, 60, 3, 3, ["-n"])])
def test_dn_against_reference_binary(gpu_ctx, tmp_path, n, length, d, edits, extra):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, length, 900 + d, edits)
    r = S.run_ref_swarm(["-d", d, "-o", tmp_path / "ro", "-i", tmp_path / "ri", "-s", tmp_path / "rs", "-l", "/dev/null"]
                        + extra + [fa])
    assert r.returncode == 0, r.stderr
    hdb = HostDb(fa, check_duplicate_sequences=True)
    gpu_ctx.upload_hostdb(hdb)
    cl = DnClusters(gpu_ctx, hdb, d, no_cluster_breaking="-n" in extra)
    cl.write_swarms(tmp_path / "o")
    cl.write_structure(tmp_path / "i")
    cl.write_stats(tmp_path / "s")
    for suffix in "ois":
        assert filecmp.cmp(tmp_path / suffix, tmp_path / ("r" + suffix), shallow=False), suffix
