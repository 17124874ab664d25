"""d = 0 (dereplication).

CPU: the oracle's restatement of src/derep.cc:276-354 drives the host clustering + writers and
every output file must equal the reference's own (tests/golden/d0_*).
GPU: swa_derep (HIP) must return exactly the oracle's array, including when distinct sequences
are forced onto the same table key."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

import support as S
from swarm_amd import D0Clusters, HostDb

G = S.GOLDEN


def _first_identical_bruteforce(db) -> np.ndarray:
    seen = {}
    out = np.zeros(db.n, dtype=np.uint32)
    for i in range(db.n):
        key = (int(db.seqlen[i]), db.seqs[int(db.seq_off[i]):int(db.seq_off[i + 1])].tobytes())
        out[i] = seen.setdefault(key, i)
    return out


def test_oracle_derep_is_first_occurrence():
    db = S.db_from_fasta(G / "d0_derep.fasta")
    got = S.oracle_derep(db)
    assert np.array_equal(got, _first_identical_bruteforce(db))
    assert (got != np.arange(db.n)).sum() == db.n - 120       # 120 clusters in the reference's log


@pytest.mark.parametrize("name,mothur,append", [("d0_derep", False, 0), ("d0_mothur", True, 1)])
def test_host_clustering_reproduces_reference_files(tmp_path, name, mothur, append):
    fa = G / "d0_derep.fasta"
    hdb = HostDb(fa, append_abundance=append)
    cl = D0Clusters(hdb, S.oracle_derep(S.db_from_fasta(fa)))
    log = (G / f"{name}.log").read_text()
    s = cl.summary()
    assert f"Number of swarms:  {s['swarms']}\n" in log
    assert f"Largest swarm:     {s['largest']}\n" in log
    assert f"Heaviest swarm:    {s['heaviest']}\n" in log
    cl.write_swarms(tmp_path / "o", mothur=mothur, append_abundance=append)
    assert filecmp.cmp(tmp_path / "o", G / f"{name}.o", shallow=False)
    if name == "d0_derep":
        cl.write_seeds(tmp_path / "w")
        cl.write_stats(tmp_path / "s")
        cl.write_structure(tmp_path / "i")
        cl.write_uclust(tmp_path / "u")
        for k in "wsiu":
            assert filecmp.cmp(tmp_path / k, G / f"{name}.{k}", shallow=False), k


def test_host_clustering_rejects_inconsistent_array():
    hdb = HostDb(G / "d0_derep.fasta")
    bad = np.arange(hdb.n, dtype=np.uint32)
    bad[3] = 7                                               # points forward
    with pytest.raises(Exception):
        D0Clusters(hdb, bad)
    bad = np.arange(hdb.n, dtype=np.uint32)
    bad[5] = 4
    bad[9] = 5                                               # points at a non-seed
    with pytest.raises(Exception):
        D0Clusters(hdb, bad)


def _make_reads(path, n_unique, length, seed, max_copies):
    rng = np.random.default_rng(seed)
    uniq = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(max(1, length - 3), length + 4)))) for _ in range(n_unique)]
    entries = []
    for k, s in enumerate(uniq):
        for c in range(int(rng.integers(1, max_copies + 1))):
            entries.append((f"u{k}c{c}_{int(rng.integers(1, 50))}", s))
    with open(path, "w") as fh:
        for j in rng.permutation(len(entries)):
            fh.write(f">{entries[j][0]}\n{entries[j][1]}\n")


@pytest.mark.gpu
@pytest.mark.parametrize("n_unique,length,seed,max_copies", [(3000, 150, 1, 6), (500, 20, 2, 40), (1, 64, 3, 5000),
                                                             (2000, 400, 4, 3)])
def test_gpu_derep_matches_oracle(gpu_ctx, tmp_path, n_unique, length, seed, max_copies):
    from swarm_amd import derep
    fa = tmp_path / "reads.fa"
    _make_reads(fa, n_unique, length, seed, max_copies)
    hdb = HostDb(fa)
    gpu_ctx.upload_hostdb(hdb)
    got = derep(gpu_ctx)
    want = S.oracle_derep(S.db_from_fasta(fa))
    assert np.array_equal(got, want)
    # the d = 1 index can be rebuilt on the same context afterwards (and reports the duplicates)
    if max_copies > 1 and hdb.n > n_unique:
        assert gpu_ctx.d1_has_duplicates() is True            # (index build or network, whichever meets them: include/swarm_amd.h)


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [1, 4, 9])
def test_gpu_derep_survives_key_collisions(tmp_path, bits):
    """SWA_DEREP_KEY_BITS narrows the table key so that different sequences share a slot: the
    verification + re-key rounds must still give the oracle's answer (own process: env hook)."""
    fa = tmp_path / "reads.fa"
    _make_reads(fa, 700, 60, 9, 5)
    want = S.oracle_derep(S.db_from_fasta(fa))
    np.save(tmp_path / "want.npy", want)
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(S.ROOT)!r})\n"
        "from swarm_amd import Context, HostDb, derep\n"
        f"hdb = HostDb({str(fa)!r}); ctx = Context(0); ctx.upload_hostdb(hdb)\n"
        f"assert np.array_equal(derep(ctx), np.load({str(tmp_path / 'want.npy')!r})), 'derep differs'\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, SWA_DEREP_KEY_BITS=str(bits)), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
