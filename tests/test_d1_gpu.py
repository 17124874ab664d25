"""B1 parity on the GPU: swa_d1_* (HIP, through the C ABI) vs the oracle on the same inputs."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _upload(ctx, db):
    ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)


def _oracle_sorted_rows(db, ncb=False, first=0, count=None):
    off, nb, dup = S.oracle_d1_network(db, ncb, first, count)
    nb = nb.copy()
    for i in range(len(off) - 1):
        nb[int(off[i]):int(off[i + 1])].sort()
    return off, nb, dup


@pytest.mark.parametrize("n,L,seed,ncb", [(1000, 150, 11, False), (3000, 150, 12, True), (500, 33, 14, False),
                                          (2000, 400, 15, False), (300, 1000, 16, False), (64, 9, 17, True)])
def test_network_matches_oracle(gpu_ctx, tmp_path, n, L, seed, ncb):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, L, seed)
    db = S.db_from_fasta(fa)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_index_build() is False
    # intermediates are bit-exact: Zobrist table, sequence hashes, table size, Bloom bitmap (the debug readers build the
    # database-wide index)
    lib = S.oracle()
    zob = S.oracle_zobrist(db.longest + 2)
    assert np.array_equal(gpu_ctx.d1_debug(2, 4 * (db.longest + 2)), zob)
    want_hash = np.array([lib.orc_zobrist_hash(S._p(zob, S.u64p), S._p(db.words(i), S.u64p), int(db.seqlen[i]))
                          for i in range(db.n)], dtype=np.uint64)
    assert np.array_equal(gpu_ctx.d1_debug(0, db.n), want_hash)
    assert gpu_ctx.d1_table_size() == lib.orc_hashtable_size(db.n)
    want_bloom = S.oracle_d1_bloom(db)
    assert np.array_equal(gpu_ctx.d1_debug(1, len(want_bloom)), want_bloom)
    off, nb = gpu_ctx.d1_network(ncb)
    woff, wnb, _ = _oracle_sorted_rows(db, ncb)
    assert np.array_equal(off, woff)
    assert np.array_equal(nb, wnb)


def test_network_subrange_and_capacity(gpu_ctx, tmp_path):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 5000, 150, 21)
    db = S.db_from_fasta(fa)
    _upload(gpu_ctx, db)
    gpu_ctx.d1_index_build()
    full_off, full_nb = gpu_ctx.d1_network()
    for first, count in [(0, 1), (17, 1000), (4000, 1000), (4999, 1)]:
        off, nb = gpu_ctx.d1_network(False, first, count)
        lo, hi = int(full_off[first]), int(full_off[first + count])
        assert np.array_equal(off, full_off[first:first + count + 1] - full_off[first])
        assert np.array_equal(nb, full_nb[lo:hi])


def test_duplicates_are_reported(gpu_ctx, tmp_path):
    fa = tmp_path / "dup.fa"
    fa.write_text(">a_3\nACGTACGTACGTAAAC\n>b_2\nACGTACGTACGTAAAC\n>c_1\nACGTACGTACGTAAAG\n")
    db = S.db_from_fasta(fa)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_has_duplicates() is True            # (16-nt reads: too short for windows, the table route's own check)


def test_duplicates_are_reported_by_the_call_that_meets_them(gpu_ctx, tmp_path):
    """Round 6: on the pair route identical sequences are met by the NETWORK call's prefix pass (two members of a prefix
    group whose every word agrees), not by the index build (which ran a second hash table over fingerprints for them until
    round 5): swa_d1_index_build is clean, swa_d1_network* returns SWA_E_DUPLICATES — equal lengths only, whatever the
    position of the twin in its group, in groups of every size kind."""
    from swarm_amd.capi import SWA_E_DUPLICATES, SwaError
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 6000, 150, 12)
    recs = S.read_fasta(fa)
    clean = S.build_db(recs)
    _upload(gpu_ctx, clean)
    assert gpu_ctx.d1_index_build() is False
    gpu_ctx.d1_network()                                           # no error
    for victim in (0, 17, 2999, 5999):
        dirty = S.build_db(recs + [(b"twin_1", recs[victim][1])])
        _upload(gpu_ctx, dirty)
        assert gpu_ctx.d1_index_build() is False                   # (nothing here builds a table)
        with pytest.raises(SwaError) as e:
            gpu_ctx.d1_network()
        assert e.value.code == SWA_E_DUPLICATES
        with pytest.raises(SwaError):
            gpu_ctx.d1_network_resident()
    # a sequence and its one-nucleotide-longer sibling whose extra nucleotide is an A (code 0: the packed words agree): not twins
    base = recs[5][1].decode()
    longer = S.build_db(recs + [(b"longer_1", (base + "A").encode())])
    _upload(gpu_ctx, longer)
    assert gpu_ctx.d1_has_duplicates() is False


def test_star_with_many_neighbours(gpu_ctx, tmp_path):
    """One centre with every one of its microvariants present: rows of > 64 and queue overflow."""
    centre = "ACGTTGCAAGCTTAGCGATCGGATCCATGCAAGTCTAGCTAGGCTAACGT"
    lib = S.oracle()
    seqs = {centre}
    for p in range(len(centre)):
        for b in "ACGT":
            if b != centre[p]:
                seqs.add(centre[:p] + b + centre[p + 1:])
            seqs.add(centre[:p] + b + centre[p:])
        seqs.add(centre[:p] + centre[p + 1:])
    for b in "ACGT":
        seqs.add(centre + b)
    seqs = sorted(seqs)
    fa = tmp_path / "star.fa"
    fa.write_text("".join(f">s{i}_{1000 if s == centre else 1}\n{s}\n" for i, s in enumerate(seqs)))
    db = S.db_from_fasta(fa)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_index_build() is False
    off, nb = gpu_ctx.d1_network()
    woff, wnb, _ = _oracle_sorted_rows(db)
    assert int(off[1] - off[0]) == len(seqs) - 1      # the centre sees everybody
    assert np.array_equal(off, woff)
    assert np.array_equal(nb, wnb)


def test_full_size_1m_properties(gpu_ctx, tmp_path):
    """BASELINE config 2 (1M x 150, d=1) through size-independent properties: the network is
    symmetric up to the abundance rule, rows are sorted/unique/self-free, and a random sample
    of rows equals the oracle's."""
    fa = tmp_path / "big.fa"
    S.gen_fasta(fa, 1_000_000, 150, 1)
    db = S.db_from_fasta(fa)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_index_build() is False
    off, nb = gpu_ctx.d1_network(True)       # -n: no abundance rule => symmetric relation
    rows = np.repeat(np.arange(db.n, dtype=np.uint64), np.diff(off).astype(np.int64))
    fwd = (rows << np.uint64(32)) | nb.astype(np.uint64)
    rev = (nb.astype(np.uint64) << np.uint64(32)) | rows
    assert np.array_equal(np.sort(fwd), np.sort(rev))
    assert (rows != nb).all()
    assert (np.diff(fwd) > 0).all()           # ascending and unique within and across rows
    off2, nb2 = gpu_ctx.d1_network(False)
    rows2 = np.repeat(np.arange(db.n, dtype=np.uint64), np.diff(off2).astype(np.int64))
    assert (db.abundance[rows2.astype(np.int64)] >= db.abundance[nb2.astype(np.int64)]).all()
    keep = db.abundance[rows.astype(np.int64)] >= db.abundance[nb.astype(np.int64)]
    assert np.array_equal(fwd[keep], (rows2 << np.uint64(32)) | nb2.astype(np.uint64))
    rng = np.random.default_rng(3)
    for first in rng.integers(0, db.n - 2000, size=3):
        woff, wnb, _ = _oracle_sorted_rows(db, False, int(first), 2000)
        lo, hi = int(off2[first]), int(off2[first + 2000])
        assert np.array_equal(nb2[lo:hi], wnb)


def test_bench_size_10m_properties(gpu_ctx):
    """The bench workload itself (10 M x 150, d=1: the size BASELINE.json's metric names), same
    generator call as bench.py: the same size-independent properties as the 1 M test, plus sampled
    row ranges against the oracle.  The database comes from the product's own reader here (the
    independent Python packer needs minutes at this size; the reader is checked against it on the
    fixtures in test_host_logic.py)."""
    import bench
    from swarm_amd import HostDb
    hdb = HostDb(bench.gen_fasta(10_000_000, 150, 1))
    db = S.Db(headers=[], seqs=hdb.seqs, seq_off=hdb.seq_off, seqlen=hdb.seqlen, abundance=hdb.abundance,
              longest=hdb.longest)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_index_build() is False
    off, nb = gpu_ctx.d1_network(True)
    rows = np.repeat(np.arange(db.n, dtype=np.uint64), np.diff(off).astype(np.int64))
    fwd = (rows << np.uint64(32)) | nb.astype(np.uint64)
    assert (np.diff(fwd) > 0).all()           # rows ascending, unique, in row order
    assert (rows != nb).all()
    rev = np.sort((nb.astype(np.uint64) << np.uint64(32)) | rows)
    assert np.array_equal(fwd, rev)           # symmetric without the abundance rule
    del rev
    off2, nb2 = gpu_ctx.d1_network(False)
    keep = db.abundance[rows.astype(np.int64)] >= db.abundance[nb.astype(np.int64)]
    rows2 = np.repeat(np.arange(db.n, dtype=np.uint64), np.diff(off2).astype(np.int64))
    assert np.array_equal(fwd[keep], (rows2 << np.uint64(32)) | nb2.astype(np.uint64))
    rng = np.random.default_rng(5)
    for first in rng.integers(0, db.n - 1000, size=3):
        woff, wnb, _ = _oracle_sorted_rows(db, False, int(first), 1000)
        lo, hi = int(off2[first]), int(off2[first + 1000])
        assert np.array_equal(nb2[lo:hi], wnb)


def _check_vs_oracle(ctx, db, ncb=False):
    _upload(ctx, db)
    assert ctx.d1_index_build() is False
    off, nb = ctx.d1_network(ncb)
    woff, wnb, _ = _oracle_sorted_rows(db, ncb)
    assert np.array_equal(off, woff)
    assert np.array_equal(nb, wnb)
    return off, nb


def _giant_group_db():
    rng = np.random.default_rng(99)
    head = "".join(rng.choice(list("ACGT"), size=40))
    tail = "".join(rng.choice(list("ACGT"), size=40))
    mids = set()
    while len(mids) < 2600:
        mids.add("".join(rng.choice(list("ACGT"), size=int(rng.integers(5, 8)))))
    seqs = [head + m + tail for m in sorted(mids)]
    # plus a few ordinary clusters so that small and big groups coexist
    for k in range(30):
        base = "".join(rng.choice(list("ACGT"), size=120))
        seqs.append(base)
        for j in range(100):
            p = int(rng.integers(0, 120))
            seqs.append(base[:p] + "ACGT"[(("ACGT".index(base[p])) + 1 + j % 3) % 4] + base[p + 1:])
    seqs = sorted(set(seqs))
    db = S.build_db([(f"s{i}_{1 + (i * 13) % 40}".encode(), s.encode()) for i, s in enumerate(seqs)])
    return db


def test_giant_anchor_groups_fall_back(gpu_ctx):
    """> 2048 amplicons sharing the same first AND last 32 nt: the anchored passes hand those
    seeds (per position range) to the plain kernel; the result must not change."""
    db = _giant_group_db()
    off, nb = _check_vs_oracle(gpu_ctx, db)
    assert len(nb) > 1000


def _length_mix_db():
    rng = np.random.default_rng(7)
    seqs = set()
    for L in (20, 31, 32, 33, 40, 63, 64, 65, 66, 70, 96, 97, 128, 129, 200):
        for k in range(6):
            base = "".join(rng.choice(list("ACGT"), size=L))
            seqs.add(base)
            for j in range(25):
                u = rng.random()
                p = int(rng.integers(0, L))
                if u < 0.4:
                    seqs.add(base[:p] + "ACGT"[int(rng.integers(0, 4))] + base[p + 1:])
                elif u < 0.7:
                    seqs.add(base[:p] + base[p + 1:])
                else:
                    seqs.add(base[:p] + "ACGT"[int(rng.integers(0, 4))] + base[p:])
    seqs = sorted(seqs)
    db = S.build_db([(f"s{i}_{1 + (i * 7) % 9}".encode(), s.encode()) for i, s in enumerate(seqs)])
    return db


def test_short_and_long_sequences_mix(gpu_ctx):
    """lengths around the anchoring thresholds (32, 64/65) and far above"""
    db = _length_mix_db()
    _check_vs_oracle(gpu_ctx, db)
    _check_vs_oracle(gpu_ctx, db, ncb=True)


def test_very_long_sequences_use_plain_kernel(gpu_ctx, tmp_path):
    fa = tmp_path / "long.fa"
    S.gen_fasta(fa, 120, 5000, 31)
    db = S.db_from_fasta(fa)
    _check_vs_oracle(gpu_ctx, db)


def test_unsorted_abundances_still_exact(gpu_ctx, tmp_path):
    """The anchored passes apply the abundance rule through ranks, which presumes the reference's
    db order (abundance descending).  A caller that uploads another order must still get the
    rule applied to the abundances themselves (the library notices and uses the plain kernel)."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 4000, 150, 61)
    db = S.db_from_fasta(fa)
    rng = np.random.default_rng(3)
    db.abundance = np.ascontiguousarray(rng.permutation(db.abundance))
    assert (np.diff(db.abundance.astype(np.int64)) > 0).any()
    off, nb = _check_vs_oracle(gpu_ctx, db)
    assert len(nb) > 500


def test_abundance_ties_follow_the_rule(gpu_ctx, tmp_path):
    """every amplicon has one of three abundances: the `>=` of the rule is decided by ties"""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 6000, 100, 62)
    recs = S.read_fasta(fa)
    db = S.build_db([(h.rsplit(b"_", 1)[0] + b"_%d" % (1 + i % 3), s) for i, (h, s) in enumerate(recs)])
    off, nb = _check_vs_oracle(gpu_ctx, db)
    assert len(nb) > 1000


def test_results_do_not_depend_on_stale_device_memory(tmp_path):
    """Device memory is poisoned (0xFF, then 0xA5) before a fresh context runs index + network +
    fastidious: any read of memory the library did not initialise itself would change the result.
    Own process: torch (used only to poison HBM) must initialise its HIP runtime first."""
    import os
    import subprocess
    import sys
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 30000, 150, 71, 1, 0.3)
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {str(S.ROOT)!r}); sys.path.insert(0, {str(S.ROOT / 'tests')!r})\n"
        "import support as S\n"
        "from swarm_amd import Context, D1Clusters, HostDb\n"
        f"fa = {str(fa)!r}\n"
        "db = S.db_from_fasta(fa)\n"
        "woff, wnb, _ = S.oracle_d1_network(db)\n"
        "wnb = wnb.copy()\n"
        "for i in range(db.n): wnb[int(woff[i]):int(woff[i + 1])].sort()\n"
        "results = []\n"
        "for pattern in (0xFF, 0xA5):\n"
        "    junk = torch.empty(6 * (1 << 30), dtype=torch.uint8, device='cuda'); junk.fill_(pattern)\n"
        "    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()\n"
        "    ctx = Context(0); hdb = HostDb(fa); ctx.upload_hostdb(hdb)\n"
        "    assert ctx.d1_index_build() is False\n"
        "    off, nb = ctx.d1_network()\n"
        "    assert np.array_equal(off, woff) and np.array_equal(nb, wnb), 'network differs'\n"
        "    cl = D1Clusters(hdb, off, nb); flags, stats = cl.light_flags(3)\n"
        "    graft, counters = ctx.d1_fastidious(flags, stats[2], 16)\n"
        "    results.append((graft.copy(), counters[:5].copy())); ctx.close()\n"
        "assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])\n"
        "want_graft, _ = S.oracle_fastidious(db, flags, 16)\n"
        "assert np.array_equal(results[0][0], want_graft), 'graft differs'\n"
        "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ), timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_duplicate_check_by_slices(gpu_ctx, tmp_path):
    """Index build + network of a slice [first, first + count): the slice reports the duplicates it contains (a twin anywhere
    in the database counts), the OR over slices equals the whole-database answer."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 3000, 60, 81)
    recs = S.read_fasta(fa)
    dup_of = recs[1200][1]
    recs.append((b"twin_1", dup_of))                       # identical to an amplicon in the middle of the db
    db = S.build_db(recs)
    _upload(gpu_ctx, db)
    assert gpu_ctx.d1_has_duplicates() is True
    where = [i for i in range(db.n) if db.seq_str(i) == dup_of.decode().upper()]
    assert len(where) == 2
    flags = []
    for first, count in [(0, 1000), (1000, 1000), (2000, db.n - 2000)]:
        got = gpu_ctx.d1_has_duplicates(first, count)            # (index build + network of that slice: whichever meets the twins)
        flags.append(got)
        assert got == any(first <= w < first + count for w in where)
    assert any(flags)
    _upload(gpu_ctx, S.db_from_fasta(fa))
    assert [gpu_ctx.d1_has_duplicates(f, c) for f, c in [(0, 1500), (1500, 1500)]] == [False, False]


def _link_keys(off, nb):
    rows = np.repeat(np.arange(len(off) - 1, dtype=np.uint64), np.diff(off).astype(np.int64))
    return (rows << np.uint64(32)) | nb.astype(np.uint64)


@pytest.mark.parametrize("which", ["generated", "giant_groups", "length_mix"])
@pytest.mark.parametrize("plain", [False, True])
def test_ownership_partials_add_up_to_the_network(tmp_path, which, plain):
    """swa_d1_set_ownership: with world = 2 and 3 every rank's network call returns partial rows;
    over the ranks every link of the complete network appears exactly once — for the anchored
    route (groups owned by key), its fallback seeds (oversized groups, short sequences) and the
    plain route (seeds by id mod world)."""
    import os
    from swarm_amd import Context
    if which == "generated":
        fa = tmp_path / "in.fa"
        S.gen_fasta(fa, 20000, 150, 41)
        db = S.db_from_fasta(fa)
    else:
        db = _giant_group_db() if which == "giant_groups" else _length_mix_db()
    ctx = Context(0)
    try:
        if plain:
            os.environ["SWA_D1_PLAIN"] = "1"
        _upload(ctx, db)
        assert ctx.d1_index_build() is False
        woff, wnb, _ = _oracle_sorted_rows(db)
        whole = _link_keys(woff, wnb)
        off, nb = ctx.d1_network()
        assert np.array_equal(_link_keys(off, nb), whole)
        for world, rebuild in ((2, False), (2, True), (3, True)):
            parts = []
            for rank in range(world):
                ctx.d1_set_ownership(rank, world)
                if rebuild:
                    # the index build of a rank that serves only its groups (no database-wide table
                    # unless a seed needs the plain kernel: short sequences, oversized groups)
                    assert ctx.d1_index_build() is False
                poff, pnb = ctx.d1_network()
                keys = _link_keys(poff, pnb)
                assert (np.diff(keys) > 0).all()              # partial rows ascending and unique, too
                parts.append(keys)
            merged = np.sort(np.concatenate(parts))
            assert np.array_equal(merged, whole), (which, plain, world)
            assert sum(len(p) > 0 for p in parts) == world    # nobody idles
        ctx.d1_set_ownership(0, 1)
        assert ctx.d1_index_build() is False
        off, nb = ctx.d1_network()
        assert np.array_equal(_link_keys(off, nb), whole)
    finally:
        os.environ.pop("SWA_D1_PLAIN", None)
        ctx.close()


def test_owner_ranks_find_duplicates_without_the_table(tmp_path):
    """Index build + network under ownership: identical sequences share a prefix group, so exactly the rank
    that owns it meets them (its prefix pass); a clean database is clean on every rank."""
    from swarm_amd import Context
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 5000, 150, 44)
    recs = S.read_fasta(fa)
    clean = S.build_db(recs)
    twin = (b"twin_1", recs[1234][1])                           # a second copy of one sequence
    dirty = S.build_db(recs + [twin])
    ctx = Context(0)
    try:
        for world in (2, 3):
            found = []
            for rank in range(world):
                ctx.d1_set_ownership(rank, world)
                _upload(ctx, clean)
                assert ctx.d1_has_duplicates() is False
                _upload(ctx, dirty)
                found.append(ctx.d1_has_duplicates())
            assert sum(found) == 1, found
    finally:
        ctx.close()


@pytest.mark.parametrize("n,L,seed,ncb", [(1000, 150, 11, False), (30000, 150, 95, False), (20000, 80, 96, True), (400, 33, 97, False)])
def test_device_clustering_equals_the_host_walk(gpu_ctx, tmp_path, n, L, seed, ncb):
    """swa_d1_cluster_device (agglomeration on the network that stayed in HBM) against the serial walk of
    cluster_d1.cpp over the downloaded CSR: swarm, generation, parent of every amplicon, and every output file."""
    import filecmp
    import os
    from swarm_amd import D1Clusters, HostDb
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, n, L, seed)
    hdb = HostDb(fa)
    gpu_ctx.upload_hostdb(hdb)
    assert gpu_ctx.d1_index_build() is False
    total = gpu_ctx.d1_network_resident(ncb)
    off, nb = gpu_ctx.d1_network_fetch(total)
    woff, wnb = gpu_ctx.d1_network(ncb)
    assert np.array_equal(off, woff) and np.array_equal(nb, wnb)
    gpu_ctx.d1_network_resident(ncb)
    dev = D1Clusters.from_resident(gpu_ctx, hdb)
    os.environ["SWARM_AMD_CLUSTER"] = "serial"
    try:
        host = D1Clusters(hdb, off, nb)
    finally:
        del os.environ["SWARM_AMD_CLUSTER"]
    assert dev.summary() == host.summary()
    assert np.array_equal(dev.swarmid(), host.swarmid())
    assert np.array_equal(dev.generation(), host.generation())
    assert np.array_equal(dev.parent(), host.parent())
    for name, fn in (("o", "write_swarms"), ("s", "write_stats"), ("i", "write_structure"), ("w", "write_seeds")):
        getattr(dev, fn)(tmp_path / f"d{name}")
        getattr(host, fn)(tmp_path / f"h{name}")
        assert filecmp.cmp(tmp_path / f"d{name}", tmp_path / f"h{name}", shallow=False), name


def _same_clustering(dev, host, tmp_path, tag=""):
    import filecmp
    assert dev.summary() == host.summary()
    assert np.array_equal(dev.swarmid(), host.swarmid())
    assert np.array_equal(dev.generation(), host.generation())
    assert np.array_equal(dev.parent(), host.parent())
    for name, fn in (("o", "write_swarms"), ("s", "write_stats"), ("i", "write_structure"), ("w", "write_seeds")):
        getattr(dev, fn)(tmp_path / f"d{name}{tag}")
        getattr(host, fn)(tmp_path / f"h{name}{tag}")
        assert filecmp.cmp(tmp_path / f"d{name}{tag}", tmp_path / f"h{name}{tag}", shallow=False), name


def test_device_clustering_of_deep_chains_and_ties(gpu_ctx, tmp_path):
    """What the frontier walk of cluster_gpu.hip has to get right beyond the generated families: swarms a hundred generations
    deep (more frontiers than one batch of launches holds), chains of EQUAL abundances (links in both directions: the smallest
    id that reaches a vertex may sit in the middle of its chain), a chain that branches, members without links of their own at
    the deepest generation — and the three forms of the result: eager, lazy, prepared (pinned arrays)."""
    import os
    from swarm_amd import D1Clusters, HostDb
    rng = np.random.default_rng(5)
    L = 150
    recs = []
    uid = 0

    def chain(steps, abundance_of, branch_at=None):
        nonlocal uid
        seq = list(rng.choice(list("ACGT"), L))
        positions = rng.permutation(L)
        members = []
        for k in range(steps):
            if k > 0:
                p = positions[k - 1]
                seq[p] = "ACGT"[("ACGT".index(seq[p]) + 1 + int(rng.integers(0, 3))) % 4]
            members.append("".join(seq))
            recs.append((f"c{uid}_{abundance_of(k)}".encode(), "".join(seq)))
            uid += 1
            if branch_at is not None and k == branch_at:
                side = list(seq)
                for j in range(40):                                  # a side branch off the chain, on other positions
                    p = positions[L - 1 - j]
                    side[p] = "ACGT"[("ACGT".index(side[p]) + 1) % 4]
                    recs.append((f"b{uid}_{max(1, abundance_of(k) - 1 - j)}".encode(), "".join(side)))
                    uid += 1

    chain(100, lambda k: 5000 - k)                                   # strictly falling: 100 generations
    chain(70, lambda k: 77)                                          # all equal: links both ways
    chain(90, lambda k: 900 - 2 * k, branch_at=30)
    chain(45, lambda k: 300 if k % 2 == 0 else 299)
    for _ in range(300):                                             # and company: singletons and small families
        chain(int(rng.integers(1, 4)), lambda k: 10 - k)
    fa = tmp_path / "chains.fa"
    with open(fa, "wb") as fh:
        for h, q in recs:
            fh.write(b">" + h + b"\n" + q.encode() + b"\n")
    if S.have_reference():                                           # the command line (prepared / pinned form) against the reference itself
        import filecmp
        import subprocess
        from pathlib import Path
        exe = Path(__file__).resolve().parents[1] / "swarm_amd" / "bin" / "swarm"
        for extra in ([], ["-n"], ["-f"]):
            ref_cmd, our_cmd = ["-d", "1"] + extra, [str(exe), "-d", "1"] + extra
            for k, flag in (("o", "-o"), ("s", "-s"), ("i", "-i"), ("w", "-w")):
                ref_cmd += [flag, str(tmp_path / f"r{k}")]
                our_cmd += [flag, str(tmp_path / f"g{k}")]
            r = S.run_ref_swarm(ref_cmd + ["-l", "/dev/null", str(fa)])
            assert r.returncode == 0, r.stderr
            g = subprocess.run(our_cmd + ["-l", "/dev/null", str(fa)], capture_output=True, text=True)
            assert g.returncode == 0, g.stderr
            for k in "osiw":
                assert filecmp.cmp(tmp_path / f"r{k}", tmp_path / f"g{k}", shallow=False), (extra, k)
    hdb = HostDb(fa)
    gpu_ctx.upload_hostdb(hdb)
    assert gpu_ctx.d1_index_build() is False
    for ncb in (False, True):
        total = gpu_ctx.d1_network_resident(ncb)
        off, nb = gpu_ctx.d1_network_fetch(total)
        os.environ["SWARM_AMD_CLUSTER"] = "serial"
        try:
            host = D1Clusters(hdb, off, nb)
        finally:
            del os.environ["SWARM_AMD_CLUSTER"]
        assert host.summary()["maxgen"] >= 60
        for form in ({}, {"lazy": True}, {"pinned": True}):
            dev = D1Clusters.from_resident(gpu_ctx, hdb, **form)
            _same_clustering(dev, host, tmp_path, f"_{int(ncb)}_{'_'.join(form) or 'eager'}")
            dev.close()
        host.close()


def test_runs_across_the_anchor_boundary(gpu_ctx, tmp_path):
    """The passes divide a seed's microvariants by whether they keep its first 32 nucleotides (swa_aux::pb).  The cases
    that rule exists for: homopolymer runs that cross position 31 / 32 with deletions and insertions inside the run
    (generate_variants lists them at the run's FIRST position), and sequences whose first 32+ nucleotides are one run."""
    rng = np.random.default_rng(17)
    seqs = set()
    for fam in range(300):
        left = "".join(rng.choice(list("ACGT"), int(rng.integers(0, 31))))
        run = str(rng.choice(list("ACGT"))) * int(rng.integers(2, 40))
        if fam % 10 == 0:
            left = ""                                          # the run starts at position 0: first 32 nt one run
            run = run[0] * int(rng.integers(33, 45))
        tail = "".join(rng.choice(list("ACGT"), 120))
        base = (left + run + tail)[:130]
        seqs.add(base)
        for _ in range(12):
            p = int(rng.integers(max(0, len(left) - 2), min(len(base), len(left) + len(run) + 2)))
            k = int(rng.integers(0, 3))
            b = str(rng.choice(list("ACGT")))
            seqs.add(base[:p] + b + base[p + 1:] if k == 0 else (base[:p] + base[p + 1:] if k == 1 else base[:p] + b + base[p:]))
    seqs = sorted(seqs)
    fa = tmp_path / "runs.fa"
    fa.write_text("".join(f">r{i}_{int(rng.integers(1, 4))}\n{s}\n" for i, s in enumerate(seqs)))
    db = S.db_from_fasta(fa)
    off, nb = _check_vs_oracle(gpu_ctx, db)
    assert len(nb) > 3000
    _check_vs_oracle(gpu_ctx, db, ncb=True)


def _conserved_flank_set(path, n, seed, flank=40):
    """Every amplicon starts and ends with the same `flank` nucleotides (primers / conserved regions left on): with the
    default anchors (first / last 32 nt) everybody lands in one prefix and one suffix group."""
    src = path.with_suffix(".src.fa")
    S.gen_fasta(src, n, 150, seed)
    rng = np.random.default_rng(seed)
    head = "".join(rng.choice(list("ACGT"), flank))
    tail = "".join(rng.choice(list("ACGT"), flank))
    seen, out = set(), []
    for h, s in S.read_fasta(src):
        s = s.decode().upper()
        t = head + s[flank:len(s) - flank] + tail
        if t not in seen:
            seen.add(t)
            out.append(b">" + h + b"\n" + t.encode() + b"\n")
    path.write_bytes(b"".join(out))


@pytest.mark.parametrize("n,seed,width", [(30000, 71, 32), (150000, 72, 32), (30000, 71, 0)])
def test_conserved_flanks_move_the_anchor_windows(tmp_path, monkeypatch, n, seed, width):
    """Groups far beyond the LDS limit under 32-nt anchors at the ends: the index build moves the windows inwards (window
    mode: pair kernels, tiled for groups of 65..2048, filtered plain kernel beyond) — same network as the oracle's.
    width 0 = the default choice (round 4): 64-nt windows at the ends reach past the 40 conserved nucleotides."""
    from swarm_amd import Context
    if width:
        monkeypatch.setenv("SWA_D1_ANCHOR_W", str(width))
    fa = tmp_path / "flanks.fa"
    _conserved_flank_set(fa, n, seed)
    db = S.db_from_fasta(fa)
    ctx = Context(0)
    try:
        off, nb = _check_vs_oracle(ctx, db)
        assert len(nb) > n // 2
        if width == 32:
            assert ctx.d1_anchor_windows()[0] >= 32            # the windows did move
        else:
            assert ctx.d1_anchor_windows() == (0, 0) and ctx.d1_anchor_width() == 64
        _check_vs_oracle(ctx, db, ncb=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("n,seed,flank", [(30000, 73, 70), (120000, 74, 70), (30000, 75, 64)])
def test_flanks_wider_than_a_window_are_served_exactly(tmp_path, n, seed, flank):
    """Conserved flanks that swallow a whole 64-nt window (primers left on): every amplicon shares its
    first and last 64 / 70 nt, so both default indexes put everybody in one group — whatever the build chooses instead
    (windows moved inwards, the tiled or the enumerating kernel), the network is the oracle's."""
    from swarm_amd import Context
    fa = tmp_path / "flanks70.fa"
    _conserved_flank_set(fa, n, seed, flank)
    db = S.db_from_fasta(fa)
    ctx = Context(0)
    try:
        off, nb = _check_vs_oracle(ctx, db)
        assert len(nb) > n // 2
        _check_vs_oracle(ctx, db, ncb=True)
    finally:
        ctx.close()


@pytest.mark.parametrize("n,seed", [(40000, 81), (150000, 82)])
def test_v4_like_reads_with_conserved_stretches(tmp_path, n, seed):
    """250-nt reads whose centroids agree in 60 % of their positions (stretches of 8..40 nt, both ends conserved:
    tools/gen_amplicons GEN_CONSERVED): windows at the ends are shared by far more than a family."""
    from swarm_amd import Context
    fa = tmp_path / "v4.fa"
    S.gen_fasta(fa, n, 250, seed, env={"GEN_CONSERVED": "60"})
    db = S.db_from_fasta(fa)
    ctx = Context(0)
    try:
        off, nb = _check_vs_oracle(ctx, db)
        assert len(nb) > n // 2
    finally:
        ctx.close()


def test_tiled_pair_kernel_on_big_groups(tmp_path, monkeypatch):
    """k_d1_pairs_tiled (64 x 64 tiles) in place of the enumerating kernel for groups of 65..2048, default anchors."""
    from swarm_amd import Context
    monkeypatch.setenv("SWA_D1_PAIRS_TILED", "1")
    rng = np.random.default_rng(5)
    cent = "".join(rng.choice(list("ACGT"), 150))
    seqs = {cent}
    while len(seqs) < 1500:                                   # one family of 1500: one prefix and one suffix group of ~1200
        s = cent
        for _ in range(int(rng.integers(1, 3))):
            p = int(rng.integers(0, len(s)))
            k = int(rng.integers(0, 3))
            b = str(rng.choice(list("ACGT")))
            s = s[:p] + b + s[p + 1:] if k == 0 else (s[:p] + s[p + 1:] if k == 1 else s[:p] + b + s[p:])
        seqs.add(s)
    fa = tmp_path / "family.fa"
    fa.write_text("".join(f">m{i}_{int(rng.integers(1, 50))}\n{s}\n" for i, s in enumerate(sorted(seqs))))
    db = S.db_from_fasta(fa)
    ctx = Context(0)
    try:
        off, nb = _check_vs_oracle(ctx, db)
        assert len(nb) > 2000
    finally:
        ctx.close()


def test_every_anchor_its_own_group(gpu_ctx):
    """30 000 unrelated sequences (plus a few neighbours so that there is a network): as many anchor groups as
    amplicons, the fullest the key tables get."""
    rng = np.random.default_rng(808)
    seqs = set()
    while len(seqs) < 30000:
        seqs.add("".join(rng.choice(list("ACGT"), size=int(rng.integers(90, 130)))))
    seqs = sorted(seqs)
    extra = set()
    for s in seqs[:400]:
        p = int(rng.integers(0, len(s)))
        extra.add(s[:p] + s[p + 1:])
    seqs = sorted(set(seqs) | extra)
    db = S.build_db([(f"s{i}_{1 + (i * 7) % 23}".encode(), s.encode()) for i, s in enumerate(seqs)])
    off, nb = _check_vs_oracle(gpu_ctx, db)
    assert len(nb) >= 400


def _family(rng, length, members, max_edits=2, alphabet="ACGT"):
    cent = "".join(rng.choice(list(alphabet), length))
    seqs = {cent}
    while len(seqs) < members:
        s = cent
        for _ in range(int(rng.integers(1, max_edits + 1))):
            p = int(rng.integers(0, len(s)))
            k = int(rng.integers(0, 3))
            b = str(rng.choice(list("ACGT")))
            s = s[:p] + b + s[p + 1:] if k == 0 else (s[:p] + s[p + 1:] if k == 1 else s[:p] + b + s[p:])
        seqs.add(s)
    return seqs


@pytest.mark.parametrize("length,sizes", [(150, [70, 100, 128, 129, 200, 255, 256, 257, 300]),        # one workgroup per group
                                          (240, [66, 130, 250, 270]),                                  # 8 words per member
                                          (150, [2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65] * 3),   # every size class
                                          (90, [4, 8, 16, 32, 64, 90]),
                                          (150, [300, 600, 1000]),            # tiled pair kernel, no hashes built
                                          (150, [300, 700, 1100])])           # a group beyond its range: enumerating kernel for all three
def test_pair_kernel_group_sizes(gpu_ctx, length, sizes):
    """k_d1_group_pairs: groups on both sides of every class boundary (4, 8, 16, 32, 64 | one workgroup up to 256 |
    enumerating kernel beyond), ties in abundance, both widths of the member records."""
    rng = np.random.default_rng(length * 1000 + len(sizes))
    seqs = set()
    for g in sizes:
        seqs |= _family(rng, length, g)
    seqs = sorted(seqs)
    db = S.build_db([(f"s{i}_{1 + (i * 11) % 5}".encode(), s.encode()) for i, s in enumerate(seqs)])
    for ncb in (False, True):
        off, nb = _check_vs_oracle(gpu_ctx, db, ncb)
    assert len(nb) > len(seqs) // 2


def test_pair_kernel_on_runs_and_length_extremes(gpu_ctx):
    """Low-complexity families (homopolymer runs: an indel anywhere in a run is the same sequence; the end-aligned
    comparison sees them from the other side), members of 33..160 nt in the same groups, sequences that are prefixes /
    suffixes of one another."""
    rng = np.random.default_rng(4242)
    seqs = set()
    for length in (70, 100, 159, 160):
        seqs |= _family(rng, length, 40, max_edits=2, alphabet="AC")
    for _ in range(20):                                       # run-heavy centroids
        cent = "".join(str(rng.choice(list("ACGT"))) * int(rng.integers(1, 9)) for _ in range(40))[:150]
        seqs.add(cent)
        for _ in range(25):
            p = int(rng.integers(0, len(cent)))
            k = int(rng.integers(0, 3))
            b = str(rng.choice(list("ACGT")))
            seqs.add(cent[:p] + b + cent[p + 1:] if k == 0 else (cent[:p] + cent[p + 1:] if k == 1 else cent[:p] + b + cent[p:]))
    base = "".join(rng.choice(list("ACGT"), 160))
    for cut in range(1, 60):                                  # chains of prefixes and of suffixes (lengths 101..160)
        seqs.add(base[:160 - cut]); seqs.add(base[cut:])
    seqs = sorted(s for s in seqs if 1 <= len(s) <= 160)
    db = S.build_db([(f"s{i}_{1 + (i * 3) % 4}".encode(), s.encode()) for i, s in enumerate(seqs)])
    for ncb in (False, True):
        _check_vs_oracle(gpu_ctx, db, ncb)


@pytest.mark.parametrize("seed", range(10))
def test_pair_kernels_against_the_oracle_on_random_families(gpu_ctx, seed):
    """Fuzz of the pair kernels: families of random size (1..400) and length (40..256), two- or four-letter alphabets,
    one to three edits from the centroid (so that pairs at distance 0, 1, 2, .. sit in the same anchor groups), many
    abundance ties, both settings of cluster breaking — the network must equal the oracle's."""
    rng = np.random.default_rng(9000 + seed)
    longest = int(rng.choice([90, 150, 160, 200, 256]))
    seqs = set()
    while len(seqs) < 2500:
        length = int(rng.integers(40, longest - 3))
        alphabet = "AC" if rng.random() < 0.3 else "ACGT"
        size = int(rng.choice([1, 2, 3, 5, 9, 20, 40, 70, 130, 260, 400]))
        fam = _family(rng, length, size, max_edits=3, alphabet=alphabet)
        seqs |= {s for s in fam if 1 <= len(s) <= longest}
    seqs = sorted(seqs)
    ab = rng.choice([1, 1, 1, 2, 2, 3, 7, 50], size=len(seqs))
    db = S.build_db([(f"s{i}_{int(ab[i])}".encode(), s.encode()) for i, s in enumerate(seqs)])
    _check_vs_oracle(gpu_ctx, db, ncb=bool(seed & 1))


@pytest.mark.parametrize("which,world", [("generated", 2), ("generated", 5), ("length_mix", 3), ("giant_groups", 2), ("flanks", 3)])
def test_record_routed_index_build_equals_the_network(tmp_path, which, world):
    """swa_d1_route_slice_records + swa_d1_index_build_records (round 6: what the multi-GPU drivers use): the rank that holds
    an amplicon's slice makes its key records, they travel to the owners of their keys (here: through the host, one context
    playing the ranks in turn), every owner starts at the partition.  The records are the ones the owner's own k_keys would
    have made (the id-routed build of rounds 4-5, below, is the same network); over the ranks every link appears exactly
    once."""
    from swarm_amd import Context
    if which == "generated":
        fa = tmp_path / "in.fa"
        S.gen_fasta(fa, 20000, 150, 43)
        db = S.db_from_fasta(fa)
    elif which == "flanks":
        fa = tmp_path / "flank.fa"
        _conserved_flank_set(fa, 20000, 77)
        db = S.db_from_fasta(fa)
    else:
        db = _giant_group_db() if which == "giant_groups" else _length_mix_db()
    woff, wnb, _ = _oracle_sorted_rows(db)
    whole = _link_keys(woff, wnb)
    ctx = Context(0)
    try:
        _upload(ctx, db)
        inbox = _route_records(ctx, db.n, world)
        for index in range(2):                                 # every amplicon long enough went to exactly one owner per index
            ids = np.sort(np.concatenate([np.concatenate(inbox[o][index]) for o in range(world)]) & np.uint64(0xFFFFFFFF))
            assert (np.diff(ids.astype(np.int64)) > 0).all() and len(ids) <= db.n
        parts = []
        for rank in range(world):
            ctx.d1_set_ownership(rank, world)
            assert _build_from_records(ctx, inbox[rank]) is False
            poff, pnb = ctx.d1_network()
            parts.append(_link_keys(poff, pnb))
        merged = np.sort(np.concatenate(parts))
        assert np.array_equal(merged, whole), (which, world)
        if which in ("generated", "flanks"):
            assert sum(len(p) > 0 for p in parts) == world
    finally:
        ctx.close()


def _route_records(ctx, n, world):
    """steps 1 + 2 through the host: every slice routed, the record lists collected per owner: inbox[owner][index]"""
    cap = 3 * n // (2 * world) + 1024
    bounds = [n * r // world for r in range(world + 1)]
    inbox = [[[], []] for _ in range(world)]
    d_rec, d_counts = S.DeviceArray(2 * world * cap, np.uint64), S.DeviceArray(2 * world + 1)
    for r in range(world):
        ctx.d1_route_slice_records(bounds[r], bounds[r + 1] - bounds[r], world, d_rec, cap, d_counts)
        counts = d_counts.to_host()
        assert counts[2 * world] == 0
        rec = d_rec.to_host()
        for owner in range(world):
            for index in range(2):
                k = index * world + owner
                got = rec[k * cap: k * cap + counts[k]]
                assert ((got & np.uint64(0xFFFFFFFF)) >= bounds[r]).all() and ((got & np.uint64(0xFFFFFFFF)) < bounds[r + 1]).all()
                inbox[owner][index].append(got)
    d_rec.free(); d_counts.free()
    return inbox


def _build_from_records(ctx, lists) -> bool:
    recs = [np.concatenate(lists[index]).astype(np.uint64) for index in range(2)]
    bufs = [S.DeviceArray(len(r), np.uint64) for r in recs]
    for b, r in zip(bufs, recs):
        if len(r):
            b.from_host(r)
    try:
        return ctx.d1_index_build_records(bufs[0], bufs[1])
    finally:
        for b in bufs:
            b.free()


def test_record_routed_build_finds_identical_sequences(tmp_path):
    """Two identical sequences in DIFFERENT slices of a record-routed job: the one rank that owns their prefix group meets them
    in its network call (SWA_E_DUPLICATES), nobody else does."""
    from swarm_amd import Context
    from swarm_amd.capi import SWA_E_DUPLICATES, SwaError
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 6000, 150, 9)
    recs = S.read_fasta(fa)
    db = S.build_db(recs + [(b"twin_1", recs[17][1])])
    world = 3
    ctx = Context(0)
    try:
        _upload(ctx, db)
        inbox = _route_records(ctx, db.n, world)
        found = []
        for rank in range(world):
            ctx.d1_set_ownership(rank, world)
            dup = _build_from_records(ctx, inbox[rank])
            try:
                ctx.d1_network()
            except SwaError as e:
                assert e.code == SWA_E_DUPLICATES
                dup = True
            found.append(dup)
        assert sum(found) == 1, found
    finally:
        ctx.close()


@pytest.mark.parametrize("which,world", [("generated", 2), ("length_mix", 3), ("flanks", 3)])
def test_routed_index_build_equals_the_network(tmp_path, which, world):
    """swa_d1_route_slice + swa_d1_index_build_routed: every rank keys only its slice, the ids travel to the owners of
    their keys (here: through the host, one context playing the ranks in turn), every rank builds its indexes from the
    lists it received — and over the ranks every link of the network appears exactly once.  Databases with sequences
    under 65 nt or oversized groups take the database-wide route inside the call; conserved flanks move the windows."""
    from swarm_amd import Context
    if which == "generated":
        fa = tmp_path / "in.fa"
        S.gen_fasta(fa, 20000, 150, 43)
        db = S.db_from_fasta(fa)
    elif which == "flanks":
        fa = tmp_path / "flank.fa"
        _conserved_flank_set(fa, 20000, 77)
        db = S.db_from_fasta(fa)
    else:
        db = _giant_group_db() if which == "giant_groups" else _length_mix_db()
    woff, wnb, _ = _oracle_sorted_rows(db)
    whole = _link_keys(woff, wnb)
    ctx = Context(0)
    try:
        _upload(ctx, db)
        n = db.n
        cap = 3 * n // (2 * world) + 1024
        # step 1 + 2: every slice routed, the lists collected per owner
        bounds = [n * r // world for r in range(world + 1)]
        inbox = [[[], []] for _ in range(world)]
        d_ids, d_counts = S.DeviceArray(2 * world * cap), S.DeviceArray(2 * world + 1)
        for r in range(world):
            ctx.d1_route_slice(bounds[r], bounds[r + 1] - bounds[r], world, d_ids, cap, d_counts)
            counts = d_counts.to_host()
            assert counts[2 * world] == 0
            ids = d_ids.to_host()
            for index in range(2):
                for owner in range(world):
                    k = index * world + owner
                    inbox[owner][index].append(ids[k * cap: k * cap + counts[k]])
        d_ids.free(); d_counts.free()
        # every amplicon long enough went to exactly one owner per index
        for index in range(2):
            everything = np.sort(np.concatenate([np.concatenate(inbox[o][index]) for o in range(world)]))
            assert (np.diff(everything) > 0).all() and len(everything) <= n
        # step 3: each rank from its lists
        parts = []
        for rank in range(world):
            lists = [np.concatenate(inbox[rank][index]).astype(np.uint32) for index in range(2)]
            bufs = [S.DeviceArray(max(1, len(l))) for l in lists]
            for b, l in zip(bufs, lists):
                b.from_host(l)
            ctx.d1_set_ownership(rank, world)
            assert ctx.d1_index_build_routed(bufs[0], len(lists[0]), bufs[1], len(lists[1])) is False
            poff, pnb = ctx.d1_network()
            parts.append(_link_keys(poff, pnb))
            for b in bufs:
                b.free()
        merged = np.sort(np.concatenate(parts))
        assert np.array_equal(merged, whole), (which, world)
        if which in ("generated", "flanks"):
            assert sum(len(p) > 0 for p in parts) == world
    finally:
        ctx.close()
