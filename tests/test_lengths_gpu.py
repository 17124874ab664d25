"""Width classes of the d=1 pair kernels: the record width is chosen per GROUP by its longest member
— 5, 8, 15 or 21 words: up to 160 / 256 / 480 / 672 nt — not per database by its longest sequence; groups with a longer
member go to the plain kernel on a table of those members.  The reference is length-agnostic
(src/variants.cc:184-249 enumerates any sequence); so must the network be, whatever mix of lengths the file holds.
Every set here is compared with the oracle, rows and all."""
import numpy as np
import pytest

import support as S

pytestmark = pytest.mark.gpu


def _family(rng, centre, members, edits=(1, 2)):
    out = {centre}
    tries = 0
    while len(out) < members and tries < 50 * members:
        tries += 1
        s = centre
        for _ in range(int(rng.integers(edits[0], edits[1] + 1))):
            p = int(rng.integers(0, len(s)))
            k = int(rng.integers(0, 3))
            b = str(rng.choice(list("ACGT")))
            s = s[:p] + b + s[p + 1:] if k == 0 else (s[:p] + s[p + 1:] if k == 1 else s[:p] + b + s[p:])
        out.add(s)
    return out


def _write(path, seqs, rng):
    seqs = sorted(set(seqs))
    path.write_text("".join(f">a{i}_{int(rng.integers(1, 40))}\n{s}\n" for i, s in enumerate(seqs)))
    return S.db_from_fasta(path)


def _check(db, ncb=False):
    from swarm_amd import Context
    ctx = Context(0)
    try:
        ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
        assert ctx.d1_index_build() is False
        off, nb = ctx.d1_network(ncb)
        woff, wnb, _ = S.oracle_d1_network(db, ncb)
        wnb = wnb.copy()
        for i in range(db.n):
            wnb[int(woff[i]):int(woff[i + 1])].sort()
        assert np.array_equal(off, woff), f"row lengths differ in {int((np.diff(off) != np.diff(woff)).sum())} rows"
        assert np.array_equal(nb, wnb)
        return len(nb), ctx.d1_anchor_width()
    finally:
        ctx.close()


@pytest.mark.parametrize("ncb", [False, True])
def test_a_few_long_reads_in_a_file_of_short_ones(tmp_path, ncb):
    """10 000 x 150 nt with one per cent of 420-480 nt reads, a handful of 600-nt and of 700-1200-nt ones, every kind in
    families one or two edits apart: four width classes and the too-long route in one database"""
    rng = np.random.default_rng(31)
    src = tmp_path / "short.fa"
    S.gen_fasta(src, 10000, 150, 31)
    seqs = [s.decode().upper() for _, s in S.read_fasta(src)]
    for length, families, size in [(450, 6, 18), (200, 4, 15), (600, 3, 12), (800, 3, 10), (1200, 1, 8)]:
        for _ in range(families):
            centre = "".join(rng.choice(list("ACGT"), int(rng.integers(length - 30, length + 31))))
            seqs += list(_family(rng, centre, size))
    links, width = _check(_write(tmp_path / "mixed.fa", seqs, rng), ncb)
    assert links > 10000 and width == 64


@pytest.mark.parametrize("boundary", [160, 256, 480, 672])
def test_pairs_across_a_class_boundary(tmp_path, boundary):
    """families whose members' lengths straddle the last length of a width class: a deletion / insertion pair has one member
    in each class, and the group takes the wider one"""
    rng = np.random.default_rng(boundary)
    seqs = []
    for k in range(12):
        centre = "".join(rng.choice(list("ACGT"), boundary + (k % 3) - 1))      # boundary - 1, boundary, boundary + 1
        seqs += list(_family(rng, centre, 25, edits=(1, 1)))
    links, _ = _check(_write(tmp_path / "edge.fa", seqs, rng))
    assert links > 300


def test_short_and_long_members_of_one_group(tmp_path):
    """sequences of 150 and of 460 nt that share their first 64 nucleotides land in one prefix group: the group is served at
    the long one's width, the short members end-aligned by whole-word shifts (pair_stage's barrel shifter)"""
    rng = np.random.default_rng(77)
    seqs = []
    for _ in range(10):
        head = "".join(rng.choice(list("ACGT"), 70))
        short = head + "".join(rng.choice(list("ACGT"), 80))
        long_ = head + "".join(rng.choice(list("ACGT"), 390))
        seqs += list(_family(rng, short, 20)) + list(_family(rng, long_, 20))
    links, _ = _check(_write(tmp_path / "shared_head.fa", seqs, rng))
    assert links > 200


@pytest.mark.parametrize("n,length,seed,want_width", [(20000, 460, 61, 128), (6000, 600, 62, 128), (20000, 250, 63, 64)])
def test_long_amplicons_stay_on_the_pair_kernels(tmp_path, n, length, seed, want_width):
    """V3-V4-like (460 nt: 15-word records, 128-byte lines), 600 nt (21 words, 256-byte lines), 250 nt (8 words)"""
    fa = tmp_path / "long.fa"
    S.gen_fasta(fa, n, length, seed)
    links, width = _check(S.db_from_fasta(fa))
    assert links > n // 2 and width == want_width


def test_sequences_beyond_the_widest_kernel(tmp_path):
    """900-nt amplicons: every group has a member too long for the pair kernels — the plain kernel on the member table"""
    fa = tmp_path / "beyond.fa"
    S.gen_fasta(fa, 3000, 900, 64)
    links, _ = _check(S.db_from_fasta(fa))
    assert links > 1500


def test_a_window_shared_by_more_amplicons_than_sixteen_bits_count(tmp_path):
    """70 000 amplicons behind one 64-nt prefix window (a dominant organism's variants in a deep run): the bucket of that
    key holds more records than k_group1's 16-bit ranks count, so all of them are left to the plain kernel — the network
    must still be the oracle's, and two identical sequences among them must still be reported"""
    rng = np.random.default_rng(77)
    head = "".join(rng.choice(list("ACGT"), 64))
    tails = set()
    while len(tails) < 69000:
        tails.add("".join(rng.choice(list("ACGT"), 86)))
    tails = sorted(tails)
    seqs = {head + t for t in tails}
    for t in tails[:1500]:                                       # ... 1500 of them with a neighbour one edit away, in the tail
        p = int(rng.integers(0, 86))
        k = int(rng.integers(0, 3))
        b = str(rng.choice(list("ACGT")))
        seqs.add(head + (t[:p] + b + t[p + 1:] if k == 0 else (t[:p] + t[p + 1:] if k == 1 else t[:p] + b + t[p:])))
    db = _write(tmp_path / "giant.fa", list(seqs), rng)
    assert db.n > 65534
    links, _ = _check(db)
    assert links >= 1300                                         # (one direction per pair: the lighter one does not link up)
    # the same set with one sequence twice: the duplicate is found by the plain kernel's table
    from swarm_amd import Context
    twice = sorted(seqs)
    (tmp_path / "twice.fa").write_text("".join(f">b{i}_{1 + i % 7}\n{s}\n" for i, s in enumerate(twice + [twice[12345]])))
    db2 = S.db_from_fasta(tmp_path / "twice.fa")
    ctx = Context(0)
    try:
        ctx.upload_db(db2.seqs, db2.seq_off, db2.seqlen, db2.abundance, db2.longest)
        assert ctx.d1_has_duplicates() is True
    finally:
        ctx.close()
