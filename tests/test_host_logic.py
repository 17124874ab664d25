"""Host-side product code (no GPU): the C ABI's exported surface, the FASTA database reader and
the d=1 clustering / grafting / writers, checked against the reference's own output files under
tests/golden/.  The neighbour lists fed to the host logic here come from the oracle (the GPU
tests feed it the HIP path's lists)."""
import filecmp
import os
import re

import numpy as np
import pytest

import support as S
from swarm_amd import D1Clusters, HostDb, SwaError, capi

G = S.GOLDEN
INCLUDE = S.ROOT / "include"


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    declared = set()
    for hdr in INCLUDE.glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", hdr.read_text(), flags=re.S)
        declared |= set(re.findall(r"\b(swa_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 30
    missing = sorted(s for s in declared if not hasattr(lib, s))
    assert not missing, missing
    assert set(capi.EXPORTS) <= declared
    assert lib.swa_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    import ctypes as C
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.swa_ctx_create(99, None, C.byref(h))       # device ordinal that cannot exist
    assert rc == capi.SWA_E_DEVICE and not h


@pytest.mark.parametrize("name", ["d1_1k", "d1_short", "d3_400", "d1_fastidious"])
def test_hostdb_matches_independent_packing(name):
    hdb = HostDb(G / f"{name}.fasta")
    db = S.db_from_fasta(G / f"{name}.fasta")
    assert (hdb.n, hdb.longest) == (db.n, db.longest)
    assert np.array_equal(hdb.seq_off, db.seq_off)
    assert np.array_equal(hdb.seqs, db.seqs[:len(hdb.seqs)])
    assert np.array_equal(hdb.seqlen, db.seqlen)
    assert np.array_equal(hdb.abundance, db.abundance)
    assert [hdb.header(i) for i in range(db.n)] == db.headers
    assert hdb.nucleotides == int(db.seqlen.astype(np.uint64).sum())
    info = re.search(r"Database info:\s+(\d+) nt in (\d+) sequences, longest (\d+) nt", (G / f"{name}.log").read_text())
    assert (hdb.nucleotides, hdb.n, hdb.longest) == tuple(int(x) for x in info.groups())


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hostdb_line_shapes_fuzz(tmp_path, seed):
    """Sequences of every length 1..300 wrapped at random widths (the reader packs eight
    nucleotides per turn on clean stretches and byte by byte elsewhere), lower case, U for T,
    CR LF line ends, blank-free: against the independent Python packer."""
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(600):
        L = 1 + (i % 300)
        seq = "".join(rng.choice(list("ACGT"), size=L))
        if rng.random() < 0.3:
            seq = seq.lower()
        if rng.random() < 0.3:
            seq = seq.replace("T", "U").replace("t", "u")
        eol = "\r\n" if rng.random() < 0.2 else "\n"
        lines.append(f">r{i}x{seed}_{1 + int(rng.integers(0, 50))} some description{eol}")
        width = int(rng.choice([1, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 60, 70, 1000]))
        for at in range(0, L, width):
            lines.append(seq[at:at + width] + eol)
    fa = tmp_path / "shapes.fa"
    fa.write_bytes("".join(lines).encode())
    hdb = HostDb(fa)
    db = S.db_from_fasta(fa)
    assert (hdb.n, hdb.longest) == (db.n, db.longest) == (600, 300)
    assert np.array_equal(hdb.seq_off, db.seq_off)
    assert np.array_equal(hdb.seqs, db.seqs[:len(hdb.seqs)])
    assert np.array_equal(hdb.seqlen, db.seqlen)
    assert np.array_equal(hdb.abundance, db.abundance)
    assert [hdb.header(i) for i in range(db.n)] == db.headers


def test_hostdb_db_order_of_a_large_file_sorted_on_several_threads(tmp_path):
    """Above 100 000 entries and 4 MB of text the reader parses on several threads and orders the entries with its
    sample sort (fasta_db.cpp) on 16-byte records whose abundance saturates at 32 bits: abundance descending, then
    header bytes ascending (src/db.cc:388-413).  Many ties on the abundance, headers that agree in their first
    8 bytes (decided by strcmp), and abundances beyond 32 bits (decided through the entries)."""
    rng = np.random.default_rng(11)
    n = 150_000
    abund = np.maximum(1, (1.0 / rng.random(n) ** 1.5).astype(np.int64) % 50)
    abund[rng.integers(0, n, 6)] = [2 ** 32 - 1, 2 ** 32, 2 ** 32 + 5, 2 ** 40, 2 ** 32 + 5, 2 ** 33]
    names = [f"amplicon{int(x):07d}" if i % 3 else f"a{int(x)}" for i, x in enumerate(rng.permutation(n))]
    seqs = rng.integers(0, 4, size=(n, 40), dtype=np.uint8)
    text = "".join(f">{names[i]}_{int(abund[i])}\n{''.join('ACGT'[c] for c in seqs[i])}\n" for i in range(n))
    assert len(text) > 8 << 20
    fa = tmp_path / "large.fa"
    fa.write_text(text)
    hdb = HostDb(fa)
    want = sorted(range(n), key=lambda i: (-int(abund[i]), f"{names[i]}_{int(abund[i])}".encode()))
    assert hdb.n == n
    assert [hdb.header(k) for k in range(0, n, 997)] == [f"{names[i]}_{int(abund[i])}".encode() for i in want[::997]]
    assert np.array_equal(hdb.abundance, np.array([int(abund[i]) for i in want], dtype=np.uint64))
    got_names = [hdb.header(k) for k in range(n)]
    assert got_names == [f"{names[i]}_{int(abund[i])}".encode() for i in want]
    first_word = np.array([sum(int(c) << (2 * j) for j, c in enumerate(seqs[i][:32])) for i in want[:2000]], dtype=np.uint64)
    assert np.array_equal(hdb.seqs[hdb.seq_off[:2000].astype(np.int64)], first_word)


@pytest.mark.parametrize("style,radix", [("mostly_distinct_prefixes", True), ("ties_beyond_16_bytes", True), ("short_names", True), ("wide_abundances", True),
                                         ("shared_prefixes", False), ("saturated_abundances", True)])
def test_hostdb_db_order_by_radix_and_by_comparisons(tmp_path, monkeypatch, capfd, style, radix):
    """The db-order sort of a large file takes the records' integer key — abundance, first 8 identifier bytes — through a
    radix sort (fasta_db.cpp: parallel_radix_sort) and settles runs of equal keys by the next 8 identifier bytes, then by
    strcmp; identifiers that mostly share their first 8 bytes take the comparison sort; records whose abundance saturates the
    key's 32 bits (2^32 - 1 and more) are put in front and ordered among themselves through their entries.  Either way:
    abundance descending, then header bytes ascending (src/db.cc:388-413)."""
    rng = np.random.default_rng(23)
    n = 140_000
    ids = rng.permutation(n)
    abund = np.maximum(1, (1.0 / rng.random(n) ** 1.5).astype(np.int64) % 300)
    if style == "mostly_distinct_prefixes":       # every fifth identifier shares its first 8 bytes with up to nine others
        names = [f"q{i // 5:08d}" if i % 5 == 0 else f"s{int(x)}" for i, x in enumerate(ids)]      # (q0000123 0 .. 9: runs of ten)
    elif style == "ties_beyond_16_bytes":         # ... in runs of seven whose members agree in their first 16 bytes (and some end there)
        names = [(f"r{i // 35:07d}________{i // 5 % 7}" if i // 5 % 7 else f"r{i // 35:07d}________") if i % 5 == 0 else f"s{int(x)}"
                 for i, x in enumerate(ids)]
    elif style == "short_names":                  # shorter than 8 bytes, some a prefix of others ("a1" < "a10" < "a2")
        names = [f"a{int(x)}" for x in ids]
    elif style == "wide_abundances":              # every byte of the 32-bit abundance in use, none saturated
        names = [f"s{int(x)}" for x in ids]
        abund = (2.0 ** (rng.random(n) * 31.9)).astype(np.int64) + 1
        abund[:3] = [2 ** 32 - 2, 2 ** 32 - 2, 2 ** 31]
    elif style == "shared_prefixes":              # the first 8 bytes say nothing
        names = [f"amplicon{int(x):07d}" for x in ids]
    else:                                         # 2^32 - 1 and more: in front, ordered among themselves through their entries
        names = [f"s{int(x)}" for x in ids]
        abund[[77, 5, 139_000, 60_001, 99_999, 3, 4]] = [2 ** 32 - 1, 2 ** 32, 2 ** 40, 2 ** 32 + 5, 2 ** 32 + 5, 2 ** 32 - 1, 2 ** 32 - 2]
    seq = ["".join("ACGT"[c] for c in rng.integers(0, 4, 60)) for _ in range(97)]
    fa = tmp_path / "large.fa"
    fa.write_text("".join(f">{names[i]}_{int(abund[i])}\n{seq[i % 97]}{'ACGT'[i % 4] * (i % 7)}\n" for i in range(n)))
    assert fa.stat().st_size > 8 << 20
    monkeypatch.setenv("SWARM_AMD_DB_TIMING", "1")
    capfd.readouterr()
    hdb = HostDb(fa)
    err = capfd.readouterr().err
    assert ("sorted by radix" in err) == radix, err
    want = sorted(range(n), key=lambda i: (-int(abund[i]), f"{names[i]}_{int(abund[i])}".encode()))
    assert [hdb.header(k) for k in range(n)] == [f"{names[i]}_{int(abund[i])}".encode() for i in want]
    assert np.array_equal(hdb.abundance, np.array([int(abund[i]) for i in want], dtype=np.uint64))
    assert np.array_equal(hdb.seqlen, np.array([60 + i % 7 for i in want], dtype=np.uint32))


def test_repeated_identifiers_in_a_large_file_are_reported_like_the_sequential_scan(tmp_path):
    """The identifier check partitions the identifiers' hashes into buckets and looks at each bucket on its own
    (fasta_db.cpp: earliest_repetition), on several threads: of several repeated identifiers — with different abundance
    annotations, so the headers differ — the one whose SECOND occurrence comes first in the file is reported
    (src/db.cc:680-758 walks the entries in order)."""
    from swarm_amd.capi import SwaError
    rng = np.random.default_rng(5)
    n = 160_000
    names = [f"id{int(x)}" for x in rng.permutation(n)]
    names[120_000] = names[300]          # second occurrence at 120 000
    names[90_000] = names[70_000]        # second occurrence at 90 000: the earliest
    names[150_000] = names[90_000]       # a third copy
    names[140_000] = names[5]
    seq = ["".join("ACGT"[c] for c in rng.integers(0, 4, 45)) for _ in range(64)]
    fa = tmp_path / "dups.fa"
    fa.write_text("".join(f">{names[i]}_{1 + i % 7}\n{seq[i % 64]}{'ACGT'[i % 4] * (i % 5)}\n" for i in range(n)))
    assert fa.stat().st_size > 8 << 20
    with pytest.raises(SwaError) as e:
        HostDb(fa)
    assert f"Duplicated sequence identifier: {names[90_000]}\n" in str(e.value)


def test_unordered_view_and_the_staging_notice_describe_the_same_database(tmp_path):
    """The reader keeps the packed words where its parser threads wrote them (file order, one pool per thread) and tells a
    caller about the pools as soon as the parse is done (swa_hostdb_read_fasta_staged: what the command line's GPU helper
    thread copies while the reader sorts).  swa_hostdb_unordered_view — what swa_db_upload_unordered takes — must point at
    the same pools, and pools + src_off must spell the db-order sequences of swa_hostdb_view word for word."""
    import ctypes as C
    from swarm_amd.capi import DbUnorderedView, DbView
    rng = np.random.default_rng(3)
    n = 150_000
    text = "".join(f">r{i}_{1 + int(a)}\n{''.join('ACGT'[c] for c in rng.integers(0, 4, int(L)))}\n"
                   for i, (a, L) in enumerate(zip(rng.integers(0, 30, n), rng.integers(20, 90, n))))
    fa = tmp_path / "staged.fa"
    fa.write_text(text)
    assert fa.stat().st_size > 8 << 20                       # several parser threads
    lib = capi.load_library()
    told = {}
    NOTICE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32)

    def on_words(user, pools, counts, pieces):
        told["pools"] = [pools[p] for p in range(pieces)]
        told["counts"] = [int(counts[p]) for p in range(pieces)]

    h = C.c_void_p()
    lib.swa_hostdb_read_fasta_staged.argtypes = [C.c_char_p, C.c_int, C.c_int64, C.c_int, NOTICE, C.c_void_p, C.POINTER(C.c_void_p)]
    assert lib.swa_hostdb_read_fasta_staged(str(fa).encode(), 0, 0, 0, NOTICE(on_words), None, C.byref(h)) == 0
    try:
        u = DbUnorderedView()
        lib.swa_hostdb_unordered_view(h, C.byref(u))
        assert u.n == n and u.pieces == len(told["pools"]) >= 2
        pools = C.cast(u.piece_words, C.POINTER(C.c_void_p))
        counts = C.cast(u.piece_word_count, C.POINTER(C.c_uint64))
        assert [pools[p] for p in range(u.pieces)] == told["pools"] and [int(counts[p]) for p in range(u.pieces)] == told["counts"]
        words = np.concatenate([np.frombuffer((C.c_char * (8 * c)).from_address(a), dtype=np.uint64) for a, c in zip(told["pools"], told["counts"]) if c])
        src_off = np.frombuffer((C.c_char * (8 * n)).from_address(u.src_off), dtype=np.uint64).astype(np.int64)
        seqlen = np.frombuffer((C.c_char * (4 * n)).from_address(u.seqlen), dtype=np.uint32)
        v = DbView()
        lib.swa_hostdb_view(h, C.byref(v))
        seq_off = np.frombuffer((C.c_char * (8 * (n + 1))).from_address(v.seq_off), dtype=np.uint64).astype(np.int64)
        seqs = np.frombuffer((C.c_char * (8 * int(seq_off[n]))).from_address(v.seqs), dtype=np.uint64)
        nw = (seqlen.astype(np.int64) + 31) >> 5
        assert np.array_equal(seq_off[1:] - seq_off[:-1], nw)
        for k in range(int(nw.max())):                      # word k of every amplicon that has one
            has = nw > k
            assert np.array_equal(words[src_off[has] + k], seqs[seq_off[:-1][has] + k])
    finally:
        lib.swa_hostdb_free(h)


def test_the_avx2_packer_and_the_byte_loop_pack_the_same_words(tmp_path):
    """fasta_db.cpp packs 32 nucleotides a turn with AVX2 where the CPU has it (pack32_avx2) and falls back to 8 / 1 a turn
    at odd characters and on other CPUs (a line's last nucleotides: pack_tail_avx2, one masked turn): the same file read with SWARM_AMD_NO_AVX2=1 in a process of its own
    (the switch is read once per process) must give the same database, word for word."""
    import hashlib
    import subprocess
    import sys
    rng = np.random.default_rng(17)
    lines = []
    for i in range(3000):
        L = 1 + int(rng.integers(0, 700))
        seq = "".join(rng.choice(list("ACGTUacgtu"), size=L))
        lines.append(f">q{i}_{1 + int(rng.integers(0, 9))}\n")
        width = int(rng.choice([31, 32, 33, 63, 64, 65, 100, 1000]))
        nl = "\r\n" if i % 10 == 3 else "\n"                 # (a carriage return ends the 32-at-a-time turns: byte loop)
        lines.extend(seq[at:at + width] + nl for at in range(0, L, width))
    fa = tmp_path / "packers.fa"
    fa.write_bytes("".join(lines).encode()[:-1])             # (the last line ends with the file: no 32 bytes to load there)

    def digest_here():
        hdb = HostDb(fa)
        return hashlib.md5(hdb.seqs.tobytes() + hdb.seq_off.tobytes() + hdb.seqlen.tobytes()).hexdigest()

    code = ("import sys, hashlib; sys.path.insert(0, sys.argv[1]); from swarm_amd import HostDb; h = HostDb(sys.argv[2]); "
            "print(hashlib.md5(h.seqs.tobytes() + h.seq_off.tobytes() + h.seqlen.tobytes()).hexdigest())")
    import os
    other = subprocess.run([sys.executable, "-c", code, str(S.ROOT), str(fa)], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, SWARM_AMD_NO_AVX2="1"))
    assert other.returncode == 0, other.stderr[-1000:]
    assert other.stdout.strip() == digest_here()
    db = S.db_from_fasta(fa)                                  # ... and the independent Python packer's
    hdb = HostDb(fa)
    assert np.array_equal(hdb.seqs, db.seqs[:len(hdb.seqs)]) and np.array_equal(hdb.seq_off, db.seq_off)


def test_the_streamed_reader_equals_the_mapped_one_whatever_the_buffer(tmp_path, monkeypatch):
    """A regular file is pread() through a few megabytes of buffer per parser thread instead of being mapped (fasta_db.cpp:
    parse_piece_streamed; the 1.6 GB mapping of a 10 M file was page tables to take apart at exit).  The same file through
    buffers of 1 KB and 64 KB — smaller than one record: 30 000-nt sequences wrapped at 70 — and through the mapping
    (SWARM_AMD_INPUT=mmap) must give the same database; a late error must carry the same absolute line number."""
    import hashlib
    rng = np.random.default_rng(23)
    lines = []
    for i in range(40000):
        L = 30000 if i % 9000 == 17 else 1 + int(rng.integers(0, 400))
        seq = "".join(rng.choice(list("ACGT"), size=L))
        lines.append(f">rec{i}_{1 + int(rng.integers(0, 30))}\n")
        width = 70 if L > 1000 else int(rng.choice([60, 80, 1000]))
        lines.extend(seq[at:at + width] + "\n" for at in range(0, L, width))
    text = "".join(lines)
    assert len(text) > 6 << 20                                 # (several parser threads, several buffers each)
    fa = tmp_path / "streamed.fa"
    fa.write_text(text)

    def digest():
        hdb = HostDb(fa)
        d = hashlib.md5(hdb.seqs.tobytes() + hdb.seq_off.tobytes() + hdb.seqlen.tobytes() + hdb.abundance.tobytes()
                        + b"".join(hdb.header(k) for k in range(0, hdb.n, 37))).hexdigest()
        return hdb.n, d

    monkeypatch.setenv("SWARM_AMD_INPUT", "mmap")
    mapped = digest()
    assert mapped[0] == 40000
    monkeypatch.delenv("SWARM_AMD_INPUT")
    assert digest() == mapped
    for kb in ("1", "64"):
        monkeypatch.setenv("SWARM_AMD_READ_CHUNK_KB", kb)
        assert digest() == mapped, kb
    db = S.db_from_fasta(fa)                                   # ... and the independent Python packer's
    hdb = HostDb(fa)
    assert np.array_equal(hdb.seqs, db.seqs[:len(hdb.seqs)]) and np.array_equal(hdb.seq_off, db.seq_off)
    # an illegal character near the end: line number and text as the mapped reader gives them (which the tests above pin
    # against the compiled reference)
    bad = tmp_path / "bad.fa"
    cut = text.rfind("\n>", 0, len(text) - 200000)
    at = text.index("\n", cut + 2) + 5
    bad.write_text(text[:at] + "!" + text[at + 1:])
    errors = []
    for env in ({"SWARM_AMD_INPUT": "mmap"}, {"SWARM_AMD_READ_CHUNK_KB": "1"}, {"SWARM_AMD_READ_CHUNK_KB": "64"}, {}):
        monkeypatch.delenv("SWARM_AMD_INPUT", raising=False)
        monkeypatch.delenv("SWARM_AMD_READ_CHUNK_KB", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with pytest.raises(Exception) as e:
            HostDb(bad)
        errors.append(str(e.value))
    assert "Illegal character '!'" in errors[0] and " on line " in errors[0]
    assert all(x == errors[0] for x in errors), errors


def test_repeated_sequences_in_a_large_file_for_d_above_one(tmp_path):
    """The d > 1 duplicate-sequence check (src/db.cc:763-790) goes through the same partitioned tables as the identifier check:
    a large file with three copies of one sequence and two of another ends in the reference's text (SWA_E_DUPLICATES), the
    same file without them reads fine."""
    from swarm_amd.capi import SwaError
    rng = np.random.default_rng(23)
    n = 140_000
    seqs = ["".join("ACGT"[c] for c in rng.integers(0, 4, int(L))) for L in rng.integers(40, 70, n)]
    clean = tmp_path / "clean.fa"
    clean.write_text("".join(f">u{i}_{1 + i % 5}\n{s}\n" for i, s in enumerate(seqs)))
    assert clean.stat().st_size > 8 << 20
    assert HostDb(clean, check_duplicate_sequences=True).n == n
    seqs[100_000] = seqs[77]
    seqs[130_000] = seqs[77]
    seqs[60_001] = seqs[60_000]
    dups = tmp_path / "dups.fa"
    dups.write_text("".join(f">u{i}_{1 + i % 5}\n{s}\n" for i, s in enumerate(seqs)))
    with pytest.raises(SwaError) as e:
        HostDb(dups, check_duplicate_sequences=True)
    assert e.value.code == capi.SWA_E_DUPLICATES and "identical sequences" in str(e.value)
    assert HostDb(dups, check_duplicate_sequences=False).n == n     # (d = 1 finds them on the GPU instead)


def test_hostdb_usearch_and_append_abundance():
    hdb = HostDb(G / "d1_usearch.fasta", usearch_abundance=True, append_abundance=2)
    db = S.build_db([(h, s) for h, s in S.read_fasta(G / "d1_short.fasta")])
    assert hdb.n == db.n
    # same sequences, same abundances except the one entry that took -a 2
    assert sorted(hdb.seqlen.tolist()) == sorted(db.seqlen.tolist())
    assert int((hdb.abundance == 2).sum()) >= 1


@pytest.mark.parametrize("text,msg", [
    ("ACGT\n", "Illegal header line in fasta file."),
    (">a_1\nACGN\n", "Illegal character 'N' in sequence on line 2."),
    (">a_1\nAC\x01GT\n", "Illegal character (ascii no 1) in sequence on line 2."),
    (">a_1\n>b_1\nACGT\n", "Empty sequence found on line 1."),
    (">a_1\nACGT\n>a_2\nACGA\n", "Duplicated sequence identifier: a"),
    (">a\nACGT\n", "Abundance annotations not found for 1 sequences, starting on line 1."),
    (">a_0\nACGT\n", "Illegal abundance value on line 1"),
    (">_3\nACGT\n", "Empty sequence identifier."),
])
def test_hostdb_error_messages(tmp_path, text, msg):
    fa = tmp_path / "bad.fa"
    fa.write_text(text)
    with pytest.raises(SwaError) as e:
        HostDb(fa)
    assert msg in str(e.value)


def _cluster(name, ncb=False):
    hdb = HostDb(G / f"{name}.fasta", usearch_abundance=name == "d1_usearch", append_abundance=2 if name == "d1_usearch" else 0)
    db = S.Db(headers=[hdb.header(i) for i in range(hdb.n)], seqs=hdb.seqs, seq_off=hdb.seq_off, seqlen=hdb.seqlen,
              abundance=hdb.abundance, longest=hdb.longest)
    off, nb, dup = S.oracle_d1_network(db, ncb)
    assert not dup
    return hdb, db, D1Clusters(hdb, off, nb)


@pytest.mark.parametrize("name", ["d1_1k", "d1_nobreak", "d1_mothur", "d1_short", "d1_usearch"])
def test_d1_outputs_byte_identical(tmp_path, name):
    args = (G / f"{name}.args").read_text().split()
    hdb, db, cl = _cluster(name, "-n" in args)
    usearch = "-z" in args
    append = 2 if "-a" in args else 0
    cl.write_swarms(tmp_path / "o", mothur="-r" in args, usearch=usearch, append_abundance=append)
    assert filecmp.cmp(tmp_path / "o", G / f"{name}.o", shallow=False)
    for suffix, writer in (("s", cl.write_stats), ("i", cl.write_structure), ("w", cl.write_seeds)):
        if (G / f"{name}.{suffix}").exists():
            writer(tmp_path / suffix, usearch=usearch)
            assert filecmp.cmp(tmp_path / suffix, G / f"{name}.{suffix}", shallow=False), suffix
    if (G / f"{name}.j").exists():
        cl.write_network(tmp_path / "j", usearch=usearch, append_abundance=append)
        assert filecmp.cmp(tmp_path / "j", G / f"{name}.j", shallow=False)
    log = (G / f"{name}.log").read_text() if (G / f"{name}.log").exists() else ""
    if log:
        s = cl.summary()
        assert f"Number of swarms:  {s['swarms']}\n" in log
        assert f"Largest swarm:     {s['largest']}\n" in log
        assert f"Max generations:   {s['maxgen']}\n" in log


@pytest.mark.parametrize("name,boundary,bits", [("d1_fastidious", 3, 16), ("d1_fastidious_b10_y8", 10, 8)])
def test_fastidious_outputs_byte_identical(tmp_path, name, boundary, bits):
    hdb, db, cl = _cluster(name)
    log = (G / f"{name}.log").read_text()
    before = cl.summary()
    assert f"Number of swarms:  {before['swarms']}\n" in log.split("Heavy swarms")[0]
    flags, stats = cl.light_flags(boundary)
    assert f"Heavy swarms: {stats[3]}, with {stats[4]} amplicons" in log
    assert f"Light swarms: {stats[0]}, with {stats[1]} amplicons" in log
    assert f"Total length of amplicons in light swarms: {stats[2]}" in log
    graft, counters = S.oracle_fastidious(db, flags, bits)
    grafts = cl.graft(graft)
    assert f"Made {grafts} grafts" in log
    cl.write_swarms(tmp_path / "o")
    cl.write_stats(tmp_path / "s")
    cl.write_structure(tmp_path / "i")
    for suffix in "osi":
        assert filecmp.cmp(tmp_path / suffix, G / f"{name}.{suffix}", shallow=False), suffix
    if (G / f"{name}.w").exists():
        cl.write_seeds(tmp_path / "w")
        assert filecmp.cmp(tmp_path / "w", G / f"{name}.w", shallow=False)
    after = cl.summary()
    tail = log.split("Made")[1]
    assert f"Number of swarms:  {after['swarms']}\n" in tail and f"Largest swarm:     {after['largest']}\n" in tail


def test_d1_uclust_byte_identical(tmp_path):
    from swarm_amd import d1_write_uclust
    hdb, db, cl = _cluster("d1_uclust")
    d1_write_uclust(cl, tmp_path / "u")
    assert filecmp.cmp(tmp_path / "u", G / "d1_uclust.u", shallow=False)
    cl.write_swarms(tmp_path / "o")
    assert filecmp.cmp(tmp_path / "o", G / "d1_uclust.o", shallow=False)


def test_parallel_writers_equal_serial(tmp_path):
    """-o / -r / -i / -w / -u of a big result are formatted by several threads (ranges of swarms)
    and written in order: byte-identical to the single-threaded path (OMP_NUM_THREADS=1 in a
    child process)."""
    import subprocess
    import sys
    fa = tmp_path / "big.fa"
    S.gen_fasta(fa, 260000, 24, 5)
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(S.ROOT)!r})\n"
        "from swarm_amd import D0Clusters, D1Clusters, HostDb, d1_write_uclust\n"
        f"hdb = HostDb({str(fa)!r})\n"
        "rng = np.random.default_rng(1)\n"
        "# a synthetic network: chains i -> i+1 inside blocks of random length (valid for the host\n"
        "# clustering: any CSR is), so that swarms have many different sizes\n"
        "n = hdb.n\n"
        "cut = rng.random(n) < 0.3\n"
        "deg = np.where(cut | (np.arange(n) == n - 1), 0, 1).astype(np.uint64)\n"
        "off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(deg)\n"
        "nb = (np.arange(n, dtype=np.uint32) + 1)[deg == 1]\n"
        "cl = D1Clusters(hdb, off, nb)\n"
        "cl.write_swarms(sys.argv[1]); cl.write_swarms(sys.argv[1] + '.r', mothur=True)\n"
        "cl.write_structure(sys.argv[1] + '.i'); cl.write_seeds(sys.argv[1] + '.w'); d1_write_uclust(cl, sys.argv[1] + '.u')\n"
        "# d = 0 writers on a synthetic (valid) first-identical array: runs of 1..4 equal neighbours\n"
        "first = np.arange(n, dtype=np.uint32); first -= (first % np.uint32(4)) * (rng.random(n) < 0.5)\n"
        "first = first[first]            # idempotent: every non-seed points at a seed\n"
        "d0 = D0Clusters(hdb, first)\n"
        "d0.write_swarms(sys.argv[1] + '.d0o'); d0.write_seeds(sys.argv[1] + '.d0w')\n"
        "# --fastidious: light swarms grafted onto heavy ones print with them and nothing in their own place (the pieces are cut\n"
        "# by what the swarms PRINT: swa_format_in_weighted_pieces) — synthetic candidates: every other light amplicon -> a heavy one\n"
        "flags, stats = cl.light_flags(40)\n"
        "heavy = np.flatnonzero(flags == 0).astype(np.uint32); light = np.flatnonzero(flags == 1)\n"
        "assert len(heavy) > 1000 and len(light) > 1000, (len(heavy), len(light))\n"
        "graft = np.full(n, 0xFFFFFFFF, dtype=np.uint32)\n"
        "pick = light[rng.random(len(light)) < 0.5]\n"
        "graft[pick] = heavy[rng.integers(0, len(heavy), len(pick))]\n"
        "assert len(pick) > 60000, len(pick)      # (from 50 000 pairs on the attach loop runs on all threads: graft_sorted_pairs_in_parallel)\n"
        "assert cl.graft(graft) > 100\n"
        "cl.write_swarms(sys.argv[1] + '.go'); cl.write_structure(sys.argv[1] + '.gi'); cl.write_stats(sys.argv[1] + '.gs')\n")
    outs = []
    for threads in ("1", "4"):
        out = tmp_path / f"o{threads}"
        r = subprocess.run([sys.executable, "-c", code, str(out)], capture_output=True, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS=threads), timeout=600)
        assert r.returncode == 0, r.stderr
        outs.append(out)
    assert outs[0].stat().st_size > 2_000_000
    assert filecmp.cmp(outs[0], outs[1], shallow=False)
    for ext in (".r", ".i", ".w", ".u", ".d0o", ".d0w", ".go", ".gi", ".gs"):
        assert os.path.getsize(str(outs[0]) + ext) > (100_000 if ext == ".gs" else 1_000_000), ext
        assert filecmp.cmp(str(outs[0]) + ext, str(outs[1]) + ext, shallow=False), ext


def _cluster_both_ways(hdb, off, nb, tmp_path):
    out = {}
    for mode in ("serial", "parallel"):
        os.environ["SWARM_AMD_CLUSTER"] = mode
        try:
            cl = D1Clusters(hdb, off, nb)
        finally:
            os.environ.pop("SWARM_AMD_CLUSTER", None)
        cl.write_swarms(tmp_path / f"o_{mode}")
        cl.write_stats(tmp_path / f"s_{mode}")
        cl.write_structure(tmp_path / f"i_{mode}")
        out[mode] = (cl.swarmid().copy(), cl.parent().copy(), cl.generation().copy(), cl.summary())
    a, b = out["serial"], out["parallel"]
    assert np.array_equal(a[0], b[0]), "swarm ids differ"
    assert np.array_equal(a[2], b[2]), "generations differ"
    assert np.array_equal(a[1], b[1]), "parents differ"
    assert a[3] == b[3]
    for k in "osi":
        assert filecmp.cmp(tmp_path / f"{k}_serial", tmp_path / f"{k}_parallel", shallow=False), k


@pytest.mark.parametrize("name", ["d1_1k", "d1_nobreak", "d1_short", "d1_fastidious"])
def test_order_free_clustering_equals_the_walk_on_fixtures(tmp_path, name):
    """The data-parallel formulation of the d=1 clustering (smallest reaching id, distance, smallest
    previous-level pointer) against the serial walk, on the oracle's networks of the fixtures."""
    fa = G / f"{name}.fasta"
    hdb = HostDb(fa)
    db = S.db_from_fasta(fa)
    ncb = "-n" in (G / f"{name}.args").read_text().split()
    off, nb, _ = S.oracle_d1_network(db, ncb)
    _cluster_both_ways(hdb, off, nb, tmp_path)


@pytest.mark.parametrize("seed,n,avg_deg,symmetric", [(1, 3000, 1.5, False), (2, 3000, 3.0, True), (3, 20000, 0.7, False),
                                                      (4, 5000, 6.0, False), (5, 400, 2.0, True)])
def test_order_free_clustering_equals_the_walk_on_random_graphs(tmp_path, seed, n, avg_deg, symmetric):
    """Any CSR is a valid input for the host clustering: random digraphs (with the abundance-rule
    shape: edges point to larger ids) and symmetric ones (the -n shape), long chains included."""
    rng = np.random.default_rng(seed)
    fa = tmp_path / "g.fa"
    S.gen_fasta(fa, n, 30, 100 + seed)
    hdb = HostDb(fa)
    n = hdb.n
    m = int(avg_deg * n)
    src = rng.integers(0, n, size=m)
    dst = np.clip(src + rng.integers(1, 40, size=m), 0, n - 1)          # local edges: long generation chains
    far = rng.random(m) < 0.1
    dst[far] = rng.integers(0, n, size=int(far.sum()))
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    else:
        lo, hi = np.minimum(src, dst), np.maximum(src, dst)
        src, dst = lo, hi
    pairs = np.unique(np.stack([src, dst], axis=1), axis=0)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.add.at(off, pairs[:, 0] + 1, 1)
    off = np.cumsum(off).astype(np.uint64)
    nb = pairs[:, 1].astype(np.uint32)
    _cluster_both_ways(hdb, off, nb, tmp_path)


@pytest.mark.skipif(not S.have_reference(), reason="compiled reference not available")
@pytest.mark.parametrize("text,d", [
    (">a_0\nACGT\n>b_1\nAC\nGN\n", 1),                       # illegal abundance on line 1, illegal character on line 5: the read comes first
    (">_3\nACGT\n>b_1\nACXT\n", 1),                          # empty identifier, then an illegal character
    (">a_1\nACGT\n>a_2\nACGA\n>c_0\nACGG\n", 1),             # repeated identifier (entry 2) before an illegal abundance (entry 3)
    (">a_1\nACGT\n>c_0\nACGG\n>a_2\nACGA\n", 1),             # the other way round
    (">a_1\nACGT\n>b_1\nACGT\n>c_0\nACGG\n", 2),             # d = 2: repeated sequence at entry 2 ends the walk before entry 3
    (">a_1\nACGT\n>c_0\nACGG\n>b_1\nACGT\n", 2),             # illegal abundance at entry 2 comes before the repeated sequence
    (">a_1\nACGT\n>a_1\nACGT\n", 2),                          # same entry: identifier before sequence
    (">a\nACGT\n>b_0\nACGA\n", 1),                            # missing annotation (reported after the walk) vs illegal value (during it)
    (">a\nACGT\n>b_1\nACGA\n>b_2\nACGG\n", 1),                # missing annotation vs repeated identifier
])
def test_hostdb_errors_in_the_reference_order(tmp_path, text, d):
    """Inputs with more than one defect: the message is the one the reference gives (it reads the whole file first,
    then walks the entries once)."""
    fa = tmp_path / "bad.fa"
    fa.write_text(text)
    r = S.run_ref_swarm(["-d", d, "-o", "/dev/null", "-l", "/dev/null", fa])
    assert r.returncode != 0
    want = [ln for ln in r.stderr.splitlines() if ln.startswith("Error:")]
    assert want, r.stderr
    with pytest.raises(SwaError) as e:
        HostDb(fa, check_duplicate_sequences=d > 1)
    assert want[0].strip() in str(e.value), (want[0], str(e.value))


def test_xcd_tile_mapping_visits_every_tile_once():
    """swarm_amd/csrc/d1_stream.inc: xcd_tile(v) — which tile the workgroup of turn v takes, so that runs of 2^run_bits
    consecutive tiles stay on XCD v mod 8.  The partition kernels walk v over [0, tiles rounded up to 8 << run_bits) and
    skip tiles beyond the end: that must reach every tile exactly once, and a run must stay on one XCD.  (The formula
    restated; the kernels themselves are checked end to end by tests/test_stream_gpu.py.)"""
    def xcd_tile(v, run_bits):
        xcd, w = v & 7, v >> 3
        return ((((w >> run_bits) << 3) + xcd) << run_bits) + (w & ((1 << run_bits) - 1))

    for run_bits in range(0, 7):
        span = 8 << run_bits
        for ntiles in (1, 7, 8, 127, 128, 129, 2443, 8194):
            vmax = (ntiles + span - 1) // span * span
            seen = [xcd_tile(v, run_bits) for v in range(vmax)]
            assert sorted(seen) == list(range(vmax))                     # a permutation of the padded range
            assert {t for t in seen if t < ntiles} == set(range(ntiles))
            for v in range(vmax):                                        # the tiles of one run share their XCD
                t = seen[v]
                assert (t >> run_bits) % 8 == v % 8
