#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the UNMODIFIED reference
(oracle/_ref/swarm and oracle/_ref/libswarmref.so, built by `make -C oracle ref` from
/root/reference).  Runs only where /root/reference exists; the fixtures it writes are
plain data (input FASTA + the reference's outputs, and function-level known answers)
and are committed, so the tests run anywhere.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import ctypes as C
import json
import re
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import support as S  # noqa: E402

# name -> (generator args, swarm args, outputs to keep {flag: suffix})
CASES = {
    # d = 1 plumbing case of BASELINE configs[0]: 1k x 150
    "d1_1k": (dict(n=1000, length=150, seed=11), ["-d", "1"], "osiwj"),
    "d1_nobreak": (dict(n=600, length=80, seed=12), ["-d", "1", "-n"], "osij"),
    "d1_mothur": (dict(n=300, length=40, seed=13), ["-d", "1", "-r"], "o"),
    "d1_short": (dict(n=200, length=12, seed=14), ["-d", "1"], "osij"),
    "d1_fastidious": (dict(n=1500, length=100, seed=21, light_frac=0.3), ["-d", "1", "-f"], "osiw"),
    "d1_fastidious_b10_y8": (dict(n=1200, length=60, seed=22, light_frac=0.3), ["-d", "1", "-f", "-b", "10", "-y", "8"], "osi"),
    "d1_uclust": (dict(n=250, length=70, seed=15), ["-d", "1"], "ou"),
    "d2_small": (dict(n=300, length=60, seed=31, max_edits=2), ["-d", "2"], "osiw"),
    "d3_400": (dict(n=400, length=120, seed=32, max_edits=3), ["-d", "3"], "osiwu"),
    "d5_ties": (dict(n=250, length=90, seed=33, max_edits=4), ["-d", "5"], "osi"),
    "d8_16bit": (dict(n=200, length=100, seed=34, max_edits=6), ["-d", "8"], "oi"),
}
FLAG = {"o": "-o", "s": "-s", "i": "-i", "w": "-w", "j": "-j", "u": "-u"}


def run_case(name: str, gen: dict, args: list, keep: str) -> None:
    fa = HERE / f"{name}.fasta"
    S.gen_fasta(fa, gen["n"], gen["length"], gen["seed"], gen.get("max_edits", 1), gen.get("light_frac", 0.0))
    cmd = list(args)
    for k in keep:
        cmd += [FLAG[k], str(HERE / f"{name}.{k}")]
    log = HERE / f"{name}.log"
    cmd += ["-l", str(log), str(fa)]
    r = S.run_ref_swarm(cmd)
    assert r.returncode == 0, (name, r.stderr)
    # keep only the deterministic, machine-independent lines of the log
    lines = [ln for ln in log.read_text().splitlines()
             if re.match(r"^(Database info|Number of swarms|Largest swarm|Max generations|Heavy swarms|Light swarms|"
                         r"Total length|Bloom filter|Generated|Heavy variants|Got|Made|Results before|Resolution|"
                         r"Break swarms|Fastidious)", ln)]
    log.write_text("\n".join(lines) + "\n")
    (HERE / f"{name}.args").write_text(" ".join(args) + "\n")


def usearch_case() -> None:
    """-z / -a handling: headers in ;size=N; form, one header without annotation."""
    src = S.read_fasta(HERE / "d1_short.fasta")
    name = "d1_usearch"
    fa = HERE / f"{name}.fasta"
    with open(fa, "wb") as fh:
        for k, (h, s) in enumerate(src):
            ident, ab = h.rsplit(b"_", 1)
            if k % 3 == 0:
                hdr = b"size=" + ab + b";" + ident
            elif k % 3 == 1:
                hdr = ident + b";size=" + ab + b";"
            else:
                hdr = ident + b";size=" + ab
            if k == 7:
                hdr = ident                      # no annotation: takes -a 2
            fh.write(b">" + hdr + b" some description\n" + s.lower()[:30] + b"\n" + s[30:] + b"\n")
    args = ["-d", "1", "-z", "-a", "2"]
    cmd = list(args)
    for k in "osiwj":
        cmd += [FLAG[k], str(HERE / f"{name}.{k}")]
    cmd += ["-l", "/dev/null", str(fa)]
    r = S.run_ref_swarm(cmd)
    assert r.returncode == 0, r.stderr
    (HERE / f"{name}.args").write_text(" ".join(args) + "\n")


def derep_case() -> None:
    """-d 0: raw-read-like input with many identical sequences (different headers / abundances /
    letter case), abundance-1 entries, and clusters that tie on mass."""
    src = S.read_fasta(HERE / "d1_short.fasta")[:120]
    rng = np.random.default_rng(404)
    entries = []
    for k, (h, s) in enumerate(src):
        copies = 1 + int(rng.geometric(0.45)) if k % 4 else 1
        for c in range(copies):
            ab = int(rng.choice([1, 1, 2, 3, 5, 8, 40]))
            body = s.lower() if (k + c) % 3 == 0 else s
            entries.append((f"r{k}c{c}_{ab}".encode(), body))
    order = rng.permutation(len(entries))
    fa = HERE / "d0_derep.fasta"
    with open(fa, "wb") as fh:
        for j in order:
            h, body = entries[j]
            fh.write(b">" + h + b"\n" + body + b"\n")
    for name, args, keep in (("d0_derep", ["-d", "0"], "osiwu"), ("d0_mothur", ["-d", "0", "-r", "-a", "1"], "o")):
        cmd = list(args)
        for k in keep:
            cmd += [FLAG[k], str(HERE / f"{name}.{k}")]
        log = HERE / f"{name}.log"
        cmd += ["-l", str(log), str(fa)]
        r = S.run_ref_swarm(cmd)
        assert r.returncode == 0, (name, r.stderr)
        lines = [ln for ln in log.read_text().splitlines()
                 if re.match(r"^(Database info|Number of swarms|Largest swarm|Heaviest swarm|Resolution)", ln)]
        log.write_text("\n".join(lines) + "\n")
        (HERE / f"{name}.args").write_text(" ".join(args) + "\n")


TINY = {
    "tiny_one": b">a_1\nA\n",
    "tiny_mix": b">a_2\nA\n>b_1\nC\n>c_1\nAA\n>d_1\nAC\n>e_3\nG\n>f_1\nacgu\n>g_1\nACGTT\n>h_2\nACG\n",
    "tiny_empty": b"",
}


def tiny_cases() -> None:
    """Edge inputs: one amplicon of one nucleotide, lengths 1..5 (deletions down to the empty
    sequence, lower case, U), and an empty file — at d = 0, 1, 2."""
    for name, text in TINY.items():
        fa = HERE / f"{name}.fasta"
        fa.write_bytes(text)
        for d in (0, 1, 2):
            case = f"{name}_d{d}"
            cmd = ["-d", str(d)]
            for k in "osi":
                cmd += [FLAG[k], str(HERE / f"{case}.{k}")]
            cmd += ["-l", "/dev/null", str(fa)]
            r = S.run_ref_swarm(cmd)
            assert r.returncode == 0, (case, r.stderr)
            (HERE / f"{case}.args").write_text(f"-d {d}\n")


def function_vectors() -> None:
    """Known answers of the reference's hot-path functions (through oracle/_ref/libswarmref.so)."""
    code = r'''
import ctypes as C, json, sys
import numpy as np
sys.path.insert(0, %r)
import support as S
ref = C.CDLL(str(S.REF_DIR / "libswarmref.so"))
for f in ("ref_zobrist_table", "ref_zobrist_hash", "ref_zobrist_hash_delete_first", "ref_zobrist_hash_insert_first",
          "ref_hashtable_size", "ref_nw"):
    getattr(ref, f).restype = C.c_uint64
out = {}
ZL = 450
assert ref.ref_zobrist_init(ZL) == 0
tab = np.zeros(4 * ZL, dtype=np.uint64)
ref.ref_zobrist_table(tab.ctypes.data_as(S.u64p), C.c_uint64(4 * ZL))
out["zobrist_first_32"] = [int(x) for x in tab[:32]]
out["zobrist_xor_all_1800"] = int(np.bitwise_xor.reduce(tab))
pat = np.zeros(1024, dtype=np.uint64); ref.ref_bloom_patterns(pat.ctypes.data_as(S.u64p))
out["bloom_patterns_first_8"] = [int(x) for x in pat[:8]]
out["bloom_patterns_xor"] = int(np.bitwise_xor.reduce(pat))
fp = np.zeros(65536, dtype=np.uint64); ref.ref_bloomflex_patterns(6, fp.ctypes.data_as(S.u64p))
out["bloomflex_k6_first_8"] = [int(x) for x in fp[:8]]
out["bloomflex_k6_xor"] = int(np.bitwise_xor.reduce(fp))
out["hashtable_size"] = {str(n): int(ref.ref_hashtable_size(C.c_uint64(n))) for n in
                         [0, 1, 2, 10, 11, 12, 178, 179, 1000, 2867, 45875, 734003, 1000000, 10000000, 11744051, 100000000]}
rng = np.random.default_rng(2024)
seqs = ["A", "AC", "ACGT", "AAAAAAAA", "ACACACACACACACACACACACACACACACACAC",
        "".join(rng.choice(list("ACGT"), size=31)), "".join(rng.choice(list("ACGT"), size=32)),
        "".join(rng.choice(list("ACGT"), size=33)), "".join(rng.choice(list("ACGT"), size=64)),
        "".join(rng.choice(list("ACGT"), size=150)), "".join(rng.choice(list("AC"), size=97)),
        "".join(rng.choice(list("ACGT"), size=400))]
vecs = []
for s in seqs:
    L = len(s)
    w = S.pack_seq(s.encode())
    cp = w.ctypes.data_as(C.c_char_p)
    h = int(ref.ref_zobrist_hash(cp, L))
    N = 7 * L + 5
    oh = np.zeros(N, dtype=np.uint64); op = np.zeros(N, dtype=np.uint32); ot = np.zeros(N, dtype=np.uint8); ob = np.zeros(N, dtype=np.uint8)
    n = ref.ref_generate_variants(cp, L, C.c_uint64(h), oh.ctypes.data_as(S.u64p), op.ctypes.data_as(S.u32p),
                                  ot.ctypes.data_as(S.u8p), ob.ctypes.data_as(S.u8p))
    q = np.zeros(128, dtype=np.uint8)
    ref.ref_findqgrams(cp, C.c_uint64(L), q.ctypes.data_as(S.u8p))
    vecs.append({"seq": s, "hash": h,
                 "hash_delete_first": int(ref.ref_zobrist_hash_delete_first(cp, L)),
                 "hash_insert_first": int(ref.ref_zobrist_hash_insert_first(cp, L)),
                 "variants": [[int(oh[i]), int(op[i]), int(ot[i]), int(ob[i])] for i in range(n)] if L <= 64 else None,
                 "n_variants": int(n),
                 "variant_hash_xor": int(np.bitwise_xor.reduce(oh[:n])),
                 "qgram_hex": q.tobytes().hex()})
out["sequences"] = vecs
# alignment known answers: nw() with the default reduced penalties 18/24/13 and two more scoring systems
pairs = []
def mutate(s, k):
    s = list(s)
    for _ in range(k):
        u = rng.random(); p = int(rng.integers(0, len(s)))
        if u < 0.5: s[p] = "ACGT"[(("ACGT".index(s[p])) + 1 + int(rng.integers(0, 3))) %% 4]
        elif u < 0.75: del s[p]
        else: s.insert(p, "ACGT"[int(rng.integers(0, 4))])
    return "".join(s)
for trial in range(120):
    L = int(rng.integers(5, 140))
    alphabet = "ACGT" if trial %% 3 else "AC"
    a = "".join(rng.choice(list(alphabet), size=L))
    b = mutate(a, int(rng.integers(0, 7))) if trial %% 5 else "".join(rng.choice(list(alphabet), size=int(rng.integers(3, 140))))
    for (mm, go, ge) in ((18, 24, 13), (4, 5, 1), (1, 1, 1)):
        wa = S.pack_seq(a.encode()); wb = S.pack_seq(b.encode())
        aln = C.create_string_buffer(len(a) + len(b) + 8); alen = C.c_uint64(0)
        d = int(ref.ref_nw(wb.ctypes.data_as(C.c_char_p), C.c_uint64(len(b)), wa.ctypes.data_as(C.c_char_p), C.c_uint64(len(a)),
                           C.c_int64(mm), C.c_uint64(go), C.c_uint64(ge), aln, C.byref(alen)))
        pairs.append({"q": a, "d": b, "mismatch": mm, "gapopen": go, "gapextend": ge, "diff": d, "alnlen": int(alen.value)})
out["nw_pairs"] = pairs
json.dump(out, open(%r, "w"))
''' % (str(HERE.parent), str(HERE / "function_vectors.json"))
    # separate process: the reference's table generators are only valid on their first call
    subprocess.run([sys.executable, "-c", code], check=True)


def main() -> None:
    assert S.have_reference(), "needs /root/reference (oracle/_ref) — run in the build container"
    if sys.argv[1:] == ["derep"]:                      # add the d = 0 fixtures only
        derep_case()
        return
    if sys.argv[1:] == ["tiny"]:
        tiny_cases()
        return
    derep_case()
    tiny_cases()
    for name, (gen, args, keep) in CASES.items():
        run_case(name, gen, args, keep)
    usearch_case()
    function_vectors()
    total = sum(p.stat().st_size for p in HERE.iterdir() if p.is_file())
    print(f"golden fixtures written: {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
