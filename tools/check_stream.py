#!/usr/bin/env python3
"""tools/check_stream.py [n=200000] — triage of the streaming index build / CSR assembly (d1_stream.inc) on the GPU:
the d=1 network of one synthetic set under the streaming index with the streaming and with the counting (table) CSR
assembly, each compared with the C oracle's network (whole database and one sub-range), and the amplicon lines read
back and compared with the database, so that a difference names the stage that made it.  Exit status 0 = all identical."""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S  # noqa: E402
from swarm_amd import Context  # noqa: E402


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    fa = f"/tmp/check_stream_{n}.fa"
    S.gen_fasta(fa, n, 150, 5)
    db = S.db_from_fasta(fa)
    woff, wnb, _ = S.oracle_d1_network(db)
    wnb = wnb.copy()
    for i in range(db.n):
        wnb[int(woff[i]):int(woff[i + 1])].sort()
    bad = 0
    order = sys.argv[2].split(",") if len(sys.argv) > 2 else ["stream"]
    # group sizes of every amplicon's prefix / suffix window (for the diagnosis of missing links)
    off0 = db.seq_off[:-1].astype(np.int64)
    pre = db.seqs[off0]
    ln = db.seqlen.astype(np.int64)
    pos = ln - 32
    w, sh = pos >> 5, ((pos & 31) << 1).astype(np.uint64)
    nxt = np.concatenate([db.seqs, np.zeros(2, np.uint64)])[off0 + w + 1]
    suf = (db.seqs[off0 + w] >> sh) | np.where(sh == 0, np.uint64(0), nxt << ((np.uint64(64) - sh) & np.uint64(63)))
    def gsize(keys):
        u, inv, cnt = np.unique(keys, return_inverse=True, return_counts=True)
        return cnt[inv]
    gpre, gsuf = gsize(pre), gsize(suf)
    for build in order:
        for csr in ("stream", "table"):
            os.environ["SWA_D1_CSR"] = csr
            ctx = Context(0)
            ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
            t0 = time.perf_counter()
            dup = ctx.d1_index_build()
            if build == "stream":
                # the amplicon lines against the database they were made from (64-byte lines: length | rank, then the words)
                raw = np.zeros(db.n * 8, dtype=np.uint64)
                ctx._check(ctx.lib.swa_d1_debug_read(ctx.h, 15, raw.ctypes.data, raw.nbytes))
                L = raw.reshape(db.n, 8)
                nw = (db.seqlen.astype(np.int64) + 31) >> 5
                wrong = np.zeros(db.n, dtype=bool)
                for k in range(7):
                    want = np.where(k < nw, np.concatenate([db.seqs, np.zeros(8, np.uint64)])[off0 + k], np.uint64(0))
                    wrong |= L[:, 1 + k] != want
                meta = L[:, 0]
                rank_want = np.searchsorted(-db.abundance.astype(np.int64), -db.abundance.astype(np.int64), side="left")
                wrong_len = (meta & np.uint64(0xFFFFFFFF)) != db.seqlen.astype(np.uint64)
                wrong_rank = (meta >> np.uint64(32)) != rank_want.astype(np.uint64)
                print(f"    lines: wrong words {int(wrong.sum())}, wrong length {int(wrong_len.sum())}, wrong rank {int(wrong_rank.sum())}" +
                      (f"; first wrong ids {np.nonzero(wrong | wrong_len | wrong_rank)[0][:10].tolist()}" if (wrong | wrong_len | wrong_rank).any() else ""), flush=True)
            off, nb = ctx.d1_network(False)
            dt = time.perf_counter() - t0
            same = (not dup) and np.array_equal(off, woff) and np.array_equal(nb, wnb)
            msg = "identical to the oracle" if same else f"DIFFERENT (dup={dup}, links {len(nb)} vs {len(wnb)}, offsets equal: {np.array_equal(off, woff)})"
            print(f"index {build:6s} csr {csr:6s}: {msg}  [{1e3 * dt:.1f} ms first call]", flush=True)
            if not same and len(off) == len(woff):
                rows = np.nonzero(np.diff(off) != np.diff(woff))[0]
                print(f"    rows with a different length: {len(rows)}; first {rows[:8].tolist()}")
                got = set(zip(np.repeat(np.arange(db.n), np.diff(off).astype(np.int64)).tolist(), nb.tolist()))
                want = set(zip(np.repeat(np.arange(db.n), np.diff(woff).astype(np.int64)).tolist(), wnb.tolist()))
                missing, extra = sorted(want - got), sorted(got - want)
                print(f"    missing links {len(missing)}, extra links {len(extra)}")
                for (a, b) in missing[:12]:
                    shared = "prefix" if pre[a] == pre[b] else "suffix"
                    g = int(gpre[a]) if shared == "prefix" else int(gsuf[a])
                    back = (b, a) in got
                    print(f"      {a} -> {b}: share the {shared} window, group of {g}; lengths {int(ln[a])}, {int(ln[b])}; reverse link present: {back}; abundances {int(db.abundance[a])}, {int(db.abundance[b])}")
                import collections
                hist = collections.Counter()
                for (a, b) in missing:
                    shared = "prefix" if pre[a] == pre[b] else "suffix"
                    hist[(shared, int(gpre[a]) if shared == "prefix" else int(gsuf[a]))] += 1
                print(f"    missing by (pass, group size): {sorted(hist.items())[:40]}")
                if len(rows) == 0 and len(nb) == len(wnb):
                    d = np.nonzero(nb != wnb)[0]
                    print(f"    differing entries: {len(d)}; first at {d[:8].tolist()}")
            bad += 0 if same else 1
            first, count = n // 3, n // 4
            soff, snb = ctx.d1_network(False, first, count)
            lo, hi = int(woff[first]), int(woff[first + count])
            sub_ok = np.array_equal(soff, woff[first:first + count + 1] - woff[first]) and np.array_equal(snb, wnb[lo:hi])
            print(f"    sub-range [{first}, {first + count}): {'identical' if sub_ok else 'DIFFERENT'}", flush=True)
            bad += 0 if sub_ok else 1
            ncb_off, ncb_nb = ctx.d1_network(True)
            voff, vnb, _ = S.oracle_d1_network(db, True) if n <= 200_000 else (None, None, None)
            if voff is not None:
                vnb = vnb.copy()
                for i in range(db.n):
                    vnb[int(voff[i]):int(voff[i + 1])].sort()
                ok = np.array_equal(ncb_off, voff) and np.array_equal(ncb_nb, vnb)
                print(f"    no-cluster-breaking: {'identical' if ok else 'DIFFERENT'}", flush=True)
                bad += 0 if ok else 1
            ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
