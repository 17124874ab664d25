#!/usr/bin/env python3
"""tools/check_index.py [n=200000] — the streaming anchor indexes as they lie in HBM, validated on the host: every
amplicon once in each member list, every work item a set of amplicons that share the window, every window group of
two or more members present exactly once.  Names what is wrong if the network differs from the oracle's."""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S  # noqa: E402
from swarm_amd import Context  # noqa: E402


def windows(db, which, nwin=1):
    """a number per amplicon that is equal exactly for equal anchor windows (first / last 32 nwin nt)"""
    off = db.seq_off[:-1].astype(np.int64)
    padded = np.concatenate([db.seqs, np.zeros(8, np.uint64)])
    ln = db.seqlen.astype(np.int64)
    cols = []
    for q in range(nwin):
        pos = np.full(db.n, 32 * q, dtype=np.int64) if which == 0 else ln - 32 * nwin + 32 * q
        w, sh = pos >> 5, ((pos & 31) << 1).astype(np.uint64)
        lo = padded[off + w] >> sh
        hi = np.where(sh == 0, np.uint64(0), padded[off + w + 1] << ((np.uint64(64) - sh) & np.uint64(63)))
        cols.append(lo | hi)
    if nwin == 1:
        return cols[0]
    _, inverse = np.unique(np.stack(cols, axis=1), axis=0, return_inverse=True)
    return inverse.reshape(-1).astype(np.uint64)


def main() -> None:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    fa = f"/tmp/check_stream_{n}.fa"
    S.gen_fasta(fa, n, 150, 5)
    db = S.db_from_fasta(fa)
    ctx = Context(0)
    ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
    assert not ctx.d1_index_build()
    nwin = ctx.d1_anchor_width() // 32
    print(f"anchor windows: {32 * nwin} nt")
    counters = np.zeros(128, dtype=np.uint64)
    ctx._check(ctx.lib.swa_d1_debug_read(ctx.h, 14, counters.ctypes.data, counters.nbytes))
    counters = counters.view(np.uint32)
    # where the lists lie: [width class][size kind] first item, [.][7] end; then the buffer's length and the quads per line
    layout = np.zeros(4 * 8 + 2, dtype=np.uint64)
    ctx._check(ctx.lib.swa_d1_debug_read(ctx.h, 16, layout.ctypes.data, layout.nbytes))
    region = layout[:32].reshape(4, 8).astype(np.int64)
    cap_items = int(layout[32])
    bad = 0
    for which in range(2):
        members = np.zeros((n + 1) // 2, dtype=np.uint64)
        ctx._check(ctx.lib.swa_d1_debug_read(ctx.h, 10 + which, members.ctypes.data, members.nbytes))
        members = members.view(np.uint32)[:n]
        items = np.zeros((cap_items * 12 + 7) // 8, dtype=np.uint64)
        ctx._check(ctx.lib.swa_d1_debug_read(ctx.h, 12 + which, items.ctypes.data, items.nbytes))
        items = items.view(np.uint32)[:cap_items * 3].reshape(-1, 3)
        cnt = np.bincount(members, minlength=n)
        print(f"index {which}: members once each: {bool((cnt == 1).all())} (missing {int((cnt == 0).sum())}, repeated {int((cnt > 1).sum())})")
        bad += 0 if (cnt == 1).all() else 1
        win = windows(db, which, nwin)
        lists, chunk_lists = [], []
        for cls in range(4):
            base = 64 + (which * 4 + cls) * 8
            lists += [items[region[cls, k]:region[cls, k] + int(counters[base + k])] for k in range(6)]
            chunk_lists.append(items[region[cls, 6]:region[cls, 6] + int(counters[base + 6])])
        chunks = np.concatenate(chunk_lists)
        groups = np.concatenate(lists + [chunks[chunks[:, 2] == 0]])
        print(f"   items per width class and size kind {[len(x) for x in lists]}, row tiles {len(chunks)}")
        # every item: members share the window; sizes fit the class
        covered = np.zeros(n, dtype=np.int64)
        item_of = np.full(n, -1, dtype=np.int64)
        mixed = 0
        for gi, (begin, size, _) in enumerate(groups):
            ids = members[begin:begin + size]
            covered[ids] += 1
            item_of[ids] = gi
            if len(np.unique(win[ids])) != 1:        # two windows with the same 32-bit key share a group: harmless (every
                mixed += 1                            # pair test is exact), and rare: about n^2 / 2^33 group pairs
        print(f"   groups listed {len(groups)}, with mixed windows {mixed}, members covered twice {int((covered > 1).sum())}")
        # truth: window groups of >= 2 members
        order = np.argsort(win, kind="stable")
        sw = win[order]
        starts = np.nonzero(np.concatenate([[True], sw[1:] != sw[:-1]]))[0]
        sizes = np.diff(np.concatenate([starts, [n]]))
        in_group = np.repeat(sizes >= 2, sizes)
        should = np.zeros(n, dtype=bool)
        should[order] = in_group
        miss = np.nonzero(should & (covered == 0))[0]
        extra = np.nonzero(~should & (covered > 0))[0]
        print(f"   true groups >= 2: {int((sizes >= 2).sum())}; members of such groups not covered by any item: {len(miss)}; singletons covered: {len(extra)}")
        bad += (1 if len(miss) or (covered > 1).any() else 0)
        if len(miss):
            i = int(miss[0])
            pos = int(np.nonzero(members == i)[0][0])
            print(f"   e.g. amplicon {i} at member position {pos}; neighbours in the member list {members[max(0, pos - 3):pos + 4].tolist()}, same window: {(win[members[max(0, pos - 3):pos + 4]] == win[i]).tolist()}")
        # every true group contiguous in the member list?
        posof = np.zeros(n, dtype=np.int64)
        posof[members] = np.arange(n)
        p = posof[order]
        gid = np.repeat(np.arange(len(sizes)), sizes)
        lo = np.full(len(sizes), n, dtype=np.int64)
        hi = np.zeros(len(sizes), dtype=np.int64)
        np.minimum.at(lo, gid, p)
        np.maximum.at(hi, gid, p)
        loose = (hi - lo + 1) != sizes
        # (a true group interleaved with another one is fine when both lie in ONE listed group: a merge by equal 32-bit keys)
        first_item = np.full(len(sizes), -2, dtype=np.int64)
        np.maximum.at(first_item, gid, item_of[order])
        min_item = np.full(len(sizes), 1 << 60, dtype=np.int64)
        np.minimum.at(min_item, gid, item_of[order])
        torn = int(((first_item != min_item) & (sizes >= 2)).sum())
        print(f"   true groups not contiguous in the member list: {int(loose.sum())} (merged with another group by equal 32-bit keys); torn over several items: {torn}")
        bad += 1 if torn or mixed > 64 else 0      # (29-bit keys: about n^2 / 2^30 group pairs share one)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
