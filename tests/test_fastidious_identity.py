"""The identities the GPU's pair route for --fastidious rests on (swarm_amd/csrc/d1_fast.inc), checked
on the CPU against the oracle (which is pinned to the reference):

  * generate_variants (src/variants.cc:184-249) lists DISTINCT sequences: exactly the set V1(s) of
    sequences one edit away from s, |V1(s)| = 6 L + 4 + runs(s);
  * graft candidates (src/algod1.cc:244-258, 339-450) = sum over (heavy h, light x) of |V1(h) & V1(x)|,
    graft_cand[x] = the smallest h with a non-empty intersection, non-empty <=> edit distance <= 2;
  * two sequences of >= 112 nt within two edits share their first 32 nt, or their last 32, or the window
    [40, 72) of one equals the window at 39 / 40 / 41 of the other."""
import numpy as np

import support as S


def v1(s: str) -> set:
    out = set()
    for p in range(len(s)):
        for b in "ACGT":
            if b != s[p]:
                out.add(s[:p] + b + s[p + 1:])
        out.add(s[:p] + s[p + 1:])
    for p in range(len(s) + 1):
        for b in "ACGT":
            out.add(s[:p] + b + s[p:])
    out.discard(s)
    return out


def runs(s: str) -> int:
    return 1 + sum(1 for p in range(1, len(s)) if s[p] != s[p - 1])


def test_generate_variants_lists_distinct_sequences():
    rng = np.random.default_rng(5)
    lib = S.oracle()
    tab = S.oracle_zobrist(80)
    for t in range(400):
        L = int(rng.integers(1, 60))
        s = "".join(rng.choice(list("AC" if t % 2 else "ACGT"), L))
        w = S.pack_seq(s.encode())
        h = lib.orc_zobrist_hash(S._p(tab, S.u64p), S._p(w, S.u64p), L)
        got = S.oracle_variants(tab, w, L, h)
        assert len(got) == len(v1(s)) == 6 * L + 4 + runs(s)
        assert len({g[0] for g in got}) == len(got)


def test_candidates_are_intersections_of_microvariant_sets(tmp_path):
    fa = tmp_path / "f.fa"
    S.gen_fasta(fa, 600, 36, 91, 1, 0.4)
    recs = S.read_fasta(fa)
    db = S.build_db(recs)
    seqs = ["".join("ACGT"[int(c)] for c in ((db.words(i)[p >> 5] >> np.uint64((p & 31) * 2)) & np.uint64(3)
                                            for p in range(int(db.seqlen[i])))) for i in range(db.n)]
    rng = np.random.default_rng(1)
    is_light = (db.abundance <= 1).astype(np.uint8)
    is_light[rng.integers(0, db.n, 20)] ^= 1            # any split is a valid input of the seam
    graft, counters = S.oracle_fastidious(db, is_light)
    sets = [v1(s) for s in seqs]
    want = np.full(db.n, 0xFFFFFFFF, dtype=np.uint32)
    cand = 0
    for x in np.flatnonzero(is_light):
        for h in np.flatnonzero(is_light == 0):
            if abs(len(seqs[x]) - len(seqs[h])) > 2:
                continue
            c = len(sets[x] & sets[h])
            if c:
                cand += c
                want[x] = min(want[x], h)
    assert cand == int(counters[2]) and cand > 0
    assert np.array_equal(want, graft)
    assert int(counters[0]) == sum(len(sets[i]) for i in np.flatnonzero(is_light))
    assert int(counters[1]) == sum(len(sets[i]) for i in np.flatnonzero(is_light == 0))


def test_two_edits_leave_a_shared_window():
    rng = np.random.default_rng(9)

    def edit(s):
        p = int(rng.integers(0, len(s) + 1))
        k = int(rng.integers(0, 3))
        if k == 0 and p < len(s):
            return s[:p] + str(rng.choice(list("ACGT"))) + s[p + 1:]
        if k == 1 and p < len(s):
            return s[:p] + s[p + 1:]
        return s[:p] + str(rng.choice(list("ACGT"))) + s[p:]

    seen_classes = set()
    for t in range(20000):
        L = int(rng.integers(114, 200))
        alphabet = "ACGT" if t % 3 else "AC"
        h = "".join(rng.choice(list(alphabet), L))
        x = edit(edit(h))
        if len(x) < 112 or len(h) < 112:
            continue
        if h[:32] == x[:32]:
            seen_classes.add("P")
        elif h[-32:] == x[-32:]:
            seen_classes.add("S")
        else:
            seen_classes.add("M")
            assert h[40:72] in (x[39:71], x[40:72], x[41:73]), (h, x)
    assert seen_classes == {"P", "S", "M"}


def _canonical_slots(s: str):
    """(position, microvariant) for every slot of generate_variants (src/variants.cc:184-249): substitutions at p,
    the deletion of a run filed under the run's first position, insertions of bases that differ from the left neighbour."""
    out = []
    for p in range(len(s)):
        for b in "ACGT":
            if b != s[p]:
                out.append((p, s[:p] + b + s[p + 1:]))
        if p == 0 or s[p] != s[p - 1]:
            out.append((p, s[:p] + s[p + 1:]))
    for p in range(len(s) + 1):
        for b in "ACGT":
            if p == 0 or b != s[p - 1]:
                out.append((p, s[:p] + b + s[p:]))
    return out


def test_common_microvariants_come_from_the_difference_interval():
    """k_fast_count_sites only expands the canonical positions [runstart(h, min(P, E) - 1) - 1, max(P, E) + 1] of h,
    P = first mismatch of h and x, E = last mismatch of the end-aligned comparison: that must find every common
    microvariant — also in periodic sequences, where the two edits can sit anywhere inside a repeat."""
    rng = np.random.default_rng(3)

    def edit(s, alpha):
        p = int(rng.integers(0, len(s) + 1))
        k = int(rng.integers(0, 3))
        b = str(rng.choice(list(alpha)))
        if k == 0 and p < len(s):
            return s[:p] + b + s[p + 1:]
        if k == 1 and p < len(s):
            return s[:p] + s[p + 1:]
        return s[:p] + b + s[p:]

    checked = 0
    for t in range(12000):
        alpha = "ACGT" if t % 3 == 0 else ("AC" if t % 3 == 1 else "AAAAAAC")
        h = "".join(rng.choice(list(alpha), int(rng.integers(3, 40))))
        x = h
        for _ in range(int(rng.integers(1, 3))):
            x = edit(x, alpha)
        if x == h:
            continue
        lh, lx = len(h), len(x)
        dl = lx - lh
        P = 0
        while P < min(lh, lx) and h[P] == x[P]:
            P += 1
        E = -1
        for i in range(lh - 1, -1, -1):
            if i + dl < 0 or h[i] != x[i + dl]:
                E = i
                break
        P, E = min(P, lh - 1), min(max(E, 0), lh - 1)
        q = max(min(P, E) - 1, 0)
        while q > 0 and h[q - 1] == h[q]:
            q -= 1
        lo, hi = max(q - 1, 0), min(max(P, E) + 1, lh)
        vx = v1(x)
        want = len(v1(h) & vx)
        got = sum(1 for p, v in _canonical_slots(h) if lo <= p <= hi and v in vx)
        assert got == want, (h, x, want, got, lo, hi)
        checked += 1
    assert checked > 10000


def test_far_apart_edits_need_only_the_two_neighbourhoods():
    """k_fast_count_sites, second form: with P <= E and the neighbourhoods of the two edits apart, only the canonical
    positions [runstart(P - 1) - 1, P + 1] and [runstart(E - 1) - 1, E + 1] of h are expanded (a common microvariant undoes
    one of the two edits); otherwise the whole interval as above.  Random, low-complexity and periodic sequences."""
    rng = np.random.default_rng(17)

    def edit(s, alpha):
        p = int(rng.integers(0, len(s) + 1))
        k = int(rng.integers(0, 3))
        b = str(rng.choice(list(alpha)))
        if k == 0 and p < len(s):
            return s[:p] + b + s[p + 1:]
        if k == 1 and p < len(s):
            return s[:p] + s[p + 1:]
        return s[:p] + b + s[p:]

    def runstart(h, q):
        q = max(q, 0)
        while q > 0 and h[q - 1] == h[q]:
            q -= 1
        return q

    def make(t):
        L = int(rng.integers(20, 90))
        m = t % 5
        if m == 0:
            return "".join(rng.choice(list("ACGT"), L))
        if m == 1:
            return "".join(rng.choice(list("AC"), L))
        if m == 2:
            return "".join(rng.choice(list("AAAAAAC"), L))
        if m == 3:                                      # a repeat with a few point changes
            unit = "".join(rng.choice(list("ACGT"), int(rng.integers(1, 5))))
            s = list((unit * L)[:L])
            for _ in range(int(rng.integers(0, 3))):
                s[int(rng.integers(0, L))] = str(rng.choice(list("ACGT")))
            return "".join(s)
        a = "".join(rng.choice(list("ACGT"), L // 2))  # a repeat between two random flanks
        unit = "".join(rng.choice(list("AC"), int(rng.integers(1, 4))))
        return a[:L // 4] + (unit * L)[:L // 2] + a[L // 4:]

    checked = split = 0
    for t in range(9000):
        h = make(t)
        x = h
        alpha = "ACGT" if t % 2 else "AC"
        for _ in range(2):
            x = edit(x, alpha)
        if x == h:
            continue
        lh, lx = len(h), len(x)
        dl = lx - lh
        P = 0
        while P < min(lh, lx) and h[P] == x[P]:
            P += 1
        E = -1
        for i in range(lh - 1, -1, -1):
            if i + dl < 0 or h[i] != x[i + dl]:
                E = i
                break
        P, E = min(P, lh - 1), min(max(E, 0), lh - 1)
        first, last = min(P, E), max(P, E)
        lo, hi = max(runstart(h, first - 1) - 1, 0), min(last + 1, lh)
        lo_b, hi_a = max(runstart(h, last - 1) - 1, 0), min(first + 1, lh)
        two = P <= E and lo_b > hi_a
        vx = v1(x)
        want = len(v1(h) & vx)
        if two:
            got = sum(1 for p, v in _canonical_slots(h) if (lo <= p <= hi_a or lo_b <= p <= hi) and v in vx)
            split += 1
        else:
            got = sum(1 for p, v in _canonical_slots(h) if lo <= p <= hi and v in vx)
        assert got == want, (h, x, want, got, (lo, hi_a), (lo_b, hi))
        checked += 1
    assert checked > 8000 and split > 2000


# ---- the two-edit test of k_fast_pairs_lines, bit for bit (swarm_amd/csrc/d1_fast.inc: diagonal_mask, first_difference,
# reg_lce, within_two_edits_reg), restated on Python integers as 64-bit words ----------------------------------------------
_M64 = (1 << 64) - 1


def _pack(s: str, W: int):
    w = [0] * W
    for p, c in enumerate(s):
        w[p >> 5] |= "ACGT".index(c) << ((p & 31) * 2)
    return w


def _diagonal_mask(A, B, k, W):
    s = 2 * abs(k)
    D = []
    for w in range(W):
        b = B[w]
        if k > 0:
            b = ((B[w] >> s) | ((B[w + 1] if w + 1 < W else 0) << (64 - s))) & _M64
        elif k < 0:
            b = ((B[w] << s) | ((B[w - 1] if w > 0 else 0) >> (64 - s))) & _M64
        D.append(A[w] ^ b)
    return D


def _first_difference(D, frm, W):
    pos = 32 * W
    for w in range(W - 1, -1, -1):
        rel = frm - 32 * w
        mask = _M64 if rel <= 0 else (0 if rel >= 32 else (_M64 << (2 * rel)) & _M64)
        t = D[w] & mask
        if t:
            pos = 32 * w + (((t & -t).bit_length() - 1) >> 1)
    return pos


def _reg_lce(D, i, k, m, n, W):
    return min(_first_difference(D, i, W) - i, min(m - i, n - i - k))


def _within_two_edits_reg(a: str, b: str, W: int) -> bool:
    A, B, m, n = _pack(a, W), _pack(b, W), len(a), len(b)
    dk = n - m
    if dk > 2 or dk < -2:
        return False
    D0, Dp, Dm = _diagonal_mask(A, B, 0, W), _diagonal_mask(A, B, 1, W), _diagonal_mask(A, B, -1, W)
    l0 = _reg_lce(D0, 0, 0, m, n, W)
    if dk == 0 and l0 >= m:
        return True
    l1 = [-1] * 5
    l1[2] = l0 + 1 + _reg_lce(D0, l0 + 1, 0, m, n, W) if (l0 + 1 <= m and l0 + 1 <= n) else -1
    l1[3] = l0 + _reg_lce(Dp, l0, 1, m, n, W) if l0 + 1 <= n else -1
    l1[1] = l0 + 1 + _reg_lce(Dm, l0 + 1, -1, m, n, W) if l0 + 1 <= m else -1
    if -1 <= dk <= 1 and l1[dk + 2] >= m:
        return True
    same = {0: l1[2], 1: l1[3], -1: l1[1]}.get(dk, -1)
    below = {1: l1[2], 0: l1[1], 2: l1[3]}.get(dk, -1)
    above = {-1: l1[2], 0: l1[3], -2: l1[1]}.get(dk, -1)
    best = -1
    if same >= 0 and same + 1 <= m and same + 1 + dk <= n:
        best = same + 1
    if below >= 0 and below + dk <= n and below > best:
        best = below
    if above >= 0 and above + 1 <= m and above + 1 > best:
        best = above + 1
    if best < 0:
        return False
    D = {0: D0, 1: Dp, -1: Dm}.get(dk) or _diagonal_mask(A, B, dk, W)
    return best + _reg_lce(D, best, dk, m, n, W) >= m


def _edit_distance_at_most_2(a: str, b: str) -> bool:
    if abs(len(a) - len(b)) > 2:
        return False
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[-1] <= 2


def test_two_edit_test_on_register_words_is_exact():
    """Sequences zero padded to W words ('A' is 00: a tail of A's looks like padding), lengths across a word boundary,
    0..4 random edits, low-complexity alphabets: the difference-mask form of Landau-Vishkin decides "edit distance <= 2"
    exactly as the dynamic programme does."""
    rng = np.random.default_rng(29)

    def edit(s, alpha):
        p = int(rng.integers(0, len(s) + 1))
        k = int(rng.integers(0, 3))
        c = str(rng.choice(list(alpha)))
        if k == 0 and p < len(s):
            return s[:p] + c + s[p + 1:]
        if k == 1 and p < len(s) and len(s) > 1:
            return s[:p] + s[p + 1:]
        return s[:p] + c + s[p:]

    yes = no = 0
    for t in range(6000):
        alpha = ["ACGT", "AC", "A", "AAAC"][t % 4]
        L = int(rng.integers(1, 100))
        a = "".join(rng.choice(list(alpha), L))
        b = a
        for _ in range(int(rng.integers(0, 5))):
            b = edit(b, alpha)
        if not b or len(b) > 150:
            continue
        want = _edit_distance_at_most_2(a, b)
        assert _within_two_edits_reg(a, b, 5) == want, (a, b, want)
        assert _within_two_edits_reg(b, a, 5) == want, (b, a, want)
        yes += want
        no += not want
    assert yes > 1500 and no > 500
