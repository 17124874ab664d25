"""The identities the d=1 pair kernels (swarm_amd/csrc/d1_anchor.inc: pair_stage / pair_near) decide links with, pinned
on the CPU against the definition the reference works from (src/variants.cc: b is a microvariant of a <=> one
substitution, deletion or insertion turns a into b):

    one edit apart  <=>  lcp + lcs == n - 1            equal lengths n (identical sequences: never)
                         lcp + lcs >= n                lengths n and n + 1, lcp and lcs capped at n

and, bit for bit, the way the kernels compute lcp and lcs: forward packed words (32 nt per 64-bit word, LSB first) and
the same words END-ALIGNED (shifted up so that the last nucleotide is the top 2-bit group of word W - 1), first
differing bit from the bottom (v_ffbl per dword, OR 32 k, minimum) resp. from the top (v_ffbh)."""
import itertools

import numpy as np
import pytest

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def one_edit_apart(a: str, b: str) -> bool:
    if a == b or abs(len(a) - len(b)) > 1:
        return False
    if len(a) == len(b):
        return sum(x != y for x, y in zip(a, b)) == 1
    s, l = (a, b) if len(a) < len(b) else (b, a)
    return any(l[:p] + l[p + 1:] == s for p in range(len(l)))


def lcp_lcs_rule(a: str, b: str) -> bool:
    n = min(len(a), len(b))
    apart = max(len(a), len(b)) - n
    lcp = next((i for i in range(n) if a[i] != b[i]), n)
    lcs = next((i for i in range(n) if a[-1 - i] != b[-1 - i]), n)
    return (lcp + lcs == n - 1) if apart == 0 else (apart == 1 and lcp + lcs >= n)


@pytest.mark.parametrize("alphabet,maxlen", [("AC", 7), ("ACG", 5)])
def test_rule_equals_definition_exhaustively(alphabet, maxlen):
    words = ["".join(t) for n in range(1, maxlen + 1) for t in itertools.product(alphabet, repeat=n)]
    for a in words:
        for b in words:
            if abs(len(a) - len(b)) <= 1:
                assert lcp_lcs_rule(a, b) == one_edit_apart(a, b), (a, b)


def pack(seq: str, W: int):
    """forward dwords (2 W of them) and end-aligned dwords, as pair_stage makes them"""
    v = 0
    for i, ch in enumerate(seq):
        v |= CODE[ch] << (2 * i)
    shift = 2 * (32 * W - len(seq))
    t = (v << shift) & ((1 << (64 * W)) - 1)
    dw = lambda x: [(x >> (32 * k)) & 0xFFFFFFFF for k in range(2 * W)]
    return dw(v), dw(t)


def ffbl(x):            # v_ffbl_b32: 0xFFFFFFFF for 0
    return 0xFFFFFFFF if x == 0 else (x & -x).bit_length() - 1


def ffbh(x):            # v_ffbh_u32
    return 0xFFFFFFFF if x == 0 else 32 - x.bit_length()


def kernel_near(a: str, b: str, W: int) -> bool:
    wa, ta = pack(a, W)
    wb, tb = pack(b, W)
    f = l = 0xFFFFFFFF
    for k in range(2 * W):
        f = min(f, ffbl(wa[k] ^ wb[k]) | (32 * k))
        l = min(l, ffbh(ta[k] ^ tb[k]) | (32 * (2 * W - 1 - k)))
    n = min(len(a), len(b))
    apart = max(len(a), len(b)) - n
    total = min(f >> 1, n) + min(l >> 1, n)
    return (total + 1 == n) if apart == 0 else (apart == 1 and total >= n)


@pytest.mark.parametrize("W,alphabet", [(5, "ACGT"), (5, "AC"), (8, "ACGT"), (8, "A")])
def test_bit_level_form_equals_definition(W, alphabet):
    """random pairs at 0..3 edits, lengths around every word boundary up to 32 W, runs included"""
    rng = np.random.default_rng(17 * W + len(alphabet))
    lengths = sorted({1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 32 * W - 1, 32 * W})
    for length in lengths:
        if length > 32 * W:
            continue
        for _ in range(120):
            a = "".join(rng.choice(list(alphabet), length))
            b = a
            for _ in range(int(rng.integers(0, 4))):
                p = int(rng.integers(0, max(1, len(b))))
                kind = int(rng.integers(0, 3))
                ch = str(rng.choice(list("ACGT")))
                b = b[:p] + ch + b[p + 1:] if kind == 0 else (b[:p] + b[p + 1:] if kind == 1 else b[:p] + ch + b[p:])
            if not 1 <= len(b) <= 32 * W:
                continue
            assert kernel_near(a, b, W) == one_edit_apart(a, b), (a, b)
            assert kernel_near(b, a, W) == one_edit_apart(a, b), (b, a)


def test_division_between_the_passes_covers_every_link_once():
    """A pair one edit apart (both >= 65 nt) shares its first 32 nt or its last 32 nt (or both): the prefix pass takes
    the pairs with equal first windows, the suffix pass the others — which then share the last window."""
    rng = np.random.default_rng(5)
    for _ in range(3000):
        length = int(rng.integers(65, 100))
        a = "".join(rng.choice(list("AC" if rng.random() < 0.5 else "ACGT"), length))
        p = int(rng.integers(0, len(a)))
        kind = int(rng.integers(0, 3))
        ch = str(rng.choice(list("ACGT")))
        b = a[:p] + ch + a[p + 1:] if kind == 0 else (a[:p] + a[p + 1:] if kind == 1 else a[:p] + ch + a[p:])
        if not one_edit_apart(a, b) or min(len(a), len(b)) < 65:
            continue
        same_prefix, same_suffix = a[:32] == b[:32], a[-32:] == b[-32:]
        assert same_prefix or same_suffix, (a, b)


# ---- round 4: anchor windows of 32 NW nucleotides (NW = 1, 2, 4) --------------------------------------------------------
def kernel_pair_near_full(a: str, b: str, W: int, NW: int, PASS: int):
    """pair_near<PASS, W, NW> with `ends` as rounds 4-5 had it: the window dwords for equality only, both chains over every
    other dword.  Kept as the yardstick of the shortened chains below."""
    wa, ta = pack(a, W)
    wb, tb = pack(b, W)
    f = l = 0xFFFFFFFF
    head = tail = 0
    for k in range(2 * W):
        if PASS == 0 and k < 2 * NW:
            head |= wa[k] ^ wb[k]
        else:
            f = min(f, ffbl(wa[k] ^ wb[k]) | (32 * k))
        if PASS == 1 and k >= 2 * W - 2 * NW:
            tail |= ta[k] ^ tb[k]
        else:
            l = min(l, ffbh(ta[k] ^ tb[k]) | (32 * (2 * W - 1 - k)))
    ok = tail == 0 if PASS == 1 else True
    same = head == 0 if PASS == 0 else f >= 64 * NW
    n = min(len(a), len(b))
    apart = max(len(a), len(b)) - n
    total = min(f >> 1, n) + min(l >> 1, n)
    near = (total + 1 == n) if apart == 0 else (apart == 1 and total >= n)
    kernel_pair_near_full.identical = PASS == 0 and apart == 0 and f == 0xFFFFFFFF and same
    return ok and near and (same if PASS == 0 else not same)


def kernel_pair_near(a: str, b: str, W: int, NW: int, PASS: int):
    """pair_near<PASS, W, NW> with `ends` (d1_anchor.inc), dword for dword, as round 6 has it.  PASS 0: the first 2 NW forward
    dwords for equality only (`same`), the forward chain over the rest, the suffix chain WITHOUT the lowest 2 NW end-aligned
    dwords.  PASS 1: the top 2 NW end-aligned dwords for equality only, the suffix chain over the rest, the forward chain
    over the first 2 NW dwords ONLY (an empty chain = equal first windows = `same`)."""
    wa, ta = pack(a, W)
    wb, tb = pack(b, W)
    f = l = 0xFFFFFFFF
    head = tail = 0
    for k in range(2 * W):
        if PASS == 0:
            if k < 2 * NW:
                head |= wa[k] ^ wb[k]
            else:
                f = min(f, ffbl(wa[k] ^ wb[k]) | (32 * k))
                l = min(l, ffbh(ta[k] ^ tb[k]) | (32 * (2 * W - 1 - k)))
        else:
            if k < 2 * NW:
                f = min(f, ffbl(wa[k] ^ wb[k]) | (32 * k))
            if k >= 2 * W - 2 * NW:
                tail |= ta[k] ^ tb[k]
            else:
                l = min(l, ffbh(ta[k] ^ tb[k]) | (32 * (2 * W - 1 - k)))
    ok = tail == 0 if PASS == 1 else True
    same = head == 0 if PASS == 0 else f == 0xFFFFFFFF
    n = min(len(a), len(b))
    apart = max(len(a), len(b)) - n
    total = min(f >> 1, n) + min(l >> 1, n)
    near = (total + 1 == n) if apart == 0 else (apart == 1 and total >= n)
    kernel_pair_near.identical = PASS == 0 and apart == 0 and f == 0xFFFFFFFF and same       # (round 6: pair_near's fourth result)
    return ok and near and (same if PASS == 0 else not same)


def _edit(rng, a):
    p = int(rng.integers(0, len(a)))
    kind = int(rng.integers(0, 3))
    ch = str(rng.choice(list("ACGT")))
    return a[:p] + ch + a[p + 1:] if kind == 0 else (a[:p] + a[p + 1:] if kind == 1 else a[:p] + ch + a[p:])


@pytest.mark.parametrize("W,NW", [(5, 1), (5, 2), (8, 2), (13, 2), (13, 4)])
def test_wide_windows_divide_every_link_between_the_passes_once(W, NW):
    """Two sequences of >= 2 w + 1 nt one edit apart share their first w or their last w nucleotides (w = 32 NW).  Inside a
    prefix group (first windows equal) PASS 0 reports the pair, inside a suffix group (last windows equal) PASS 1 reports it
    unless the first windows are equal as well: exactly one report per link, none for pairs that are not one edit apart —
    also for pairs whose shorter member has only 2 w nucleotides (a target, never a source), runs, and low-complexity
    alphabets, where deletions and insertions move through repeats."""
    w = 32 * NW
    rng = np.random.default_rng(100 * W + NW)
    checked = 0
    for trial in range(4000):
        length = int(rng.integers(2 * w, min(32 * W, 2 * w + 40) + 1))
        alphabet = ["A", "AC", "ACGT"][trial % 3]
        a = "".join(rng.choice(list(alphabet), length))
        b = a
        for _ in range(int(rng.integers(0, 3))):
            b = _edit(rng, b)
        if not (2 * w <= len(b) <= 32 * W) or abs(len(a) - len(b)) > 1:
            continue
        truth = one_edit_apart(a, b) and max(len(a), len(b)) >= 2 * w + 1
        same_prefix, same_suffix = a[:w] == b[:w], a[-w:] == b[-w:]
        if truth:
            assert same_prefix or same_suffix, (a, b)
        reports = 0
        if same_prefix:                                  # the two meet in a prefix group
            reports += kernel_pair_near(a, b, W, NW, 0)
            assert kernel_pair_near(a, b, W, NW, 0) == kernel_pair_near(b, a, W, NW, 0)
        if same_suffix:                                  # ... and / or in a suffix group
            reports += kernel_pair_near(a, b, W, NW, 1)
            assert kernel_pair_near(a, b, W, NW, 1) == kernel_pair_near(b, a, W, NW, 1)
        assert reports == (1 if one_edit_apart(a, b) else 0), (a, b, same_prefix, same_suffix)
        checked += 1
    assert checked > 1500


@pytest.mark.parametrize("NW", [1, 2, 4])
def test_members_of_a_colliding_suffix_group_are_not_paired(NW):
    """Two window keys with equal 32-bit values share a group: members whose windows differ must not be reported by that
    group (their true groups, if any, report them).  PASS 1 checks the last window for equality, PASS 0 the first."""
    W = 13
    w = 32 * NW
    rng = np.random.default_rng(NW)
    for _ in range(500):
        a = "".join(rng.choice(list("ACGT"), 2 * w + 20))
        p = int(rng.integers(len(a) - w, len(a)))            # an edit inside the last window
        b = a[:p] + "ACGT"[("ACGT".index(a[p]) + 1) % 4] + a[p + 1:]
        assert one_edit_apart(a, b)
        assert not kernel_pair_near(a, b, W, NW, 1)          # would be a pair of the suffix group by position, but the windows differ
        assert kernel_pair_near(a, b, W, NW, 0)              # the prefix group has it


@pytest.mark.parametrize("W,NW", [(5, 1), (5, 2), (8, 2), (13, 4)])
def test_the_prefix_pass_meets_identical_sequences_and_nothing_else(W, NW):
    """Round 6: the reference's duplicate check (src/algod1.cc:1131-1150) as the prefix pass of the pair kernels sees it —
    `identical` of pair_near<0>: equal lengths, the window dwords equal, no differing bit in the rest of the forward words.
    True for a sequence and its copy; false for every pair that differs anywhere: one edit, an edit inside the window, an edit
    in the LAST nucleotide, and a sequence against itself plus trailing A (code 0: the packed words agree, the lengths do not)."""
    w = 32 * NW
    rng = np.random.default_rng(7 * W + NW)
    for trial in range(1500):
        length = int(rng.integers(2 * w + 1, 32 * W))
        alphabet = ["A", "AC", "ACGT"][trial % 3]
        a = "".join(rng.choice(list(alphabet), length))
        assert kernel_pair_near(a, a, W, NW, 0) is False and kernel_pair_near.identical is True
        kernel_pair_near(a, a + "A", W, NW, 0)
        assert kernel_pair_near.identical is False
        b = _edit(rng, a)
        if len(b) > 32 * W or len(b) < 2 * w or b == a:
            continue
        kernel_pair_near(a, b, W, NW, 0)
        assert kernel_pair_near.identical is False, (a, b)
        last = a[:-1] + ("C" if a[-1] != "C" else "G")
        kernel_pair_near(a, last, W, NW, 0)
        assert kernel_pair_near.identical is False
        kernel_pair_near(a, a, W, NW, 1)                       # PASS 1 never reports identity (the prefix pass has)
        assert kernel_pair_near.identical is False


@pytest.mark.parametrize("W,NW", [(5, 1), (5, 2), (8, 2), (15, 4), (21, 4)])
def test_shortened_chains_answer_as_the_full_chains(W, NW):
    """Round 6: the chains of pair_near leave out the dwords that cannot change the answer (the lowest 2 NW end-aligned dwords
    in PASS 0, every forward dword beyond the first window in PASS 1).  Same answer and same `identical` as the full chains for
    ANY two members a group may hold — every length from 2 w up to 32 W (members of a wide group may be much shorter than
    the group's width), 0-3 edits anywhere, edits placed at the seams (around nucleotide w, around n - w, the last
    nucleotide), unrelated sequences (a colliding key), runs and low-complexity alphabets."""
    w = 32 * NW
    rng = np.random.default_rng(1000 * W + NW)
    checked = linked = 0
    for trial in range(6000):
        length = int(rng.integers(2 * w, 32 * W + 1)) if trial % 4 else int(rng.integers(2 * w, min(32 * W, 2 * w + 6) + 1))
        alphabet = ["A", "AC", "ACGT", "ACGT"][trial % 4]
        a = "".join(rng.choice(list(alphabet), length))
        kind = trial % 5
        if kind == 0:                                             # unrelated (only the lengths are close)
            b = "".join(rng.choice(list(alphabet), max(2 * w, length + int(rng.integers(-1, 2)))))
        elif kind == 1:                                           # one edit at a seam
            at = [w - 1, w, w + 1, len(a) - w - 1, len(a) - w, len(a) - w + 1, len(a) - 1, 0][int(rng.integers(0, 8))]
            at = min(max(at, 0), len(a) - 1)
            how = int(rng.integers(0, 3))
            ch = str(rng.choice(list("ACGT")))
            b = a[:at] + ch + a[at + 1:] if how == 0 else (a[:at] + a[at + 1:] if how == 1 else a[:at] + ch + a[at:])
        else:
            b = a
            for _ in range(int(rng.integers(0, 4))):
                b = _edit(rng, b)
        if not (2 * w <= len(b) <= 32 * W) or abs(len(a) - len(b)) > 1:
            continue
        for PASS in (0, 1):
            for x, y in ((a, b), (b, a)):
                full = kernel_pair_near_full(x, y, W, NW, PASS)
                short = kernel_pair_near(x, y, W, NW, PASS)
                assert short == full, (x, y, PASS)
                assert kernel_pair_near.identical == kernel_pair_near_full.identical, (x, y, PASS)
                linked += int(short)
        checked += 1
    assert checked > 3000 and linked > 500


# ---- round 6: the tiled kernel's items and the room the host gives them -----------------------------------------------------------
def _tiled_item_count(members: int, cols: int = 4) -> int:
    """tiled_item_count (d1_anchor.inc): a row tile of 64 members against its column tiles, `cols` (kTiledCols) at a time"""
    tiles = (members + 63) // 64
    return sum((tiles - r + cols - 1) // cols for r in range(tiles))


def test_the_item_regions_hold_the_tiled_kernels_items_whatever_the_groups():
    """list_regions (d1.hip) gives the tiled list per_row_tile * (pop / 64 + pop / 257) items, per_row_tile = what the largest
    group (kStreamGroupCap = 4096 members) makes of one of its row tiles, rounded up.  Whatever mix of groups of 257..4096
    members a population holds, their items fit: items(g) / row_tiles(g) grows with g, row_tiles(g) <= g / 64 + 1, and there
    are at most pop / 257 groups."""
    cap, least = 4096, 257
    per_row_tile = -(-_tiled_item_count(cap) // (cap // 64))
    ratios = [_tiled_item_count(g) / ((g + 63) // 64) for g in range(least, cap + 1)]
    assert max(ratios) <= per_row_tile and ratios[-1] == max(ratios)
    rng = np.random.default_rng(6)
    for _ in range(300):
        sizes = []
        pop = int(rng.integers(least, 200_000))
        left = pop
        while left >= least:
            kind = int(rng.integers(0, 4))
            g = [least, cap, int(rng.integers(least, cap + 1)), 64 * int(rng.integers(5, 65)) + 1][kind]
            g = min(g, left, cap)
            if g < least:
                break
            sizes.append(g)
            left -= g
        items = sum(_tiled_item_count(g) for g in sizes)
        room = 64 + per_row_tile * (pop // 64 + pop // least)
        assert items <= room, (pop, len(sizes), items, room)
    # the 16 bits an item's chunk has: row tile (6 bits) | part << 6
    assert (cap // 64 - 1) | (((cap // 64 + 3) // 4 - 1) << 6) < 1 << 16
