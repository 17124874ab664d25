"""The two ways k_csr_bucket (swarm_amd/csrc/d1_stream.inc) sorts the rows of the CSR, restated on the CPU with the same
index arithmetic and checked against sorted(): the bitonic network over a row padded with the largest value to 4 / 8 /
16 / 32 entries (sort_row_in_registers: comparator (i, i ^ j) ascending where i & k == 0, descending otherwise, for
k = 2, 4, .. N and j = k / 2, .. 1), and the rank sort of a row of up to 64 targets by one wave (wave_rank_sort64: a
target's place is the number of targets that are smaller, or equal and earlier).  The reference's rows are in the order
its variant generator meets the neighbours (src/algod1.cc:558-603); ours are ascending — the tests compare sorted rows,
the clustering does not depend on the order inside a row."""
import numpy as np
import pytest

PAD = 0xFFFFFFFF


def bitonic_network(n: int):
    """the comparators of sort_row_in_registers<N>, in the order the kernel applies them: (low index, high index)"""
    out = []
    k = 2
    while k <= n:
        j = k >> 1
        while j > 0:
            for i in range(n):
                l = i ^ j
                if l > i:
                    out.append((i, l) if (i & k) == 0 else (l, i))
            j >>= 1
        k <<= 1
    return out


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_bitonic_network_sorts_every_padded_row(n):
    rng = np.random.default_rng(n)
    net = bitonic_network(n)
    assert len(net) == {4: 6, 8: 24, 16: 80, 32: 240}[n]
    lengths = list(range(2, n + 1))
    for length in lengths:
        for trial in range(40):
            if trial == 0:
                row = np.arange(length, 0, -1, dtype=np.uint64)                # descending
            elif trial == 1:
                row = np.full(length, 7, dtype=np.uint64)                      # all equal
            elif trial == 2:
                row = np.concatenate([[PAD], rng.integers(0, 1 << 32, length - 1)]).astype(np.uint64)   # a real 0xFFFFFFFF
            else:
                row = rng.integers(0, 1 << 32 if trial % 2 else 50, length).astype(np.uint64)
            v = [int(x) for x in row] + [PAD] * (n - length)
            for lo, hi in net:                                                 # order2(v[lo], v[hi])
                if v[lo] > v[hi]:
                    v[lo], v[hi] = v[hi], v[lo]
            assert v[:length] == sorted(int(x) for x in row)
            assert all(x == PAD for x in v[length:])


def test_zero_one_principle_for_the_smallest_networks():
    """a comparator network sorts everything iff it sorts every 0/1 input: exhaustively for 4, 8 and 16 entries"""
    for n in (4, 8, 16):
        net = bitonic_network(n)
        for bits in range(1 << n):
            v = [(bits >> i) & 1 for i in range(n)]
            for lo, hi in net:
                if v[lo] > v[hi]:
                    v[lo], v[hi] = v[hi], v[lo]
            assert v == sorted(v)


@pytest.mark.parametrize("length", [33, 40, 49, 63, 64])
def test_rank_sort_places_every_target_once(length):
    rng = np.random.default_rng(length)
    for trial in range(30):
        row = rng.integers(0, 1 << 32 if trial % 3 else 20, length).astype(np.uint64)        # (few distinct values: ties by index)
        out = [None] * length
        for x in range(length):                                                             # lane x
            mine = int(row[x])
            rank = sum(1 for y in range(length) if int(row[y]) < mine or (int(row[y]) == mine and y < x))
            assert out[rank] is None
            out[rank] = mine
        assert out == sorted(int(v) for v in row)
