"""The LDS key table of k_group1 (swarm_amd/csrc/d1_stream.inc) as a protocol, restated on the CPU and run under random
interleavings of its atomic steps (one LDS instruction = one step; the kernel's lanes run them in any order):

  key table     lines of four entries; a record reads its key's home line, finds the key or the first empty entry, swaps
                its key in (compare-and-swap against 0), on a lost swap reads the line again, on a full line moves one line
                on; the counting add returns its rank.  Whatever the interleaving: every key ends in exactly one entry,
                every entry's count is the number of its key's records, the ranks of a key's records are 0 .. count - 1.

(Rounds 4-5 also restated a second table here — 16-bit tags of sequence fingerprints, the search for identical sequences;
round 6 removed it from the kernel: the prefix pass of the pair kernels meets identical sequences on the sequences
themselves, tests/test_pair_identity.py.)  The reference keeps the same facts in its hash table of sequences
(src/algod1.cc:630-670, src/hashtable.cc); here they are what the group kernel's stage 1 must deliver for any schedule of the
hardware."""
import numpy as np
import pytest

LINES = 8


class Lane:
    """one record on its way into the key table (g1_count_once, then g1_count_keys<1>)"""

    def __init__(self, key, home):
        self.key, self.line, self.state, self.q, self.slot, self.rank = key, home, "read", None, None, None

    def step(self, tkey, tcnt):
        if self.state == "read":
            self.q = list(tkey[4 * self.line: 4 * self.line + 4])                 # ds_read_b128
            if self.key in self.q:
                self.slot, self.state = 4 * self.line + self.q.index(self.key), "add"
            elif 0 in self.q:
                self.slot, self.state = 4 * self.line + self.q.index(0), "cas"
            else:
                self.line = (self.line + 1) % LINES                                # a full line without the key
        elif self.state == "cas":
            old = tkey[self.slot]                                                  # ds_cmpst_rtn_b32
            if old == 0:
                tkey[self.slot] = self.key
            self.state = "add" if old in (0, self.key) else "read"                 # (lost to another key: the line again)
        elif self.state == "add":
            self.rank = tcnt[self.slot]                                            # ds_add_rtn_u32
            tcnt[self.slot] += 1
            self.state = "done"


@pytest.mark.parametrize("seed", range(12))
def test_key_table_under_any_interleaving(seed):
    rng = np.random.default_rng(seed)
    nkeys = int(rng.integers(3, 4 * LINES - 2))
    homes = {k + 1: int(rng.integers(0, LINES)) for k in range(nkeys)}            # (keys are nonzero: key | 1 in the kernel)
    records = [int(rng.integers(1, nkeys + 1)) for _ in range(int(rng.integers(20, 200)))]
    tkey, tcnt = [0] * (4 * LINES), [0] * (4 * LINES)
    lanes = [Lane(k, homes[k]) for k in records]
    live = list(range(len(lanes)))
    steps = 0
    while live:
        i = live[int(rng.integers(0, len(live)))]
        lanes[i].step(tkey, tcnt)
        if lanes[i].state == "done":
            live.remove(i)
        steps += 1
        assert steps < 100000
    for k in set(records):
        assert tkey.count(k) == 1
        slot = tkey.index(k)
        mine = [ln for ln in lanes if ln.key == k]
        assert tcnt[slot] == len(mine)
        assert all(ln.slot == slot for ln in mine)
        assert sorted(ln.rank for ln in mine) == list(range(len(mine)))
    assert sum(tcnt) == len(records)
