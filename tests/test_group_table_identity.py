"""The two LDS tables of k_group1 (swarm_amd/csrc/d1_stream.inc) as protocols, restated on the CPU and run under random
interleavings of their atomic steps (one LDS instruction = one step; the kernel's lanes run them in any order):

  key table     lines of four entries; a record reads its key's home line, finds the key or the first empty entry, swaps
                its key in (compare-and-swap against 0), on a lost swap reads the line again, on a full line moves one line
                on; the counting add returns its rank.  Whatever the interleaving: every key ends in exactly one entry,
                every entry's count is the number of its key's records, the ranks of a key's records are 0 .. count - 1.
  duplicate     entries tag | record; a record compares its tag with the entries of its home line it has read, swaps
  table         itself into the first empty one and walks on entry by entry when it loses (comparing with what it
                meets).  Whatever the interleaving: of two records with equal fingerprints at least one meets the other.

The reference finds both facts in its hash table of sequences (src/algod1.cc:630-670, src/hashtable.cc); here they are what
the group kernel's stages 1 and 1b must deliver for any schedule of the hardware."""
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


class DupLane:
    """one record on its way into the duplicate table (g1_dup_once, then g1_dup_keys<1> from the start on a failure)"""

    def __init__(self, rec, fp, home, slots):
        self.rec, self.fp, self.home, self.slots = rec, fp, home, slots
        self.state, self.pos, self.e, self.met, self.generic = "read", None, None, set(), False

    def tag(self, entry):
        return entry[0] if entry else None

    def step(self, dtab):
        if self.state == "read":
            q = dtab[self.home: self.home + 4]
            e = next((i for i, x in enumerate(q) if x is None), 4)
            if not self.generic:
                if e == 4:
                    self.generic = True                                            # full line: the generic walk, from the start
                else:
                    self.q, self.e, self.state = q, e, "once_cas"
            else:
                for x in q:
                    if x is not None and x[0] == self.fp:
                        self.met.add(x[1])
                self.pos, self.state = (self.home + e) % self.slots, "walk"
        elif self.state == "once_cas":
            at = self.home + self.e
            if dtab[at] is None:
                dtab[at] = (self.fp, self.rec)
                for x in self.q[:self.e]:                                          # (entries before e were set when read)
                    if x[0] == self.fp:
                        self.met.add(x[1])
                self.state = "done"
            else:
                self.generic, self.state = True, "read"
        elif self.state == "walk":
            old = dtab[self.pos]
            if old is None:
                dtab[self.pos] = (self.fp, self.rec)
                self.state = "done"
            else:
                if old[0] == self.fp:
                    self.met.add(old[1])
                self.pos = (self.pos + 1) % self.slots


@pytest.mark.parametrize("seed", range(12))
def test_duplicate_table_under_any_interleaving(seed):
    rng = np.random.default_rng(100 + seed)
    lines = 6
    slots = 4 * lines
    nrec = int(rng.integers(4, slots - 4))
    nfp = max(2, nrec // 2)
    homes = {f: 4 * int(rng.integers(0, lines)) for f in range(nfp)}              # equal fingerprints share their home line
    fps = [int(rng.integers(0, nfp)) for _ in range(nrec)]
    dtab = [None] * slots
    lanes = [DupLane(r, fps[r], homes[fps[r]], slots) for r in range(nrec)]
    live = list(range(nrec))
    steps = 0
    while live:
        i = live[int(rng.integers(0, len(live)))]
        lanes[i].step(dtab)
        if lanes[i].state == "done":
            live.remove(i)
        steps += 1
        assert steps < 100000
    assert sorted(x[1] for x in dtab if x is not None) == list(range(nrec))      # every record entered once
    for a in range(nrec):
        for b in range(a + 1, nrec):
            if fps[a] == fps[b]:
                assert b in lanes[a].met or a in lanes[b].met
