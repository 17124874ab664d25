#!/usr/bin/env python3
"""tools/gpu_selfcheck.py [rounds] — end-to-end triage for the d=1 path.

Builds the index and the network of the 1 M x 150 set `rounds` times on a fresh context each
time, compares every network with the C oracle's (computed once), and prints the GPU's serial
number.  SWARM_AMD_LIB=<path> checks another build of the library.

Why it exists: an experimental index build (anchor indexes built inside swa_d1_index_build right
behind the sequence hashes, database-wide table only on demand) gave, on SOME boxes of the pool
and on every run there, wrong suffix-anchor keys for a few wavefronts' worth of amplicons
(differently each run; prefix keys always right), and was right on every run on other boxes;
the committed build was right everywhere, including the boxes where the experiment failed.  The
experiment was dropped; this tool is what told the two apart."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import support as S  # noqa: E402
from swarm_amd import Context  # noqa: E402  (SWARM_AMD_LIB is honoured by swarm_amd.capi)



def main() -> None:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    serial = subprocess.run("rocm-smi --showserial 2>/dev/null | grep -i 'serial number:' | head -1", shell=True,
                            capture_output=True, text=True).stdout.strip()
    print("device:", serial or "(rocm-smi gave no serial)")
    fa = "/tmp/selfcheck_1M.fa"
    S.gen_fasta(fa, 1_000_000, 150, 1)
    db = S.db_from_fasta(fa)
    woff, wnb, _ = S.oracle_d1_network(db)
    wnb = wnb.copy()
    for i in range(db.n):
        wnb[int(woff[i]):int(woff[i + 1])].sort()
    bad = 0
    for r in range(rounds):
        ctx = Context(0)
        ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
        assert ctx.d1_index_build() is False
        off, nb = ctx.d1_network(False)
        same = np.array_equal(off, woff) and np.array_equal(nb, wnb)
        print(f"round {r}: {len(nb)} links, {'identical to the oracle' if same else 'DIFFERENT from the oracle (' + str(len(wnb)) + ')'}")
        bad += 0 if same else 1
        # a sharded rank's view: the index rebuilt for a sub-range (insert + lookup kernels)
        first, count = 250_000 * (r % 3), 250_000
        soff, snb = ctx.d1_network(False, first, count)
        lo, hi = int(woff[first]), int(woff[first + count])
        sub_ok = np.array_equal(soff, woff[first:first + count + 1] - woff[first]) and np.array_equal(snb, wnb[lo:hi])
        print(f"         sub-range [{first}, {first + count}): {'identical' if sub_ok else 'DIFFERENT'}")
        bad += 0 if sub_ok else 1
        ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
