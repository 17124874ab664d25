#!/usr/bin/env python3
"""tools/make_reads.py <unique.fasta> <n_reads> <seed> <out.fasta> — raw-read-like input for -d 0:
n_reads draws (Zipf-weighted, with replacement) from the sequences of unique.fasta, each written
as its own entry `>r<i>_1`."""
import sys

import numpy as np


def main() -> None:
    src, n_reads, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    seqs = [ln.strip() for ln in open(src, "rb") if not ln.startswith(b">")]
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(seqs) + 1) ** 0.9
    pick = rng.choice(len(seqs), size=n_reads, p=w / w.sum())
    with open(out, "wb") as fh:
        chunk = []
        for i, k in enumerate(pick):
            chunk.append(b">r%d_1\n%s\n" % (i, seqs[k]))
            if len(chunk) == 65536:
                fh.write(b"".join(chunk))
                chunk = []
        fh.write(b"".join(chunk))


if __name__ == "__main__":
    main()
