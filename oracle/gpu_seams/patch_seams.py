#!/usr/bin/env python3
"""oracle/gpu_seams/patch_seams.py <reference src dir> <out dir> — TEST INFRASTRUCTURE ONLY.

Writes scratch copies of two reference translation units into <out dir> (oracle/_ref/gpu_src, git-ignored) with
the call seams of INTEGRATION.md spliced in BY LINE NUMBER: the replaced line ranges become `#include`s of
oracle/gpu_seams/b1.inc / b2.inc, and one call is added behind search_begin().  No reference text lives in this
repository; a few tokens per anchor line are checked so that a different reference version fails loudly instead of
being patched in the wrong place."""
import sys
from pathlib import Path

src, out = Path(sys.argv[1]), Path(sys.argv[2])
here = Path(__file__).resolve().parent
out.mkdir(parents=True, exist_ok=True)


def lines_of(name):
    return (src / name).read_text().split("\n")


def expect(lines, number, token):
    assert token in lines[number - 1], f"{number}: expected {token!r}, found {lines[number - 1]!r} (another reference version?)"


# ---- algod1.cc: B1 = lines 1129-1171 (hash_insert loop ... ThreadRunner(network_thread) ... progress_done),
#                 B2 = lines 1404-1467 (bloomflex_init, table reset, light pass + its log line, heavy pass, mutexes)
a = lines_of("algod1.cc")
expect(a, 1129, 'progress_init("Hashing sequences:"')
expect(a, 1166, "network_thread")
expect(a, 1171, "progress_done(parameters);")
expect(a, 1403, "const uint64_t n_bytes")
expect(a, 1404, "struct bloomflex_s bloomflex_filter;")
expect(a, 1437, "Generated %")
expect(a, 1465, "bloomflex_exit(bloomflex_filter);")
expect(a, 1467, "pthread_mutex_destroy(&graft_mutex);")
expect(a, 1469, "Heavy variants:")
b1 = [f'#include "{here / "b1.inc"}"']
b2 = [f'#include "{here / "b2.inc"}"']
# (from the bottom up, so that the line numbers above stay valid)
a[1403 - 1:1467] = b2            # n_bytes, bloomflex filter, both thread passes, the "Generated ..." line (b2.inc prints it)
a[1129 - 1:1171] = b1
a.insert(0, '#include "swarm_amd.h"')
(out / "algod1.cc").write_text("\n".join(a))

# ---- algo.cc: tell the bridge d right behind search_begin() (line 335)
g = lines_of("algo.cc")
expect(g, 335, "search_begin(search_data_v);")
g.insert(335, "  { extern void gpu_bridge_configure(int64_t); gpu_bridge_configure(parameters.opt_differences); }")
(out / "algo.cc").write_text("\n".join(g))
print("seams spliced into", out)
