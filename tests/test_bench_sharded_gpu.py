"""SURVEY 8(e) on a real GPU: the N = 2 flow of bench.py (query range split over ranks, duplicate
flags all-reduced, CSR slices all-gathered) with both ranks sharing GPU 0 and the collectives over
gloo — the same code the driver launches over RCCL, minus the transport.  Rank 0 then computes the
whole network alone and compares it with the gathered CSR."""
import json
import os
import subprocess
import sys

import pytest

import support as S

pytestmark = pytest.mark.gpu


def test_two_ranks_gather_the_whole_network():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", str(S.ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--per-gpu", "150000", "--seed", "5", "--dev-backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["db_amplicons"] == 300000
    assert out["sharded_csr_equals_whole"] is True
    assert out["config"]["neighbour_links"] > 0
