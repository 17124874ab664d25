"""SURVEY 8(e) on a real GPU: the N = 2 flow of bench.py (query range split over ranks, duplicate
flags all-reduced, CSR slices all-gathered) with both ranks sharing GPU 0 and the collectives over
gloo — the same code the driver launches over RCCL, minus the transport.  Rank 0 then computes the
whole network alone and compares it with the gathered CSR."""
import json
import os
import socket
import subprocess
import sys

import pytest

import support as S

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("build", ["records", "routed", "streamed"])
def test_two_ranks_gather_the_whole_network(build):
    """build = records (the default): every rank keys its own slice, the finished key records travel all-to-all
    (sharding.exchange_routed_records), the owners start at the partition; routed: the ids travel (exchange_routed_ids) and
    the owners key them again; streamed: every rank walks the whole database for the keys it owns."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(S.ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--per-gpu", "150000", "--seed", "5", "--dev-backend", "gloo", "--build", build]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["db_amplicons"] == 300000
    assert out["sharded_csr_equals_whole"] is True
    assert out["config"]["neighbour_links"] > 0


EDGE_LIST_SCRIPT = """
import sys
import numpy as np
import torch
torch.zeros(1, device="cuda:0")            # torch initialises the GPU first (as in bench.py)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import support as S
from swarm_amd import Context
db = S.db_from_fasta(sys.argv[2])
ctx = Context(0, torch.cuda.current_stream().cuda_stream)
ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
assert ctx.d1_index_build() is False
def keys_of(off, nb):
    rows = np.repeat(np.arange(len(off) - 1, dtype=np.uint64), np.diff(off).astype(np.int64))
    return (rows << np.uint64(32)) | nb.astype(np.uint64)
for world in (1, 2):
    for rank in range(world):
        ctx.d1_set_ownership(rank, world)
        want = keys_of(*ctx.d1_network())
        buf = torch.zeros(len(want) + 5, dtype=torch.int64, device="cuda:0")
        total = ctx.d1_network_edges_device(buf, buf.numel())
        got = np.sort(buf[:total].cpu().numpy().view(np.uint64))
        assert total == len(want) and np.array_equal(got, want), (world, rank)
        small = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        try:
            ctx.d1_network_edges_device(small, small.numel())
            raise SystemExit("capacity error expected")
        except Exception as e:
            assert "too small" in str(e), e
ctx.close()
print("edge lists ok")
"""


def test_edge_list_holds_the_links_of_the_csr(tmp_path):
    """swa_d1_network_edges_device (the form a multi-GPU job exchanges) against the CSR of the same
    call, complete and per owner; a too-small buffer is reported, not overrun."""
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 20000, 150, 43)
    r = subprocess.run([sys.executable, "-c", EDGE_LIST_SCRIPT, str(S.ROOT), str(fa)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "edge lists ok" in r.stdout
