"""The guard of the d=1 step (swarm_amd/csrc/d1.hip: guard_check): a network whose index lost or
misfiled records is never handed out — SWA_E_INTERNAL instead.  The reference cannot return a partial network
(src/algod1.cc:630-670 is serialised by a mutex); this pipeline is twenty kernels, and the counts they hand to one
another must balance.  Faults are injected on purpose in k_keys (SWA_D1_GUARD_TEST), in fresh processes."""
import os
import subprocess
import sys
import textwrap

import pytest

import support as S

pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import sys
    import numpy as np
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
    import support as S
    from swarm_amd import Context, capi
    db = S.db_from_fasta(sys.argv[2])
    ctx = Context(0)
    ctx.upload_db(db.seqs, db.seq_off, db.seqlen, db.abundance, db.longest)
    try:
        dup = ctx.d1_index_build()
        off, nb = ctx.d1_network()
    except capi.SwaError as e:
        print("ERROR", e.code, str(e))
        sys.exit(0)
    woff, wnb, _ = S.oracle_d1_network(db)
    wnb = wnb.copy()
    for i in range(db.n):
        wnb[int(woff[i]):int(woff[i + 1])].sort()
    print("RETRIES", ctx.d1_guard_retries())
    print("NETWORK", "equal" if (np.array_equal(off, woff) and np.array_equal(nb, wnb)) else "DIFFERENT", len(nb), len(wnb), "width", ctx.d1_anchor_width())
''')


def _run(tmp_path, fault):
    fa = tmp_path / "in.fa"
    S.gen_fasta(fa, 60000, 150, 11)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ)
    env.pop("SWA_D1_GUARD_TEST", None)
    if fault:
        env["SWA_D1_GUARD_TEST"] = fault
    r = subprocess.run([sys.executable, str(script), str(S.ROOT), str(fa)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_a_sound_index_passes_the_guard(tmp_path):
    out = _run(tmp_path, None)
    assert "NETWORK equal" in out, out


def test_misfiled_members_are_caught(tmp_path):
    """a few suffix-side keys are wrong (what the round-1 anomaly looked like): the members sit in groups whose key is not
    theirs, their links would be lost silently — the pair kernels notice (pair_misfiled) and the call fails"""
    out = _run(tmp_path, "misfile")
    assert "ERROR 6" in out and ("not theirs" in out or "key records" in out), out


def test_lost_records_are_caught(tmp_path):
    """records made by k_keys that never reach a group: the counts do not balance and the call fails"""
    out = _run(tmp_path, "drop")
    assert "ERROR 6" in out and "key records" in out, out


@pytest.mark.parametrize("fault", ["misfile-once", "drop-once"])
def test_a_transient_fault_costs_one_repeated_step_not_the_run(tmp_path, fault):
    """the reference cannot fail here (src/algod1.cc:630-670): when the guard trips, the library rebuilds everything
    derived from the packed database and repeats the step once — a fault that does not recur (injected into the first
    build only) ends in the right network, one retry counted, one line on stderr"""
    out = _run(tmp_path, fault)
    assert "NETWORK equal" in out and "RETRIES 1" in out, out


def test_a_sound_run_repeats_nothing(tmp_path):
    assert "RETRIES 0" in _run(tmp_path, None)
