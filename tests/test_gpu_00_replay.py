"""Virtual-rank replay (tools/replay_rank.py): ONE rank of a P-rank sharded evaluation on the one GPU of the test box
with the DEFAULT multi-rank configuration — filler launches on (one work-group per panel CU), the other ranks' block rows played into
the message buffer by a rate-limited copy kernel with a collective's footprint — and its results against the single-GPU path.
Runs in a child process (this file sorts in front of the in-process GPU tests, see test_gpu_00_sharded_lists.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replay(tmp_path, *args, **env):
    out = tmp_path / "replay.json"
    e = dict(os.environ, GPP_SHARD_TIMEOUT_MS="20000", **env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_rank.py"), *[str(a) for a in args], "--json", str(out)],
                       capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.load(open(out)), p.stdout


@pytest.mark.gpu
def test_one_rank_of_four_replayed_against_the_single_gpu_result(tmp_path):
    """P = 4, rank 1, N = 13 000 (13 block rows of 1024, fused groups of 2 steps): rank 1's REAL lists (gpp_shard_list_begin ... _end with
    fillers on, gpp_shard_back_list) with the block rows of ranks 0, 2, 3 replayed at 70 GB/s.  Its owned block rows of the factor, its
    column blocks of L^-1 and of Ky^-1 against gpp_potrf_ws + gpp_trtri + gpp_lauum on one GPU at 1e-11 (bar 1e-5: north_star), both
    lists complete with status 0.  (Reference call sites: optim/mll_torch.py:114-117.)"""
    rec, text = _replay(tmp_path, "--n", 13000, "--d", 8, "--P", 4, "--ranks", 1, "--rates", 70, "--check")
    r = rec["rates"][0]
    assert [q["status"] for q in r["ranks"]] == [[0, 0]], text
    for k, e in r["errors"].items():
        assert e < 1e-11, (k, e, text)
    assert r["ranks"][0]["ff_ms"] > 0 and r["ranks"][0]["back_ms"] > 0


@pytest.mark.gpu
def test_every_rank_of_eight_converges_to_one_timeline(tmp_path):
    """All 8 ranks of a P = 8 run at N = 20 000 (C2), replayed in turn until the times at which the owners can send their block rows
    stop moving: every list completes (fillers on + message kernels resident never executed before round 6), every rank's results
    agree with the single-GPU path, and the sweep converges (the dependencies are triangular in the block index)."""
    rec, text = _replay(tmp_path, "--config", "C2", "--P", 8, "--rates", "150", "--sweeps", 6, "--check")
    r = rec["rates"][0]
    assert all(q["status"] == [0, 0] for q in r["ranks"]) and len(r["ranks"]) == 8, text
    for k, e in r["errors"].items():
        assert e < 1e-11, (k, e, text)
    # (the ready times of a chain-bound C2 rank wander by 0.3-1.5 ms from one replay to the next, the noise of a run; an unconverged
    #  sweep moves them by tens of ms.  Converged = the last sweep moved no ready time by more than 5 % of the evaluation.)
    assert r["sweeps"] <= 6 and r["last_move_ms"] < 0.05 * r["total_ms"], text
