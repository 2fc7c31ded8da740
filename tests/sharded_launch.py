"""Launcher shared by the sharded GPU tests: ``world`` ranks of tests/workers/sharded_worker.py through torch.distributed.run (on the
1-GPU test box the ranks share cuda:0 and talk over gloo), rank 0's RESULT line parsed, every rank's agreement with rank 0 checked."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(args, world=2, port=29531, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPP_SHARD_DEBUG="1", **extra_env)  # (debug: a list's time-out status is printed)
    env.setdefault("GPP_SHARD_TIMEOUT_MS", "20000")  # (a stalled list gives up after 20 s and the evaluation falls back to the launches)
    if world > 1:  # the ranks share the one GPU: the ticket lists' persistent work-groups of all ranks must fit on it together
        env.setdefault("GPP_SHARD_WORKERS", str(448 // world))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "sharded_worker.py")] + [str(a) for a in args]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    import re
    # (robust against ranks sharing a line of the launcher's pipe: parse from the marker, not by lines)
    res = [json.JSONDecoder().raw_decode(p.stdout, m.end())[0] for m in re.finditer(r"RESULT (?=\{)", p.stdout)]
    same = re.findall(r"same_as_rank0=(True|False)", p.stdout)
    assert len(res) == 1 and len(same) == world, p.stdout[-3000:]
    assert all(v == "True" for v in same), same
    res[0]["status_lines"] = [l[:300] for l in p.stdout.splitlines() if l.startswith("[sharded rank")]
    return res[0]


def config_values(name, nb, world, n=None, nograd=False, port=29977, timeout=1500, meta=None, only=None, **extra_env):
    """Loss and gradients of a BASELINE config through GP_Plus: (single-GPU path, sharded over ``world`` ranks); ``only="sharded"``
    skips the single-GPU run (its slot is None) — for comparisons with a committed fixture instead."""
    import re

    worker = os.path.join(ROOT, "tests", "workers", "sharded_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPP_SHARD_DEBUG="1", **extra_env)
    env.setdefault("GPP_SHARD_TIMEOUT_MS", "20000")
    if world > 1:
        env.setdefault("GPP_SHARD_WORKERS", str(448 // world))
    extra = ([str(n)] if n else []) + (["nograd"] if nograd else [])
    if nograd and not n:
        extra = ["0", "nograd"]

    def values(cmd):
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        res = [json.JSONDecoder().raw_decode(p.stdout, m.end())[0] for m in re.finditer(r"RESULT (?=\{)", p.stdout)]
        assert len(res) == 1, p.stdout[-3000:]
        same = re.findall(r"same_as_rank0=(True|False)", p.stdout)
        assert all(v == "True" for v in same), same
        if meta is not None:
            meta.update({k: v for k, v in res[0].items() if k != "values"})
            meta["status_lines"] = [l[:300] for l in p.stdout.splitlines() if l.startswith("[sharded rank")]
        return res[0]["values"]

    single = None if only == "sharded" else values([sys.executable, worker, "config", name, "single", str(nb)] + extra)
    shard = values([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                    "127.0.0.1", "--master-port", str(port), worker, "config", name, "sharded", str(nb)] + extra)
    return single, shard


def assert_close_values(single, shard, tol):
    assert set(single) == set(shard)
    for k, ref in single.items():
        a, b = (shard[k], ref) if isinstance(ref, list) else ([shard[k]], [ref])
        scale = max(max(abs(v) for v in b), 1e-300)
        assert max(abs(x - y) for x, y in zip(a, b)) <= tol * scale, (k, a, b)
