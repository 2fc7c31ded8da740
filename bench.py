#!/usr/bin/env python3
"""bench.py — MLL evals/sec (fwd+grad) of the exact-GP hot path on MI355X.

A *step* is one evaluation of ``optim/mll_torch.py:112-117`` without ``optimizer.step``:
    output = model(*model.train_inputs); loss = -mll(output, y); loss.backward()
on BASELINE.json's config C2: synthetic Borehole, N = 20 000, d = 8, fp64, Rough_RBF exact GP at
theta1 = (omega = -1, raw_outputscale = 0.3, raw_noise = -6, mean constant = 0.4), inputs resident in HBM.

``python bench.py --gpus N``: with N > 1 and no torch.distributed environment the process starts N ranks of itself
(``python -m torch.distributed.run --nproc-per-node N``, one rank per GPU over RCCL) BEFORE it touches a GPU, relays rank
0's JSON line and exits with the launcher's status; started under torch.distributed.run it is a rank.  Two legs:
  * replicas (the headline ``value``): every rank evaluates its own copy of C2 — the restart-parallel mode of the
    reference's multistart fit (optim/mll_scipy.py:287-293), no data-path collective, ``scaling: weak``;
  * ``sharded`` (N > 1 only, second object of the same JSON line): ALL ranks evaluate ONE C5 problem (N = 60 000, d = 16)
    cooperatively (gp-plus_amd/sharded.py: block-cyclic rows, RCCL broadcasts of factor / inverse slabs), strong scaling.
Every timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.

The JSON line also carries
  roofline      the O(N^3) stage furthest from the fp64 MFMA peak (the factorisation), timed with HIP events on the stream
                the library is driven from; ``entries`` lists all three O(N^3) stages (potrf, trtri, and the single LAUUM
                launch gpp_gemm_f64<2,64,64,1,16,2>) and the whole evaluation;
  stages        per-stage mean milliseconds and rates from the same events;
  cpu_baseline  the CPU oracle (oracle/gp_oracle.py, plain PyTorch fp64) on this box's host cores at the SAME N = 20 000
                workload (rank 0, N = 1 only; see ``cpu_baseline``).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix peak (AMD spec sheet; = 32 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz)
N_C2, D_C2 = 20000, 8
THETA1 = dict(omega=-1.0, raw_os=0.3, raw_noise=-6.0, const=0.4)


def make_c2_data(n=N_C2):
    """Sobol(d=8, seed=0) points scaled to the Borehole bounds, no shuffle (unique rows),
    z-scored with the population std, y = Borehole."""
    from gpplus_amd.test_functions.baseline_configs import make_config

    X, y, _, _ = make_config("C2", n)
    return X, y


def set_theta1(model):
    with torch.no_grad():
        model.covar_module.base_kernel.raw_lengthscale.fill_(THETA1["omega"])
        model.covar_module.raw_outputscale.fill_(THETA1["raw_os"])
        model.likelihood.noise_covar.raw_noise.fill_(THETA1["raw_noise"])
        model.mean_module.constant.fill_(THETA1["const"])


# ---------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------
def _host_cpus():
    """(physical cores, logical CPUs, model name) of this host from /proc/cpuinfo."""
    cores, model, phys, core = set(), "", None, None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return (len(cores) or logical), logical, model


def _free_host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        pass
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def cpu_baseline(mode="full"):
    """Oracle loss+grad (= optim/mll_torch.py:114-117 on the CPU oracle, plain PyTorch fp64 with the Cholesky forced) on
    this box's host cores, on the C2 generator — BASELINE.md §3.

    Thread count: PyTorch's CPU ops oversubscribe badly at these sizes with every hardware thread (256 threads were 10x
    slower than 32 on the 2 x 64-core host of round 1), so the count is CHOSEN by timing N = 8192 with 32, 64 and 128 threads
    (capped at the host's physical cores) and the fastest is used for what follows.
      mode "full" (default)    BASELINE.md §3's protocol: 1 warm-up + median of 3 evaluations at N = 20 000 (~4 min);
      mode "single"            ONE timed evaluation at N = 20 000 after the thread sweep as warm-up (~1.5 min);
      mode "ladder"            the bounded sample only, scaled by N^3 (forced when the host has < 64 GB of free memory).
    ``threads`` is the number of threads used, ``cores`` the host's physical cores."""
    from oracle.gp_oracle import OracleGP

    phys, logical, model = _host_cpus()
    X, y = make_c2_data(N_C2)

    def one(n, threads):
        torch.set_num_threads(threads)
        o = OracleGP(X[:n], y[:n])
        o.params[o.ls_key].fill_(THETA1["omega"])
        o.params["covar_module.raw_outputscale"].fill_(THETA1["raw_os"])
        o.params["likelihood.noise_covar.raw_noise"].fill_(THETA1["raw_noise"])
        o.params["mean_module.constant"].fill_(THETA1["const"])
        t0 = time.perf_counter()
        o.loss_and_grad()
        return time.perf_counter() - t0

    cands = sorted({min(t, phys, logical) for t in (32, 64, 128)})
    one(512, cands[0])  # thread pool, allocator
    ladder = [(n, one(n, cands[0])) for n in (2048, 4096)]
    n_s = 8192
    sweep = {t: one(n_s, t) for t in cands}
    best_th = min(sweep, key=sweep.get)
    t_s = sweep[best_th]
    ladder.append((n_s, t_s))
    free_gb = _free_host_gb()
    if mode != "ladder" and free_gb < 64:
        mode = "ladder"  # autograd through the dense N = 20 000 Cholesky holds ~45 GB of N x N fp64 temporaries
    base = {"unit": "evals/s", "cores": phys, "threads": best_th, "host_physical_cores": phys,
            "host_logical_cpus": logical, "host_cpu": model, "kind": "port",
            "ladder_s": {str(a): round(b, 3) for a, b in ladder}, "ladder_threads": cands[0],
            "thread_sweep_s_at_8192": {str(t): round(v, 3) for t, v in sweep.items()}}
    if mode == "ladder":
        est = t_s * (N_C2 / n_s) ** 3
        base.update(value=1.0 / est, measured_at_N=n_s, seconds_per_eval=est,
                    sample=f"oracle loss+grad timed at N={n_s} ({t_s:.2f} s, {best_th} threads) and scaled by "
                           f"(20000/{n_s})^3 = {est:.0f} s/eval (extrapolated: {free_gb:.0f} GB of host memory free)")
        return base
    if mode == "full":
        one(N_C2, best_th)  # warm-up at full size
        times = [one(N_C2, best_th) for _ in range(3)]
        sec = float(np.median(times))
        what = f"1 warm-up + median of 3 evaluations at N={N_C2} ({[round(t, 1) for t in times]} s)"
    else:
        sec = one(N_C2, best_th)
        times = [sec]
        what = f"ONE evaluation at N={N_C2} after the thread sweep as warm-up"
    base.update(value=1.0 / sec, measured_at_N=N_C2, seconds_per_eval=sec, samples_s=[round(t, 2) for t in times],
                sample=f"oracle loss+grad, {what}: {sec:.1f} s/eval with {best_th} threads (fastest of {cands} at N={n_s}) on "
                       f"{phys} physical cores ({logical} logical, {model}); same C2 generator and theta1 as the GPU leg")
    return base


# ---------------------------------------------------------------------------------------------------
# self-launch
# ---------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus():
    """GPUs of this node WITHOUT touching the HIP runtime (the launcher must stay a process that never initialised a GPU):
    ``GPP_BENCH_NGPUS`` if set, else the KFD topology (nodes with SIMDs are GPUs, CPU nodes have none), narrowed by
    ROCR_/HIP_VISIBLE_DEVICES.  None when the topology is not readable."""
    if os.environ.get("GPP_BENCH_NGPUS"):
        return int(os.environ["GPP_BENCH_NGPUS"])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(root, d, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args, argv):
    """``--gpus N`` (N > 1) outside torch.distributed.run: start N fresh ranks as a child process tree (this process never
    touches the HIP runtime), relay their output line by line and return 0 iff rank 0's JSON line was relayed — whatever
    became of the ranks afterwards (a hung second leg is ended by the ranks' own watchdogs, or here after ``--launch-timeout``)."""
    if not args.dry_run:
        have = count_gpus()
        if not args.share_gpu and (have is None or have < args.gpus):
            print(f"bench.py --gpus {args.gpus}: this node exposes {have or 0} GPU(s)"
                  + ("" if have is not None else " (no KFD topology under /sys/class/kfd)"), file=sys.stderr)
            return 2
    # the ranks' own arguments travel in the environment: on the launcher's command line an option such as ``--n`` is an
    # ambiguous abbreviation of torch.distributed.run's own options and is rejected before it reaches the script
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", GPP_BENCH_ARGV=json.dumps(argv))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)]
    import threading

    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    relayed = {"line": False}

    def pump():
        for line in proc.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
            if line.startswith("{") and '"metric"' in line:
                relayed["line"] = True

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        proc.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the ranks did not finish within {args.launch_timeout} s; ending them", file=sys.stderr)
        try:
            os.killpg(proc.pid, 15)  # the process group this launcher started (start_new_session), nothing else
            proc.wait(timeout=20)
        except Exception:
            try:
                os.killpg(proc.pid, 9)
            except Exception:
                pass
    t.join(timeout=10)
    if relayed["line"]:
        return 0
    return proc.returncode if proc.returncode not in (0, None) else 1


# ---------------------------------------------------------------------------------------------------
# the legs
# ---------------------------------------------------------------------------------------------------
def _bracket(dist, dev, fn):
    """barrier + synchronize, ``fn()``, synchronize + barrier; returns the MAX over ranks of the elapsed seconds."""
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def _stage_table(events, N, share=1):
    """Mean ms per stage and TFLOP/s of the O(N^3) stages (``share`` = ranks that split each of them)."""
    stages = {}
    for name, e0, e1 in events:
        stages.setdefault(name, []).append(e0.elapsed_time(e1))
    ms = {k: float(np.mean(v)) for k, v in stages.items()}
    third = N ** 3 / 3
    flops = {"potrf": third, "trtri": third, "lauum": third, "shard_factor": third, "shard_inverse": third, "shard_backsolve": third}
    rate = {k: flops[k] / share / (ms[k] * 1e-3) / 1e12 for k in flops if k in ms}
    return ms, rate


def _pmc_record(name):
    """HBM-side bytes of a stage from a committed rocprofv3 --pmc collection (it cannot run inside this process).  A record
    names the kernel build it was collected with (``lib_signature`` = the source hash in gpp_version()); one collected with
    another build is REFUSED — traffic is then reported as null with the reason, never silently stale."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    with open(p) as fh:
        rec = json.load(fh)
    from gpplus_amd import _lib

    have = _lib.load().gpp_version().decode()
    sig = rec.get("lib_signature")
    if not sig or sig not in have:
        return {"traffic_bytes_per_launch": None,
                "note": f"profiles/{name} was collected with kernel build '{sig}', this library is '{have}': refused"}
    return rec


def run_replicas(args, dist, dev, rank, world, local_rank):
    from gpplus_amd import linalg
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.models import GP_Plus

    X, y = make_c2_data(args.n)
    S = max(1, args.streams)
    replicas = []
    for k in range(S):
        model = GP_Plus(X, y, dtype=torch.float64, device=dev)
        set_theta1(model)
        model.train()
        replicas.append((model, ExactMarginalLogLikelihood(model.likelihood, model),
                         [p for p in model.parameters() if p.requires_grad]))

    def step(k=0):
        model, mll, params = replicas[k]
        for p in params:
            p.grad = None
        output = model(*model.train_inputs)
        loss = -mll(output, model.train_targets)
        loss.backward()
        return loss

    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []

    def run_steps(nsteps):
        """``nsteps`` evaluations in total; with S > 1 they are dealt to S threads, each on its own stream and slot."""
        if S == 1:
            out = None
            for _ in range(nsteps):
                out = step()
            return out
        import threading

        results = [None] * S
        counts = [nsteps // S + (1 if k < nsteps % S else 0) for k in range(S)]

        def worker(k):
            torch.cuda.set_device(local_rank)
            with torch.cuda.stream(streams[k]), linalg.eval_slot(k):
                for _ in range(counts[k]):
                    results[k] = step(k)
                streams[k].synchronize()

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return next(r for r in results if r is not None)

    run_steps(max(args.warmup, S if S > 1 else 0))
    torch.cuda.synchronize()
    linalg.STAGE_EVENTS = []  # list.append is atomic: the replica threads share it
    elapsed, loss = _bracket(dist, dev, lambda: run_steps(args.steps))
    events, linalg.STAGE_EVENTS = (linalg.STAGE_EVENTS or []), None
    out = None
    if rank == 0:
        N = args.n
        stage_ms, stage_rate = _stage_table(events, N)
        eval_tf = N ** 3 / (elapsed / args.steps) / 1e12
        # HBM-side bytes come from committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot run inside this
        # process); they only apply to the size they were collected at
        pmc_l = _pmc_record("r05_lauum_pmc.json") if N == N_C2 else None
        pmc_p = _pmc_record("r05_potrf_pmc.json") if N == N_C2 else None
        pmc_t = _pmc_record("r05_trtri_pmc.json") if N == N_C2 else None
        third = N ** 3 / 3
        # Above 16 384 rows the leading block of the inverse (8192 rows at C2) is built INSIDE the factorisation's ticket list, i.e.
        # inside the "potrf" stage's time, and gpp_trtri merges only the rest: the stages' flop follow the work each really does
        lead = 0
        try:
            import ctypes as _ct
            from gpplus_amd.backend import get_context as _gc
            _info = (_ct.c_int64 * 10)()
            _c = _gc(dev)
            if _c.lib.gpp_debug_dag_info(_c.h, _info) == 0 and int(_info[9]) == N and 0 < int(_info[8]) < N:
                lead = int(_info[8])
        except Exception:  # noqa: BLE001 (a library without the debug entry point: no shift)
            lead = 0
        shift = (lead ** 3) / 3
        stage_flops = {"potrf": third + shift, "trtri": third - shift, "lauum": third}
        for k, fl in stage_flops.items():
            if k in stage_ms:
                stage_rate[k] = fl / (stage_ms[k] * 1e-3) / 1e12
        entries = []
        for name, kernel, pmc in (
                ("potrf", "gpp_potrf_ws: blocked Cholesky as ONE ticket list of 128 x 128 tile tasks (row solves, trailing updates, strip "
                          "copies; host-planned topological order, gpp_dag.hip) taken by the persistent work-groups of gpp_dag_f64 — 448 "
                          "on the 224 throughput CUs + filler launches of 64 on the panel's 32 CUs between gpp_panel_potrf_inv launches.  "
                          "Traffic record: the same kernel over the same list as a sequence of launches (GPP_DAG_PHASED=1: counter "
                          "collection serialises dispatches)", pmc_p),
                ("trtri", "gpp_trtri: batched pair merges, gpp_gemm_f64<2,64,64,0,16,2>", pmc_t),
                ("lauum", "gpp_gemm_f64<2, 64, 64, 1, 16, 2> (Kinv = Linv^T Linv, ONE lower-triangular TN launch)", pmc_l)):
            if name in stage_rate:
                entries.append({"stage": name, "kernel": kernel, "achieved": stage_rate[name],
                                "frac": stage_rate[name] / PEAK_FP64_MFMA_TFLOPS, "flops": stage_flops[name], "ms": stage_ms[name],
                                "traffic": None if pmc is None else pmc["traffic_bytes_per_launch"],
                                "traffic_source": None if pmc is None else pmc.get("note")})
        entries.append({"stage": "whole evaluation (N^3 flop, driver-timed)", "achieved": eval_tf,
                        "frac": eval_tf / PEAK_FP64_MFMA_TFLOPS, "flops": N ** 3, "ms": 1e3 * elapsed / args.steps})
        o3 = [e for e in entries if e["stage"] in ("potrf", "trtri", "lauum")]
        worst = min(o3, key=lambda e: e["frac"]) if o3 else None
        out = {
            "metric": "MLL evals/sec (fwd+grad), NxN exact GP, N=20k d=8",
            "value": world * args.steps / elapsed, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: synthetic Borehole (Sobol seed 0, unique rows, z-scored) N={N} d={D_C2} fp64, "
                                   "GP_Plus Rough_RBF exact GP at theta1, one replica per GPU (restart-parallel)",
                       "N": N, "d": D_C2, "loss": float(loss.item()), "streams_per_gpu": S},
            "roofline": {"bound": "mfma",
                         "kernel": None if worst is None else f"{worst['stage']} stage — {worst['kernel']}",
                         "achieved": None if worst is None else worst["achieved"], "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": None if worst is None else worst["frac"],
                         "traffic": None if worst is None else worst["traffic"], "traffic_unit": "B/launch",
                         "traffic_source": None if worst is None else worst["traffic_source"],
                         "flops_per_launch": None if worst is None else worst["flops"], "ms_per_launch": None if worst is None else worst["ms"],
                         "note": "the O(N^3) stage furthest from peak; 'entries' lists all three and the whole evaluation"
                                 + (f"; the leading {lead} rows of the inverse ({shift:.3e} flop) are built inside the potrf stage's "
                                    "ticket list and counted there, not under trtri" if lead else ""),
                         "entries": entries},
            "stages": {"ms": stage_ms, "tflops": stage_rate, "eval_tflops_N3": eval_tf},
        }
    del replicas
    linalg._workspaces.clear()
    torch.cuda.empty_cache()
    return out


def run_sharded(args, dist, dev, rank, world, n, steps, warmup):
    """ONE evaluation shared by all ranks (C5 generator at size ``n``; the C2 generator when ``--mode sharded --n 20000``)."""
    from gpplus_amd import linalg
    from gpplus_amd import settings as gpp_settings
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config

    cfg_name = "C5" if n > 30000 else "C2"
    if not args.nb:
        # (the ticket lists of round 5 need the cooperative panel: blocks of 1024 rows; the launch-per-product path of rounds 2-4 was
        #  faster with 2048 from N = 40 000)
        lists = os.environ.get("GPP_SHARD_LIST", "1") not in ("", "0")
        args.nb = 1024 if (lists or n < 40000) else 2048
    X, y, kw, theta = make_config(cfg_name, n)
    model = GP_Plus(X, y, dtype=torch.float64, device=dev, **kw)
    apply_theta(model, theta)
    model.train()
    mll = ExactMarginalLogLikelihood(model.likelihood, model)
    params = [p for p in model.parameters() if p.requires_grad]
    shard_cfg = {"group": None, "nb": args.nb}

    def step():
        for p in params:
            p.grad = None
        with gpp_settings.sharded_evaluation(shard_cfg):
            loss = -mll(model(*model.train_inputs), model.train_targets)
            loss.backward()
        return loss

    def run_steps(k):
        out = None
        for _ in range(k):
            out = step()
        return out

    from gpplus_amd import sharded as _sh0

    run_steps(warmup)
    lists0 = (_sh0.LIST_EVALS, _sh0.BACK_LIST_EVALS)
    linalg.STAGE_EVENTS = []
    _sh0.COMM_LOG = []  # every collective of the timed evaluations is bracketed by events on its stream
    elapsed, loss = _bracket(dist, dev, lambda: run_steps(steps))
    events, linalg.STAGE_EVENTS = (linalg.STAGE_EVENTS or []), None
    comm_log, _sh0.COMM_LOG = _sh0.COMM_LOG, None
    torch.cuda.synchronize()
    comm = {k: {"calls": v["calls"] // steps, "bytes": v["bytes"] // steps, "comm_ms": v["comm_ms"] / steps}
            for k, v in _sh0.comm_report(comm_log).items()}
    out = None
    # what a reader needs to see that the collective backend really formed `world` ranks on `world` devices, and what each holds
    mine = {"rank": rank, "device": int(torch.cuda.current_device()),
            "matrix_gb": round(sum(w.nbytes() for w in _sh0._workspaces.values()) / 1e9, 3)}
    ranks = [mine]
    rccl = None
    if dist is not None and dist.is_initialized():
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, mine)
        if dist.get_backend() == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # (reported, not fatal: the line still says which backend ran)
                rccl = f"unknown ({type(e).__name__})"
    if rank == 0:
        stage_ms, stage_rate = _stage_table(events, n, share=world)
        per_gpu_tf = n ** 3 / (elapsed / steps) / 1e12 / world
        D = X.shape[1]
        out = {"metric": "MLL evals/sec (fwd+grad), ONE evaluation sharded over all GPUs", "value": steps / elapsed,
               "unit": "evals/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
               "scaling": "strong", "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{cfg_name}: synthetic N={n} d={D} fp64 exact GP, kernel build + blocked Cholesky + "
                                      f"inverse + gradient sharded block-cyclically (nb={args.nb}) over {world} GPU(s), "
                                      "RCCL broadcasts of factor / inverse slabs", "N": n, "d": D, "nb": args.nb,
                          "loss": float(loss.item()), "backend": "none" if dist is None else dist.get_backend(),
                          "world": 1 if dist is None else dist.get_world_size(), "rccl_version": rccl, "ranks": ranks,
                          "full_matrix_gb": round(8e-9 * n * n, 3)},
               "roofline": {"bound": "mfma", "kernel": "whole sharded evaluation, N^3 flop / (time x GPUs)",
                            "achieved": per_gpu_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s per GPU",
                            "frac": per_gpu_tf / PEAK_FP64_MFMA_TFLOPS, "traffic": None},
               "stages": {"ms": stage_ms, "tflops_per_gpu": stage_rate},
               # timed evaluations whose factorisation + forward sweep / back-substitution ran as ticket lists to completion
               "ticket_lists": {"factor_forward": _sh0.LIST_EVALS - lists0[0], "back": _sh0.BACK_LIST_EVALS - lists0[1], "of": steps},
               # per evaluation, rank 0's view: collectives issued, bytes they carried and the time they occupied their stream (the
               # factor's broadcasts run on the communication stream beside the updates: comm_ms is NOT all exposed time)
               "comm": comm}
    del model
    from gpplus_amd import sharded as _sh
    _sh._workspaces.clear()
    torch.cuda.empty_cache()
    return out


def start_watchdog(seconds, rank, line, dist):
    """Timer on EVERY rank around the sharded leg: when it fires, rank 0 prints the replica line (``line``, a dict) with the
    failure recorded, then each rank tries to tear the process group down (bounded) and leaves with a NON-zero status — a hung
    collective must not read as success, and no rank stays behind inside it.  The launcher (spawn_ranks) returns 0 iff the
    line was relayed.  Returns the timer (cancel it when the leg returns) or None."""
    if seconds <= 0:
        return None
    import threading

    def give_up():
        if rank == 0:
            line["sharded"] = {"error": f"no result within {seconds} s (hang in the sharded leg)"}
            print(json.dumps(line), flush=True)
        closer = threading.Thread(target=lambda: dist.destroy_process_group(), daemon=True)
        closer.start()
        closer.join(timeout=10)
        os._exit(3)

    t = threading.Timer(seconds, give_up)
    t.daemon = True
    t.start()
    return t


def dry_run(args, rank, world):
    """Launch-path check without GPUs: gloo rendezvous, barrier, MAX all-reduce, one JSON line from rank 0."""
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world - 1
    line = {"metric": "MLL evals/sec (fwd+grad), NxN exact GP, N=20k d=8", "value": None, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True}
    if args.fake_hang and world > 1:
        # a second leg that never returns on the other ranks: the watchdog path of the real run, without GPUs
        watchdog = start_watchdog(args.sharded_timeout, rank, line, dist)
        if rank != 0:
            time.sleep(10 ** 6)
        dist.barrier()  # (never completes)
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=None, help="problem size (default: 20000 = C2; 60000 = C5 for --mode sharded)")
    ap.add_argument("--cpu-baseline", choices=["single", "full", "ladder", "none"], default="full",
                    help="full (default): BASELINE.md section 3 - 1 warm-up + median of 3 evaluations at N=20000 (~4 min of host time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent replicas evaluated concurrently on this GPU (one host thread + HIP stream + "
                         "workspace slot each), the per-GPU form of restart parallelism")
    ap.add_argument("--mode", choices=["auto", "replicas", "sharded"], default="auto",
                    help="auto (default): the replica leg, plus the sharded C5 leg when N > 1.  replicas / sharded: that "
                         "leg only (sharded with one rank measures the algorithm without communication)")
    ap.add_argument("--nb", type=int, default=0,
                    help="block height of the sharded evaluation (0 = 1024, what the ticket lists of round 5 need — one rank at C5: "
                         "3404 ms, at C2: 137 ms; with GPP_SHARD_LIST=0, the launches of rounds 2-4: 2048 from N = 40000 — 3450 ms "
                         "against 3554 with 1024 —, 1024 below)")
    ap.add_argument("--sharded-n", type=int, default=60000, help="size of the sharded leg (C5)")
    ap.add_argument("--sharded-steps", type=int, default=2)
    ap.add_argument("--sharded-warmup", type=int, default=1)
    ap.add_argument("--sharded-timeout", type=float, default=600.0, help="seconds before rank 0 gives up on the sharded leg")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launcher waits for its ranks")
    ap.add_argument("--dry-run", action="store_true", help="exercise the launch path only (gloo, no GPU work)")
    ap.add_argument("--fake-hang", action="store_true", help="with --dry-run: the other ranks never return (tests the watchdog)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE: all ranks use cuda:0 and talk over gloo (exercises the N > 1 code path of both legs on a "
                         "1-GPU box; not a measurement)")
    forwarded = os.environ.get("GPP_BENCH_ARGV") if "WORLD_SIZE" in os.environ and len(sys.argv) == 1 else None
    args = ap.parse_args(json.loads(forwarded)) if forwarded else ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the exact-GP path has no CPU fallback")
    if args.share_gpu or args.streams > 1:
        # several tenants on ONE GPU: two cooperative panel launches (one work-group per CU of the same 32-CU set) can each hold
        # part of those CUs and wait for the rest — the library would detect that after ~1 s and the host would switch the panel
        # off for the context (tools/tenant_probe.py); start without it instead
        os.environ["GPP_COOP_PANEL"] = "0"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    elif args.mode == "sharded":
        import torch.distributed as dist  # the sharded evaluation wants a process group even when it has one rank

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("gloo", rank=0, world_size=1)

    out = None
    if args.mode in ("auto", "replicas"):
        if args.n is None:
            args.n = N_C2
        out = run_replicas(args, dist, dev, rank, world, local_rank)
        if args.mode == "auto" and world > 1:
            # the second leg must never cost the first its JSON line: a failure is reported inside the line (all ranks reach
            # the same branch: the sharded evaluation fails or succeeds collectively, and a hang is bounded by the
            # process-group timeout)
            # ... and neither can a hang: after ``--sharded-timeout`` seconds rank 0 prints the replica line with the failure
            # recorded and every rank leaves (non-zero)
            watchdog = start_watchdog(args.sharded_timeout, rank, out, dist)
            try:
                sh = run_sharded(args, dist, dev, rank, world, args.sharded_n, args.sharded_steps, args.sharded_warmup)
            except Exception as exc:  # noqa: BLE001
                sh = {"error": f"{type(exc).__name__}: {exc}"[:500]}
            if watchdog is not None:
                watchdog.cancel()
            if rank == 0:
                out["sharded"] = sh
    else:
        out = run_sharded(args, dist, dev, rank, world, args.n or args.sharded_n, args.steps, args.warmup)
        if rank == 0:
            out["higher_is_better"], out["vs_baseline"] = True, None
    if rank == 0:
        if world == 1 and args.mode != "sharded" and not args.no_cpu_baseline and args.cpu_baseline != "none":
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
