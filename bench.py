#!/usr/bin/env python3
"""bench.py — MLL evals/sec (fwd+grad) of the exact-GP hot path on MI355X.

A *step* is one evaluation of ``optim/mll_torch.py:112-117`` without ``optimizer.step``:
    output = model(*model.train_inputs); loss = -mll(output, y); loss.backward()
on BASELINE.json's config C2: synthetic Borehole, N = 20 000, d = 8, fp64, Rough_RBF exact GP at
theta1 = (omega = -1, raw_outputscale = 0.3, raw_noise = -6, mean constant = 0.4), inputs resident in HBM.

Multi-GPU (``--gpus N``, launched by torch.distributed.run, one rank per GPU): every rank runs its own replica of the
workload — the restart-parallel mode of the reference's multistart fit (optim/mll_scipy.py:287-293) — with no
data-path collective; the timed region is bracketed by barrier + synchronize and the MAX over ranks is reported
(``scaling: weak``).

The JSON line also carries
  roofline      the dominant MFMA kernel (the lower-triangular TN launch of the fp64 GEMM that forms Ky^-1 = Linv^T Linv,
                N^3/3 flop in ONE launch) timed with HIP events on its own stream inside the timed region;
  stages        per-stage mean milliseconds and rates from the same events;
  cpu_baseline  the CPU oracle (oracle/gp_oracle.py, plain PyTorch fp64) on this box's host cores, on a bounded sample
                (smaller N, scaled by N^3), rank 0 and N=1 only.
"""
import argparse
import json
import os
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
    from gpplus_amd.preprocessing import standard
    from gpplus_amd.test_functions.analytical import borehole

    X, y = borehole(n=n, random_state=0, shuffle=False)
    Xs, _, _ = standard(torch.tensor(X), {})
    return Xs.double(), torch.tensor(y).double()


def set_theta1(model):
    with torch.no_grad():
        model.covar_module.base_kernel.raw_lengthscale.fill_(THETA1["omega"])
        model.covar_module.raw_outputscale.fill_(THETA1["raw_os"])
        model.likelihood.noise_covar.raw_noise.fill_(THETA1["raw_noise"])
        model.mean_module.constant.fill_(THETA1["const"])


def cpu_baseline(budget_s=30.0):
    """Oracle loss+grad (= optim/mll_torch.py:114-117 on the CPU oracle) on this box's host cores.  Bounded sample: one
    untimed warm-up, then N = 2048, 4096, 8192 while the N^3 projection of the next size fits the budget; the largest
    timed size is scaled to N = 20000 by N^3 (the evaluation is dominated by the O(N^3) Cholesky backward).  Thread
    count: PyTorch CPU ops oversubscribe badly at these sizes with every hardware thread (256 threads are 10x slower
    than 32 on a 2 x 64-core host), so the ladder runs with min(32, cores) threads and the last size is re-timed with
    4x as many when the budget allows; the faster of the two is reported together with its thread count."""
    from oracle.gp_oracle import OracleGP

    ncpu = os.cpu_count() or 1
    X, y = make_c2_data(8192)

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

    th = min(32, ncpu)
    one(512, th)  # warm-up: thread pool, allocator
    used, sizes = 0.0, []
    for n in (2048, 4096, 8192):
        if sizes and used + sizes[-1][1] * 8.0 > budget_s:
            break
        t = one(n, th)
        used += t
        sizes.append((n, t))
    n_s, t_s = sizes[-1]
    best_th = th
    th2 = min(4 * th, ncpu)
    if th2 > th and used + 2.0 * t_s < budget_s:
        t2 = one(n_s, th2)
        if t2 < t_s:
            t_s, best_th = t2, th2
    est = t_s * (N_C2 / n_s) ** 3
    return {"value": 1.0 / est, "unit": "evals/s", "cores": best_th, "kind": "port",
            "sample": f"oracle loss+grad timed at N={n_s} ({t_s:.2f} s with {best_th} threads of {ncpu}, same C2 "
                      f"generator, 1 warm-up) and scaled by (20000/{n_s})^3 = {est:.0f} s/eval; "
                      f"ladder at {th} threads {[(a, round(b, 2)) for a, b in sizes]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=N_C2, help="problem size (default: the C2 config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent replicas evaluated concurrently on this GPU (one host thread + HIP stream + "
                         "workspace slot each), the per-GPU form of restart parallelism")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="replicas (default): every GPU evaluates its own copy of the C2 problem (restart parallelism, weak "
                         "scaling).  sharded: ALL GPUs evaluate ONE problem cooperatively (gp-plus_amd/sharded.py, strong "
                         "scaling; use --n 60000 for the C5 size)")
    ap.add_argument("--nb", type=int, default=1024, help="block height of the sharded evaluation")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the exact-GP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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

    from gpplus_amd import settings as gpp_settings
    shard_cfg = {"group": None, "nb": args.nb} if (args.mode == "sharded" and world > 1) else None

    def step(k=0):
        model, mll, params = replicas[k]
        for p in params:
            p.grad = None
        with gpp_settings.sharded_evaluation(shard_cfg):
            output = model(*model.train_inputs)
            loss = -mll(output, model.train_targets)
            loss.backward()
        return loss

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

    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []
    run_steps(max(args.warmup, S if S > 1 else 0))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    linalg.STAGE_EVENTS = []  # list.append is atomic: the replica threads share it
    t0 = time.perf_counter()
    loss = run_steps(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events, linalg.STAGE_EVENTS = (linalg.STAGE_EVENTS or []), None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        N = args.n
        stages = {}
        for name, e0, e1 in events:
            stages.setdefault(name, []).append(e0.elapsed_time(e1))
        if os.environ.get("GPP_BENCH_DEBUG"):
            print({k: [round(x, 2) for x in v] for k, v in stages.items()}, file=sys.stderr)
        stage_ms = {k: float(np.mean(v)) for k, v in stages.items()}
        flops = {"potrf": N ** 3 / 3, "trtri": N ** 3 / 3, "lauum": N ** 3 / 3}
        stage_rate = {k: flops[k] / (stage_ms[k] * 1e-3) / 1e12 for k in flops if k in stage_ms}
        lauum_tflops = stage_rate.get("lauum")  # None in sharded mode (no single LAUUM launch there)
        sharded = shard_cfg is not None
        value = (1 if sharded else world) * args.steps / elapsed  # sharded: the ranks share every evaluation
        # HBM-side bytes of the roofline kernel come from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # cannot run inside this process); they only apply to the size they were collected at
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_lauum_pmc.json")
        if N == 20000 and os.path.exists(pmc):
            with open(pmc) as fh:
                rec = json.load(fh)
            traffic, traffic_src = rec["traffic_bytes_per_launch"], "profiles/r01_lauum_pmc.json: " + rec["note"]
        out = {
            "metric": "MLL evals/sec (fwd+grad), NxN exact GP, N=20k d=8",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: synthetic Borehole (Sobol seed 0, unique rows, z-scored) N={N} d={D_C2} fp64, "
                                   "GP_Plus Rough_RBF exact GP at theta1, " +
                                   ("ONE evaluation sharded over all GPUs (block-cyclic rows, RCCL panel broadcasts)" if sharded
                                    else "one replica per GPU (restart-parallel)"),
                       "N": N, "d": D_C2, "loss": float(loss.item()), "streams_per_gpu": S},
            "roofline": {"bound": "mfma", "kernel": "gpp_gemm_f64<2, 64, 64, 1, 16, 2> (LAUUM: Kinv = Linv^T Linv, the one lower-triangular TN launch)",
                         "achieved": lauum_tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": None if lauum_tflops is None else lauum_tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "B/launch",
                         "traffic_source": traffic_src,
                         "flops_per_launch": N ** 3 / 3, "ms_per_launch": stage_ms.get("lauum")},
            "stages": {"ms": stage_ms, "tflops": stage_rate,
                       "eval_tflops_N3": N ** 3 / (elapsed / args.steps) / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
