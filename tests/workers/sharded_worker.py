"""Worker of tests/test_gpu_sharded.py: launched by torch.distributed.run, one process per rank.  With fewer GPUs than
ranks (the 1-GPU test box) the ranks share cuda:0 and talk over gloo (host-staged broadcasts); with one GPU per rank
the backend is nccl (= RCCL).  Every rank evaluates the sharded MLL + gradients and rank 0 compares them with the
single-GPU path on the same inputs."""
import os, sys, json


def emit(line: str) -> None:
    """One write() per line: the ranks share the launcher's stdout pipe, and print() may split a line into two writes that
    interleave with another rank's."""
    sys.stdout.flush()
    os.write(1, (line + "\n").encode())

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpplus_amd.linalg import KernelSpec, exact_mll
from gpplus_amd import settings


def same_as_rank0(flat, dev):
    """Every rank must hold the sharded result rank 0 holds, bit for bit: rank 0's values are broadcast and compared."""
    flat = flat.detach().to(dev).reshape(-1).clone()
    ref = flat.clone()
    if dist.get_backend() == "nccl":
        dist.broadcast(ref, 0)
    else:
        h = ref.cpu(); dist.broadcast(h, 0); ref = h.to(dev)
    return bool(torch.equal(flat, ref))


def pick_device_and_backend():
    """One GPU per rank and RCCL when the node has enough GPUs; otherwise every rank on cuda:0 over gloo (host-staged)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    one_each = torch.cuda.device_count() >= world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if one_each else 0)
    torch.cuda.set_device(dev)
    force = os.environ.get("GPP_TEST_BACKEND")  # "nccl": a one-rank RCCL group on the 1-GPU box
    backend = force or ("nccl" if one_each else "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    return dev


def model_case(rank, world, dev, N, nb):
    """The same comparison through the GP_Plus API (mixed inputs: manifold-encoded categoricals, Examples/02 shape)."""
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.test_functions.analytical import borehole_mixed_variables
    from gpplus_amd.preprocessing import standard
    from gpplus_amd.utils import set_seed
    set_seed(4)
    np.random.seed(4)  # (the level draws use the global numpy RNG, as in the reference: every rank must agree)
    X, y = borehole_mixed_variables(n=N, qual_dict={0: 5, 5: 5}, random_state=4, shuffle=False)
    X = torch.tensor(X); y = torch.tensor(y)
    X, _, _ = standard(X, {0: 5, 5: 5})
    out = {}
    for mode in ("sharded", "single"):
        set_seed(7)
        m = GP_Plus(X, y, qual_dict={0: 5, 5: 5}, dtype=torch.float64, device=str(dev))
        m.train()
        mll = ExactMarginalLogLikelihood(m.likelihood, m)
        cfg = {"group": None, "nb": nb} if mode == "sharded" else None
        with settings.sharded_evaluation(cfg):
            loss = -mll(m(*m.train_inputs), m.train_targets)
            loss.backward()
        out[mode] = torch.cat([loss.detach().reshape(1)] + [p.grad.reshape(-1) for p in m.parameters() if p.requires_grad]).cpu()
    if rank == 0:
        e = float((out["sharded"] - out["single"]).abs().max() / out["single"].abs().max())
        from gpplus_amd import sharded as _sh
        emit("RESULT " + json.dumps({"err": {"loss_and_grads": e}, "mll": float(out["single"][0]), "backend": dist.get_backend(),
                                     "list_evals": _sh.LIST_EVALS, "back_list_evals": _sh.BACK_LIST_EVALS}))
    emit(f"RANK{rank} same_as_rank0={same_as_rank0(out['sharded'], dev)}")
    dist.barrier()
    dist.destroy_process_group()


def config_case(name, sharded, nb, n=None, nograd=False):
    """A BASELINE config (FULL size unless ``n`` is given) through GP_Plus (baseline_configs.make_config), sharded over the ranks
    or — one process, no process group — on the single-GPU path; rank 0 prints loss and gradients (``nograd``: the loss of a
    gradient-free evaluation, which skips the back-substitution)."""
    from gpplus_amd.models import GP_Plus
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood
    from gpplus_amd.test_functions.baseline_configs import apply_theta, make_config
    rank = int(os.environ.get("RANK", "0"))
    if sharded:
        dev = pick_device_and_backend()
    else:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
    X, y, kw, theta = make_config(name, n)
    torch.manual_seed(0)
    m = GP_Plus(X, y, dtype=torch.float64, device=str(dev), **kw)
    apply_theta(m, theta)
    m.train()
    mll = ExactMarginalLogLikelihood(m.likelihood, m)
    with settings.sharded_evaluation({"group": None, "nb": nb} if sharded else None):
        if nograd:
            with torch.no_grad():
                loss = -mll(m(*m.train_inputs), m.train_targets)
        else:
            loss = -mll(m(*m.train_inputs), m.train_targets)
            loss.backward()
    if rank == 0:
        vals = {"loss": float(loss)}
        for n, p in m.named_parameters():
            if p.grad is not None:
                vals[n] = p.grad.detach().cpu().reshape(-1).tolist()
        from gpplus_amd import sharded as _sh
        emit("RESULT " + json.dumps({"values": vals, "list_evals": _sh.LIST_EVALS, "back_list_evals": _sh.BACK_LIST_EVALS}))
    if sharded:
        mine = torch.cat([loss.detach().reshape(1)] + [p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])
        emit(f"RANK{rank} same_as_rank0={same_as_rank0(mine, dev)}")
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "config":
        return config_case(sys.argv[2], sys.argv[3] == "sharded", int(sys.argv[4]), (int(sys.argv[5]) or None) if len(sys.argv) > 5 else None,
                           len(sys.argv) > 6 and sys.argv[6] == "nograd")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = pick_device_and_backend()
    N, D, nb, kind, S, dU = (int(a) for a in sys.argv[1:7])
    if len(sys.argv) > 7 and sys.argv[7] == "model":
        return model_case(rank, world, dev, N, nb)
    g = torch.Generator().manual_seed(1234)
    U = torch.rand(N, D, generator=g, dtype=torch.float64)
    y = torch.sin(3.0 * U[:, 0]) + U[:, 1] ** 2 + 0.05 * torch.randn(N, generator=g, dtype=torch.float64)
    grp = (torch.arange(N) % S).to(torch.int32) if S > 1 else None
    variant = sys.argv[7] if len(sys.argv) > 7 else ""
    tau0 = 2e-3
    if variant in ("jitter", "notpsd"):
        # every other row duplicated and a slightly NEGATIVE noise: K + tau I is indefinite until the jitter schedule
        # (1e-8, 1e-7, 1e-6) lifts it ("jitter": from 1e-7 on) or never ("notpsd"): every rank must retry / give up together
        U[1::2] = U[0:2 * (N // 2):2]
        tau0 = -5e-8 if variant == "jitter" else -1e-6
    res = {}
    raised = {}
    import warnings
    warnings.simplefilter("ignore", RuntimeWarning)
    for mode in ("sharded", "single"):
        if mode == "single" and rank != 0:
            continue
        Ud = U.to(dev).requires_grad_(dU > 0)
        w = torch.full((D,), 2.5, dtype=torch.float64, device=dev).requires_grad_(True)
        sf2 = torch.tensor(0.8, dtype=torch.float64, device=dev).requires_grad_(True)
        tau = torch.full((S,), tau0, dtype=torch.float64, device=dev)
        tau = (tau * (1.0 + torch.arange(S, device=dev, dtype=torch.float64))).requires_grad_(True)
        mean = torch.full((N,), 0.1, dtype=torch.float64, device=dev).requires_grad_(True)
        spec = KernelSpec(w=w, sf2=sf2, kind=kind, d_split=2 if kind else 0)
        cfg = {"group": None, "nb": nb} if mode == "sharded" else None
        if mode == "sharded" and variant.startswith("cdriver"):
            # the whole evaluation through the C driver gpp_shard_eval (collectives: callbacks into torch.distributed, or RCCL itself)
            from gpplus_amd.sharded_c import sharded_eval_c
            with torch.no_grad():
                out = sharded_eval_c(Ud, w, sf2, tau, mean, y.to(dev), grp=None if grp is None else grp.to(dev), kind=kind,
                                     d_split=2 if kind else 0, n_grad_dims=dU, group=None, nb=nb, rccl=variant == "cdriver_rccl")
            m_, a_, gw_, gs_, gt_, gU_ = out
            res[mode] = [m_.cpu().reshape(1), gw_.cpu(), gs_.cpu().reshape(1), gt_.cpu(), a_.cpu()] + ([gU_.cpu().reshape(-1)] if dU > 0 else [])
            continue
        try:
            with settings.sharded_evaluation(cfg):
                mll = exact_mll(Ud, spec, tau, mean, y.to(dev), grp=None if grp is None else grp.to(dev), n_grad_dims=dU)
        except Exception as exc:  # noqa: BLE001
            raised[mode] = type(exc).__name__
            continue
        mll.backward()
        res[mode] = [mll.detach().cpu().reshape(1), w.grad.cpu(), sf2.grad.cpu().reshape(1), tau.grad.cpu(),
                     mean.grad.cpu()] + ([Ud.grad.cpu()[:, :dU].reshape(-1)] if dU > 0 else [])
    if variant == "notpsd":
        if rank == 0:
            emit("RESULT " + json.dumps({"raised": raised, "backend": dist.get_backend()}))
        emit(f"RANK{rank} same_as_rank0={raised.get('sharded') == 'NotPSDError'}")
        dist.barrier()
        dist.destroy_process_group()
        return
    # every rank must hold the same sharded result
    same = same_as_rank0(torch.cat([t.reshape(-1) for t in res["sharded"]]), dev)
    if rank == 0:
        names = ["mll", "g_w", "g_sf2", "g_tau", "g_mean"] + (["g_U"] if dU > 0 else [])
        err = {}
        for n, a, b in zip(names, res["sharded"], res["single"]):
            err[n] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
        from gpplus_amd import sharded as _sh
        calls = sum(getattr(w, "comm_calls", 0) for w in _sh._workspaces.values())
        # storage: ONE N x N matrix per rank (the replicated factor), the inverse and Ky^-1 only as owned column blocks
        wsx = next(iter(_sh._workspaces.values()), None)  # (none when the C driver ran: it has buffers of its own)
        emit("RESULT " + json.dumps({"err": err, "mll": float(res["single"][0]), "backend": dist.get_backend(), "collectives": calls,
                                     "matrix_bytes": wsx.nbytes() if wsx else 0, "full_matrix_bytes": 8 * N * (wsx.A.stride(0) if wsx else N),
                                     "owned_cols": wsx.Lc.shape[1] if wsx else 0, "nb": nb, "world": world, "list_evals": _sh.LIST_EVALS, "back_list_evals": _sh.BACK_LIST_EVALS,
                                     # (bit-for-bit comparisons between transports: the sharded result's bytes; messages moved by push)
                                     "digest": __import__("hashlib").sha256(torch.cat([t.reshape(-1).double() for t in res["sharded"]]).numpy().tobytes()).hexdigest(),
                                     "push_messages": _sh._push.MESSAGES}))
    emit(f"RANK{rank} same_as_rank0={same}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
