"""CPU tests of the host side: C-ABI export list, the gpytorch-protocol mirror (names, transforms, priors, state_dict
keys), the data-prep helpers against the reference-generated fixtures, and the fail-loudly rule.  No compute call is
made into the HIP library here."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def test_library_exports_every_declared_symbol():
    from gpplus_amd import _lib

    header = open(os.path.join(ROOT, "include", "gpp.h")).read()
    declared = set(re.findall(r"\b(gpp_[a-z_0-9]+)\s*\(", header))
    declared -= {"gpp_handle_s"}
    lib = _lib.load()
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.gpp_version()


def test_data_pipeline_reproduces_reference_fixtures():
    """Our borehole / split / standard must give exactly what the reference's own functions gave (fixture inputs)."""
    from gpplus_amd.preprocessing import train_test_split_normalizeX
    from gpplus_amd.test_functions.analytical import borehole, borehole_mixed_variables
    from gpplus_amd.utils import set_seed

    fx = load("c1_borehole_n500.npz")
    set_seed(1245)
    X, y = borehole(n=10000, random_state=12345)
    Xtr, Xte, ytr, yte = train_test_split_normalizeX(X, y, test_size=0.95)
    np.testing.assert_array_equal(Xtr.numpy(), fx["Xtrain"])
    np.testing.assert_array_equal(ytr.numpy(), fx["ytrain"])
    np.testing.assert_array_equal(Xte.numpy()[:200], fx["Xtest"])
    assert len(np.unique(fx["Xtrain"], axis=0)) == 490  # duplicate rows of the with-replacement shuffle (B-1)

    fx = load("c3_borehole_mixed_n100.npz")
    set_seed(4)
    qd = {0: 5, 5: 5}
    U, y = borehole_mixed_variables(n=10000, qual_dict=qd, random_state=4)
    Utr, Ute, ytr, yte = train_test_split_normalizeX(U, y, test_size=0.99, qual_dict=qd)
    np.testing.assert_array_equal(Utr.numpy(), fx["Utrain"])
    np.testing.assert_array_equal(ytr.numpy(), fx["ytrain"])


def test_setlevels_and_standard():
    from gpplus_amd.preprocessing import setlevels, standard

    X = torch.tensor([[3.5, 10.0], [1.5, 20.0], [3.5, 30.0], [2.0, 10.0]])
    out, labels = setlevels(X, qual_index=[0], return_label=True)
    np.testing.assert_array_equal(out[:, 0].numpy(), [2, 0, 2, 1])
    assert labels == [[1.5, 2.0, 3.5]]
    Xs, mean, std = standard(X.clone().double(), {0: 3})
    np.testing.assert_allclose(Xs[:, 1].numpy(), (X[:, 1].numpy() - 17.5) / np.std([10, 20, 30, 10]), rtol=1e-12)


def _mixed_model(**kw):
    from gpplus_amd.models import GP_Plus

    fx = load("c3_borehole_mixed_n100.npz")
    return fx, GP_Plus(torch.tensor(fx["Utrain"]), torch.tensor(fx["ytrain"]), qual_dict={0: 5, 5: 5}, dtype=torch.float64, **kw)


def test_gp_plus_structure_matches_reference_names():
    fx, m = _mixed_model()
    sd = m.state_dict()
    for k in ("likelihood.noise_covar.raw_noise", "covar_module.raw_outputscale",
              "covar_module.base_kernel.kernels.0.raw_lengthscale", "covar_module.base_kernel.kernels.1.raw_lengthscale",
              "mean_module.constant", "latent[0, 5]", "y_min", "y_std", "y_scaled", "quant_index", "qual_dict_list"):
        assert k in sd, k
    assert sd["covar_module.base_kernel.kernels.1.raw_lengthscale"].shape == (1, 6)
    assert sd["latent[0, 5]"].shape == (2, 10)
    assert not m.covar_module.base_kernel.kernels[0].raw_lengthscale.requires_grad
    names = [n for n, *_ in m.named_priors()]
    assert names == ["latent_prior_latent[0, 5]", "likelihood.noise_prior", "covar_module.outputscale_prior",
                     "covar_module.base_kernel.kernels.1.lengthscale_prior", "mean_module.mean_prior"]
    # y scaling (gpregression.py:67-69)
    y = torch.tensor(fx["ytrain"])
    np.testing.assert_allclose(m.train_targets.numpy(), ((y - y.min()) / (y.max() - y.min())).numpy(), rtol=1e-14)


def test_weights_transforms_and_lazy_forward():
    fx, m = _mixed_model()
    with torch.no_grad():
        m.covar_module.base_kernel.kernels[1].raw_lengthscale.copy_(torch.tensor([[-1.0, 0.0, 0.5, 1.0, -2.0, 0.25]]))
        m.covar_module.raw_outputscale.fill_(0.3)
        m.likelihood.noise_covar.raw_noise.fill_(-6.0)
    out = m(*m.train_inputs)
    cov = out.lazy_covariance_matrix
    omega = np.array([-1.0, 0.0, 0.5, 1.0, -2.0, 0.25])
    np.testing.assert_allclose(cov.spec.w.detach().numpy(), np.concatenate([[0.5, 0.5], 10.0 ** omega]), rtol=1e-13)
    np.testing.assert_allclose(cov.spec.sf2.item(), np.log1p(np.exp(0.3)), rtol=1e-14)
    assert cov.shape == (100, 100) and cov.n_grad_dims == 2
    noisy = m.likelihood(out).lazy_covariance_matrix
    np.testing.assert_allclose(noisy.tau.detach().numpy(), [np.exp(-6.0) + 1e-8], rtol=1e-14)
    np.testing.assert_allclose(noisy.diag().detach().numpy(), np.log1p(np.exp(0.3)) + np.exp(-6.0) + 1e-8, rtol=1e-13)
    # manifold features: z = zeta[index] @ A^T with the reference's str(list) dictionary order
    A = m.state_dict()["latent[0, 5]"]
    Utr = torch.tensor(fx["Utrain"])
    idx = [m.perm_dict[0][str(r.tolist())] for r in Utr[:, [0, 5]].to(torch.int64)]
    z = m.zeta[0][idx].double() @ A.T
    np.testing.assert_allclose(cov.U1[:, :2].detach().numpy(), z.numpy(), rtol=1e-13)
    np.testing.assert_array_equal(cov.U1[:, 2:].detach().numpy(), fx["Utrain"][:, [1, 2, 3, 4, 6, 7]])


def test_multifidelity_noise_and_means():
    from gpplus_amd.models import GP_Plus

    fx = load("c4_wing_mf_n300.npz")
    m = GP_Plus(torch.tensor(fx["Xtrain"]), torch.tensor(fx["ytrain"]), qual_dict={10: 3}, multiple_noise=True,
                m_gp="multiple_constant", dtype=torch.float64)
    with torch.no_grad():
        m.likelihood.noise_covar.raw_noise.copy_(torch.log(torch.tensor([1e-4, 4e-4, 9e-4], dtype=torch.float64)))
        m.mean_module_1.constant.fill_(0.1)
        m.mean_module_2.constant.fill_(-0.2)
    out = m(*m.train_inputs)
    src = fx["Xtrain"][:, -1].astype(int)
    np.testing.assert_allclose(out.mean.detach().numpy(), np.array([0.0, 0.1, -0.2])[src], rtol=1e-14)
    noisy = m.likelihood(out).lazy_covariance_matrix
    np.testing.assert_allclose(noisy.noise_vector().detach().numpy(), (np.array([1e-4, 4e-4, 9e-4]) + 1e-8)[src], rtol=1e-10)
    assert isinstance(m.mean_module_0, type(m.mean_module_0)) and not list(m.mean_module_0.parameters())  # ZeroMean
    assert m.likelihood.raw_noise.shape == (3,)


def test_reset_parameters_and_fixed_noise():
    from gpplus_amd.models import GP_Plus

    fx = load("c1_borehole_n500.npz")
    m = GP_Plus(torch.tensor(fx["Xtrain"]), torch.tensor(fx["ytrain"]), dtype=torch.float64, fix_noise=True)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(0)
    m.reset_parameters()
    after = m.state_dict()
    assert torch.equal(before["likelihood.noise_covar.raw_noise"], after["likelihood.noise_covar.raw_noise"])  # fixed
    for k in ("covar_module.raw_outputscale", "covar_module.base_kernel.raw_lengthscale", "mean_module.constant"):
        assert not torch.equal(before[k], after[k]), k
    assert torch.isfinite(after["covar_module.raw_outputscale"]).all()


def test_argument_validation_and_scope():
    from gpplus_amd.models import GP_Plus, GPR

    X, y = torch.randn(20, 3, dtype=torch.float64), torch.randn(20, dtype=torch.float64)
    with pytest.raises(ValueError):
        GP_Plus(X, y, qual_dict=[0])
    with pytest.raises(ValueError):
        GP_Plus(X, y, quant_correlation_class="Cubic")
    with pytest.raises(NotImplementedError):
        GP_Plus(X, y, embedding_type="probabilistic", qual_dict={})
    with pytest.raises(RuntimeError):
        GPR(X.numpy(), y, "Rough_RBF", [])
    with pytest.raises(RuntimeError):
        GPR(X, y[:5], "Rough_RBF", [])
    g = GPR(X, y, "Rough_RBF", [])
    assert g.covar_module.base_kernel.raw_lengthscale.shape == (1, 3)
    np.testing.assert_allclose(g.covar_module.base_kernel.feature_weights(3).detach().numpy(), np.ones(3), rtol=1e-14)  # w = exp(raw)


def test_no_cpu_fallback():
    from gpplus_amd._lib import GppError
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood

    fx, m = _mixed_model()
    with pytest.raises(GppError, match="no CPU fallback"):
        ExactMarginalLogLikelihood(m.likelihood, m)(m(*m.train_inputs), m.train_targets)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.fit()
    with pytest.raises(GppError):
        m(*m.train_inputs).lazy_covariance_matrix.evaluate()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gp-plus_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|gp_oracle|importlib.*oracle", src, re.M), os.path.join(d, f)


def test_bench_self_launches_its_ranks_dry_run():
    """``python bench.py --gpus 2`` outside torch.distributed.run starts two ranks of itself before touching a GPU and
    relays rank 0's single JSON line (launch path only: gloo rendezvous, barrier, MAX all-reduce; no GPU work)."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True
    # without enough GPUs the real launch refuses loudly instead of running one replica and reporting n_gpus = 1
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0 and "GPU" in p.stderr


def test_bench_watchdog_ends_a_hung_second_leg_on_every_rank():
    """``--dry-run --fake-hang``: rank 1 never returns from the (fake) sharded leg.  After ``--sharded-timeout`` seconds rank 0 prints
    the line with the failure recorded, EVERY rank leaves with a non-zero status, and the self-launcher still returns 0 because
    the line was relayed — within seconds, not after a process-group timeout."""
    import json
    import subprocess
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run", "--fake-hang", "--sharded-timeout", "5"], capture_output=True, text=True, timeout=300, env=env)
    took = time.time() - t0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert "hang" in rec["sharded"]["error"] and rec["n_gpus"] == 2
    assert took < 120, took
    assert "exitcode  : 3" in p.stderr or "exitcode: 3" in p.stderr or "(exitcode: 3)" in p.stderr, p.stderr[-1500:]


def test_panel_timeout_status_is_not_reported_as_indefinite():
    """info = 2**30 is the cooperative panel kernel giving up a wait (gpp_leaf.hip), not a failing leading minor: the host raises
    GppError for it instead of retrying with jitter."""
    from gpplus_amd._lib import GppError
    from gpplus_amd.backend import INFO_PANEL_TIMEOUT, check_status

    check_status(0)
    check_status(251)  # LAPACK-style "leading minor 251": left to the jitter policy
    with pytest.raises(GppError, match="timed out"):
        check_status(INFO_PANEL_TIMEOUT)


def test_panel_hand_off_waits_for_the_write_back_before_raising_a_flag():
    """gpp_panel_potrf_inv publishes data to other work-groups with  fence(release, agent) + flag atomic.  hipcc 7.2 lowered that to
    ``buffer_wbl2 sc1`` followed DIRECTLY by the ``global_atomic_add`` (no ``s_waitcnt vmcnt(0)`` in between), so a flag could become
    visible before the data: one wrong factor in ~50 000 launches under memory pressure.  The source now carries an explicit wait;
    this test reads the ISA hipcc produces for gfx950 and fails if any L2 write-back is again followed by an atomic without the
    wait (whatever future compiler or edit causes it)."""
    import re
    import shutil
    import subprocess
    import tempfile

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    # gpp_leaf.hip: panel_publish / panel_leave; gpp_gemm.hip: the DAG executor's counter increments (gpp_dag_f64) and the one-wave
    # signal kernel of the panel stream
    for name in ("gpp_leaf.hip", "gpp_gemm.hip"):
        src = os.path.join(ROOT, "gp-plus_amd", "csrc", name)
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            p = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                               capture_output=True, text=True, timeout=900)
            assert p.returncode == 0, p.stderr[-2000:]
            lines = [l.strip() for l in open(out) if l.strip() and not l.strip().startswith((";", ".", "//"))]
        instr = [l for l in lines if re.match(r"^[a-z_0-9]+(\s|$)", l)]
        wb = [i for i, l in enumerate(instr) if l.startswith("buffer_wbl2")]
        assert len(wb) >= 2, f"{name}: L2 write-backs missing: has a kernel lost its release fences?"
        for i in wb:
            for l in instr[i + 1:i + 40]:
                if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
                    break
                assert not l.startswith(("global_atomic", "flat_atomic", "buffer_atomic")), \
                    f"{name}: an atomic follows buffer_wbl2 without s_waitcnt vmcnt(0): " + " | ".join(instr[i:i + 12])
            else:
                raise AssertionError(f"{name}: no s_waitcnt vmcnt(0) within 40 instructions of a buffer_wbl2")


# ---- the DAG executor's planner (gpp_dag.hip): host logic, no GPU ---------------------------------------------------------------------
def _dag_check():
    import ctypes

    from gpplus_amd import _lib

    lib = _lib.load()
    f = lib.gpp_debug_dag_check
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                  ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    return f, (ctypes.c_int64 * 4)()


@pytest.mark.parametrize("N,nb,flags,chain_tile,W,fill", [
    (4096, 1024, 1, 64, 448, 64), (7168, 1024, 1, 64, 448, 64), (10000, 1024, 1, 64, 448, 64), (10000, 1024, 0, 128, 448, 0),
    (9300, 1024, 1, 64, 30, 6),      # a last block of 84 rows joins its neighbour
    (5000, 512, 1, 64, 1, 0),        # ONE worker executes the list in order: the order itself must be topological
    (11264, 1024, 1, 128, 7, 2), (13500, 1024, 1, 2064, 448, 64), (15000, 1024, 0, 4064, 448, 64), (20001, 1024, 0, 64, 448, 64),
    (20000, 1024, 1 + (8192 << 2), 4064, 448, 64),   # as at C2: fused groups of 4 steps, the leading 8192 rows of the inverse inside
    (15000, 1024, 1, 4064, 1, 0), (9300, 1024, 1, 2064, 30, 6)])
def test_dag_plan_is_a_valid_schedule_under_any_interleaving(N, nb, flags, chain_tile, W, fill):
    """The ticket list of the DAG executor (gpp_dag.hip: factorisation, and with flags = 1 the right-looking inverse beside it),
    executed on the host by W workers + filler launches in random and adversarial interleavings that respect only what the device
    respects (tickets in list order, counters, stream order of the panel stream, the fillers' ticket limit): every tile gets its
    updates in order and exactly once, solves read fully updated, not yet overwritten block rows of a factored diagonal block,
    updates read completely solved strips, panels start on fully updated blocks, the inverse's sums take their contributions in
    order from finished rows, nothing deadlocks (a filler launch in front of a panel never holds a task that needs that panel),
    everything is complete at the end.  (chain_tile // 1000 = the fusion factor: far tiles take that many steps' updates — and the
    inverse's sums that many contributions — in one task; flags >> 2 = rows of the inverse's leading block built inside the list.)"""
    f, st = _dag_check()
    for seed in range(8):
        rc = f(N, nb, flags, chain_tile, W, fill, seed, st, 0)
        assert rc == 0, (seed, rc)
    assert st[0] > 0 and st[1] > 0 and st[2] > 0


def test_dag_plan_check_notices_a_missing_wait():
    """The test of the test above: with ONE wait removed, and the tasks that raise that counter made slow (they run only when
    nothing else can; for a panel's counter: a lazy panel stream), the host execution must find a violation in most cases.  Some
    waits are implied by the others for every schedule the given number of workers can produce (a worker holds one ticket, so only
    so much of the list is in flight at once): with 4096 workers nearly every removed wait is exposed, with the device's 448 most."""
    f, st = _dag_check()
    for W, need, need_panel in ((448, 0.75, 0.0), (4096, 0.9, 0.5)):
        caught = total = 0
        panel = [0, 0]
        for mut in range(1, 68000, 1511):
            rcs = [f(10000, 1024, 1, 64, W, 64, seed, st, mut) for seed in range(4)]
            total += 1
            caught += any(rcs)
            if int(st[3]) % 10 == 0:  # the removed wait was for a panel (PD)
                panel[0] += 1
                panel[1] += any(rcs)
        assert caught >= need * total, (W, caught, total)
        assert panel[0] > 0 and panel[1] >= need_panel * panel[0], (W, panel)


def _shard_check():
    import ctypes

    from gpplus_amd import _lib

    lib = _lib.load()
    f = lib.gpp_debug_shard_check
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                  ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    return f, (ctypes.c_int64 * 4)()


@pytest.mark.parametrize("N,nb,P,chain_tile,W,fill", [
    (9000, 1024, 1, 64, 448, 64), (9000, 512, 2, 4064, 448, 0), (10000, 1024, 3, 64, 144, 0), (20000, 1024, 2, 4064, 448, 64),
    (20000, 1024, 8, 4064, 448, 64),   # as the C2 test case on 8 ranks: fused groups of 4 steps, fillers between the owned panels
    (6100, 384, 4, 2064, 3, 2),        # P does not divide the block count (16 blocks), three workers per rank
    (4200, 384, 3, 128, 1, 1),         # ONE worker per rank executes its list in order
    (5000, 512, 12, 64, 5, 1),         # more ranks than blocks: planned for the ranks that own one (the others run the launches)
    (30000, 1024, 5, 4064, 448, 64),
    (60000, 1024, 8, 4064, 448, 64)])  # C5 on 8 GPUs: 59 blocks, 1.8 million tasks over the ranks
def test_sharded_lists_are_valid_schedules_under_any_interleaving(N, nb, P, chain_tile, W, fill):
    """The per-rank ticket lists of the sharded evaluation (gpp_dag.hip, DAG_SHARD: each rank's panels, row solves, its share of the
    trailing updates and its column blocks of L^-1), ALL ranks executed together on the host — W workers + filler launches per rank,
    each rank's panel stream, and the communication streams as gp-plus_amd/sharded.py drives them (head then tail of every block
    row, the owner behind its gates, a receiver's signal behind the owner's send) — in random and adversarial interleavings.  Checked
    from the tasks' geometry alone: a task reads another rank's block row only after its message arrived, its own only after the
    copy into place (or, on the chain, the complete solve in the scratch row it still holds), scratch rows are reused only when
    copied out and no longer read, updates and the inverse's sums are applied in order and exactly once on tiles the rank owns, a
    message is packed only from complete strips, nothing deadlocks, everything is complete on every rank at the end."""
    f, st = _shard_check()
    if P > -(-N // nb):
        assert f(N, nb, P, chain_tile, W, fill, 0, st, 0) == 1  # (no plan for a rank without blocks: the caller's launch path)
        return
    by_fillers = 0
    for seed in range(6 if N <= 40000 else 1):
        rc = f(N, nb, P, chain_tile, W, fill, seed, st, 0)
        assert rc == 0, (seed, rc)
        by_fillers = max(by_fillers, int(st[3]))
    assert st[0] > 0 and st[1] > 0 and st[2] > 0
    assert (by_fillers > 0) == (fill > 0)


@pytest.mark.parametrize("piece_cols", [0, 1024, 2048])
def test_sharded_lists_with_other_tail_piece_sizes(piece_cols):
    """The tail of a block row travels in pieces of gpp_shard_piece_cols() columns (gpp.h; round 6): the joint check with ONE piece
    (round 5's messages), pieces of one block's width and of two — in a child process, the size is read once per process.  Also
    a mutation run: a task that loses its wait for a PIECE reads columns that have not arrived."""
    import subprocess

    code = (
        "import ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "from gpplus_amd import _lib\n"
        "lib = _lib.load()\n"
        "assert lib.gpp_shard_piece_cols() == %d\n"
        "f = lib.gpp_debug_shard_check\n"
        "f.restype = ctypes.c_int\n"
        "f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]\n"
        "st = (ctypes.c_int64 * 4)()\n"
        "for N, nb, P, ct, W, fill in ((9000, 512, 3, 4064, 448, 0), (13000, 1024, 4, 2064, 448, 64), (6100, 384, 2, 64, 3, 2)):\n"
        "    for seed in range(3):\n"
        "        rc = f(N, nb, P, ct, W, fill, seed, st, 0)\n"
        "        assert rc == 0, (N, nb, P, seed, rc)\n"
        "f(9000, 512, 3, 4064, 448, 0, 0, st, 0)\n"
        "nw = int(st[1]); caught = total = piece = 0\n"
        "for mut in range(1, nw + 1, max(1, nw // 150)):\n"
        "    rc = f(9000, 512, 3, 4064, 448, 0, mut, st, mut)\n"
        "    total += 1; caught += rc != 0\n"
        "    piece += (int(st[3]) %% 10 == 9)\n"
        "print('CAUGHT', caught, total, piece)\n" % (ROOT, piece_cols))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GPP_SHARD_PIECE_COLS=str(piece_cols)))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    caught, total, piece = (int(v) for v in p.stdout.split("CAUGHT", 1)[1].split()[:3])
    assert caught >= 0.75 * total and piece > 0, (caught, total, piece)  # (some waits are implied by others: 79-90 %)


def test_sharded_list_check_notices_a_missing_wait():
    """With ONE wait removed from one rank's list — and what raises that counter made slow: its tasks, the panel, or the arrival of
    the message — the joint execution must find a violation in most cases (some waits are implied by the others: the scratch row's
    reuse behind the copy, the own panel behind the strip's copy)."""
    f, st = _shard_check()
    for P, W, need in ((3, 448, 0.8), (1, 448, 0.8), (2, 1024, 0.8)):
        f(9000, 512, P, 4064, W, 0, 0, st, 0)
        nwaits = int(st[1])
        caught = total = 0
        remote = [0, 0]
        for mut in range(1, nwaits, 1613):
            rcs = [f(9000, 512, P, 4064, W, 0, seed, st, mut) for seed in range(4)]
            total += 1
            caught += any(rcs)
            if int(st[3]) % 10 == 9:  # the removed wait was for a tail message (ART)
                remote[0] += 1
                remote[1] += any(rcs)
        assert caught >= need * total, (P, W, caught, total)
        assert P == 1 or (remote[0] > 0 and remote[1] >= 0.7 * remote[0]), (P, remote)


def _shard_back_check():
    import ctypes

    from gpplus_amd import _lib

    lib = _lib.load()
    f = lib.gpp_debug_shard_back_check
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                  ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    return f, (ctypes.c_int64 * 4)()


@pytest.mark.parametrize("N,nb,P,fuse,W", [(3400, 512, 1, 1, 7), (9000, 512, 3, 4, 448), (6100, 384, 4, 2, 3), (4200, 384, 3, 1, 1),
                                           (20000, 1024, 1, 4, 512), (20000, 1024, 8, 4, 512), (30000, 1024, 5, 4, 512)])
def test_sharded_back_substitution_list_is_a_valid_schedule(N, nb, P, fuse, W):
    """The back-substitution's ticket list of every rank (gpp_dag.hip, DAG_BACK), executed on the host by W workers in random and
    adversarial interleavings: a row of Z is built from rows of Y that have taken every update, an update reads complete rows of Z
    and is applied in order and exactly once (fused: f steps at once, only on tiles no step of the group reads), only at and below
    the diagonal of column blocks the rank owns; nothing deadlocks; every owned tile of Ky^-1 is complete at the end."""
    f, st = _shard_back_check()
    for rank in range(P):
        for seed in range(4):
            rc = f(N, nb, P, rank, fuse, W, seed, st, 0)
            assert rc == 0, (rank, seed, rc)
        assert st[0] > 0 and st[1] > 0 and st[2] > 0


def test_sharded_back_list_check_notices_a_missing_wait():
    f, st = _shard_back_check()
    for P, rank, fuse in ((1, 0, 4), (3, 1, 2)):
        f(9000, 512, P, rank, fuse, 448, 0, st, 0)
        nwaits = int(st[1])
        caught = total = 0
        for mut in range(1, nwaits, 211 if P == 1 else 67):
            total += 1
            caught += any(f(9000, 512, P, rank, fuse, 448, seed, st, mut) for seed in range(4))
        assert total > 20 and caught >= 0.9 * total, (P, caught, total)


def test_shard_eval_buffer_sizes_and_argument_checks():
    """Host-only parts of the C driver of the sharded evaluation (gpp_shard.hip): the buffer sizes a caller allocates, and that bad
    arguments are refused before anything touches a device."""
    import ctypes

    from gpplus_amd import _lib

    lib = _lib.load()
    N, nb = 60000, 1024
    nblk = -(-N // nb)
    assert lib.gpp_shard_buffer_doubles(N, nb, 0, 8, 0) == N * N + 128        # A: ld = N (a multiple of 16) + the read slack
    assert lib.gpp_shard_buffer_doubles(10001, nb, 0, 8, 0) == 10001 * 10016 + 128   # ld rounded up to 16
    for rank in range(8):
        owned = len(range(rank, nblk, 8))
        assert lib.gpp_shard_buffer_doubles(N, nb, rank, 8, 1) == N * owned * nb + 128   # Kc / Lc: the owned column blocks, whole
    assert lib.gpp_shard_buffer_doubles(N, nb, 3, 8, 2) == nblk * nb * nb
    assert lib.gpp_shard_buffer_doubles(N, nb, 3, 8, 3) == nb * N + 128
    assert lib.gpp_shard_buffer_doubles(N, nb, 3, 8, 4) == nb * (N + 2 * nb)
    assert lib.gpp_shard_buffer_doubles(N, nb, 3, 8, 9) == 0
    assert lib.gpp_set_comm(None, None, 0, 1) == -1 and lib.gpp_comm_init_rccl(None, None, 0, 1) == -1
    info = ctypes.c_int(0)
    assert lib.gpp_shard_eval(None, N, nb, None, 8, None, None, None, None, 1, 0, 0, 0.0, 0, 1, None, ctypes.byref(info)) == -1


def test_reference_fp32_theta_round_trip_is_a_switch():
    """optim/mll_scipy.py:32-35,97: the reference loads float32(theta) into the model whatever its dtype.  Default here: theta keeps
    the model's dtype; ``settings.reference_fp32_theta(True)`` reproduces the round trip — the two modes differ by that rounding only."""
    from gpplus_amd import settings
    from gpplus_amd.optim import MLLObjective

    _, m = _mixed_model()
    obj = MLLObjective(m, True, [0, 0])
    x = obj.pack_parameters() + 1e-9
    assert not np.array_equal(x, x.astype(np.float32).astype(np.float64))
    flat = lambda d: np.concatenate([v.detach().double().numpy().ravel() for v in d.values()])  # noqa: E731
    np.testing.assert_array_equal(flat(obj.unpack_parameters(x)), x)
    with settings.reference_fp32_theta(True):
        np.testing.assert_array_equal(flat(obj.unpack_parameters(x)), x.astype(np.float32).astype(np.float64))
        assert all(v.dtype == torch.float64 for v in obj.unpack_parameters(x).values())  # (cast back to the model's dtype on load)
    np.testing.assert_array_equal(flat(obj.unpack_parameters(x)), x)


def test_push_transport_describes_its_parts_in_bytes():
    """gp-plus_amd/push.py::_arrays (host logic of the GPP_SHARD_PUSH transport): a part is a strided 2-D view + its place in the message;
    what reaches gpp_push_send / gpp_push_recv are pointers, pitches, offsets, widths in BYTES and heights in rows; views the 2-D copies
    cannot express are refused; the transport is off unless asked for, and never on in a group of one."""
    import torch

    from gpplus_amd import push

    A = torch.zeros(64, 100, dtype=torch.float64)
    D = torch.zeros(3, 16, 16, dtype=torch.float64)
    nbk, wh = 16, 40
    n, ptr, vp, off, sp, wd, ht = push._arrays([(A[16:32, 10:50], 0, wh), (D[1, :nbk, :nbk], nbk * wh, nbk), (A[5:6, 0:8], 7, 8)])
    assert n == 3
    assert [int(p) for p in ptr] == [A[16:32, 10:50].data_ptr(), D[1].data_ptr(), A[5:6].data_ptr()]
    assert list(vp) == [100 * 8, 16 * 8, 8 * 8]  # (one row: its width stands for the pitch)
    assert list(sp) == [wh * 8, nbk * 8, 8 * 8]
    assert list(off) == [0, nbk * wh * 8, 7 * 8] and list(wd) == [wh * 8, nbk * 8, 8 * 8] and list(ht) == [16, 16, 1]
    with pytest.raises(ValueError):
        push._arrays([(A[:, ::2], 0, 50)])          # column stride 2
    with pytest.raises(ValueError):
        push._arrays([(A[0], 0, 100)])              # not 2-D
    with pytest.raises(ValueError):
        push._arrays([(A.float(), 0, 100)])         # not float64
    assert not push.active(1) and (push.active(4) == push.ENABLED)
