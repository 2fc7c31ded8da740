"""GPU parity of the individual HIP kernels (called through the C ABI) against numpy/scipy fp64."""
import numpy as np
import pytest
import scipy.linalg as sla
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def _sq(n, fill=None):
    from gpplus_amd.backend import square_buffer

    m = square_buffer(n, "cuda")
    m.fill_(float("nan") if fill is None else fill)
    return m


def _spd(n, d=6, seed=0, noise=1e-3):
    rng = np.random.default_rng(seed)
    U = rng.standard_normal((n, d))
    w = rng.uniform(0.05, 0.5, d)
    d2 = ((U[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    K = 0.8 * np.exp(-d2) + noise * np.eye(n)
    return U, w, K


@pytest.mark.parametrize("variant", ["NT", "NN", "TN"])
@pytest.mark.parametrize("shape", [(128, 128, 16), (200, 333, 77), (513, 130, 1000), (64, 700, 5)])
def test_gemm_plain(gpu_ctx, variant, shape):
    M, N, K = shape
    rng = np.random.default_rng(1)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    ld = lambda n: (n + 15) // 16 * 16
    tA, tB = {"NT": (0, 1), "NN": (0, 0), "TN": (1, 0)}[variant]
    As = A.T if tA else A
    Bs = B.T if tB else B
    dA = torch.zeros(As.shape[0], ld(As.shape[1]), dtype=torch.float64, device="cuda")
    dA[:, : As.shape[1]] = _dev(As)
    dB = torch.zeros(Bs.shape[0], ld(Bs.shape[1]), dtype=torch.float64, device="cuda")
    dB[:, : Bs.shape[1]] = _dev(Bs)
    dC = torch.zeros(M, ld(N), dtype=torch.float64, device="cuda")
    dC[:, :N] = _dev(C0)
    gpu_ctx.gemm(tA, tB, M, N, K, 0.7, dA[:, : As.shape[1]], dB[:, : Bs.shape[1]], -1.3, dC[:, :N])
    ref = 0.7 * A @ B - 1.3 * C0
    np.testing.assert_allclose(dC[:, :N].cpu().numpy(), ref, rtol=1e-12, atol=1e-11)


def test_gemm_masks_and_lower(gpu_ctx):
    n = 300
    rng = np.random.default_rng(2)
    Lo = np.tril(rng.standard_normal((n, n)))
    junk = rng.standard_normal((n, n)) * 1e6
    Lj = Lo + np.triu(junk, 1)  # garbage above the diagonal must be ignored
    X = rng.standard_normal((n, n))
    dL, dX = _sq(n), _sq(n)
    dL.copy_(_dev(Lj)); dX.copy_(_dev(X))
    # NT, B lower [n][k] keep k<=n: X @ Lo^T with khi limited per column tile
    out = _sq(n)
    gpu_ctx.gemm(0, 1, n, n, n, 1.0, dX, dL, 0.0, out, b_mask=1, khi_mode=2)
    np.testing.assert_allclose(out.cpu().numpy(), X @ Lo.T, rtol=1e-12, atol=1e-10)
    # NN, B lower [k][n] keep k>=n: X @ Lo
    gpu_ctx.gemm(0, 0, n, n, n, 1.0, dX, dL, 0.0, out, b_mask=2, klo_mode=2)
    np.testing.assert_allclose(out.cpu().numpy(), X @ Lo, rtol=1e-12, atol=1e-10)
    # NN, A lower [m][k] keep k<=m: Lo @ X
    gpu_ctx.gemm(0, 0, n, n, n, 1.0, dL, dX, 0.0, out, a_mask=1, khi_mode=1)
    np.testing.assert_allclose(out.cpu().numpy(), Lo @ X, rtol=1e-12, atol=1e-10)
    # TN lauum: Lo^T Lo, lower part only; upper part of `out` must stay untouched
    out.fill_(7.0)
    gpu_ctx.gemm(1, 0, n, n, n, 1.0, dL, dL, 0.0, out, a_mask=2, b_mask=2, klo_mode=3, c_tri=1)
    got = out.cpu().numpy()
    ref = Lo.T @ Lo
    np.testing.assert_allclose(np.tril(got), np.tril(ref), rtol=1e-12, atol=1e-10)
    assert np.all(np.triu(got, 1) == np.triu(np.full((n, n), 7.0), 1))
    # TN syrk-style update of the UPPER triangle only: C -= X^T X
    out.fill_(7.0)
    gpu_ctx.gemm(1, 0, n, n, n, -1.0, dX, dX, 1.0, out, c_tri=2)
    got = out.cpu().numpy()
    np.testing.assert_allclose(np.triu(got), np.triu(7.0 - X.T @ X), rtol=1e-12, atol=1e-10)
    assert np.all(np.tril(got, -1) == np.tril(np.full((n, n), 7.0), -1))


@pytest.mark.parametrize("n,d", [(1, 3), (64, 8), (130, 2), (500, 8), (1000, 12)])
def test_kernel_build_and_cross(gpu_ctx, n, d):
    rng = np.random.default_rng(3)
    U = rng.standard_normal((n, d))
    w = rng.uniform(0.05, 2.0, d)
    sf2 = 0.77
    tau = np.array([1e-3, 2e-2, 0.3])
    grp = rng.integers(0, 3, n).astype(np.int32)
    d2 = ((U[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    ref = sf2 * np.exp(-d2) + np.diag(tau[grp] + 1e-6)
    dU, dw = _dev(U), _dev(w)
    dsf2 = torch.tensor([sf2], dtype=torch.float64, device="cuda")
    out = _sq(n)
    gpu_ctx.kernel_build(dU, dw, dsf2, _dev(tau), _dev(grp), out, jitter=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-13, atol=1e-15)
    out2 = _sq(n, fill=-5.0)
    gpu_ctx.kernel_build(dU, dw, dsf2, _dev(tau), _dev(grp), out2, jitter=1e-6, uplo=1)
    np.testing.assert_allclose(np.tril(out2.cpu().numpy()), np.tril(ref), rtol=1e-13, atol=1e-15)
    out3 = _sq(n, fill=-5.0)
    gpu_ctx.kernel_build(dU, dw, dsf2, _dev(tau), _dev(grp), out3, jitter=1e-6, uplo=2)
    np.testing.assert_allclose(np.triu(out3.cpu().numpy()), np.triu(ref), rtol=1e-13, atol=1e-15)
    # cross block against a different point set
    m = 77
    Ua = rng.standard_normal((m, d))
    cross = torch.empty(m, (n + 15) // 16 * 16, dtype=torch.float64, device="cuda")[:, :n]
    gpu_ctx.cross_kernel(_dev(Ua), dU, dw, dsf2, cross)
    d2c = ((Ua[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    np.testing.assert_allclose(cross.cpu().numpy(), sf2 * np.exp(-d2c), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("n,use_ws", [(1, False), (5, True), (128, False), (129, True), (300, False), (777, True),
                                      (2048, True), (4097, False), (4097, True), (6700, True), (6700, False),
                                      (11500, True)])
def test_potrf_trtri_lauum(gpu_ctx, n, use_ws):
    """use_ws: pass the scratch to the factorisation.  The drivers behind the same entry points: leaf steps on one stream
    (n < 3840, or < 6144 without scratch); look-ahead on internal streams above, which with the scratch inverts its diagonal
    block rows, solves panels with one GEMM and — for 3840 <= n <= 11264 — also builds the whole inverse by bordering, so
    trtri finds nothing left (6700, True); above that trtri skips only the merged levels (11500, True); without the scratch
    the panels are solved recursively and trtri does all the merging (6700, False)."""
    U, w, K = _spd(n, seed=n)
    A, Li, T, Ki = _sq(n), _sq(n), _sq(n), _sq(n)
    A.copy_(_dev(np.triu(K) + np.tril(np.full((n, n), np.nan), -1)))  # the strict lower triangle must never be read
    info = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info, T if use_ws else None)
    assert int(info.item()) == 0
    Lref = np.linalg.cholesky(K)
    Ufac = np.triu(A.cpu().numpy())
    np.testing.assert_allclose(Ufac, Lref.T, rtol=0, atol=1e-10 * np.abs(Lref).max())
    assert np.isnan(np.tril(A.cpu().numpy(), -1)[np.tril_indices(n, -1)]).all()  # ... and never written
    gpu_ctx.trtri(A, Li, T)
    full = Li.cpu().numpy()
    Linv = np.tril(full)
    np.testing.assert_allclose(Linv @ Lref, np.eye(n), rtol=0, atol=1e-8)
    np.testing.assert_array_equal(np.triu(full, 1), np.tril(full, -1).T)  # mirror image L^-T in the upper triangle
    gpu_ctx.lauum(Li, Ki)
    Kinv = np.tril(Ki.cpu().numpy())
    Kinv = Kinv + np.tril(Kinv, -1).T
    np.testing.assert_allclose(Kinv @ K, np.eye(n), rtol=0, atol=1e-6)


def test_potrf_reports_failing_minor(gpu_ctx):
    n = 400
    _, _, K = _spd(n, seed=9)
    K[250, 250] = -1.0  # leading minor 251 is not positive definite
    A, Li = _sq(n), _sq(n)
    A.copy_(_dev(K))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info)
    assert int(info.item()) == 251


@pytest.mark.parametrize("n,d,S,dU", [(300, 5, 1, 0), (777, 8, 3, 2), (1500, 12, 3, 12)])
def test_mll_and_grad_reduce(gpu_ctx, n, d, S, dU):
    rng = np.random.default_rng(n)
    U = rng.standard_normal((n, d))
    w = rng.uniform(0.05, 0.6, d)
    sf2 = 0.9
    tau = rng.uniform(1e-3, 1e-2, S)
    grp = rng.integers(0, S, n).astype(np.int32)
    r = rng.standard_normal(n)
    diff = U[:, None, :] - U[None, :, :]
    d2 = (diff**2 * w).sum(-1)
    Kc = sf2 * np.exp(-d2)
    Ky = Kc + np.diag(tau[grp])
    cf = sla.cho_factor(Ky, lower=True)
    alpha = sla.cho_solve(cf, r)
    quad = r @ alpha
    logdet = 2 * np.log(np.diag(cf[0])).sum()
    mll = -0.5 * (quad + logdet + n * np.log(2 * np.pi))
    Kinv = sla.cho_solve(cf, np.eye(n))
    W = 0.5 * (np.outer(alpha, alpha) - Kinv)
    g_w = np.array([(W * Kc * (-(diff[:, :, k] ** 2))).sum() for k in range(d)])
    g_sf2 = (W * Kc).sum() / sf2
    g_tau = np.array([np.diag(W)[grp == s].sum() for s in range(S)])
    g_U = np.stack([2 * (W * Kc * (-2 * w[k]) * diff[:, :, k]).sum(1) for k in range(dU)], 1) if dU else None

    dUm, dw = _dev(U), _dev(w)
    dsf2 = torch.tensor([sf2], dtype=torch.float64, device="cuda")
    dgrp = _dev(grp)
    A, Li, T, Ki = _sq(n), _sq(n), _sq(n), _sq(n)
    gpu_ctx.kernel_build(dUm, dw, dsf2, _dev(tau), dgrp, A, uplo=2)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info)
    assert int(info.item()) == 0
    gpu_ctx.trtri(A, Li, T)
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    out3 = torch.empty(3, dtype=torch.float64, device="cuda")
    gpu_ctx.mll_reduce(A, Li, _dev(r), z, out3)
    o = out3.cpu().numpy()
    np.testing.assert_allclose(o, [quad, logdet, mll], rtol=1e-9)
    al = torch.empty(n, dtype=torch.float64, device="cuda")
    gpu_ctx.alpha(Li, z, al)
    np.testing.assert_allclose(al.cpu().numpy(), alpha, rtol=1e-7, atol=1e-9 * np.abs(alpha).max())
    gpu_ctx.lauum(Li, Ki)
    gw = torch.empty(d, dtype=torch.float64, device="cuda")
    gs = torch.empty(1, dtype=torch.float64, device="cuda")
    gt = torch.empty(S, dtype=torch.float64, device="cuda")
    gU = torch.empty(n, dU, dtype=torch.float64, device="cuda") if dU else None
    gpu_ctx.grad_reduce(dUm, dw, dsf2, dgrp, S, al, Ki, dU, gw, gs, gt, gU)
    sc = lambda x: 1e-7 * np.abs(x).max() + 1e-12
    np.testing.assert_allclose(gw.cpu().numpy(), g_w, rtol=1e-6, atol=sc(g_w))
    np.testing.assert_allclose(gs.cpu().numpy()[0], g_sf2, rtol=1e-6, atol=sc(g_sf2))
    np.testing.assert_allclose(gt.cpu().numpy(), g_tau, rtol=1e-6, atol=sc(g_tau))
    if dU:
        np.testing.assert_allclose(gU.cpu().numpy(), g_U, rtol=1e-6, atol=sc(g_U))


def test_predict(gpu_ctx):
    n, m, d = 600, 130, 7
    rng = np.random.default_rng(5)
    U, w, K = _spd(n, d=d, seed=4)
    Us = rng.standard_normal((m, d))
    r = rng.standard_normal(n)
    d2c = ((Us[:, None, :] - U[None, :, :]) ** 2 * w).sum(-1)
    Ks = 0.8 * np.exp(-d2c)
    cf = sla.cho_factor(K, lower=True)
    alpha = sla.cho_solve(cf, r)
    mean = Ks @ alpha
    V = sla.solve_triangular(cf[0], Ks.T, lower=True)
    var = 0.8 - (V**2).sum(0)
    A, Li, T = _sq(n), _sq(n), _sq(n)
    A.copy_(_dev(K))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info)
    gpu_ctx.trtri(A, Li, T)
    ldn = (n + 15) // 16 * 16
    dKs = torch.empty(m, ldn, dtype=torch.float64, device="cuda")[:, :n]
    dKs.copy_(_dev(Ks))
    dV = torch.empty(m, ldn, dtype=torch.float64, device="cuda")[:, :n]
    mo = torch.empty(m, dtype=torch.float64, device="cuda")
    vo = torch.empty(m, dtype=torch.float64, device="cuda")
    kss = torch.full((m,), 0.8, dtype=torch.float64, device="cuda")
    gpu_ctx.predict(Li, _dev(alpha), dKs, kss, dV, mo, vo)
    np.testing.assert_allclose(mo.cpu().numpy(), mean, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(vo.cpu().numpy(), var, rtol=1e-6, atol=1e-9)


# ---- entry points of the sharded evaluation (gp-plus_amd/sharded.py), one rank at a time ------------------------------
def test_gemm_batched(gpu_ctx):
    """gpp_gemm_batched: products at a regular column spacing with a shared A (the layout of the sharded inverse)."""
    rng = np.random.default_rng(3)
    K, M, nb, P, nbatch = 96, 200, 128, 3, 4
    A = rng.standard_normal((K, M))                      # stored K x M (transA = 1)
    Bfull = rng.standard_normal((K, P * nb * nbatch))    # the batch elements are the column blocks b * P * nb .. + nb
    C0 = rng.standard_normal((M, P * nb * nbatch))
    dA, dB, dC = _dev(A), _dev(Bfull), _dev(C0)
    gpu_ctx.gemm_batched(1, 0, M, nb, K, 0.5, dA, 0, dB[:, :nb], P * nb, 2.0, dC[:, :nb], P * nb, nbatch)
    ref = C0.copy()
    for b in range(nbatch):
        c = slice(b * P * nb, b * P * nb + nb)
        ref[:, c] = 2.0 * C0[:, c] + 0.5 * A.T @ Bfull[:, c]
    np.testing.assert_allclose(dC.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_syrk_rows_covers_the_upper_triangle_once(gpu_ctx, nranks):
    """gpp_syrk_rows: the ranks' launches together apply C(upper) -= U^T U exactly once; each touches only its block rows."""
    rng = np.random.default_rng(4)
    K, Nt, nb, first = 160, 1100, 256, 5
    U = rng.standard_normal((K, Nt))
    C0 = rng.standard_normal((Nt, Nt))
    dU = _dev(U)
    C = _sq(Nt, 0.0)
    C.copy_(_dev(C0))
    full = C0 - U.T @ U
    for r in range(nranks):
        before = C.cpu().numpy().copy()
        gpu_ctx.syrk_rows(dU, C, nb, first, r, nranks)
        after = C.cpu().numpy()
        for i0 in range(0, Nt, nb):
            rows = slice(i0, min(i0 + nb, Nt))
            mine = (first + i0 // nb) % nranks == r
            blk_ref = np.triu(full)[rows] if mine else None
            if mine:
                np.testing.assert_allclose(np.triu(after)[rows][:, i0:], blk_ref[:, i0:], rtol=1e-11, atol=1e-11)
            else:
                np.testing.assert_array_equal(after[rows], before[rows])
    np.testing.assert_allclose(np.triu(C.cpu().numpy()), np.triu(full), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("nranks", [2, 5])
def test_lauum_rows_and_grad_reduce_rows(gpu_ctx, nranks):
    """gpp_lauum_rows / gpp_grad_reduce_rows: the ranks' tile-row shares tile Kinv exactly, and their partial gradient
    sums add up to gpp_grad_reduce."""
    n, d, S, dU = 900, 6, 2, 2
    rng = np.random.default_rng(8)
    U = rng.standard_normal((n, d))
    w = rng.uniform(0.05, 0.6, d)
    tau = rng.uniform(1e-3, 1e-2, S)
    grp = rng.integers(0, S, n).astype(np.int32)
    dUm, dw, dgrp = _dev(U), _dev(w), _dev(grp)
    dsf2 = torch.tensor([0.9], dtype=torch.float64, device="cuda")
    A, Li, T, Ki = _sq(n), _sq(n), _sq(n), _sq(n)
    gpu_ctx.kernel_build(dUm, dw, dsf2, _dev(tau), dgrp, A, uplo=2)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info)
    gpu_ctx.trtri(A, Li, T)
    gpu_ctx.lauum(Li, Ki)
    Kref = np.tril(Ki.cpu().numpy())
    al = _dev(rng.standard_normal(n))
    ref = [torch.empty(d, dtype=torch.float64, device="cuda"), torch.empty(1, dtype=torch.float64, device="cuda"),
           torch.empty(S, dtype=torch.float64, device="cuda"), torch.empty(n, dU, dtype=torch.float64, device="cuda")]
    gpu_ctx.grad_reduce(dUm, dw, dsf2, dgrp, S, al, Ki, dU, *ref)
    acc = [torch.zeros_like(t) for t in ref]
    Kshare = _sq(n)
    covered = np.zeros(n, dtype=int)
    for r in range(nranks):
        Kshare.fill_(float("nan"))
        gpu_ctx.lauum_rows(Li, Kshare, r, nranks)
        got = Kshare.cpu().numpy()
        for t0 in range(0, n, 128):
            rows = slice(t0, min(t0 + 128, n))
            if (t0 // 128) % nranks == r:
                covered[rows] += 1
                np.testing.assert_allclose(np.tril(got)[rows], Kref[rows], rtol=1e-12, atol=1e-12)
            else:
                assert np.isnan(got[rows]).all()       # another rank's tile rows are not touched
        part = [torch.empty_like(t) for t in ref]
        gpu_ctx.grad_reduce_rows(dUm, dw, dsf2, dgrp, S, al, Kshare, dU, 128, r, nranks, *part)
        for a, p_ in zip(acc, part):
            a += p_
    assert (covered == 1).all()
    for a, t in zip(acc, ref):
        np.testing.assert_allclose(a.cpu().numpy(), t.cpu().numpy(), rtol=1e-10, atol=1e-10 * float(t.abs().max()))


@pytest.mark.parametrize("n,d,S,dU,shared", [(300, 5, 1, 0, True), (777, 8, 3, 2, False), (130, 4, 2, 4, False)])
def test_batched_evaluation_matches_single(gpu_ctx, n, d, S, dU, shared):
    """The *_batched entry points (B independent problems per launch) against B runs of the single-problem ones."""
    B = 5
    rng = np.random.default_rng(n + 1)
    Ub = rng.standard_normal((1 if shared else B, n, d))
    w = rng.uniform(0.05, 0.6, (B, d))
    sf2 = rng.uniform(0.5, 1.5, B)
    tau = rng.uniform(1e-3, 1e-2, (B, S))
    grp = rng.integers(0, S, n).astype(np.int32)
    r = rng.standard_normal((B, n))
    dU_, dw, ds, dt, dg = _dev(Ub[0] if shared else Ub), _dev(w), _dev(sf2), _dev(tau), _dev(grp)
    A, Li, Ki = (gpu_ctx.batched_buffer(B, n) for _ in range(3))
    info = torch.zeros(B, dtype=torch.int32, device="cuda")
    dr, z, al = (gpu_ctx.batched_vector(B, n) for _ in range(3))   # odd n: rows padded to an even stride
    dr.copy_(_dev(r))
    out3 = torch.empty(B, 3, dtype=torch.float64, device="cuda")
    gw = torch.empty(B, d, dtype=torch.float64, device="cuda"); gs = torch.empty(B, dtype=torch.float64, device="cuda")
    gt = torch.empty(B, S, dtype=torch.float64, device="cuda")
    gU = torch.empty(B, n, dU, dtype=torch.float64, device="cuda") if dU else None
    gpu_ctx.kernel_build_batched(dU_, dw, ds, dt, dg, A, uplo=2)
    gpu_ctx.potrf_batched(A, Li, info)
    assert int(info.abs().max().item()) == 0
    gpu_ctx.trtri_batched(A, Li, Ki)
    gpu_ctx.mll_reduce_batched(A, Li, dr, z, out3)
    gpu_ctx.alpha_batched(Li, z, al)
    gpu_ctx.lauum_batched(Li, Ki)
    gpu_ctx.grad_reduce_batched(dU_, dw, ds, dg, S, al, Ki, dU, gw, gs, gt, gU)
    for b in range(B):
        Us = _dev(Ub[0] if shared else Ub[b])
        A1, L1, T1, K1 = _sq(n), _sq(n), _sq(n), _sq(n)
        i1 = torch.zeros(1, dtype=torch.int32, device="cuda")
        gpu_ctx.kernel_build(Us, dw[b].contiguous(), ds[b:b + 1].contiguous(), dt[b].contiguous(), dg, A1, uplo=2)
        gpu_ctx.potrf(A1, L1, i1)
        gpu_ctx.trtri(A1, L1, T1)
        z1 = torch.empty(n, dtype=torch.float64, device="cuda"); a1 = torch.empty_like(z1)
        o1 = torch.empty(3, dtype=torch.float64, device="cuda")
        gpu_ctx.mll_reduce(A1, L1, dr[b].contiguous(), z1, o1)
        gpu_ctx.alpha(L1, z1, a1)
        gpu_ctx.lauum(L1, K1)
        w1 = torch.empty(d, dtype=torch.float64, device="cuda"); s1 = torch.empty(1, dtype=torch.float64, device="cuda")
        t1 = torch.empty(S, dtype=torch.float64, device="cuda")
        u1 = torch.empty(n, dU, dtype=torch.float64, device="cuda") if dU else None
        gpu_ctx.grad_reduce(Us, dw[b].contiguous(), ds[b:b + 1].contiguous(), dg, S, a1, K1, dU, w1, s1, t1, u1)
        tol = dict(rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(out3[b].cpu().numpy(), o1.cpu().numpy(), **tol)
        np.testing.assert_allclose(al[b].cpu().numpy(), a1.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(a1.abs().max()))
        # (the strict upper triangle of K1 is never written and still holds the NaN fill: scale from the lower triangle.  The batched
        #  path factors in leaf steps, the single one with the cooperative panel: equal to rounding, not bitwise)
        K1l = np.tril(K1.cpu().numpy())
        np.testing.assert_allclose(np.tril(Ki[b].cpu().numpy()), K1l, rtol=1e-9, atol=1e-9 * float(np.abs(K1l).max()))
        np.testing.assert_allclose(gw[b].cpu().numpy(), w1.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(w1.abs().max()))
        np.testing.assert_allclose(gs[b].item(), s1.item(), rtol=1e-9)
        np.testing.assert_allclose(gt[b].cpu().numpy(), t1.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(t1.abs().max()))
        if dU:
            np.testing.assert_allclose(gU[b].cpu().numpy(), u1.cpu().numpy(), rtol=1e-9, atol=1e-9 * float(u1.abs().max()))


def test_batched_mll_function_matches_single(gpu_ctx):
    """batched.BatchedMLLFunction (values and every gradient, per batch element) against linalg.ExactMLLFunction; a
    non-positive-definite element returns NaN with zero gradients and does not disturb the others."""
    from gpplus_amd.batched import batched_mll
    from gpplus_amd.linalg import KernelSpec, exact_mll

    B, n, d, S, dU = 4, 301, 5, 2, 2
    rng = np.random.default_rng(21)
    Ub = torch.tensor(rng.standard_normal((B, n, d)), device="cuda", requires_grad=True)
    w = torch.tensor(rng.uniform(0.05, 0.6, (B, d)), device="cuda", requires_grad=True)
    sf2 = torch.tensor(rng.uniform(0.5, 1.5, B), device="cuda", requires_grad=True)
    tau0 = rng.uniform(1e-3, 1e-2, (B, S)); tau0[2] = -5.0          # element 2: negative "noise" -> not p.d.
    tau = torch.tensor(tau0, device="cuda", requires_grad=True)
    mean = torch.tensor(rng.standard_normal((B, n)) * 0.1, device="cuda", requires_grad=True)
    y = torch.tensor(rng.standard_normal(n), device="cuda")
    grp = torch.tensor(rng.integers(0, S, n).astype(np.int32), device="cuda")
    mll = batched_mll(Ub, w, sf2, tau, mean, y, grp, n_grad_dims=dU)
    assert torch.isnan(mll[2]) and torch.isfinite(mll[[0, 1, 3]]).all()
    coef = torch.tensor([1.0, -2.0, 3.0, 0.5], device="cuda", dtype=torch.float64)
    torch.nansum(mll * coef).backward()
    for b in (0, 1, 3):
        Us = Ub[b].detach().clone().requires_grad_(True)
        ws_, ss, ts, ms = (t[b].detach().clone().requires_grad_(True) for t in (w, sf2, tau, mean))
        one = exact_mll(Us, KernelSpec(ws_, ss), ts, ms, y, grp=grp, n_grad_dims=dU)
        (one * coef[b]).backward()
        assert abs(one.item() - mll[b].item()) <= 1e-10 * abs(one.item())
        for name, gb, g1 in (("U", Ub.grad[b], Us.grad), ("w", w.grad[b], ws_.grad), ("sf2", sf2.grad[b], ss.grad),
                             ("tau", tau.grad[b], ts.grad), ("mean", mean.grad[b], ms.grad)):
            np.testing.assert_allclose(gb.cpu().numpy(), g1.cpu().numpy(), rtol=1e-8, atol=1e-9 * float(g1.abs().max()), err_msg=name)
    for g in (Ub.grad[2], w.grad[2], sf2.grad[2], tau.grad[2], mean.grad[2]):
        assert float(g.abs().max()) == 0.0


@pytest.mark.parametrize("n", [4224, 6700, 9984, 12000, 13500, 16384])
def test_lookahead_driver_is_bitwise_repeatable(gpu_ctx, n):
    """Race screen for the look-ahead factorisation's internal streams (panel, throughput, fill, unmasked): every tile
    product accumulates in a fixed order, so repeating the same factorisation + inverse + Ky^-1 must reproduce the
    first result bit for bit; a missing event between two streams — or a missing counter wait in the DAG executor's ticket list —
    shows up as a difference.  4224 and 6700 take the bordered-inverse path of the launches, 9984 and above the ticket list
    (gpp_dag_f64: any work-group may run any tile, the order of a tile's updates is enforced by its version counter; 12000 and
    13500 with fused pairs of steps, 13500 ragged, 16384 a multiple of the block height with groups of four)."""
    U, w, K = _spd(n, seed=n)
    Kd = _dev(np.triu(K))
    A, Li, Ki = _sq(n), _sq(n), _sq(n)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ref = None
    for rep in range(5):
        A.copy_(Kd); Li.zero_(); Ki.zero_()
        gpu_ctx.potrf(A, Li, info, Ki)
        gpu_ctx.trtri(A, Li, Ki)
        fac, inv = torch.triu(A).clone(), Li.clone()
        gpu_ctx.lauum(Li, Ki)
        cur = (fac, inv, torch.tril(Ki).clone())
        assert int(info.item()) == 0
        if ref is None:
            ref = cur
        else:
            for a, b in zip(ref, cur):
                assert torch.equal(a, b), f"repetition {rep} differs by {float((a - b).abs().max()):.3e}"


@pytest.mark.gpu
def test_transpose_and_lauum_row_ranges(gpu_ctx):
    """``gpp_transpose`` (the sharded inverse's mirror) on ragged shapes and padded views, and ``gpp_lauum_rows_range``: the
    ranges of block rows of every rank's cyclic share add up to the full lower triangle of Linv^T Linv."""
    g = torch.Generator(device="cuda").manual_seed(5)
    for rows, cols in ((1, 1), (63, 65), (130, 70), (1000, 257)):
        src = torch.randn(rows + 3, cols + 5, dtype=torch.float64, device="cuda", generator=g)[2:2 + rows, 1:1 + cols]
        dst = torch.full((cols + 2, rows + 4), -7.0, dtype=torch.float64, device="cuda")
        gpu_ctx.transpose(src, dst[1:1 + cols, 3:3 + rows])
        assert torch.equal(dst[1:1 + cols, 3:3 + rows], src.t())
        assert float(dst[0].abs().min()) == 7.0 and float(dst[:, :3].abs().min()) == 7.0  # nothing outside the view
    n = 1100
    Li = _sq(n)
    Li.copy_(torch.tril(torch.randn(n, n, dtype=torch.float64, device="cuda", generator=g)))
    ref = torch.tril(Li.T @ Li)
    for nranks in (1, 3):
        Ki = _sq(n, fill=0.0)
        bounds = [0, 256, 640, 1024, n]
        for rank in range(nranks):
            for r0, r1 in zip(bounds[:-1], bounds[1:]):
                gpu_ctx.lauum_rows_range(Li, Ki, rank, nranks, r0, r1)
        np.testing.assert_allclose(torch.tril(Ki).cpu().numpy(), ref.cpu().numpy(), rtol=1e-11, atol=1e-9)
