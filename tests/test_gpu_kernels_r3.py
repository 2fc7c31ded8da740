"""Round-3 kernel parity through the C ABI against numpy fp64:
  * wide feature counts (D = 17, 33, 49, 64: the DT = 32 / 64 instantiations of gpp_grad_tiles, the > 48 KiB LDS opt-in of
    gpp_cov_tile) for the RBF and both Matern kinds, with and without feature gradients;
  * the column-sharded pieces of the back-substituted sharded evaluation: gpp_gemm_lower_cols, gpp_trmv_lower_cols,
    gpp_mll_scalars, gpp_grad_reduce_cols;
  * argument checks that guard the kernels' 16-byte accesses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), device="cuda")


def _sq(n, fill=None):
    from gpplus_amd.backend import square_buffer

    m = square_buffer(n, "cuda")
    m.fill_(float("nan") if fill is None else fill)
    return m


def _kernel_and_derivs(Ua, Ub, w, sf2, kind, d_split):
    """K, dK/d(-r2_rbf), dK/d(-r2_mat) of sf2 * exp(-r2_rbf) * m(r2_mat) with r2 = sum_d w_d (ua_d - ub_d)^2 over the RBF dims
    (d < d_split, or all of them for kind 0) and the Matern dims (kernels/matern.py:4-8: nu = 3/2, 5/2 with l_d = (2 w_d)^-1/2)."""
    diff = Ua[:, None, :] - Ub[None, :, :]
    dsp = Ua.shape[1] if kind == 0 else d_split
    r2r = (diff[:, :, :dsp] ** 2 * w[:dsp]).sum(-1)
    r2m = (diff[:, :, dsp:] ** 2 * w[dsp:]).sum(-1)
    er = np.exp(-r2r)
    if kind == 0:
        m, dm = np.ones_like(r2m), np.zeros_like(r2m)
    elif kind == 1:
        a = np.sqrt(6.0 * r2m)
        m, dm = (1 + a) * np.exp(-a), 3.0 * np.exp(-a)
    else:
        a = np.sqrt(10.0 * r2m)
        m, dm = (1 + a + a * a / 3.0) * np.exp(-a), (5.0 / 3.0) * (1 + a) * np.exp(-a)
    return sf2 * er * m, sf2 * er * m, sf2 * er * dm, diff, dsp


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("d,dU", [(17, 0), (17, 3), (33, 0), (33, 5), (49, 2), (64, 0), (64, 4)])
def test_wide_feature_counts(gpu_ctx, d, dU, kind):
    n, m, S = 200, 70, 2
    rng = np.random.default_rng(100 * d + kind)
    U = rng.standard_normal((n, d)) * 0.5
    Ua = rng.standard_normal((m, d)) * 0.5
    w = rng.uniform(0.02, 0.2, d)
    sf2 = 0.83
    d_split = 0 if kind == 0 else max(dU, 2)  # the manifold dims stay RBF (models/gp_plus.py:223-226), the others Matern
    tau = np.array([2e-3, 3e-2])
    grp = rng.integers(0, S, n).astype(np.int32)
    K, _, _, _, _ = _kernel_and_derivs(U, U, w, sf2, kind, d_split)
    ref = K + np.diag(tau[grp] + 1e-6)
    dUm, dw = _dev(U), _dev(w)
    dsf2 = torch.tensor([sf2], dtype=torch.float64, device="cuda")
    for uplo, pick in ((0, lambda x: x), (2, np.triu)):
        out = _sq(n, fill=-7.0)
        gpu_ctx.kernel_build(dUm, dw, dsf2, _dev(tau), _dev(grp), out, jitter=1e-6, kind=kind, d_split=d_split, uplo=uplo)
        np.testing.assert_allclose(pick(out.cpu().numpy()), pick(ref), rtol=1e-12, atol=1e-14)
    cross = torch.empty(m, (n + 15) // 16 * 16, dtype=torch.float64, device="cuda")[:, :n]
    gpu_ctx.cross_kernel(_dev(Ua), dUm, dw, dsf2, cross, kind=kind, d_split=d_split)
    Kc, _, _, _, _ = _kernel_and_derivs(Ua, U, w, sf2, kind, d_split)
    np.testing.assert_allclose(cross.cpu().numpy(), Kc, rtol=1e-12, atol=1e-14)

    # gradient reduction against the written-out sums, with an arbitrary symmetric matrix in the place of Ky^-1
    alpha = rng.standard_normal(n)
    Kinv = rng.standard_normal((n, n))
    Kinv = Kinv + Kinv.T
    W = 0.5 * (np.outer(alpha, alpha) - Kinv)
    K, Gr, Gm, diff, dsp = _kernel_and_derivs(U, U, w, sf2, kind, d_split)
    g_w = np.array([(W * (Gr if k < dsp else Gm) * (-(diff[:, :, k] ** 2))).sum() for k in range(d)])
    g_sf2 = (W * K).sum() / sf2
    g_tau = np.array([np.diag(W)[grp == s].sum() for s in range(S)])
    g_U = np.stack([2 * (W * (Gr if k < dsp else Gm) * (-2 * w[k]) * diff[:, :, k]).sum(1) for k in range(dU)], 1) if dU else None
    Ki = _sq(n)
    Ki.copy_(_dev(np.tril(Kinv) + np.triu(np.full((n, n), 1e30), 1)))  # the strict upper triangle must never be used
    gw = torch.empty(d, dtype=torch.float64, device="cuda")
    gs = torch.empty(1, dtype=torch.float64, device="cuda")
    gt = torch.empty(S, dtype=torch.float64, device="cuda")
    gU = torch.empty(n, dU, dtype=torch.float64, device="cuda") if dU else None
    gpu_ctx.grad_reduce(dUm, dw, dsf2, _dev(grp), S, _dev(alpha), Ki, dU, gw, gs, gt, gU, kind=kind, d_split=d_split)
    sc = lambda x: 1e-10 * np.abs(x).max() + 1e-13
    np.testing.assert_allclose(gw.cpu().numpy(), g_w, rtol=1e-9, atol=sc(g_w))
    np.testing.assert_allclose(gs.cpu().numpy()[0], g_sf2, rtol=1e-9, atol=sc(g_sf2))
    np.testing.assert_allclose(gt.cpu().numpy(), g_tau, rtol=1e-9, atol=sc(g_tau))
    if dU:
        np.testing.assert_allclose(gU.cpu().numpy(), g_U, rtol=1e-9, atol=sc(g_U))


def test_grad_reduce_rejects_views_its_vector_loads_cannot_take(gpu_ctx):
    """gpp_grad_tiles reads Kinv with 16-byte loads: an odd leading dimension or a base that is not 16-byte aligned is a bad
    argument (LAPACK-style negative status -> GppError), never a misaligned access."""
    from gpplus_amd._lib import GppError

    n, d = 130, 3
    U = torch.randn(n, d, dtype=torch.float64, device="cuda")
    w = torch.full((d,), 0.1, dtype=torch.float64, device="cuda")
    sf2 = torch.tensor([1.0], dtype=torch.float64, device="cuda")
    al = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = [torch.empty(d, dtype=torch.float64, device="cuda"), torch.empty(1, dtype=torch.float64, device="cuda"),
           torch.empty(1, dtype=torch.float64, device="cuda")]
    odd = torch.zeros(n, n + 1, dtype=torch.float64, device="cuda")  # ld = 131
    with pytest.raises(GppError, match="bad argument"):
        gpu_ctx.grad_reduce(U, w, sf2, None, 1, al, odd[:, :n], 0, *out, None)
    shifted = torch.zeros(n, n + 14, dtype=torch.float64, device="cuda")[:, 1:n + 1]  # even ld, base 8 bytes off
    with pytest.raises(GppError, match="bad argument"):
        gpu_ctx.grad_reduce(U, w, sf2, None, 1, al, shifted, 0, *out, None)


@pytest.mark.parametrize("nranks,nb,n", [(1, 128, 700), (2, 128, 700), (3, 256, 1500), (4, 128, 900)])
def test_gemm_lower_cols_covers_the_owned_column_blocks_once(gpu_ctx, nranks, nb, n):
    """C(lower) = beta C + alpha A^T B restricted to block-cyclic column blocks and to a band of rows: the ranks' shares (in
    two row bands each) tile the lower triangle exactly once, nothing else is touched."""
    K = 192
    rng = np.random.default_rng(n + nranks)
    A, B, C0 = rng.standard_normal((K, n)), rng.standard_normal((K, n)), rng.standard_normal((n, n))
    ref = C0 - A.T @ B
    dA = torch.zeros(K, (n + 15) // 16 * 16, dtype=torch.float64, device="cuda")[:, :n]
    dB = torch.zeros(K, (n + 15) // 16 * 16, dtype=torch.float64, device="cuda")[:, :n]
    dA.copy_(_dev(A))
    dB.copy_(_dev(B))
    first_block = 2  # the region's first column block is global block 2
    split = (n // 2) // 128 * 128
    hits = np.zeros((n, n), dtype=int)
    for r in range(nranks):
        C = _sq(n)
        C.copy_(_dev(C0))
        gpu_ctx.gemm_lower_cols(dA, dB, C, -1.0, 1.0, nb, first_block, r, nranks, split, n)  # lower band first (as the sweep does)
        gpu_ctx.gemm_lower_cols(dA, dB, C, -1.0, 1.0, nb, first_block, r, nranks, 0, split)
        got = C.cpu().numpy()
        ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        mine = (jj <= ii) & (((jj // nb) + first_block) % nranks == r)
        np.testing.assert_allclose(got[mine], ref[mine], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(got[~mine], C0[~mine])  # other ranks' column blocks and the upper triangle: untouched
        hits += mine
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    assert (hits[jj <= ii] == 1).all() and (hits[jj > ii] == 0).all()


@pytest.mark.parametrize("nranks,nb,n", [(1, 128, 513), (2, 128, 1000), (3, 256, 1500)])
def test_trmv_lower_cols_and_mll_scalars(gpu_ctx, nranks, nb, n):
    """z = L^-1 r and alpha = L^-T z from column blocks: the ranks' partial results add up to the dense products, whatever sits
    in the blocks a rank does not own (NaN here) or above the diagonal."""
    rng = np.random.default_rng(n)
    T = np.tril(rng.standard_normal((n, n)))
    x = rng.standard_normal(n)
    z_ref, a_ref = T @ x, T.T @ x
    zs, as_ = np.zeros(n), np.zeros(n)
    for r in range(nranks):
        Tm = np.full((n, n), np.nan)
        for c0 in range(0, n, nb):
            if (c0 // nb) % nranks == r:
                Tm[:, c0:c0 + nb] = T[:, c0:c0 + nb] + np.triu(np.full((n, n), 1e30), 1)[:, c0:c0 + nb]
        dT = _sq(n)
        dT.copy_(_dev(Tm))
        y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        gpu_ctx.trmv_lower_cols(dT, _dev(x), y, nb, r, nranks, trans=False)
        zs += y.cpu().numpy()
        y2 = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        gpu_ctx.trmv_lower_cols(dT, _dev(x), y2, nb, r, nranks, trans=True)
        got = y2.cpu().numpy()
        own = ((np.arange(n) // nb) % nranks) == r
        assert (got[~own] == 0).all()
        as_ += got
    np.testing.assert_allclose(zs, z_ref, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(as_, a_ref, rtol=1e-11, atol=1e-11)
    # scalars from a given z: quad = z'z, logdet = 2 sum log U_ii
    Uf = _sq(n)
    diag = rng.uniform(0.5, 2.0, n)
    Uf.copy_(_dev(np.diag(diag) + np.triu(rng.standard_normal((n, n)), 1)))
    out3 = torch.empty(3, dtype=torch.float64, device="cuda")
    gpu_ctx.mll_scalars(Uf, _dev(z_ref), out3)
    quad, logdet = z_ref @ z_ref, 2 * np.log(diag).sum()
    np.testing.assert_allclose(out3.cpu().numpy(), [quad, logdet, -0.5 * (quad + logdet + n * np.log(2 * np.pi))], rtol=1e-12)


@pytest.mark.parametrize("nranks,nb", [(2, 128), (3, 256)])
def test_grad_reduce_cols_partial_sums_add_up(gpu_ctx, nranks, nb):
    n, d, S, dU = 900, 6, 2, 2
    rng = np.random.default_rng(9)
    U = _dev(rng.standard_normal((n, d)))
    w = _dev(rng.uniform(0.05, 0.6, d))
    grp = _dev(rng.integers(0, S, n).astype(np.int32))
    sf2 = torch.tensor([0.9], dtype=torch.float64, device="cuda")
    al = _dev(rng.standard_normal(n))
    Kf = rng.standard_normal((n, n))
    Kf = np.tril(Kf + Kf.T)
    Ki = _sq(n)
    Ki.copy_(_dev(Kf))
    mk = lambda: [torch.empty(d, dtype=torch.float64, device="cuda"), torch.empty(1, dtype=torch.float64, device="cuda"),
                  torch.empty(S, dtype=torch.float64, device="cuda"), torch.empty(n, dU, dtype=torch.float64, device="cuda")]
    ref = mk()
    gpu_ctx.grad_reduce(U, w, sf2, grp, S, al, Ki, dU, *ref)
    acc = [torch.zeros_like(t) for t in ref]
    for r in range(nranks):
        Km = np.full((n, n), np.nan)  # a rank holds its own column blocks only
        for c0 in range(0, n, nb):
            if (c0 // nb) % nranks == r:
                Km[:, c0:c0 + nb] = Kf[:, c0:c0 + nb]
        Kr = _sq(n)
        Kr.copy_(_dev(Km))
        part = mk()
        gpu_ctx.grad_reduce_cols(U, w, sf2, grp, S, al, Kr, dU, nb, r, nranks, *part)
        for a, p_ in zip(acc, part):
            assert torch.isfinite(p_).all()
            a += p_
    for a, t in zip(acc, ref):
        np.testing.assert_allclose(a.cpu().numpy(), t.cpu().numpy(), rtol=1e-10, atol=1e-10 * float(t.abs().max()))


def test_exp_for_nonpositive_arguments_is_accurate_to_a_few_ulp(gpu_ctx):
    """gpp_exp_nonpos (the branch-free exp of the covariance kernels) against numpy.exp over [-745, 0]: the parity bars of the
    kernels built on it are 1e-12 and tighter."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "exp_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "exp_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rel, ulp, den = mod.max_rel_err(400_000)
    assert ulp <= 4.0 and rel < 1e-15, (rel, ulp)
    assert den < 1e-307


@pytest.mark.parametrize("n,m", [(600, 130), (1000, 1), (257, 700), (5000, 3000)])
def test_predict_tn_matches_scipy_and_the_nt_form(gpu_ctx, n, m):
    """gpp_predict_tn (transposed cross block, V = Kns^T L^-T as a TN product against the mirror, mean = V z) against scipy and
    against gpp_predict on the same factor; the mean-only form of gpp_predict (V = NULL) as well."""
    import scipy.linalg as sla

    d = 7
    rng = np.random.default_rng(n + m)
    U = rng.standard_normal((n, d))
    w = rng.uniform(0.05, 0.5, d)
    def sqd(P, Q):  # weighted squared distances without the (p, q, d) intermediate (n = 5000: 1.4 GB)
        Pw, Qw = P * np.sqrt(w), Q * np.sqrt(w)
        return np.maximum((Pw ** 2).sum(1)[:, None] + (Qw ** 2).sum(1)[None, :] - 2.0 * Pw @ Qw.T, 0.0)

    K = 0.8 * np.exp(-sqd(U, U)) + 1e-3 * np.eye(n)
    Us = rng.standard_normal((m, d))
    r = rng.standard_normal(n)
    Ks = 0.8 * np.exp(-sqd(Us, U))
    cf = sla.cho_factor(K, lower=True)
    alpha = sla.cho_solve(cf, r)
    mean = Ks @ alpha
    Vref = sla.solve_triangular(cf[0], Ks.T, lower=True).T  # m x n
    var = 0.8 - (Vref ** 2).sum(1)
    A, Li, T = _sq(n), _sq(n), _sq(n)
    A.copy_(_dev(K))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    # n >= 3840 with the scratch: the look-ahead driver + cooperative panel + bordered inverse, the path every config but C1 takes
    gpu_ctx.potrf(A, Li, info, T if n >= 3840 else None)
    gpu_ctx.trtri(A, Li, T)
    assert int(info.item()) == 0
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    out3 = torch.empty(3, dtype=torch.float64, device="cuda")
    gpu_ctx.mll_reduce(A, Li, _dev(r), z, out3)
    ld = lambda k: max(16, (k + 15) // 16 * 16)
    Kns = torch.empty(n, ld(m), dtype=torch.float64, device="cuda")[:, :m]
    Kns.copy_(_dev(Ks.T))
    V = torch.full((m, ld(n)), float("nan"), dtype=torch.float64, device="cuda")[:, :n]
    mo = torch.empty(m, dtype=torch.float64, device="cuda")
    vo = torch.empty(m, dtype=torch.float64, device="cuda")
    kss = torch.full((m,), 0.8, dtype=torch.float64, device="cuda")
    gpu_ctx.predict_tn(Li, z, Kns, kss, V, mo, vo)
    sc = 1e-9 * np.abs(mean).max() + 1e-12
    np.testing.assert_allclose(mo.cpu().numpy(), mean, rtol=1e-8, atol=sc)
    np.testing.assert_allclose(vo.cpu().numpy(), var, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(V.cpu().numpy(), Vref, rtol=1e-7, atol=1e-9)
    # the [test][train] form, with and without the variance
    Ksn = torch.empty(m, ld(n), dtype=torch.float64, device="cuda")[:, :n]
    Ksn.copy_(_dev(Ks))
    al = torch.empty(n, dtype=torch.float64, device="cuda")
    gpu_ctx.alpha(Li, z, al)
    m2 = torch.empty(m, dtype=torch.float64, device="cuda")
    gpu_ctx.predict(Li, al, Ksn, None, None, m2, None)
    np.testing.assert_allclose(m2.cpu().numpy(), mean, rtol=1e-8, atol=sc)


# ---- cooperative panel (gpp_panel_potrf_inv): one launch factors AND inverts a block of up to 2048 rows ---------------------------
def _spd_block(n, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, 5))
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    return np.exp(-0.35 * d2) + 2e-3 * np.eye(n)


@pytest.mark.parametrize("n,use_ws", [(257, True), (384, False), (500, True), (777, False), (1000, True), (1024, True),
                                      (1152, False), (2000, True), (2048, False)])
def test_panel_factors_and_inverts_in_one_launch(gpu_ctx, n, use_ws):
    """256 < n <= 2048 on the caller's stream: the panel leaves U AND the complete inverse (lower + mirror), so gpp_trtri has
    nothing to do — checked by calling it on a poisoned scratch.  Ragged last leaves (n not a multiple of 128) included; the
    strict lower triangle of A is neither read nor written."""
    K = _spd_block(n, n)
    A, Li, T = _sq(n), _sq(n, 0.0), _sq(n)
    A.copy_(_dev(np.triu(K) + np.tril(np.full((n, n), np.nan), -1)))
    info = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info, T if use_ws else None)
    assert int(info.item()) == 0
    Lref = np.linalg.cholesky(K)
    Ah = A.cpu().numpy()
    np.testing.assert_allclose(np.triu(Ah), Lref.T, rtol=0, atol=1e-11 * np.abs(Lref).max())
    assert np.isnan(Ah[np.tril_indices(n, -1)]).all()
    before = Li.clone()
    gpu_ctx.trtri(A, Li, T)  # T is all NaN: any merge still done here would show
    assert torch.equal(Li, before)
    full = Li.cpu().numpy()
    np.testing.assert_allclose(np.tril(full) @ Lref, np.eye(n), rtol=0, atol=1e-9)
    np.testing.assert_array_equal(np.triu(full, 1), np.tril(full, -1).T)


def test_panel_reports_the_failing_minor_and_is_repeatable(gpu_ctx):
    n = 1000
    K = _spd_block(n, 3)
    bad = K.copy()
    bad[700, 700] = -1.0
    A, Li = _sq(n), _sq(n, 0.0)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    A.copy_(_dev(bad))
    gpu_ctx.potrf(A, Li, info)
    assert int(info.item()) == 701
    # the same block twice, and beside another stream's traffic: bitwise the same factor and inverse (every strip has one owner)
    outs = []
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, dtype=torch.float64, device="cuda")
    for rep in range(3):
        A.copy_(_dev(K))
        Li.zero_()
        torch.cuda.synchronize()
        if rep == 2:
            with torch.cuda.stream(side):
                for _ in range(8):
                    noise.fill_(float(rep))
        gpu_ctx.potrf(A, Li, info)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        outs.append((torch.triu(A).clone(), Li.clone()))
    for U2, L2 in outs[1:]:
        assert torch.equal(U2, outs[0][0]) and torch.equal(L2, outs[0][1])


def test_panel_inside_the_lookahead_reports_the_failing_minor(gpu_ctx):
    """n = 5000: the look-ahead's diagonal blocks are panel launches on the CU-masked stream; a bad pivot in the third block."""
    n = 5000
    K = _spd_block(n, 11)
    K[1300, 1300] = -5.0
    A, Li, T = _sq(n), _sq(n, 0.0), _sq(n)
    A.copy_(_dev(K))
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu_ctx.potrf(A, Li, info, T)
    assert int(info.item()) == 1301


def test_static_schedule_matches_the_launch_per_product_driver(gpu_ctx):
    """gpp_potrf_ws through the DAG executor (gpp_dag_f64, planned by gpp_dag.hip: factorisation and right-looking inverse as one
    ticket list) against launches per product + pair merges (GPP_OPT_DAG_SCHED = 0, of which GPP_OPT_EXEC_SCHED is an alias): same
    factor and inverse to rounding, both against scipy; and a bad pivot inside the list's steps is reported as the failing leading
    minor (the launch runs to its end on whatever the panel left: no counter waits on a value)."""
    import scipy.linalg as sla

    from gpplus_amd.backend import OPT_EXEC_SCHED

    n = 13500
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n, 6)) * np.sqrt(0.3)
    G = X @ X.T
    sq = np.diag(G)
    K = np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2 * G, 0.0)) + 2e-3 * np.eye(n)
    del G
    Kd = _dev(np.triu(K))
    A, Li, T = _sq(n), _sq(n, 0.0), _sq(n)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {}
    try:
        for on in (0, 1):
            gpu_ctx.set_option(OPT_EXEC_SCHED, on)
            A.copy_(Kd); Li.zero_()
            gpu_ctx.potrf(A, Li, info, T)
            gpu_ctx.trtri(A, Li, T)
            assert int(info.item()) == 0
            res[on] = (torch.triu(A).clone(), torch.tril(Li).clone())
        scale = float(res[0][0].abs().max())
        assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-12 * scale
        assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-10 * float(res[0][1].abs().max())
        Lref = sla.cholesky(K, lower=True)
        np.testing.assert_allclose(res[1][0].cpu().numpy().T, Lref, rtol=0, atol=1e-11 * scale)
        del Lref
        K[2500, 2500] = -5.0  # third diagonal block, inside the scheduled steps
        A.copy_(_dev(np.triu(K)))
        gpu_ctx.potrf(A, Li, info, T)
        assert int(info.item()) == 2501
    finally:
        gpu_ctx.set_option(OPT_EXEC_SCHED, 1)


@pytest.mark.parametrize("n", [7300, 20200])
def test_dag_list_equals_launches_at_its_range_ends(gpu_ctx, n):
    """The DAG executor at the two ends of its range that the other tests do not reach: just above its lower threshold (the whole
    inverse inside the list, a ragged last block) and above GPP_DAG_INV_MAX = 19 456 (only the leading 8192-row block of the inverse
    inside the list, gpp_trtri merges the rest around it) — against launches per product + pair merges on the same matrix: same
    factor and inverse to rounding, L X = I on a probe vector, the mirror exact, and bit-for-bit repeatable."""
    from gpplus_amd.backend import OPT_DAG_SCHED

    g = torch.Generator(device="cuda").manual_seed(n)
    U = torch.randn(n, 6, dtype=torch.float64, device="cuda", generator=g)
    w = torch.full((6,), 0.2, dtype=torch.float64, device="cuda")
    sf2 = torch.tensor([0.9], dtype=torch.float64, device="cuda")
    tau = torch.tensor([3e-3], dtype=torch.float64, device="cuda")
    A, Li, T = _sq(n), _sq(n, 0.0), _sq(n)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {}
    try:
        for on in (0, 1, 1):
            gpu_ctx.set_option(OPT_DAG_SCHED, on)
            gpu_ctx.kernel_build(U, w, sf2, tau, None, A, uplo=2)
            Li.zero_()
            gpu_ctx.potrf(A, Li, info, T)
            gpu_ctx.trtri(A, Li, T)
            assert int(info.item()) == 0
            cur = (torch.triu(A).clone(), Li.clone())
            if on in res:
                assert torch.equal(res[on][0], cur[0]) and torch.equal(res[on][1], cur[1])  # repeatable bit for bit
            res[on] = cur
    finally:
        gpu_ctx.set_option(OPT_DAG_SCHED, 1)
    scale = float(res[0][0].abs().max())
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-12 * scale
    li0, li1 = torch.tril(res[0][1]), torch.tril(res[1][1])
    assert float((li0 - li1).abs().max()) <= 1e-10 * float(li0.abs().max())
    assert float((torch.triu(res[1][1], 1) - torch.tril(res[1][1], -1).T).abs().max()) == 0.0
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    Lv = torch.mv(res[1][0].T, torch.mv(li1, v))  # L (X v), L = U^T
    assert float((Lv - v).norm() / v.norm()) <= 1e-9


def test_panel_timeout_is_recovered_by_the_host(gpu_ctx):
    """GPP_OPT_PANEL_FAULT makes the next panel launch report the time-out status (what a wait inside the kernel reports after ~1 s
    when another tenant of the GPU holds part of its CUs).  The C ABI returns it in ``info``; ``linalg`` switches the panel off for
    the context, factors again with leaf-step launches and the evaluation comes out as without the incident."""
    import warnings

    from gpplus_amd.backend import INFO_PANEL_TIMEOUT, OPT_COOP_PANEL, OPT_PANEL_FAULT
    from gpplus_amd.models import GP_Plus

    if not gpu_ctx.coop_panel:
        pytest.skip("the cooperative panel is switched off (GPP_COOP_PANEL=0)")
    n = 1000
    K = _spd_block(n, 5)
    A, Li = _sq(n), _sq(n, 0.0)
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    A.copy_(_dev(K))
    gpu_ctx.set_option(OPT_PANEL_FAULT, 1)
    gpu_ctx.potrf(A, Li, info)
    assert int(info.item()) == INFO_PANEL_TIMEOUT
    A.copy_(_dev(K))
    gpu_ctx.potrf(A, Li, info)  # one shot: the next launch is a real one
    assert int(info.item()) == 0
    # through the model: same loss and gradients as a clean evaluation, one warning, the panel off afterwards
    rng = np.random.default_rng(1)
    X = rng.uniform(size=(600, 4))
    y = np.sin(X.sum(1))
    m = GP_Plus(torch.tensor(X), torch.tensor(y), dtype=torch.float64, device=torch.device("cuda:0"))
    from gpplus_amd.gpcore import ExactMarginalLogLikelihood

    mll = ExactMarginalLogLikelihood(m.likelihood, m)

    def evaluate():
        m.zero_grad()
        loss = -mll(m(*m.train_inputs), m.train_targets)
        loss.backward()
        return loss.item(), [p.grad.clone() for p in m.parameters() if p.grad is not None]

    m.train()
    clean = evaluate()
    try:
        gpu_ctx.set_option(OPT_PANEL_FAULT, 1)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            hit = evaluate()
        assert any("panel launch timed out" in str(x.message) for x in w)
        assert not gpu_ctx.coop_panel
        # (the retry factors with leaf-step launches: equal to rounding, not bitwise)
        assert hit[0] == pytest.approx(clean[0], rel=1e-12)
        for a, b in zip(hit[1], clean[1]):
            torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-12)
    finally:
        gpu_ctx.set_option(OPT_PANEL_FAULT, 0)
        gpu_ctx.set_option(OPT_COOP_PANEL, 1)
    assert gpu_ctx.coop_panel


def test_replay_copy_kernel_obeys_its_rate_and_its_start_time(gpu_ctx):
    """``gpp_debug_replay_copy`` (gpp_shard.hip; tools/replay_rank.py plays other ranks' block rows with it): a strided 2-D copy by a
    few work-groups that (a) moves the right data, (b) takes at least bytes / rate, (c) does not start before epoch + not_before on
    the device's 100 MHz clock, and reports both stamps relative to the epoch."""
    import ctypes

    lib = gpu_ctx.lib
    lib.gpp_debug_replay_copy.restype = ctypes.c_int
    lib.gpp_debug_replay_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_longlong,
                                          ctypes.c_void_p]
    lib.gpp_debug_replay_stamp.restype = ctypes.c_int
    lib.gpp_debug_replay_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    dev = "cuda"
    rows, cols, ld = 512, 4096, 5000  # 16 MiB out of a wider matrix
    src = torch.randn(rows, ld, dtype=torch.float64, device=dev)
    dst = torch.zeros(rows, cols, dtype=torch.float64, device=dev)
    epoch = torch.zeros(1, dtype=torch.int64, device=dev)
    stamps = torch.zeros(2, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.gpp_debug_replay_stamp(s, epoch.data_ptr(), None) == 0
    rate, delay_ticks = 20.0, 300_000  # 20 GB/s; not before 3 ms after the epoch
    assert lib.gpp_debug_replay_copy(s, dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), rows, cols, 16, 512, rate,
                                     epoch.data_ptr(), delay_ticks, stamps.data_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dst, src[:, :cols])
    start, end = (int(v) for v in stamps.tolist())
    assert start >= delay_ticks and start < delay_ticks + 50_000, (start, delay_ticks)
    need = rows * cols * 8 / (rate * 10.0)  # ticks: rate GB/s = 10 rate bytes per tick of 10 ns
    assert end - start >= 0.98 * need and end - start < 1.5 * need, (end - start, need)
    # unthrottled, at once: faster than the paced copy by a wide margin
    stamps.zero_()
    assert lib.gpp_debug_replay_copy(s, dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), rows, cols, 32, 512, 0.0,
                                     epoch.data_ptr(), -1, stamps.data_ptr()) == 0
    torch.cuda.synchronize()
    start, end = (int(v) for v in stamps.tolist())
    assert 0 < end - start < 0.25 * need
    # odd column counts / leading dimensions are refused (16-byte vectors)
    assert lib.gpp_debug_replay_copy(s, dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), rows, cols - 1, 16, 512, 0.0,
                                     epoch.data_ptr(), -1, stamps.data_ptr()) != 0


def test_push_channel_waits_are_bounded_and_arguments_checked(gpu_ctx, monkeypatch):
    """``gpp_push_*`` (csrc/gpp_push.hip; the one-to-all push transport of gp-plus_amd/push.py, GPP_SHARD_PUSH=1): a receiver whose message
    never arrives gives up after GPP_SHARD_TIMEOUT_MS with the caller's code in the caller's status word — the GPU is not left hanging —
    and the entry points refuse message numbers below 1, parts that do not fit a slot and calls on an unconnected channel.  (The data
    path itself needs a peer: tests/test_gpu_00_sharded_lists.py::test_push_transport_equals_the_broadcast_bit_for_bit.)"""
    import ctypes
    import time

    from gpplus_amd.backend import INFO_EXEC_TIMEOUT
    from gpplus_amd.push import HANDLE_BYTES, _arrays

    monkeypatch.setenv("GPP_SHARD_TIMEOUT_MS", "250")
    lib = gpu_ctx.lib
    h = ctypes.c_void_p()
    rec = ctypes.create_string_buffer(HANDLE_BYTES)
    assert lib.gpp_push_create(gpu_ctx.index, 0, 1, 1 << 20, ctypes.byref(h), rec) == 0
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    dst = torch.zeros(4, 8, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    n, ptr, vp, off, sp, wd, ht = _arrays([(dst, 0, 8)])
    assert lib.gpp_push_recv(h, s, 1, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht) == -1  # not connected yet
    assert lib.gpp_push_connect(h, rec) == 0  # a group of one: nobody to map
    sb, kind = ctypes.c_int64(), ctypes.c_int()
    assert lib.gpp_push_info(h, ctypes.byref(sb), ctypes.byref(kind)) == 0 and sb.value == 1 << 20 and kind.value in (0, 1, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert lib.gpp_push_recv(h, s, 1, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht) == 0
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert 0.2 < waited < 5.0, waited
    assert int(status.item()) == INFO_EXEC_TIMEOUT
    assert lib.gpp_push_recv(h, s, 0, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht) == -3
    assert lib.gpp_push_ack(h, s, 0) == -3 and lib.gpp_push_ack(h, s, 1) == 0
    big = torch.zeros(1, (1 << 17) + 1, dtype=torch.float64, device="cuda")  # one double more than a slot holds
    n2, ptr2, vp2, off2, sp2, wd2, ht2 = _arrays([(big, 0, big.shape[1])])
    assert lib.gpp_push_recv(h, s, 2, status.data_ptr(), INFO_EXEC_TIMEOUT, n2, ptr2, vp2, off2, sp2, wd2, ht2) == -7
    assert lib.gpp_push_send(h, s, 2, status.data_ptr(), INFO_EXEC_TIMEOUT, n2, ptr2, vp2, off2, sp2, wd2, ht2) == -7
    assert lib.gpp_push_send(h, s, 2, status.data_ptr(), INFO_EXEC_TIMEOUT, n, ptr, vp, off, sp, wd, ht) == 0  # no peers: nothing moves
    torch.cuda.synchronize()
    assert lib.gpp_push_destroy(h, 2) == 0
