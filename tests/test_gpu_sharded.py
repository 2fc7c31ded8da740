"""Sharded single evaluation (gp-plus_amd/sharded.py, SURVEY.md §8(e) mode 2): two and more ranks against the single-GPU path.
On a box with one GPU the ranks share it and communicate over gloo; with one GPU per rank they use RCCL.  The RCCL branch
itself is exercised on one GPU by a group of ONE rank with every collective forced (GPP_SHARDED_FORCE_COLLECTIVES)."""
import json, os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from sharded_launch import assert_close_values as _assert_close_values, config_values as _config_values, run_ranks as _run  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,nb,kind,S,dU", [
    (1500, 5, 256, 0, 1, 0),     # ragged last block row (1500 = 5 * 256 + 220)
    (1024, 8, 128, 0, 3, 2),     # per-group noise + gradients w.r.t. the first two feature columns (manifold dims)
    (900, 6, 384, 2, 1, 0),      # Matern 5/2 on the dims >= 2, three block rows on two ranks
])
def test_sharded_matches_single_gpu(N, D, nb, kind, S, dU):
    out = _run([N, D, nb, kind, S, dU], port=29531 + (N % 7))
    for name, e in out["err"].items():
        assert e < 1e-9, (name, e, out)   # bar: 1e-5 relative (BASELINE north_star); observed ~1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("world,N,D,nb,kind,S,dU", [
    (3, 8200, 6, 1024, 0, 1, 0),   # 9 block rows (8 x 1024 + 8) on 3 ranks: the look-ahead / batched inverse sweep at C5-like block counts
    (4, 4500, 5, 512, 0, 2, 2),    # 9 block rows on 4 ranks: P does not divide the block count; per-group noise; manifold gradients
    (3, 300, 4, 256, 0, 1, 0),     # 2 block rows on 3 ranks: a rank that owns nothing (P > number of block rows)
    (4, 128, 3, 128, 0, 1, 0),     # one block: three idle ranks
    (2, 2176, 7, 128, 1, 1, 0),    # 17 leaf-sized block rows (nb = 128): owners alternate every 128 rows; Matern 3/2
    (2, 20000, 8, 1024, 0, 1, 0),  # the C2 size on two ranks: 20 block rows of 1024, the grouped LAUUM pipeline with real groups
])
def test_sharded_more_ranks_and_block_counts(world, N, D, nb, kind, S, dU):
    # (the launch-per-product choreography of rounds 2-4; the ticket lists have tests of their own in test_gpu_00_sharded_lists.py)
    out = _run([N, D, nb, kind, S, dU], world=world, port=29560 + (N * 7 + world * 13 + nb) % 400, GPP_SHARD_LIST="0")
    for name, e in out["err"].items():
        assert e < 1e-9, (name, e, out)
    # storage per rank ("28.8 GB (3.6 GB/GPU sharded)"): only the factor is N x N; L^-1 and Ky^-1 exist as the owned
    # column blocks — ceil(blocks / P) blocks of nb columns each
    nblk = -(-N // nb)
    assert out["owned_cols"] == max(-(-nblk // world), 1) * nb, out
    if nblk >= 2 * world:
        # (factor + two compact matrices + five nb-row strips: diagonal inverses, three row-solve scratches, packing buffer)
        assert out["matrix_bytes"] < (1.0 + 2.0 * (-(-nblk // world)) / nblk + 5.0 * nb / N + 0.1) * out["full_matrix_bytes"], out


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,nb,kind,S,dU", [
    (20000, 8, 1024, 0, 1, 0),   # the C2 size: 20 slab broadcasts of up to 164 MB, the split updates and the panel stream around them
    (3000, 6, 256, 0, 2, 2),     # ragged last block, per-group noise, manifold gradients (the all-reduce carries N dU doubles)
])
def test_sharded_rccl_branch_with_one_rank(N, D, nb, kind, S, dU):
    """``init_process_group("nccl", world_size=1)`` + GPP_SHARDED_FORCE_COLLECTIVES=1: ``_Comm.direct`` — dist.broadcast /
    all_reduce on device memory from the CU-masked side stream, the packing buffers' reuse, RCCL's own stream ordered against
    the library's three internal streams — runs on the 1-GPU box, and must reproduce the single-GPU path."""
    out = _run([N, D, nb, kind, S, dU], world=1, port=29900 + N % 50, GPP_TEST_BACKEND="nccl",
               GPP_SHARDED_FORCE_COLLECTIVES="1")
    assert out["backend"] == "nccl"
    if N == 20000:  # the ticket lists at the C2 size, RCCL's kernels beside the executor's
        assert (out["list_evals"], out["back_list_evals"]) == (1, 1), (out["list_evals"], out["back_list_evals"], out["status_lines"])
    nblk = -(-N // nb)
    assert out["collectives"] >= 2 * nblk + 2, out  # head + tail per block row (the last has no tail), info, z, alpha, gradient
    for name, e in out["err"].items():
        assert e < 1e-9, (name, e, out)


@pytest.mark.gpu
def test_sharded_jitter_retries_and_failure_are_collective():
    """A covariance that is indefinite at first (duplicated rows, slightly negative noise): every rank learns the failure from the
    all-reduced status word, retries with the same jitter (gpytorch's schedule) and ends on the single-GPU path's values — or, when
    no jitter suffices, EVERY rank raises NotPSDError instead of one of them hanging in a collective the others never enter."""
    out = _run([1024, 5, 256, 0, 1, 0, "jitter"], world=2, port=29610)
    for name, e in out["err"].items():
        assert e < 1e-4, (name, e, out)  # (a matrix lifted by 5e-8: condition ~1e7, both paths round differently)
    out = _run([1024, 5, 256, 0, 1, 0, "notpsd"], world=3, port=29611)
    assert out["raised"] == {"sharded": "NotPSDError", "single": "NotPSDError"}, out


@pytest.mark.gpu
def test_sharded_through_gp_plus_api():
    """settings.sharded_evaluation routes GP_Plus's own loss through the cooperative evaluation (mixed-input model)."""
    out = _run([700, 8, 256, 0, 1, 2, "model"], port=29547)
    assert out["err"]["loss_and_grads"] < 1e-8, out


@pytest.mark.gpu
def test_sharded_multifidelity_model_through_gp_plus():
    """BASELINE config C4's model (three sources: per-source noise levels, per-source means, the source column manifold-encoded)
    at N = 1500 on three ranks, block height 256: loss and every gradient — latent map, three noises, two means, lengthscales —
    against the single-GPU path."""
    single, shard = _config_values("C4", 256, 3, n=1500, port=29655)
    assert any("raw_noise" in k for k in single) and any(k.startswith("latent") for k in single)
    _assert_close_values(single, shard, 1e-8)


@pytest.mark.gpu
def test_sharded_evaluation_without_gradient():
    """A gradient-free evaluation (``torch.no_grad()``: validation loss) skips the back-substitution and alpha; the value is the
    same."""
    single, shard = _config_values("C3", 256, 2, n=1200, nograd=True, port=29656)
    assert list(single) == ["loss"]
    _assert_close_values(single, shard, 1e-9)


@pytest.mark.gpu
def test_sharded_c5_size_two_ranks():
    """BASELINE config C5 at FULL size (N = 60 000, d = 16: 59 block rows of 1024) through GP_Plus, sharded over two ranks
    (one GPU, gloo: 2 x 87 GB) against the single-GPU path run on its own beforehand (86 GB): loss and every gradient."""
    import torch

    if torch.cuda.get_device_properties(0).total_memory < 200 * 2 ** 30:
        pytest.skip("needs ~175 GiB of device memory")
    single, shard = _config_values("C5", 1024, 2, port=29977, GPP_SHARD_LIST="0")  # (the launches; the lists: test_gpu_00_sharded_lists.py)
    _assert_close_values(single, shard, 1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("backend,force,N,nb", [("gloo", "0", 4300, 256), ("nccl", "1", 4300, 256), ("nccl", "1", 9000, 512), ("gloo", "0", 9000, 512)])
def test_sharded_choreography_is_bitwise_repeatable(backend, force, N, nb):
    """The sharded evaluation enqueues on five streams (panel, throughput, bulk, collectives, the caller's) tied by events; every
    kernel is deterministic, so repeating one evaluation must reproduce loss and gradients BIT FOR BIT — any difference is a race
    (tools/stress_sharded.py; 17 block rows, per-group noise, manifold gradients, 20 repetitions; N = 9000 with nb = 512 runs the
    ticket lists, 4300 with 256 — a last block too short for the panel — the launches)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPP_SHARDED_FORCE_COLLECTIVES=force,
               MASTER_PORT=str(29720 + (backend == "nccl") + 2 * (N == 9000)))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_sharded.py"), str(N), str(nb), "20", backend],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "20 repetitions, 0 differ" in p.stdout, p.stdout[-1000:]
