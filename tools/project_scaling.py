"""Falsifiable projection of both multi-GPU legs from MEASURED one-GPU numbers (no hardware needed; DESIGN.md section 7 prints this
table).  Nothing here has run on more than one GPU: when an 8-GPU node produces SCALE_rNN.json, read it against these rows.

Model of the sharded evaluation (gp-plus_amd/sharded.py), per evaluation, P ranks, block-cyclic blocks of nb rows:
  factor   right-looking; step k: the owner factors the diagonal block (panel) and solves the block row (nb^2 rem flop, NOT divided
           by P), the slab (8 nb (N - o_k) bytes) is broadcast, every rank updates the block rows it owns (nb rem^2 / P flop, the
           most loaded rank counts).  With one step of look-ahead the step costs
               T_k = max( update_k(P),  chain + solve_{k+1} + bcast_{k+1} )
  forward  / back-substitution: each rank's own column blocks, no communication: the most loaded rank's share of N^3/3 at the
           measured one-rank rate of that stage
  vectors, gradient: 1/P of the one-rank time + three small all-reduces
Rates are the one-rank stage times of profiles/r04_restarts_and_sharded_1rank.txt (re-measured in round 5: unchanged);
the broadcast rate is a PARAMETER (xGMI: 7 links x ~153 GB/s bidirectional per GPU = ~77 GB/s per direction and link; a large
RCCL broadcast from one root is bound by one link's direction): rows for 50 and 70 GB/s.
usage: python tools/project_scaling.py"""
import math

ONE_RANK = {  # measured, one rank, no communication (ms): factor, forward sweeps, back-substitution, vectors + gradient, build
    "C5": dict(N=60000, nb=2048, factor=1205.9, forward=1138.2, back=1103.9, small=17.1, panel_ms=1.3, chain_ms=0.5),
    "C2": dict(N=20000, nb=1024, factor=56.6, forward=51.0, back=48.3, small=1.6, panel_ms=0.58, chain_ms=0.35),
}
SINGLE_GPU_EVALS = {"C2": 7.70, "C5": 1 / 3.32}  # evals/s of the single-GPU path (bench.py / tools/run_configs.py, round 5 / round 4)


def blocks(N, nb):
    offs = list(range(0, N, nb)) + [N]
    return offs, len(offs) - 1


def factor_ms(cfg, P, bw_gbs, rate_scale=None):
    """(total ms, ms of broadcasts inside it, steps by binding term).  The flop rate is calibrated so that P = 1 reproduces the
    measured one-rank factor."""
    if rate_scale is None:
        f1 = factor_ms(cfg, 1, 1.0, rate_scale=1.0)[0]
        rate_scale = f1 / cfg["factor"]
    N, nb = cfg["N"], cfg["nb"]
    offs, nblk = blocks(N, nb)
    # flop of the one-rank factor: updates + solves (+ panels, negligible); its measured time gives the rate of both
    upd = [nb_k * (N - offs[k + 1]) ** 2 for k, nb_k in ((k, offs[k + 1] - offs[k]) for k in range(nblk))]
    sol = [(offs[k + 1] - offs[k]) ** 2 * (N - offs[k + 1]) for k in range(nblk)]
    chain_total = nblk * (cfg["panel_ms"] + cfg["chain_ms"])
    rate = rate_scale * (sum(upd) + sum(sol)) / max(cfg["factor"] - 0.3 * chain_total, 1e-9)  # flop per ms
    total, comm_total, bound = 0.0, 0.0, {"update": 0, "chain+solve+bcast": 0}
    for k in range(nblk):
        rem_blocks = list(range(k + 1, nblk))
        # most loaded rank: block rows i > k owned by r, row i costs ~ (N - o_i) * nb_i * nb_k * 2 / 2 ... use exact per-row flop
        per_rank = [0.0] * P
        for i in rem_blocks:
            per_rank[i % P] += (offs[k + 1] - offs[k]) * (offs[i + 1] - offs[i]) * (N - offs[i]) * 2 / 2 * 2  # nb_k * nb_i * (N - o_i) * 2 / ... upper part
        scale = upd[k] / max(sum(per_rank), 1e-9)
        t_upd = max(per_rank) * scale / rate if rem_blocks else 0.0
        t_next = 0.0
        if k + 1 < nblk:
            bytes_next = 8.0 * (offs[k + 2] - offs[k + 1]) * (N - offs[k + 1])
            t_bcast = bytes_next / (bw_gbs * 1e6) if P > 1 else 0.0
            comm_total += t_bcast
            t_next = cfg["panel_ms"] + cfg["chain_ms"] + sol[k + 1] / rate + t_bcast
        step = max(t_upd, t_next)
        bound["update" if t_upd >= t_next else "chain+solve+bcast"] += 1
        total += step
    total += cfg["panel_ms"] + sol[0] / rate + (8.0 * (offs[1] - offs[0]) * N / (bw_gbs * 1e6) if P > 1 else 0.0)  # step 0's own panel, solve, broadcast
    return total, comm_total, bound


def sweep_share(cfg, P):
    """Most loaded rank's share of a sweep's flop: column block c costs ~ (N - o_c)^2 * width_c."""
    N, nb = cfg["N"], cfg["nb"]
    offs, nblk = blocks(N, nb)
    w = [(N - offs[c]) ** 2 * (offs[c + 1] - offs[c]) for c in range(nblk)]
    per = [sum(w[c] for c in range(r, nblk, P)) for r in range(P)]
    return max(per) / sum(w)


def main():
    print("Sharded single evaluation (strong scaling), projected from one-rank stage times; NOT measured on more than one GPU")
    print(f"{'cfg':3s} {'P':>2s} {'GB/s':>5s} | {'factor':>8s} {'(comm in it)':>12s} {'steps bound by update / chain+solve+bcast':>42s} | "
          f"{'forward':>8s} {'back':>8s} {'small':>6s} | {'total ms':>9s} {'evals/s':>8s} {'speed-up':>8s} {'GB recv/GPU':>11s}")
    for name, cfg in ONE_RANK.items():
        base = None
        for P in (1, 2, 4, 8):
            for bw in ((50.0, 70.0) if P > 1 else (0.0,)):
                f, comm, bound = factor_ms(cfg, P, bw if P > 1 else 1.0)
                share = sweep_share(cfg, P)
                fwd, back = cfg["forward"] * share, cfg["back"] * share
                small = cfg["small"] / P + (0.15 if P > 1 else 0.0)  # three small all-reduces
                total = f + fwd + back + small
                if base is None:
                    base = total
                N = cfg["N"]
                recv = 8.0 * N * N / 2 / 1e9 if P > 1 else 0.0
                print(f"{name:3s} {P:2d} {bw:5.0f} | {f:8.1f} {comm:12.1f} {bound['update']:20d} / {bound['chain+solve+bcast']:<19d} | "
                      f"{fwd:8.1f} {back:8.1f} {small:6.1f} | {total:9.1f} {1e3 / total:8.3f} {base / total:8.2f} {recv:11.1f}")
    print()
    print("Replicas (weak scaling, bench.py --gpus P): no data-path collective; projected = P x the single-GPU rate x [0.95, 1.00]")
    print("(independent processes; the pool's boxes differ by up to 5 % between GPUs)")
    for name, r in SINGLE_GPU_EVALS.items():
        print("  " + name + ": " + "  ".join(f"P={P}: {0.95 * P * r:.2f}-{P * r:.2f} evals/s" for P in (1, 2, 4, 8)))
    print()
    N, nb = 60000, 2048
    print("Capacity with the replicated factor (one N x N fp64 matrix per rank + owned column blocks of two more + five nb-row strips):")
    for P in (1, 8):
        for hbm in (288.0,):
            # bytes(N) = 8 N^2 (1 + 2 ceil(nblk / P) nb / N) + 5 * 8 nb N  <= 0.92 * HBM
            lo, hi = 1000, 400000
            while hi - lo > 64:
                mid = (lo + hi) // 2
                nblk = math.ceil(mid / nb)
                b = 8.0 * mid * mid * (1 + 2 * math.ceil(nblk / P) * nb / mid) + 5 * 8.0 * nb * mid
                lo, hi = (mid, hi) if b <= 0.92 * hbm * 1e9 else (lo, mid)
            print(f"  P={P}: largest N ~ {lo}  ({8e-9 * lo * lo:.0f} GB for the factor alone)")
    print("  2-D block-cyclic (Pr x Pc = 2 x 4) would store N^2 / P of every matrix: N ~ "
          f"{int(math.sqrt(0.92 * 288e9 * 8 / (3 * 8.0)))} at P = 8 — at the price of row AND column broadcasts per step "
          "(2 x 8 nb (N - o) / sqrt(P)-ish bytes) and a distributed panel; not built")


if __name__ == "__main__":
    main()
