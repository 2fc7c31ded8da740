"""Falsifiable projection of both multi-GPU legs from MEASURED one-GPU numbers (no hardware needed; DESIGN.md section 7 prints this
table).  Nothing here has run on more than one GPU: when an 8-GPU node produces SCALE_rNN.json, read it against these rows.

Model of the sharded evaluation with round 5's ticket lists (gp-plus_amd/sharded.py + csrc/gpp_dag.hip DAG_SHARD), P ranks, blocks of nb rows:
  factor + forward sweep (ONE list per rank): a rank's tile tasks — its share of the trailing updates (block rows it owns), the running
           sums of ITS column blocks of L^-1 against every block row, its own row solves — are taken in priority order whenever their
           operands are there, so the stage takes the LARGEST of
               work:   the most loaded rank's flop / the one-rank rate (measured: the one-rank list runs 2 N^3 / 3 flop in `ff` ms)
               chain:  sum over steps of max( diagonal update + panel + head solve + copy + gate + head message + signal ,
                                              the step's bytes / broadcast rate )   — heads and tails travel in order on one stream
           plus the first step's start-up.  (The launch-per-product path of rounds 2-4 paid per step max(update, chain + ROW SOLVE +
           broadcast): the owner's row solve is now one more set of tasks beside everybody's updates.)
  back-substitution (one list per rank, no communication): the most loaded rank's share of its flop at the measured one-rank rate
  vectors, gradient: 1/P of the one-rank time + three small all-reduces
The broadcast rate is a PARAMETER (xGMI: 7 links x ~153 GB/s bidirectional per GPU = ~77 GB/s per direction and link; a large RCCL
broadcast from one root is bound by one link's direction): rows for 50 and 70 GB/s; `overlap` = how much of the smaller of work and
chain hides behind the larger (1.0: all of it; 0.7: a third of it shows) — rows for both.
usage: python tools/project_scaling.py"""
import math

ONE_RANK = {  # measured, one rank, ticket lists (profiles/r05_sharded_lists_1rank.txt), ms: factor + forward list (incl. build, mirror),
              # back-substitution list, vectors + gradient; chain pieces from the list's task traces (profiles/r05_dag_traces.txt)
    "C5": dict(N=60000, nb=1024, ff=2272.2, back=1140.8, small=20.5, chain_ms=0.15 + 0.63 + 0.16 + 0.16 + 0.10),
    "C2": dict(N=20000, nb=1024, ff=90.3, back=43.6, small=2.8, chain_ms=0.15 + 0.63 + 0.16 + 0.16 + 0.10),
}
SINGLE_GPU_EVALS = {"C2": 7.80, "C5": 1 / 3.21}  # evals/s of the single-GPU path (bench.py / tools/run_configs.py, round 5)


def blocks(N, nb):
    offs = list(range(0, N, nb)) + [N]
    return offs, len(offs) - 1


def ff_ms(cfg, P, bw_gbs, overlap):
    """(stage ms, ms the messages occupy the communication stream, work ms of the most loaded rank, chain ms)."""
    N, nb = cfg["N"], cfg["nb"]
    offs, nblk = blocks(N, nb)
    h = [offs[k + 1] - offs[k] for k in range(nblk)]
    per = [0.0] * P
    for k in range(nblk):
        rem = N - offs[k + 1]
        per[k % P] += h[k] * h[k] * rem  # row solve (triangular: half of 2 nb^2 rem)
        for i in range(k + 1, nblk):
            per[i % P] += 2.0 * h[k] * h[i] * (N - offs[i] - h[i] / 2.0)  # update of block row i: columns from its diagonal on
        for r in range(P):
            cols = sum(h[c] for c in range(r, k + 1, P))  # owned column blocks up to k
            per[r] += 2.0 * h[k] * rem * cols                      # running sums of every row below block k
            per[r] += h[k] * h[k] * sum(h[c] for c in range(r, k, P))  # rows of X in block k (triangular)
    rate = sum(per) / cfg["ff"]  # flop per ms, calibrated at one rank (sum(per) is P-independent)
    work = max(per) / rate
    comm = chain = 0.0
    for k in range(nblk):
        msg = 8.0 * h[k] * (N - offs[k] + h[k]) / (bw_gbs * 1e6) if P > 1 else 0.0
        comm += msg
        head = 8.0 * h[k] * (min(2 * nb, N - offs[k]) + h[k]) / (bw_gbs * 1e6) if P > 1 else 0.0
        chain += max(cfg["chain_ms"] + head + (0.05 if P > 1 else 0.0), msg)
    lo, hi = min(work, chain), max(work, chain)
    return hi + (1.0 - overlap) * lo, comm, work, chain


def back_share(cfg, P):
    """Most loaded rank's share of the back-substitution's flop: column block c costs ~ (N - o_c)^2 * width_c."""
    N, nb = cfg["N"], cfg["nb"]
    offs, nblk = blocks(N, nb)
    w = [(N - offs[c]) ** 2 * (offs[c + 1] - offs[c]) for c in range(nblk)]
    per = [sum(w[c] for c in range(r, nblk, P)) for r in range(P)]
    return max(per) / sum(w)


def main():
    print("Sharded single evaluation (strong scaling) with round 5's ticket lists, projected from one-rank stage times; NOT measured on more")
    print("than one GPU (multi-rank lists run and match on ONE GPU shared by 2-4 ranks: tests/test_gpu_sharded.py)")
    print(f"{'cfg':3s} {'P':>2s} {'GB/s':>5s} {'overlap':>7s} | {'factor+forward':>14s} {'(work':>8s} {'chain':>8s} {'messages)':>9s} | "
          f"{'back':>8s} {'small':>6s} | {'total ms':>9s} {'evals/s':>8s} {'speed-up':>8s} {'GB recv/GPU':>11s}")
    for name, cfg in ONE_RANK.items():
        base = None
        for P in (1, 2, 4, 8):
            for bw in ((50.0, 70.0) if P > 1 else (0.0,)):
                for ov in ((0.7, 1.0) if P > 1 else (1.0,)):
                    f, comm, work, chain = ff_ms(cfg, P, bw if P > 1 else 1.0, ov)
                    back = cfg["back"] * back_share(cfg, P)
                    small = cfg["small"] / P + (0.15 if P > 1 else 0.0)  # three small all-reduces
                    total = f + back + small
                    if base is None:
                        base = total
                    N = cfg["N"]
                    recv = 8.0 * N * N / 2 / 1e9 if P > 1 else 0.0
                    print(f"{name:3s} {P:2d} {bw:5.0f} {ov:7.1f} | {f:14.1f} {work:8.1f} {chain:8.1f} {comm:9.1f} | "
                          f"{back:8.1f} {small:6.1f} | {total:9.1f} {1e3 / total:8.3f} {base / total:8.2f} {recv:11.1f}")
    print()
    print("Replicas (weak scaling, bench.py --gpus P): no data-path collective; projected = P x the single-GPU rate x [0.95, 1.00]")
    print("(independent processes; the pool's boxes differ by up to 5 % between GPUs)")
    for name, r in SINGLE_GPU_EVALS.items():
        print("  " + name + ": " + "  ".join(f"P={P}: {0.95 * P * r:.2f}-{P * r:.2f} evals/s" for P in (1, 2, 4, 8)))
    print()
    N, nb = 60000, 1024
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
