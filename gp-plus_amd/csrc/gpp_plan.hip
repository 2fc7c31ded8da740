// gpp_plan.hip — host-side planner of the static-schedule executor (gpp_exec_f64, gpp_gemm.hip) for the throughput-bound
// steps of the look-ahead Cholesky (gpp_api.hip::potrf_lookahead).  No reference counterpart: the reference's factorisation is
// torch.linalg.cholesky_ex behind gpytorch's psd_safe_cholesky (call site optim/mll_torch.py:116).
//
// What is planned.  Steps k = 0 .. K-1 of the right-looking factorisation over block rows of nb = bt * 128 rows (upper storage,
// A = U^T U), as ONE persistent launch of W workers on the throughput CUs plus, per step, one short "filler" launch of <= F
// workers on the panel's CUs between two diagonal-block factorisations.  Tasks are 128 x 128 tiles of three products per step:
//   S(b; r, c)   row solve      T[b rows, c] = W_bb^T A[b rows, c]          (K <= nb, triangular; output to the scratch T so that the
//                                                                            8 row tiles of a column strip may run in parallel)
//   U(k; i, j)   trailing update A[i, j] -= T[k rows, i]^T T[k rows, j]     (K = nb; upper triangle only on diagonal tiles)
//   CP(k; c)     copy            A[k rows, c] = T[k rows, c]                (the factor's block row k into place)
// and the dependencies are monotone counters (ids below).  Ownership makes most of them implicit: tile t (row-major index in the
// upper triangle of tiles) belongs to main worker t mod W for the whole factorisation — the active tiles of a step are a SUFFIX of
// that order, so a cyclic deal is balanced to +-1 tile at every step and successive updates of a tile need no flag — except the
// last F * m_k tiles (the bottom rows), which step k's filler launch updates; that region only shrinks, and a tile handed back to
// its main owner waits once for the filler launch that last wrote it.
#include "gpp_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

enum { C_PD = 0, C_G1D = 1, C_HR = 2, C_RR = 3, C_SH = 4, C_SA = 5, C_FD = 6 };
inline int cid(int b, int which) { return 1 + 8 * b + which; }

struct Builder {
  std::vector<ExecTask> list;
  std::vector<std::pair<int, int>> known;  // (counter, value) this worker has already observed
  bool knows(int c, int v) const {
    for (auto& k : known)
      if (k.first == c && k.second >= v) return true;
    return false;
  }
  void learn(int c, int v) {
    for (auto& k : known)
      if (k.first == c) {
        k.second = std::max(k.second, v);
        return;
      }
    known.emplace_back(c, v);
  }
  // waits: up to two (counter, value) pairs with counter < 0 meaning none; those already observed by this worker are dropped
  bool add(int group, int tm, int tn, int w0, int v0, int w1, int v1, int i0, int i1) {
    ExecTask t;
    t.group = group;
    t.tm = (int16_t)tm;
    t.tn = (int16_t)tn;
    int n = 0;
    t.wait_id[0] = t.wait_id[1] = -1;
    t.wait_val[0] = t.wait_val[1] = 0;
    auto want = [&](int c, int v) {
      if (c < 0 || v <= 0 || knows(c, v)) return;
      t.wait_id[n] = c;
      t.wait_val[n] = v;
      ++n;
      learn(c, v);
    };
    want(w0, v0);
    want(w1, v1);
    t.inc_id[0] = i0;
    t.inc_id[1] = i1;
    list.push_back(t);
    return true;
  }
};

}  // namespace

void gpp_plan_bind(PotrfExecPlan* P, double* A, int64_t ld, double* Li, int64_t ldi, double* T, int64_t ldt) {
  const int64_t N = P->N, nb = P->nb;
  P->groups.assign((size_t)3 * P->K, GemmArgs{});
  for (int k = 0; k < P->K; ++k) {
    const int64_t o = (int64_t)k * nb, c0 = o + nb, rem = N - c0;
    GemmArgs s{};  // S(k): T[o.., c0..) = W_oo^T A[o.., c0..)   (W_oo: mirror of the block's inverse, keep k <= row)
    s.A = Li + o * ldi + o; s.lda = ldi;
    s.B = A + o * ld + c0; s.ldb = ld;
    s.C = T + o * ldt + c0; s.ldc = ldt;
    s.M = (int)nb; s.N = (int)rem; s.K = (int)nb;
    s.alpha = 1.0; s.beta = 0.0;
    s.a_mask = 1; s.khi_mode = 1;
    s.pad_ok = 1;  // (rows o .. o+nb-1 of A with o + nb < N: reading past column N stays inside the buffer)
    P->groups[3 * k] = s;
    GemmArgs u{};  // U(k): A[c0.., c0..) -= T[o.., c0..)^T T[o.., c0..), upper triangle
    u.A = T + o * ldt + c0; u.lda = ldt;
    u.B = u.A; u.ldb = ldt;
    u.C = A + c0 * ld + c0; u.ldc = ld;
    u.M = u.N = (int)rem; u.K = (int)nb;
    u.alpha = -1.0; u.beta = 1.0;
    u.c_lower = 2;
    u.pad_ok = 1;  // (rows o .. o+nb-1 of T, never the buffer's last row)
    P->groups[3 * k + 1] = u;
    GemmArgs c{};  // CP(k): A[o.., c0..) = T[o.., c0..)
    c.B = T + o * ldt + c0; c.ldb = ldt;
    c.C = A + o * ld + c0; c.ldc = ld;
    c.M = (int)nb; c.N = (int)rem;
    c.op = 1;
    P->groups[3 * k + 2] = c;
  }
  P->A = A; P->ld = ld; P->Li = Li; P->ldi = ldi; P->T = T; P->ldt = ldt;
}

void gpp_plan_free(PotrfExecPlan* P) {
  if (!P) return;
  if (P->d_groups) (void)hipFree(P->d_groups);
  if (P->d_tasks) (void)hipFree(P->d_tasks);
  if (P->d_offsets) (void)hipFree(P->d_offsets);
  if (P->d_counters) (void)hipFree(P->d_counters);
  if (P->d_trace) (void)hipFree(P->d_trace);
  delete P;
}

// Builds the task lists (pointer independent).  W main workers, F filler work-groups per step (0: no filler), K steps.
//
// Order of a main worker's list — ONE step of look-ahead, as in the launch-per-product driver: while the bulk of step k's update
// (rows below block row k+2) runs, the worker also (1) solves its share of block row k+1 — whose diagonal block the panel stream
// factored during the PREVIOUS phase —, (2) applies step k+1's update to its tiles of block row k+2, which completes that block
// row and releases the panel of diagonal block k+2.  Every counter a task waits for is therefore raised about half a phase before
// it is needed, and a worker that runs ahead or behind by less than that never stalls:
//   prologue : S(0) share, U(0) on block row 1
//   phase k  : bulk[0, ps) | S(k+1) share | bulk[ps, pl) | U(k+1) on block row k+2 | bulk[pl, ...) | CP(k) share
// (bulk = the worker's U(k) tiles in rows >= block row k+2 in row-major order, so its tiles of block row k+2 come first: they
// precede the look-ahead tasks on the same tiles.)  Panel stream: panel(0), signal, gate(1), panel(1), signal, gate(2), panel(2),
// signal, F(0), gate(3), panel(3), signal, F(1), ...: the filler launch F(k) — step k's update of the bottom rows — follows the
// diagonal block k+2 and ends before block k+3 is due.
PotrfExecPlan* gpp_plan_potrf_exec(int64_t N, int64_t nb, int K, int W, int F, const PotrfExecTuning& tune) {
  if (nb % GPP_TILE != 0 || K < 1 || W < 1 || (int64_t)K * nb >= N) return nullptr;
  const int bt = (int)(nb / GPP_TILE);
  const int nt = (int)((N + GPP_TILE - 1) / GPP_TILE);
  if (nt >= 32000) return nullptr;  // tile coordinates are 16-bit
  auto tix = [&](int64_t i, int64_t j) -> int64_t { return i * nt - i * (i - 1) / 2 + (j - i); };
  const int64_t total = (int64_t)nt * (nt + 1) / 2;
  const int Fs = std::max(F, 1);

  PotrfExecPlan* P = new PotrfExecPlan();
  P->N = N; P->nb = nb; P->K = K; P->W = W; P->F = F;
  P->ncounters = 1 + 8 * (K + 2);
  P->gate_target.assign(K + 2, 0);
  P->fill_workers.assign(K, 0);

  struct Proto { int group, tm, tn, w0, v0, w1, v1, i0, i1, row; };
  std::vector<std::vector<std::vector<Proto>>> pS(K), pLA(K), pBulk(K), pCP(K);  // [step][worker]
  std::vector<Builder> fill((size_t)K * Fs);
  std::vector<int> hr(K + 2, 0), rr(K + 2, 0), sh(K + 2, 0), sa(K + 2, 0), fd(K + 2, 0);
  std::vector<int64_t> endk(K + 1, total);
  int next_s = 0, next_c = 0;

  for (int k = 0; k < K; ++k) {
    const int lo = bt * (k + 1);
    pS[k].assign(W, {});
    pLA[k].assign(W, {});
    pBulk[k].assign(W, {});
    pCP[k].assign(W, {});
    // the classes of step k's tiles inside block row k+1 (targets of the counters that release block k+1's solve and panel)
    int g1 = 0, h = 0, r = 0;
    for (int i = lo; i < std::min(lo + bt, nt); ++i)
      for (int j = i; j < nt; ++j) {
        if (j < lo + bt) ++g1; else if (j < lo + 2 * bt) ++h; else ++r;
      }
    P->gate_target[k + 1] = g1;
    hr[k + 1] = h;
    rr[k + 1] = r;
    // S(k): head columns (those of diagonal block k+1) first, longest K (last row tile) first, dealt round-robin
    for (int pass = 0; pass < 2; ++pass)
      for (int rt = bt - 1; rt >= 0; --rt)
        for (int c = lo; c < nt; ++c) {
          const bool head = c < lo + bt;
          if (head != (pass == 0)) continue;
          sa[k] += 1;
          if (head) sh[k] += 1;
          pS[k][next_s].push_back({3 * k, rt, c - lo, cid(k, C_PD), 1, k > 0 ? cid(k, head ? C_HR : C_RR) : -1,
                                   k > 0 ? (head ? hr[k] : rr[k]) : 0, cid(k, C_SA), head ? cid(k, C_SH) : -1, 0});
          next_s = (next_s + 1) % W;
        }
    // filler share of this step: m tiles per filler work-group, taken from the END of the row-major order (the bottom rows)
    const int64_t start = tix(lo, lo), n_act = total - start;
    int64_t e = total;
    if (F > 0 && tune.fill) {
      const double t_main = (double)n_act * tune.t_tile / W;
      int64_t m = (int64_t)((t_main - tune.t_block) / (tune.t_tile * (1.0 + (double)F / W)));
      m = std::max<int64_t>(m, 0);
      // never a tile of the next three block rows: those are on the look-ahead's path within a phase
      const int64_t first_ok = (int64_t)bt * (k + 4) < nt ? tix((int64_t)bt * (k + 4), (int64_t)bt * (k + 4)) : total;
      const int64_t want = std::max(total - (int64_t)F * m, first_ok);
      e = k > 0 ? std::max(endk[k - 1], want) : want;
    }
    endk[k] = e;
    fd[k] = (int)(total - e);
    P->fill_workers[k] = (int)std::min<int64_t>(F, total - e);
    for (int i = lo; i < nt; ++i)
      for (int j = i; j < nt; ++j) {
        const int64_t t = tix(i, j);
        int cls = 3;
        if (i < lo + bt) cls = j < lo + bt ? 0 : (j < lo + 2 * bt ? 1 : 2);
        if (t >= e) {
          fill[(size_t)k * Fs + (size_t)((t - e) % F)].add(3 * k + 1, i - lo, j - lo, cid(k, C_SA), sa[k], -1, 0, cid(k, C_FD), -1);
          continue;
        }
        const int inc = cls == 0 ? cid(k + 1, C_G1D) : cls == 1 ? cid(k + 1, C_HR) : cls == 2 ? cid(k + 1, C_RR) : -1;
        const bool handed_back = k > 0 && t >= endk[k - 1] && fd[k - 1] > 0;
        Proto pr{3 * k + 1, i - lo, j - lo, cid(k, cls == 0 ? C_SH : C_SA), cls == 0 ? sh[k] : sa[k],
                 handed_back ? cid(k - 1, C_FD) : -1, handed_back ? fd[k - 1] : 0, inc, -1, i};
        (cls < 3 ? pLA[k] : pBulk[k])[(size_t)(t % W)].push_back(pr);  // (row-major enumeration: already sorted by t; LA by class below)
      }
    for (int w = 0; w < W; ++w)
      std::stable_sort(pLA[k][w].begin(), pLA[k][w].end(), [&](const Proto& a, const Proto& b) {
        auto cls = [&](const Proto& x) { return x.tn + lo < lo + bt ? 0 : (x.tn + lo < lo + 2 * bt ? 1 : 2); };
        return cls(a) < cls(b);
      });
    // CP(k): the factor's block row into place, dealt round-robin
    for (int c = lo; c < nt; ++c) {
      pCP[k][next_c].push_back({3 * k + 2, 0, c - lo, cid(k, C_SA), sa[k], -1, 0, -1, -1, 0});
      next_c = (next_c + 1) % W;
    }
  }

  std::vector<Builder> main(W);
  auto put = [&](Builder& b, const Proto& p) { b.add(p.group, p.tm, p.tn, p.w0, p.v0, p.w1, p.v1, p.i0, p.i1); };
  for (int w = 0; w < W; ++w) {
    Builder& b = main[w];
    for (auto& p : pS[0][w]) put(b, p);
    for (auto& p : pLA[0][w]) put(b, p);
    for (int k = 0; k < K; ++k) {
      const auto& bulk = pBulk[k][w];
      const size_t len = bulk.size();
      size_t n_a0 = 0;  // bulk tiles of block row k+2: the look-ahead tasks on the same tiles come after them
      while (n_a0 < len && bulk[n_a0].row < bt * (k + 2) + bt) ++n_a0;
      size_t ps = std::min<size_t>(len, (size_t)(k == 0 ? tune.solve_pos : tune.solve_pos_later));
      size_t pl = std::max<size_t>(std::max(n_a0, ps), (size_t)(len * tune.la_frac));
      pl = std::min(pl, len);
      // (la_frac2 > la_frac: the look-ahead tasks OUTSIDE the next diagonal block — they wait for ALL of block row k+1's solves,
      //  the diagonal block's only for the head's — run later in the phase than those inside it, which release the panel)
      size_t pl2 = std::max<size_t>(pl, (size_t)(len * tune.la_frac2));
      pl2 = std::min(pl2, len);
      size_t q = 0;
      for (; q < ps; ++q) put(b, bulk[q]);
      if (k + 1 < K)
        for (auto& p : pS[k + 1][w]) put(b, p);
      for (; q < pl; ++q) put(b, bulk[q]);
      if (k + 1 < K)
        for (auto& p : pLA[k + 1][w])
          if (p.i0 == cid(k + 2, C_G1D)) put(b, p);
      for (; q < pl2; ++q) put(b, bulk[q]);
      if (k + 1 < K)
        for (auto& p : pLA[k + 1][w])
          if (p.i0 != cid(k + 2, C_G1D)) put(b, p);
      for (; q < len; ++q) put(b, bulk[q]);
      for (auto& p : pCP[k][w]) put(b, p);
    }
  }

  // the panel stream: diagonal blocks 0 .. K (block K is where the launch-per-product steps take over), each behind its gate,
  // each but the last followed by its signal; the filler launch of step k follows diagonal block k+2, the last ones close the stream
  for (int b = 0; b <= K; ++b) {
    if (b > 0) P->stream_ops.push_back({0, b});
    P->stream_ops.push_back({1, b});
    if (b < K) P->stream_ops.push_back({2, b});
    if (b >= 2 && P->fill_workers[b - 2] > 0) P->stream_ops.push_back({3, b - 2});
  }
  if (K >= 1 && P->fill_workers[K - 1] > 0) P->stream_ops.push_back({3, K - 1});

  // flatten
  P->offsets.assign((size_t)W + (size_t)K * Fs, 0);
  ExecTask endt{};
  endt.group = GPP_EXEC_END;
  endt.wait_id[0] = endt.wait_id[1] = endt.inc_id[0] = endt.inc_id[1] = -1;
  for (int w = 0; w < W; ++w) {
    P->offsets[w] = (int32_t)P->tasks.size();
    P->tasks.insert(P->tasks.end(), main[w].list.begin(), main[w].list.end());
    P->tasks.push_back(endt);
  }
  for (int k = 0; k < K; ++k)
    for (int f = 0; f < Fs; ++f) {
      P->offsets[(size_t)W + (size_t)k * Fs + f] = (int32_t)P->tasks.size();
      auto& l = fill[(size_t)k * Fs + f].list;
      P->tasks.insert(P->tasks.end(), l.begin(), l.end());
      P->tasks.push_back(endt);
    }
  if (P->tasks.size() >= ((size_t)1 << 31)) {
    delete P;
    return nullptr;
  }
  if (getenv("GPP_EXEC_VERBOSE")) {
    fprintf(stderr, "libgpp_hip: exec plan N=%lld nb=%lld K=%d W=%d F=%d: %zu tasks, filler tiles per step:", (long long)N, (long long)nb, K, W, F,
            P->tasks.size());
    for (int k = 0; k < K; ++k) fprintf(stderr, " %d", fd[k]);
    fprintf(stderr, "\n");
  }
  return P;
}

// ONE trailing update as a wait-free task list (debug / profiling: the executor's kernel can then run alone, e.g. under a counter
// collection that serialises dispatches): step 0's update tiles dealt cyclically to W workers, nothing else.
PotrfExecPlan* gpp_plan_single_update(int64_t N, int64_t nb, int W) {
  if (nb % GPP_TILE != 0 || W < 1 || nb >= N) return nullptr;
  const int bt = (int)(nb / GPP_TILE), nt = (int)((N + GPP_TILE - 1) / GPP_TILE);
  PotrfExecPlan* P = new PotrfExecPlan();
  P->N = N; P->nb = nb; P->K = 1; P->W = W; P->F = 0;
  P->ncounters = 1 + 8 * 3;
  P->gate_target.assign(3, 0);
  P->fill_workers.assign(1, 0);
  std::vector<Builder> main(W);
  int64_t t = 0;
  for (int i = bt; i < nt; ++i)
    for (int j = i; j < nt; ++j, ++t) main[(size_t)(t % W)].add(1, i - bt, j - bt, -1, 0, -1, 0, -1, -1);
  P->offsets.assign((size_t)W + 1, 0);
  ExecTask endt{};
  endt.group = GPP_EXEC_END;
  endt.wait_id[0] = endt.wait_id[1] = endt.inc_id[0] = endt.inc_id[1] = -1;
  for (int w = 0; w < W; ++w) {
    P->offsets[w] = (int32_t)P->tasks.size();
    P->tasks.insert(P->tasks.end(), main[w].list.begin(), main[w].list.end());
    P->tasks.push_back(endt);
  }
  P->offsets[W] = (int32_t)P->tasks.size();
  P->tasks.push_back(endt);
  return P;
}

// Device copies of the plan (synchronous; once per plan, and again when the operands' addresses change).
hipError_t gpp_plan_upload(PotrfExecPlan* P) {
  hipError_t e;
  if (!P->d_tasks) {
    if ((e = hipMalloc(&P->d_tasks, P->tasks.size() * sizeof(ExecTask))) != hipSuccess) return e;
    if ((e = hipMemcpy(P->d_tasks, P->tasks.data(), P->tasks.size() * sizeof(ExecTask), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(&P->d_offsets, P->offsets.size() * sizeof(int32_t))) != hipSuccess) return e;
    if ((e = hipMemcpy(P->d_offsets, P->offsets.data(), P->offsets.size() * sizeof(int32_t), hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMalloc(&P->d_counters, (size_t)P->ncounters * sizeof(int))) != hipSuccess) return e;
    if ((e = hipMalloc(&P->d_groups, P->groups.size() * sizeof(GemmArgs))) != hipSuccess) return e;
  }
  return hipMemcpy(P->d_groups, P->groups.data(), P->groups.size() * sizeof(GemmArgs), hipMemcpyHostToDevice);
}

// ---- host-side verification of a plan (tests/test_host_cpu.py) ---------------------------------------------------------------------
// Executes the lists on the host in a RANDOM interleaving that respects only what the device respects — list order per worker, the
// counters, stream order on the panel stream (gate, panel, signal, filler launch) — and checks what the arithmetic needs: every tile
// receives the updates of steps 0, 1, 2, ... in order and exactly once, a solve reads fully updated and not yet overwritten tiles
// of a factored block row, an update reads completely solved strips, a diagonal block is fully updated when its panel starts, and
// everything is complete at the end.  Returns 0, or a positive code naming the first violation (stats[0..3]: tasks run, waits
// carried, increments, largest number of tiles a filler work-group runs in one step).
// `mutate` > 0 removes the mutate-th wait of the plan first: the check must then FAIL (the test of the test).
extern "C" int gpp_debug_plan_check(int64_t N, int64_t nb, int K, int W, int F, int fill, int solve_pos, unsigned seed, int64_t* stats,
                                    int mutate) {
  PotrfExecTuning tune{275.0, 600.0, solve_pos, 0, 0.15, 0.5, fill};
  PotrfExecPlan* P = gpp_plan_potrf_exec(N, nb, K, W, F, tune);
  if (!P) return 1;
  int mutated = -1;  // what was removed: 100 * task kind + 10 * (filler list) + counter kind
  if (mutate > 0) {
    int seen = 0;
    for (auto& t : P->tasks)
      for (int q = 0; q < 2 && mutate > 0; ++q)
        if (t.group >= 0 && t.wait_id[q] >= 0 && ++seen == mutate) {
          mutated = 100 * (t.group % 3) + (t.wait_id[q] - 1) % 8 + 10 * ((&t - P->tasks.data()) >= P->offsets[W] ? 1 : 0);
          t.wait_id[q] = -1;
          mutate = 0;
        }
  }
  gpp_plan_bind(P, nullptr, 0, nullptr, 0, nullptr, 0);
  const int bt = (int)(nb / GPP_TILE), nt = (int)((N + GPP_TILE - 1) / GPP_TILE), Fs = std::max(F, 1);
  auto tix = [&](int64_t i, int64_t j) -> int64_t { return i * nt - i * (i - 1) / 2 + (j - i); };
  std::vector<int> counters(P->ncounters, 0), version((size_t)nt * (nt + 1) / 2, 0);
  std::vector<char> solved((size_t)K * bt * nt, 0), copied((size_t)K * nt, 0), panel_done(K + 2, 0);
  uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  int rc = 0;
  int64_t ran = 0, waits = 0, incs = 0, fill_max = 0;
  // one task; returns false when its waits are not satisfied
  auto try_task = [&](const ExecTask& t) -> bool {
    for (int q = 0; q < 2; ++q)
      if (t.wait_id[q] >= 0 && counters[t.wait_id[q]] < t.wait_val[q]) return false;
    const int k = t.group / 3, kind = t.group % 3, lo = bt * (k + 1);
    if (kind == 0) {  // S(k; r, c)
      const int r = t.tm, c = t.tn + lo;
      if (!panel_done[k]) rc = rc ? rc : 10;
      if (copied[(size_t)k * nt + c]) rc = rc ? rc : 11;
      for (int rr = 0; rr <= r; ++rr)
        if (version[tix(bt * k + rr, c)] != k) rc = rc ? rc : 12;
      if (solved[((size_t)k * bt + r) * nt + c]) rc = rc ? rc : 13;
      solved[((size_t)k * bt + r) * nt + c] = 1;
    } else if (kind == 1) {  // U(k; i, j)
      const int i = t.tm + lo, j = t.tn + lo;
      if (i > j || j >= nt) rc = rc ? rc : 20;
      for (int r = 0; r < bt; ++r)
        if (!solved[((size_t)k * bt + r) * nt + i] || !solved[((size_t)k * bt + r) * nt + j]) rc = rc ? rc : 21;
      if (version[tix(i, j)] != k) rc = rc ? rc : 22;
      version[tix(i, j)] = k + 1;
    } else {  // CP(k; c)
      const int c = t.tn + lo;
      for (int r = 0; r < bt; ++r)
        if (!solved[((size_t)k * bt + r) * nt + c]) rc = rc ? rc : 30;
      if (copied[(size_t)k * nt + c]) rc = rc ? rc : 31;
      copied[(size_t)k * nt + c] = 1;
    }
    for (int q = 0; q < 2; ++q) {
      if (t.wait_id[q] >= 0) ++waits;
      if (t.inc_id[q] >= 0) {
        ++counters[t.inc_id[q]];
        ++incs;
      }
    }
    ++ran;
    return true;
  };
  std::vector<int64_t> pos(W);  // main workers' positions
  for (int w = 0; w < W; ++w) pos[w] = P->offsets[w];
  // panel stream: the plan's launches in order; a filler launch is a set of positions that all have to reach their end
  size_t op = 0;
  bool in_fill = false;
  std::vector<int64_t> fpos;
  bool stream_done = P->stream_ops.empty();
  auto stream_step = [&]() -> bool {  // true when the stream made progress
    if (stream_done) return false;
    const PotrfExecPlan::Op o = P->stream_ops[op];
    auto next = [&]() {
      if (++op == P->stream_ops.size()) stream_done = true;
      return true;
    };
    if (o.kind == 0) {
      if (counters[gpp_plan_counter(o.arg, 1)] < P->gate_target[o.arg]) return false;
      return next();
    }
    if (o.kind == 1) {
      const int pb = o.arg;
      for (int i = bt * pb; i < std::min(bt * (pb + 1), nt); ++i)
        for (int j = i; j < std::min(bt * (pb + 1), nt); ++j)
          if (version[tix(i, j)] != pb) rc = rc ? rc : 40;
      panel_done[pb] = 1;
      return next();
    }
    if (o.kind == 2) {
      if (!panel_done[o.arg]) rc = rc ? rc : 41;
      ++counters[gpp_plan_counter(o.arg, 0)];
      return next();
    }
    if (!in_fill) {
      in_fill = true;
      fpos.clear();
      for (int f = 0; f < P->fill_workers[o.arg]; ++f) fpos.push_back(P->offsets[(size_t)W + (size_t)o.arg * Fs + f]);
      for (auto q : fpos) {
        int64_t n = 0;
        while (P->tasks[q + n].group >= 0) ++n;
        fill_max = std::max(fill_max, n);
      }
    }
    // filler launch: advance a random work-group that can run
    bool any_left = false, progressed = false;
    const size_t nf = fpos.size(), s0 = nf ? (size_t)(rnd() % nf) : 0;
    for (size_t q = 0; q < nf; ++q) {
      int64_t& fp = fpos[(s0 + q) % nf];
      if (P->tasks[fp].group < 0) continue;
      any_left = true;
      if (!progressed && try_task(P->tasks[fp])) {
        ++fp;
        progressed = true;
      }
    }
    if (!any_left) {
      in_fill = false;
      return next();
    }
    return progressed;
  };
  // Interleavings by seed & 3: 0 random; 1 random with a LAZY panel stream (it only moves when no worker can: a solve that does not
  // wait for its panel runs too early); 2 / 3 lazy stream and a fixed worker priority, descending / ascending (the other end of the
  // grid lags as far behind as the counters allow).
  const int mode = (int)(seed & 3u);
  int64_t idle_rounds = 0;
  for (;;) {
    bool all_done = stream_done;
    bool progressed = false;
    if (mode == 0 && rnd() % 8 == 0) progressed = stream_step();
    // bursts of 1-3 tasks mostly, now and then a worker runs as far ahead as its waits allow (what exposes a missing wait)
    const int w0 = mode >= 2 ? 0 : (int)(rnd() % W), burst = (mode >= 2 || rnd() % 16 == 0) ? (1 << 30) : 1 + (int)(rnd() % 3);
    for (int q = 0; q < W; ++q) {
      const int w = mode == 2 ? W - 1 - q : (w0 + q) % W;
      if (P->tasks[pos[w]].group < 0) continue;
      all_done = false;
      for (int b = 0; b < burst && P->tasks[pos[w]].group >= 0 && try_task(P->tasks[pos[w]]); ++b) {
        ++pos[w];
        progressed = true;
      }
      if (progressed) break;
    }
    if (!progressed) progressed = stream_step();
    if (all_done && stream_done) break;
    if (!progressed) {
      if (++idle_rounds > 4) {
        rc = rc ? rc : 2;  // deadlock
        break;
      }
    } else {
      idle_rounds = 0;
    }
    if (rc) break;
  }
  if (!rc) {
    for (int i = bt * K; i < nt && !rc; ++i)
      for (int j = i; j < nt; ++j)
        if (version[tix(i, j)] != K) { rc = 50; break; }
    for (int k = 0; k < K && !rc; ++k)
      for (int c = bt * (k + 1); c < nt; ++c)
        if (!copied[(size_t)k * nt + c]) { rc = 51; break; }
    if (!rc && !panel_done[K]) rc = 52;
  }
  if (stats) {
    stats[0] = ran;
    stats[1] = waits;
    stats[2] = incs;
    stats[3] = mutated >= 0 ? mutated : fill_max;
  }
  gpp_plan_free(P);
  return rc;
}
