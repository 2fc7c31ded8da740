// gpp_dag.hip — host-side planner of the DAG executor (gpp_dag_f64, gpp_gemm.hip): the blocked Cholesky factorisation and,
// beside it, the triangular inverse built right-looking, as ONE list of tile tasks in a topological order of the whole task graph.
// No reference counterpart: the reference's factorisation and its backward are torch.linalg.cholesky_ex / ATen cholesky_backward
// behind gpytorch (call sites optim/mll_torch.py:116-117).
//
// Storage (gpp.h): A keeps the upper factor U (A = U^T U), Linv keeps X = L^-1 lower and its mirror upper, T is an N x N scratch.
// Blocks of nb rows (a multiple of 128; block b = tiles [tb[b], tb[b+1]) of 128).  Per block k, with "k rows" = the rows of block k:
//   P(k)          panel: factor AND invert the diagonal block (a cooperative launch on the panel stream, gpp_leaf.hip)
//   S(k; r, c)    row solve      T[k rows, c] = W_kk^T A[k rows, c]              c right of block k   (K <= nb, triangular)
//   U(k; i, j)    update         A[i, j] -= T[k rows, i]^T T[k rows, j]           i <= j right of block k   (K = nb)
//   CP(k; c)      copy           A[k rows, c] = T[k rows, c]                      (the factor's block row into place)
//   XB(k; i, j)   inverse, sums  T[i, j] (+)= T[k rows, i]^T X[k rows, j]         i below block k, j up to block k   (K = nb)
//   XA(m; r, j)   inverse, rows  X[m rows, j] = -X_mm T[m rows, j] (+ mirror)     j left of block m   (K <= nb, triangular)
// (XB / XA: with S_m = sum_{k<m} L[m,k] X[k,:] accumulated in the lower-left part of T, X[m,:m) = -X_mm S_m — the sharded forward
// sweep's recurrence; the solved block rows live in the upper-right part of T, so the two uses of the scratch never meet.)
// (Above GPP_DAG_INV_MAX rows only the leading inv_rows x inv_rows block of X is built here, gpp_trtri merges the rest around it.)
// FUSED STEPS: a tile at least f + 1 block rows below an aligned group of f steps takes the group's f updates (or the sum's f
// contributions) in ONE task at the group's last step, with K = the f blocks' rows — see generate().
// Every dependency is a monotone counter: strip counters (all tasks of a strip done) and per-tile version counters (the k-th
// contribution to a tile follows the (k-1)-th; a fused task raises its tile's version by f).  The ORDER of the list is that of a list-scheduling simulation (priority = longest
// path to the end of the graph, cost model in DagTuning): with it a work-group that takes the next ticket finds its task ready or
// nearly so, chain tasks (head solve, next diagonal block's update) are taken the moment they can run, and the inverse's tasks fill
// whatever the factorisation's chain leaves idle.  Any topological order is CORRECT; the simulation only decides how good it is.
#include "gpp_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include <vector>

namespace {

constexpr int ALL = -7;  // wait value: "every task that increments this counter"

struct Node {
  int group = -1, tm = 0, tn = 0, kind = 0;  // kind < 0: panel of block `tm`
  int lvl = 0;                               // the last block whose panel the task (transitively) needs
  int w[3] = {-1, -1, -1}, v[3] = {0, 0, 0};
  int inc[2] = {-1, -1};
  int incv[2] = {0, 0};  // amounts (a fused task raises its tile's version by the number of steps it applies)
  float cost = 0;
  double bl = 0;  // bottom level
};

struct Planner {
  int64_t N, nb, ld, ldi, ldt, ldk;
  int flags;
  DagTuning tune;
  int nt = 0, B = 0;
  int inv_tiles = 0;                       // the inverse is built for the tile rows below this (block aligned)
  std::vector<int> tb;
  std::vector<GemmArgs> groups;
  std::vector<std::pair<int, int>> gmeta;  // per group: (kind, block)
  std::vector<Node> nodes;
  int ncounters = 2;                      // 0 abort, 1 ticket
  std::vector<char> chain;                // per counter: version counter (the v-th increment follows the (v-1)-th)
  std::vector<std::vector<int>> incs;     // per counter: incrementing nodes in generation order
  int c_pd = 0, c_g1d = 0, c_ur = 0, c_ss = 0, c_xr = 0, c_xs = 0, c_va = 0, c_vt = 0;
  int cur_lvl = 0;
  // DAG_SHARD: this rank's list of the sharded evaluation
  int rank = 0, nranks = 1;
  int c_cpd = 0, c_cph = 0, c_cpt = 0, c_art = 0, c_own = 0;
  std::vector<int> cph_n, cpt_n;  // copies that raise CPH(k) / CPT(k, g)
  int piece_tiles = 0, GP = 1;    // the tail of a block row's message in pieces of piece_tiles column tiles (DagPlan::piece_tiles)
  int hi_of(int k) const { return tb[std::min(k + 2, B)]; }  // first column tile of block row k's tail
  int npieces(int k) const { return hi_of(k) >= nt ? 0 : (nt - hi_of(k) + piece_tiles - 1) / piece_tiles; }
  int piece_of(int k, int c) const { return (c - hi_of(k)) / piece_tiles; }  // c >= hi_of(k)
  bool own(int b) const { return b % nranks == rank; }
  int c_zs = 0, c_yr = 0, c_vy = 0;  // DAG_BACK
  int ZS(int b, int c) const { return c_zs + b * nt + c; }
  int YR(int b, int c) const { return c_yr + b * nt + c; }
  int VY(int i, int j) const { return c_vy + (int)((int64_t)i * (i + 1) / 2 + j); }  // i >= j
  int CPD(int b, int c) const { return c_cpd + b * nt + c; }
  int CPH(int b) const { return c_cph + b; }
  int CPT(int b, int g) const { return c_cpt + b * GP + g; }
  int ART(int b, int g) const { return c_art + b * GP + g; }
  int OWNM(int b, int m) const { return c_own + b * (GP + 1) + m; }  // this rank's own message m (0 head, 1 + g piece g) is on the wire:
                                                                     // raised by the simulation's communication stream only

  int blk_of(int tile) const { return (int)(std::upper_bound(tb.begin(), tb.end(), tile) - tb.begin()) - 1; }
  int64_t rows_of(int b) const { return std::min<int64_t>((int64_t)tb[b + 1] * 128, N) - (int64_t)tb[b] * 128; }
  int PD(int b) const { return c_pd + b; }
  int G1D(int b) const { return c_g1d + b; }
  int UR(int b, int c) const { return c_ur + b * nt + c; }
  int SS(int b, int c) const { return c_ss + b * nt + c; }
  int XR(int b, int c) const { return c_xr + b * nt + c; }
  int XS(int b, int c) const { return c_xs + b * nt + c; }
  int VA(int i, int j) const { return c_va + (int)((int64_t)i * nt - (int64_t)i * (i - 1) / 2 + (j - i)); }  // i <= j
  int VT(int i, int j) const { return c_vt + (int)((int64_t)i * (i - 1) / 2 + j); }                          // i > j

  static const double* off(int64_t elems) { return reinterpret_cast<const double*>(static_cast<uintptr_t>(elems * 8)); }
  static double* offw(int64_t elems) { return reinterpret_cast<double*>(static_cast<uintptr_t>(elems * 8)); }

  int add_group(const GemmArgs& g, int kind, int b) {
    cur_lvl = b;
    groups.push_back(g);
    gmeta.emplace_back(kind, b);
    return (int)groups.size() - 1;
  }
  double tile_cost(int etile, int64_t K) const {
    const double ch = (double)((K + 15) / 16);
    if (etile == 64) return tune.t0_64 + tune.tc_64 * ch;
    if (etile == 32) return tune.t0_32 + tune.tc_32 * ch;
    return tune.t0_big + tune.tc_big * ch;
  }
  int add_node(int group, int tm, int tn, int kind, double cost) {
    Node n;
    n.group = group; n.tm = tm; n.tn = tn; n.kind = kind; n.cost = (float)cost;
    n.lvl = cur_lvl;  // (every task of a group of block b needs the panel of block b and none after it)
    nodes.push_back(n);
    return (int)nodes.size() - 1;
  }
  void wait(int node, int c, int v) {
    if (c < 0 || v == 0) return;
    Node& n = nodes[node];
    for (int q = 0; q < 3; ++q)
      if (n.w[q] < 0) {
        n.w[q] = c;
        n.v[q] = v;
        return;
      }
    fprintf(stderr, "libgpp_hip: dag planner: more than three waits on a task\n");
    abort();
  }
  void inc(int node, int c, int amount = 1) {
    if (c < 0) return;
    Node& n = nodes[node];
    for (int q = 0; q < 2; ++q)
      if (n.inc[q] < 0) {
        n.inc[q] = c;
        n.incv[q] = amount;
        for (int a = 0; a < amount; ++a) incs[c].push_back(node);  // (one entry per unit: "the v-th increment" stays meaningful)
        return;
      }
    fprintf(stderr, "libgpp_hip: dag planner: more than two increments on a task\n");
    abort();
  }

  bool layout() {
    if (nb % GPP_TILE != 0 || nb < GPP_TILE || N < 2 * GPP_TILE) return false;
    nt = (int)((N + GPP_TILE - 1) / GPP_TILE);
    if (nt >= 32000) return false;
    const int bt = (int)(nb / GPP_TILE);
    tb.clear();
    // (measured and dropped: a shorter first block — nothing runs beside the first panel — 13.60 vs 13.45 ms at N = 10 000 with 512 rows)
    for (int t = 0; t < nt; t += bt) tb.push_back(t);
    // a last block too short for the cooperative panel (<= 256 rows) joins its neighbour
    if (tb.size() >= 2 && N - (int64_t)tb.back() * 128 <= 256) tb.pop_back();
    tb.push_back(nt);
    B = (int)tb.size() - 1;
    if (B < 2) return false;
    inv_tiles = 0;
    if (flags & DAG_INV) {
      inv_tiles = nt;
      if (tune.inv_rows > 0 && tune.inv_rows < N) {
        inv_tiles = (int)(tune.inv_rows / GPP_TILE);
        bool aligned = false;
        for (int b = 1; b <= B; ++b) aligned |= (tb[b] == inv_tiles);
        if (!aligned || tune.inv_rows % GPP_TILE != 0) return false;
      }
    }
    if (flags & DAG_BACK) {
      if (!(flags & DAG_SHARD) || B != (nt + bt - 1) / bt || rank < 0 || rank >= nranks || rank >= B) return false;
      c_zs = ncounters; ncounters += B * nt;
      c_yr = ncounters; ncounters += B * nt;
      c_vy = ncounters; ncounters += (int)((int64_t)nt * (nt + 1) / 2);
      chain.assign(ncounters, 0);
      for (int c = c_vy; c < ncounters; ++c) chain[c] = 1;
      incs.assign(ncounters, {});
      return true;
    }
    c_pd = ncounters; ncounters += B;
    c_g1d = ncounters; ncounters += B;
    c_ur = ncounters; ncounters += B * nt;
    c_ss = ncounters; ncounters += B * nt;
    if (flags & DAG_INV) {
      c_xr = ncounters; ncounters += B * nt;
      c_xs = ncounters; ncounters += B * nt;
    }
    c_va = ncounters; ncounters += (int)((int64_t)nt * (nt + 1) / 2);
    if (flags & DAG_INV) {
      c_vt = ncounters; ncounters += (int)((int64_t)nt * (nt - 1) / 2);
    }
    const int n_ver = ncounters;
    if (flags & DAG_SHARD) {
      // (uniform blocks: the sharded buffers address block k at k * nb; a last block too short for the panel is the caller's to avoid)
      if (B != (nt + bt - 1) / bt || rank < 0 || rank >= nranks || rank >= B) return false;
      piece_tiles = tune.piece_cols >= GPP_TILE ? (int)(tune.piece_cols / GPP_TILE) : nt;
      GP = std::max(1, npieces(0));
      c_cpd = ncounters; ncounters += B * nt;
      c_cph = ncounters; ncounters += B;
      c_cpt = ncounters; ncounters += B * GP;
      c_art = ncounters; ncounters += B * GP;
      c_own = ncounters; ncounters += B * (GP + 1);
      cph_n.assign(B, 0);
      cpt_n.assign((size_t)B * GP, 0);
    }
    chain.assign(ncounters, 0);
    for (int c = c_va; c < n_ver; ++c) chain[c] = 1;  // VA, VT
    incs.assign(ncounters, {});
    return true;
  }

  void generate() {
    if (flags & DAG_BACK) {
      generate_back();
      return;
    }
    if (flags & DAG_SHARD) {
      generate_sharded();
      return;
    }
    const int ct = tune.chain_tile == 64 ? 64 : 128;  // (a 32 x 32 body beside the two others made the kernel spill 26 VGPRs)
    for (int k = 0; k < B; ++k) {
      const int lo = tb[k + 1], btk = tb[k + 1] - tb[k];
      const int64_t o = (int64_t)tb[k] * 128, nbk = rows_of(k), c0 = (int64_t)lo * 128, rem = N - c0;
      // the panel of block k
      {
        const int p = add_node(-1, k, 0, -1, tune.t_gate + tune.t_panel0 + tune.t_panel_leaf * (double)((nbk + 127) / 128));
        if (k > 0) wait(p, G1D(k), ALL);
        inc(p, PD(k));
      }
      if ((flags & DAG_INV) && k > 0 && tb[k + 1] <= inv_tiles) {
        // XA(k; r, j): X[k rows, j] = -W_kk^T T[k rows, j] (+ mirror), j left of block k
        GemmArgs g{};
        g.A = off(o * ldi + o); g.lda = ldi; g.buf[0] = 1;
        g.B = off(o * ldt); g.ldb = ldt; g.buf[1] = 2;
        g.C = offw(o * ldi); g.ldc = ldi; g.buf[2] = 1;
        g.C2 = offw(o); g.ldc2 = ldi; g.buf[3] = 1;
        g.M = (int)nbk; g.N = (int)o; g.K = (int)nbk;
        g.alpha = -1.0; g.beta = 0.0;
        g.a_mask = 1; g.khi_mode = 1;
        g.pad_ok = (k < B - 1) ? 1 : 0;  // (the last block's rows end the buffer)
        const int gi = add_group(g, DK_XA, k);
        for (int r = btk - 1; r >= 0; --r)
          for (int j = 0; j < tb[k]; ++j) {
            const int n = add_node(gi, r, j, DK_XA, tile_cost(128, std::min<int64_t>(nbk, (int64_t)(r + 1) * 128)));
            wait(n, PD(k), 1);
            wait(n, XR(k, j), ALL);
            inc(n, XS(k, j));
          }
      }
      if (rem <= 0) continue;
      const int nbk1 = (int)rows_of(k + 1), hi = tb[k + 2];  // next block: rows, end tile
      // ---- S(k): row solve ----------------------------------------------------------------------------------------------------
      GemmArgs s{};
      s.A = off(o * ldi + o); s.lda = ldi; s.buf[0] = 1;
      s.B = off(o * ld + c0); s.ldb = ld; s.buf[1] = 0;
      s.C = offw(o * ldt + c0); s.ldc = ldt; s.buf[2] = 2;
      s.buf[3] = -1;
      s.M = (int)nbk; s.N = (int)rem; s.K = (int)nbk;
      s.alpha = 1.0; s.beta = 0.0;
      s.a_mask = 1; s.khi_mode = 1;
      s.pad_ok = 1;  // (rows of block k with further rows below them: reading past column N stays inside the buffers)
      const int gS = add_group(s, DK_S, k);
      int gSh = -1;
      if (ct != 128) {
        GemmArgs sh = s;
        sh.N = nbk1;
        sh.etile = ct;
        gSh = add_group(sh, DK_SH, k);
      }
      for (int pass = 0; pass < 2; ++pass) {  // head columns (those of diagonal block k+1) first
        if (pass == 0 && gSh >= 0) {
          const int rt = (int)((nbk + ct - 1) / ct), ctiles = (nbk1 + ct - 1) / ct;
          for (int r = rt - 1; r >= 0; --r)
            for (int c = 0; c < ctiles; ++c) {
              const int c128 = lo + c * ct / 128;
              const int n = add_node(gSh, r, c, DK_SH, tile_cost(ct, std::min<int64_t>(nbk, (int64_t)(r + 1) * ct)));
              wait(n, PD(k), 1);
              if (k > 0) wait(n, UR(k, c128), ALL);
              inc(n, SS(k, c128));
            }
          continue;
        }
        for (int r = btk - 1; r >= 0; --r)
          for (int c = lo; c < nt; ++c) {
            const bool head = c < hi;
            if (head != (pass == 0)) continue;
            const int n = add_node(gS, r, c - lo, DK_S, tile_cost(128, std::min<int64_t>(nbk, (int64_t)(r + 1) * 128)));
            wait(n, PD(k), 1);
            if (k > 0) wait(n, UR(k, c), ALL);
            inc(n, SS(k, c));
          }
      }
      // ---- U(k): trailing update ----------------------------------------------------------------------------------------------
      GemmArgs u{};
      u.A = off(o * ldt + c0); u.lda = ldt; u.buf[0] = 2;
      u.B = u.A; u.ldb = ldt; u.buf[1] = 2;
      u.C = offw(c0 * ld + c0); u.ldc = ld; u.buf[2] = 0;
      u.buf[3] = -1;
      u.M = u.N = (int)rem; u.K = (int)nbk;
      u.alpha = -1.0; u.beta = 1.0;
      u.c_lower = 2;
      u.pad_ok = 1;
      const int gU = add_group(u, DK_U, k);
      int gUd = -1;
      if (ct != 128) {
        GemmArgs ud = u;
        ud.M = ud.N = nbk1;
        ud.etile = ct;
        gUd = add_group(ud, DK_UD, k);
        const int rt = (nbk1 + ct - 1) / ct;
        for (int a = 0; a < rt; ++a)
          for (int b = a; b < rt; ++b) {
            const int i = lo + a * ct / 128, j = lo + b * ct / 128;
            const int n = add_node(gUd, a, b, DK_UD, tile_cost(ct, nbk));
            if (k > 0) wait(n, VA(i, j), k);
            wait(n, SS(k, i), ALL);
            if (j != i) wait(n, SS(k, j), ALL);
            inc(n, G1D(k + 1));
          }
      }
      // Fused steps (tune.fuse = 2 or 4): a tile far enough below the chain skips the updates of steps k0 .. k0 + f - 2 of an
      // aligned group of f steps and takes all f together at step k0 + f - 1, as ONE task with K = the f blocks' rows (adjacent
      // rows of T): one prologue, one read-modify-write of the tile and one ticket instead of f.  "Far enough": its block row is
      // at least f + 1 below k0, so nothing of the next f steps' chain or look-ahead reads it.  The strips of the earlier steps are
      // complete when those of the last one are (block row k + 1's solves follow its own last update, which followed step k's
      // solves), so the last step's two strip counters and the version suffice — the host checker verifies that from the geometry.
      auto fuse_of = [&](int bi, int kk) {  // how many steps the update of a tile in block row bi is grouped by at step kk
        for (int f = std::min(tune.fuse, 4); f >= 2; f >>= 1) {
          const int g0 = kk - kk % f;
          if (g0 + f - 1 <= B - 2 && bi >= g0 + f + 1) return f;
        }
        return 1;
      };
      int gUf[5] = {-1, -1, -1, -1, -1};  // fused groups ending at this step, by f
      int64_t Kf[5] = {0, 0, 0, 0, 0};
      for (int f = 2; f <= std::min(tune.fuse, 4); f <<= 1) {
        if (k % f != f - 1 || k - (f - 1) < 0) continue;
        const int g0 = k - (f - 1);
        GemmArgs uf = u;
        const int64_t o_first = (int64_t)tb[g0] * 128;
        Kf[f] = (int64_t)tb[k + 1] * 128 - o_first;  // rows of blocks g0 .. k
        uf.A = off(o_first * ldt + c0);
        uf.B = uf.A;
        uf.K = (int)Kf[f];
        gUf[f] = add_group(uf, DK_U, k);
      }
      for (int i = lo; i < nt; ++i) {
        const int f = fuse_of(blk_of(i), k);
        for (int j = i; j < nt; ++j) {
          const bool in_next = i < hi;          // block row k+1: this is the tile's LAST update
          const bool diag = in_next && j < hi;  // inside diagonal block k+1
          if (diag && gUd >= 0) continue;
          if (f > 1) {
            const int g0 = k - k % f;
            if (k != g0 + f - 1) continue;  // taken together with the group's last step
            const int n = add_node(gUf[f], i - lo, j - lo, DK_U, tile_cost(128, Kf[f]));
            if (g0 > 0) wait(n, VA(i, j), g0);
            wait(n, SS(k, i), ALL);
            if (j != i) wait(n, SS(k, j), ALL);
            inc(n, VA(i, j), f);
            continue;
          }
          const int n = add_node(gU, i - lo, j - lo, DK_U, tile_cost(128, nbk));
          if (k > 0) wait(n, VA(i, j), k);
          wait(n, SS(k, i), ALL);
          if (j != i) wait(n, SS(k, j), ALL);
          if (diag) inc(n, G1D(k + 1));
          else if (in_next) inc(n, UR(k + 1, j));
          else inc(n, VA(i, j));
        }
      }
      // ---- CP(k): the factor's block row into place -----------------------------------------------------------------------------
      GemmArgs c{};
      c.A = off(0); c.buf[0] = 2;
      c.B = off(o * ldt + c0); c.ldb = ldt; c.buf[1] = 2;
      c.C = offw(o * ld + c0); c.ldc = ld; c.buf[2] = 0;
      c.buf[3] = -1;
      c.M = (int)nbk; c.N = (int)rem;
      c.op = 1;
      const int gC = add_group(c, DK_CP, k);
      for (int cc = lo; cc < nt; ++cc) {
        const int n = add_node(gC, 0, cc - lo, DK_CP, tune.t_copy);
        wait(n, SS(k, cc), ALL);
      }
      // ---- XB(k): the inverse's running sums ------------------------------------------------------------------------------------
      if ((flags & DAG_INV) && tb[k + 1] < inv_tiles) {
        GemmArgs x0{};  // columns of block k: the first contribution (beta = 0), X_kk lower triangular
        x0.A = off(o * ldt + c0); x0.lda = ldt; x0.buf[0] = 2;
        x0.B = off(o * ldi + o); x0.ldb = ldi; x0.buf[1] = 1;
        x0.C = offw(c0 * ldt + o); x0.ldc = ldt; x0.buf[2] = 2;
        x0.buf[3] = -1;
        x0.M = (int)rem; x0.N = (int)nbk; x0.K = (int)nbk;
        x0.alpha = 1.0; x0.beta = 0.0;
        x0.b_mask = 2; x0.klo_mode = 2;
        x0.pad_ok = 1;
        const int g0 = add_group(x0, DK_XB, k);
        int g1 = -1;
        if (k > 0) {
          GemmArgs x1 = x0;  // columns left of block k: accumulate
          x1.B = off(o * ldi);
          x1.C = offw(c0 * ldt);
          x1.N = (int)o;
          x1.beta = 1.0;
          x1.b_mask = 0; x1.klo_mode = 0;
          g1 = add_group(x1, DK_XB, k);
        }
        // fused steps for the sums as for the updates: a far tile whose column lies at or left of the group's first block takes
        // the group's f contributions at its last step, with K = the f blocks' rows of T and of X (both adjacent); X's rows of the
        // earlier blocks are final when the last one's are (each XA follows the sums the previous step completed)
        int gXf0[5] = {-1, -1, -1, -1, -1}, gXf1[5] = {-1, -1, -1, -1, -1};
        for (int f = 2; f <= std::min(tune.fuse, 4); f <<= 1) {
          if (k % f != f - 1 || k - (f - 1) < 0) continue;
          const int q0 = k - (f - 1);
          const int64_t o_first = (int64_t)tb[q0] * 128;
          GemmArgs y0 = x0;
          y0.A = off(o_first * ldt + c0);
          y0.B = off(o_first * ldi + o_first);
          y0.C = offw(c0 * ldt + o_first);
          y0.N = (int)rows_of(q0);
          y0.K = (int)Kf[f];
          gXf0[f] = add_group(y0, DK_XB, k);
          if (q0 > 0) {
            GemmArgs y1 = y0;
            y1.B = off(o_first * ldi);
            y1.C = offw(c0 * ldt);
            y1.N = (int)o_first;
            y1.beta = 1.0;
            y1.b_mask = 0; y1.klo_mode = 0;
            gXf1[f] = add_group(y1, DK_XB, k);
          }
        }
        for (int i = lo; i < inv_tiles; ++i) {
          const bool fin = i < hi;  // block row k+1: the sums of that block row are complete after this step
          const int fz = fuse_of(blk_of(i), k), q0 = k - k % fz;
          for (int j = 0; j < lo; ++j) {
            const bool own = j >= tb[k];
            const int bj = blk_of(j);
            if (fz > 1 && bj <= q0) {
              if (k != q0 + fz - 1) continue;  // taken together with the group's last step
              const bool own0 = bj == q0;
              const int64_t K = own0 ? Kf[fz] - (int64_t)(j - tb[q0]) * 128 : Kf[fz];
              const int n = add_node(own0 ? gXf0[fz] : gXf1[fz], i - lo, own0 ? j - tb[q0] : j, DK_XB, tile_cost(128, K));
              wait(n, SS(k, i), ALL);
              wait(n, XS(k, j), ALL);
              if (q0 - bj > 0) wait(n, VT(i, j), q0 - bj);
              inc(n, VT(i, j), fz);
              continue;
            }
            const int64_t K = own ? nbk - (int64_t)(j - tb[k]) * 128 : nbk;
            const int n = add_node(own ? g0 : g1, i - lo, own ? j - tb[k] : j, DK_XB, tile_cost(128, K));
            wait(n, SS(k, i), ALL);
            if (own) wait(n, PD(k), 1);
            else wait(n, XS(k, j), ALL);
            if (k - bj > 0) wait(n, VT(i, j), k - bj);
            if (fin) inc(n, XR(k + 1, j));
            else inc(n, VT(i, j));
          }
        }
      }
    }
  }

  // ---- one rank's list of the SHARDED evaluation (gp-plus_amd/sharded.py; DAG_SHARD) ------------------------------------------------
  // 1-D block-cyclic: block row k of the factor AND column block k of X = L^-1 belong to rank k % P.  Buffers (DagBases):
  // 0 A (N x N: the replicated factor, upper), 1 Kc (N x owned columns, compact: X), 2 Lc (the same shape: the forward sweep's running
  // sums), 3 D (per block nb x nb: the diagonal blocks' inverses + mirrors), 4-6 W (three nb x N scratch rows of the row solves).
  // The plan's ldi is the compact buffers' leading dimension, ldt the scratch rows'.
  //   own k:     P(k) on the panel stream (inverse into D[k]);  S(k; r, c) into W[q % 3], q = k's ordinal among the owned blocks;
  //              CP(k; c): the solved strip into A's block row, raising CPD(k, c) and CPH(k) / CPT(k) — the gates of the head / tail
  //              broadcasts, which the caller packs from A on its communication stream
  //   other k:   the caller signals PD(k) behind the unpacked head message (diagonal block, D[k], the columns of block k+1) and
  //              ART(k) behind the tail's
  //   U(k; i, j) for the block rows i this rank owns, reading block row k of the factor from A: available for column tile c at
  //              CPD(k, c) on its owner, PD(k) (head) / ART(k) (tail) elsewhere.  Where one rank owns k and k+1 (P = 1) the chain's
  //              tiles — diagonal block k+1 and the rest of block row k+1 — read the solve's output in W instead (no copy on the
  //              chain); W[.][., c] is rewritten by S(k + 3P; ., c), which follows U(k; k+1, c) through the chain of column c's
  //              strips and waits for CP(k; c) explicitly.
  //   XB(k; i, j) / XA(m; r, j) for the column blocks j this rank owns: the owned blocks left of any block sit side by side at the
  //              start of Kc / Lc, so the products address compact tiles directly; their counters keep the global tile numbers.
  //   Fused steps: the last step's availability implies the earlier ones' on every rank (messages arrive in order; a block row's
  //              solves follow the updates that needed the previous slab), as in the one-GPU list.
  // For the ORDER the remote events are nodes of a serial "communication stream" with estimated costs (kind -2: head, -3: tail);
  // they are not tasks.
  int glob_tile(int jc) const {  // compact tile -> global tile
    const int bt = (int)(nb / GPP_TILE);
    return (rank + (jc / bt) * nranks) * bt + jc % bt;
  }
  int nleft(int k) const { return k > rank ? (k - rank + nranks - 1) / nranks : 0; }  // owned blocks left of block k
  void generate_sharded() {
    const int ct = tune.chain_tile == 64 ? 64 : 128;
    const int P = nranks, me = rank;
    const int bt = (int)(nb / GPP_TILE);
    const int64_t ldc = ldi, ldw = ldt;
    auto avail = [&](int node, int k, int c) {  // block row k of the factor, column tile c, is in A
      if (own(k)) wait(node, CPD(k, c), 1);
      else wait(node, c < hi_of(k) ? PD(k) : ART(k, piece_of(k, c)), 1);
    };
    auto avail2 = [&](int node, int k, int i, int j) {  // column tiles i <= j
      if (own(k)) {
        wait(node, CPD(k, i), 1);
        if (j != i) wait(node, CPD(k, j), 1);
      } else {
        // (the head, then the tail's pieces in column order, arrive on ONE stream: the later tile's message implies the earlier's)
        wait(node, j < hi_of(k) ? PD(k) : ART(k, piece_of(k, j)), 1);
      }
    };
    auto fuse_of = [&](int bi, int kk) {
      for (int f = std::min(tune.fuse, 4); f >= 2; f >>= 1) {
        const int g0 = kk - kk % f;
        if (g0 + f - 1 <= B - 2 && bi >= g0 + f + 1) return f;
      }
      return 1;
    };
    int last_comm = -1;  // the communication stream's previous event
    // Every message of the communication stream is a node of the simulation, in the stream's order: per block row its head, then
    // the pieces of its tail.  Another rank's message raises PD(k) / ART(k, g) when it has arrived; this rank's own occupies the
    // stream behind its gate (the copies of those strips: CPH(k) / CPT(k, g)) for its bytes at ~50 GB/s and raises OWNM, which
    // nothing on the device waits for.
    auto piece_mb = [&](int k, int g) {
      const int t0 = hi_of(k) + g * piece_tiles, t1 = std::min(t0 + piece_tiles, nt);
      return 1e-6 * 8.0 * (double)rows_of(k) * (double)(std::min<int64_t>((int64_t)t1 * 128, N) - (int64_t)t0 * 128);
    };
    auto own_messages = [&](int k) {  // (behind the generation of block row k's copies: their counts are the gates' targets)
      const int64_t o_ = (int64_t)tb[k] * 128, nbk_ = rows_of(k);
      const double mb_head = 1e-6 * 8.0 * (double)(nbk_ * (std::min<int64_t>((int64_t)hi_of(k) * 128, N) - o_ + nbk_));
      const int h = add_node(-1, k, 0, -2, 50.0 + mb_head / 0.05);
      if (last_comm >= 0) wait(h, nodes[last_comm].inc[0], 1);
      if (cph_n[k] > 0) wait(h, CPH(k), ALL);
      else wait(h, PD(k), 1);
      inc(h, OWNM(k, 0));
      last_comm = h;
      for (int g = 0; g < npieces(k); ++g) {
        const int t = add_node(-1, k, g, -3, 50.0 + piece_mb(k, g) / 0.05);
        wait(t, nodes[last_comm].inc[0], 1);
        if (cpt_n[(size_t)k * GP + g] > 0) wait(t, CPT(k, g), ALL);
        inc(t, OWNM(k, 1 + g));
        last_comm = t;
      }
    };
    for (int k = 0; k < B; ++k) {
      const int lo = tb[k + 1], btk = tb[k + 1] - tb[k];
      const int64_t o = (int64_t)tb[k] * 128, nbk = rows_of(k), c0 = (int64_t)lo * 128, rem = N - c0;
      const int hi = hi_of(k);
      const int64_t dk = (int64_t)k * nb * nb;  // D[k]
      const int nl = nleft(k);                  // owned column blocks left of block k: compact tiles [0, nl * bt)
      cur_lvl = k;
      if (own(k)) {
        const int p = add_node(-1, k, 0, -1, tune.t_gate + tune.t_panel0 + tune.t_panel_leaf * (double)((nbk + 127) / 128));
        if (k > 0) wait(p, G1D(k), ALL);
        inc(p, PD(k));
      } else if (nranks > 1) {
        // the head of a remote slab: its owner needed the previous slab's head first
        const double mb_head = 1e-6 * 8.0 * (double)(nbk * (std::min<int64_t>(c0 + nb, N) - o + nbk));
        const int h = add_node(-1, k, 0, -2, 1300.0 + mb_head / 0.05);  // update + panel + head solve + copy, the message at ~50 GB/s
        if (last_comm >= 0) wait(h, nodes[last_comm].inc[0], 1);  // (the stream's order: behind the previous message)
        inc(h, PD(k));
        last_comm = h;
        for (int g = 0; g < npieces(k); ++g) {
          // (its first piece: one piece's update + solve + copies behind the previous block row's first piece; then the wire)
          const int t = add_node(-1, k, g, -3, (g == 0 ? 200.0 : 20.0) + piece_mb(k, g) / 0.05);
          wait(t, nodes[last_comm].inc[0], 1);
          inc(t, ART(k, g));
          last_comm = t;
        }
      }
      // XA(k; r, jc): X[k rows, j] = -W_kk^T S_k[., j] for the owned column blocks left of block k
      if (k > 0 && nl > 0) {
        GemmArgs g{};
        g.A = off(dk); g.lda = nb; g.buf[0] = 3;
        g.B = off(o * ldc); g.ldb = ldc; g.buf[1] = 2;
        g.C = offw(o * ldc); g.ldc = ldc; g.buf[2] = 1;
        g.buf[3] = -1;
        g.M = (int)nbk; g.N = (int)(nl * nb); g.K = (int)nbk;
        g.alpha = -1.0; g.beta = 0.0;
        g.a_mask = 1; g.khi_mode = 1;
        g.pad_ok = 1;  // (D[k] is a full nb x nb block, the compact buffers are whole blocks wide)
        const int gi = add_group(g, DK_XA, k);
        for (int r = btk - 1; r >= 0; --r)
          for (int jc = 0; jc < nl * bt; ++jc) {
            const int j = glob_tile(jc);
            const int n = add_node(gi, r, jc, DK_XA, tile_cost(128, std::min<int64_t>(nbk, (int64_t)(r + 1) * 128)));
            wait(n, PD(k), 1);
            wait(n, XR(k, j), ALL);
            inc(n, XS(k, j));
          }
      }
      if (rem <= 0) {
        if (own(k) && nranks > 1) own_messages(k);
        continue;
      }
      const int nbk1 = (int)rows_of(k + 1);
      const bool chain_here = own(k) && own(k + 1);  // this rank's next panel reads this solve's output directly
      const int slot = 4 + ((k - me) / P) % 3;
      // ---- S(k), CP(k): the owner's row solve into the scratch row, then into place ------------------------------------------------
      if (own(k)) {
        GemmArgs s{};
        s.A = off(dk); s.lda = nb; s.buf[0] = 3;
        s.B = off(o * ld + c0); s.ldb = ld; s.buf[1] = 0;
        s.C = offw(c0); s.ldc = ldw; s.buf[2] = slot;
        s.buf[3] = -1;
        s.M = (int)nbk; s.N = (int)rem; s.K = (int)nbk;
        s.alpha = 1.0; s.beta = 0.0;
        s.a_mask = 1; s.khi_mode = 1;
        s.pad_ok = 1;
        const int gS = add_group(s, DK_S, k);
        int gSh = -1;
        if (ct != 128) {
          GemmArgs sh = s;
          sh.N = nbk1;
          sh.etile = ct;
          gSh = add_group(sh, DK_SH, k);
        }
        auto solve_waits = [&](int n, int c128) {
          wait(n, PD(k), 1);
          if (k > 0) wait(n, UR(k, c128), ALL);
          if (k >= 3 * P) wait(n, CPD(k - 3 * P, c128), 1);  // the scratch row's previous strip has been copied out
          inc(n, SS(k, c128));
        };
        if (gSh >= 0) {
          const int rt = (int)((nbk + ct - 1) / ct), ctiles = (nbk1 + ct - 1) / ct;
          for (int r = rt - 1; r >= 0; --r)
            for (int c = 0; c < ctiles; ++c) {
              const int n = add_node(gSh, r, c, DK_SH, tile_cost(ct, std::min<int64_t>(nbk, (int64_t)(r + 1) * ct)));
              solve_waits(n, lo + c * ct / 128);
            }
        }
        for (int r = btk - 1; r >= 0; --r)
          for (int c = lo; c < nt; ++c) {
            if (gSh >= 0 && c < hi) continue;
            const int n = add_node(gS, r, c - lo, DK_S, tile_cost(128, std::min<int64_t>(nbk, (int64_t)(r + 1) * 128)));
            solve_waits(n, c);
          }
        GemmArgs cp{};
        cp.A = off(0); cp.buf[0] = slot;
        cp.B = off(c0); cp.ldb = ldw; cp.buf[1] = slot;
        cp.C = offw(o * ld + c0); cp.ldc = ld; cp.buf[2] = 0;
        cp.buf[3] = -1;
        cp.M = (int)nbk; cp.N = (int)rem;
        cp.op = 1;
        const int gC = add_group(cp, DK_CP, k);
        for (int c = lo; c < nt; ++c) {
          const int n = add_node(gC, 0, c - lo, DK_CP, tune.t_copy);
          wait(n, SS(k, c), ALL);
          // one rank owning consecutive blocks: a column's strips are copied in step order, so that "block row k is in A" implies
          // the same of the rows above it (fused tasks wait for their last step only); with P > 1 the messages' order does that
          if (k > 0 && own(k - 1)) wait(n, CPD(k - 1, c), 1);
          inc(n, CPD(k, c));
          if (c < hi) {
            inc(n, CPH(k));
            ++cph_n[k];
          } else {
            inc(n, CPT(k, piece_of(k, c)));
            ++cpt_n[(size_t)k * GP + piece_of(k, c)];
          }
        }
        if (nranks > 1) own_messages(k);
      }
      // ---- U(k): the trailing update of the block rows this rank owns -----------------------------------------------------------
      GemmArgs u{};
      u.A = off(o * ld + c0); u.lda = ld; u.buf[0] = 0;
      u.B = u.A; u.ldb = ld; u.buf[1] = 0;
      u.C = offw(c0 * ld + c0); u.ldc = ld; u.buf[2] = 0;
      u.buf[3] = -1;
      u.M = u.N = (int)rem; u.K = (int)nbk;
      u.alpha = -1.0; u.beta = 1.0;
      u.c_lower = 2;
      u.pad_ok = 1;
      const int gU = add_group(u, DK_U, k);
      GemmArgs uw = u;  // the chain's tiles on the rank that owns k and k+1: operands from the scratch row
      uw.A = off(c0); uw.lda = ldw; uw.buf[0] = slot;
      uw.B = uw.A; uw.ldb = ldw; uw.buf[1] = slot;
      const int gUw = chain_here ? add_group(uw, DK_U, k) : -1;
      auto chain_waits = [&](int n, int i, int j) {
        if (chain_here) {
          wait(n, SS(k, i), ALL);
          if (j != i) wait(n, SS(k, j), ALL);
        } else {
          avail2(n, k, i, j);
        }
      };
      int gUd = -1;
      if (ct != 128 && own(k + 1)) {
        GemmArgs ud = chain_here ? uw : u;
        ud.M = ud.N = nbk1;
        ud.etile = ct;
        gUd = add_group(ud, DK_UD, k);
        const int rt = (nbk1 + ct - 1) / ct;
        for (int a = 0; a < rt; ++a)
          for (int b = a; b < rt; ++b) {
            const int i = lo + a * ct / 128, j = lo + b * ct / 128;
            const int n = add_node(gUd, a, b, DK_UD, tile_cost(ct, nbk));
            if (k > 0) wait(n, VA(i, j), k);
            chain_waits(n, i, j);
            inc(n, G1D(k + 1));
          }
      }
      int gUf[5] = {-1, -1, -1, -1, -1};
      int64_t Kf[5] = {0, 0, 0, 0, 0};
      for (int f = 2; f <= std::min(tune.fuse, 4); f <<= 1) {
        if (k % f != f - 1 || k - (f - 1) < 0) continue;
        GemmArgs uf = u;
        const int64_t o_first = (int64_t)tb[k - (f - 1)] * 128;
        Kf[f] = (int64_t)tb[k + 1] * 128 - o_first;
        uf.A = off(o_first * ld + c0);
        uf.B = uf.A;
        uf.K = (int)Kf[f];
        gUf[f] = add_group(uf, DK_U, k);
      }
      for (int i = lo; i < nt; ++i) {
        if (!own(blk_of(i))) continue;
        const int f = fuse_of(blk_of(i), k);
        for (int j = i; j < nt; ++j) {
          const bool in_next = i < hi, diag = in_next && j < hi;
          if (diag && gUd >= 0) continue;
          if (f > 1) {
            const int g0 = k - k % f;
            if (k != g0 + f - 1) continue;
            const int n = add_node(gUf[f], i - lo, j - lo, DK_U, tile_cost(128, Kf[f]));
            if (g0 > 0) wait(n, VA(i, j), g0);
            avail2(n, k, i, j);
            inc(n, VA(i, j), f);
            continue;
          }
          const int n = add_node(in_next && chain_here ? gUw : gU, i - lo, j - lo, DK_U, tile_cost(128, nbk));
          if (k > 0) wait(n, VA(i, j), k);
          if (in_next) chain_waits(n, i, j);
          else avail2(n, k, i, j);
          if (diag) inc(n, G1D(k + 1));
          else if (in_next) inc(n, UR(k + 1, j));
          else inc(n, VA(i, j));
        }
      }
      // ---- XB(k): the running sums of the owned column blocks -------------------------------------------------------------------
      int gx0 = -1, gx1 = -1;
      if (own(k)) {  // this rank's own column block k: the first contribution, X_kk from D[k] (lower triangular)
        GemmArgs x0{};
        x0.A = off(o * ld + c0); x0.lda = ld; x0.buf[0] = 0;
        x0.B = off(dk); x0.ldb = nb; x0.buf[1] = 3;
        x0.C = offw(c0 * ldc + (int64_t)nl * nb); x0.ldc = ldc; x0.buf[2] = 2;
        x0.buf[3] = -1;
        x0.M = (int)rem; x0.N = (int)nbk; x0.K = (int)nbk;
        x0.alpha = 1.0; x0.beta = 0.0;
        x0.b_mask = 2; x0.klo_mode = 2;
        x0.pad_ok = 1;
        gx0 = add_group(x0, DK_XB, k);
      }
      if (nl > 0) {
        GemmArgs x1{};
        x1.A = off(o * ld + c0); x1.lda = ld; x1.buf[0] = 0;
        x1.B = off(o * ldc); x1.ldb = ldc; x1.buf[1] = 1;
        x1.C = offw(c0 * ldc); x1.ldc = ldc; x1.buf[2] = 2;
        x1.buf[3] = -1;
        x1.M = (int)rem; x1.N = (int)(nl * nb); x1.K = (int)nbk;
        x1.alpha = 1.0; x1.beta = 1.0;
        x1.pad_ok = 1;
        gx1 = add_group(x1, DK_XB, k);
      }
      int gXf[5] = {-1, -1, -1, -1, -1};
      int nlf[5] = {0, 0, 0, 0, 0};
      for (int f = 2; f <= std::min(tune.fuse, 4); f <<= 1) {
        if (k % f != f - 1 || k - (f - 1) < 1) continue;
        const int q0 = k - (f - 1);
        nlf[f] = nleft(q0);
        if (nlf[f] == 0) continue;
        const int64_t o_first = (int64_t)tb[q0] * 128;
        GemmArgs y1{};
        y1.A = off(o_first * ld + c0); y1.lda = ld; y1.buf[0] = 0;
        y1.B = off(o_first * ldc); y1.ldb = ldc; y1.buf[1] = 1;
        y1.C = offw(c0 * ldc); y1.ldc = ldc; y1.buf[2] = 2;
        y1.buf[3] = -1;
        y1.M = (int)rem; y1.N = (int)(nlf[f] * nb); y1.K = (int)Kf[f];
        y1.alpha = 1.0; y1.beta = 1.0;
        y1.pad_ok = 1;
        gXf[f] = add_group(y1, DK_XB, k);
      }
      const int ncols = (nl + (own(k) ? 1 : 0)) * bt;  // compact tiles of the owned column blocks up to block k
      for (int i = lo; i < nt; ++i) {
        const bool fin = i < hi;
        const int fz = fuse_of(blk_of(i), k), q0 = k - k % fz;
        for (int jc = 0; jc < ncols; ++jc) {
          const int j = glob_tile(jc), bj = blk_of(j);
          if (j >= lo) continue;  // (a ragged last block cannot be k here: rem > 0)
          const bool ownb = bj == k;
          if (fz > 1 && bj < q0) {  // columns left of the GROUP's first block: f blocks of X's rows at once, with the group's last step
            if (k != q0 + fz - 1) continue;
            const int n = add_node(gXf[fz], i - lo, jc, DK_XB, tile_cost(128, Kf[fz]));
            avail(n, k, i);
            wait(n, XS(k, j), ALL);
            if (q0 - bj > 0) wait(n, VT(i, j), q0 - bj);
            inc(n, VT(i, j), fz);
            continue;
          }
          const int64_t K = ownb ? nbk - (int64_t)(j - tb[k]) * 128 : nbk;
          const int n = add_node(ownb ? gx0 : gx1, i - lo, ownb ? j - tb[k] : jc, DK_XB, tile_cost(128, K));
          avail(n, k, i);
          if (ownb) wait(n, PD(k), 1);
          else wait(n, XS(k, j), ALL);
          if (k - bj > 0) wait(n, VT(i, j), k - bj);
          if (fin) inc(n, XR(k + 1, j));
          else inc(n, VT(i, j));
        }
      }
    }
  }

  // ---- the sharded evaluation's BACK-substitution of one rank (gp-plus_amd/sharded.py::_backward; DAG_SHARD | DAG_BACK) -----------------
  // The owned column blocks Y of L^-1 (Kc, compact) become the same column blocks of Ky^-1 = L^-T L^-1 (Lc), rows at and below each
  // block's diagonal, right-looking from the last block row up:  Z_j = X_jj^T Y_j, then Y_i -= U[i, j] Z_j for the rows i above.
  // Buffers: 0 A (the factor's mirror L in its strict lower triangle: L[j rows, i] = U[i, j]^T, row-contiguous), 1 Kc, 2 Lc, 3 D.
  //   ZA(j; r, jc)  Lc[j rows, jc] = X_jj^T Kc[j rows, jc] for the owned column blocks left of block j (tile row r: K from its own rows
  //                 down);  after the block row's last update (YR(j, .), raised by step j + 1's tasks on it);  raises ZS(j, jc)
  //   ZD(j; r, c)   the owned diagonal block of Ky^-1: both operands lower triangular, lower tiles only; nothing depends on it
  //   ZU(j; i, jc)  Kc[i, jc] -= L[j rows, i]^T Lc[j rows, jc] for the tiles i above block j, at and below the column's diagonal tile:
  //                 after the strip (ZS) and the tile's previous update (VY); far tiles take f steps' updates at once (K = the f
  //                 blocks' rows, adjacent in the mirror and in Lc), as in the factorisation's list
  // No panels, no messages: one launch over every CU.
  void generate_back() {
    const int me = rank;
    const int bt = (int)(nb / GPP_TILE);
    const int64_t ldc = ldi;
    cur_lvl = 0;
    auto group_of = [&](int j, int f) {  // the aligned group of f steps that contains step j: (first processed = largest j, last)
      const int t = B - 1 - j, tg0 = t - t % f;
      return std::make_pair(B - 1 - tg0, B - 1 - tg0 - f + 1);
    };
    auto fuse_of = [&](int bi, int j) {
      for (int f = std::min(tune.fuse, 4); f >= 2; f >>= 1) {
        const auto g = group_of(j, f);
        if (g.second >= me + 1 && bi <= g.second - 2) return f;
      }
      return 1;
    };
    for (int j = B - 1; j >= me; --j) {
      const int btk = tb[j + 1] - tb[j];
      const int64_t o = (int64_t)tb[j] * 128, nbj = rows_of(j);
      const int64_t dk = (int64_t)j * nb * nb;
      const int nl = nleft(j);
      if (nl > 0) {
        GemmArgs g{};
        g.A = off(dk); g.lda = nb; g.buf[0] = 3;
        g.B = off(o * ldc); g.ldb = ldc; g.buf[1] = 1;
        g.C = offw(o * ldc); g.ldc = ldc; g.buf[2] = 2;
        g.buf[3] = -1;
        g.M = (int)nbj; g.N = (int)(nl * nb); g.K = (int)nbj;
        g.alpha = 1.0; g.beta = 0.0;
        g.a_mask = 2; g.klo_mode = 1;
        g.pad_ok = 1;
        const int gi = add_group(g, DK_XA, j);
        for (int r = 0; r < btk; ++r)
          for (int jc = 0; jc < nl * bt; ++jc) {
            const int gj = glob_tile(jc);
            const int n = add_node(gi, r, jc, DK_XA, tile_cost(128, nbj - (int64_t)r * 128));
            if (j < B - 1) wait(n, YR(j, gj), ALL);
            inc(n, ZS(j, gj));
          }
      }
      if (own(j)) {
        GemmArgs g{};
        g.A = off(dk); g.lda = nb; g.buf[0] = 3;
        g.B = off(o * ldc + (int64_t)nl * nb); g.ldb = ldc; g.buf[1] = 1;
        g.C = offw(o * ldc + (int64_t)nl * nb); g.ldc = ldc; g.buf[2] = 2;
        g.buf[3] = -1;
        g.M = g.N = g.K = (int)nbj;
        g.alpha = 1.0; g.beta = 0.0;
        g.a_mask = 2; g.b_mask = 2; g.klo_mode = 3;
        g.c_lower = 1;
        g.pad_ok = 1;
        const int gi = add_group(g, DK_S, j);
        for (int r = 0; r < btk; ++r)
          for (int c = 0; c <= r; ++c) {
            const int n = add_node(gi, r, c, DK_S, tile_cost(128, nbj - (int64_t)r * 128));
            if (j < B - 1) wait(n, YR(j, tb[j] + c), ALL);
          }
      }
      if (nl == 0 || j <= me) continue;
      GemmArgs u{};
      u.A = off(o * ld); u.lda = ld; u.buf[0] = 0;
      u.B = off(o * ldc); u.ldb = ldc; u.buf[1] = 2;
      u.C = offw(0); u.ldc = ldc; u.buf[2] = 1;
      u.buf[3] = -1;
      u.M = (int)o; u.N = (int)(nl * nb); u.K = (int)nbj;
      u.alpha = -1.0; u.beta = 1.0;
      u.pad_ok = 1;
      const int gU = add_group(u, DK_U, j);
      int gUf[5] = {-1, -1, -1, -1, -1};
      int64_t Kf[5] = {0, 0, 0, 0, 0};
      int jg_of[5] = {0, 0, 0, 0, 0};
      for (int f = 2; f <= std::min(tune.fuse, 4); f <<= 1) {
        const auto g = group_of(j, f);
        if (g.second != j || g.first > B - 1 || j < me + 1) continue;
        GemmArgs uf = u;
        Kf[f] = std::min<int64_t>((int64_t)tb[g.first + 1] * 128, N) - o;  // rows of blocks j .. j + f - 1
        uf.K = (int)Kf[f];
        gUf[f] = add_group(uf, DK_U, j);
        jg_of[f] = g.first;
      }
      for (int i = tb[me]; i < tb[j]; ++i) {
        const int bi = blk_of(i);
        const int f = fuse_of(bi, j);
        for (int jc = 0; jc < nl * bt; ++jc) {
          const int gj = glob_tile(jc);
          if (i < gj) continue;  // above the column's diagonal tile
          if (f > 1) {
            const auto g = group_of(j, f);
            if (j != g.second) continue;  // taken with the group's last step
            // (the columns of the group's own blocks take part from their own steps on: those tiles are never "far")
            const int n = add_node(gUf[f], i, jc, DK_U, tile_cost(128, Kf[f]));
            wait(n, ZS(j, gj), ALL);
            if (B - 1 - g.first > 0) wait(n, VY(i, gj), B - 1 - g.first);
            inc(n, VY(i, gj), f);
            continue;
          }
          const int n = add_node(gU, i, jc, DK_U, tile_cost(128, nbj));
          wait(n, ZS(j, gj), ALL);
          if (B - 1 - j > 0) wait(n, VY(i, gj), B - 1 - j);
          if (bi == j - 1) inc(n, YR(j - 1, gj));
          else inc(n, VY(i, gj));
        }
      }
    }
  }

  // ALL -> the number of incrementers; predecessor lists; bottom levels.  False when the graph is not a DAG (a planner bug).
  std::vector<std::vector<int>> preds;
  bool resolve() {
    for (Node& n : nodes)
      for (int q = 0; q < 3; ++q)
        if (n.w[q] >= 0 && n.v[q] == ALL) n.v[q] = (int)incs[n.w[q]].size();
    const int nn = (int)nodes.size();
    preds.assign(nn, {});
    std::vector<int> nsucc(nn, 0);
    for (int t = 0; t < nn; ++t) {
      const Node& n = nodes[t];
      for (int q = 0; q < 3; ++q) {
        if (n.w[q] < 0) continue;
        const auto& in = incs[n.w[q]];
        if (n.v[q] < 1 || n.v[q] > (int)in.size()) return false;  // would never be satisfied
        if (chain[n.w[q]]) preds[t].push_back(in[n.v[q] - 1]);
        else {
          if (n.v[q] != (int)in.size()) return false;  // "any v of n" would depend on the execution order
          for (int p : in) preds[t].push_back(p);
        }
      }
      for (int p : preds[t]) ++nsucc[p];
    }
    // a version counter's v-th increment must itself follow the (v-1)-th: its incrementers wait for it with v-1 (checked here)
    for (int c = 0; c < ncounters; ++c)
      if (chain[c])
        for (size_t q = 1; q < incs[c].size(); ++q) {
          if (incs[c][q] == incs[c][q - 1]) continue;  // the second increment of a fused pair
          const Node& n = nodes[incs[c][q]];
          bool ok = false;
          for (int z = 0; z < 3; ++z) ok |= (n.w[z] == c && n.v[z] == (int)q);
          if (!ok) return false;
        }
    // bottom levels by reverse topological order (Kahn on the reversed graph)
    std::vector<int> stack;
    for (int t = 0; t < nn; ++t) {
      nodes[t].bl = nodes[t].cost;
      if (nsucc[t] == 0) stack.push_back(t);
    }
    int seen = 0;
    while (!stack.empty()) {
      const int t = stack.back();
      stack.pop_back();
      ++seen;
      for (int p : preds[t]) {
        nodes[p].bl = std::max(nodes[p].bl, nodes[p].cost + nodes[t].bl);
        if (--nsucc[p] == 0) stack.push_back(p);
      }
    }
    return seen == nn;
  }

  // List scheduling on `workers` identical workers (panels on their own stream, one at a time): returns the nodes in start order.
  // fill_m[b] > 0: after the panel of block b a filler launch of tune.fill work-groups on the panel's CUs takes up to fill_m[b] tasks
  // each from the same list; it stops taking tasks when the next panel's gate opens, and that panel starts when the launch has ended.
  std::vector<int> order;
  std::vector<int> fill_m;            // per block (empty: no filler launches)
  std::vector<double> p_end, g_open;  // per block: end of its panel, time its gate opened (simulated)
  double makespan = 0, busy = 0;
  void simulate() {
    const int nn = (int)nodes.size(), W = std::max(1, tune.workers), F = std::max(0, tune.fill);
    std::vector<int> value(ncounters, 0), unsat(nn, 0);
    struct Waiter { int need, node; };
    std::vector<std::vector<Waiter>> waiters(ncounters);
    std::vector<int> wpos(ncounters, 0);
    for (int t = 0; t < nn; ++t)
      for (int q = 0; q < 3; ++q)
        if (nodes[t].w[q] >= 0) {
          ++unsat[t];
          waiters[nodes[t].w[q]].push_back({nodes[t].v[q], t});
        }
    for (auto& wl : waiters) std::sort(wl.begin(), wl.end(), [](const Waiter& a, const Waiter& b) { return a.need < b.need; });
    auto better = [&](int a, int b) {  // a runs before b
      if (nodes[a].bl != nodes[b].bl) return nodes[a].bl > nodes[b].bl;
      return a < b;
    };
    auto cmp = [&](int a, int b) { return better(b, a); };
    std::priority_queue<int, std::vector<int>, decltype(cmp)> ready(cmp);
    struct Ev {
      double t;
      int node, wid;  // node >= 0: completion (wid < 0 main worker, else filler worker); node == -2: the filler launch of block wid starts
      bool operator>(const Ev& o) const { return t > o.t; }
    };
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
    std::vector<int> panel_ready;
    double now = 0, panel_free = 0, comm_free = 0;
    int wfree = W;
    std::vector<int> fbudget(F, 0), ffree;  // filler workers of the current launch: tasks left, ids of the idle ones
    int falive = 0;                         // filler workers that have not left yet
    bool fquit = false;                     // the next gate is open: no filler takes a further task
    int pending_panel = -1;                 // a panel whose gate is open but whose CUs still run a filler launch
    order.clear();
    order.reserve(nn);
    p_end.assign(B, 0);
    g_open.assign(B, 0);
    auto make_ready = [&](int t) {
      if (nodes[t].kind <= -2) {  // a remote slab's arrival (sharded lists): the communication stream, serial
        comm_free = std::max(now, comm_free) + nodes[t].cost;
        events.push({comm_free, t, -1});
        order.push_back(t);
      } else if (nodes[t].kind < 0) panel_ready.push_back(t);
      else ready.push(t);
    };
    auto start_panel = [&](int t) {
      const double st = std::max(now, panel_free);
      panel_free = st + nodes[t].cost;
      events.push({panel_free, t, -1});
      order.push_back(t);
    };
    for (int t = 0; t < nn; ++t)
      if (unsat[t] == 0) make_ready(t);
    busy = 0;
    for (;;) {
      for (int t : panel_ready) {  // (panels become ready in block order: the stream runs them in that order)
        g_open[nodes[t].tm] = now;
        fquit = true;
        falive -= (int)ffree.size();  // idle filler workers leave at once
        ffree.clear();
        if (falive > 0) pending_panel = t;
        else start_panel(t);
      }
      panel_ready.clear();
      while (!ready.empty() && (wfree > 0 || !ffree.empty())) {
        const int t = ready.top();
        ready.pop();
        int wid = -1;
        if (wfree > 0) --wfree;
        else {
          wid = ffree.back();
          ffree.pop_back();
          --fbudget[wid];
        }
        order.push_back(t);
        events.push({now + nodes[t].cost, t, wid});
        busy += nodes[t].cost;
      }
      if (events.empty()) break;
      const Ev e = events.top();
      events.pop();
      now = e.t;
      if (e.node == -2) {  // filler launch of block e.wid
        if (!fquit && F > 0) {
          falive = F;
          for (int f = 0; f < F; ++f) {
            fbudget[f] = fill_m[e.wid];
            ffree.push_back(f);
          }
        }
        continue;
      }
      const int t = e.node;
      if (nodes[t].kind <= -2) {
      } else if (nodes[t].kind < 0) {
        const int b = nodes[t].tm;
        p_end[b] = now;
        fquit = false;
        if (F > 0 && !fill_m.empty() && fill_m[b] > 0) events.push({now + 12.0, -2, b});
      } else if (e.wid < 0) {
        ++wfree;
      } else {
        if (fquit || fbudget[e.wid] <= 0) {
          if (--falive == 0 && pending_panel >= 0) {
            start_panel(pending_panel);
            pending_panel = -1;
          }
        } else {
          ffree.push_back(e.wid);
        }
      }
      for (int q = 0; q < 2; ++q) {
        const int c = nodes[t].inc[q];
        if (c < 0) continue;
        const int v = (value[c] += nodes[t].incv[q]);
        auto& wl = waiters[c];
        while (wpos[c] < (int)wl.size() && wl[wpos[c]].need <= v) {
          const int u = wl[wpos[c]++].node;
          if (--unsat[u] == 0) make_ready(u);
        }
      }
    }
    makespan = now;
    busy = makespan > 0 ? busy / (makespan * W) : 0;
  }
  // Two passes: without filler launches first, which tells how long the panel's CUs idle between panel b and the gate of block
  // b + 1; the filler launches are sized from that (whole tasks of the big tile, a margin of 0.35 of one) and the list re-ordered.
  void schedule() {
    fill_m.clear();
    simulate();
    if (tune.fill <= 0) return;
    const double t_tile = tile_cost(128, nb);
    fill_m.assign(B, 0);
    bool any = false;
    std::vector<int> pb;  // the blocks whose panels run here, in order (a sharded list: the owned ones)
    for (const Node& n : nodes)
      if (n.kind == -1) pb.push_back(n.tm);
    if (pb.empty()) {
      fill_m.clear();
      return;
    }
    for (size_t q = 0; q + 1 < pb.size(); ++q) {
      const int b = pb[q], nx = pb[q + 1];
      const double idle = g_open[nx] - p_end[b] - 12.0;
      const int m = (int)std::floor(idle / t_tile - 0.35);
      fill_m[b] = std::max(0, std::min(m, 64 * (nx - b)));
      any |= fill_m[b] > 0;
    }
    // behind the LAST panel its CUs are free for good: a launch without a budget works the list to its end (the inverse's tasks)
    if (makespan - p_end[pb.back()] > 2.0 * t_tile) {
      fill_m[pb.back()] = 1 << 20;
      any = true;
    }
    if (any) simulate();
    else fill_m.clear();
  }
};

// a plan from the planner's state (tasks in `order`, panels as stream operations)
DagPlan* emit(Planner& pl) {
  DagPlan* P = new DagPlan();
  P->N = pl.N; P->nb = pl.nb; P->ld = pl.ld; P->ldi = pl.ldi; P->ldt = pl.ldt; P->ldk = pl.ldk;
  P->flags = pl.flags;
  P->inv_rows = (pl.flags & DAG_INV) ? (int64_t)pl.inv_tiles * GPP_TILE : 0;
  if (P->inv_rows > pl.N) P->inv_rows = pl.N;
  P->B = pl.B; P->nt = pl.nt; P->tb = pl.tb;
  P->groups = pl.groups;
  P->ncounters = pl.ncounters;
  P->c_pd = pl.c_pd; P->c_g1d = pl.c_g1d;
  P->gate_target.assign(pl.B, 0);
  P->rank = pl.rank; P->nranks = pl.nranks;
  P->c_cph = pl.c_cph; P->c_cpt = pl.c_cpt; P->c_art = pl.c_art;
  P->cph_target = pl.cph_n; P->cpt_target = pl.cpt_n;
  P->piece_tiles = pl.piece_tiles; P->GP = pl.GP;
  P->sim_ms = pl.makespan * 1e-3;
  P->sim_busy = pl.busy;
  // first ticket of each level: a filler launch behind panel b stops in front of the first task that needs panel b + 1
  std::vector<int> first_of(pl.B + 1, 1 << 30);
  {
    int pos = 0;
    for (int t : pl.order) {
      const Node& n = pl.nodes[t];
      if (n.kind < 0) continue;
      first_of[n.lvl] = std::min(first_of[n.lvl], pos);
      ++pos;
    }
    for (int b = pl.B - 1; b >= 0; --b) first_of[b] = std::min(first_of[b], first_of[b + 1]);  // "level >= b"
  }
  P->level_first = first_of;
  for (int t : pl.order) {
    const Node& n = pl.nodes[t];
    if (n.kind <= -2) continue;  // (a remote slab's arrival: the caller's communication stream signals it)
    if (n.kind < 0) {
      const int b = n.tm;
      if (b > 0) {
        P->gate_target[b] = n.v[0];
        P->stream_ops.push_back({0, b, 0, 0});
      }
      P->stream_ops.push_back({1, b, 0, 0});
      P->stream_ops.push_back({2, b, 0, 0});
      int nx = -1;  // the next block whose panel runs on this stream (a sharded list: the next owned one)
      for (int c = b + 1; c < pl.B && nx < 0; ++c)
        if (pl.own(c)) nx = c;
      if (!pl.fill_m.empty() && pl.fill_m[b] > 0 && (nx < 0 || first_of[nx] > 0))
        P->stream_ops.push_back({3, nx >= 0 ? pl.fill_m[b] : 0, nx, nx >= 0 ? first_of[nx] : 0});
      continue;
    }
    DagTask d;
    d.group = n.group;
    d.tm = (int16_t)n.tm;
    d.tn = (int16_t)n.tn;
    for (int q = 0; q < 3; ++q) {
      d.wait_id[q] = n.w[q];
      d.wait_val[q] = n.w[q] >= 0 ? n.v[q] : 0;
    }
    d.inc_id[0] = n.inc[0];
    d.inc_id[1] = n.inc[1];
    d.inc_val[0] = (int16_t)n.incv[0];
    d.inc_val[1] = (int16_t)n.incv[1];
    d.kind = n.kind;
    P->tasks.push_back(d);
  }
  return P;
}

}  // namespace

// Columns per piece of a block row's tail message (gpp.h).  Measured with tools/replay_rank.py at C5 on 8 virtual ranks: see
// profiles/r06_virtual_rank.txt.  GPP_SHARD_PIECE_COLS=0: one piece, as in round 5.
extern "C" int64_t gpp_shard_piece_cols(void) {
  static const int64_t cols = [] {
    const int64_t v = getenv("GPP_SHARD_PIECE_COLS") ? atol(getenv("GPP_SHARD_PIECE_COLS")) : 8192;
    return v <= 0 ? (int64_t)0 : std::max<int64_t>(GPP_TILE, v / GPP_TILE * GPP_TILE);
  }();
  return cols;
}

DagTuning gpp_dag_default_tuning() {
  DagTuning t;
  // two work-groups per CU, measured per task with TRACE=1 tools/dag_check.py (profiles/r05_dag_traces.txt): the big tile 264 us at
  // K = 1024 and 1035 at 4096; the 64-tile 60 us at a mean K of 576 and 105 at 1024; a strip copy 150-180 us (latency-bound beside
  // the MFMA work); a panel 620-640 us for 8 leaves.  The ORDER of the list is only as good as these: with the first guesses
  // (3.7 us per chunk, 0.75 for the 64-tile, 12 us per copy, 580 us per panel) chain tasks were taken ~70 us before they could run.
  t.t0_big = getenv("GPP_DAG_T0") ? atof(getenv("GPP_DAG_T0")) : 14.0;
  t.tc_big = getenv("GPP_DAG_TC") ? atof(getenv("GPP_DAG_TC")) : 3.9;
  t.t0_64 = 15.0; t.tc_64 = 1.4;
  t.t0_32 = 6.0; t.tc_32 = 0.55;
  t.t_copy = 160.0;
  t.t_panel0 = getenv("GPP_DAG_TP0") ? atof(getenv("GPP_DAG_TP0")) : 20.0;
  t.t_panel_leaf = getenv("GPP_DAG_TPL") ? atof(getenv("GPP_DAG_TPL")) : 77.0;
  t.t_gate = getenv("GPP_DAG_TGATE") ? atof(getenv("GPP_DAG_TGATE")) : 15.0;
  t.chain_tile = getenv("GPP_DAG_CHAIN_TILE") ? atoi(getenv("GPP_DAG_CHAIN_TILE")) : 64;
  t.workers = 448;
  t.inv_rows = 0;
  t.piece_cols = gpp_shard_piece_cols();
  t.fuse = getenv("GPP_DAG_FUSE") ? atoi(getenv("GPP_DAG_FUSE")) : 1;  // (potrf_dag chooses by size)
  t.fill = getenv("GPP_DAG_FILL") ? atoi(getenv("GPP_DAG_FILL")) : 64;
  return t;
}

DagPlan* gpp_dag_plan(int64_t N, int64_t nb, int64_t ld, int64_t ldi, int64_t ldt, int64_t ldk, int flags, const DagTuning& tune, int rank,
                      int nranks) {
  Planner pl;
  pl.rank = rank; pl.nranks = std::max(nranks, 1);
  pl.N = N; pl.nb = nb; pl.ld = ld; pl.ldi = ldi; pl.ldt = ldt; pl.ldk = ldk;
  pl.flags = flags;
  pl.tune = tune;
  if (!pl.layout()) return nullptr;
  pl.generate();
  if (!pl.resolve()) {
    fprintf(stderr, "libgpp_hip: dag planner: the task graph of N = %lld, nb = %lld is not a DAG (planner bug); not used\n", (long long)N, (long long)nb);
    return nullptr;
  }
  pl.schedule();
  if (pl.order.size() != pl.nodes.size()) {
    fprintf(stderr, "libgpp_hip: dag planner: the simulation of N = %lld left %zu of %zu tasks unscheduled (planner bug); not used\n",
            (long long)N, pl.nodes.size() - pl.order.size(), pl.nodes.size());
    return nullptr;
  }
  DagPlan* P = emit(pl);
  if (getenv("GPP_EXEC_VERBOSE"))
    fprintf(stderr, "libgpp_hip: dag plan N=%lld nb=%lld flags=%d: %d blocks, %zu tasks, %zu groups, %d counters; simulated %.2f ms, "
                    "%.0f %% busy on %d workers\n", (long long)N, (long long)nb, flags, P->B, P->tasks.size(), P->groups.size(), P->ncounters,
            P->sim_ms, 100.0 * P->sim_busy, tune.workers);
  if (getenv("GPP_DAG_DUMP")) {
    auto fam = [&](int c) {
      static char buf[64];
      if (c < 0) return (const char*)"-";
      const char* nm = "?";
      int base = 0;
      struct { const char* n; int b; } fs[] = {{"PD", pl.c_pd}, {"G1D", pl.c_g1d}, {"UR", pl.c_ur}, {"SS", pl.c_ss}, {"XR", pl.c_xr}, {"XS", pl.c_xs},
                                               {"VA", pl.c_va}, {"VT", pl.c_vt}, {"CPD", pl.c_cpd}, {"CPH", pl.c_cph}, {"CPT", pl.c_cpt}, {"ART", pl.c_art}};
      for (auto& f : fs)
        if (f.b > 0 && c >= f.b && f.b >= base) { nm = f.n; base = f.b; }
      snprintf(buf, sizeof buf, "%s+%d", nm, c - base);
      return (const char*)buf;
    };
    const int nd = atoi(getenv("GPP_DAG_DUMP"));
    for (int t = 0; t < nd && t < (int)P->tasks.size(); ++t) {
      const DagTask& d = P->tasks[t];
      fprintf(stderr, "  r%d ticket %d kind %d blk %d tile (%d,%d) waits", rank, t, d.kind, pl.gmeta[d.group].second, d.tm, d.tn);
      for (int q = 0; q < 3; ++q)
        if (d.wait_id[q] >= 0) fprintf(stderr, " %s>=%d", fam(d.wait_id[q]), d.wait_val[q]);
      fprintf(stderr, " incs");
      for (int q = 0; q < 2; ++q)
        if (d.inc_id[q] >= 0) fprintf(stderr, " %s", fam(d.inc_id[q]));
      fprintf(stderr, "\n");
    }
  }
  if (getenv("GPP_EXEC_VERBOSE") && !pl.fill_m.empty()) {
    fprintf(stderr, "libgpp_hip: dag plan filler tasks per work-group and block:");
    for (int m : pl.fill_m) fprintf(stderr, " %d", m);
    fprintf(stderr, "\n");
  }
  return P;
}

hipError_t gpp_dag_upload(DagPlan* P) {
  hipError_t e;
  if (P->d_tasks) return hipSuccess;
  if ((e = hipMalloc(&P->d_tasks, P->tasks.size() * sizeof(DagTask))) != hipSuccess) return e;
  if ((e = hipMemcpy(P->d_tasks, P->tasks.data(), P->tasks.size() * sizeof(DagTask), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipMalloc(&P->d_groups, P->groups.size() * sizeof(GemmArgs))) != hipSuccess) return e;
  if ((e = hipMemcpy(P->d_groups, P->groups.data(), P->groups.size() * sizeof(GemmArgs), hipMemcpyHostToDevice)) != hipSuccess) return e;
  if ((e = hipMalloc(&P->d_groups_abs, P->groups.size() * sizeof(GemmArgs))) != hipSuccess) return e;
  if ((e = hipMalloc(&P->d_counters, (size_t)P->ncounters * sizeof(int))) != hipSuccess) return e;
  return hipEventCreateWithFlags(&P->last_use, hipEventDisableTiming);
}

void gpp_dag_free(DagPlan* P) {
  if (!P) return;
  if (P->last_use) {
    (void)hipEventSynchronize(P->last_use);  // the launches that read the device copies are over
    (void)hipEventDestroy(P->last_use);
  }
  // (hipFree synchronises the DEVICE: an eviction from the handle's LRU — more than GPP_DAG_PLANS shapes alive — is not free of
  //  device-wide waits; it happens in front of a re-plan that costs far more host time, before anything of that call is enqueued)
  if (P->d_groups) (void)hipFree(P->d_groups);
  if (P->d_groups_abs) (void)hipFree(P->d_groups_abs);
  if (P->d_tasks) (void)hipFree(P->d_tasks);
  if (P->d_counters) (void)hipFree(P->d_counters);
  if (P->d_trace) (void)hipFree(P->d_trace);
  delete P;
}

// ---- host-side verification of a plan (tests/test_host_cpu.py) ---------------------------------------------------------------------
// Executes the ticket list on the host with W workers in random and adversarial interleavings that respect only what the device
// respects — tickets are taken in list order, a task runs when its counters allow, the panel stream runs gate / panel / signal in
// order — and checks what the ARITHMETIC needs, from the tasks' geometry alone (never from their counters): a solve reads a fully
// updated, not yet overwritten block row and a factored diagonal block; an update reads completely solved strips and is the k-th
// on its tile; a panel starts on a fully updated block; the inverse's sums take their contributions in order from final rows of X;
// a row of X is built from complete sums; everything is complete at the end; nothing deadlocks (W = 1 executes the list in order,
// so the order itself must be topological).  Returns 0 or a code naming the first violation.  stats[0..3]: tasks run, waits,
// increments, tasks run by filler launches — or, with `mutate` > 0, which removes the mutate-th wait of the plan first (the check
// must then FAIL), the kind of the task that lost its wait * 10 + which counter family.
extern "C" int gpp_debug_dag_check(int64_t N, int64_t nb, int flags, int chain_tile, int W, int fill, unsigned seed, int64_t* stats,
                                   int mutate) {
  // (flags >= 4: flags >> 2 = rows of the leading block whose inverse is built inside the list; flags & 1 = DAG_INV;
  //  chain_tile / 1000 = the fusion factor (0: 1), chain_tile % 1000 the tile)
  DagTuning tune = gpp_dag_default_tuning();
  tune.inv_rows = (int64_t)(flags >> 2);
  flags &= 3;
  if (chain_tile >= 1000) {
    tune.fuse = chain_tile / 1000;
    chain_tile %= 1000;
  }
  tune.chain_tile = chain_tile;
  tune.fill = fill;
  tune.workers = std::max(W, 1);
  DagPlan* P = gpp_dag_plan(N, nb, N, N, N, N, flags, tune);
  if (!P) return 1;
  // geometry, recomputed independently of the planner's counters
  const int nt = P->nt, B = P->B;
  const std::vector<int>& tb = P->tb;
  auto blk_of = [&](int tile) { return (int)(std::upper_bound(tb.begin(), tb.end(), tile) - tb.begin()) - 1; };
  // which block / kind a group belongs to: from its operands (offsets), not from planner metadata
  struct GInfo { int kind, k; };
  std::vector<GInfo> gi(P->groups.size());
  for (size_t g = 0; g < P->groups.size(); ++g) {
    const GemmArgs& a = P->groups[g];
    const int64_t offC = (int64_t)(reinterpret_cast<uintptr_t>(a.C) / 8), offB = (int64_t)(reinterpret_cast<uintptr_t>(a.B) / 8);
    int kind, k;
    if (a.op == 1) { kind = DK_CP; k = blk_of((int)(offB / N / 128)); }
    else if (a.buf[2] == 0) { kind = a.etile && a.etile != 128 ? DK_UD : DK_U; k = blk_of((int)(offB / N / 128)); }          // C in A: update
    else if (a.buf[2] == 1) { kind = DK_XA; k = blk_of((int)(offC / N / 128)); }                                               // C in Linv
    else if (a.buf[1] == 0) { kind = a.etile && a.etile != 128 ? DK_SH : DK_S; k = blk_of((int)(offC / N / 128)); }          // B from A: solve
    else { kind = DK_XB; k = blk_of((int)(offB / N / 128)); }                                                                  // B from Linv
    gi[g] = {kind, k};
  }
  int mutated = -1, lazy_counter = -1;
  long mut_task = -1;
  const int mutate_in = mutate;
  if (mutate > 0) {
    int seen = 0;
    for (auto& t : P->tasks)
      for (int q = 0; q < 3 && mutate > 0; ++q)
        if (t.wait_id[q] >= 0 && ++seen == mutate) {
          const int c = t.wait_id[q];
          const int fam = c < P->c_g1d ? 0 : c < P->c_g1d + B ? 1 : 2 + (int)(((int64_t)c - (P->c_g1d + B)) / ((int64_t)B * nt));
          mutated = 10 * gi[t.group].kind + std::min(fam, 9);
          if (getenv("GPP_DAG_CHECK_VERBOSE"))
            fprintf(stderr, "mutate: task %ld kind %d block %d tile (%d, %d) loses wait %d (counter %d >= %d, family %d)\n",
                    (long)(&t - P->tasks.data()), gi[t.group].kind, gi[t.group].k, (int)t.tm, (int)t.tn, q, c, t.wait_val[q], fam);
          t.wait_id[q] = -1;
          mutate = 0;
          lazy_counter = c;
          mut_task = (long)(&t - P->tasks.data());
        }
  }
  // With a wait removed the adversary is TARGETED: the tasks (and the panel-stream signal) that raise the counter the removed wait
  // referred to run only when nothing else can — on the device: those work-groups are slow —, so the task that lost its wait runs
  // before them whenever the remaining waits allow it at all.
  std::vector<char> lazy(P->tasks.size(), 0);
  if (lazy_counter >= 0)
    for (size_t t = 0; t < P->tasks.size(); ++t)
      lazy[t] = P->tasks[t].inc_id[0] == lazy_counter || P->tasks[t].inc_id[1] == lazy_counter;
  auto tixU = [&](int i, int j) { return (size_t)((int64_t)i * nt - (int64_t)i * (i - 1) / 2 + (j - i)); };
  std::vector<int> counters(P->ncounters, 0);
  std::vector<int> a_upd((size_t)nt * (nt + 1) / 2, 0);              // updates applied to A[i][j], i <= j
  std::vector<int> ud_done((size_t)nt * (nt + 1) / 2, 0), ud_need((size_t)nt * (nt + 1) / 2, 0);  // small-tile updates of a tile in flight
  std::vector<int> t_done((size_t)nt * nt, 0), t_need((size_t)nt * nt, 0);  // solve tasks done / covering T[i][c] (upper right)
  std::vector<int> t_acc((size_t)nt * nt, 0);                        // contributions in T[i][j] (lower left)
  std::vector<char> x_done((size_t)nt * nt, 0), copied((size_t)B * nt, 0), panel_done(B, 0);
  // how many solve / small update tasks cover each 128-tile
  for (const DagTask& t : P->tasks) {
    const GInfo g = gi[t.group];
    const int et = P->groups[t.group].etile ? P->groups[t.group].etile : 128;
    if (g.kind == DK_S || g.kind == DK_SH) {
      const int lo = tb[g.k + 1];
      const int i = tb[g.k] + t.tm * et / 128, c = lo + t.tn * et / 128;
      ++t_need[(size_t)i * nt + c];
    } else if (g.kind == DK_UD) {
      const int lo = tb[g.k + 1];
      ++ud_need[tixU(lo + t.tm * et / 128, lo + t.tn * et / 128)];
    }
  }
  uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  int rc = 0;
  int64_t ran = 0, waits = 0, nincs = 0;
  auto fail = [&](int code) { if (!rc) rc = code; };
  auto solved = [&](int k, int c) {  // every tile of T[k rows, c] is completely solved
    for (int i = tb[k]; i < tb[k + 1]; ++i)
      if (t_need[(size_t)i * nt + c] == 0 || t_done[(size_t)i * nt + c] != t_need[(size_t)i * nt + c]) return false;
    return true;
  };
  auto try_task = [&](const DagTask& t) -> bool {
    for (int q = 0; q < 3; ++q)
      if (t.wait_id[q] >= 0 && counters[t.wait_id[q]] < t.wait_val[q]) return false;
    if (getenv("GPP_DAG_CHECK_VERBOSE") && lazy_counter >= 0 && (&t - P->tasks.data()) == mut_task)
      fprintf(stderr, "mutated task runs: lazy counter %d = %d, tasks run so far %ld\n", lazy_counter, counters[lazy_counter], (long)ran);
    const GInfo g = gi[t.group];
    const GemmArgs& a = P->groups[t.group];
    const int et = a.etile ? a.etile : 128, k = g.k;
    if (g.kind == DK_S || g.kind == DK_SH) {
      const int lo = tb[k + 1];
      const int i = tb[k] + t.tm * et / 128, c = lo + t.tn * et / 128;
      if (!panel_done[k]) fail(10);
      if (copied[(size_t)k * nt + c]) fail(11);
      const int rmax = std::min(tb[k + 1] - 1, tb[k] + ((t.tm + 1) * et - 1) / 128);  // K range ends with this tile's last row
      for (int r = tb[k]; r <= rmax; ++r)
        if (a_upd[tixU(r, c)] != k || ud_done[tixU(r, c)] != 0) fail(12);
      if (t_done[(size_t)i * nt + c] >= t_need[(size_t)i * nt + c]) fail(13);
      ++t_done[(size_t)i * nt + c];
    } else if (g.kind == DK_U || g.kind == DK_UD) {
      // a fused pair (K spans two blocks of T's rows): the updates of steps k and k + 1 at once, tiles relative to block k + 2
      int f = 1;  // steps this task applies: K spans the rows of blocks k .. k + f - 1
      if (g.kind == DK_U)
        while (k + f < B && (int64_t)a.K > std::min<int64_t>((int64_t)tb[k + f] * 128, N) - (int64_t)tb[k] * 128) ++f;
      const int lo = tb[k + f];
      const int i = lo + t.tm * et / 128, j = lo + t.tn * et / 128;
      if (i > j || j >= nt) fail(20);
      for (int q = 0; q < f; ++q)
        if (!solved(k + q, i) || !solved(k + q, j)) fail(q == 0 ? 21 : 24);
      if (a_upd[tixU(i, j)] != k) fail(22);
      if (g.kind == DK_U) {
        if (ud_need[tixU(i, j)] != 0 && blk_of(i) == k + 1 && blk_of(j) == k + 1) fail(23);  // tile covered twice
        if (f > 1 && blk_of(i) < k + f + 1) fail(25);  // a fused task on a tile the group's own steps depend on
        a_upd[tixU(i, j)] = k + f;
      } else {
        if (++ud_done[tixU(i, j)] == ud_need[tixU(i, j)]) {
          ud_done[tixU(i, j)] = 0;
          a_upd[tixU(i, j)] = k + 1;
        }
      }
    } else if (g.kind == DK_CP) {
      const int c = tb[k + 1] + t.tn;
      if (!solved(k, c)) fail(30);
      if (copied[(size_t)k * nt + c]) fail(31);
      copied[(size_t)k * nt + c] = 1;
    } else if (g.kind == DK_XB) {
      int f = 1;  // contributions this task adds: K spans the rows of blocks k .. k + f - 1
      while (k + f < B && (int64_t)a.K > std::min<int64_t>((int64_t)tb[k + f] * 128, N) - (int64_t)tb[k] * 128) ++f;
      const int lo = tb[k + f];
      const int i = lo + t.tm;
      const bool own = a.beta == 0.0;
      const int j = own ? tb[k] + t.tn : t.tn;
      const int bj = blk_of(j);
      if (own ? bj != k : bj >= k) fail(42);
      if (f > 1 && blk_of(i) < k + f + 1) fail(45);
      for (int q = 0; q < f; ++q) {
        if (!solved(k + q, i)) fail(40);
        if (k + q == bj) {
          if (!panel_done[k + q]) fail(41);
        } else {
          for (int r = tb[k + q]; r < tb[k + q + 1]; ++r)
            if (!x_done[(size_t)r * nt + j]) fail(43);
        }
      }
      if (t_acc[(size_t)i * nt + j] != k - bj) fail(44);
      t_acc[(size_t)i * nt + j] += f;
    } else if (g.kind == DK_XA) {
      const int i = tb[k] + t.tm, j = t.tn, bj = blk_of(j);
      if (!panel_done[k]) fail(50);
      if (bj >= k) fail(51);
      for (int r = tb[k]; r <= i; ++r)
        if (t_acc[(size_t)r * nt + j] != k - bj) fail(52);
      if (x_done[(size_t)i * nt + j]) fail(53);
      x_done[(size_t)i * nt + j] = 1;
    } else {
      fail(60);
    }
    for (int q = 0; q < 3; ++q)
      if (t.wait_id[q] >= 0) ++waits;
    for (int q = 0; q < 2; ++q)
      if (t.inc_id[q] >= 0) {
        counters[t.inc_id[q]] += t.inc_val[q];
        ++nincs;
      }
    ++ran;
    return true;
  };
  size_t op = 0;
  bool stream_done = P->stream_ops.empty();
  const int64_t ntasks = (int64_t)P->tasks.size();
  int64_t head = 0, finished = 0, fill_tasks = 0;
  // a filler launch in progress: its work-groups' current tickets (-1: none), tasks left, and whether each has left
  std::vector<int64_t> fcur;
  std::vector<int> fleft;
  bool in_fill = false;
  auto stream_step = [&]() -> bool {
    if (stream_done) return false;
    const DagPlan::Op o = P->stream_ops[op];
    if (o.kind == 0) {
      if (counters[P->c_g1d + o.arg] < P->gate_target[o.arg]) return false;
    } else if (o.kind == 1) {
      const int b = o.arg;
      for (int i = tb[b]; i < tb[b + 1]; ++i)
        for (int j = i; j < tb[b + 1]; ++j)
          if (a_upd[tixU(i, j)] != b || ud_done[tixU(i, j)] != 0) fail(70);
      if (panel_done[b]) fail(71);
      panel_done[b] = 1;
    } else if (o.kind == 2) {
      if (!panel_done[o.arg]) fail(72);
      ++counters[P->c_pd + o.arg];
    } else {
      // filler launch: `fill` work-groups, each takes up to o.arg tasks and none once the gate counter of block o.n has moved
      if (!in_fill) {
        in_fill = true;
        fcur.assign(std::max(fill, 1), -1);
        fleft.assign(std::max(fill, 1), o.arg > 0 ? o.arg : 1 << 30);
      }
      bool progressed = false, any_alive = false;
      const size_t nf = fcur.size(), s0 = (size_t)(rnd() % nf);
      for (size_t q = 0; q < nf; ++q) {
        const size_t f = (s0 + q) % nf;
        if (fcur[f] < 0) {
          if (fleft[f] <= 0) continue;  // has left
          if ((o.n >= 0 && counters[P->c_g1d + o.n] >= 1) || head >= ntasks || (o.lim > 0 && head >= o.lim)) {
            fleft[f] = 0;
            progressed = true;
            continue;
          }
          if (progressed) { any_alive = true; continue; }
          fcur[f] = head++;
          --fleft[f];
          progressed = true;
        }
        any_alive = true;
        if (!progressed || fcur[f] >= 0) {
          if (try_task(P->tasks[fcur[f]])) {
            fcur[f] = -1;
            ++finished;
            ++fill_tasks;
            progressed = true;
          }
        }
      }
      if (any_alive) return progressed;
      for (size_t f = 0; f < nf; ++f)
        if (fcur[f] >= 0 || fleft[f] > 0) return progressed;
      in_fill = false;
    }
    if (++op == P->stream_ops.size()) stream_done = true;
    return true;
  };
  const int mode = (lazy_counter >= 0 && (seed & 3u) == 0) ? 1 : (int)(seed & 3u);
  std::vector<int64_t> cur(W, -1);
  bool allow_lazy = lazy_counter < 0;
  while (!rc) {
    bool progressed = false;
    if ((mode == 0 && rnd() % 8 == 0) || (in_fill && rnd() % 3 == 0)) progressed = stream_step();
    const int w0 = mode >= 2 ? 0 : (int)(rnd() % W);
    for (int q = 0; q < W && !progressed; ++q) {
      const int w = mode == 2 ? W - 1 - q : (w0 + q) % W;
      const int burst = (mode >= 2 || rnd() % 16 == 0) ? (1 << 30) : 1 + (int)(rnd() % 3);
      // taking a ticket and running its task are SEPARATE steps: a work-group may sit on a runnable task for any length of time
      for (int b = 0; b < burst; ++b) {
        if (cur[w] < 0) {
          if (head >= ntasks) break;
          cur[w] = head++;
          progressed = true;
          if (mode < 2 && rnd() % 4 != 0) break;
          continue;
        }
        if (lazy[cur[w]] && !allow_lazy) break;
        if (!try_task(P->tasks[cur[w]])) break;
        cur[w] = -1;
        ++finished;
        progressed = true;
      }
    }
    if (!progressed) progressed = stream_step();
    if (finished == ntasks && stream_done) break;
    if (!progressed && !allow_lazy) {
      allow_lazy = true;  // nothing else can run: one round with the slow tasks
      continue;
    }
    if (lazy_counter >= 0) allow_lazy = false;
    if (!progressed) fail(2);  // deadlock
  }
  if (!rc) {
    for (int b = 0; b < B && !rc; ++b) {
      if (!panel_done[b]) fail(80);
      for (int c = tb[b + 1]; c < nt && !rc; ++c)
        if (!copied[(size_t)b * nt + c]) fail(81);
      if ((flags & DAG_INV) && (int64_t)tb[b + 1] * 128 <= std::max<int64_t>(P->inv_rows, 0) + 127)
        for (int i = tb[b]; i < tb[b + 1] && !rc; ++i)
          for (int j = 0; j < tb[b]; ++j)
            if (!x_done[(size_t)i * nt + j]) { fail(82); break; }
    }
  }
  if (stats) {
    stats[0] = ran;
    stats[1] = waits;
    stats[2] = nincs;
    stats[3] = mutate_in > 0 ? mutated : fill_tasks;
  }
  gpp_dag_free(P);
  return rc;
}

// the planner's own estimate for a size (tools): stats = {tasks, simulated us, busy per mille, blocks}
extern "C" int gpp_debug_dag_sim(int64_t N, int64_t nb, int flags, int chain_tile, int workers, int64_t* stats) {
  DagTuning tune = gpp_dag_default_tuning();
  tune.chain_tile = chain_tile;
  tune.workers = workers;
  DagPlan* P = gpp_dag_plan(N, nb, N, N, N, N, flags, tune);
  if (!P) return 1;
  stats[0] = (int64_t)P->tasks.size();
  stats[1] = (int64_t)(P->sim_ms * 1000.0);
  stats[2] = (int64_t)(P->sim_busy * 1000.0);
  stats[3] = P->B;
  gpp_dag_free(P);
  return 0;
}

// ---- host-side verification of the SHARDED lists (tests/test_host_cpu.py) ------------------------------------------------------------
// The lists of all `nranks` ranks executed TOGETHER on the host: per rank W workers taking tickets in list order, the panel stream
// (gate / panel / signal of the owned blocks) and the communication stream as gp-plus_amd/sharded.py drives it — for k = 0, 1, ...: the
// head message of block row k, then its tail; the owner's side waits at the gates (panel done + its copies of the head's / tail's strips
// counted), a receiver's side completes only after the owner's has, and then raises PD(k) / ART(k).  Interleavings are random or
// adversarial as in gpp_debug_dag_check, and so are the checks: what the arithmetic needs, from the tasks' geometry alone — the
// operands' buffers and offsets — never from their counters.  Returns 0 or a code naming the first violation (+ 1000 * rank).
// stats: tasks run, waits, increments, (with `mutate` > 0, which removes the mutate-th wait over all ranks' lists: the check must
// then fail) kind * 10 + counter family of the task that lost its wait.
extern "C" int gpp_debug_shard_check(int64_t N, int64_t nb, int nranks, int chain_tile, int W, int fill, unsigned seed, int64_t* stats,
                                     int mutate) {
  DagTuning tune = gpp_dag_default_tuning();
  if (chain_tile >= 1000) {
    tune.fuse = chain_tile / 1000;
    chain_tile %= 1000;
  }
  tune.chain_tile = chain_tile;
  tune.fill = fill;
  tune.workers = std::max(W, 1);
  const int P = std::max(nranks, 1);
  const int64_t nblk = (N + nb - 1) / nb;
  struct Rank {
    DagPlan* plan = nullptr;
    std::vector<int> counters, a_upd, ud_done, ud_need, t_done, t_need, t_acc, w_content;
    std::vector<char> x_done, copied, panel_done, arr_head, arr_tail, sent_head, sent_tail, lazy;
    std::vector<int64_t> cur;
    size_t op = 0, cop = 0;
    int64_t head = 0, finished = 0, fill_tasks = 0;
    int64_t ldc = 0;
    // a filler launch in progress: its work-groups' current tickets (-1: none), tasks left
    std::vector<int64_t> fcur;
    std::vector<int> fleft;
    bool in_fill = false;
  };
  std::vector<Rank> R(P);
  auto cleanup = [&]() {
    for (Rank& r : R)
      if (r.plan) gpp_dag_free(r.plan);
  };
  for (int r = 0; r < P; ++r) {
    int64_t nq = 0;
    for (int64_t b = r; b < nblk; b += P) ++nq;
    R[r].ldc = std::max<int64_t>(nq, 1) * nb;
    R[r].plan = gpp_dag_plan(N, nb, N, R[r].ldc, N, N, DAG_INV | DAG_SHARD, tune, r, P);
    if (!R[r].plan) {
      cleanup();
      return 1;
    }
  }
  const int nt = R[0].plan->nt, B = R[0].plan->B;
  const std::vector<int> tb = R[0].plan->tb;
  const int bt = (int)(nb / GPP_TILE);
  auto blk_of = [&](int tile) { return (int)(std::upper_bound(tb.begin(), tb.end(), tile) - tb.begin()) - 1; };
  auto tixU = [&](int i, int j) { return (size_t)((int64_t)i * nt - (int64_t)i * (i - 1) / 2 + (j - i)); };
  auto hi_of = [&](int k) { return tb[std::min(k + 2, B)]; };
  // the tail's pieces, recomputed from the documented rule (gpp.h), not read from the plans
  const int pt = gpp_shard_piece_cols() > 0 ? (int)(gpp_shard_piece_cols() / GPP_TILE) : nt;
  auto npieces = [&](int k) { return hi_of(k) >= nt ? 0 : (nt - hi_of(k) + pt - 1) / pt; };
  const int GP = std::max(1, npieces(0));
  if (R[0].plan->GP != GP || R[0].plan->piece_tiles != pt) {
    cleanup();
    return 3;
  }
  auto elems = [](const void* p) { return (int64_t)(reinterpret_cast<uintptr_t>(p) / 8); };
  for (int r = 0; r < P; ++r) {
    Rank& q = R[r];
    q.counters.assign(q.plan->ncounters, 0);
    q.a_upd.assign((size_t)nt * (nt + 1) / 2, 0);
    q.ud_done.assign((size_t)nt * (nt + 1) / 2, 0);
    q.ud_need.assign((size_t)nt * (nt + 1) / 2, 0);
    q.t_done.assign((size_t)nt * nt, 0);
    q.t_need.assign((size_t)nt * nt, 0);
    q.t_acc.assign((size_t)nt * nt, 0);
    q.w_content.assign((size_t)3 * nt, -1);
    q.x_done.assign((size_t)nt * nt, 0);
    q.copied.assign((size_t)B * nt, 0);
    q.panel_done.assign(B, 0);
    q.arr_head.assign(B, 0); q.arr_tail.assign((size_t)B * GP, 0); q.sent_head.assign(B, 0); q.sent_tail.assign((size_t)B * GP, 0);
    q.cur.assign(std::max(W, 1), -1);
    q.lazy.assign(q.plan->tasks.size(), 0);
  }
  // a task's kind and blocks from its group's operands
  struct GInfo { int kind, k, f, slot; bool from_w, own0; };
  auto ginfo = [&](const Rank& q, const GemmArgs& a) {
    GInfo g{-1, 0, 1, -1, false, false};
    const int64_t ld = q.plan->ld, ldc = q.plan->ldi;
    if (a.op == 1) {
      g.kind = DK_CP;
      g.k = (int)(elems(a.C) / ld / nb);
      g.slot = a.buf[1] - 4;
    } else if (a.buf[2] == 0) {
      g.kind = a.etile && a.etile != 128 ? DK_UD : DK_U;
      const int64_t c0 = elems(a.C) / ld;
      g.f = (int)((a.K + nb - 1) / nb);
      g.k = (int)(c0 / nb) - g.f;  // FIRST step of the group
      g.from_w = a.buf[0] >= 4;
      g.slot = a.buf[0] - 4;
    } else if (a.buf[2] == 1) {
      g.kind = DK_XA;
      g.k = (int)(elems(a.A) / (nb * nb));
    } else if (a.buf[2] == 2) {
      g.kind = DK_XB;
      const int64_t c0 = elems(a.C) / ldc;
      g.f = (int)((a.K + nb - 1) / nb);
      g.k = (int)(c0 / nb) - g.f;
      g.own0 = a.beta == 0.0;
    } else if (a.buf[2] >= 4) {
      g.kind = a.etile && a.etile != 128 ? DK_SH : DK_S;
      g.k = (int)(elems(a.A) / (nb * nb));
      g.slot = a.buf[2] - 4;
    }
    return g;
  };
  // the wait to remove
  int mutated = -1, lazy_rank = -1, lazy_counter = -1;
  const int mutate_in = mutate;
  if (mutate > 0) {
    int seen = 0;
    for (int r = 0; r < P && mutate > 0; ++r)
      for (auto& t : R[r].plan->tasks)
        for (int z = 0; z < 3 && mutate > 0; ++z)
          if (t.wait_id[z] >= 0 && ++seen == mutate) {
            const DagPlan* pl = R[r].plan;
            const int c = t.wait_id[z];
            int fam;
            if (c >= pl->c_art) fam = 9;
            else if (c >= pl->c_cph) fam = 8;          // CPH / CPT (never waited for by tasks)
            else if (c >= pl->c_cph - B * nt) fam = 7;  // CPD
            else fam = c < pl->c_g1d ? 0 : c < pl->c_g1d + B ? 1 : 2 + (int)std::min<int64_t>(((int64_t)c - (pl->c_g1d + B)) / ((int64_t)B * nt), 4);
            mutated = 10 * ginfo(R[r], pl->groups[t.group]).kind + fam;
            if (getenv("GPP_DAG_CHECK_VERBOSE"))
              fprintf(stderr, "mutate: rank %d task %ld kind %d tile (%d, %d) loses wait %d (counter %d >= %d, family %d)\n", r,
                      (long)(&t - pl->tasks.data()), mutated / 10, (int)t.tm, (int)t.tn, z, c, t.wait_val[z], fam);
            t.wait_id[z] = -1;
            mutate = 0;
            lazy_rank = r;
            lazy_counter = c;
          }
    if (lazy_rank >= 0)
      for (size_t t = 0; t < R[lazy_rank].plan->tasks.size(); ++t) {
        const DagTask& d = R[lazy_rank].plan->tasks[t];
        R[lazy_rank].lazy[t] = d.inc_id[0] == lazy_counter || d.inc_id[1] == lazy_counter;
      }
  }
  for (int r = 0; r < P; ++r)
    for (const DagTask& t : R[r].plan->tasks) {
      const GemmArgs& a = R[r].plan->groups[t.group];
      const GInfo g = ginfo(R[r], a);
      const int et = a.etile ? a.etile : 128;
      if (g.kind == DK_S || g.kind == DK_SH) ++R[r].t_need[(size_t)(tb[g.k] + t.tm * et / 128) * nt + tb[g.k + 1] + t.tn * et / 128];
      else if (g.kind == DK_UD) ++R[r].ud_need[tixU(tb[g.k + 1] + t.tm * et / 128, tb[g.k + 1] + t.tn * et / 128)];
    }
  uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  int rc = 0;
  int64_t ran = 0, waits = 0, nincs = 0;
  auto fail = [&](int r, int code) { if (!rc) rc = code + 1000 * r; };
  auto own = [&](int r, int b) { return b % P == r; };
  auto glob = [&](int r, int jc) { return (r + (jc / bt) * P) * bt + jc % bt; };
  auto solved = [&](const Rank& q, int k, int c) {
    for (int i = tb[k]; i < tb[k + 1]; ++i)
      if (q.t_need[(size_t)i * nt + c] == 0 || q.t_done[(size_t)i * nt + c] != q.t_need[(size_t)i * nt + c]) return false;
    return true;
  };
  auto in_a = [&](int r, const Rank& q, int k, int c) {  // block row k of the factor, column tile c, is in this rank's A
    if (own(r, k)) return (bool)q.copied[(size_t)k * nt + c];
    return (bool)(c < hi_of(k) ? q.arr_head[k] : q.arr_tail[(size_t)k * GP + (c - hi_of(k)) / pt]);
  };
  auto d_here = [&](int r, const Rank& q, int k) { return (bool)(own(r, k) ? q.panel_done[k] : q.arr_head[k]); };
  auto try_task = [&](int r, const DagTask& t) -> bool {
    Rank& q = R[r];
    for (int z = 0; z < 3; ++z)
      if (t.wait_id[z] >= 0 && q.counters[t.wait_id[z]] < t.wait_val[z]) return false;
    const GemmArgs& a = q.plan->groups[t.group];
    const GInfo g = ginfo(q, a);
    const int et = a.etile ? a.etile : 128, k = g.k;
    if (g.kind == DK_S || g.kind == DK_SH) {
      const int lo = tb[k + 1];
      const int i = tb[k] + t.tm * et / 128, c = lo + t.tn * et / 128;
      if (!own(r, k) || g.slot != ((k - r) / P) % 3) fail(r, 14);
      if (!q.panel_done[k]) fail(r, 10);
      if (q.copied[(size_t)k * nt + c]) fail(r, 11);
      const int rmax = std::min(tb[k + 1] - 1, tb[k] + ((t.tm + 1) * et - 1) / 128);
      for (int rr = tb[k]; rr <= rmax; ++rr)
        if (q.a_upd[tixU(rr, c)] != k || q.ud_done[tixU(rr, c)] != 0) fail(r, 12);
      if (q.t_done[(size_t)i * nt + c] >= q.t_need[(size_t)i * nt + c]) fail(r, 13);
      // the scratch row's strip: free of its previous user (copied out, and no update of the chain still reads it)
      int& wc = q.w_content[(size_t)g.slot * nt + c];
      if (wc != k) {
        if (wc >= 0) {
          const int kp = wc;
          if (!q.copied[(size_t)kp * nt + c]) fail(r, 15);
          if (own(r, kp + 1))  // the chain's tiles of step kp read W: block row kp+1 against column c
            for (int ii = tb[kp + 1]; ii < tb[kp + 2]; ++ii)
              if (ii <= c && q.a_upd[tixU(ii, c)] < kp + 1) fail(r, 16);
        }
        wc = k;
      }
      ++q.t_done[(size_t)i * nt + c];
    } else if (g.kind == DK_U || g.kind == DK_UD) {
      const int f = g.f, lo = tb[k + f];
      const int i = lo + t.tm * et / 128, j = lo + t.tn * et / 128;
      if (i > j || j >= nt) fail(r, 20);
      else {
        if (!own(r, blk_of(i))) fail(r, 26);
        if (g.from_w) {
          if (f != 1 || !own(r, k) || !solved(q, k, i) || !solved(q, k, j)) fail(r, 21);
          if (q.w_content[(size_t)g.slot * nt + i] != k || q.w_content[(size_t)g.slot * nt + j] != k || g.slot != ((k - r) / P) % 3) fail(r, 27);
          if (blk_of(i) != k + 1) fail(r, 28);  // only the chain's tiles may read the scratch row (its reuse is ordered for those alone)
        } else {
          for (int z = 0; z < f; ++z)
            if (!in_a(r, q, k + z, i) || !in_a(r, q, k + z, j)) fail(r, z == f - 1 ? 21 : 24);
        }
        if (q.a_upd[tixU(i, j)] != k) fail(r, 22);
        if (g.kind == DK_U) {
          if (q.ud_need[tixU(i, j)] != 0 && blk_of(i) == k + 1 && blk_of(j) == k + 1) fail(r, 23);
          if (f > 1 && blk_of(i) < k + f + 1) fail(r, 25);
          q.a_upd[tixU(i, j)] = k + f;
        } else {
          if (++q.ud_done[tixU(i, j)] == q.ud_need[tixU(i, j)]) {
            q.ud_done[tixU(i, j)] = 0;
            q.a_upd[tixU(i, j)] = k + 1;
          }
        }
      }
    } else if (g.kind == DK_CP) {
      const int c = tb[k + 1] + t.tn;
      if (!own(r, k) || !solved(q, k, c) || q.w_content[(size_t)g.slot * nt + c] != k) fail(r, 30);
      if (q.copied[(size_t)k * nt + c]) fail(r, 31);
      q.copied[(size_t)k * nt + c] = 1;
    } else if (g.kind == DK_XB) {
      const int f = g.f, lo = tb[k + f];
      const int i = lo + t.tm;
      const int j = g.own0 ? tb[k] + t.tn : glob(r, t.tn);
      const int bj = j < nt ? blk_of(j) : B;
      if (i >= nt || j >= nt || !own(r, bj)) fail(r, 46);
      else {
        if (g.own0 ? (bj != k || f != 1) : bj >= k) fail(r, 42);
        if (f > 1 && blk_of(i) < k + f + 1) fail(r, 45);
        for (int z = 0; z < f; ++z) {
          if (!in_a(r, q, k + z, i)) fail(r, 40);
          if (k + z == bj) {
            if (!q.panel_done[k + z]) fail(r, 41);
          } else {
            for (int rr = tb[k + z]; rr < tb[k + z + 1]; ++rr)
              if (!q.x_done[(size_t)rr * nt + j]) fail(r, 43);
          }
        }
        if (q.t_acc[(size_t)i * nt + j] != k - bj) fail(r, 44);
        q.t_acc[(size_t)i * nt + j] += f;
      }
    } else if (g.kind == DK_XA) {
      const int i = tb[k] + t.tm, j = glob(r, t.tn), bj = j < nt ? blk_of(j) : B;
      if (i >= nt || j >= nt || !own(r, bj) || bj >= k) fail(r, 51);
      else {
        if (!d_here(r, q, k)) fail(r, 50);
        for (int rr = tb[k]; rr <= i; ++rr)
          if (q.t_acc[(size_t)rr * nt + j] != k - bj) fail(r, 52);
        if (q.x_done[(size_t)i * nt + j]) fail(r, 53);
        q.x_done[(size_t)i * nt + j] = 1;
      }
    } else {
      fail(r, 60);
    }
    for (int z = 0; z < 3; ++z)
      if (t.wait_id[z] >= 0) ++waits;
    for (int z = 0; z < 2; ++z)
      if (t.inc_id[z] >= 0) {
        q.counters[t.inc_id[z]] += t.inc_val[z];
        ++nincs;
      }
    ++ran;
    return true;
  };
  bool allow_lazy_g = true;
  auto stream_step = [&](int r) -> bool {  // the panel stream of rank r
    Rank& q = R[r];
    const DagPlan* pl = q.plan;
    if (q.op >= pl->stream_ops.size()) return false;
    const DagPlan::Op o = pl->stream_ops[q.op];
    if (o.kind == 0) {
      if (q.counters[pl->c_g1d + o.arg] < pl->gate_target[o.arg]) return false;
    } else if (o.kind == 1) {
      const int b = o.arg;
      if (r == lazy_rank && pl->c_pd + b == lazy_counter && !allow_lazy_g) return false;  // (a slow panel)
      if (!own(r, b)) fail(r, 73);
      for (int i = tb[b]; i < tb[b + 1]; ++i)
        for (int j = i; j < tb[b + 1]; ++j)
          if (q.a_upd[tixU(i, j)] != b || q.ud_done[tixU(i, j)] != 0) fail(r, 70);
      if (q.panel_done[b]) fail(r, 71);
      q.panel_done[b] = 1;
    } else if (o.kind == 2) {
      if (!q.panel_done[o.arg]) fail(r, 72);
      if (r == lazy_rank && pl->c_pd + o.arg == lazy_counter && !allow_lazy_g) return false;  // (slow, not stuck)
      ++q.counters[pl->c_pd + o.arg];
    } else {
      // filler launch: `fill` work-groups, each takes up to o.arg tasks and none once the gate counter of block o.n has moved
      const int64_t ntasks = (int64_t)pl->tasks.size();
      if (!q.in_fill) {
        q.in_fill = true;
        q.fcur.assign(std::max(fill, 1), -1);
        q.fleft.assign(std::max(fill, 1), o.arg > 0 ? o.arg : 1 << 30);
      }
      bool progressed = false, any_alive = false;
      const size_t nf = q.fcur.size(), s0 = (size_t)(rnd() % nf);
      for (size_t z = 0; z < nf; ++z) {
        const size_t f = (s0 + z) % nf;
        if (q.fcur[f] < 0) {
          if (q.fleft[f] <= 0) continue;  // has left
          if ((o.n >= 0 && q.counters[pl->c_g1d + o.n] >= 1) || q.head >= ntasks || (o.lim > 0 && q.head >= o.lim)) {
            q.fleft[f] = 0;
            progressed = true;
            continue;
          }
          if (progressed) { any_alive = true; continue; }
          q.fcur[f] = q.head++;
          --q.fleft[f];
          progressed = true;
        }
        any_alive = true;
        if (!progressed || q.fcur[f] >= 0) {
          if (q.lazy[q.fcur[f]] && !allow_lazy_g) continue;
          if (try_task(r, pl->tasks[q.fcur[f]])) {
            q.fcur[f] = -1;
            ++q.finished;
            ++q.fill_tasks;
            progressed = true;
          }
        }
      }
      if (any_alive) return progressed;
      for (size_t f = 0; f < nf; ++f)
        if (q.fcur[f] >= 0 || q.fleft[f] > 0) return progressed;
      q.in_fill = false;
    }
    ++q.op;
    return true;
  };
  // the communication stream of rank r: operation k (GP + 1) + m = message m of block row k — 0 its head, 1 + g piece g of its tail
  const size_t nops = (size_t)B * (GP + 1);
  auto skip_absent = [&](size_t c) {
    while (c < nops && (int)(c % (GP + 1)) > npieces((int)(c / (GP + 1)))) ++c;
    return c;
  };
  auto comm_step = [&](int r) -> bool {
    Rank& q = R[r];
    const DagPlan* pl = q.plan;
    if (P == 1) return false;
    q.cop = skip_absent(q.cop);
    if (q.cop >= nops) return false;
    const int k = (int)(q.cop / (GP + 1)), m = (int)(q.cop % (GP + 1));
    const bool tail = m > 0;
    const size_t pc = (size_t)k * GP + (tail ? m - 1 : 0);  // the piece's slot
    if (own(r, k)) {
      // gates: the panel's signal, then the copies counted
      if (q.counters[pl->c_pd + k] < 1) return false;
      if (!tail && q.counters[pl->c_cph + k] < pl->cph_target[k]) return false;
      if (tail && q.counters[pl->c_cpt + (int)pc] < pl->cpt_target[pc]) return false;
      if (!q.panel_done[k]) fail(r, 90);
      const int cb = tail ? hi_of(k) + (m - 1) * pt : tb[k + 1], ce = tail ? std::min(cb + pt, nt) : hi_of(k);
      for (int c = cb; c < ce; ++c)
        if (!q.copied[(size_t)k * nt + c]) fail(r, 91);
      if (tail) q.sent_tail[pc] = 1;
      else q.sent_head[k] = 1;
    } else {
      const Rank& ow = R[k % P];
      if (!(tail ? ow.sent_tail[pc] : ow.sent_head[k])) return false;
      const int cid = tail ? pl->c_art + (int)pc : pl->c_pd + k;
      if (r == lazy_rank && cid == lazy_counter && !allow_lazy_g) return false;
      if (tail) q.arr_tail[pc] = 1;
      else q.arr_head[k] = 1;
      ++q.counters[cid];
    }
    ++q.cop;
    return true;
  };
  auto comm_done = [&](int r) { return P == 1 || skip_absent(R[r].cop) >= nops; };
  const int mode = (lazy_counter >= 0 && (seed & 3u) == 0) ? 1 : (int)(seed & 3u);
  allow_lazy_g = lazy_counter < 0;
  const int Wn = std::max(W, 1);
  while (!rc) {
    bool progressed = false;
    const int r0 = (int)(rnd() % P);
    for (int rq = 0; rq < P && !rc; ++rq) {
      const int r = (r0 + rq) % P;
      Rank& q = R[r];
      const int64_t ntasks = (int64_t)q.plan->tasks.size();
      bool prog_r = false;
      if ((mode == 0 && rnd() % 8 == 0) || (q.in_fill && rnd() % 3 == 0)) prog_r = stream_step(r) || comm_step(r);
      const int w0 = mode >= 2 ? 0 : (int)(rnd() % Wn);
      for (int z = 0; z < Wn && !prog_r; ++z) {
        const int w = mode == 2 ? Wn - 1 - z : (w0 + z) % Wn;
        const int burst = (mode >= 2 || rnd() % 16 == 0) ? (1 << 30) : 1 + (int)(rnd() % 3);
        for (int b = 0; b < burst; ++b) {
          if (q.cur[w] < 0) {
            if (q.head >= ntasks) break;
            q.cur[w] = q.head++;
            prog_r = true;
            if (mode < 2 && rnd() % 4 != 0) break;
            continue;
          }
          if (q.lazy[q.cur[w]] && !allow_lazy_g) break;
          if (!try_task(r, q.plan->tasks[q.cur[w]])) break;
          q.cur[w] = -1;
          ++q.finished;
          prog_r = true;
        }
      }
      // a rank's streams move when its workers made no progress in this round
      if (!prog_r) prog_r = stream_step(r);
      if (!prog_r) prog_r = comm_step(r);
      progressed |= prog_r;
    }
    bool all_done = true;
    for (int r = 0; r < P; ++r)
      all_done &= R[r].finished == (int64_t)R[r].plan->tasks.size() && R[r].op >= R[r].plan->stream_ops.size() && comm_done(r);
    if (all_done) break;
    if (!progressed && !allow_lazy_g) {
      allow_lazy_g = true;  // nothing else can run: one round with the slow tasks / signals
      continue;
    }
    if (lazy_counter >= 0) allow_lazy_g = false;
    if (!progressed) fail(0, 2);  // deadlock
  }
  if (!rc)
    for (int r = 0; r < P && !rc; ++r) {
      const Rank& q = R[r];
      for (int b = 0; b < B && !rc; ++b) {
        if (own(r, b)) {
          if (!q.panel_done[b]) fail(r, 80);
          for (int c = tb[b + 1]; c < nt && !rc; ++c)
            if (!q.copied[(size_t)b * nt + c]) fail(r, 81);
        } else if (P > 1) {
          if (!q.arr_head[b]) fail(r, 83);
          for (int g = 0; g < npieces(b); ++g)
            if (!q.arr_tail[(size_t)b * GP + g]) fail(r, 83);
        }
        for (int i = tb[b]; i < tb[b + 1] && !rc; ++i)
          for (int j = 0; j < tb[b]; ++j)
            if (own(r, blk_of(j)) && !q.x_done[(size_t)i * nt + j]) { fail(r, 82); break; }
      }
    }
  if (stats) {
    stats[0] = ran;
    stats[1] = waits;
    stats[2] = nincs;
    stats[3] = mutate_in > 0 ? mutated : 0;
    if (mutate_in <= 0)
      for (const Rank& q : R) stats[3] += q.fill_tasks;
  }
  cleanup();
  return rc;
}

// The back-substitution's list of ONE rank (no messages, no panels) on the host, as above: W workers, random / adversarial
// interleavings, the checks from the tasks' geometry: a row of Z is built from final rows of Y, an update reads complete rows of Z and
// is applied in order and exactly once, only to tiles at and below the diagonal of column blocks the rank owns; everything complete.
extern "C" int gpp_debug_shard_back_check(int64_t N, int64_t nb, int nranks, int rank, int fuse, int W, unsigned seed, int64_t* stats,
                                          int mutate) {
  DagTuning tune = gpp_dag_default_tuning();
  tune.fuse = std::max(fuse, 1);
  tune.fill = 0;
  tune.workers = std::max(W, 1);
  const int P = std::max(nranks, 1);
  const int64_t nblk = (N + nb - 1) / nb;
  int64_t nq = 0;
  for (int64_t b = rank; b < nblk; b += P) ++nq;
  const int64_t ldc = std::max<int64_t>(nq, 1) * nb;
  DagPlan* pl = gpp_dag_plan(N, nb, N, ldc, N, N, DAG_SHARD | DAG_BACK, tune, rank, P);
  if (!pl) return 1;
  const int nt = pl->nt, B = pl->B, bt = (int)(nb / GPP_TILE);
  const std::vector<int> tb = pl->tb;
  auto blk_of = [&](int tile) { return (int)(std::upper_bound(tb.begin(), tb.end(), tile) - tb.begin()) - 1; };
  auto elems = [](const void* p) { return (int64_t)(reinterpret_cast<uintptr_t>(p) / 8); };
  auto glob = [&](int jc) { return (rank + (jc / bt) * P) * bt + jc % bt; };
  auto own = [&](int b) { return b % P == rank; };
  int mutated = -1, lazy_counter = -1;
  const int mutate_in = mutate;
  if (mutate > 0) {
    int seen = 0;
    for (auto& t : pl->tasks)
      for (int z = 0; z < 3 && mutate > 0; ++z)
        if (t.wait_id[z] >= 0 && ++seen == mutate) {
          lazy_counter = t.wait_id[z];
          mutated = 10 * t.kind + (lazy_counter < 2 + B * nt ? 0 : lazy_counter < 2 + 2 * B * nt ? 1 : 2);  // ZS, YR, VY
          t.wait_id[z] = -1;
          mutate = 0;
        }
  }
  std::vector<char> lazy(pl->tasks.size(), 0);
  if (lazy_counter >= 0)
    for (size_t t = 0; t < pl->tasks.size(); ++t) lazy[t] = pl->tasks[t].inc_id[0] == lazy_counter || pl->tasks[t].inc_id[1] == lazy_counter;
  std::vector<int> counters(pl->ncounters, 0), y_upd((size_t)nt * nt, 0);
  std::vector<char> z_done((size_t)nt * nt, 0), zd_done((size_t)nt * nt, 0);
  uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  int rc = 0;
  int64_t ran = 0, waits = 0, nincs = 0;
  auto fail = [&](int code) { if (!rc) rc = code; };
  auto try_task = [&](const DagTask& t) -> bool {
    for (int z = 0; z < 3; ++z)
      if (t.wait_id[z] >= 0 && counters[t.wait_id[z]] < t.wait_val[z]) return false;
    const GemmArgs& a = pl->groups[t.group];
    if (a.buf[2] == 2 && a.c_lower == 0) {  // ZA
      const int j = (int)(elems(a.A) / (nb * nb)), i = tb[j] + t.tm, gj = glob(t.tn);
      if (j < rank || j >= B || i >= tb[j + 1] || gj >= tb[j] || !own(blk_of(gj))) fail(10);
      else {
        for (int ii = i; ii < tb[j + 1]; ++ii)
          if (y_upd[(size_t)ii * nt + gj] != B - 1 - j) fail(11);  // a row of Y still to be updated
        if (z_done[(size_t)i * nt + gj]) fail(12);
        z_done[(size_t)i * nt + gj] = 1;
      }
    } else if (a.buf[2] == 2) {  // ZD
      const int j = (int)(elems(a.A) / (nb * nb)), i = tb[j] + t.tm, c = tb[j] + t.tn;
      if (j >= B || !own(j) || i >= tb[j + 1] || c > i) fail(20);
      else {
        for (int ii = i; ii < tb[j + 1]; ++ii)
          if (y_upd[(size_t)ii * nt + c] != B - 1 - j) fail(21);
        if (zd_done[(size_t)i * nt + c]) fail(22);
        zd_done[(size_t)i * nt + c] = 1;
      }
    } else if (a.buf[2] == 1) {  // ZU
      const int jl = (int)(elems(a.A) / pl->ld / nb);
      int f = 1;
      while (jl + f < B && (int64_t)a.K > std::min<int64_t>((int64_t)tb[jl + f] * 128, N) - (int64_t)tb[jl] * 128) ++f;
      const int i = t.tm, gj = glob(t.tn);
      if (jl + f > B || i >= tb[jl] || gj > i || gj >= nt || !own(blk_of(gj))) fail(30);
      else {
        for (int z = 0; z < f; ++z)
          for (int rr = tb[jl + z]; rr < tb[jl + z + 1]; ++rr)
            if (!z_done[(size_t)rr * nt + gj]) fail(z == 0 ? 31 : 34);
        if (y_upd[(size_t)i * nt + gj] != B - 1 - (jl + f - 1)) fail(32);
        if (f > 1 && blk_of(i) > jl - 2) fail(33);
        y_upd[(size_t)i * nt + gj] += f;
      }
    } else {
      fail(60);
    }
    for (int z = 0; z < 3; ++z)
      if (t.wait_id[z] >= 0) ++waits;
    for (int z = 0; z < 2; ++z)
      if (t.inc_id[z] >= 0) {
        counters[t.inc_id[z]] += t.inc_val[z];
        ++nincs;
      }
    ++ran;
    return true;
  };
  const int mode = (lazy_counter >= 0 && (seed & 3u) == 0) ? 1 : (int)(seed & 3u);
  const int Wn = std::max(W, 1);
  std::vector<int64_t> cur(Wn, -1);
  const int64_t ntasks = (int64_t)pl->tasks.size();
  int64_t head = 0, finished = 0;
  bool allow_lazy = lazy_counter < 0;
  while (!rc && finished < ntasks) {
    bool progressed = false;
    const int w0 = mode >= 2 ? 0 : (int)(rnd() % Wn);
    for (int z = 0; z < Wn && !progressed; ++z) {
      const int w = mode == 2 ? Wn - 1 - z : (w0 + z) % Wn;
      const int burst = (mode >= 2 || rnd() % 16 == 0) ? (1 << 30) : 1 + (int)(rnd() % 3);
      for (int b = 0; b < burst; ++b) {
        if (cur[w] < 0) {
          if (head >= ntasks) break;
          cur[w] = head++;
          progressed = true;
          if (mode < 2 && rnd() % 4 != 0) break;
          continue;
        }
        if (lazy[cur[w]] && !allow_lazy) break;
        if (!try_task(pl->tasks[cur[w]])) break;
        cur[w] = -1;
        ++finished;
        progressed = true;
      }
    }
    if (!progressed && !allow_lazy) {
      allow_lazy = true;
      continue;
    }
    if (lazy_counter >= 0) allow_lazy = false;
    if (!progressed) fail(2);
  }
  if (!rc)
    for (int j = rank; j < B && !rc; ++j)
      for (int i = tb[j]; i < tb[j + 1] && !rc; ++i)
        for (int gj = 0; gj <= i; ++gj) {
          const int bj = blk_of(gj);
          if (!own(bj)) continue;
          if (y_upd[(size_t)i * nt + gj] != B - 1 - j) fail(80);
          if (bj < j ? !z_done[(size_t)i * nt + gj] : !zd_done[(size_t)i * nt + gj]) fail(81);
        }
  if (stats) {
    stats[0] = ran;
    stats[1] = waits;
    stats[2] = nincs;
    stats[3] = mutate_in > 0 ? mutated : 0;
  }
  gpp_dag_free(pl);
  return rc;
}
