// gpp_internal.h — declarations shared by the HIP translation units of libgpp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/gpp.h"

#define GPP_TILE 128 /* GEMM work-group tile edge and Cholesky leaf size */
/* Plans the DAG executor keeps per handle (LRU).  A sharded evaluation holds two (factor + forward sweep, back-substitution) beside
 * the single-GPU plan of the same size, and a driver that plays every rank of a P = 8 run on one handle (tools/replay_rank.py) 16:
 * evicting a plan costs a re-plan (~1.2 s of host time at N = 60 000) and its hipFree calls synchronise the device. */
#define GPP_DAG_PLANS 20

struct gpp_handle_s {
  int device;
  hipStream_t stream;        // caller's stream (gpp_set_stream)
  void* ws;
  size_t ws_bytes;
  hipStream_t panel_stream;  // internal stream of the look-ahead Cholesky: diagonal-block factorisations (lazy)
  hipStream_t upd_stream;    // internal stream of the look-ahead Cholesky: wide trsm / trailing updates (lazy)
  hipStream_t full_stream;   // internal stream of the look-ahead Cholesky WITHOUT a CU mask: the part of a trailing update
                             // that runs after the next diagonal block is done (no leaf to starve) gets all 256 CUs (lazy)
  hipStream_t fill_stream;   // internal stream of the look-ahead Cholesky: bordering steps of the inverse (lazy; CUs of upd_stream)
  int panel_cus;             // CUs of panel_stream when cu_split == 1
  int cu_split;              // 1: the two streams own disjoint CU sets (CU masks), 0: plain priority streams, -1: unknown
  hipEvent_t events[16];     // ring of timing-disabled events for the two-stream hand-offs (lazy)
  int n_events, ev_next;
  // diagonal blocks whose inverse the last gpp_potrf_ws call completed (look-ahead path with scratch): gpp_trtri
  // skips the pair merges that lie inside one of them
  int64_t inv_N;             // N of that factorisation (0: nothing recorded)
  int inv_nblocks;
  int64_t inv_o[128], inv_n[128];
  // flag blocks of the cooperative panel launches (gpp_leaf.hip), used round-robin: a launch finds its block zeroed and the last
  // work-group to leave zeroes it again; no more than PANEL_RING panels are ever in flight on one handle, and a handle is used from
  // ONE stream at a time (gpp_set_stream orders a new stream behind the old one)
  char* panel_flags;
  int panel_next;
  // Panel launches recorded into a stream CAPTURE take their block from a second ring of GPP_PANEL_CAP_RING blocks that eager
  // launches never touch: a captured graph keeps its block for life, so two graphs captured on one handle (up to the ring's size
  // apart) may be replayed on different streams at the same time, and a replay never meets an eager launch's block.
  int cap_next;
  hipEvent_t handoff;        // gpp_set_stream: work enqueued on the previous stream is ordered before work on the new one
  int ncu;                   // CUs of the device: a panel launch never has more work-groups than its stream's CUs hold
  int coop_panel;            // GPP_OPT_COOP_PANEL
  int panel_fault;           // GPP_OPT_PANEL_FAULT: the next panel launch only reports the time-out status (tests)
  int panel_timeout_ms;      // GPP_OPT_PANEL_TIMEOUT_MS: budget of a wait inside the panel kernel (100 MHz constant clock)
  struct DagPlan* dag_plans[GPP_DAG_PLANS];  // LRU of DAG-executor plans (gpp_dag.hip), keyed by (N, nb, leading dimensions, flags, rank, ranks)
  uint64_t dag_clock;
  struct DagPlan* shard_cur;  // the sharded list between gpp_shard_list_begin and _end
  int32_t* shard_info;
  hipEvent_t shard_ready;     // counters cleared + groups bound: what the caller's communication stream waits for before a gate / signal
  int dag_sched;             // GPP_OPT_DAG_SCHED
  // gpp_shard.hip: the collectives of gpp_shard_eval (the caller's callbacks, or RCCL opened at run time) and its own stream
  gpp_comm_t comm;
  int comm_rank, comm_nranks;
  void* rccl_comm;
  hipStream_t comm_stream;
  hipEvent_t comm_event;
};
extern "C" void gpp_shard_release_comm(gpp_handle_s* h);
constexpr int GPP_PANEL_RING = 8;
constexpr int GPP_PANEL_CAP_RING = 16;

// ---- exp for the covariance kernels --------------------------------------------------------------
// exp(x) for x <= 0 (every argument of the path is -r^2 or -sqrt(.)): k = rint(x log2 e), r = x - k ln 2 in two parts (Cody-Waite,
// k ln2_hi exact), e^r by a degree-12 Horner polynomial on |r| <= ln2 / 2 (truncation 1.7e-16), scaled by 2^k with v_ldexp_f64
// (exact down to the denormals).  ~19 fp64 instructions and no branches against ~55 instructions (range checks, special cases for
// positive / huge arguments, constants re-materialised per call) of the general-purpose library exp, which made the N^2
// kernels VALU-bound instead of HBM-bound.  Max error 2 ulp on [-745, 0] (tools/attic/exp_check.py); NaN propagates; x < -800 -> 0.
#ifdef __HIPCC__
// The constants live in SGPRs for the whole kernel (wave-uniform; one constant-bus operand per v_fma_f64): left to itself the
// compiler re-materialises each coefficient in a VGPR pair per use (two v_mov_b32 in front of every v_fmac_f64, ~26 per call —
// as many issue slots as the arithmetic).  The empty asm makes them opaque, i.e. not re-materialisable.
struct GppExpConsts {
  double log2e, ln2_hi, ln2_lo, c[11];
};
__device__ __forceinline__ GppExpConsts gpp_exp_consts() {
  GppExpConsts k = {1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
                    {2.08767569878680989792e-09,    // 1/12!
                     2.50521083854417187751e-08,    // 1/11!
                     2.75573192239858906526e-07,    // 1/10!
                     2.75573192239858906526e-06,    // 1/9!
                     2.48015873015873015873e-05,    // 1/8!
                     1.98412698412698412698e-04,    // 1/7!
                     1.38888888888888888889e-03,    // 1/6!
                     8.33333333333333333333e-03,    // 1/5!
                     4.16666666666666666667e-02,    // 1/4!
                     1.66666666666666666667e-01,    // 1/3!
                     0.5}};
  asm volatile("" : "+s"(k.log2e), "+s"(k.ln2_hi), "+s"(k.ln2_lo));
#pragma unroll
  for (int i = 0; i < 10; ++i) asm volatile("" : "+s"(k.c[i]));
  return k;
}
__device__ __forceinline__ double gpp_exp_nonpos(double x, const GppExpConsts& k) {
  x = (x < -800.0) ? -800.0 : x;  // (a select, not a max: NaN stays NaN)
  const double n = __builtin_rint(x * k.log2e);
  double r = __builtin_fma(-n, k.ln2_hi, x);
  r = __builtin_fma(-n, k.ln2_lo, r);
  double p = k.c[0];
#pragma unroll
  for (int i = 1; i < 11; ++i) p = __builtin_fma(p, r, k.c[i]);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}
#endif

// ---- fp64 MFMA GEMM (gpp_gemm.hip) ------------------------------------------------------------
struct GemmArgs {
  const double* A;
  const double* B;
  double* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int a_mask, b_mask;      // 0 none, 1 keep k<=row, 2 keep k>=row
  int klo_mode, khi_mode;  // see gpp.h
  int c_lower;             // 0 full, 1 lower triangle only (n <= m), 2 upper triangle only (n >= m)
  int64_t sA, sB, sC;      // batch strides in elements (grid.y = batch)
  int64_t zA, zB, zC, zC2; // second-level batch strides (grid.z = batch2; 0 batch2 means 1): independent problems
  int batch2;
  int tiles_m, tiles_n;
  double* C2;              // optional mirrored output: C2[n][m] = C[m][n]
  int64_t ldc2, sC2;
  int k_reverse;           // walk the k range from its top down (see gpp_gemm.hip)
  int tag;                 // 1: launch the separately named instantiation (profiling label, same code)
  int col_major;           // enumerate tiles column by column (non-triangular outputs)
  int row_reverse;         // row-major order, last row tile first
  int swz;                 // set by the launcher: XCD-aware 8x8 super-tile mapping of blockIdx -> tile
  int own_mod, own_off, own_bt;  // c_lower == 2: produce the tile rows tm with (tm / own_bt + own_off) % own_mod == 0
                           // (block-cyclic block rows of own_bt tile rows; own_mod <= 1: all).  Other tiles exit at once.
                           // c_lower == 1 (plain enumeration): the same test on the tile COLUMN tn (block-cyclic column blocks)
  int skip_lead;           // c_lower == 2 only: tiles lying entirely inside the leading skip_lead x skip_lead block exit at once
                           // (that block was updated by an earlier launch; skip_lead a multiple of the tile)
  int row_limit;           // c_lower == 2 only: produce only the tiles whose rows lie below row_limit (a multiple of the tile;
                           // 0: all): the upper TRAPEZOID rows [0, row_limit) x columns [row, N) in one launch — other tiles
                           // of the triangular enumeration exit at once
  int batch_fast, nbatch;  // batch_fast: one grid dimension, tile-major with the batch element as the FAST index (nbatch of them),
                           // so that tiles of equal K length of all batch elements run together and the launch ends on the
                           // shortest tiles of every element (pair merges of gpp_trtri) instead of on the last element's longest
  int row_i0, row_i1;      // c_lower == 1 only: of the owned tile rows (index i = 0, 1, ... in increasing order) produce those
                           // with row_i0 <= i < row_i1 (row_i1 == 0: all): the sharded LAUUM, block row by block row as the
                           // column blocks of the inverse arrive
  int64_t tile_base;       // set by the launcher: first tile index of this launch in the owned-row enumeration
  int row_t0, row_t1;      // c_lower == 1, plain enumeration: produce the tile rows row_t0 <= tm < row_t1 only (row_t1 == 0: all)
  int cu_hint;             // CUs of the stream the launch goes to (0: all 256): scales the automatic tile choice
  int row_mod, row_off;    // c_lower == 1 only: produce the tile rows tm with tm % row_mod == row_off (row_mod <= 1: all);
                           // the sharded LAUUM, one launch per rank over its cyclic share of the 128-row tile rows
  int op;                  // DAG executor only (gpp_dag_f64): 0 = the product above, 1 = copy the M x 128 strip `tn` of B into C
  int compact_bc;          // c_lower == 1 with own_mod > 1: B and C hold only the owned column blocks, side by side (the q-th owned
                           // block in columns [q own_bt 128, (q+1) own_bt 128) of their buffers): the sharded back-substitution
                           // with N x (N / ranks) storage per rank
  int etile;               // DAG executor only: work-group tile of this group's tasks — 0 / 128 the big tile, 64 the small one
  int buf[4];              // DAG executor only (gpp_dag_f64): A, B, C, C2 hold BYTE OFFSETS into the launch's operand bases
                           // buf[0..3] (C2 unused: buf[3] < 0), so that a plan does not depend on the operands' addresses
  int pad_ok;              // TN variant, big tile: the operands may be READ up to the next multiple of 128 past M / N along their rows
                           // (the bytes belong to the same allocation: never set for an operand's last row of a buffer) — what
                           // is read there only reaches output entries that are not stored, and a ragged edge tile then runs the
                           // lean staging path instead of the predicated one (measured 367 -> ~250 us per K = 1024 tile)
};
// variant: 0 = NT (A[m][k], B[n][k]), 1 = NN (A[m][k], B[k][n]), 2 = TN (A[k][m], B[k][n])
// tile_m = 0: choose a square tile from the grid size; else force the work-group tile (128x128, 64x64, 32x32, 128x32)
hipError_t gpp_launch_gemm(hipStream_t s, int variant, const GemmArgs& a, int batch, int tile_m = 0, int tile_n = 0);

// one-wave kernels on a stream (gpp_gemm.hip): wait until counters[id] >= target — a wait that exceeds `budget` ticks of the 100 MHz
// clock sets the abort word counters[0] and *info = GPP_INFO_EXEC_TIMEOUT + ms —, and counters[id] += 1 behind a release fence
hipError_t gpp_launch_exec_gate(hipStream_t s, int* counters, int id, int target, int32_t* info, long long budget);
hipError_t gpp_launch_exec_signal(hipStream_t s, int* counters, int id);

// ---- DAG executor (gpp_gemm.hip: gpp_dag_f64; planned by gpp_dag.hip) -----------------------------------------------------------
// Round 5.  ONE list of tile tasks in a topological order of the whole task graph (factorisation, and optionally the triangular
// inverse built right-looking beside it); work-groups take the next task with an atomic ticket, wait for its (up to three)
// counters, run the tile and raise its (up to two) counters.  Only RUNNING work-groups hold tasks and each holds one, so the
// earliest unfinished task is always held by a running work-group whose predecessors are complete: progress needs NO co-residency
// of the grid (round 4's static per-worker lists did), faster work-groups simply take more tasks (the older / younger wave asymmetry
// balances itself), and the same list serves any number of workers — including short filler launches on the panel's CUs.
// counters[0] is the abort word, counters[1] the ticket.
struct DagTask {           // 48 bytes
  int32_t group;           // index into the groups array
  int16_t tm, tn;          // tile of that group's product (in units of the group's etile), or the strip of a copy
  int32_t wait_id[3];      // counters to wait for (-1: none) ...
  int32_t wait_val[3];     // ... until they are >= these values
  int32_t inc_id[2];       // counters to raise once the task's stores are visible (-1: none) ...
  int32_t kind;            // semantic class (gpp_dag.hip: DK_*), for traces and the host-side checker
  int16_t inc_val[2];      // ... by these amounts (a fused task that applies f steps' updates raises its tile's version by f)
};
static_assert(sizeof(DagTask) == 48, "the device reads 48-byte task records");
struct DagLaunch {
  const GemmArgs* groups;  // the ABSOLUTE copy of the plan's groups (gpp_launch_dag_bind)
  const DagTask* tasks;
  int ntasks;
  int* counters;
  int32_t* info;
  long long budget;        // ticks of the 100 MHz constant clock a single wait may take
  int max_tasks;           // > 0: a work-group leaves after this many tasks (filler launches between two panels)
  int quit_id, quit_val;   // quit_id >= 0: a work-group takes no further task once counters[quit_id] >= quit_val
  int ticket_limit;        // > 0: a work-group takes no ticket >= this (filler launches: the first task that needs the NEXT panel)
  unsigned long long* trace;  // debug: 4 words per task — ticket taken, waits over, done (100 MHz clock), work-group | launch tag << 32
  int tag;
};
hipError_t gpp_launch_dag(hipStream_t s, int nworkers, const DagLaunch& e);
struct DagBases { char* p[8]; };  // operand bases the groups' byte offsets refer to — single GPU: A, Linv, T; sharded list
                                  // (DAG_SHARD): A, Kc, Lc, D, W0, W1, W2 (gp-plus_amd/sharded.py's buffers)
hipError_t gpp_launch_dag_bind(hipStream_t s, const GemmArgs* rel, GemmArgs* abs, int n, const DagBases& bases);

enum { DK_S = 0, DK_U = 1, DK_CP = 2, DK_XB = 3, DK_XA = 4, DK_SH = 5, DK_UD = 6, DK_NKINDS = 7 };
struct DagTuning {
  double t0_big, tc_big;      // us of a 128 x 128 tile task: t0 + tc * (K / 16)
  double t0_64, tc_64;        // 64 x 64 tile
  double t0_32, tc_32;        // 32 x 32 tile
  double t_copy;              // strip copy
  double t_panel0, t_panel_leaf;  // panel launch: t_panel0 + t_panel_leaf * leaves, plus
  double t_gate;              // gate + signal hand-offs around it
  int chain_tile;             // tile of the chain's tasks (head solve, next diagonal block's update): 128 or 64
  int workers;                // workers the order is simulated for
  int fill;                   // filler work-groups per launch (0: no filler launches)
  int fuse;                   // far tiles take the updates of up to `fuse` (1, 2 or 4) consecutive steps in ONE task with K = fuse x nb
                              // (one prologue, one read-modify-write of the tile and one ticket instead of `fuse`)
  int64_t inv_rows;           // DAG_INV: rows of the inverse built inside the list — >= N: all of it; a multiple of nb below N: only
                              // the leading inv_rows x inv_rows block (gpp_trtri then merges the rest around it)
  int64_t piece_cols;         // DAG_SHARD: the tail of a block row's message travels in pieces of this many columns (a multiple of 128;
                              // 0: one piece) — see DagPlan::piece_tiles
};
struct DagPlan {
  int64_t N = 0, nb = 0, ld = 0, ldi = 0, ldt = 0, ldk = 0, inv_rows = 0;
  int rank = 0, nranks = 1, workers = 0, fill = 0;            // DAG_SHARD: block-cyclic owner of block k is k % nranks; ldi = the compact buffers' leading
                                       // dimension, ldt = the scratch rows'
  int c_cph = 0, c_cpt = 0, c_art = 0; // DAG_SHARD: first ids of "head copied" / "tail piece copied" (gates of the owner's broadcasts) and
                                       // "tail piece arrived" (signalled behind a received broadcast; the head's arrival raises c_pd + k)
  // Round 6: the tail of block row k (the columns behind block k + 1) travels in PIECES of piece_tiles column tiles, piece g = the
  // column tiles [tb[k + 2] + g piece_tiles, ...): counters c_cpt + k GP + g and c_art + k GP + g, GP = the pieces of block row 0.  A
  // task that reads block row k of another rank waits for the piece of its LATER column tile only (messages arrive in order), so the
  // owner of block row k + 1 updates, solves and sends ITS first piece while the rest of block row k is still on the wire: the
  // chain per step is the transfer plus ONE piece's update + solve instead of the whole row's (measured with tools/replay_rank.py).
  int piece_tiles = 0, GP = 1;
  std::vector<int> cph_target, cpt_target;  // cpt_target[k GP + g]
  int flags = 0;                       // DAG_INV: also the inverse (right-looking; all of it or its leading inv_rows block)
  int B = 0, nt = 0;
  std::vector<int> tb;                 // first tile of block b (tb[B] = nt)
  std::vector<GemmArgs> groups;
  std::vector<DagTask> tasks;          // in ticket order
  struct Op { int kind, arg, n, lim; };  // panel stream: 0 gate(b) 1 panel(b) 2 signal(b) 3 filler launch (arg = tasks per work-group,
                                       // n = the block whose gate ends it, lim = first ticket it must not take)
  std::vector<Op> stream_ops;
  std::vector<int> gate_target;
  std::vector<int> level_first;        // per block b: the first ticket whose task needs the panel of block b or of a later one
  int ncounters = 0;
  int c_pd = 0, c_g1d = 0;             // first ids of the panel-done / gate counters (index by block)
  double sim_ms = 0, sim_busy = 0;     // the planner's own estimate (makespan, mean worker occupancy)
  GemmArgs* d_groups = nullptr;        // as planned: operands as (buffer, byte offset)
  GemmArgs* d_groups_abs = nullptr;    // what the executor reads: rewritten by gpp_dag_bind in front of every launch
  DagTask* d_tasks = nullptr;
  int* d_counters = nullptr;
  unsigned long long* d_trace = nullptr;
  hipEvent_t last_use = nullptr;       // recorded behind the launches that read the device copies
  uint64_t stamp = 0;                  // LRU
};
enum { DAG_INV = 1, DAG_SHARD = 4, DAG_BACK = 8 };  // DAG_SHARD: one rank's list of the sharded evaluation (factor + forward sweep of its column blocks)
DagPlan* gpp_dag_plan(int64_t N, int64_t nb, int64_t ld, int64_t ldi, int64_t ldt, int64_t ldk, int flags, const DagTuning& tune,
                      int rank = 0, int nranks = 1);
DagTuning gpp_dag_default_tuning();
hipError_t gpp_dag_upload(DagPlan* P);
void gpp_dag_free(DagPlan* P);

// ---- 128x128 diagonal leaf: Cholesky + triangular inverse in LDS (gpp_leaf.hip) ---------------
// A holds the UPPER factor (A = U^T U, i.e. L = U^T read/written with swapped indices); the n x n diagonal block of
// Linv receives inv(L) in its lower triangle and the mirror image inv(L)^T in its strict upper triangle.
// batch > 1: independent blocks at A + b*sA, Linv + b*sLi, info + b (one work-group each).
// p[0..n) = value by a kernel.  Status words are cleared inside evaluations that may be captured into a HIP graph, and this stack's
// replayed memset nodes are not to be trusted (gpp_leaf.hip, the panel's flag block).
hipError_t gpp_launch_fill_i32(hipStream_t s, int32_t* p, int n, int32_t value);
// cooperative panel (gpp_leaf.hip): factor AND invert a diagonal block of n = 128 C rows in one launch
size_t gpp_panel_flag_bytes();
int gpp_panel_max_leaves();
hipError_t gpp_launch_panel(hipStream_t s, double* A, int64_t lda, double* Linv, int64_t ldi, int n, int32_t* info, int row_offset,
                            int* flags, int max_wgs, int timeout_ms);
hipError_t gpp_launch_leaf(hipStream_t s, double* A, int64_t lda, double* Linv, int64_t ldi, int n, int32_t* info,
                           int row_offset, int batch = 1, int64_t sA = 0, int64_t sLi = 0);

// ---- covariance tiles (gpp_build.hip) ---------------------------------------------------------
// batch > 1: independent parameter sets b: U + b*sU (sU = 0 shares the features), w + b*D, sf2 + b, tau + b*S, Ky + b*sK.
hipError_t gpp_launch_kernel_build(hipStream_t s, const double* U, int64_t N, int D, const double* w, const double* sf2,
                                   const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split,
                                   int uplo, double* Ky, int64_t ld, int64_t row0, int64_t nrows, int batch = 1,
                                   int64_t sU = 0, int64_t sK = 0);
hipError_t gpp_launch_cross_kernel(hipStream_t s, const double* Ua, int64_t Ma, const double* Ub, int64_t Nb, int D,
                                   const double* w, const double* sf2, int kind, int d_split, double* Kab, int64_t ld);

// ---- reductions (gpp_reduce.hip) --------------------------------------------------------------
// batch > 1 (all reductions): matrices at + b*sT, vectors at + b*sv (sv even, >= N), out3 at + 3*b
hipError_t gpp_launch_trmv_lower(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                 int batch = 1, int64_t sT = 0, int64_t sv = 0);
hipError_t gpp_launch_trmv_upper(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                 int batch = 1, int64_t sT = 0, int64_t sv = 0);
hipError_t gpp_launch_mll_scalars(hipStream_t s, const double* L, int64_t ld, int64_t N, const double* z, double* out3,
                                  int batch = 1, int64_t sL = 0, int64_t sv = 0);
size_t gpp_grad_ws_bytes(int64_t N, int D, int S, int dU);  // per batch element
hipError_t gpp_launch_grad_reduce(hipStream_t s, const double* U, int64_t N, int D, const double* w, const double* sf2,
                                  const int32_t* grp, int S, int kind, int d_split, const double* alpha,
                                  const double* Kinv, int64_t ldk, int dU, double* g_w, double* g_sf2, double* g_tau,
                                  double* g_U, void* ws, size_t ws_bytes, int shard_nb = 0, int shard_rank = 0,
                                  int shard_nranks = 1, int batch = 1, int64_t sU = 0, int64_t sK = 0, int64_t sv = 0,
                                  int shard_cols = 0);
// (shard_cols = 1: the rank owns block-cyclic COLUMN blocks of Kinv's lower triangle instead of block rows; 2: and Kinv holds
//  only those blocks, side by side: the q-th owned block in columns [q nb, (q+1) nb))
// y = sum over the owned column blocks (width nb, block b owned when b % nranks == rank) of T(lower) x  (trans = 0), or
// y_k = sum_i T[i][k] x_i for the owned columns k and 0 elsewhere (trans = 1)
hipError_t gpp_launch_trmv_lower_cols(hipStream_t s, const double* T, int64_t ldt, int64_t N, const double* x, double* y,
                                      int64_t nb, int rank, int nranks, int trans, void* ws, size_t ws_bytes, int compact = 0);
size_t gpp_trmv_t_ws_bytes(int64_t N);  // scratch of the trans = 1 form (partial sums of the row chunks)
// (batch > 1: U + b*sU, w + b*D, sf2 + b, alpha + b*sv, Kinv + b*sK; outputs g_w + b*D, g_sf2 + b, g_tau + b*S,
//  g_U + b*N*dU; the workspace holds batch * gpp_grad_ws_bytes)
// dst[c][r] = src[r][c] for r < rows, c < cols (64 x 64 tiles through LDS)
hipError_t gpp_launch_transpose(hipStream_t s, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd);
hipError_t gpp_launch_predict_reduce(hipStream_t s, const double* Ksn, int64_t lds, const double* V, int64_t ldv,
                                     int64_t M, int64_t N, const double* alpha, const double* kss, double* mean_out,
                                     double* var_out);
