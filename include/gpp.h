/*
 * gpp.h — C ABI of libgpp_hip.so, the MI355X (gfx950) native back end of the GP+ exact-GP hot path.
 *
 * The reference (Bostanabad-Research-Group/GP-Plus @ 2024_08_07) has no FFI of its own: the path
 *   optim/mll_torch.py:112-117   output = model(*train_inputs); loss = -mll(output, y); loss.backward()
 * runs entirely inside gpytorch/ATen.  Each entry point below replaces the ATen/gpytorch operator(s)
 * that the named reference call site reaches (SURVEY.md §2.1 "implicit device-op inventory", rows K1-K8).
 * The Python host in gp-plus_amd/ binds these with ctypes (gp-plus_amd/_lib.py); INTEGRATION.md shows the
 * stub a GP+ maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors), except where noted "host";
 *  - matrices are row-major fp64 with an explicit leading dimension `ld` (elements).  The covariance and its Cholesky
 *    factor keep the UPPER triangle authoritative (Ky = U^T U, U row-major == L = U^T column-major, LAPACK 'L' on a
 *    column-major array); the strict lower triangle of that buffer is never read or written.  The inverse factor
 *    buffer `Linv` holds L^-1 in its lower triangle and the mirror image L^-T in its strict upper triangle; Kinv keeps
 *    its LOWER triangle authoritative.  (This mix makes every O(N^3) product a row-contiguous "TN" GEMM.)
 *  - N x N matrix pointers must be 16-byte aligned and `ld` even (the kernels use 16-byte accesses);
 *  - all work is enqueued on the handle's stream (gpp_set_stream) and is asynchronous to the host;
 *  - return value: 0 ok, <0 bad argument (-(index of the argument)), >0 HIP runtime error code + 1000;
 *    numerical failure of the factorisation is reported LAPACK-style through `info_dev` (device int32:
 *    0 = ok, k>0 = leading minor k not positive definite), which the caller reads after syncing;
 *  - the library never allocates or frees user-visible memory; scratch comes from the workspace the caller
 *    attaches with gpp_set_workspace (size from gpp_workspace_bytes).
 */
#ifndef GPP_H
#define GPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpp_handle_s* gpp_handle_t;

/* kernel family of gpp_kernel_build / gpp_cross_kernel / gpp_grad_reduce (`kind`) */
#define GPP_KIND_RBF      0 /* sf2*exp(-sum_d w_d (u_id-u_jd)^2): gpytorch RBFKernel (models/gp_plus.py:223-253),
                               kernels/Rough_RBF.py:27-32, product of RBFs (models/gp_plus.py:297-303) */
#define GPP_KIND_MATERN32 1 /* kernels/matern.py:4-5 on the dims >= d_split, RBF on dims < d_split */
#define GPP_KIND_MATERN52 2 /* kernels/matern.py:7-8 */

#define GPP_UPLO_FULL  0
#define GPP_UPLO_LOWER 1    /* only tiles that intersect the lower triangle are written */
#define GPP_UPLO_UPPER 2    /* only tiles that intersect the upper triangle are written (what gpp_potrf reads) */

/* operation ids for gpp_workspace_bytes */
#define GPP_OP_MLL_EVAL 0   /* potrf + trtri + lauum + mll_reduce + grad_reduce for an N-point model */
#define GPP_OP_PREDICT  1   /* gpp_predict with M test points */

const char* gpp_version(void);

int gpp_create(gpp_handle_t* out, int device);
int gpp_destroy(gpp_handle_t h);
/* `stream` is a hipStream_t (passed as void* so this header needs no HIP include). */
int gpp_set_stream(gpp_handle_t h, void* stream);
/* The look-ahead factorisation inside gpp_potrf(_ws) runs on internal streams of the handle, created on first use: a
 * latency stream on 32 CUs (one per shader engine), a throughput stream and a fill stream on the other 224, and one
 * stream without a CU mask for the bulk of each trailing update once the diagonal block it ran beside is done; the
 * calling stream waits for all of them before the entry point's work is complete in stream order.
 * The first two are exposed here, with their disjoint CU sets: which = 0 the latency stream (32 CUs,
 * one per shader engine) on which gpp_potrf_ws factors diagonal blocks, which = 1 the throughput stream (the other CUs)
 * of its trailing updates, which = 2 the stream without a CU mask.  A host-side driver that overlaps its own panel
 * factorisations with its own updates (the sharded evaluation, gp-plus_amd/sharded.py) enqueues on them through
 * gpp_set_stream.  *out receives a hipStream_t. */
int gpp_internal_stream(gpp_handle_t h, int which, void** out);
/* Per-handle switches.  GPP_OPT_COOP_PANEL (default 1, or 0 with GPP_COOP_PANEL=0 in the environment): factor diagonal blocks and
 * small matrices with the cooperative panel kernel, whose work-groups wait for each other on the device and therefore must all be
 * resident together.  When several handles (threads or processes) share one GPU two such launches can each hold part of the same
 * CUs; a wait inside the kernel then gives up after GPP_OPT_PANEL_TIMEOUT_MS (default 500) milliseconds of the device's 100 MHz
 * constant clock — wall time, independent of the core clock and of how slow a poll is beside other tenants — and the factorisation
 * reports *info = GPP_INFO_PANEL_TIMEOUT + the milliseconds the abandoned wait had waited (low 20 bits) — the
 * caller switches the option off and factors again (gp-plus_amd/linalg.py does).  GPP_OPT_PANEL_FAULT (default 0): the NEXT
 * panel launch reports that time-out without running (one shot; for tests of the caller's recovery).
 * No reference counterpart: the reference's factorisation is torch.linalg.cholesky_ex behind gpytorch (optim/mll_torch.py:116). */
#define GPP_OPT_COOP_PANEL 1
#define GPP_OPT_PANEL_FAULT 2
#define GPP_OPT_PANEL_TIMEOUT_MS 3
#define GPP_OPT_EXEC_SCHED 4 /* (round 4's statically scheduled executor, replaced by the DAG executor: an alias of GPP_OPT_DAG_SCHED) */
#define GPP_OPT_DAG_SCHED 5  /* default 1 (0 with GPP_DAG_SCHED=0): factorisation (and, for 6912 <= N <= GPP_DAG_INV_MAX = 19456, the whole inverse
                              * beside it) as ONE list of tile tasks in topological order that persistent work-groups take by atomic
                              * ticket (gpp_dag.hip, gpp_dag_f64) — needs no co-residency of its work-groups; the diagonal blocks still
                              * run as cooperative panel launches (GPP_OPT_COOP_PANEL) */
#define GPP_INFO_PANEL_TIMEOUT (1 << 30)
/* a wait of one of the DAG executor's work-groups (gpp_dag_f64, its gate kernels) timed out — status =
 * GPP_INFO_EXEC_TIMEOUT + milliseconds: the caller switches GPP_OPT_DAG_SCHED off and factors again; the
 * cooperative panel itself (GPP_INFO_PANEL_TIMEOUT without the second bit) stays on unless it times out on its own */
#define GPP_INFO_EXEC_TIMEOUT ((1 << 30) | (1 << 29))
int gpp_set_option(gpp_handle_t h, int option, int value);
size_t gpp_workspace_bytes(gpp_handle_t h, int op, int64_t N, int64_t M, int D, int S);
int gpp_set_workspace(gpp_handle_t h, void* ws, size_t bytes);

/*
 * K1-K4 fused (models/gp_plus.py:472-474 covar_module(x_new).evaluate();
 * gpytorch RBFKernel/ProductKernel/ScaleKernel; likelihoods_noise/multifidelity.py:63-67 K + diag(noise)):
 *   Ky[i,j] = sf2 * k(U_i, U_j; w) + (i==j) * (tau[grp[i]] + jitter)      for row0 <= i < row0+nrows
 * U: N x D row-major; w: D weights; sf2: 1 double (device); tau: S doubles or NULL; grp: N int32 or NULL
 * (NULL => group 0); d_split: number of leading dims that stay RBF when kind != RBF.
 */
int gpp_kernel_build(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                     const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split,
                     int uplo, double* Ky, int64_t ld, int64_t row0, int64_t nrows);

/* K8 cross block (models/gpregression.py:126 self(x) -> test/train covariance): Kab[a,b] = sf2*k(Ua_a, Ub_b; w) */
int gpp_cross_kernel(gpp_handle_t h, const double* Ua, int64_t Ma, const double* Ub, int64_t Nb, int D,
                     const double* w, const double* sf2, int kind, int d_split, double* Kab, int64_t ld);

/*
 * K5 (gpytorch psd_safe_cholesky -> torch.linalg.cholesky_ex reached from optim/mll_torch.py:116):
 * in-place Cholesky of A (N x N, UPPER triangle): A = U^T U.  Right-looking in steps of one 128 x 128 leaf (factored
 * AND inverted in LDS by one work-group) below 6144 rows, block rows of 1024 / 512 with look-ahead on two CU-masked
 * streams above; every product runs on the fp64 MFMA GEMM kernel.  On return the upper triangle of A holds U = L^T and
 * the 128-aligned diagonal blocks of Linv hold inv(L_bb) (lower) mirrored with inv(L_bb)^T (upper), as gpp_trtri needs
 * them.
 */
int gpp_potrf(gpp_handle_t h, double* A, int64_t N, int64_t ld, double* Linv, int64_t ldi, int32_t* info_dev);
/* Same, with an N x N scratch T (may be the Kinv buffer): the driver then also completes the inverse of every diagonal block it
 * factors and uses it to solve each block row with GEMMs; the following gpp_trtri on the same handle skips the merges that are
 * already done.  By size (round 5; gpp_api.hip):
 *   3840 <= N < 6912    look-ahead with launches on two CU-masked streams, the WHOLE inverse by bordering on a third beside it;
 *   6912 <= N <= 65536  the DAG executor (gpp_dag.hip, gpp_dag_f64): factorisation — and up to N = 19456 the whole inverse, above
 *                       that its leading 2^j x 1024-row block — as ONE list of tile tasks in topological order that persistent
 *                       work-groups take by atomic ticket, the diagonal blocks as cooperative panel launches behind gate kernels
 *                       (22.75 -> 20.6 ms per evaluation at N = 10000, 61.6 -> 59.9 at 15000, 131.0 -> 129.2 at 20000);
 *   N > 65536           look-ahead with launches (the bulk of each update on a stream without a CU mask).
 * Linv is complete on return wherever the inverse was built beside the factorisation (gpp_trtri then launches nothing). */
int gpp_potrf_ws(gpp_handle_t h, double* A, int64_t N, int64_t ld, double* Linv, int64_t ldi, double* T, int64_t ldt,
                 int32_t* info_dev);

/* Completes Linv = inv(L) (lower triangle, mirrored into the upper) from the diagonal-block inverses left by
 * gpp_potrf.  `U` is the factored matrix (upper).  T (N x N) is scratch. */
int gpp_trtri(gpp_handle_t h, const double* U, int64_t N, int64_t ld, double* Linv, int64_t ldi, double* T, int64_t ldt);

/* Kinv(lower) = Linv^T Linv.  With gpp_trtri this is K7's "K_y^-1" (ATen cholesky_backward, optim/mll_torch.py:117). */
int gpp_lauum(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk);

/* Trailing update of a block-row-cyclic sharded factorisation, one launch per rank and step:
 *   C(upper triangle of Nt x Nt) -= Urow^T Urow   on the block rows (height nb, a multiple of 128) this rank owns,
 * block row i of C (0-based) being global block row first_block + i, owned when (first_block + i) % nranks == rank.
 * Urow: K x Nt (the just-factored block row, right of its diagonal block), C: the trailing corner of the matrix. */
int gpp_syrk_rows(gpp_handle_t h, const double* Urow, int64_t ldu, double* C, int64_t ldc, int64_t Nt, int64_t K, int64_t nb,
                  int64_t first_block, int rank, int nranks);

/* One right-looking step of the sharded evaluation's BACK-substitution ("each GPU solves for its own
 * block-columns of L^-T L^-1 E_k"): C(lower triangle of M x M) = beta C + alpha A^T B on the column blocks (width nb, a multiple
 * of 128) this rank owns — column block i of C (0-based) is global block first_block + i, owned when
 * (first_block + i) % nranks == rank; other ranks' column blocks are neither read nor written.  A: K x M (rows of the factor's
 * mirror L = U^T), B: K x M (the just-solved block row of Ky^-1), both row-contiguous.  Only the rows [row0, row1) of C are produced
 * (row0 a multiple of 128): the block row the next step depends on is issued first, the rest behind it.  Replaces the broadcast of the inverse's
 * column blocks + the LAUUM share (gpp_lauum_rows_range); reference counterpart: ATen cholesky_backward, optim/mll_torch.py:117.
 * compact != 0 (here, in gpp_trmv_lower_cols and in gpp_grad_reduce_cols): B and C — there T, Kinv — hold ONLY the owned column
 * blocks, side by side (the q-th owned block, global block rank + q nranks, in columns [q nb, (q+1) nb) of the buffer, ld >= the
 * owned width): N x (N / nranks) doubles per rank and matrix instead of N x N — what lets 8 GPUs hold a problem one cannot
 * ("28.8 GB (3.6 GB/GPU sharded)"). */
int gpp_gemm_lower_cols(gpp_handle_t h, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                        int64_t M, int64_t K, double alpha, double beta, int64_t nb, int64_t first_block, int rank, int nranks,
                        int64_t row0, int64_t row1, int compact);

/*
 * The sharded evaluation's factorisation AND forward sweep of one rank as ONE ticket list of tile tasks (the executor of
 * gpp_potrf_ws, gp-plus_amd/csrc/gpp_dag.hip, with the block-cyclic ownership built into the list): the rank's panels (diagonal blocks
 * k % nranks == rank: factor in place in A, inverse + mirror into D[k]), row solves (through the scratch rows W0..W2, nb x N, ldw),
 * its share of the trailing updates, and — right-looking beside them — the rows below the diagonal of ITS column blocks of
 * X = L^-1 into Kc (compact, as in gpp_gemm_lower_cols; Lc: the running sums).  Block rows of other ranks arrive as messages the
 * CALLER moves on a stream of its own (RCCL broadcasts in gp-plus_amd/sharded.py): per block row k a head (diagonal block, D[k], the
 * columns of block k + 1) and a tail (the columns behind, in pieces: gpp_shard_piece_cols).  The owner enqueues gpp_shard_list_gate in
 * front of packing each from A; a receiver enqueues gpp_shard_list_signal behind unpacking each into A / D.  Replaces the per-step launches of
 * gp-plus_amd/sharded.py::_factor and ::_forward (reference counterpart: torch.linalg.cholesky + the solves of optim/mll_torch.py:114-117;
 * the reference has no multi-GPU evaluation).  *used = 0: not applicable here (size, block height, options) — nothing was
 * enqueued and the caller runs its launch-per-product path.  workers: work-groups of the executor (0 = two per throughput CU; tests
 * that share one GPU between ranks pass fewer).  Between _begin and _end the handle runs nothing else.  A wait that exceeds
 * GPP_SHARD_TIMEOUT_MS (default 60 000: it covers the other ranks' progress) sets GPP_INFO_EXEC_TIMEOUT in *info.
 */
int gpp_shard_list_begin(gpp_handle_t h, int64_t N, int64_t nb, int rank, int nranks, double* A, int64_t ld, double* Kc, double* Lc,
                         int64_t ldc, double* D, double* W0, double* W1, double* W2, int64_t ldw, int32_t* info, int workers,
                         int* used);
/* Messages of block row k, in the order every rank moves them on its communication stream: the HEAD (message 0), then the TAIL in
 * pieces of gpp_shard_piece_cols() columns (messages 1 + g: the columns [(k + 2) nb + g W, min((k + 2) nb + (g + 1) W, N)), while that
 * range is not empty; W = 8192, or GPP_SHARD_PIECE_COLS, a multiple of 128; 0 = the whole tail as ONE piece [(k + 2) nb, N)).  Round 6:
 * a task waits for the piece of the column tile it reads, so the owner of block row k + 1 updates, solves and sends its first piece
 * while the rest of block row k is still on the wire.  `tail` in gpp_shard_list_gate / _signal is the message number. */
int64_t gpp_shard_piece_cols(void);
int gpp_shard_list_gate(gpp_handle_t h, void* stream, int tail, int k);
int gpp_shard_list_signal(gpp_handle_t h, void* stream, int tail, int k);
int gpp_shard_list_end(gpp_handle_t h);

/* The sharded evaluation's BACK-substitution of one rank as one ticket list on the handle's stream: the owned column blocks of L^-1 in
 * Kc (compact; rows at and below each block's diagonal, destroyed) become the same column blocks of Ky^-1 = L^-T L^-1 in Lc, against
 * the factor's mirror L in the strict lower triangle of A and the diagonal blocks' inverses D — gp-plus_amd/sharded.py::_backward's
 * per-step launches (gpp_gemm_batched + gpp_gemm_lower_cols) as tile tasks in dependency order, far tiles taking several steps'
 * updates at once.  No communication (bullet 4).  *used = 0: not applicable, nothing enqueued. */
int gpp_shard_back_list(gpp_handle_t h, int64_t N, int64_t nb, int rank, int nranks, const double* A, int64_t ld, double* Kc, double* Lc,
                        int64_t ldc, const double* D, int32_t* info, int workers, int* used);

/*
 * The WHOLE sharded evaluation behind the C ABI (SURVEY.md §8(b) sketched `gpp_set_comm(handle, ncclComm_t, rank, nranks)`): what
 * gp-plus_amd/sharded.py drives from Python — build of the owned block rows, the rank's ticket list with the block rows' messages
 * around it, z / alpha, the back-substitution's list, the gradient reduction — as ONE call per rank, for C / Fortran / MPI callers.
 * The collectives are the caller's (gpp_set_comm: two callbacks, each enqueued on or ordered against the given stream; they may
 * block the host) or RCCL's own (gpp_comm_init_rccl: librccl.so is opened at run time; rank 0 creates the 128-byte id with
 * gpp_comm_unique_id and hands it to the other ranks by any means).  gpp_shard_eval (reference counterpart: `mll(output, y)` +
 * `loss.backward()`, optim/mll_torch.py:114-117) returns 0 with *info_host = the status every rank agrees on: 0, LAPACK's "leading
 * minor not positive definite" (the caller adds jitter and calls again — gpytorch's policy stays with the caller), or a time-out's
 * bits; GPP_SHARD_UNSUPPORTED when the lists do not apply (N < 4096, a block height the panel does not take, a last block of <= 256
 * rows): nothing useful was computed.  On success b->out3 = {quad, logdet, mll}, b->alpha = Ky^-1 r, and with need_grad b->flat =
 * {g_w[D], g_sf2, g_tau[S], g_U[N x dU]} summed over the ranks (the layout of gpp_grad_reduce).  The buffers are the caller's
 * (sizes: gpp_shard_buffer_doubles); b->r = y - mean on entry; the handle's scratch workspace (gpp_set_workspace,
 * GPP_OP_MLL_EVAL) must be set.
 */
enum { GPP_COMM_SUM_F64 = 0, GPP_COMM_MAX_I32 = 1 };
#define GPP_SHARD_UNSUPPORTED 2000
typedef struct gpp_comm {
  void* user;
  int (*bcast)(void* user, void* dev_buf, size_t bytes, int root, void* stream);          /* in place, from rank `root` */
  int (*allreduce)(void* user, void* dev_buf, size_t count, int kind, void* stream);      /* in place; kind: GPP_COMM_* */
} gpp_comm_t;
typedef struct gpp_shard_buffers {
  double* A; int64_t ld;                    /* N x N: the factor (upper) and its mirror (strict lower) */
  double* Kc; double* Lc; int64_t ldc;      /* N x (owned blocks x nb) each: column blocks of L^-1, then of Ky^-1 (in Lc) */
  double* D;                                /* nblk x nb x nb: the diagonal blocks' inverses */
  double* W0; double* W1; double* W2; int64_t ldw;  /* three nb x N scratch rows */
  double* msg;                              /* nb x (N + 2 nb): a message being packed / unpacked */
  double* z; double* alpha; double* r;      /* N each */
  double* flat;                             /* D + 1 + S + N dU */
  double* out3; int32_t* info;              /* 3 doubles; 2 ints */
} gpp_shard_buffers_t;
int gpp_set_comm(gpp_handle_t h, const gpp_comm_t* comm, int rank, int nranks);
int gpp_comm_unique_id(void* out128);
int gpp_comm_init_rccl(gpp_handle_t h, const void* unique_id128, int rank, int nranks);
/* which: 0 A, 1 Kc / Lc (each), 2 D, 3 W0 / W1 / W2 (each), 4 msg — with ld = ldw = N rounded up to 16, ldc = owned blocks x nb.
 * The sizes of A, Kc / Lc and W include 128 doubles of slack behind the last row: the tile kernels read (never write) up to the next
 * multiple of 128 columns past N along a row, and a buffer that ends exactly on a page boundary must not fault there.  (The same holds
 * for the N x N operands of gpp_potrf_ws & co. when a caller allocates them with hipMalloc instead of a pooling allocator.) */
size_t gpp_shard_buffer_doubles(int64_t N, int64_t nb, int rank, int nranks, int which);
int gpp_shard_eval(gpp_handle_t h, int64_t N, int64_t nb, const double* U, int D, const double* w, const double* sf2, const double* tau,
                   const int32_t* grp, int S, int kind, int d_split, double jitter, int dU, int need_grad, const gpp_shard_buffers_t* b,
                   int32_t* info_host);

/*
 * A direct one-to-all PUSH of the sharded evaluation's messages, the alternative to a broadcast collective (SURVEY.md:204, :423-425:
 * "owner pushes the same panel on all 7 links concurrently"); gp-plus_amd/csrc/gpp_push.hip has the protocol.  Every rank creates a
 * channel (two message slots of slot_bytes + a flag page, exported as a GPP_PUSH_HANDLE_BYTES record), the ranks exchange the records
 * by any means (all-gather) and connect (hipIpcOpenMemHandle of every peer's).  Messages are numbered seq = 1, 2, ... in the one order
 * in which every rank moves them.  For message seq, on every rank, in this order:
 *   its owner:   gpp_push_send  — behind what `after_stream` has enqueued so far, on one stream per peer: wait until the peer has
 *                consumed message seq - 2, copy part i (height[i] rows of width[i] bytes at src[i], row pitch spitch[i]: straight
 *                from where it lies, no packing) to byte `offset[i]` of the peer's slot with row pitch dpitch[i], then mark the slot
 *                complete; `after_stream` continues when the message has left.
 *   the others:  gpp_push_recv  — on `stream`: wait until message seq is complete in this rank's slot, then copy part i from byte
 *                offset[i] of the slot (row pitch spitch[i]) to dst[i] (row pitch dpitch[i]).
 *   every rank:  gpp_push_ack   — on `stream`, behind whatever consumed the message: tell every peer that this rank is done with seq.
 * A wait that exceeds GPP_SHARD_TIMEOUT_MS ORs `code` into *status (device memory) and gives up.  gpp_push_destroy: stage 0 drains and
 * unmaps the peers' memory, stage 1 — after every rank's stage 0 — frees this rank's; stage 2 does both.
 */
typedef struct gpp_push* gpp_push_t;
#define GPP_PUSH_HANDLE_BYTES 128
int gpp_push_create(int device, int rank, int nranks, int64_t slot_bytes, gpp_push_t* out, void* handle_out);
int gpp_push_connect(gpp_push_t p, const void* handles /* nranks x GPP_PUSH_HANDLE_BYTES, in rank order */);
int gpp_push_send(gpp_push_t p, void* after_stream, int64_t seq, int32_t* status, int code, int nparts, const void* const* src,
                  const int64_t* spitch, const int64_t* offset, const int64_t* dpitch, const int64_t* width, const int64_t* height);
int gpp_push_recv(gpp_push_t p, void* stream, int64_t seq, int32_t* status, int code, int nparts, void* const* dst, const int64_t* dpitch,
                  const int64_t* offset, const int64_t* spitch, const int64_t* width, const int64_t* height);
int gpp_push_ack(gpp_push_t p, void* stream, int64_t seq);
int gpp_push_info(gpp_push_t p, int64_t* slot_bytes, int* flag_kind /* 0 uncached, 1 fine-grained, 2 ordinary device memory */);
int gpp_push_destroy(gpp_push_t p, int stage);

/* Products with a lower-triangular T of which only the block-cyclically owned COLUMN blocks (width nb, a multiple of 64; block
 * b owned when b % nranks == rank) exist on this rank:  trans = 0: y_i = sum over owned columns k <= i of T[i][k] x_k (this
 * rank's part of z = L^-1 r);  trans = 1: y_k = sum_{i >= k} T[i][k] x_i for the owned columns k and 0 for the others (this
 * rank's entries of alpha = L^-T z).  The caller sums the results of all ranks (one all-reduce of N doubles each). */
int gpp_trmv_lower_cols(gpp_handle_t h, const double* T, int64_t ldt, int64_t N, const double* x, double* y, int64_t nb, int rank,
                        int nranks, int trans, int compact);

/* out3 = { quad = z'z, logdet = 2 sum log U_ii, mll = -0.5*(quad + logdet + N log 2pi) } from a z the caller already holds
 * (the second half of gpp_mll_reduce; optim/mll_torch.py:116). */
int gpp_mll_scalars(gpp_handle_t h, const double* U, int64_t ld, int64_t N, const double* z, double* out3);

/* The share of gpp_lauum owned by one rank of a sharded evaluation: the 128-row tile rows t of Kinv with
 * t % nranks == rank, in ONE launch (cyclic at tile granularity: every rank gets the same mix of short and long rows).
 * Linv must be complete on this rank; the other tile rows of Kinv are not touched. */
int gpp_lauum_rows(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk, int rank, int nranks);

/* The part of that share inside the rows [row0, row1) of Kinv (row0 a multiple of 128): what one rank can form as soon as
 * the column blocks 0 .. row1 of the inverse have arrived — Kinv[i, j <= i] needs the columns i and j of Linv only — so the
 * product is pipelined with the broadcasts of the inverse's column blocks (gp-plus_amd/sharded.py).  The reference has no
 * counterpart (single-process ATen cholesky_backward, optim/mll_torch.py:117). */
int gpp_lauum_rows_range(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, double* Kinv, int64_t ldk, int rank,
                         int nranks, int64_t row0, int64_t row1);

/* dst[c][r] = src[r][c] (rows x cols, out of place, 64 x 64 tiles through LDS): the mirror L^-T of a column block of the inverse
 * that arrived from another rank (gp-plus_amd/sharded.py); on the single-GPU path the mirror is written by the GEMM epilogue. */
int gpp_transpose(gpp_handle_t h, const double* src, int64_t lds, int64_t rows, int64_t cols, double* dst, int64_t ldd);

/*
 * K6 (gpytorch MultivariateNormal.log_prob -> inv_quad_logdet, optim/mll_torch.py:116):
 *   z = Linv r;  out3 = { quad = z'z, logdet = 2 sum log U_ii, mll = -0.5*(quad + logdet + N log 2pi) }
 */
int gpp_mll_reduce(gpp_handle_t h, const double* U, int64_t ld, const double* Linv, int64_t ldi, int64_t N,
                   const double* r, double* z, double* out3);

/* alpha = Linv^T z = Ky^-1 r (gpytorch prediction_strategy mean_cache; dMLL/dmean) */
int gpp_alpha(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* z, double* alpha);

/*
 * K7 reduction (autograd backward of K1-K4, optim/mll_torch.py:117), with W = 0.5*(alpha alpha' - Kinv):
 *   g_w[d]   = sum_ij W_ij dKy_ij/dw_d        g_sf2 = sum_ij W_ij K_ij / sf2
 *   g_tau[s] = sum_{i in s} W_ii              g_U[i,d] = dMLL/dU_id for d < dU (dU may be 0)
 * K_ij is recomputed from U in registers; only the lower triangle of Kinv is read.
 */
int gpp_grad_reduce(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                    const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                    int64_t ldk, int dU, double* g_w, double* g_sf2, double* g_tau, double* g_U);

/*
 * The same reduction restricted to the block rows of Kinv this rank owns in a sharded evaluation (SURVEY.md §8(e) mode 2;
 * BASELINE config 5): rows [b*nb, (b+1)*nb) with b % nranks == rank (nb a multiple of 64).  Only those rows of Kinv are
 * read; the outputs are this rank's PARTIAL sums (the caller all-reduces them).  nranks == 1 is gpp_grad_reduce.
 */
int gpp_grad_reduce_rows(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                         const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                         int64_t ldk, int dU, int64_t nb, int rank, int nranks, double* g_w, double* g_sf2, double* g_tau,
                         double* g_U);

/* The same reduction over the block-cyclically owned COLUMN blocks of Kinv's lower triangle (columns [b*nb, (b+1)*nb) with
 * b % nranks == rank): what a rank holds after the sharded back-substitution (gpp_gemm_lower_cols). */
int gpp_grad_reduce_cols(gpp_handle_t h, const double* U, int64_t N, int D, const double* w, const double* sf2,
                         const int32_t* grp, int S, int kind, int d_split, const double* alpha, const double* Kinv,
                         int64_t ldk, int dU, int64_t nb, int rank, int nranks, double* g_w, double* g_sf2, double* g_tau,
                         double* g_U, int compact);

/*
 * K8 (models/gpregression.py:122-149 predict): V = Ksn Linv^T (M x N, scratch, may be NULL to skip var),
 *   mean_out[a] = sum_j Ksn[a,j] alpha[j],   var_out[a] = kss[a] - sum_j V[a,j]^2
 */
int gpp_predict(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* alpha, const double* Ksn,
                int64_t lds, int64_t M, const double* kss, double* V, int64_t ldv, double* mean_out, double* var_out);

/* The same prediction from the TRANSPOSED cross block Kns (N x M: gpp_cross_kernel with the training points as rows) and
 * z = Linv r (gpp_mll_reduce):  V = Kns^T Linv^T as a row-contiguous TN product against the mirror in Linv's upper triangle,
 * mean_out[a] = sum_j V[a,j] z[j],  var_out[a] = kss[a] - sum_j V[a,j]^2.  The form to use when the variance is wanted (the product
 * is ~1.7x faster than gpp_predict's); for the mean alone gpp_predict with V = NULL is O(M N).  (models/gpregression.py:122-149) */
int gpp_predict_tn(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t N, const double* z, const double* Kns, int64_t ldk,
                   int64_t M, const double* kss, double* V, int64_t ldv, double* mean_out, double* var_out);

/*
 * The fp64 MFMA GEMM behind all of the above, exported for the parity tests:
 *   C = beta*C + alpha*op(A)*op(B),  op(A): M x K, op(B): K x N, all row-major.
 * transA: 0 = A stored M x K, 1 = A stored K x M;  transB: 0 = B stored K x N, 1 = B stored N x K.
 * Supported (transA,transB): (0,1) "NT", (0,0) "NN", (1,0) "TN".
 * a_mask/b_mask: 0 none, 1 keep entries with k <= row, 2 keep entries with k >= row (row = m for A, n for B);
 * klo_mode: 0 -> 0, 1 -> tile_m*128, 2 -> tile_n*128, 3 -> max of both;  khi_mode: 0 -> K, 1 -> (tile_m+1)*128,
 * 2 -> (tile_n+1)*128;  c_tri: 0 full output, 1 only entries with n <= m, 2 only entries with n >= m (M == N).
 */
int gpp_gemm(gpp_handle_t h, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int a_mask, int b_mask,
             int klo_mode, int khi_mode, int c_tri);

/* `batch` independent products of the same shape in one launch: element b uses A + b*sA, B + b*sB, C + b*sC (strides in
 * elements, even; sA = 0 shares A).  Used by the sharded inverse, whose column blocks sit at a regular column spacing. */
int gpp_gemm_batched(gpp_handle_t h, int transA, int transB, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
                     int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB, double beta, double* C, int64_t ldc,
                     int64_t sC, int batch, int a_mask, int b_mask, int klo_mode, int khi_mode, int c_tri);

/*
 * Batched evaluation: `batch` independent problems of the SAME size N in every launch (the restarts of a multistart
 * fit, optim/mll_torch.py:99-141, evaluated together instead of one after the other).  Element b uses the matrices at
 * base + b*s? (strides in elements, even), parameters w + b*D, sf2 + b, tau + b*S, vectors r/z/alpha + b*sv (sv >= N,
 * even), out3 + 3*b, info_dev + b, and returns g_w + b*D, g_sf2 + b, g_tau + b*S, g_U + b*N*dU.  sU = 0 shares one
 * feature matrix.  N is at most 6144 for gpp_potrf_batched (the leaf-step
 * factorisation); gpp_trtri_batched completes the inverse from the 128-blocks, so it follows gpp_potrf_batched
 * directly.  The workspace must hold batch * gpp_workspace_bytes(GPP_OP_MLL_EVAL, ...).
 */
int gpp_kernel_build_batched(gpp_handle_t h, const double* U, int64_t sU, int64_t N, int D, const double* w, const double* sf2,
                             const double* tau, const int32_t* grp, int S, double jitter, int kind, int d_split, int uplo,
                             double* Ky, int64_t ld, int64_t sK, int batch);
int gpp_potrf_batched(gpp_handle_t h, double* A, int64_t N, int64_t ld, int64_t sA, double* Linv, int64_t ldi, int64_t sLi,
                      int32_t* info_dev, int batch);
int gpp_trtri_batched(gpp_handle_t h, const double* U, int64_t N, int64_t ld, int64_t sA, double* Linv, int64_t ldi,
                      int64_t sLi, double* T, int64_t ldt, int64_t sT, int batch);
int gpp_lauum_batched(gpp_handle_t h, const double* Linv, int64_t N, int64_t ldi, int64_t sLi, double* Kinv, int64_t ldk,
                      int64_t sK, int batch);
int gpp_mll_reduce_batched(gpp_handle_t h, const double* U, int64_t ld, int64_t sA, const double* Linv, int64_t ldi,
                           int64_t sLi, int64_t N, const double* r, double* z, int64_t sv, double* out3, int batch);
int gpp_alpha_batched(gpp_handle_t h, const double* Linv, int64_t ldi, int64_t sLi, int64_t N, const double* z, double* alpha,
                      int64_t sv, int batch);
int gpp_grad_reduce_batched(gpp_handle_t h, const double* U, int64_t sU, int64_t N, int D, const double* w, const double* sf2,
                            const int32_t* grp, int S, int kind, int d_split, const double* alpha, int64_t sv,
                            const double* Kinv, int64_t ldk, int64_t sK, int dU, double* g_w, double* g_sf2, double* g_tau,
                            double* g_U, int batch);

#ifdef __cplusplus
}
#endif
#endif /* GPP_H */
