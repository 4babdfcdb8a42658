/*
 * tgp_hip.h -- C ABI of libtgp_hip.so: the MI355X (gfx950) dense DirectSolver hot path.
 *
 * This is the drop-in boundary for tinygp's dense `DirectSolver` path.  The reference
 * (dfm/tinygp) is pure Python on JAX and has NO native FFI of its own; the seam a
 * replacement binds to is the `Solver` protocol (src/tinygp/solvers/solver.py:15-82)
 * consumed by `GaussianProcess(..., solver=Cls)` (src/tinygp/gp.py:101-112).  Each entry
 * point below names the reference interface it replaces.  The ctypes binding a maintainer
 * would add is shown in INTEGRATION.md and lives in tinygp_amd/_ffi.py.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature
 *     (`void* stream` is a hipStream_t passed opaquely, NULL = library-owned stream);
 *   - device matrices are COLUMN-major (element (i,j) at j*ld + i); host matrices in the
 *     `tgp_solver_*` layer are ROW-major like the reference's (N,N) arrays
 *     (a row-major lower-triangular L is byte-identical to a column-major upper L^T, so
 *     the copy-out transposes);
 *   - every function returns an int status: 0 = ok, < 0 = TGP_E_* (see tgp_last_error()),
 *     > 0 = LAPACK-style `info` (1-based index of the first non-positive pivot).
 *     Numerical failure never aborts: the factor then holds NaNs exactly like
 *     jax.scipy.linalg.cholesky (direct.py:53) and log_probability becomes -inf (gp.py:316);
 *   - blocking unless stated: results in host memory are valid on return;
 *   - a ctx / solver handle is not re-entrant (one thread at a time).
 */
#ifndef TGP_HIP_H
#define TGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an exported signature or option changes; the Python binding refuses another version
 * (round 2 -> 3: tgp_trace_factor gained nb_wide_rows, ~15 entry points added, "potf2_sync" removed;
 *  round 3 -> 4: tgp_chain_stamps, tgp_chain_task, the resident-factor solves of the block-column driver -- tgp_dist_fwd_block,
 *  _bwd_block, _trmv_partial, _cross_cov, _colsumsq_owned, _gram_owned -- and tgp_dist_abort added;
 *  round 4 -> 5: tgp_comm_* (RCCL from the C ABI), tgp_stream_* transfers, TGP_E_TIMEOUT, options poll_timeout_ms /
 *  chain_fast_update / host_join, chain_polls = 3 (stream wait-value); tgp_dist_fwd_partial / _fwd_solve_left,
 *  tgp_dist_bwd_*_multi, tgp_dist_gather_owned, tgp_dist_identity_cols,
 *  tgp_dist_grad_* -- multi-RHS backward solve and gradient on the block-column path;
 *  round 5 -> 6: tgp_chain_tasks (the chain launch's task TABLE with its K-batched updates), tgp_dist_gram_pair_owned, tgp_dist_load_matrix,
 *  options chain_fwd_tasks, chain_batch /
 *  chain_batch_lag / chain_batch_rowlag / chain_batch_minrows, trsv_groups; tgp_solver_timings ms[7]) */
#define TGP_ABI_VERSION 6

/* element types: follows the dtype of the caller's arrays (gp.py:89) */
#define TGP_F32 0
#define TGP_F64 1

/* status codes */
#define TGP_OK 0
#define TGP_E_ARG (-1)     /* bad argument (maps to Python ValueError) */
#define TGP_E_HIP (-2)     /* HIP runtime error */
#define TGP_E_NOMEM (-3)   /* device allocation failed */
#define TGP_E_UNSUPPORTED (-4)
#define TGP_E_TIMEOUT (-5) /* a device-side hand-off did not arrive within poll_timeout_ms (not a numerical failure) */

/* all device matrices are padded to a multiple of TGP_TILE rows/cols */
#define TGP_TILE 128

/* ---- kernel program -------------------------------------------------------------
 * A tinygp kernel tree (kernels/base.py:170-209 Sum/Product/Constant over the stationary
 * leaves of kernels/stationary.py:59-235) flattened to postfix.  Leaves push a value,
 * ADD/MUL pop two and push one.  `metric` selects kernels/distance.py:41-59.
 *   CONST : p0 = value                               (base.py:205-209)
 *   EXP   : exp(-dist/p0)                            (stationary.py:76-82)
 *   EXPSQ : exp(-0.5 * sqdist/p0^2)                  (stationary.py:104-106)
 *   M32   : a = sqrt(3)*dist/p0; (1+a) exp(-a)       (stationary.py:126-129)
 *   M52   : a = sqrt(5)*dist/p0; (1+a+a^2/3) exp(-a) (stationary.py:150-153)
 *   COS   : cos(2 pi dist/p0)                        (stationary.py:173-175)
 *   ESS   : exp(-p1 sin^2(pi dist/p0))               (stationary.py:202-205)
 *   RQ    : (1 + 0.5 sqdist/p0^2 / p1)^(-p1)         (stationary.py:232-235)
 * dist / sqdist: L1 -> sum|d|, (sum|d|)^2 ; L2 -> zero-safe sqrt(sum d^2), sum d^2.
 */
enum {
  TGP_K_CONST = 0, TGP_K_EXP = 1, TGP_K_EXPSQ = 2, TGP_K_M32 = 3, TGP_K_M52 = 4,
  TGP_K_COS = 5, TGP_K_ESS = 6, TGP_K_RQ = 7, TGP_K_ADD = 16, TGP_K_MUL = 17
};
enum { TGP_METRIC_L1 = 0, TGP_METRIC_L2 = 1 };

typedef struct tgp_kop {
  int32_t op;
  int32_t metric;
  double p0;
  double p1;
} tgp_kop;

#define TGP_KPROG_MAX 32  /* ops per program */
#define TGP_KSTACK_MAX 8  /* evaluation-stack depth */
#define TGP_MAX_DIM 16    /* input dimension D */

typedef struct tgp_ctx tgp_ctx;
typedef struct tgp_solver tgp_solver;

/* ---- library / context ---------------------------------------------------------- */
int tgp_abi_version(void);
const char* tgp_last_error(void); /* thread-local message of the last failing call */

/* device: HIP ordinal; stream: hipStream_t to launch on (NULL: create an own stream) */
int tgp_ctx_create(int device, void* stream, tgp_ctx** out);
int tgp_ctx_destroy(tgp_ctx* ctx);
int tgp_ctx_sync(tgp_ctx* ctx);
/* tuning knobs; returns the previous value via *old:
 *   "nb_outer"          outer panel width of the blocked Cholesky (default 1024)
 *   "lookahead"         1: factor the next panel beside the trailing update (default), 0: off
 *   "first_small_tiles" look-ahead block-column updates of at most this many 128x128 tiles
 *                       run on 64x64 tiles (default 1100)
 *   "nb_wide_rows"      panels that start with at least this many rows left are 2 nb_outer wide
 *                       (default 30000; 0: never) -- never the first panel
 *   "solve_on_update"   1 (default): the forward-substitution steps fused into the factorisation are
 *                       queued on the update stream (three busy queues); 0: on their own stream
 *   "first_split"       blocks of a panel after which its share of the next block-column
 *                       update is issued early, beside the panel's last blocks (default 5; 0 off)
 *   "fused_step"        1: the panel chain is one launch per 128-column block (potf2, the rows' own
 *                       pending update and trsm behind a device-side flag: chol.hip,
 *                       panel_step_kernel); 0 (default): potf2 | trsm | in-panel update as separate
 *                       launches.  "gate_split" (with fused_step = 1): the block-column update
 *                       between two chains in three column pieces, the chain starts behind the first
 *   "chain_kernel"      1 (default): the panel chain is ONE persistent launch per panel: tile tasks behind a ticket
 *                       counter factor the diagonal blocks, solve the rows below and apply the in-panel updates
 *                       right-looking (one 128^3 task per tile and source column), hand-offs by per-tile state words
 *                       (chol.hip, chain_kernel); no update stream, no per-block launches or events.  0: rounds 1-3's
 *                       per-block launches.  With it:
 *                       "chain_full_rows" (4096): with at most this many rows left the whole rest is ONE launch;
 *                       "chain_depth2" (1; 2 = the main stream's calls first on the host; 0 = off): the chain
 *                       pipeline -- gate, potf2, chain of the next panel -- on the priority stream, the main stream
 *                       carries trailing updates only;  "chain_lds_pad" (10240): unused dynamic LDS that keeps a
 *                       second chain workgroup off a compute unit;  "chain_pre_wait" (0): pre(p) behind the next
 *                       panel's first potf2 on chain-bound panels;  "chain_polls" (1): consumers of a block column
 *                       (forward steps, early shares) follow behind a one-wave poll, bounded by wall clock, while the
 *                       launch runs; 3: behind hipStreamWaitValue32 (the runtime's own one-wave wait kernel, no timeout:
 *                       the host then joins the pass with a deadline); 0: behind the whole launch -- the DEFAULT of a
 *                       context created under a counter-collecting profiler (ROCPROF_COUNTER_COLLECTION, i.e.
 *                       rocprofv3 --pmc, or TGP_SERIALIZED_KERNELS=1), which runs kernels one at a time in its own order;
 *                       In the merged schedule ("chain_merged") the one poll that is left -- the next panel's chain behind
 *                       the PREFIX of the trailing update -- is the first thing that panel's potf2 launch does (1; no poll
 *                       kernel at all), a one-wave kernel of its own with 2, off with 0;
 *                       "chain_sub_panel" (0 = off; measured slower, profiles/r06_i): the chain of a panel runs in sub-panels
 *                       of this many columns, each applied to the panel's remaining columns by one product of the tiled
 *                       kernel ("chain_sub_role", 4) on the same stream; "chain_sub_min_rows": only for panels that tall;
 *                       "chain_fwd_tasks" (1, round 6): the fused forward substitution (gp.py:318-320) as TASKS of the chain
 *                       launch -- no poller and no forward-step launch at all, chain_polls then only concerns the early
 *                       shares of the non-default schedules; 0: round 5's followers on the solve stream;
 *                       "chain_fast_update" (0): fp64 update tasks on the 4x4x4 MFMA form with LDS-direct operands
 *                       (measured slower: DESIGN 4.2);  "chain_stamps" (0): tgp_chain_stamps below;
 *                       "chain_batch" (1 = off; measured slower, csrc/tgp_common.h), "chain_batch_lag" (1), "chain_batch_rowlag" (4), "chain_batch_minrows" (32): K-BATCHED updates
 *                       (round 6, fp64): tile (i, c) with i >= c + rowlag takes the updates from `batch` consecutive block
 *                       columns as ONE task (K = 128 batch, the tile read and written once) as long as the batch ends at
 *                       least lag + 1 columns in front of c and at least minrows row tiles are left behind it (the launch is
 *                       still throughput-bound); csrc/chain_tasks.h, tgp_chain_tasks
 *   "tile_band"         order of the MFMA products' output tiles over the workgroup ids: bands of this many tile rows, column
 *                       by column inside a band (default 8: the 64 tiles an XCD has resident share 8 + 8 operand panels
 *                       instead of 64 + 1 -- fabric traffic of a trailing-update launch 3.28 -> 1.78 GB at N = 16 384, same
 *                       time); 0: column by column over all rows (rounds 1-4).  csrc/tile_order.h, tgp_tile_order
 *   "poll_timeout_ms"   wall-clock bound of every device-side wait (default 4000; one value per process and device).  A
 *                       wait that expires makes the call return TGP_E_TIMEOUT; tgp_solver_factor* has then already
 *                       repeated the pass ONCE on the launch-per-block path ("timeout_retries" counts them; read-only).
 *                       "host_join" (1): passes that enqueued stream wait-values are joined by polling an event with a
 *                       deadline instead of hipStreamSynchronize;  "fault_inject" (0): TEST hook, 1 = the next such
 *                       deadline has passed at once
 *   "kmat_plain_div"    1: the straight-line assembly kernel takes r / l, r^2 / l^2 by the division sequence instead
 *                       of the bit-identical reciprocal + FMA form (kmat.hip, UDiv) -- the tests' switch
 *   "chain_reserve"     workgroup slots (of two per CU) that a trailing update running beside a panel
 *                       chain leaves to the chain's kernels (default 96; 0: the update fills the chip)
 *   "reserve_max_tiles" ... when the update has at most this many 128x128 tiles (default 1200)
 *   "sub_panel"         two-level panel: the in-panel rank-128 updates stay inside sub-panels of this many
 *                       columns (a divisor of nb_outer, >= 256) and each finished sub-panel updates the
 *                       panel's remaining columns with one K = sub_panel product (0: off);
 *                       "sub_panel_min_rows": only for panels with at least this many rows
 *   "nb_first"          width of the first panel, whose chain nothing hides (0: nb_outer)
 *   "split_tail"        1: the last, partly filled round of tiles of a trailing update is split along k
 *   "profile"           1: time the trailing-update launches with events (tgp_solver_timings)
 *   "late_join"         1 (default, round 6): a fused evaluation joins the device ONCE -- the potrf `info` comes back with the two
 *                       scalars behind the reductions (0: a join behind the factorisation and another behind the reductions)
 *   "chain_reduce"      1 (default, round 6): the fused evaluation's two reductions (sum z^2, sum log L_ii) are left by the chain
 *                       launches' forward-substitution tasks, block by block in a fixed order (0: two reduction launches)
 *   "stream_trsv"       1 (default): triangular solves on a resident factor run as ONE streaming
 *                       launch (chol.hip, trsv_fwd/bwd_stream_kernel); 0: one launch pair per block;
 *                       "trsv_groups" (0 = by size: 3 / 4 / 6): workgroups per block row of the forward launch -- G - 1
 *                       helpers stream the raw tiles, the primary applies W_b, tf_b = W_b L[b,b-1] and tf2_b = W_b L[b,b-2]
 *   "keep_grad_buffers" 1: tgp_solver_grad keeps its (N + 128) x N work matrix between calls whatever its
 *                       size (default: kept up to 4 GiB, released at once above) */
int tgp_ctx_set_option(tgp_ctx* ctx, const char* key, int64_t value, int64_t* old);
int tgp_ctx_get_option(tgp_ctx* ctx, const char* key, int64_t* value);
/* name (<=255 chars), CU count, memory bytes, clock kHz of the ctx's device */
int tgp_ctx_device_info(tgp_ctx* ctx, char* name, int name_len, int32_t* cus, int64_t* mem_bytes,
                        int32_t* clock_khz);

/* raw device buffers for hosts without a HIP binding (ctypes) */
int tgp_malloc(tgp_ctx* ctx, size_t bytes, void** dev);
int tgp_free(tgp_ctx* ctx, void* dev);
int tgp_memcpy_h2d(tgp_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int tgp_memcpy_d2h(tgp_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int tgp_memset(tgp_ctx* ctx, void* dst_dev, int byte, size_t bytes);

/* ---- device-pointer kernels (async on the ctx stream) ------------------------------ */

/* K1+K3: out[i,j] = k(X1[i], X2[j]) (+ diag[i] if i==j and diag != NULL), column-major,
 * replaces Kernel.__call__ (kernels/base.py:84-103) + Diagonal._add (noise.py:77-78).
 * X1 (n1,d), X2 (n2,d) row-major device arrays.  out is rows_out x cols_out (>= n1,n2);
 * the padding region is filled with the identity (1 on i==j, else 0).
 * lower_only != 0 (square, X1==X2): only 128-tiles on/below the diagonal are written. */
int tgp_kmat(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
             int32_t d, const void* X1, const void* X2, const void* diag, void* out, int64_t ld,
             int64_t rows_out, int64_t cols_out, int lower_only);

/* K2: out[i] = k(X[i], X[i]) -- Kernel.__call__(X) / evaluate_diag (base.py:59-66,85-93) */
int tgp_kdiag(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n, int32_t d,
              const void* X, void* out);

/* K9 fused: out[i] = sum_j k(X1[i], X2[j]) * v[j]; never materialises K(X1,X2).
 * Replaces Kernel.matmul (kernels/base.py:68-82) as used by gp.py:357. */
int tgp_kmat_gemv(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
                  int32_t d, const void* X1, const void* X2, const void* v, void* out);
/* the same for nv right-hand sides at once (Kernel.matmul with a 2-D y, kernels/base.py:82):
 * V (nv, n2) and out (nv, n1) row-major device arrays; every kernel value is evaluated once per
 * group of 8 columns */
int tgp_kmat_gemv_multi(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
                        int32_t d, const void* X1, const void* X2, const void* V, int64_t nv, void* out);

/* K4: in-place lower Cholesky of the column-major n x n matrix A (n % TGP_TILE == 0);
 * replaces jax.scipy.linalg.cholesky(K, lower=True) at solvers/direct.py:53.
 * Only the lower triangle is read/written.  *info (host) = 0 or the failing pivot.
 * Blocking (reads info back). */
int tgp_potrf(tgp_ctx* ctx, int dtype, int64_t n, void* A, int64_t ld, int32_t* info);

/* K5/K6: y <- L^-1 y (transpose=0) or L^-T y (transpose=1), single right-hand side;
 * replaces solve_triangular at solvers/direct.py:66-70 for y of shape (N,). */
int tgp_trsv(tgp_ctx* ctx, int dtype, int64_t n, const void* L, int64_t ld, int transpose,
             void* y);

/* K5 with R right-hand sides in transposed form: B (m x n, column-major, rows = RHS
 * index) <- B L^-T, i.e. row r of B becomes (L^-1 b_r)^T.  m % 64 == 0, n % TGP_TILE == 0.
 * Replaces solve_triangular(L, Ks) at solvers/direct.py:94. */
int tgp_trsm_right_lt(tgp_ctx* ctx, int dtype, int64_t m, int64_t n, const void* L, int64_t ldl,
                      void* B, int64_t ldb);

/* C (m x n) <- beta*C + alpha * A (m x k) * B(n x k)^T, all column-major; m,n % TGP_TILE == 0,
* k % 16 == 0; alpha,beta in {(-1,1),(1,0)}.  lower != 0: only tiles on/below the diagonal are
 * computed (same origin for A and B rows; entries strictly above the diagonal are unspecified).  The fp64/fp32 MFMA building block (K4 trailing update,
 * K10 = A^T A at solvers/direct.py:95). */
int tgp_gemm_nt(tgp_ctx* ctx, int dtype, int64_t m, int64_t n, int64_t k, double alpha,
                const void* A, int64_t lda, const void* B, int64_t ldb, double beta, void* C,
                int64_t ldc, int lower);

/* y (m) -= P (m x k, column-major) x (k), k % TGP_TILE == 0: the update half of a blocked
 * forward substitution (used by the multi-GPU forward solve, one block column at a time) */
int tgp_gemv_sub(tgp_ctx* ctx, int dtype, int64_t m, int64_t k, const void* P, int64_t ld,
                 const void* x, void* y);

/* K7/K8 reductions: *out_host = sum log L[i,i] over i < n   /   sum y[i]^2 over i < n */
int tgp_sum_log_diag(tgp_ctx* ctx, int dtype, int64_t n, const void* L, int64_t ld,
                     double* out_host);
int tgp_sum_squares(tgp_ctx* ctx, int dtype, int64_t n, const void* y, double* out_host);

/* f64/f32 MFMA issue-rate microbenchmark: returns measured TFLOP/s of
 * v_mfma_f64_16x16x4_f64 (dtype F64) / v_mfma_f32_16x16x4_f32 (F32) over all CUs. */
int tgp_ubench_mfma(tgp_ctx* ctx, int dtype, double* tflops_out);
/* general form: kind 0 = MFMA f64 16x16x4, 1 = MFMA f32 16x16x4, 2 = VALU v_fma_f64,
 * 3 = MFMA f64 + 4 VALU f64 FMA interleaved per wave, 4 = MFMA f64 4x4x4(4b);
 * blocks_per_cu 256-thread workgroups per CU; also returns shader cycles per wave-op. */
int tgp_ubench(tgp_ctx* ctx, int kind, int blocks_per_cu, double* tflops_out,
               double* cycles_per_op_out);

/* ---- solver handle: replaces tinygp.solvers.DirectSolver (solvers/direct.py:17-95) ---- */

/* Allocates the padded n_pad^2 matrix and uploads X (n,d) row-major and the noise diagonal
 * (n,) once (DirectSolver.__init__ args, direct.py:30-37).  Host pointers. */
int tgp_solver_create(tgp_ctx* ctx, int dtype, int64_t n, int32_t d, const void* X_host,
                      const void* noise_diag_host, tgp_solver** out);
int tgp_solver_destroy(tgp_solver* s);

/* Assemble K = k(X,X) + diag (or take cov_host (n,n) row-major if non-NULL, direct.py:50-52)
 * and factor in place (direct.py:53).  Re-callable with new hyper-parameters: the
 * optimiser loop of SURVEY 3.4.  *info = potrf info. */
int tgp_solver_factor(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                      int32_t* info);
/* tgp_solver_factor + the whole of GaussianProcess.log_probability (gp.py:126-138) in one
 * call: the forward substitution alpha = L^-1 resid runs block column by block column on a
 * side stream as soon as each column of L is final, i.e. underneath the factorisation.
 * resid_host (n,) = y - loc, or NULL to use the residual of tgp_solver_set_resid.
 * *logprob = -0.5 |alpha|^2 - sum log L_ii - n/2 log(2 pi)  (not clamped: NaN when info>0) */
int tgp_solver_factor_logprob(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                              const void* resid_host, int32_t* info, double* logprob);
/* replace the noise diagonal (n,) without re-uploading X */
int tgp_solver_set_noise(tgp_solver* s, const void* noise_diag_host);

/* DirectSolver.normalization (direct.py:61-64): sum log L_ii + n/2 log(2 pi) */
int tgp_solver_normalization(tgp_solver* s, double* out);
/* DirectSolver.solve_triangular (direct.py:66-70); y,x host (n,) if nrhs==1 else (n,nrhs)
 * row-major */
int tgp_solver_solve_tri(tgp_solver* s, int transpose, int64_t nrhs, const void* y_host,
                         void* x_host);
/* DirectSolver.dot_triangular (direct.py:72-73): out = L @ y */
int tgp_solver_dot_tri(tgp_solver* s, int64_t nrhs, const void* y_host, void* out_host);
/* fused GaussianProcess._get_alpha + _compute_log_prob (gp.py:313-320) on resid = y - loc;
 * resid_host == NULL re-uses the residual uploaded by tgp_solver_set_resid */
int tgp_solver_set_resid(tgp_solver* s, const void* resid_host);
int tgp_solver_logprob(tgp_solver* s, const void* resid_host, double* out);
/* GaussianProcess._condition (gp.py:330-334): alpha = K^-1 resid (host, (n,)) and log-prob */
int tgp_solver_alpha(tgp_solver* s, const void* resid_host, void* alpha_host, double* logprob);
/* Gradient of log_probability (what the reference's callers obtain with jax.value_and_grad,
 * docs/tutorials/quickstart.ipynb cell 4):  1/2 tr((alpha alpha^T - K^-1) dK/dtheta).
 * grad_params: 2*nops doubles, [2*i + q] = d ll / d (p0 if q==0 else p1) of program op i (0 for
 * ADD/MUL and unused p1); grad_noise_host (n,) = d ll / d noise_i = 1/2 (alpha_i^2 - K^-1_ii), or
 * NULL; alpha_host (n,) = K^-1 resid = d ll / d mean_i, or NULL.  Needs tgp_solver_factor first;
 * allocates ONE (n_pad + 128) x n_pad work matrix on first use (L^-1 in both orientations, then K^-1 in the lower one's
 * place; kept between calls up to 4 GiB, see option "keep_grad_buffers").
 * grad_logscale (d doubles, or NULL): d ll / d log s_q for a scaling x_q -> s_q x_q of input dimension q of the
 * coordinates the solver holds -- what transforms.Linear / Cholesky with per-dimension scales need (reference
 * transforms.py:39-133, kernels/stationary.py:41-43); one extra pass over K^-1 per dimension. */
int tgp_solver_grad(tgp_solver* s, const void* resid_host, double* logprob, double* grad_params,
                    void* grad_noise_host, void* alpha_host, double* grad_logscale);
/* mean = K(Xt, X) alpha (gp.py:353-357; K9 fused). Xt host (m,d) row-major */
int tgp_solver_cond_mean(tgp_solver* s, const tgp_kop* prog, int nops, int64_t m,
                         const void* Xt_host, const void* alpha_host, void* mean_host);
/* DirectSolver.condition (direct.py:75-95): out (m,m) row-major = Kss + diag(noise_t) - A^T A.
 * Xt_host NULL -> X itself (m = n).  var_only != 0: out is (m,) = diag of that matrix. */
int tgp_solver_condition_cov(tgp_solver* s, const tgp_kop* prog, int nops, int64_t m,
                             const void* Xt_host, const void* noise_t_host, int var_only,
                             void* out_host);
/* DirectSolver.covariance (direct.py:58-59): K recomputed (the factor overwrote it) */
int tgp_solver_covariance(tgp_solver* s, void* out_host);
/* DirectSolver.variance (direct.py:49,55-56): k(x_i,x_i) + diag_i */
int tgp_solver_variance(tgp_solver* s, void* out_host);
/* scale_tril (direct.py:28): (n,n) row-major, upper triangle zero */
int tgp_solver_get_factor(tgp_solver* s, void* L_host);
/* device pointers of the padded factor (column-major, ld = n_pad) for zero-copy hosts */
int tgp_solver_device_factor(tgp_solver* s, void** L_dev, int64_t* n_pad);
/* last-call timings measured with HIP events on the ctx stream when option "profile"=1:
 * ms[0]=assembly of the first panel's columns (the rest is assembled beside the factorisation),
 * ms[1]=potrf total, ms[2]=sum of the 128x128-tile trailing-update launches (event spans),
 * ms[3]=number of those launches, ms[4]=separate trsv pass (0 when fused), ms[5]=HOST time from the entry of the last
 * factorisation to its last enqueue (how far ahead of the device the submitting thread runs),
 * ms[6]=algorithmic flops of those launches, ms[7]=length of the UNION of their intervals (launches of the main and the
 * priority stream overlap at large N: their sum can exceed the evaluation, the union cannot) */
int tgp_solver_timings(tgp_solver* s, double* ms, int n);

/* ---- block-cyclic column Cholesky over the GPUs of one node (BASELINE configs 4 / 5) ----------
 * The reference has NO multi-device code; these entry points are one RANK's share
 * of the same path -- `DirectSolver.__init__` (solvers/direct.py:49-53: assembly + cholesky),
 * `GaussianProcess.log_probability` (gp.py:126-138, 313-320) and the posterior mean of
 * `GaussianProcess._condition` (gp.py:330-334, 353-359) -- for a matrix cut into block columns
 * of width nb owned cyclically (block column j on rank j mod world).  One process per GPU; the
 * collectives (RCCL panel broadcast, slice broadcast of the backward solve, one all-reduce of
 * the (M,) mean) are issued by the host between these calls (tinygp_amd/distributed.py), which
 * is also where the order of calls per step is documented.  Every call is asynchronous on the
 * context's streams except tgp_dist_end / tgp_dist_cond_mean_partial / tgp_dist_get_column.
 *
 * ring0..ring2 (tgp_dist_slot_elems(n, nb) elements each) and x (n_pad = ceil(n/nb)*nb elements)
 * are device buffers of the CALLER (torch tensors, so that torch.distributed can send them):
 *   ring slot of panel k = ring[k % 3] = [(n_pad - k nb) x nb panel, column-major, ld = rows |
 *                                         (nb/128)*2048 inverse 16x16 diagonal blocks]
 *   (the panel first: a column chunk -- and the last chunk with the inverses -- is ONE contiguous message)
 *   x: the replicated right-hand side: residual -> L^-1 r (after tgp_dist_end) -> K^-1 r.
 * Per step k the host calls (tinygp_amd/distributed.py, BlockCyclicCholesky.factor):
 *   priority stream   owner of k+1: [RCCL wait for panel k] tgp_dist_lookahead(k), then per column chunk c
 *                     tgp_dist_panel_chunk(k+1, c, nch) + broadcast of the chunk; receivers of k+1:
 *                     tgp_dist_slot_ready(k+1) + the same broadcasts
 *   main stream       [RCCL wait for panel k] tgp_dist_arrived(k), tgp_dist_fwd_step(k),
 *                     tgp_dist_pre_update(k), tgp_dist_rest(k). */
typedef struct tgp_dist tgp_dist;
int64_t tgp_dist_slot_elems(int64_t n, int64_t nb);
int tgp_dist_create(tgp_ctx* ctx, int dtype, int64_t n, int32_t d, const void* X_host,
                    const void* noise_diag_host, int64_t nb, int32_t world, int32_t rank,
                    void* ring0_dev, void* ring1_dev, void* ring2_dev, void* x_dev, tgp_dist** out);
int tgp_dist_destroy(tgp_dist* h);
/* INSTEAD of tgp_dist_assemble: this rank's block columns of a host matrix (n x n row-major, symmetric, noise included) --
 * the seam's `covariance=` argument and non-diagonal noise (reference solvers/direct.py:36,50-52) on the block-column path */
int tgp_dist_load_matrix(tgp_dist* h, const void* K_host);
/* hipStream_t of the context: which = 0 main (updates), 1 panel (chain + pack); the host makes
 * its RCCL calls wait on / be waited on by these */
int tgp_dist_stream(tgp_dist* h, int which, void** stream_out);
/* K(X, X) + noise for the owned block columns (kernels/base.py:84-103, noise.py:77-78) */
int tgp_dist_assemble(tgp_dist* h, const tgp_kop* prog, int nops);
/* start of a factorisation; resid_host (n,) != NULL: also forward-substitute it (gp.py:318-320) */
int tgp_dist_begin(tgp_dist* h, const void* resid_host);
/* every rank: assembly of its remaining block columns (behind tgp_dist_panel_chunk(0, 0, ..) on the owner of
 * panel 0, so that it hides beside the first chain) */
int tgp_dist_first_panel(tgp_dist* h);
/* owner of panel k: column chunk c of nch (nch divides nb / 128) of its chain (potf2 / trsm / in-panel updates,
 * priority stream) and the pack of that chunk into the ring slot (the last chunk packs the inverses too) */
int tgp_dist_panel_chunk(tgp_dist* h, int64_t k, int64_t c, int64_t nch);
/* receiver of panel k: the priority stream waits until the slot's previous panel (k - 3) has been read */
int tgp_dist_slot_ready(tgp_dist* h, int64_t k);
/* owner of panel k+1, priority stream (already waiting for panel k's arrival): panel k applied to block
 * column k+1, behind tgp_dist_pre_update(k-1) */
int tgp_dist_lookahead(tgp_dist* h, int64_t k);
/* panel k has arrived and the main stream waits for it: the marker tgp_dist_fwd_step(k) depends on */
int tgp_dist_arrived(tgp_dist* h, int64_t k);
/* forward-substitution step k of the replicated right-hand side (solvers/direct.py:66-70 on the
 * received panel) and sum log L_ii of panel k (direct.py:61-64), on the update stream */
int tgp_dist_fwd_step(tgp_dist* h, int64_t k);
/* main stream, owner of block column k+2 only: panel k applied to that block column first (the next gate waits
 * for this launch, not for the big update: the chain pipeline runs two panels ahead of the updates) */
int tgp_dist_pre_update(tgp_dist* h, int64_t k);
/* update of all other owned block columns (> k+2) by panel k: one fp64/fp32 MFMA launch, which leaves
 * workgroup slots free when a chain of this rank runs beside it */
int tgp_dist_rest(tgp_dist* h, int64_t k);
/* joins the streams; *info = this rank's potrf info (host: MIN over ranks of the non-zero
 * ones), *sumsq = |L^-1 r|^2, *logdet_half = sum log L_ii (direct.py:61-64), identical on
 * every rank */
int tgp_dist_end(tgp_dist* h, int32_t* info, double* sumsq, double* logdet_half);
/* backward substitution (solve_triangular(..., trans=1), direct.py:68; gp.py:334), block k, on
 * its owner: x_k <- L_kk^-T (x_k - L[below, k]^T x[below]); the host then broadcasts x_k */
int tgp_dist_bwd_step(tgp_dist* h, int64_t k);
/* this rank's share of K(X*, X) alpha over its owned columns (gp.py:353-359 via
 * kernels/base.py:68-82), fused, into out_dev (m,); the host all-reduces it */
int tgp_dist_cond_mean_partial(tgp_dist* h, const tgp_kop* prog, int nops, int64_t m,
                               const void* Xt_host, void* out_dev);
/* ---- solves on the RESIDENT distributed factor (a new right-hand side costs O(N^2), not a factorisation) ------
 * Buffers are caller-owned DEVICE memory (the host reduces / broadcasts slices of them with RCCL): nrhs == 1: vectors
 * of n_pad entries; nrhs a multiple of 128: (n_pad, nrhs) ROW-major -- block k of all right-hand sides is ONE
 * contiguous chunk of nb * nrhs entries, and read column-major it is X_k^T, the MFMA kernels' operand.
 *
 * forward (solve_triangular(y), solvers/direct.py:66-70; A = L^-1 Ks, :94), fan-in: per block k the host REDUCES
 * block k of every rank's accumulator to the owner of k (north_star's reduce of the solve RHS over xGMI: nb * nrhs
 * entries per block), then the owner:  x_k = L_kk^-1 (y_k + acc_k),  acc[rows below] -= L[rows below, k] x_k.
 * No-op on the other ranks.  acc starts at zero; x is zero outside the owned blocks (one all-reduce replicates it). */
int tgp_dist_fwd_block(tgp_dist* h, int64_t k, int64_t nrhs, const void* y_dev, void* acc_dev, void* x_dev);
/* backward (solve_triangular(y, transpose=True), direct.py:68), block k on its owner, on ANY device vector of n_pad
 * entries: x_k <- L_kk^-T (x_k - L[below, k]^T x[below]); the host then broadcasts x_k */
int tgp_dist_bwd_block(tgp_dist* h, int64_t k, void* x_dev);
/* this rank's share of dot_triangular (direct.py:72-73): out = sum over the owned block columns of L[:, k] y_k;
 * device vectors of n_pad entries; the host all-reduces */
int tgp_dist_trmv_partial(tgp_dist* h, const void* y_dev, void* out_dev);
/* Ks = K(X, X*) (direct.py:87-91) as right-hand sides of the forward solve: out_dev (n_pad, m_pad) ROW-major, zero
 * padded; m_pad a multiple of 128 */
int tgp_dist_cross_cov(tgp_dist* h, const tgp_kop* prog, int nops, int64_t m, const void* Xt_host, int64_t m_pad,
                       void* out_dev);
/* this rank's share of colsum(A o A) (conditional variance without the M x M product, direct.py:94-95) and of A^T A
 * (direct.py:95; column-major nrhs x nrhs) over the rows of its OWNED blocks; the host all-reduces */
int tgp_dist_colsumsq_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* out_dev);
int tgp_dist_gram_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* out_dev);
/* block (i, j) of the same product for two sets of solved columns (round 6: condition_gram in chunks of test points):
 * out (nrhs_i x nrhs_j, column-major) = this rank's share of x_i^T x_j */
int tgp_dist_gram_pair_owned(tgp_dist* h, int64_t nrhs_i, const void* xi_dev, int64_t nrhs_j, const void* xj_dev,
                             void* out_dev);
/* after a rank-local failure: joins every stream and forgets the interrupted pass's markers (a retry starts from a
 * quiet device) */
int tgp_dist_abort(tgp_dist* h);

/* Round 5.  Backward substitution with MANY right-hand sides on the resident distributed factor (reference
 * solvers/direct.py:66-68, trans=1, y (N, R)): right-looking, block k from the last to the first --
 *   tgp_dist_bwd_block_multi   owner(k): X_k = L_kk^-T Y_k into block k of x;  the caller broadcasts it (ONE nb x R message);
 *   tgp_dist_bwd_update_multi  every rank: Y_i -= L[k, i]^T X_k for its OWN block columns i in [stop_block, k) -- ONE product.
 * x: (n_pad, nrhs) ROW-major device memory, nrhs a multiple of 128 (as tgp_dist_fwd_block); yloc: the rank's OWN blocks of the
 * right-hand sides side by side, (nloc * nb, nrhs) row-major -- what tgp_dist_fwd_solve_left leaves in xloc, or
 * tgp_dist_gather_owned of a global buffer (world size 1: the two layouts coincide, pass x itself).
 * tgp_dist_identity_cols fills such a buffer with columns c0 .. c0 + nrhs - 1 of the identity. */
/* Forward substitution with many right-hand sides, LEFT-looking fan-in (the form that scales over the ranks; the
 * right-looking tgp_dist_fwd_block is the better one at world size 1): step k --
 *   tgp_dist_fwd_partial      every rank: acc_k -= X[its columns in [first_block, k)] L[k, those columns]^T  (one product);
 *   the caller reduces acc_k to owner(k);
 *   tgp_dist_fwd_solve_left   owner(k): x_k = L_kk^-1 (y_k + acc_k) into x (global rows) AND xloc (its own solved blocks
 *                             side by side: (nloc * nb, nrhs) row-major, local column l at rows l * nb ..). */
int tgp_dist_fwd_partial(tgp_dist* h, int64_t k, int64_t nrhs, const void* xloc_dev, void* acc_dev, int64_t first_block);
int tgp_dist_fwd_solve_left(tgp_dist* h, int64_t k, int64_t nrhs, const void* y_dev, const void* acc_dev, void* x_dev,
                            void* xloc_dev);
int tgp_dist_bwd_block_multi(tgp_dist* h, int64_t k, int64_t nrhs, void* x_dev, const void* yloc_dev);
int tgp_dist_bwd_update_multi(tgp_dist* h, int64_t k, int64_t nrhs, const void* x_dev, void* yloc_dev, int64_t stop_block);
int tgp_dist_gather_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* yloc_dev);
int tgp_dist_identity_cols(tgp_dist* h, int64_t c0, int64_t nrhs, void* out_dev);

/* Gradient of log_probability on the block-column path -- what jax.value_and_grad of reference gp.py:126-138 gives a caller
 * at any size: d ll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) with alpha = K^-1 r in the handle's replicated
 * vector.  K^-1 is visited a chunk of columns at a time (the caller solves L L^T Z = E_chunk on the resident factor; the
 * chunk arrives replicated, (n_pad, nrhs) row-major): every rank adds the lower-triangle terms of ITS block rows.
 *   _begin(prog)                        zero the accumulators (the kernel program whose parameters are differentiated)
 *   _chunk(c0, nrhs, kcols_dev, dims)   += this rank's share for columns c0 .. (dims != 0: also d / d log-scale of each
 *                                       input dimension, as tgp_solver_grad's grad_logscale); keeps diag(K^-1) of the chunk
 *   _end(out, logscale|NULL, diag|NULL) this rank's PARTIAL sums: out[2 i + q] = parameter q of op i (2 nops doubles),
 *                                       grad_logscale[d]; kinv_diag_host (n entries of the solver's dtype, identical on
 *                                       every rank).  The caller all-reduces out / grad_logscale over the ranks. */
int tgp_dist_grad_begin(tgp_dist* h, const tgp_kop* prog, int nops);
int tgp_dist_grad_chunk(tgp_dist* h, int64_t c0, int64_t nrhs, const void* kcols_dev, int32_t with_logscale);
int tgp_dist_grad_end(tgp_dist* h, double* grad_params, double* grad_logscale, void* kinv_diag_host);
/* inspection: local block column l (rows from its diagonal block down, ld = rows) to the host */
int tgp_dist_get_column(tgp_dist* h, int64_t l, void* out_host);

/* ---- schedule dry run (test infrastructure of the host logic; needs no GPU) -------------
 * The kernel launches and event operations tgp_solver_factor (fused = 0) or
 * tgp_solver_factor_logprob (fused = 1) would enqueue on the library's five streams for an
 * n_pad x n_pad problem (n_pad a multiple of 128), in host order, without any HIP call.
 * Ten int64 per record: kind, stream, v[0..7].
 *   kind 1 potf2     v = {tile offset, pending-update operand offset or -1, ld}
 *        2 trsm      v = {L tile offset, B offset, rows, ld}
 *        3 gemm      v = {A, B, C offsets, m, n, k, lower | role << 8, ld}   (C -= A B^T)
 *        4 forward-substitution step   v = {L tile offset, rows below, ld}
 *        5 event record  v = {event}          6 stream wait  v = {event}
 *        7 assembly      v = {first column tile, column tiles, ld, flags}
 *        8 residual copy into the work vector     9 final reductions
 *       10 fused panel step   v = {tile offset, pending-update operand offset or -1, rows below, ld,
 *                                  has_potf2}  (potf2 + per row tile: pending update, trsm)
 *       11 persistent chain   v = {panel origin offset, ld, row tiles, first block column, end block column}
 *                                  (potf2 of blocks [max(first, 1), end), the solves of the rows below and the
 *                                  in-panel updates of those block columns: ONE launch)
 *       12 chain poll         v = {panel origin offset, ld, block column}: the stream continues once that block
 *                                  column of the chain launch in front of it (host order) is final
 * `fused`: 1 = forward substitution fused into the factorisation.  `options`: "key=value,..." over the
 * names of tgp_ctx_set_option (the format of the TGP_HIP_OPTIONS environment variable), NULL or "" for
 * the library defaults -- the dry run takes every tuning the real run takes.
 *   stream 0 main, 1 panel, 2 solve, 3 update, 4 assembly; offsets are element offsets from
 *   the matrix base (column-major, leading dimension ld).
 * tests/test_schedule.py replays the records and checks that every pair of conflicting
 * accesses is ordered by stream order or an event. */
int tgp_trace_factor(int64_t n_pad, const char* options, int32_t fused, int64_t* out, int64_t cap_records,
                     int64_t* n_records);

/* Measurement hook of the persistent panel chain (ctx option "chain_stamps" = 1 before a factorisation): per
 * tile task of the chain launches since the last tgp_potrf / tgp_solver_factor* call, 16 int64 =
 * {kind (0 bulk tile, 1 diagonal chain), row tile, block column, launch index, 12 time stamps} of the device's
 * 100 MHz real-time counter (0 = phase not run), copied to `out` (at most cap_tasks tasks; *n_tasks = recorded).
 * Phases: scripts/chain_timeline.py. */
int tgp_chain_stamps(tgp_ctx* ctx, int64_t* out, int64_t cap_tasks, int64_t* n_tasks);

/* Test hook, host only (no device, no context): the persistent chain's ticket -> task map (csrc/chain_tasks.h -- the
 * text the kernel decodes its ticket with and the launch is sized by) for a launch over block columns [cb, ce) of a
 * panel with R row tiles and nblk block columns.  *n_tasks = tickets of the launch; ticket >= 0: out5 = {kind, row
 * tile, block column, source column, part} with kind 0 solve | 1 diag | 2 update | 3 update of a diagonal tile |
 * 4 a part (an eighth) of an update | 5 xsolve.  tests/test_chain_tasks.py checks on the CPU that every task of a launch
 * exists exactly once and waits for EARLIER tickets only. */
int tgp_chain_task(int64_t R, int64_t nblk, int64_t cb, int64_t ce, int64_t ticket, int32_t* out5, int64_t* n_tasks);
/* The same for the TABLE a launch is really given (round 6): ticket order of tgp_chain_task with the K-batched updates of the
 * policy (batch, lag, rowlag, minrows: options chain_batch*) folded in -- the list launch_chain uploads, word for word.  out6 (or
 * NULL) takes {kind, row tile, block column, LAST source column, part, FIRST source column} per task for at most
 * cap_tasks tasks, *n_tasks the length of the list; kind 6 = update of the tile from block columns first .. last in one
 * product.  fwd != 0: with the forward substitution of the launch's block columns as tasks -- kind 7 fsolve(c) (z_c =
 * L_cc^-1 y_c), kind 8 fupdate(c, g) (rows of group g = `row tile` field below block c, 16 row tiles per group). */
int tgp_chain_tasks(int64_t R, int64_t nblk, int64_t cb, int64_t ce, int64_t batch, int64_t lag, int64_t rowlag, int64_t minrows,
                    int64_t fwd, int32_t* out6, int64_t cap_tasks, int64_t* n_tasks);

/* ---- RCCL from the C ABI (round 5) -------------------------------------------------------------------------------
 * The block-column driver's collectives -- north_star's "RCCL broadcast of the current panel and reduce of the solve
 * RHS over xGMI" -- issued by the library itself on ITS streams, on plain device pointers: no torch.distributed process
 * group between the panel chain and the wire.  The reference has no multi-device code; the Python side of the seam is
 * tinygp_amd/comm.py (RcclComm) under tinygp_amd/distributed.py.
 *
 * A communicator is one rank of `world` (one process per GPU).  Rank 0 calls tgp_comm_unique_id and the CALLER hands the
 * 128 bytes to every rank (TCP on MASTER_ADDR, a shared file, any launcher-side channel); every rank then calls
 * tgp_comm_create (collective: it returns when all ranks have joined).  librccl is loaded on first use.
 * Collectives are IN PLACE on `count` elements of `dtype` at the device pointer `buf`, enqueued on stream `which` of
 * the context (0 main, 1 priority/panel) and asynchronous.  tgp_comm_record / _wait order the two streams against each
 * other without a host block ("the panel broadcast on the priority stream has arrived" -> main stream). */
#define TGP_COMM_ID_BYTES 128
typedef struct tgp_comm tgp_comm;
int tgp_comm_unique_id(void* id_out /* TGP_COMM_ID_BYTES */, int32_t* rccl_version_out /* or NULL */);
int tgp_comm_create(tgp_ctx* ctx, int32_t world, int32_t rank, const void* id, tgp_comm** out);
int tgp_comm_destroy(tgp_comm* c);
int tgp_comm_info(tgp_comm* c, int32_t* world, int32_t* rank);
int tgp_comm_broadcast(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t root);
int tgp_comm_reduce(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t root); /* sum, to root */
int tgp_comm_all_reduce(tgp_comm* c, int which, void* buf, int64_t count, int dtype, int32_t op); /* 0 sum 1 min 2 max */
int tgp_comm_record(tgp_comm* c, int which, int64_t* ticket);
int tgp_comm_wait(tgp_comm* c, int which, int64_t ticket);

/* Transfers addressed to a stream of the context (0 main, 1 priority): the driver's host side owns plain device
 * buffers (tgp_malloc) and must order its fills and copies against the kernels and collectives of THAT stream.
 * h2d / d2h return when the host buffer may be reused / holds the data; d2d and memset are asynchronous. */
int tgp_stream_h2d(tgp_ctx* ctx, int which, void* dst_dev, const void* src_host, int64_t bytes);
int tgp_stream_d2h(tgp_ctx* ctx, int which, void* dst_host, const void* src_dev, int64_t bytes);
int tgp_stream_d2d(tgp_ctx* ctx, int which, void* dst_dev, const void* src_dev, int64_t bytes);
int tgp_stream_memset(tgp_ctx* ctx, int which, void* dst_dev, int byte, int64_t bytes);
int tgp_stream_sync(tgp_ctx* ctx, int which);

/* Test hook, host only: the MFMA products' workgroup-id -> output-tile map (csrc/tile_order.h, the text the kernels decode
 * their ids with) for tm x tn tiles (`lower`: tiles ti >= tj only, tn <= tm) in the order `band` selects (0: column by column;
 * > 0: bands of that many tile rows -- ctx option "tile_band").  *n_tiles = ids of the launch; id >= 0: its (ti, tj).
 * tests/test_host_logic.py checks that every order is a bijection onto the tiles. */
int tgp_tile_order(int64_t tm, int64_t tn, int32_t lower, int32_t band, int64_t id, int32_t* ti, int32_t* tj, int64_t* n_tiles);

#ifdef __cplusplus
}
#endif
#endif /* TGP_HIP_H */
