// tgp_common.h -- shared host/device declarations of libtgp_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <array>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/tgp_hip.h"

namespace tgp {

constexpr int TILE = TGP_TILE;  // 128: matrix padding and GEMM tile
constexpr int SUB = 16;         // MFMA 16x16 sub-block (diag-inverse granularity)

void set_error(const char* fmt, ...);

#define TGP_HIP_TRY(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::tgp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                       __LINE__);                                                          \
      return (_e == hipErrorOutOfMemory) ? TGP_E_NOMEM : TGP_E_HIP;                        \
    }                                                                                      \
  } while (0)

#define TGP_ARG_CHECK(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ::tgp::set_error(__VA_ARGS__);    \
      return TGP_E_ARG;                 \
    }                                   \
  } while (0)

#define TGP_TRY(expr)          \
  do {                         \
    int _s = (expr);           \
    if (_s < 0) return _s;     \
  } while (0)

// Kernel program in kernarg form (wave-uniform scalar loads on the device).
struct KProg {
  int32_t n;
  int32_t op[TGP_KPROG_MAX];
  int32_t metric[TGP_KPROG_MAX];
  double p0[TGP_KPROG_MAX];
  double p1[TGP_KPROG_MAX];
};

int make_kprog(const tgp_kop* prog, int nops, KProg* out);  // validates (TGP_E_ARG)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace tgp

// Dry-run record of one enqueued operation (tgp_trace_factor): the factorisation's schedule
// without a GPU, for the host-side dependency checker in tests/test_schedule.py.
struct tgp_trace_rec {
  int64_t kind;    // 1 potf2, 2 trsm, 3 gemm, 4 forward-substitution step, 5 event record,
                   // 6 stream wait, 7 assembly of column tiles, 10 fused panel step
  int64_t stream;  // 0 main, 1 panel, 2 solve, 3 update, 4 assembly
  int64_t v[8];    // operands as element offsets from the matrix base (see capi.hip)
};

// One HIP device + stream (+ a high-priority side stream for panel look-ahead).
struct tgp_ctx {
  // every extern "C" entry point that takes this context (or a solver built on it) holds this
  // lock for its whole duration: the scratch below (d_scal, d_work, events, asm_pending) is
  // shared by all solvers of the context, so two host threads are serialised, not raced
  std::recursive_mutex mu;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t panel_stream = nullptr;  // look-ahead panel factorisation
  hipStream_t solve_stream = nullptr;  // forward substitution overlapped with the factorisation
  hipStream_t update_stream = nullptr;  // in-panel updates beside the next potf2
  hipStream_t asm_stream = nullptr;     // assembly of the columns right of the first panel

  hipEvent_t ev_asm = nullptr;          // ... finished (potrf waits before its first update)
  bool asm_pending = false;
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, ev_e = nullptr;
  hipEvent_t ev_f = nullptr;  // second marker of the far in-panel updates (fused panel step: they alternate)
  hipEvent_t ev_g1 = nullptr, ev_g2 = nullptr;  // split gate: column block 1 / column blocks 2.. of the next panel updated
  hipEvent_t ev_h = nullptr;  // depth-2 schedule: the next panel's first potf2 has been issued (priority stream)
  hipEvent_t ev_i = nullptr, ev_j = nullptr;  // chain_gate_split: the gate's inputs are ready / the next panel's first block is factored
  bool gate_pending = false;  // set by potrf in front of a panel whose block column arrives in those pieces
  int64_t nb_outer = 1024;  // measured best for N = 4k .. 32k (profiles/r01_f_nb_sweep.txt)
  int64_t lookahead = 1;
  int64_t dist_solve_aux = 1;  // block-column driver: forward steps on the update stream (0: a solve stream of their own)
  // the fused forward-substitution steps ride on the update stream, behind the in-panel update of the same
  // block: three busy queues instead of four (c2 30.3 -> 29.8 ms, N = 4096 2.66 -> 2.45 ms; 0: own stream)
  int64_t solve_on_update = 1;
  std::chrono::steady_clock::time_point submitted;  // host time at which potrf had enqueued its last launch
  // panels that start with at least this many rows left are 2 nb_outer wide (0: never): half as many passes
  // over the trailing matrix; +1.3 % at N = 65 536, a loss below ~30 000 rows (profiles/r02_i_wide_panels.txt)
  int64_t nb_wide_rows = 30000;
  int64_t profile = 0;
  int64_t first_split = 5;  // blocks of a panel after which its share of the next block-column update is issued early (0: off)
  int64_t stream_trsv = 1;  // forward solves on a resident factor: one streaming launch (0: one launch pair per block)
  int64_t trsv_groups = 0;  // workgroups per block row of that launch (chol.hip, trsv_fwd_stream_kernel; 0: by size, 3 / 4 / 6)
  int64_t keep_grad_buffers = 0;  // tgp_solver_grad keeps its (N + 128) x N work matrix between calls
  int64_t first_small_tiles = 1100;  // look-ahead block-column updates up to this many tiles use 64x64 tiles
  // panel chain as ONE launch per 128-column block (panel_step_kernel: potf2 + the rows' own pending update
  // + trsm behind a device-side flag) instead of potf2 | trsm | update of the next column block (0: the latter)
  int64_t fused_step = 0;  // measured (profiles/r02_r): parity-green but 0-8 % slower than the separate launches
  // workgroup slots (of 2 per CU) that a trailing update which runs beside a panel chain leaves free
  // for the chain's kernels (gemm.hip: the update is persistent over its tiles, so its grid is its footprint)
  // (96 since round 6's last batches: 128 -> 96 is -0.04 ... -0.13 ms at c2 on three boxes and lifts the updates with 6 144 /
  // 5 120 / 4 096 rows from 44-45 to 45-52 TFLOP/s; 64 and 80 are flat: profiles/r06_i section 6)
  int64_t chain_reserve = 96;
  int64_t reserve_max_tiles = 1200;  // ... when the update has at most this many 128 x 128 tiles (chain-bound panels)
  // the block-column update between two chains (the `gate`) is issued column block 0 | 1 | 2..: the chain starts
  // behind the first piece and meets the others at its second and third block (fused panel step only)
  int64_t gate_split = 1;
  // two-level panel: the in-panel rank-128 updates stay inside sub-panels of this many columns and every finished
  // sub-panel updates the panel's remaining columns with ONE K = sub_panel product (0: off) -- 57 % fewer flops on
  // the rank-128 kernel at nb_outer = 1024 / sub_panel = 512; only for panels with at least sub_panel_min_rows rows
  int64_t sub_panel = 0;
  int64_t sub_panel_min_rows = 0;
  int64_t nb_first = 0;    // width of the FIRST panel, whose chain nothing hides (0: nb_outer)
  int64_t split_tail = 0;  // trailing update: the last, partly filled round of tiles is split along k (gemm.hip)
  int64_t gemm_role = 1;     // role tgp_gemm_nt launches with (measurement hook: 4 = the 64x64-tile kernel at any k)
  int64_t late_join = 1;       // fused evaluation: ONE host join per evaluation, `info` read with the scalars (0: round 5's two)
  // fused evaluation: sum z^2 and sum log L_ii are left by the chain's fsolve tasks (chol.hip, ChainArgs::red) -- no reduction
  // launch behind the factorisation (0: the two kernels of util.hip)
  int64_t chain_reduce = 1;
  double* d_chain_red = nullptr;   // [2 * CHAIN_MAX_ROW_TILES] per-block partial sums
  int64_t chain_red_total = 0;     // set by the fused evaluation around its potrf call: blocks of the matrix (0: off)
  bool reductions_done = false;    // ... and the answer: the launch with the matrix's last block left both sums in d_scal[0..1]
  bool defer_join = false;     // set by the fused evaluation around its potrf call: do not join, `info` is read with the scalars
  bool join_deferred = false;  // ... and potrf's answer: it did leave the join (and the check of d_info) to the caller
  int64_t reserve_hint = 0;  // set by potrf in front of such a launch, consumed by launch_gemm_nt
  int64_t prefix_hint_cols = 0;             // ... the same for the merged trailing update's prefix (gemm.hip, GemmArgs)
  int32_t* prefix_hint_counter = nullptr;
  const int32_t* potf2_wait_counter = nullptr;  // consumed by the next launch_potf2: poll this counter first (chol.hip)
  int64_t potf2_wait_target = 0;
  uint32_t* d_step_flag = nullptr;  // the flag potf2's workgroup publishes; value = step_epoch of the launch
  uint32_t step_epoch = 0;
  // persistent panel chain (chol.hip, chain_kernel): ONE launch per panel (two with an early share) factors its
  // block columns, solves the rows below and applies the in-panel updates -- tile tasks behind a ticket counter,
  // hand-offs by per-tile flag words that carry the launch's epoch (0: potf2 | trsm | update launches per block)
  int64_t chain_kernel = 1;
  uint32_t* d_chain_flags = nullptr;  // CHAIN_MAX_ROW_TILES x 64 words, zero at allocation, never reset (epochs)
  int32_t* d_chain_ticket = nullptr;  // [0] ticket counter, [16 + c] final tiles of block column c: zeroed per launch
  // fp64 whole-tile update tasks on the 4x4x4 MFMA form with LDS-direct staging (chain_update_fast).  MEASURED, OFF
  // (profiles/r05_b): no spill any more, but a task takes 22-25 us + 4-6 us publish against 20-21 + 1.7 for round 4's
  // form -- the task is bound by its prologue (tile + first operands: one round trip), the per-k-step LDS latency under the
  // 128-register cap and 32 scattered 8-byte write-through stores per lane, not by the MFMA form; c2 26.1 vs 25.5 ms
  int64_t chain_fast_update = 0;
  // K-batched update tasks of the chain launch (round 6, chain_tasks.h ChainPolicy; fp64 only): block columns per batch
  // (<= 1: off), columns between a batch's end and the tile's own column, rows between the tile and the diagonal, row tiles
  // that must be left.  BUILT, parity-green, MEASURED, OFF (profiles/r06_b, r06_c, r06_f): a K = 512 batch task takes 74 us
  // = 18.5 us per 128^3 against 19-20 + 1.6 for the single tasks (the product streams its operands from the Infinity Cache
  // one k-tile ahead, not from the L2 as the trailing update does with its tile order) -- 8-17 % less compute-unit time per
  // flop -- while the tile's NEXT task waits three times as long for it: N = 4 096 as one launch 1.81 vs 1.35 ms with batches
  // everywhere, N = 8 192 as ONE launch 5.1-5.6 vs 6.7 ms unbatched but 4.75 ms panel by panel, c2 25.8 vs 25.2 ms
  int64_t chain_batch = 1;
  int64_t chain_batch_lag = 1;
  int64_t chain_batch_rowlag = 4;
  int64_t chain_batch_minrows = 32;
  // task tables of the chain launches (ticket -> packed task), one per launch shape and policy, built on first use
  struct ChainTable {
    uint64_t* dev = nullptr;
    int64_t count = 0;
  };
  std::map<std::array<int64_t, 9>, ChainTable> chain_tables;
  int64_t chain_full_rows = 4096;     // with at most this many rows left the WHOLE rest is one chain launch (measured
                                      // at N = 16 384: 4096 26.6 ms, 6144 26.9, 8192 27.6; per-block chain 28.2)
  int64_t chain_depth2 = 1;           // gate + chain of the next panel on the priority stream, two panels ahead
  // round 6: ONE trailing-update launch per panel on the 128 x 128-tile kernel, the next panel's block column first
  // (write-through, counted), the next chain behind a one-wave poll of that count (chol.hip, potrf)
  int64_t chain_merged = 1;
  // two-level panel for the chain in the merged schedule (chol.hip, chain_of_panel): sub-panel width in columns (0: off), only
  // for panels with at least chain_sub_min_rows rows, the sub-panel's product on role chain_sub_role (4: 64 x 64 tiles)
  int64_t chain_sub_panel = 0;
  int64_t chain_sub_min_rows = 0;
  int64_t chain_sub_role = 4;
  // depth-2 schedule, chain-bound panels (the big update has at most reserve_max_tiles tiles): pre(p) starts behind
  // the next panel's first potf2 -- issued at once it fills every compute unit with three 48-KB workgroups, and the
  // one-workgroup potf2 (74 KB) on the chain pipeline waited 120-290 us for room (profiles/r04_c)
  int64_t kmat_plain_div = 0;    // tests: the assembly's quotients by the division instruction sequence (kmat.hip, UDiv)
  // followers of a chain launch (forward-substitution steps, early shares) start while it runs, behind
  // 1: a one-wave poll kernel on the block column's counter, bounded by wall clock (default); 3: hipStreamWaitValue32 on
  // it (the runtime's own one-wave wait kernel: as fast, no timeout -- measured, not the default); 0: they wait for the
  // whole launch (the default under a counter-collecting profiler, which runs kernels one at a time in its own order)
  // merged schedule: the poll of the update's prefix counter is the first thing the next panel's potf2 launch does (1),
  // a one-wave kernel of its own in front of it (2), off (0)
  int64_t chain_polls = 1;
  // Round 6: the fused forward substitution as TASKS of the chain launch (chain_tasks.h F(c), chol.hip chain_fsolve /
  // chain_fupdate): no poller, no forward-step launch at all.  0: round 5's followers on the solve stream (chain_polls)
  int64_t chain_fwd_tasks = 1;
  bool wait_values_inflight = false;  // this factorisation enqueued stream wait-values: join with a deadline (join_bounded)
  // tile order of the MFMA products (tile_order.h): bands of this many tile rows, column by column inside a band; 0 = column
  // by column over all rows (rounds 1-4).  Round 5, measured at c2 (profiles/r05_j, r05_k): fabric traffic of a
  // trailing-update launch 2 x 1 471 + 334 MB = 3.28 GB at 0 -> 2 x 725 + 334 = 1.78 GB at 8 (714 at 4, 932 at 16, 1 296 at 32);
  // 64 x 64-tile kernel 663 -> 284 MB; the evaluation's time does not move (25.41 / 25.48 vs 25.45 / 25.58 ms; N = 65 536
  // 1 381.7 vs 1 379.6): the fabric was never the bound -- a third of its traffic is simply not needed
  int64_t tile_band = 8;
  int64_t asm_defer = 0;         // 1: the side-stream assembly of the columns right of the first panel starts behind the first potf2
  std::function<int()> deferred_asm;  // ... the launch that was held back (cleared when run)
  hipEvent_t ev_asm_gate = nullptr;
  int64_t fault_inject = 0;      // TEST hook. 1: the next bounded join expires at once (exercises rescue + retry), then clears
  int64_t host_join = 1;         // 0: plain hipStreamSynchronize even then (A/B of the polling join)
  bool serializing_tool = false; // a counter-collecting profiler is attached (ROCPROF_COUNTER_COLLECTION): chain_polls defaults to 0
  hipStream_t rescue_stream = nullptr;
  hipEvent_t ev_join = nullptr;
  bool has_device = false;       // (false: the schedule tracer's context, tgp_trace_factor)
  bool can_wait_value = false;   // hipDeviceAttributeCanUseStreamWaitValue (else chain_polls 1 falls back to the poll kernel)
  int64_t poll_timeout_ms = 4000;  // wall-clock bound of every device-side wait (chol.hip, PollClock)
  int64_t timeout_retries = 0;     // factorisations repeated on the launch-per-block path after a TGP_E_TIMEOUT
  int64_t chain_pre_wait = 0;  // (measured: no effect at N = 16 384, -3 % at N = 8 192 -- off)
  // depth-2 schedule (round 6): the next panel's FIRST diagonal block is updated (a 128 x 128 product of its own) and
  // factored on the idle update stream BESIDE the gate, which skips that block: the one-workgroup potf2 launch waits 50-200 us
  // for a compute unit while a trailing update fills the chip (rocprofv3: 177 us per launch on average at N = 16 384 against
  // 27 us of work) -- between the gate and the chain launch, on the chain pipeline.  BUILT, race-checked, MEASURED FLAT
  // (profiles/r06_b: c2 25.43 / 25.31 vs 25.35 / 25.34 ms, N = 8 192 4.78 vs 4.81): the chain pipeline is not what the
  // evaluation waits for.  Off.
  int64_t chain_gate_split = 0;
  int64_t chain_lds_pad = 10240;      // dynamic LDS per chain workgroup that nobody uses: one chain workgroup per CU
  int64_t chain_stamps = 0;           // 1: every chain task records its phases' time stamps (tgp_chain_stamps)
  long long* d_chain_stamps = nullptr;  // CHAIN_STAMP_TASKS x 16, allocated on first use
  int64_t chain_stamp_base = 0;       // tasks recorded so far in this factorisation
  int64_t chain_launches = 0;
  bool chain_polls_pending = false;  // ev_f marks the last poller of the previous chain launch (solve stream)
  // small device scratch: scal[0..15] doubles, info int
  double* d_scal = nullptr;
  int32_t* d_info = nullptr;
  void* d_dinv = nullptr;  // inverse 16x16 diagonal blocks, grown on demand
  size_t dinv_bytes = 0;
  void* d_gemm_ws = nullptr;  // split-tail workspace of the trailing update: partial tiles + per-tile counters
  size_t gemm_ws_bytes = 0;
  void* d_work = nullptr;  // generic workspace, grown on demand
  size_t work_bytes = 0;
  // profiling (option "profile"): event pairs around trailing-update launches
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  double prof_syrk_ms = 0, prof_syrk_flops = 0, prof_syrk_union_ms = 0;
  int64_t prof_syrk_launches = 0;
  int cus = 0;
  // dry run: launches and event operations are recorded here instead of being issued
  std::vector<tgp_trace_rec>* trace = nullptr;
  const void* trace_base = nullptr;  // matrix base pointer of the traced factorisation
};

namespace tgp {

// ---- dry-run helpers ----------------------------------------------------------------
inline int64_t trace_stream_id(const tgp_ctx* ctx, hipStream_t st) {
  if (st == ctx->stream) return 0;
  if (st == ctx->panel_stream) return 1;
  if (st == ctx->solve_stream) return 2;
  if (st == ctx->update_stream) return 3;
  if (st == ctx->asm_stream) return 4;
  return -1;
}
inline int64_t trace_event_id(const tgp_ctx* ctx, hipEvent_t ev) {
  if (ev == ctx->ev_a) return 0;
  if (ev == ctx->ev_b) return 1;
  if (ev == ctx->ev_c) return 2;
  if (ev == ctx->ev_d) return 3;
  if (ev == ctx->ev_e) return 4;
  if (ev == ctx->ev_asm) return 5;
  if (ev == ctx->ev_f) return 6;
  if (ev == ctx->ev_g1) return 7;
  if (ev == ctx->ev_g2) return 8;
  if (ev == ctx->ev_h) return 9;
  if (ev == ctx->ev_i) return 10;
  if (ev == ctx->ev_j) return 11;
  return -1;
}
template <typename T>
inline int64_t trace_off(const tgp_ctx* ctx, const T* p) {
  return p == nullptr ? -1 : int64_t(p - static_cast<const T*>(ctx->trace_base));
}
inline void trace_push(tgp_ctx* ctx, int64_t kind, hipStream_t st, int64_t a = 0, int64_t b = 0,
                       int64_t c = 0, int64_t d = 0, int64_t e = 0, int64_t f = 0, int64_t g = 0,
                       int64_t h = 0) {
  ctx->trace->push_back(tgp_trace_rec{kind, trace_stream_id(ctx, st), {a, b, c, d, e, f, g, h}});
}
// event record / stream wait that honour the dry run
inline int ev_record(tgp_ctx* ctx, hipEvent_t ev, hipStream_t st) {
  if (ctx->trace) {
    trace_push(ctx, 5, st, trace_event_id(ctx, ev));
    return TGP_OK;
  }
  TGP_HIP_TRY(hipEventRecord(ev, st));
  return TGP_OK;
}
inline int st_wait(tgp_ctx* ctx, hipStream_t st, hipEvent_t ev) {
  if (ctx->trace) {
    trace_push(ctx, 6, st, trace_event_id(ctx, ev));
    return TGP_OK;
  }
  TGP_HIP_TRY(hipStreamWaitEvent(st, ev, 0));
  return TGP_OK;
}

// width of the factorisation's FIRST panel (potrf's width(0)); the assembly of the columns to its right may
// still be running when the first chain starts (capi.hip, assemble_lower)
inline int64_t first_panel_cols(const tgp_ctx* ctx, int64_t n) {
  int64_t w = ctx->nb_outer < TILE ? TILE : ctx->nb_outer / TILE * TILE;
  if (ctx->nb_first >= TILE) w = ctx->nb_first / TILE * TILE;
  if (ctx->chain_kernel != 0 && n <= ctx->chain_full_rows && n / TILE <= 64) w = n;
  return n < w ? n : w;
}

int ensure_dinv(tgp_ctx* ctx, size_t bytes);
int ensure_work(tgp_ctx* ctx, size_t bytes);
int ensure_solve_stream(tgp_ctx* ctx);  // the solve stream exists only once something asks for it

// ---- launchers (all async on the given stream) -------------------------------------
template <typename T>
int launch_kmat(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                const T* X2, const T* diag, T* out, int64_t ld, int64_t rows_out, int64_t cols_out,
                int flags);
// column tiles [tc0, tc0 + ntc) of the same matrix (128 columns each) on stream `st`
template <typename T>
int launch_kmat_cols(tgp_ctx* ctx, hipStream_t st, const KProg& kp, int64_t n1, int64_t n2, int d,
                     const T* X1, const T* X2, const T* diag, T* out, int64_t ld, int64_t rows_out,
                     int64_t cols_out, int flags, int64_t tc0, int64_t ntc);
constexpr int KMAT_LOWER = 1;         // only tiles on/below the diagonal
constexpr int KMAT_PAD_IDENTITY = 2;  // padding = identity (else zeros)
constexpr int KMAT_PLAIN_DIV = 4;     // straight-line evaluator: the division itself instead of UDiv (ctx kmat_plain_div; tests)

template <typename T>
int launch_kdiag(tgp_ctx* ctx, const KProg& kp, int64_t n, int d, const T* X, const T* add, T* out);
template <typename T>
int launch_kmat_gemv(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                     const T* X2, const T* v, T* out);
template <typename T>
int launch_kmat_gemv_multi(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                           const T* X2, const T* v, int64_t nv, T* out);

// C (m x n) = beta*C + alpha*A*B^T ; mode 0: C -= A B^T ; mode 1: C = A B^T.
// role: 0 = trailing update (profiled as the dominant kernel), 1 = everything else (64x64
// tiles when k <= 256), 3 = in-panel update that skips the first 128x128 diagonal tile (potf2
// folds it in), 4 = 64x64 tiles at any k (latency-bound look-ahead block-column update).
template <typename T>
int launch_gemm_nt(tgp_ctx* ctx, hipStream_t st, int64_t m, int64_t n, int64_t k, const T* A,
                   int64_t lda, const T* B, int64_t ldb, T* C, int64_t ldc, int lower, int mode,
                   int role);

// batched C_q = A_q B_q^T with a triangular A on the 128 x 128-tile kernel in patch order (gemm.hip, ROLE 3)
template <typename T>
int launch_gemm_tri(tgp_ctx* ctx, hipStream_t st, int64_t m, int64_t n, int64_t k, const T* A, int64_t lda,
                    const T* B, int64_t ldb, T* C, int64_t ldc, int lower, int mode, int batch, int64_t sA,
                    int64_t sB, int64_t sC);

template <typename T>
int launch_gemm_nt_dist(tgp_ctx* ctx, hipStream_t st, int64_t n_rows, int64_t nb, int64_t k,
                        const T* P, int64_t ldp, T* Cloc, int64_t ldc, int G, int rank, int64_t l0,
                        int64_t nloc, int64_t reserve = 0);

template <typename T>
int launch_potf2(tgp_ctx* ctx, hipStream_t st, T* A, int64_t ld, T* dinv, int32_t* info,
                 int32_t pivot_base, const T* Xp = nullptr, int64_t ldx = 0);
template <typename T>
int launch_trsm(tgp_ctx* ctx, hipStream_t st, int64_t m, const T* L, int64_t ldl, const T* dinv,
                T* B, int64_t ldb);
// fused panel step on the 128-block at Ljj: potf2 (has_p; folding the pending update from Xp when
// Xp != NULL) + per 128-row tile of the m rows below: pending update from Xp's block column, trsm
template <typename T>
int launch_panel_step(tgp_ctx* ctx, hipStream_t st, int64_t m, T* Ljj, int64_t ld, T* dj, int32_t* info,
                      int32_t pivot_base, const T* Xp, bool has_p);

constexpr int64_t CHAIN_MAX_ROW_TILES = 4096;  // panels of up to 524 288 rows
constexpr int64_t CHAIN_STAMP_TASKS = 32768;
// persistent chain over block columns [cb, ce) of the panel at A0 (R row tiles, nblk block columns; chol.hip)
template <typename T>
int launch_chain(tgp_ctx* ctx, hipStream_t st, T* A0, int64_t ld, T* dinv0, int64_t pivot_base, int64_t R,
                 int64_t nblk, int64_t cb, int64_t ce, bool head_done, hipEvent_t counters_ready = nullptr,
                 T* y0 = nullptr, bool fprev = false);
int launch_chain_poll(tgp_ctx* ctx, hipStream_t st, const void* A0, int64_t ld, int64_t R, int64_t c, bool first_external);
int launch_prefix_poll(tgp_ctx* ctx, hipStream_t st, const int32_t* counter, int64_t target, int64_t c_off, int64_t ld);
int set_poll_limit(tgp_ctx* ctx, int64_t ms);
int64_t poll_limit_ms();  // the process-wide value behind every context's "poll_timeout_ms"
int join_bounded(tgp_ctx* ctx, hipStream_t st, int64_t n);
int run_deferred_asm(tgp_ctx* ctx, hipStream_t behind);
template <typename T>
int panel_potf2(tgp_ctx* ctx, hipStream_t st, T* A, int64_t ld, T* dinv, int64_t pivot_off,
                int64_t j0, bool pend);
template <typename T>
int panel_chain(tgp_ctx* ctx, hipStream_t st, int64_t n, T* A, int64_t ld, T* dinv,
                int64_t pivot_off, int64_t k0, int64_t kb, bool head_done, T* y,
                int64_t after_blocks, const std::function<int(hipEvent_t)>& mid, int64_t blk_begin = 0,
                int64_t blk_end = -1);
template <typename T>
int launch_trsv_fwd_step(tgp_ctx* ctx, hipStream_t st, int64_t m_below, const T* Ljj, int64_t ld,
                         const T* dj, T* yj);
template <typename T>
int potrf(tgp_ctx* ctx, int64_t n, T* A, int64_t ld, T* dinv, int32_t* info_host, T* y = nullptr);
// winv != NULL (inverses of the 128 x 128 diagonal blocks, compute_winv) and transpose == 0: the
// single-launch streaming solve; otherwise one pair of launches per 128-block
template <typename T>
int trsv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, const T* dinv, int transpose, T* y,
         const T* winv = nullptr, const T* yin = nullptr);  // yin: right-hand side in a buffer of its own (streaming solve only)
template <typename T>
int compute_winv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* winv);
template <typename T>
int trsm_right_lt(tgp_ctx* ctx, int64_t m, int64_t n, const T* L, int64_t ldl, const T* dinv,
                  T* B, int64_t ldb);
template <typename T>
int gemv_sub(tgp_ctx* ctx, int64_t m, int64_t k, const T* P, int64_t ld, const T* x, T* y);
// K^-1 (lower tiles) from the factor L and its 128 x 128 diagonal inverses (compute_winv), through L^-1 by halves:
// S is ONE (n + 128) x n work matrix (lds >= n + 128); on return K^-1 is at S + 128 with leading dimension lds.
template <typename T>
int spd_inverse_lower(tgp_ctx* ctx, int64_t n, const T* L, int64_t ldl, const T* winv, T* S, int64_t lds);
// upper triangle of A (n x n, n a multiple of 64) <- transpose of its lower triangle (the diagonal 64 x 64 blocks are left alone)
template <typename T>
int symmetrize_lower(tgp_ctx* ctx, int64_t n, T* A, int64_t ld);
template <typename T>
int compute_dinv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* dinv);

// reductions into ctx->d_scal[slot] (device), deterministic order
template <typename T>
int launch_sum_log_diag(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, int slot);
template <typename T>
int launch_sum_squares(tgp_ctx* ctx, int64_t n, const T* y, int slot);
template <typename T>
int launch_sum_log_diag_at(tgp_ctx* ctx, hipStream_t st, int64_t n, const T* L, int64_t ld, double* out);
template <typename T>
int launch_sum_squares_at(tgp_ctx* ctx, hipStream_t st, int64_t n, const T* y, double* out);
template <typename T>
int launch_row_sumsq(tgp_ctx* ctx, int64_t m, int64_t n, const T* B, int64_t ldb, const T* base,
                     T* out);  // out[i] = base[i] - sum_j B[i,j]^2
template <typename T>
int launch_trmv_lower(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, const T* y, T* out);
template <typename T>
int launch_extract_lower_rowmajor(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* out);
template <typename T>
int launch_set_lower_from_rowmajor(tgp_ctx* ctx, int64_t n, int64_t npad, const T* src, T* A,
                                   int64_t ld);
template <typename T>
int launch_add_diag(tgp_ctx* ctx, int64_t n, T* A, int64_t ld, const T* diag);

// gradient of the log-probability (row 8f-1): partial[pass][block] sums of
//   w_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta over the lower triangle, w = 1/2 on the diagonal
template <typename T>
int launch_kgrad_cols(tgp_ctx* ctx, const KProg& kp, int which_op, int which_param, int64_t n, int d, const T* X,
                      const T* alpha, const T* Kc, int64_t R, int64_t c0, int64_t nb, int G, int rank,
                      double* out_accum);
template <typename T>
int launch_kcols_diag(tgp_ctx* ctx, int64_t n, const T* Kc, int64_t R, int64_t c0, T* diag);
template <typename T>
int launch_kgrad(tgp_ctx* ctx, const KProg& kp, int which_op, int which_param, int64_t n, int d,
                 const T* X, const T* alpha, const T* Kinv, int64_t ld, double* out_dev);
template <typename T>
int launch_noise_grad(tgp_ctx* ctx, int64_t n, const T* alpha, const T* Kinv, int64_t ld, T* out);
// "leaf" / "amp * leaf" programs: both kernel-parameter sums in ONE pass over K^-1 (returns 0 when
// the program has another shape: use launch_kgrad per parameter)
template <typename T>
int launch_kgrad_fast(tgp_ctx* ctx, const KProg& kp, int64_t n, int d, const T* X, const T* alpha,
                      const T* Kinv, int64_t ld, double* out_dev, int* leaf, int* konst);

int ubench_mfma(tgp_ctx* ctx, int dtype, double* tflops);
int ubench(tgp_ctx* ctx, int kind, int blocks_per_cu, double* tflops, double* cycles_per_op);

}  // namespace tgp
