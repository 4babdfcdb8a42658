// capi.hip -- the extern "C" boundary of libtgp_hip.so (include/tgp_hip.h).
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>

#include "tgp_common.h"
#define CHAIN_HD __host__ __device__
#include "chain_tasks.h"
#define TILE_HD
#include "tile_order.h"

namespace tgp {

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

static int grow(void** p, size_t* cur, size_t bytes) {
  if (bytes <= *cur) return TGP_OK;
  if (*p) TGP_HIP_TRY(hipFree(*p));
  *p = nullptr;
  *cur = 0;
  TGP_HIP_TRY(hipMalloc(p, bytes));
  *cur = bytes;
  return TGP_OK;
}

int ensure_dinv(tgp_ctx* ctx, size_t bytes) { return grow(&ctx->d_dinv, &ctx->dinv_bytes, bytes); }
int ensure_solve_stream(tgp_ctx* ctx) {
  if (ctx->solve_stream != nullptr || ctx->trace) return TGP_OK;
  int lo = 0, hi = 0;
  TGP_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  TGP_HIP_TRY(hipStreamCreateWithPriority(&ctx->solve_stream, hipStreamNonBlocking, hi));
  return TGP_OK;
}

int ensure_work(tgp_ctx* ctx, size_t bytes) { return grow(&ctx->d_work, &ctx->work_bytes, bytes); }

template <typename F>
static int dispatch(int dtype, F&& f) {
  if (dtype == TGP_F64) return f(double{});
  if (dtype == TGP_F32) return f(float{});
  set_error("dtype must be TGP_F32 (0) or TGP_F64 (1), got %d", dtype);
  return TGP_E_ARG;
}

static inline size_t esize(int dtype) { return dtype == TGP_F64 ? 8 : 4; }

}  // namespace tgp

using namespace tgp;

struct tgp_solver {
  tgp_ctx* ctx = nullptr;
  int dtype = TGP_F64;
  int64_t n = 0, npad = 0;
  int d = 1;
  void* X = nullptr;     // (n, d) row-major
  void* diag = nullptr;  // (n,)
  void* A = nullptr;     // npad x npad column-major: K, then L in place
  void* dinv = nullptr;  // (npad/128) * 8 * 256 inverse 16x16 diagonal blocks
  void* vec = nullptr;   // npad work vectors
  void* vec2 = nullptr;
  void* resid = nullptr;  // resident residual (tgp_solver_set_resid)
  void* scratch = nullptr;  // per-solver workspace (multi-RHS / conditional products)
  void* Minv = nullptr;     // gradient path: (npad + 128) x npad, L^-1 in both orientations, then K^-1 (lower tiles at +128)
  void* winv = nullptr;     // inverses of the 128 x 128 diagonal blocks (streaming forward solve), lazy
  bool winv_valid = false;
  size_t scratch_bytes = 0;
  KProg kp{};
  bool has_prog = false, factored = false, has_resid = false;
  int32_t info = 0;
  double logdet_half = 0;  // sum log L_ii
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// argument check + context lock (held until the entry point returns) + device selection
#define CTX_GUARD(ctx)                                                \
  TGP_ARG_CHECK((ctx) != nullptr, "null context");                    \
  std::unique_lock<std::recursive_mutex> _tgp_lock((ctx)->mu);        \
  TGP_HIP_TRY(hipSetDevice((ctx)->device))

#define SOLVER_GUARD(s)                                               \
  TGP_ARG_CHECK((s) != nullptr && (s)->ctx != nullptr, "null solver"); \
  std::unique_lock<std::recursive_mutex> _tgp_lock((s)->ctx->mu);     \
  TGP_HIP_TRY(hipSetDevice((s)->ctx->device))

// K(X, X) + noise into the lower tiles of A.  Only the first panel's columns gate the
// factorisation: the rest of K is assembled on its own stream beside the first panel's
// potf2/trsm chain (potrf waits for it before the first trailing update).
template <typename T>
static int assemble_lower(tgp_ctx* ctx, const tgp::KProg& kp, int64_t n, int d, const T* X,
                          const T* diag, T* A, int64_t npad) {
  using namespace tgp;
  const int64_t tc = npad / 128;
  int64_t t1 = tc;
  if (ctx->lookahead != 0 && ctx->asm_stream != nullptr) {
    int64_t nb = ctx->nb_outer / 128;
    if (nb < 1) nb = 1;
    // (the first panel can be wider than nb_outer: nb_first, or the whole matrix as ONE persistent chain launch)
    nb = std::max<int64_t>(nb, first_panel_cols(ctx, npad) / 128);
    if (nb < tc) t1 = nb;
  }
  const int flags = KMAT_LOWER | KMAT_PAD_IDENTITY;
  if (t1 < tc) {
    TGP_TRY(ev_record(ctx, ctx->ev_asm, ctx->stream));  // X / noise uploads are on the main stream
    TGP_TRY(st_wait(ctx, ctx->asm_stream, ctx->ev_asm));
  }
  TGP_TRY(launch_kmat_cols<T>(ctx, ctx->stream, kp, n, n, d, X, X, diag, A, npad, npad, npad, flags, 0, t1));
  if (t1 < tc) {
    auto side = [=]() -> int {
      TGP_TRY(launch_kmat_cols<T>(ctx, ctx->asm_stream, kp, n, n, d, X, X, diag, A, npad, npad, npad,
                                  flags, t1, tc - t1));
      return ev_record(ctx, ctx->ev_asm, ctx->asm_stream);
    };
    if (ctx->asm_defer != 0) {
      // Round 5 experiment: the other columns' assembly saturates the memory system for 0.25 ms, and the first panel's
      // potf2 -- one workgroup, latency-bound -- took 163 us beside it instead of 27 (profiles/r05_e).  Deferred: potrf
      // launches it behind that potf2 (run_deferred_asm in chol.hip).
      ctx->deferred_asm = side;
    } else {
      TGP_TRY(side());
    }
    ctx->asm_pending = true;
  }
  return TGP_OK;
}

// W_b = L_bb^-1 of every diagonal block, once per factorisation, on first use
template <typename T>
static int ensure_winv(tgp_solver* s) {
  if (s->winv_valid) return TGP_OK;
  if (!s->winv) TGP_HIP_TRY(hipMalloc(&s->winv, 4 * size_t(s->npad / TILE) * 16384 * sizeof(T)));  // W, W^T, tf, tf2
  TGP_TRY(compute_winv<T>(s->ctx, s->npad, (const T*)s->A, s->npad, (T*)s->winv));
  s->winv_valid = true;
  return TGP_OK;
}

extern "C" {

int tgp_abi_version(void) { return TGP_ABI_VERSION; }
const char* tgp_last_error(void) { return g_error.c_str(); }

int tgp_ctx_create(int device, void* stream, tgp_ctx** out) {
  TGP_ARG_CHECK(out != nullptr, "null output pointer");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_error("no HIP device visible (%s); tinygp_amd has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return TGP_E_HIP;
  }
  TGP_ARG_CHECK(device >= 0 && device < count, "device %d out of range (0..%d)", device, count - 1);
  TGP_HIP_TRY(hipSetDevice(device));
  tgp_ctx* ctx = new tgp_ctx();
  ctx->device = device;
  ctx->has_device = true;
  if (stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    TGP_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  int lo = 0, hi = 0;
  TGP_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  {
    int can = 0;  // stream memory operations (chain_polls = 3: followers behind hipStreamWaitValue32)
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess) can = 0;
    (void)hipGetLastError();
    ctx->can_wait_value = can == 1;
  }
  {
    // A profiler that collects hardware counters (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION in the profiled
    // process) runs kernels one at a time in an order of its own: followers that wait for a chain launch of ANOTHER
    // stream -- by a poll kernel or by a stream wait-value -- can then be scheduled in front of it for good (measured:
    // profiles/r05_b).  Under such a tool the followers go behind the whole launch (chain_polls = 0) unless the caller
    // sets the option; TGP_SERIALIZED_KERNELS=1 says the same for tools this check does not know.
    const char* a = getenv("ROCPROF_COUNTER_COLLECTION");
    const char* b = getenv("TGP_SERIALIZED_KERNELS");
    auto on = [](const char* v) { return v != nullptr && *v && strcmp(v, "0") != 0 && strcasecmp(v, "false") != 0; };
    if (on(a) || on(b)) {
      ctx->serializing_tool = true;
      ctx->chain_polls = 0;
    }
  }
  TGP_HIP_TRY(hipStreamCreateWithPriority(&ctx->panel_stream, hipStreamNonBlocking, hi));
  // (the solve stream is created on first use -- ensure_solve_stream: a FIFTH stream in use costs every
  // dependent launch of the panel chains, profiles/r02_m_stream_count.txt, and the default schedule
  // does not need it)
  TGP_HIP_TRY(hipStreamCreateWithPriority(&ctx->update_stream, hipStreamNonBlocking, hi));
  TGP_HIP_TRY(hipStreamCreateWithPriority(&ctx->asm_stream, hipStreamNonBlocking, lo));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_asm, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_d, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_e, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_f, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_g1, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_g2, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_h, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_i, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_j, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_c, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
  TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
  TGP_HIP_TRY(hipMalloc(&ctx->d_scal, 16 * sizeof(double)));
  TGP_HIP_TRY(hipMalloc(&ctx->d_info, sizeof(int32_t)));
  TGP_HIP_TRY(hipMalloc(&ctx->d_step_flag, 64));
  TGP_HIP_TRY(hipMemset(ctx->d_step_flag, 0, 64));
  TGP_HIP_TRY(hipMalloc(&ctx->d_chain_flags, size_t(tgp::CHAIN_MAX_ROW_TILES) * 64 * sizeof(uint32_t)));
  TGP_HIP_TRY(hipMemset(ctx->d_chain_flags, 0, size_t(tgp::CHAIN_MAX_ROW_TILES) * 64 * sizeof(uint32_t)));
  TGP_HIP_TRY(hipMalloc(&ctx->d_chain_ticket, 4096));  // (CHAIN_TICKET_WORDS of chol.hip: 544 words)
  TGP_HIP_TRY(hipMemset(ctx->d_chain_ticket, 0, 4096));
  TGP_HIP_TRY(hipMalloc(&ctx->d_chain_red, size_t(2) * CHAIN_MAX_ROW_TILES * sizeof(double)));
  hipDeviceProp_t prop;
  TGP_HIP_TRY(hipGetDeviceProperties(&prop, device));
  ctx->cus = prop.multiProcessorCount;
  *out = ctx;
  return TGP_OK;
}

int tgp_ctx_destroy(tgp_ctx* ctx) {
  if (!ctx) return TGP_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->panel_stream) hipStreamSynchronize(ctx->panel_stream);
  if (ctx->solve_stream) {
    hipStreamSynchronize(ctx->solve_stream);
    hipStreamDestroy(ctx->solve_stream);
  }
  if (ctx->ev_c) hipEventDestroy(ctx->ev_c);
  if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
  if (ctx->ev_asm_gate) hipEventDestroy(ctx->ev_asm_gate);
  if (ctx->rescue_stream) hipStreamDestroy(ctx->rescue_stream);
  if (ctx->update_stream) {
    hipStreamSynchronize(ctx->update_stream);
    hipStreamDestroy(ctx->update_stream);
  }
  if (ctx->asm_stream) {
    hipStreamSynchronize(ctx->asm_stream);
    hipStreamDestroy(ctx->asm_stream);
  }
  if (ctx->ev_asm) hipEventDestroy(ctx->ev_asm);
  if (ctx->ev_d) hipEventDestroy(ctx->ev_d);
  if (ctx->ev_e) hipEventDestroy(ctx->ev_e);
  if (ctx->ev_f) hipEventDestroy(ctx->ev_f);
  if (ctx->ev_g1) hipEventDestroy(ctx->ev_g1);
  if (ctx->ev_g2) hipEventDestroy(ctx->ev_g2);
  if (ctx->ev_h) hipEventDestroy(ctx->ev_h);
  if (ctx->ev_i) hipEventDestroy(ctx->ev_i);
  if (ctx->ev_j) hipEventDestroy(ctx->ev_j);
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  if (ctx->ev_a) hipEventDestroy(ctx->ev_a);
  if (ctx->ev_b) hipEventDestroy(ctx->ev_b);
  if (ctx->d_scal) hipFree(ctx->d_scal);
  if (ctx->d_info) hipFree(ctx->d_info);
  if (ctx->d_step_flag) hipFree(ctx->d_step_flag);
  if (ctx->d_chain_flags) hipFree(ctx->d_chain_flags);
  if (ctx->d_chain_ticket) hipFree(ctx->d_chain_ticket);
  if (ctx->d_chain_red) hipFree(ctx->d_chain_red);
  if (ctx->d_chain_stamps) hipFree(ctx->d_chain_stamps);
  for (auto& kv : ctx->chain_tables)
    if (kv.second.dev) hipFree(kv.second.dev);
  if (ctx->d_dinv) hipFree(ctx->d_dinv);
  if (ctx->d_work) hipFree(ctx->d_work);
  if (ctx->d_gemm_ws) hipFree(ctx->d_gemm_ws);
  if (ctx->panel_stream) hipStreamDestroy(ctx->panel_stream);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return TGP_OK;
}

int tgp_ctx_sync(tgp_ctx* ctx) {
  CTX_GUARD(ctx);
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

// option name -> field of the context (NULL: unknown)
static int64_t* option_slot(tgp_ctx* ctx, const char* key) {
  if (!strcmp(key, "nb_outer")) return &ctx->nb_outer;
  if (!strcmp(key, "lookahead")) return &ctx->lookahead;
  if (!strcmp(key, "profile")) return &ctx->profile;
  if (!strcmp(key, "first_split")) return &ctx->first_split;
  if (!strcmp(key, "first_small_tiles")) return &ctx->first_small_tiles;
  if (!strcmp(key, "keep_grad_buffers")) return &ctx->keep_grad_buffers;
  if (!strcmp(key, "stream_trsv")) return &ctx->stream_trsv;
  if (!strcmp(key, "trsv_groups")) return &ctx->trsv_groups;
  if (!strcmp(key, "nb_wide_rows")) return &ctx->nb_wide_rows;
  if (!strcmp(key, "dist_solve_aux")) return &ctx->dist_solve_aux;
  if (!strcmp(key, "solve_on_update")) return &ctx->solve_on_update;
  if (!strcmp(key, "fused_step")) return &ctx->fused_step;
  if (!strcmp(key, "chain_kernel")) return &ctx->chain_kernel;
  if (!strcmp(key, "chain_stamps")) return &ctx->chain_stamps;
  if (!strcmp(key, "chain_full_rows")) return &ctx->chain_full_rows;
  if (!strcmp(key, "chain_lds_pad")) return &ctx->chain_lds_pad;
  if (!strcmp(key, "chain_depth2")) return &ctx->chain_depth2;
  if (!strcmp(key, "chain_merged")) return &ctx->chain_merged;
  if (!strcmp(key, "chain_sub_panel")) return &ctx->chain_sub_panel;
  if (!strcmp(key, "chain_sub_min_rows")) return &ctx->chain_sub_min_rows;
  if (!strcmp(key, "chain_sub_role")) return &ctx->chain_sub_role;
  if (!strcmp(key, "chain_pre_wait")) return &ctx->chain_pre_wait;
  if (!strcmp(key, "chain_gate_split")) return &ctx->chain_gate_split;
  if (!strcmp(key, "chain_polls")) return &ctx->chain_polls;
  if (!strcmp(key, "chain_fwd_tasks")) return &ctx->chain_fwd_tasks;
  if (!strcmp(key, "chain_fast_update")) return &ctx->chain_fast_update;
  if (!strcmp(key, "chain_batch")) return &ctx->chain_batch;
  if (!strcmp(key, "chain_batch_lag")) return &ctx->chain_batch_lag;
  if (!strcmp(key, "chain_batch_rowlag")) return &ctx->chain_batch_rowlag;
  if (!strcmp(key, "chain_batch_minrows")) return &ctx->chain_batch_minrows;
  if (!strcmp(key, "kmat_plain_div")) return &ctx->kmat_plain_div;
  if (!strcmp(key, "chain_reserve")) return &ctx->chain_reserve;
  if (!strcmp(key, "gate_split")) return &ctx->gate_split;
  if (!strcmp(key, "reserve_max_tiles")) return &ctx->reserve_max_tiles;
  if (!strcmp(key, "sub_panel")) return &ctx->sub_panel;
  if (!strcmp(key, "sub_panel_min_rows")) return &ctx->sub_panel_min_rows;
  if (!strcmp(key, "nb_first")) return &ctx->nb_first;
  if (!strcmp(key, "split_tail")) return &ctx->split_tail;
  if (!strcmp(key, "gemm_role")) return &ctx->gemm_role;
  if (!strcmp(key, "host_join")) return &ctx->host_join;
  if (!strcmp(key, "late_join")) return &ctx->late_join;
  if (!strcmp(key, "chain_reduce")) return &ctx->chain_reduce;
  if (!strcmp(key, "asm_defer")) return &ctx->asm_defer;
  if (!strcmp(key, "tile_band")) return &ctx->tile_band;
  if (!strcmp(key, "fault_inject")) return &ctx->fault_inject;
  if (!strcmp(key, "poll_timeout_ms")) return &ctx->poll_timeout_ms;
  if (!strcmp(key, "timeout_retries")) return &ctx->timeout_retries;  // (read: passes repeated after a device-side timeout)
  return nullptr;
}

static int set_option_checked(tgp_ctx* ctx, const char* key, int64_t value, int64_t* old) {
  int64_t* slot = option_slot(ctx, key);
  TGP_ARG_CHECK(slot != nullptr, "unknown option '%s'", key);
  if (slot == &ctx->nb_outer)
    TGP_ARG_CHECK(value >= TILE && value % TILE == 0, "nb_outer must be a positive multiple of %d", TILE);
  if (slot == &ctx->sub_panel || slot == &ctx->nb_first)
    TGP_ARG_CHECK(value >= 0 && value % TILE == 0, "%s must be a multiple of %d (0: off)", key, TILE);
  if (slot == &ctx->chain_batch) TGP_ARG_CHECK(value >= 0 && value <= 32, "chain_batch must be in [0, 32]");
  if (slot == &ctx->chain_batch_lag) TGP_ARG_CHECK(value >= 1 && value <= 64, "chain_batch_lag must be in [1, 64]");
  if (slot == &ctx->chain_batch_minrows) TGP_ARG_CHECK(value >= 0 && value <= 4096, "chain_batch_minrows must be in [0, 4096]");
  if (slot == &ctx->chain_batch_rowlag) TGP_ARG_CHECK(value >= 2 && value <= 4096, "chain_batch_rowlag must be in [2, 4096]");
  if (old) *old = *slot;
  // (the bound lives in a device global of the library: it holds for every context of the process on this device)
  if (slot == &ctx->poll_timeout_ms && ctx->has_device) return set_poll_limit(ctx, value);
  *slot = value;
  return TGP_OK;
}

int tgp_ctx_set_option(tgp_ctx* ctx, const char* key, int64_t value, int64_t* old) {
  TGP_ARG_CHECK(ctx != nullptr && key != nullptr, "null argument");
  return set_option_checked(ctx, key, value, old);
}

int tgp_ctx_get_option(tgp_ctx* ctx, const char* key, int64_t* value) {
  TGP_ARG_CHECK(ctx != nullptr && key != nullptr && value != nullptr, "null argument");
  int64_t* slot = option_slot(ctx, key);
  TGP_ARG_CHECK(slot != nullptr, "unknown option '%s'", key);
  if (slot == &ctx->poll_timeout_ms && ctx->has_device) ctx->poll_timeout_ms = tgp::poll_limit_ms();  // (one value per process)
  *value = *slot;
  return TGP_OK;
}

// "key=value,key=value" (the format of TGP_HIP_OPTIONS) applied to a context
static int apply_options(tgp_ctx* ctx, const char* options) {
  if (options == nullptr) return TGP_OK;
  std::string all(options);
  size_t pos = 0;
  while (pos < all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    const std::string item = all.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    TGP_ARG_CHECK(eq != std::string::npos && eq > 0, "option '%s' is not key=value", item.c_str());
    char* stop = nullptr;
    const long long v = strtoll(item.c_str() + eq + 1, &stop, 10);
    TGP_ARG_CHECK(stop != nullptr && *stop == 0 && eq + 1 < item.size(), "option '%s': bad integer", item.c_str());
    TGP_TRY(set_option_checked(ctx, item.substr(0, eq).c_str(), (int64_t)v, nullptr));
  }
  return TGP_OK;
}

int tgp_ctx_device_info(tgp_ctx* ctx, char* name, int name_len, int32_t* cus, int64_t* mem_bytes,
                        int32_t* clock_khz) {
  CTX_GUARD(ctx);
  hipDeviceProp_t prop;
  TGP_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_len > 0) {
    snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cus) *cus = prop.multiProcessorCount;
  if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
  if (clock_khz) *clock_khz = prop.clockRate;
  return TGP_OK;
}

int tgp_malloc(tgp_ctx* ctx, size_t bytes, void** dev) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(dev != nullptr, "null output pointer");
  TGP_HIP_TRY(hipMalloc(dev, bytes ? bytes : 8));
  return TGP_OK;
}
int tgp_free(tgp_ctx* ctx, void* dev) {
  CTX_GUARD(ctx);
  if (dev) {
    TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    TGP_HIP_TRY(hipFree(dev));
  }
  return TGP_OK;
}
int tgp_memcpy_h2d(tgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  CTX_GUARD(ctx);
  TGP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}
int tgp_memcpy_d2h(tgp_ctx* ctx, void* dst, const void* src, size_t bytes) {
  CTX_GUARD(ctx);
  TGP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}
int tgp_memset(tgp_ctx* ctx, void* dst, int byte, size_t bytes) {
  CTX_GUARD(ctx);
  TGP_HIP_TRY(hipMemsetAsync(dst, byte, bytes, ctx->stream));
  return TGP_OK;
}

// ---- device-pointer kernels ----------------------------------------------------------
int tgp_kmat(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
             int32_t d, const void* X1, const void* X2, const void* diag, void* out, int64_t ld,
             int64_t rows_out, int64_t cols_out, int lower_only) {
  CTX_GUARD(ctx);
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  TGP_ARG_CHECK(n1 >= 0 && n2 >= 0 && X1 && X2 && out, "kmat: null or negative argument");
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    const int flags = (lower_only ? KMAT_LOWER : 0) | KMAT_PAD_IDENTITY;
    return launch_kmat<T>(ctx, kp, n1, n2, d, (const T*)X1, (const T*)X2, (const T*)diag, (T*)out,
                          ld, rows_out, cols_out, flags);
  });
}

int tgp_kdiag(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n, int32_t d,
              const void* X, void* out) {
  CTX_GUARD(ctx);
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kdiag<T>(ctx, kp, n, d, (const T*)X, (const T*)nullptr, (T*)out);
  });
}

int tgp_kmat_gemv(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
                  int32_t d, const void* X1, const void* X2, const void* v, void* out) {
  CTX_GUARD(ctx);
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kmat_gemv<T>(ctx, kp, n1, n2, d, (const T*)X1, (const T*)X2, (const T*)v, (T*)out);
  });
}

int tgp_kmat_gemv_multi(tgp_ctx* ctx, int dtype, const tgp_kop* prog, int nops, int64_t n1, int64_t n2,
                        int32_t d, const void* X1, const void* X2, const void* V, int64_t nv, void* out) {
  CTX_GUARD(ctx);
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  TGP_ARG_CHECK(nv >= 1 && X1 && X2 && V && out, "kmat_gemv_multi: bad argument");
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kmat_gemv_multi<T>(ctx, kp, n1, n2, d, (const T*)X1, (const T*)X2, (const T*)V, nv, (T*)out);
  });
}

int tgp_potrf(tgp_ctx* ctx, int dtype, int64_t n, void* A, int64_t ld, int32_t* info) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(A != nullptr && n >= 0, "potrf: null matrix");
  TGP_TRY(ensure_dinv(ctx, size_t(n / TILE + 1) * 2048 * esize(dtype)));
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return potrf<T>(ctx, n, (T*)A, ld, (T*)ctx->d_dinv, info);
  });
}

int tgp_trsv(tgp_ctx* ctx, int dtype, int64_t n, const void* L, int64_t ld, int transpose, void* y) {
  CTX_GUARD(ctx);
  TGP_TRY(ensure_dinv(ctx, size_t(n / TILE + 1) * 2048 * esize(dtype)));
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    TGP_TRY(compute_dinv<T>(ctx, n, (const T*)L, ld, (T*)ctx->d_dinv));
    return trsv<T>(ctx, n, (const T*)L, ld, (const T*)ctx->d_dinv, transpose, (T*)y);
  });
}

int tgp_trsm_right_lt(tgp_ctx* ctx, int dtype, int64_t m, int64_t n, const void* L, int64_t ldl,
                      void* B, int64_t ldb) {
  CTX_GUARD(ctx);
  TGP_TRY(ensure_dinv(ctx, size_t(n / TILE + 1) * 2048 * esize(dtype)));
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    TGP_TRY(compute_dinv<T>(ctx, n, (const T*)L, ldl, (T*)ctx->d_dinv));
    return trsm_right_lt<T>(ctx, m, n, (const T*)L, ldl, (const T*)ctx->d_dinv, (T*)B, ldb);
  });
}

int tgp_gemv_sub(tgp_ctx* ctx, int dtype, int64_t m, int64_t k, const void* P, int64_t ld,
                 const void* x, void* y) {
  CTX_GUARD(ctx);
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return gemv_sub<T>(ctx, m, k, (const T*)P, ld, (const T*)x, (T*)y);
  });
}

int tgp_gemm_nt(tgp_ctx* ctx, int dtype, int64_t m, int64_t n, int64_t k, double alpha,
                const void* A, int64_t lda, const void* B, int64_t ldb, double beta, void* C,
                int64_t ldc, int lower) {
  CTX_GUARD(ctx);
  int mode;
  if (alpha == -1.0 && beta == 1.0) mode = 0;
  else if (alpha == 1.0 && beta == 0.0) mode = 1;
  else {
    set_error("gemm_nt: (alpha, beta) must be (-1, 1) or (1, 0)");
    return TGP_E_UNSUPPORTED;
  }
  return dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_gemm_nt<T>(ctx, ctx->stream, m, n, k, (const T*)A, lda, (const T*)B, ldb, (T*)C,
                             ldc, lower, mode, (int)ctx->gemm_role);
  });
}

int tgp_sum_log_diag(tgp_ctx* ctx, int dtype, int64_t n, const void* L, int64_t ld, double* out) {
  CTX_GUARD(ctx);
  TGP_TRY(dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_sum_log_diag<T>(ctx, n, (const T*)L, ld, 0);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(out, ctx->d_scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_sum_squares(tgp_ctx* ctx, int dtype, int64_t n, const void* y, double* out) {
  CTX_GUARD(ctx);
  TGP_TRY(dispatch(dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_sum_squares<T>(ctx, n, (const T*)y, 0);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(out, ctx->d_scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_ubench_mfma(tgp_ctx* ctx, int dtype, double* tflops_out) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(tflops_out != nullptr && (dtype == TGP_F32 || dtype == TGP_F64), "bad argument");
  return ubench_mfma(ctx, dtype, tflops_out);
}

int tgp_ubench(tgp_ctx* ctx, int kind, int blocks_per_cu, double* tflops_out,
               double* cycles_per_op_out) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(tflops_out != nullptr, "null output pointer");
  return ubench(ctx, kind, blocks_per_cu, tflops_out, cycles_per_op_out);
}

// ---- solver handle ----------------------------------------------------------------------
static int solver_scratch(tgp_solver* s, size_t bytes) {
  return tgp::grow(&s->scratch, &s->scratch_bytes, bytes);
}

int tgp_solver_create(tgp_ctx* ctx, int dtype, int64_t n, int32_t d, const void* X_host,
                      const void* noise_diag_host, tgp_solver** out) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(out != nullptr, "null output pointer");
  TGP_ARG_CHECK(dtype == TGP_F32 || dtype == TGP_F64, "dtype must be TGP_F32 or TGP_F64");
  TGP_ARG_CHECK(n >= 1, "need at least one data point (n = %lld)", (long long)n);
  TGP_ARG_CHECK(d >= 1 && d <= TGP_MAX_DIM, "input dimension must be 1..%d (got %d)", TGP_MAX_DIM, d);
  TGP_ARG_CHECK(X_host != nullptr && noise_diag_host != nullptr, "null input array");
  tgp_solver* s = new tgp_solver();
  s->ctx = ctx;
  s->dtype = dtype;
  s->n = n;
  s->npad = round_up(n, TILE);
  s->d = d;
  const size_t es = esize(dtype);
  auto fail = [&](int code) { tgp_solver_destroy(s); return code; };
#define S_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    set_error("%s failed: %s", #expr, hipGetErrorString(_e));            \
    return fail(_e == hipErrorOutOfMemory ? TGP_E_NOMEM : TGP_E_HIP); } } while (0)
  S_TRY(hipMalloc(&s->X, size_t(n) * d * es));
  S_TRY(hipMalloc(&s->diag, size_t(n) * es));
  S_TRY(hipMalloc(&s->A, size_t(s->npad) * s->npad * es));
  S_TRY(hipMalloc(&s->dinv, size_t(s->npad / TILE) * 2048 * es));
  S_TRY(hipMalloc(&s->vec, size_t(s->npad) * es));
  S_TRY(hipMalloc(&s->vec2, size_t(s->npad) * es));
  S_TRY(hipMalloc(&s->resid, size_t(s->npad) * es));
  S_TRY(hipMemcpyAsync(s->X, X_host, size_t(n) * d * es, hipMemcpyHostToDevice, ctx->stream));
  S_TRY(hipMemcpyAsync(s->diag, noise_diag_host, size_t(n) * es, hipMemcpyHostToDevice, ctx->stream));
  S_TRY(hipMemsetAsync(s->vec, 0, size_t(s->npad) * es, ctx->stream));
  S_TRY(hipMemsetAsync(s->vec2, 0, size_t(s->npad) * es, ctx->stream));
  S_TRY(hipMemsetAsync(s->resid, 0, size_t(s->npad) * es, ctx->stream));
  S_TRY(hipStreamSynchronize(ctx->stream));
#undef S_TRY
  *out = s;
  return TGP_OK;
}

int tgp_solver_destroy(tgp_solver* s) {
  if (!s) return TGP_OK;
  if (s->ctx) {
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
  }
  void* bufs[] = {s->X, s->diag, s->A, s->dinv, s->vec, s->vec2, s->resid, s->scratch, s->Minv, s->winv};
  for (void* b : bufs)
    if (b) hipFree(b);
  delete s;
  return TGP_OK;
}

int tgp_solver_set_noise(tgp_solver* s, const void* noise_diag_host) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(noise_diag_host != nullptr, "null input array");
  TGP_HIP_TRY(hipMemcpyAsync(s->diag, noise_diag_host, size_t(s->n) * esize(s->dtype),
                             hipMemcpyHostToDevice, s->ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
  return TGP_OK;
}

// upload a host vector (n,) into a zero-padded device vector
static int upload_vec(tgp_solver* s, void* dst, const void* src_host) {
  const size_t es = esize(s->dtype);
  TGP_HIP_TRY(hipMemcpyAsync(dst, src_host, size_t(s->n) * es, hipMemcpyHostToDevice, s->ctx->stream));
  if (s->npad > s->n)
    TGP_HIP_TRY(hipMemsetAsync((char*)dst + size_t(s->n) * es, 0, size_t(s->npad - s->n) * es,
                               s->ctx->stream));
  return TGP_OK;
}

// fused != 0: also alpha = L^-1 resid (into s->vec) overlapped with the factorisation and
// *logprob = -0.5 |alpha|^2 - normalization.  resid_host NULL -> the resident residual.
static int factor_body(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                       int32_t* info, int fused, const void* resid_host, double* logprob);

// An error return in the middle of the multi-stream schedule must not leave the context
// half-way: side streams are drained, the assembly marker is cleared and the profiling
// events go back to the pool, so that the next call starts from a clean state.
static int factor_impl(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                       int32_t* info, int fused, const void* resid_host, double* logprob) {
  int st = factor_body(s, prog, nops, cov_host, info, fused, resid_host, logprob);
  if (st == TGP_E_TIMEOUT && s->ctx->chain_kernel != 0) {
    // A device-side hand-off of the persistent chain timed out: a starved launch (a chip shared with other processes,
    // a tool that holds kernels back), not a numerical result.  The matrix was factored in place, so the pass is
    // repeated from the assembly -- ONCE, on the launch-per-block path, which has no device-side waits at all.
    tgp_ctx* ctx = s->ctx;
    for (hipStream_t q : {ctx->asm_stream, ctx->panel_stream, ctx->update_stream, ctx->solve_stream, ctx->stream})
      if (q) (void)hipStreamSynchronize(q);
    (void)hipGetLastError();
    ctx->asm_pending = false;
    ctx->deferred_asm = nullptr;
    ctx->ev_used = 0;
    ctx->timeout_retries++;
    const int64_t keep = ctx->chain_kernel;
    ctx->chain_kernel = 0;
    st = factor_body(s, prog, nops, cov_host, info, fused, resid_host, logprob);
    ctx->chain_kernel = keep;
  }
  if (st < 0) {
    tgp_ctx* ctx = s->ctx;
    ctx->asm_pending = false;
    ctx->deferred_asm = nullptr;
    ctx->ev_used = 0;
    s->factored = false;
    for (hipStream_t q : {ctx->asm_stream, ctx->panel_stream, ctx->update_stream, ctx->solve_stream,
                          ctx->stream})
      if (q) (void)hipStreamSynchronize(q);
    (void)hipGetLastError();
  }
  return st;
}

static int factor_body(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                       int32_t* info, int fused, const void* resid_host, double* logprob) {
  tgp_ctx* ctx = s->ctx;
  if (nops > 0) {
    TGP_TRY(make_kprog(prog, nops, &s->kp));
    s->has_prog = true;
  } else {
    s->has_prog = false;
  }
  TGP_ARG_CHECK(s->has_prog || cov_host != nullptr, "factor needs a kernel program or a covariance");
  s->winv_valid = false;
  const auto host0 = std::chrono::steady_clock::now();  // -> ms[5]: how long the HOST takes to submit the evaluation
  const size_t es = esize(s->dtype);
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  const bool prof = ctx->profile != 0;
  if (prof) {
    TGP_HIP_TRY(hipEventCreate(&e0));
    TGP_HIP_TRY(hipEventCreate(&e1));
    TGP_HIP_TRY(hipEventCreate(&e2));
    TGP_HIP_TRY(hipEventRecord(e0, ctx->stream));
  }
  int status = dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    T* A = (T*)s->A;
    // the residual goes to the work vector FIRST: queued behind the assembly, the 131-KB copy ran beside the side
    // stream's assembly of the other columns and took 170-240 us -- on the main stream, in front of the marker the
    // first panel's chain waits for (profiles/r04_c)
    if (fused) {
      if (resid_host) {
        TGP_TRY(upload_vec(s, s->vec, resid_host));
      } else {
        TGP_ARG_CHECK(s->has_resid, "no resident residual: call tgp_solver_set_resid first");
        TGP_HIP_TRY(hipMemcpyAsync(s->vec, s->resid, size_t(s->npad) * es, hipMemcpyDeviceToDevice,
                                   ctx->stream));
      }
    }
    if (cov_host != nullptr) {
      TGP_TRY(solver_scratch(s, size_t(s->n) * s->n * es));
      TGP_HIP_TRY(hipMemcpyAsync(s->scratch, cov_host, size_t(s->n) * s->n * es,
                                 hipMemcpyHostToDevice, ctx->stream));
      TGP_TRY(launch_set_lower_from_rowmajor<T>(ctx, s->n, s->npad, (const T*)s->scratch, A, s->npad));
    } else {
      TGP_TRY(assemble_lower<T>(ctx, s->kp, s->n, s->d, (const T*)s->X, (const T*)s->diag, A, s->npad));
    }
    if (prof) TGP_HIP_TRY(hipEventRecord(e1, ctx->stream));
    int32_t inf = 0;
    ctx->defer_join = ctx->late_join != 0;  // (one host round trip per evaluation: `info` comes back with the scalars below)
    ctx->join_deferred = false;
    // (the chain's fsolve tasks leave both sums when the forward substitution rides in the chain launches: chol.hip)
    ctx->chain_red_total = fused && ctx->chain_reduce != 0 ? s->npad / 128 : 0;
    ctx->reductions_done = false;
    int st = potrf<T>(ctx, s->npad, A, s->npad, (T*)s->dinv, &inf, fused ? (T*)s->vec : (T*)nullptr);
    ctx->defer_join = false;
    ctx->chain_red_total = 0;
    if (st < 0) return st;
    s->info = inf;
    if (prof) TGP_HIP_TRY(hipEventRecord(e2, ctx->stream));
    if (!ctx->reductions_done) {
      TGP_TRY(launch_sum_log_diag<T>(ctx, s->n, A, s->npad, 1));
      if (fused) TGP_TRY(launch_sum_squares<T>(ctx, s->n, (const T*)s->vec, 0));
    }
    ctx->reductions_done = false;
    return TGP_OK;
  });
  if (status < 0) return status;
  double two[2] = {0, 0};
  int32_t inf_late = 0;
  const bool late = ctx->join_deferred;
  ctx->join_deferred = false;
  TGP_HIP_TRY(hipMemcpyAsync(two, ctx->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (late) TGP_HIP_TRY(hipMemcpyAsync(&inf_late, ctx->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (late) {  // what potrf does behind its own join: a timed-out hand-off is an error (retried by factor_impl), a pivot is `info`
    if (inf_late == INT32_MIN) {
      tgp::set_error("potrf: a device-side hand-off did not arrive within poll_timeout_ms (a lost or starved chain launch)");
      if (prof) { hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2); }
      return TGP_E_TIMEOUT;
    }
    s->info = inf_late;
  }
  s->ms[5] = std::chrono::duration<double, std::milli>(ctx->submitted - host0).count();  // (potrf's last enqueue)
  s->logdet_half = two[1];
  if (fused && logprob)
    *logprob = -0.5 * two[0] - (s->logdet_half + 0.5 * double(s->n) * std::log(2.0 * M_PI));
  if (prof) {
    float a = 0, b = 0;
    TGP_HIP_TRY(hipEventElapsedTime(&a, e0, e1));
    TGP_HIP_TRY(hipEventElapsedTime(&b, e1, e2));
    s->ms[0] = a;
    s->ms[1] = b;
    s->ms[2] = ctx->prof_syrk_ms;
    s->ms[3] = (double)ctx->prof_syrk_launches;
    s->ms[6] = ctx->prof_syrk_flops;
    s->ms[7] = ctx->prof_syrk_union_ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipEventDestroy(e2);
  }
  s->factored = true;
  if (info) *info = s->info;
  return s->info > 0 ? s->info : TGP_OK;
}

int tgp_solver_factor(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                      int32_t* info) {
  SOLVER_GUARD(s);
  return factor_impl(s, prog, nops, cov_host, info, 0, nullptr, nullptr);
}

int tgp_solver_factor_logprob(tgp_solver* s, const tgp_kop* prog, int nops, const void* cov_host,
                              const void* resid_host, int32_t* info, double* logprob) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(logprob != nullptr, "null output pointer");
  return factor_impl(s, prog, nops, cov_host, info, 1, resid_host, logprob);
}

#define NEED_FACTOR(s) TGP_ARG_CHECK((s)->factored, "solver has not been factored yet")

int tgp_solver_normalization(tgp_solver* s, double* out) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(out != nullptr, "null output pointer");
  *out = s->logdet_half + 0.5 * double(s->n) * std::log(2.0 * M_PI);
  return TGP_OK;
}

int tgp_solver_solve_tri(tgp_solver* s, int transpose, int64_t nrhs, const void* y_host,
                         void* x_host) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(nrhs >= 1 && y_host && x_host, "solve_tri: bad argument");
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  return dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* L = (const T*)s->A;
    if (nrhs == 1) {
      TGP_TRY(upload_vec(s, s->vec, y_host));
      const T* winv = nullptr;
      if (ctx->stream_trsv != 0 && s->info == 0) {
        TGP_TRY(ensure_winv<T>(s));
        winv = (const T*)s->winv;
      }
      TGP_TRY(trsv<T>(ctx, s->npad, L, s->npad, (const T*)s->dinv, transpose, (T*)s->vec, winv));
      TGP_HIP_TRY(hipMemcpyAsync(x_host, s->vec, size_t(s->n) * es, hipMemcpyDeviceToHost, ctx->stream));
      TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
      return TGP_OK;
    }
    if (!transpose) {
      // host (n, nrhs) row-major == column-major (nrhs x n): the transposed form B L^-T
      const int64_t mpad = round_up(nrhs, TILE);
      TGP_TRY(solver_scratch(s, size_t(mpad) * s->npad * es));
      TGP_HIP_TRY(hipMemsetAsync(s->scratch, 0, size_t(mpad) * s->npad * es, ctx->stream));
      TGP_HIP_TRY(hipMemcpy2DAsync(s->scratch, size_t(mpad) * es, y_host, size_t(nrhs) * es,
                                   size_t(nrhs) * es, size_t(s->n), hipMemcpyHostToDevice, ctx->stream));
      TGP_TRY(trsm_right_lt<T>(ctx, mpad, s->npad, L, s->npad, (const T*)s->dinv, (T*)s->scratch, mpad));
      TGP_HIP_TRY(hipMemcpy2DAsync(x_host, size_t(nrhs) * es, s->scratch, size_t(mpad) * es,
                                   size_t(nrhs) * es, size_t(s->n), hipMemcpyDeviceToHost, ctx->stream));
      TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
      return TGP_OK;
    }
    // L^T X = Y with several right-hand sides: one backward sweep per column
    const T* winv_t = nullptr;
    if (ctx->stream_trsv != 0 && s->info == 0) {
      TGP_TRY(ensure_winv<T>(s));
      winv_t = (const T*)s->winv;
    }
    for (int64_t r = 0; r < nrhs; ++r) {
      TGP_HIP_TRY(hipMemsetAsync(s->vec, 0, size_t(s->npad) * es, ctx->stream));
      TGP_HIP_TRY(hipMemcpy2DAsync(s->vec, es, (const char*)y_host + r * es, size_t(nrhs) * es, es,
                                   size_t(s->n), hipMemcpyHostToDevice, ctx->stream));
      TGP_TRY(trsv<T>(ctx, s->npad, L, s->npad, (const T*)s->dinv, 1, (T*)s->vec, winv_t));
      TGP_HIP_TRY(hipMemcpy2DAsync((char*)x_host + r * es, size_t(nrhs) * es, s->vec, es, es,
                                   size_t(s->n), hipMemcpyDeviceToHost, ctx->stream));
    }
    TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TGP_OK;
  });
}

int tgp_solver_dot_tri(tgp_solver* s, int64_t nrhs, const void* y_host, void* out_host) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(nrhs >= 1 && y_host && out_host, "dot_tri: bad argument");
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  return dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    if (nrhs >= 8) {
      // (L Z)^T = Z^T L^T as one MFMA GEMM: the host (n, R) row-major array IS the
      // column-major (R x n) operand, and so is the result; the k-loop stops at the
      // diagonal tile (TRMM), so the never-written upper tiles of L are not touched.
      const int64_t mpad = round_up(nrhs, TILE);
      const size_t bytes = size_t(mpad) * s->npad * es;
      TGP_TRY(solver_scratch(s, 2 * bytes));
      char* Zt = (char*)s->scratch;
      char* Ct = Zt + bytes;
      TGP_HIP_TRY(hipMemsetAsync(Zt, 0, bytes, ctx->stream));
      TGP_HIP_TRY(hipMemcpy2DAsync(Zt, size_t(mpad) * es, y_host, size_t(nrhs) * es, size_t(nrhs) * es,
                                   size_t(s->n), hipMemcpyHostToDevice, ctx->stream));
      TGP_TRY(launch_gemm_nt<T>(ctx, ctx->stream, mpad, s->npad, s->npad, (const T*)Zt, mpad,
                                (const T*)s->A, s->npad, (T*)Ct, mpad, 0, 3, 1));
      TGP_HIP_TRY(hipMemcpy2DAsync(out_host, size_t(nrhs) * es, Ct, size_t(mpad) * es, size_t(nrhs) * es,
                                   size_t(s->n), hipMemcpyDeviceToHost, ctx->stream));
      TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
      return TGP_OK;
    }
    for (int64_t r = 0; r < nrhs; ++r) {
      TGP_HIP_TRY(hipMemcpy2DAsync(s->vec, es, (const char*)y_host + r * es, size_t(nrhs) * es, es,
                                   size_t(s->n), hipMemcpyHostToDevice, ctx->stream));
      TGP_TRY(launch_trmv_lower<T>(ctx, s->n, (const T*)s->A, s->npad, (const T*)s->vec, (T*)s->vec2));
      TGP_HIP_TRY(hipMemcpy2DAsync((char*)out_host + r * es, size_t(nrhs) * es, s->vec2, es, es,
                                   size_t(s->n), hipMemcpyDeviceToHost, ctx->stream));
    }
    TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TGP_OK;
  });
}

int tgp_solver_set_resid(tgp_solver* s, const void* resid_host) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(resid_host != nullptr, "null input array");
  TGP_TRY(upload_vec(s, s->resid, resid_host));
  TGP_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
  s->has_resid = true;
  return TGP_OK;
}

// alpha = L^-1 resid into s->vec; returns -0.5 |alpha|^2 - normalization
static int logprob_device(tgp_solver* s, const void* resid_host, double* out) {
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  // the resident residual is read in place by the streaming solve (no copy into the work vector); the launch-per-block
  // path (stream_trsv = 0, or a factor with a failed pivot) solves in place and needs the copy
  const bool in_place_rhs = !resid_host && ctx->stream_trsv != 0 && s->info == 0;
  if (resid_host) {
    TGP_TRY(upload_vec(s, s->vec, resid_host));
  } else {
    TGP_ARG_CHECK(s->has_resid, "no resident residual: call tgp_solver_set_resid first");
    if (!in_place_rhs)
      TGP_HIP_TRY(hipMemcpyAsync(s->vec, s->resid, size_t(s->npad) * es, hipMemcpyDeviceToDevice, ctx->stream));
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool prof = ctx->profile != 0;
  if (prof) {
    TGP_HIP_TRY(hipEventCreate(&e0));
    TGP_HIP_TRY(hipEventCreate(&e1));
    TGP_HIP_TRY(hipEventRecord(e0, ctx->stream));
  }
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* winv = nullptr;
    if (ctx->stream_trsv != 0 && s->info == 0) {
      TGP_TRY(ensure_winv<T>(s));
      winv = (const T*)s->winv;
    }
    TGP_TRY(trsv<T>(ctx, s->npad, (const T*)s->A, s->npad, (const T*)s->dinv, 0, (T*)s->vec, winv,
                    in_place_rhs ? (const T*)s->resid : (const T*)nullptr));
    return launch_sum_squares<T>(ctx, s->n, (const T*)s->vec, 0);
  }));
  if (prof) TGP_HIP_TRY(hipEventRecord(e1, ctx->stream));
  double ss = 0;
  TGP_HIP_TRY(hipMemcpyAsync(&ss, ctx->d_scal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (prof) {
    float t = 0;
    TGP_HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    s->ms[4] = t;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  *out = -0.5 * ss - (s->logdet_half + 0.5 * double(s->n) * std::log(2.0 * M_PI));
  return TGP_OK;
}

int tgp_solver_logprob(tgp_solver* s, const void* resid_host, double* out) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(out != nullptr, "null output pointer");
  return logprob_device(s, resid_host, out);
}

int tgp_solver_alpha(tgp_solver* s, const void* resid_host, void* alpha_host, double* logprob) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(alpha_host != nullptr && logprob != nullptr, "null output pointer");
  TGP_TRY(logprob_device(s, resid_host, logprob));
  tgp_ctx* ctx = s->ctx;
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* winv = (ctx->stream_trsv != 0 && s->info == 0 && s->winv_valid) ? (const T*)s->winv : nullptr;
    return trsv<T>(ctx, s->npad, (const T*)s->A, s->npad, (const T*)s->dinv, 1, (T*)s->vec, winv);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(alpha_host, s->vec, size_t(s->n) * esize(s->dtype),
                             hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

// Gradient of the log-probability (the reference gets it from JAX autodiff through
// cholesky):  d ll / d theta = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij / d theta.
int tgp_solver_grad(tgp_solver* s, const void* resid_host, double* logprob, double* grad_params,
                    void* grad_noise_host, void* alpha_host, double* grad_logscale) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(s->has_prog, "gradients need a kernel program (not a host covariance)");
  TGP_ARG_CHECK(logprob && grad_params, "null output pointer");
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  const int64_t ldk = s->npad + TILE;
  const size_t mat = size_t(ldk) * s->npad * es;
  // ONE more matrix of (N_pad + 128) x N_pad: L^-1 in both orientations, then K^-1 in the lower one's place
  // (spd_inverse_lower).  Released again below unless the context option "keep_grad_buffers" is set (an optimiser
  // loop at moderate N), and always on failure: an out-of-memory here must not pin the device for the solver's
  // lifetime.
  auto release = [&]() {
    if (s->Minv) (void)hipFree(s->Minv);
    s->Minv = nullptr;
  };
  if (!s->Minv) {
    hipError_t e = hipMalloc(&s->Minv, mat);
    if (e != hipSuccess) {
      s->Minv = nullptr;
      (void)hipGetLastError();
      set_error("the gradient needs one %lld x %lld work matrix (%.1f GB) and the device is out of memory: %s",
                (long long)ldk, (long long)s->npad, double(mat) / 1e9, hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? TGP_E_NOMEM : TGP_E_HIP;
    }
  }
  int gst = logprob_device(s, resid_host, logprob);  // s->vec = L^-1 r
  if (gst < 0) {
    release();
    return gst;
  }
  std::vector<double> g(size_t(2 * s->kp.n), 0.0);
  std::vector<double> gdim(size_t(grad_logscale ? s->d : 0), 0.0);
  gst = (dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* L = (const T*)s->A;
    T* alpha = (T*)s->vec;
    TGP_TRY(trsv<T>(ctx, s->npad, L, s->npad, (const T*)s->dinv, 1, alpha,
                    (ctx->stream_trsv != 0 && s->info == 0 && s->winv_valid) ? (const T*)s->winv : (const T*)nullptr));  // alpha = K^-1 r
    TGP_TRY(ensure_winv<T>(s));
    TGP_TRY(spd_inverse_lower<T>(ctx, s->npad, L, s->npad, (const T*)s->winv, (T*)s->Minv, ldk));
    const T* Kinv = (const T*)s->Minv + TILE;  // lower tiles, leading dimension ldk
    int fleaf = -1, fkonst = -1;
    const int fast = launch_kgrad_fast<T>(ctx, s->kp, s->n, s->d, (const T*)s->X, (const T*)alpha,
                                          Kinv, ldk, ctx->d_scal + 2, &fleaf, &fkonst);
    if (fast < 0) return fast;
    if (fast == 1) {  // d_scal[2] = d/d constant, d_scal[3] = d/d scale
      if (fkonst >= 0)
        TGP_HIP_TRY(hipMemcpyAsync(&g[size_t(2 * fkonst)], ctx->d_scal + 2, sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
      TGP_HIP_TRY(hipMemcpyAsync(&g[size_t(2 * fleaf)], ctx->d_scal + 3, sizeof(double), hipMemcpyDeviceToHost,
                                 ctx->stream));
    }
    for (int i = 0; i < s->kp.n && fast == 0; ++i) {
      const int op = s->kp.op[i];
      if (op >= TGP_K_ADD) continue;
      const int nparam = (op == TGP_K_ESS || op == TGP_K_RQ) ? 2 : 1;
      for (int q = 0; q < nparam; ++q) {
        TGP_TRY(launch_kgrad<T>(ctx, s->kp, i, q, s->n, s->d, (const T*)s->X, (const T*)alpha,
                                Kinv, ldk, ctx->d_scal + 2));
        TGP_HIP_TRY(hipMemcpyAsync(&g[size_t(2 * i + q)], ctx->d_scal + 2, sizeof(double),
                                   hipMemcpyDeviceToHost, ctx->stream));
      }
    }
    // d / d log(scale of input dimension q): one pass per dimension over the same K^-1 (transforms with a
    // per-dimension scale; which_op = -1 - q selects the coordinate mode of the derivative kernel)
    for (int q = 0; q < (int)gdim.size(); ++q) {
      TGP_TRY(launch_kgrad<T>(ctx, s->kp, -1 - q, 0, s->n, s->d, (const T*)s->X, (const T*)alpha, Kinv, ldk,
                              ctx->d_scal + 2));
      TGP_HIP_TRY(hipMemcpyAsync(&gdim[size_t(q)], ctx->d_scal + 2, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (grad_noise_host) {
      TGP_TRY(launch_noise_grad<T>(ctx, s->n, (const T*)alpha, Kinv, ldk, (T*)s->vec2));
      TGP_HIP_TRY(hipMemcpyAsync(grad_noise_host, s->vec2, size_t(s->n) * es, hipMemcpyDeviceToHost,
                                 ctx->stream));
    }
    if (alpha_host)
      TGP_HIP_TRY(hipMemcpyAsync(alpha_host, alpha, size_t(s->n) * es, hipMemcpyDeviceToHost, ctx->stream));
    return TGP_OK;
  }));
  if (gst >= 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) {
    set_error("hipStreamSynchronize failed in tgp_solver_grad");
    gst = TGP_E_HIP;
  }
  // kept for the next call when it is small (an optimiser loop: allocating and freeing 2 GB cost more than the 52 ms
  // of the gradient itself on every other call) or when the option says so; a large one goes back at once
  const bool keep = ctx->keep_grad_buffers != 0 || mat <= (size_t(4) << 30);
  if (gst < 0 || !keep) {
    (void)hipStreamSynchronize(ctx->stream);
    release();
  }
  if (gst < 0) return gst;
  for (size_t i = 0; i < g.size(); ++i) grad_params[i] = g[i];
  for (size_t i = 0; i < gdim.size(); ++i) grad_logscale[i] = gdim[i];
  return TGP_OK;
}

int tgp_solver_cond_mean(tgp_solver* s, const tgp_kop* prog, int nops, int64_t m,
                         const void* Xt_host, const void* alpha_host, void* mean_host) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(m >= 1 && Xt_host && alpha_host && mean_host, "cond_mean: bad argument");
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  const size_t xt_bytes = round_up(size_t(m) * s->d * es, 256), out_bytes = round_up(size_t(m) * es, 256);
  TGP_TRY(solver_scratch(s, xt_bytes + out_bytes));
  char* xt = (char*)s->scratch;
  char* outv = xt + xt_bytes;
  TGP_HIP_TRY(hipMemcpyAsync(xt, Xt_host, size_t(m) * s->d * es, hipMemcpyHostToDevice, ctx->stream));
  TGP_TRY(upload_vec(s, s->vec2, alpha_host));
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kmat_gemv<T>(ctx, kp, m, s->n, s->d, (const T*)xt, (const T*)s->X,
                               (const T*)s->vec2, (T*)outv);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(mean_host, outv, size_t(m) * es, hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_solver_condition_cov(tgp_solver* s, const tgp_kop* prog, int nops, int64_t m,
                             const void* Xt_host, const void* noise_t_host, int var_only,
                             void* out_host) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(out_host != nullptr, "null output pointer");
  if (!Xt_host) m = s->n;
  TGP_ARG_CHECK(m >= 1, "condition: need at least one test point");
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  const int64_t mpad = round_up(m, TILE);
  // scratch layout: [Bp mpad x npad][Kss mpad x mpad | base,outv][Xt][noise]
  const size_t b_bytes = size_t(mpad) * s->npad * es;
  const size_t k_bytes = var_only ? 2 * size_t(mpad) * es : size_t(mpad) * mpad * es;
  const size_t xt_bytes = round_up(size_t(m) * s->d * es, 256);
  const size_t nz_bytes = round_up(size_t(mpad) * es, 256);
  TGP_TRY(solver_scratch(s, b_bytes + round_up(k_bytes, 256) + xt_bytes + nz_bytes));
  char* Bp = (char*)s->scratch;
  char* Kss = Bp + b_bytes;
  char* xt = Kss + round_up(k_bytes, 256);
  char* nz = xt + xt_bytes;
  const void* Xt = s->X;
  if (Xt_host) {
    TGP_HIP_TRY(hipMemcpyAsync(xt, Xt_host, size_t(m) * s->d * es, hipMemcpyHostToDevice, ctx->stream));
    Xt = xt;
  }
  const void* nzp = nullptr;
  if (noise_t_host) {
    TGP_HIP_TRY(hipMemcpyAsync(nz, noise_t_host, size_t(m) * es, hipMemcpyHostToDevice, ctx->stream));
    nzp = nz;
  }
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    // Bp[i, j] = k(Xt[i], X[j]) = Ks^T (direct.py:87-91), zero padding
    TGP_TRY(launch_kmat<T>(ctx, kp, m, s->n, s->d, (const T*)Xt, (const T*)s->X, (const T*)nullptr,
                           (T*)Bp, mpad, mpad, s->npad, 0));
    // rows of Bp <- (L^-1 k(X, x_t))^T  (A = L^-1 Ks, direct.py:94)
    TGP_TRY(trsm_right_lt<T>(ctx, mpad, s->npad, (const T*)s->A, s->npad, (const T*)s->dinv, (T*)Bp, mpad));
    if (var_only) {
      T* base = (T*)Kss;
      T* outv = base + mpad;
      TGP_TRY(launch_kdiag<T>(ctx, kp, m, s->d, (const T*)Xt, (const T*)nzp, base));
      TGP_TRY(launch_row_sumsq<T>(ctx, m, s->npad, (const T*)Bp, mpad, base, outv));
      TGP_HIP_TRY(hipMemcpyAsync(out_host, outv, size_t(m) * es, hipMemcpyDeviceToHost, ctx->stream));
    } else {
      // Kss + noise - A^T A (direct.py:92,95): SYRK-shaped MFMA GEMM, K = n
      TGP_TRY(launch_kmat<T>(ctx, kp, m, m, s->d, (const T*)Xt, (const T*)Xt, (const T*)nzp,
                             (T*)Kss, mpad, mpad, mpad, 0));
      // (lower tiles only -- the product is symmetric -- and the upper triangle mirrored from them: half the flops)
      TGP_TRY(launch_gemm_nt<T>(ctx, ctx->stream, mpad, mpad, s->npad, (const T*)Bp, mpad,
                                (const T*)Bp, mpad, (T*)Kss, mpad, 1, 0, 1));
      TGP_TRY(symmetrize_lower<T>(ctx, mpad, (T*)Kss, mpad));
      TGP_HIP_TRY(hipMemcpy2DAsync(out_host, size_t(m) * es, Kss, size_t(mpad) * es, size_t(m) * es,
                                   size_t(m), hipMemcpyDeviceToHost, ctx->stream));
    }
    return TGP_OK;
  }));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_solver_covariance(tgp_solver* s, void* out_host) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(s->has_prog, "covariance needs a kernel program");
  TGP_ARG_CHECK(out_host != nullptr, "null output pointer");
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  TGP_TRY(solver_scratch(s, size_t(s->n) * s->n * es));
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kmat<T>(ctx, s->kp, s->n, s->n, s->d, (const T*)s->X, (const T*)s->X,
                          (const T*)s->diag, (T*)s->scratch, s->n, s->n, s->n, 0);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(out_host, s->scratch, size_t(s->n) * s->n * es, hipMemcpyDeviceToHost,
                             ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_solver_variance(tgp_solver* s, void* out_host) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(s->has_prog, "variance needs a kernel program");
  TGP_ARG_CHECK(out_host != nullptr, "null output pointer");
  tgp_ctx* ctx = s->ctx;
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_kdiag<T>(ctx, s->kp, s->n, s->d, (const T*)s->X, (const T*)s->diag, (T*)s->vec2);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(out_host, s->vec2, size_t(s->n) * esize(s->dtype),
                             hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_solver_get_factor(tgp_solver* s, void* L_host) {
  SOLVER_GUARD(s);
  NEED_FACTOR(s);
  TGP_ARG_CHECK(L_host != nullptr, "null output pointer");
  tgp_ctx* ctx = s->ctx;
  const size_t es = esize(s->dtype);
  TGP_TRY(solver_scratch(s, size_t(s->n) * s->n * es));
  TGP_TRY(dispatch(s->dtype, [&](auto tag) {
    using T = decltype(tag);
    return launch_extract_lower_rowmajor<T>(ctx, s->n, (const T*)s->A, s->npad, (T*)s->scratch);
  }));
  TGP_HIP_TRY(hipMemcpyAsync(L_host, s->scratch, size_t(s->n) * s->n * es, hipMemcpyDeviceToHost,
                             ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TGP_OK;
}

int tgp_solver_device_factor(tgp_solver* s, void** L_dev, int64_t* n_pad) {
  SOLVER_GUARD(s);
  if (L_dev) *L_dev = s->A;
  if (n_pad) *n_pad = s->npad;
  return TGP_OK;
}

int tgp_solver_timings(tgp_solver* s, double* ms, int n) {
  SOLVER_GUARD(s);
  TGP_ARG_CHECK(ms != nullptr && n >= 0, "bad argument");
  for (int i = 0; i < n && i < 8; ++i) ms[i] = s->ms[i];
  return TGP_OK;
}

// the chain kernel's ticket -> task map on the host (chain_tasks.h): ticket < 0 -> *n_tasks only
int tgp_tile_order(int64_t tm, int64_t tn, int32_t lower, int32_t band, int64_t id, int32_t* ti, int32_t* tj, int64_t* n_tiles) {
  if (!(tm >= 1 && tn >= 1 && tm <= 65535 && tn <= 65535 && band >= 0 && (!lower || tn <= tm))) {
    tgp::set_error("tgp_tile_order: bad shape");
    return TGP_E_ARG;
  }
  const int64_t n = tile_count((int)tm, (int)tn, lower);
  if (n_tiles) *n_tiles = n;
  if (id >= 0) {
    if (id >= n || !ti || !tj) {
      tgp::set_error("tgp_tile_order: id out of range");
      return TGP_E_ARG;
    }
    int a = 0, b = 0;
    tile_decode((int)id, (int)tm, (int)tn, lower, band, a, b);
    *ti = a;
    *tj = b;
  }
  return TGP_OK;
}

int tgp_chain_task(int64_t R, int64_t nblk, int64_t cb, int64_t ce, int64_t ticket, int32_t* out5, int64_t* n_tasks) {
  if (!(R >= 1 && cb >= 0 && cb < ce && ce <= nblk && nblk <= 64 && nblk <= R && R <= (1 << 20))) {
    tgp::set_error("tgp_chain_task: bad panel shape");
    return TGP_E_ARG;
  }
  const int64_t n = chain_task_count((int)R, (int)nblk, (int)cb, (int)ce);
  if (n_tasks) *n_tasks = n;
  if (ticket >= 0) {
    if (ticket >= n || out5 == nullptr) {
      tgp::set_error("tgp_chain_task: ticket out of range");
      return TGP_E_ARG;
    }
    const ChainTask t = chain_decode_ticket((int)ticket, (int)R, (int)nblk, (int)cb, (int)ce);
    out5[0] = t.kind; out5[1] = t.i; out5[2] = t.c; out5[3] = t.k; out5[4] = t.part;
  }
  return TGP_OK;
}

int tgp_chain_tasks(int64_t R, int64_t nblk, int64_t cb, int64_t ce, int64_t batch, int64_t lag, int64_t rowlag, int64_t minrows,
                    int64_t fwd, int32_t* out6, int64_t cap_tasks, int64_t* n_tasks) {
  if (!(R >= 1 && cb >= 0 && cb < ce && ce <= nblk && nblk <= 64 && nblk <= R && R <= tgp::CHAIN_MAX_ROW_TILES) ||
      !(batch >= 0 && batch <= 32 && lag >= 1 && rowlag >= 2 && minrows >= 0) || n_tasks == nullptr || cap_tasks < 0) {
    tgp::set_error("tgp_chain_tasks: bad panel shape or policy");
    return TGP_E_ARG;
  }
  const std::vector<ChainTask> list = chain_build((int)R, (int)nblk, (int)cb, (int)ce, ChainPolicy{(int)batch, (int)lag, (int)rowlag, (int)minrows}, fwd != 0 ? 1 : 0);
  *n_tasks = (int64_t)list.size();
  for (int64_t u = 0; out6 != nullptr && u < (int64_t)list.size() && u < cap_tasks; ++u) {
    const ChainTask t = chain_unpack(chain_pack(list[size_t(u)]));  // through the table's packing: what the kernel sees
    int32_t* o = out6 + 6 * u;
    o[0] = t.kind; o[1] = t.i; o[2] = t.c; o[3] = t.k; o[4] = t.part; o[5] = t.k0;
  }
  return TGP_OK;
}

int tgp_chain_stamps(tgp_ctx* ctx, int64_t* out, int64_t cap_tasks, int64_t* n_tasks) {
  CTX_GUARD(ctx);
  TGP_ARG_CHECK(out != nullptr && n_tasks != nullptr && cap_tasks >= 0, "chain_stamps: bad argument");
  const int64_t n = std::min<int64_t>(ctx->d_chain_stamps ? ctx->chain_stamp_base : 0, cap_tasks);
  *n_tasks = n;
  if (n > 0) {
    for (hipStream_t q : {ctx->stream, ctx->panel_stream, ctx->update_stream})
      if (q) TGP_HIP_TRY(hipStreamSynchronize(q));
    TGP_HIP_TRY(hipMemcpy(out, ctx->d_chain_stamps, size_t(n) * 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
  }
  return TGP_OK;
}

// Dry run of tgp_solver_factor / tgp_solver_factor_logprob: the sequence of kernel launches and
// event operations the five streams would receive for an n_pad x n_pad problem, without a GPU
// (no HIP call is made).  Ten int64 per record: kind, stream, v[0..7] (tgp_trace_rec).
int tgp_trace_factor(int64_t n_pad, const char* options, int32_t fused, int64_t* out, int64_t cap_records,
                     int64_t* n_records) {
  using namespace tgp;
  TGP_ARG_CHECK(n_pad > 0 && n_pad % 128 == 0 && out != nullptr && n_records != nullptr,
                "trace: n_pad must be a positive multiple of 128");
  tgp_ctx ctx;
  auto fake = [](uintptr_t v) { return reinterpret_cast<void*>(v); };
  ctx.stream = (hipStream_t)fake(0x10);
  ctx.panel_stream = (hipStream_t)fake(0x20);
  ctx.solve_stream = (hipStream_t)fake(0x30);
  ctx.update_stream = (hipStream_t)fake(0x40);
  ctx.asm_stream = (hipStream_t)fake(0x50);
  ctx.ev_a = (hipEvent_t)fake(0x100);
  ctx.ev_b = (hipEvent_t)fake(0x110);
  ctx.ev_c = (hipEvent_t)fake(0x120);
  ctx.ev_d = (hipEvent_t)fake(0x130);
  ctx.ev_e = (hipEvent_t)fake(0x140);
  ctx.ev_asm = (hipEvent_t)fake(0x150);
  ctx.ev_f = (hipEvent_t)fake(0x160);
  ctx.ev_g1 = (hipEvent_t)fake(0x170);
  ctx.ev_g2 = (hipEvent_t)fake(0x180);
  ctx.ev_h = (hipEvent_t)fake(0x190);
  ctx.ev_i = (hipEvent_t)fake(0x1a0);
  ctx.ev_j = (hipEvent_t)fake(0x1b0);
  TGP_TRY(apply_options(&ctx, options));  // every option of tgp_ctx_set_option, library defaults otherwise
  std::vector<tgp_trace_rec> recs;
  ctx.trace = &recs;
  // never dereferenced: only differences of these pointers are recorded
  double* base = static_cast<double*>(fake(uintptr_t(1) << 40));
  double* dinv = static_cast<double*>(fake(uintptr_t(2) << 40));
  double* y = static_cast<double*>(fake(uintptr_t(3) << 40));
  double* X = static_cast<double*>(fake(uintptr_t(4) << 40));
  ctx.trace_base = base;
  KProg kp{};
  if (fused & 1) trace_push(&ctx, 8, ctx.stream);  // residual -> work vector (main stream, in front of the assembly)
  TGP_TRY(assemble_lower<double>(&ctx, kp, n_pad, 1, X, X, base, n_pad));
  int32_t info = 0;
  const int st = potrf<double>(&ctx, n_pad, base, n_pad, dinv, &info, (fused & 1) ? y : nullptr);
  if (st < 0) return st;
  trace_push(&ctx, 9, ctx.stream);  // reductions over diag(L) and the solved vector (main stream)
  *n_records = (int64_t)recs.size();
  TGP_ARG_CHECK((int64_t)recs.size() <= cap_records, "trace: %lld records, room for %lld",
                (long long)recs.size(), (long long)cap_records);
  for (size_t i = 0; i < recs.size(); ++i) {
    out[10 * i] = recs[i].kind;
    out[10 * i + 1] = recs[i].stream;
    for (int q = 0; q < 8; ++q) out[10 * i + 2 + q] = recs[i].v[q];
  }
  return TGP_OK;
}

}  // extern "C"
