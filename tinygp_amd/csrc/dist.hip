// dist.hip -- one rank's share of the 1-D block-cyclic column Cholesky (SURVEY.md 8e,
// BASELINE configs 4 and 5).  The reference has no multi-device code; this is new design.
//
// Layout.  The N x N matrix is cut into block columns of width nb (a multiple of 128); block
// column j lives on rank j mod G.  A rank keeps its nloc block columns side by side in ONE
// column-major matrix A_loc (n_pad x nloc*nb, ld = n_pad, rows GLOBAL): local block column l
// is global block column l*G + rank.  Only the part on/below each column's diagonal block is
// meaningful (full-height storage costs 2x the minimum and buys plain pointer arithmetic:
// config 4 is 17 GB per GPU of 288).
//
// Step k.  The owner factors panel k in place with the single-GPU panel chain (chol.hip,
// potf2 / trsm / in-panel updates on the priority stream) and PACKS it -- [dinv of its 128-
// blocks | rows k*nb.. x nb, ld = rows] -- into ring slot k mod 3.  The host (Python,
// tinygp_amd/distributed.py) broadcasts that slot with RCCL.  Every rank then
//   * runs forward-substitution step k of the (replicated) right-hand side straight from the
//     received panel on the solve stream -- log_probability needs no further exchange,
//   * updates its block column k+1 first if it owns it (look-ahead) and starts that panel's
//     chain on the priority stream,
//   * updates the rest of its block columns with ONE MFMA launch over all of them
//     (gemm_nt's block-cyclic tile map).
// The collectives are the host's; this file only orders its streams around the two points
// where the host calls RCCL (after tgp_dist_panel / tgp_dist_after_recv, before tgp_dist_rest).
#include <cmath>

#include "tgp_common.h"

using namespace tgp;

struct tgp_dist {
  tgp_ctx* ctx = nullptr;
  int dtype = TGP_F64;
  int64_t n = 0, npad = 0, nb = 0, nblk = 0, nloc = 0;
  int d = 1, G = 1, rank = 0;
  void* X = nullptr;       // (n, d) replicated
  void* diag = nullptr;    // (n,)
  void* A = nullptr;       // n_pad x nloc*nb, ld = n_pad
  void* dinv = nullptr;    // (n_pad/128) * 2048 (only the owned panels' entries are written)
  static constexpr int NSLOT = 3;
  void* ring[3] = {nullptr, nullptr, nullptr};  // caller-owned broadcast slots (panel k in slot k mod 3)
  hipEvent_t ev_solve[3] = {nullptr, nullptr, nullptr};  // forward step that read the slot has finished
  bool ev_solve_set[3] = {false, false, false};
  hipEvent_t ev_arrived = nullptr;  // panel k is in its slot (recorded on the main stream by after_recv)
  void* x = nullptr;       // caller-owned replicated vector (n_pad): residual -> L^-1 r -> K^-1 r
  void* Xown = nullptr;    // coordinates of the owned columns, compacted (cond-mean partial)
  void* aown = nullptr;    // alpha at the owned columns, compacted
  double* d_logdet = nullptr;  // per-panel sum log L_ii (nblk) + [nblk] = sum of squares
  int64_t n_own = 0;
  bool solving = false, asm_pending = false;
  bool asm_deferred = false;  // owned block columns l >= 1 are still to be assembled (tgp_dist_first_panel)
  KProg kp{};
};

namespace {

inline size_t esz(int dtype) { return dtype == TGP_F64 ? 8 : 4; }
inline int64_t slot_dinv_elems(const tgp_dist* h) { return (h->nb / TILE) * 2048; }
inline int64_t rows_of(const tgp_dist* h, int64_t k) { return h->npad - k * h->nb; }
inline int owner_of(const tgp_dist* h, int64_t k) { return int(k % h->G); }

// panel (rows x nb at src, leading dimension ld) -> dst (leading dimension rows); 16-byte moves
template <typename T>
__global__ __launch_bounds__(256) void pack_panel_kernel(const T* __restrict__ src, int64_t ld,
                                                         T* __restrict__ dst, int64_t rows) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int64_t c = blockIdx.y;
  const vec_t* s = reinterpret_cast<const vec_t*>(src + c * ld);
  vec_t* o = reinterpret_cast<vec_t*>(dst + c * rows);
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < rows / V; i += int64_t(gridDim.x) * 256)
    o[i] = s[i];
}

// y[c] -= sum_r P[r, c] x[r], P (m x ncols) column-major: one workgroup per column, lanes along
// the contiguous rows, fixed-order LDS tree (deterministic).  The L^T half of the distributed
// backward substitution.
template <typename T>
__global__ __launch_bounds__(256) void gemv_t_sub_kernel(int64_t m, const T* __restrict__ P, int64_t ld,
                                                         const T* __restrict__ x, T* __restrict__ y) {
  __shared__ T red[256];
  const T* col = P + int64_t(blockIdx.x) * ld;
  T a0 = 0, a1 = 0;
  int64_t r = threadIdx.x;
  for (; r + 256 < m; r += 512) {
    a0 += col[r] * x[r];
    a1 += col[r + 256] * x[r + 256];
  }
  if (r < m) a0 += col[r] * x[r];
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) y[blockIdx.x] -= red[0];
}

template <typename T>
int join_assembly(tgp_dist* h) {
  tgp_ctx* ctx = h->ctx;
  if (h->asm_pending) {
    TGP_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_asm, 0));
    h->asm_pending = false;
  }
  return TGP_OK;
}

// chain of the owned panel k on the priority stream + pack into its ring slot.  `head_done`:
// the first 128-block's potf2 is already on the main stream (behind the look-ahead update).
template <typename T>
int factor_and_pack(tgp_dist* h, int64_t k, bool head_done) {
  tgp_ctx* ctx = h->ctx;
  hipStream_t S0 = ctx->stream, S1 = ctx->panel_stream;
  const int64_t l = k / h->G, rows = rows_of(h, k), ld = h->npad;
  T* Ap = (T*)h->A + l * h->nb * ld + k * h->nb;
  T* dk = (T*)h->dinv + (k * h->nb / TILE) * 2048;
  // everything the main stream has queued so far (look-ahead update of this column, the reads of
  // the slot's previous panel) precedes the chain and the pack
  TGP_HIP_TRY(hipEventRecord(ctx->ev_a, S0));
  TGP_HIP_TRY(hipStreamWaitEvent(S1, ctx->ev_a, 0));
  const std::function<int(hipEvent_t)> no_mid = [](hipEvent_t) { return TGP_OK; };
  TGP_TRY(panel_chain<T>(ctx, S1, rows, Ap, ld, dk, k * h->nb, 0, h->nb, head_done, (T*)nullptr, 0, no_mid));
  T* slot = (T*)h->ring[k % tgp_dist::NSLOT];
  const int64_t nd = slot_dinv_elems(h);
  TGP_HIP_TRY(hipMemcpyAsync(slot, dk, size_t(nd) * sizeof(T), hipMemcpyDeviceToDevice, S1));
  unsigned gx = (unsigned)((rows / (16 / sizeof(T)) + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL((pack_panel_kernel<T>), dim3(gx, (unsigned)h->nb), dim3(256), 0, S1, Ap, ld,
                     slot + nd, rows);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// K(X, X) + noise for the owned block columns (lower part of each), identity padding.  Only the
// first owned column is assembled here (it gates this rank's first panel or first update); the
// others follow on the same main stream in tgp_dist_first_panel, BEHIND the point where the first
// panel's chain branches off to the priority stream -- hidden beside that chain without a stream of
// their own (a rank with peers should have no more than four streams in use, the collective
// library's included: profiles/r02_m_stream_count.txt).
template <typename T>
int assemble_columns(tgp_dist* h, int64_t l_begin, int64_t l_end) {
  tgp_ctx* ctx = h->ctx;
  const int flags = KMAT_LOWER | KMAT_PAD_IDENTITY;
  for (int64_t l = l_begin; l < l_end; ++l) {
    const int64_t j0 = (l * h->G + h->rank) * h->nb;
    const int64_t n1 = std::max<int64_t>(h->n - j0, 0), n2 = std::max<int64_t>(std::min(h->nb, h->n - j0), 0);
    const int64_t x0 = std::min(j0, h->n);
    TGP_TRY(launch_kmat_cols<T>(ctx, ctx->stream, h->kp, n1, n2, h->d, (const T*)h->X + x0 * h->d,
                                (const T*)h->X + x0 * h->d, (const T*)h->diag + x0,
                                (T*)h->A + l * h->nb * h->npad + j0, h->npad, h->npad - j0, h->nb,
                                flags, 0, h->nb / TILE));
  }
  return TGP_OK;
}

}  // namespace

#define DIST_GUARD(h)                                                      \
  TGP_ARG_CHECK((h) != nullptr && (h)->ctx != nullptr, "null dist handle"); \
  std::unique_lock<std::recursive_mutex> _tgp_lock((h)->ctx->mu);          \
  TGP_HIP_TRY(hipSetDevice((h)->ctx->device))

template <typename F>
static int ddispatch(int dtype, F&& f) {
  if (dtype == TGP_F64) return f(double{});
  return f(float{});
}

extern "C" {

int64_t tgp_dist_slot_elems(int64_t n, int64_t nb) {
  if (n <= 0 || nb <= 0 || nb % TILE) return -1;
  const int64_t npad = round_up(n, nb);
  return (nb / TILE) * 2048 + npad * nb;
}

int tgp_dist_create(tgp_ctx* ctx, int dtype, int64_t n, int32_t d, const void* X_host,
                    const void* noise_diag_host, int64_t nb, int32_t world, int32_t rank,
                    void* ring0_dev, void* ring1_dev, void* ring2_dev, void* x_dev, tgp_dist** out) {
  TGP_ARG_CHECK(ctx != nullptr && out != nullptr, "null argument");
  std::unique_lock<std::recursive_mutex> lk(ctx->mu);
  TGP_HIP_TRY(hipSetDevice(ctx->device));
  TGP_ARG_CHECK(dtype == TGP_F32 || dtype == TGP_F64, "dtype must be TGP_F32 or TGP_F64");
  TGP_ARG_CHECK(n >= 1 && d >= 1 && d <= TGP_MAX_DIM, "bad problem size");
  TGP_ARG_CHECK(nb >= TILE && nb % TILE == 0, "nb must be a positive multiple of %d", TILE);
  TGP_ARG_CHECK(world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
  TGP_ARG_CHECK(X_host && noise_diag_host && ring0_dev && ring1_dev && ring2_dev && x_dev, "null buffer");
  TGP_ARG_CHECK(ctx->panel_stream != nullptr, "the context has no panel stream");
  tgp_dist* h = new tgp_dist();
  h->ctx = ctx; h->dtype = dtype; h->n = n; h->d = d; h->nb = nb; h->G = world; h->rank = rank;
  h->nblk = (n + nb - 1) / nb;
  h->npad = h->nblk * nb;
  h->nloc = (h->nblk - rank + world - 1) / world;
  if (h->nloc < 0) h->nloc = 0;
  h->ring[0] = ring0_dev; h->ring[1] = ring1_dev; h->ring[2] = ring2_dev; h->x = x_dev;
  const size_t es = esz(dtype);
  auto fail = [&](int code) { tgp_dist_destroy(h); return code; };
#define D_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) {                    \
    set_error("%s failed: %s", #expr, hipGetErrorString(_e));                               \
    return fail(_e == hipErrorOutOfMemory ? TGP_E_NOMEM : TGP_E_HIP); } } while (0)
  D_TRY(hipMalloc(&h->X, size_t(n) * d * es));
  D_TRY(hipMalloc(&h->diag, size_t(n) * es));
  D_TRY(hipMalloc(&h->A, std::max<size_t>(size_t(h->npad) * size_t(h->nloc * nb) * es, 8)));
  D_TRY(hipMalloc(&h->dinv, size_t(h->npad / TILE) * 2048 * es));
  D_TRY(hipMalloc((void**)&h->d_logdet, size_t(h->nblk + 1) * sizeof(double)));
  for (auto& e : h->ev_solve) D_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  D_TRY(hipEventCreateWithFlags(&h->ev_arrived, hipEventDisableTiming));
  D_TRY(hipMemcpyAsync(h->X, X_host, size_t(n) * d * es, hipMemcpyHostToDevice, ctx->stream));
  D_TRY(hipMemcpyAsync(h->diag, noise_diag_host, size_t(n) * es, hipMemcpyHostToDevice, ctx->stream));
  // compacted coordinates of the owned columns (for the conditional-mean partial products)
  std::vector<char> xo;
  for (int64_t l = 0; l < h->nloc; ++l) {
    const int64_t j0 = (l * world + rank) * nb;
    const int64_t cnt = std::min<int64_t>(nb, n - j0);
    if (cnt <= 0) break;
    const char* src = (const char*)X_host + size_t(j0) * d * es;
    xo.insert(xo.end(), src, src + size_t(cnt) * d * es);
    h->n_own += cnt;
  }
  D_TRY(hipMalloc(&h->Xown, std::max<size_t>(xo.size(), 8)));
  D_TRY(hipMalloc(&h->aown, std::max<size_t>(size_t(h->n_own) * es, 8)));
  if (!xo.empty()) D_TRY(hipMemcpyAsync(h->Xown, xo.data(), xo.size(), hipMemcpyHostToDevice, ctx->stream));
  D_TRY(hipStreamSynchronize(ctx->stream));
#undef D_TRY
  *out = h;
  return TGP_OK;
}

int tgp_dist_destroy(tgp_dist* h) {
  if (!h) return TGP_OK;
  if (h->ctx) {
    hipSetDevice(h->ctx->device);
    for (hipStream_t q : {h->ctx->asm_stream, h->ctx->panel_stream, h->ctx->update_stream,
                          h->ctx->solve_stream, h->ctx->stream})
      if (q) hipStreamSynchronize(q);
  }
  void* bufs[] = {h->X, h->diag, h->A, h->dinv, h->Xown, h->aown, (void*)h->d_logdet};
  for (void* b : bufs)
    if (b) hipFree(b);
  if (h->ev_arrived) hipEventDestroy(h->ev_arrived);
  for (auto e : h->ev_solve)
    if (e) hipEventDestroy(e);
  delete h;
  return TGP_OK;
}

int tgp_dist_stream(tgp_dist* h, int which, void** stream_out) {
  TGP_ARG_CHECK(h != nullptr && stream_out != nullptr, "null argument");
  TGP_ARG_CHECK(which == 0 || which == 1, "stream index must be 0 (main) or 1 (panel)");
  *stream_out = which == 0 ? (void*)h->ctx->stream : (void*)h->ctx->panel_stream;
  return TGP_OK;
}

int tgp_dist_assemble(tgp_dist* h, const tgp_kop* prog, int nops) {
  DIST_GUARD(h);
  TGP_TRY(make_kprog(prog, nops, &h->kp));
  h->asm_pending = false;
  h->asm_deferred = h->nloc > 1;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    return assemble_columns<T>(h, 0, std::min<int64_t>(h->nloc, 1));
  });
}

// Start of a factorisation.  resid_host != NULL: the right-hand side (n,) is uploaded into the
// replicated vector and forward-substituted panel by panel as the panels arrive.
int tgp_dist_begin(tgp_dist* h, const void* resid_host) {
  DIST_GUARD(h);
  tgp_ctx* ctx = h->ctx;
  const size_t es = esz(h->dtype);
  TGP_HIP_TRY(hipMemsetAsync(ctx->d_info, 0, sizeof(int32_t), ctx->stream));
  TGP_HIP_TRY(hipMemsetAsync(h->d_logdet, 0, size_t(h->nblk + 1) * sizeof(double), ctx->stream));
  h->solving = resid_host != nullptr;
  if (h->solving) {
    TGP_HIP_TRY(hipMemcpyAsync(h->x, resid_host, size_t(h->n) * es, hipMemcpyHostToDevice, ctx->stream));
    if (h->npad > h->n)
      TGP_HIP_TRY(hipMemsetAsync((char*)h->x + size_t(h->n) * es, 0, size_t(h->npad - h->n) * es, ctx->stream));
  }
  for (bool& b : h->ev_solve_set) b = false;
  return TGP_OK;
}

// Owner of panel 0: its chain + pack.  Every rank: the rest of its block columns is assembled now.
// (Later panels are started by tgp_dist_after_recv.)
int tgp_dist_first_panel(tgp_dist* h) {
  DIST_GUARD(h);
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    if (owner_of(h, 0) == h->rank) TGP_TRY(factor_and_pack<T>(h, 0, false));
    if (h->asm_deferred) {
      h->asm_deferred = false;
      TGP_TRY(assemble_columns<T>(h, 1, h->nloc));
    }
    return TGP_OK;
  });
}

// Panel k sits in its ring slot on this rank and the MAIN stream has been made to wait for
// its arrival by the caller (RCCL work.wait()).  If this rank owns panel k+1: the look-ahead
// update of that block column, its first potf2, its chain (priority stream) and its pack.
// The host then starts the broadcast of panel k+1 and calls tgp_dist_fwd_step(k).
int tgp_dist_after_recv(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  tgp_ctx* ctx = h->ctx;
  hipStream_t S0 = ctx->stream;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb, nd = slot_dinv_elems(h);
    const int sl = int(k % tgp_dist::NSLOT), sl_next = int((k + 1) % tgp_dist::NSLOT);
    const T* slot = (const T*)h->ring[sl];
    const T* P = slot + nd;  // rows x nb, ld = rows, row 0 = global row k*nb
    TGP_HIP_TRY(hipEventRecord(h->ev_arrived, S0));
    // The slot panel k+1 will be written into (by the owner's pack or by RCCL, both ordered
    // behind the main stream from here on) was last read by the forward step of panel k-2:
    // two panels of slack for the forward solve, which shares the chip with the updates.
    if (h->ev_solve_set[sl_next]) TGP_HIP_TRY(hipStreamWaitEvent(S0, h->ev_solve[sl_next], 0));
    const int64_t k1 = k + 1;
    if (k1 < h->nblk && owner_of(h, k1) == h->rank) {
      TGP_TRY(join_assembly<T>(h));
      const int64_t l1 = k1 / h->G, m = rows_of(h, k1), ld = h->npad;
      T* C = (T*)h->A + l1 * nb * ld + k1 * nb;
      const int64_t tiles = (m / TILE) * (nb / TILE) - (nb / TILE) * (nb / TILE - 1) / 2;
      const int role = tiles <= ctx->first_small_tiles ? 4 : 0;
      TGP_TRY(launch_gemm_nt<T>(ctx, S0, m, nb, nb, P + nb, rows, P + nb, rows, C, ld, 1, 0, role));
      // the panel's first potf2 goes in front of the big update on the main stream: issued
      // beside it, it waits a whole round of tiles for a free CU
      TGP_TRY(panel_potf2<T>(ctx, S0, C, ld, (T*)h->dinv + (k1 * nb / TILE) * 2048, k1 * nb, 0, false));
      TGP_TRY(factor_and_pack<T>(h, k1, true));
    }
    return TGP_OK;
  });
}

// Forward-substitution step k of the replicated right-hand side (and sum log L_ii of panel k),
// straight from the received panel, on the solve stream.  A separate entry point so that the host
// can start the broadcast of panel k+1 first.
int tgp_dist_fwd_step(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  tgp_ctx* ctx = h->ctx;
  // stream: the update stream (behind the in-panel updates of the chain that was queued just before;
  // the steps have two panels of slack), so that a rank uses three streams of its own
  if (ctx->dist_solve_aux == 0) TGP_TRY(ensure_solve_stream(ctx));
  hipStream_t S2 = ctx->dist_solve_aux != 0 ? ctx->update_stream : ctx->solve_stream;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb, nd = slot_dinv_elems(h);
    const int sl = int(k % tgp_dist::NSLOT);
    const T* slot = (const T*)h->ring[sl];
    const T* P = slot + nd;
    TGP_HIP_TRY(hipStreamWaitEvent(S2, h->ev_arrived, 0));
    if (h->solving) {
      T* xk = (T*)h->x + k * nb;
      for (int64_t j = 0; j < nb; j += TILE)
        TGP_TRY(launch_trsv_fwd_step<T>(ctx, S2, rows - (j + TILE), P + j * rows + j, rows,
                                        slot + (j / TILE) * 2048, xk + j));
    }
    TGP_TRY(launch_sum_log_diag_at<T>(ctx, S2, nb, P, rows, h->d_logdet + k));
    TGP_HIP_TRY(hipEventRecord(h->ev_solve[sl], S2));
    h->ev_solve_set[sl] = true;
    return TGP_OK;
  });
}

// The rest of step k: every owned block column right of k (and of k+1, done by the look-ahead)
// in one MFMA launch.
int tgp_dist_rest(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb;
    const T* P = (const T*)h->ring[k % tgp_dist::NSLOT] + slot_dinv_elems(h);
    // first owned block column j = l*G + rank with j > k, skipping k+1 (look-ahead)
    int64_t l0 = (k + 1 - h->rank + h->G - 1) / h->G;
    if (l0 < 0) l0 = 0;
    if (l0 * h->G + h->rank == k + 1) ++l0;
    const int64_t cnt = h->nloc - l0;
    if (cnt <= 0) return TGP_OK;
    TGP_TRY(join_assembly<T>(h));
    return launch_gemm_nt_dist<T>(ctx, ctx->stream, h->npad, nb, nb, P - k * nb, rows, (T*)h->A, h->npad,
                                  h->G, h->rank, l0, cnt);
  });
}

// End of the factorisation (every panel received and applied): joins the streams and returns
// this rank's potrf info (owners see their own pivots only: the host takes the MIN over ranks),
// |L^-1 r|^2 (0 when no right-hand side was given) and sum log L_ii -- both computed
// redundantly from the received panels, so identical on every rank.
int tgp_dist_end(tgp_dist* h, int32_t* info, double* sumsq, double* logdet_half) {
  DIST_GUARD(h);
  tgp_ctx* ctx = h->ctx;
  hipStream_t S0 = ctx->stream;
  for (int i = 0; i < tgp_dist::NSLOT; ++i)
    if (h->ev_solve_set[i]) TGP_HIP_TRY(hipStreamWaitEvent(S0, h->ev_solve[i], 0));
  TGP_TRY(ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    if (h->solving) TGP_TRY(launch_sum_squares_at<T>(ctx, S0, h->npad, (const T*)h->x, h->d_logdet + h->nblk));
    return TGP_OK;
  }));
  std::vector<double> part(size_t(h->nblk + 1), 0.0);
  int32_t inf = 0;
  TGP_HIP_TRY(hipMemcpyAsync(part.data(), h->d_logdet, part.size() * sizeof(double), hipMemcpyDeviceToHost, S0));
  TGP_HIP_TRY(hipMemcpyAsync(&inf, ctx->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, S0));
  TGP_HIP_TRY(hipStreamSynchronize(S0));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->panel_stream));
  if (inf == INT32_MIN) {  // panel_step_kernel's bounded wait ran out (chol.hip)
    set_error("block-column driver: a panel step's hand-off flag never arrived (device-side timeout)");
    return TGP_E_HIP;
  }
  double ld = 0;
  for (int64_t k = 0; k < h->nblk; ++k) ld += part[size_t(k)];  // fixed order
  if (info) *info = inf;
  if (sumsq) *sumsq = part[size_t(h->nblk)];
  if (logdet_half) *logdet_half = ld;
  return TGP_OK;
}

// Backward substitution L^T x = z, block k (the owner only; others return at once):
//   x_k <- L_kk^-T (z_k - L[rows below, block k]^T x[rows below])
// in place in the replicated vector; the host then broadcasts x_k (n b entries) from the owner.
int tgp_dist_bwd_step(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  if (owner_of(h, k) != h->rank) return TGP_OK;
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t l = k / h->G, nb = h->nb, ld = h->npad, below = h->npad - (k + 1) * nb;
    const T* col = (const T*)h->A + l * nb * ld;
    T* xk = (T*)h->x + k * nb;
    if (below > 0) {
      hipLaunchKernelGGL((gemv_t_sub_kernel<T>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, below,
                         col + (k + 1) * nb, ld, (const T*)h->x + (k + 1) * nb, xk);
      TGP_HIP_TRY(hipGetLastError());
    }
    return trsv<T>(ctx, nb, col + k * nb, ld, (const T*)h->dinv + (k * nb / TILE) * 2048, 1, xk);
  });
}

// This rank's share of the conditional mean  K(X*, X) alpha  (reference gp.py:353-359,
// kernels/base.py:68-82): sum over the OWNED columns only, K never formed (fused kmat_gemv).
// out_dev (m,) is a device buffer of the caller (summed over ranks with one all-reduce).
int tgp_dist_cond_mean_partial(tgp_dist* h, const tgp_kop* prog, int nops, int64_t m,
                               const void* Xt_host, void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(m >= 1 && Xt_host && out_dev, "cond_mean_partial: bad argument");
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  tgp_ctx* ctx = h->ctx;
  const size_t es = esz(h->dtype);
  if (h->n_own == 0) {
    TGP_HIP_TRY(hipMemsetAsync(out_dev, 0, size_t(m) * es, ctx->stream));
    return TGP_OK;
  }
  void* xt = nullptr;
  TGP_HIP_TRY(hipMalloc(&xt, size_t(m) * h->d * es));
  int st = TGP_OK;
  do {
    if (hipMemcpyAsync(xt, Xt_host, size_t(m) * h->d * es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
      st = TGP_E_HIP;
      break;
    }
    int64_t off = 0;
    for (int64_t l = 0; l < h->nloc && st == TGP_OK; ++l) {
      const int64_t j0 = (l * h->G + h->rank) * h->nb;
      const int64_t cnt = std::min<int64_t>(h->nb, h->n - j0);
      if (cnt <= 0) break;
      if (hipMemcpyAsync((char*)h->aown + size_t(off) * es, (const char*)h->x + size_t(j0) * es,
                         size_t(cnt) * es, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
        st = TGP_E_HIP;
      off += cnt;
    }
    if (st != TGP_OK) break;
    st = ddispatch(h->dtype, [&](auto tag) {
      using T = decltype(tag);
      return launch_kmat_gemv<T>(ctx, kp, m, h->n_own, h->d, (const T*)xt, (const T*)h->Xown,
                                 (const T*)h->aown, (T*)out_dev);
    });
  } while (0);
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(xt);
  if (st == TGP_E_HIP) set_error("HIP error in tgp_dist_cond_mean_partial");
  return st;
}

// test / inspection hook: copy local block column l (rows from its diagonal block down,
// column-major, ld = rows) to the host
int tgp_dist_get_column(tgp_dist* h, int64_t l, void* out_host) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(l >= 0 && l < h->nloc && out_host, "bad local block column");
  const size_t es = esz(h->dtype);
  const int64_t j0 = (l * h->G + h->rank) * h->nb, rows = h->npad - j0;
  TGP_HIP_TRY(hipMemcpy2DAsync(out_host, size_t(rows) * es,
                               (const char*)h->A + (size_t(l) * h->nb * h->npad + size_t(j0)) * es,
                               size_t(h->npad) * es, size_t(rows) * es, size_t(h->nb),
                               hipMemcpyDeviceToHost, h->ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(h->ctx->stream));
  return TGP_OK;
}

}  // extern "C"
