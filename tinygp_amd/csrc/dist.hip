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
// Step k (round 3: the chain pipeline lives on the priority stream, two panels ahead of the big updates).
//   * The owner of panel k+1 -- as soon as panel k has ARRIVED, whatever the main stream is still doing -- applies
//     panel k to block column k+1 (the gate), factors it with the single-GPU panel chain and PACKS it into ring
//     slot (k+1) mod 3 = [rows (k+1) nb.. x nb, ld = rows | dinv of its 128-blocks], in column CHUNKS: a chunk is
//     contiguous in the column-major panel, and it is final (and its broadcast can start) while the chain is still
//     factoring the columns to its right.  All of that on the priority stream.
//   * The host (tinygp_amd/distributed.py) broadcasts each chunk with RCCL.
//   * Every rank, on the main stream: forward-substitution step k of the (replicated) right-hand side straight
//     from the received panel (update stream) -- log_probability needs no further exchange --, then panel k
//     applied to block column k+2 on ITS owner first (so that the gate of the next step never waits for the big
//     update), then to all other owned block columns with ONE MFMA launch (gemm_nt's block-cyclic tile map),
//     which leaves workgroup slots free when a chain of this rank runs beside it.
// A slot is rewritten (pack or RCCL) only behind the last readers of the panel it held three steps ago.
// The collectives are the host's; this file only orders its streams around them.
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
  hipEvent_t ev_arrived = nullptr;  // panel k is in its slot (recorded on the main stream by tgp_dist_arrived)
  hipEvent_t ev_rest[3] = {nullptr, nullptr, nullptr};  // main-stream readers of the slot (pre-update, rest) have finished
  bool ev_rest_set[3] = {false, false, false};
  hipEvent_t ev_pre[2] = {nullptr, nullptr};  // block column j (parity j & 1) carries every panel < j - 1 ... see pre_update
  bool ev_pre_set[2] = {false, false};
  hipEvent_t ev_asm_done = nullptr;  // every owned block column is assembled (main stream)
  bool s1_saw_asm = false;           // the priority stream has waited for it in this factorisation
  void* x = nullptr;       // caller-owned replicated vector (n_pad): residual -> L^-1 r -> K^-1 r
  void* Xown = nullptr;    // coordinates of the owned columns, compacted (cond-mean partial)
  void* aown = nullptr;    // alpha at the owned columns, compacted
  double* d_logdet = nullptr;  // per-panel sum log L_ii (nblk) + [nblk] = sum of squares
  void* bwd_ws = nullptr;      // multi-RHS backward solve: transposed blocks of a block row | M = P L_kk^T P | its 16 x 16
  size_t bwd_ws_bytes = 0;     //   inverses | the reversed right-hand sides (grown on demand)
  double* d_grad = nullptr;    // gradient accumulators (2 * TGP_KPROG_MAX + TGP_MAX_DIM doubles) of tgp_dist_grad_*
  void* kinv_diag = nullptr;   // diag(K^-1) collected chunk by chunk (n_pad entries)
  int64_t n_own = 0;
  bool solving = false, asm_pending = false;
  bool asm_deferred = false;  // owned block columns l >= 1 are still to be assembled (tgp_dist_first_panel)
  KProg kp{};
};

namespace {

inline size_t esz(int dtype) { return dtype == TGP_F64 ? 8 : 4; }
inline int64_t slot_dinv_elems(const tgp_dist* h) { return (h->nb / TILE) * 2048; }
// slot of panel k: [rows x nb panel, ld = rows | dinv]: the panel first, so that a column chunk -- and the last
// chunk together with the inverses -- is one contiguous message
template <typename T>
inline const T* slot_dinv(const tgp_dist* h, int64_t k) {
  return (const T*)h->ring[k % tgp_dist::NSLOT] + (h->npad - k * h->nb) * h->nb;
}
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

// x = y + a (a: the NEGATIVE sum of the updates that reached this block: the fan-in forward solve)
template <typename T>
__global__ __launch_bounds__(256) void add_into_kernel(int64_t cnt, const T* __restrict__ y, const T* __restrict__ a,
                                                       T* __restrict__ x) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < cnt) x[i] = y[i] + a[i];
}

// partial[c][r] = sum over the rows of chunk c (256 rows) of x[i, r]^2 -- x (rows, R) ROW-major; the second pass adds
// the chunks in a fixed order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void colsumsq_chunk_kernel(int64_t rows, int64_t R, const T* __restrict__ x,
                                                             double* __restrict__ partial) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t i0 = int64_t(blockIdx.y) * 256, i1 = i0 + 256 < rows ? i0 + 256 : rows;
  if (r >= R) return;
  double s = 0;
  for (int64_t i = i0; i < i1; ++i) {
    const double v = double(x[i * R + r]);
    s += v * v;
  }
  partial[int64_t(blockIdx.y) * R + r] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void negate_kernel(int64_t cnt, T* __restrict__ x) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < cnt) x[i] = -x[i];
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_final_kernel(int64_t chunks, int64_t R, const double* __restrict__ partial,
                                                           T* __restrict__ out, int accumulate) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r >= R) return;
  double s = accumulate ? double(out[r]) : 0.0;
  for (int64_t c = 0; c < chunks; ++c) s += partial[c * R + r];
  out[r] = T(s);
}

// dst (cols x rows, leading dimension ldd) = src (rows x cols, leading dimension lds)^T, both column-major: 32 x 32
// tiles through LDS (padded: conflict-free), coalesced on both sides
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int64_t rows, int64_t cols, const T* __restrict__ src, int64_t lds,
                                                        T* __restrict__ dst, int64_t ldd) {
  __shared__ T tile[32][33];
  const int64_t r0 = int64_t(blockIdx.x) * 32, c0 = int64_t(blockIdx.y) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int q = ty; q < 32; q += 8)
    if (r0 + tx < rows && c0 + q < cols) tile[q][tx] = src[(c0 + q) * lds + r0 + tx];
  __syncthreads();
  for (int q = ty; q < 32; q += 8)
    if (c0 + tx < cols && r0 + q < rows) dst[(r0 + q) * ldd + c0 + tx] = tile[tx][q];
}

// M = P L^T P for a lower-triangular nb x nb block L (P: the reversal permutation): M[i][j] = L[nb-1-j][nb-1-i] is
// LOWER triangular again, and B L^-1 = ((B P) M^-T) P -- the right solve WITHOUT transposition on the kernels that
// exist for the one with it (trsm_right_lt).  Entries above the diagonal are written as zeros.
template <typename T>
__global__ __launch_bounds__(256) void reverse_transpose_kernel(int64_t nb, const T* __restrict__ L, int64_t ld,
                                                                T* __restrict__ M) {
  __shared__ T tile[32][33];
  const int64_t i0 = int64_t(blockIdx.x) * 32, j0 = int64_t(blockIdx.y) * 32;  // tile of M: rows i0.., columns j0..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // M[i][j] = L[nb-1-j][nb-1-i]: read L rows (nb-1-j0-31 .. nb-1-j0) x columns (nb-1-i0-31 .. nb-1-i0), rows contiguous
  const int64_t lr0 = nb - 32 - j0, lc0 = nb - 32 - i0;
  for (int q = ty; q < 32; q += 8) tile[q][tx] = L[(lc0 + q) * ld + lr0 + tx];  // tile[c][r] = L[lr0 + r][lc0 + c]
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    const int64_t i = i0 + tx, j = j0 + q;  // M[i][j] = L[nb-1-j][nb-1-i] = tile[nb-1-i - lc0][nb-1-j - lr0]
    M[j * nb + i] = (i >= j) ? tile[31 - tx][31 - q] : T(0);
  }
}

// dst[:, j] = src[:, nb - 1 - j] for an (R x nb) column-major pair with leading dimension R (R contiguous entries per column)
template <typename T>
__global__ __launch_bounds__(256) void reverse_cols_kernel(int64_t R, int64_t nb, const T* __restrict__ src,
                                                           T* __restrict__ dst) {
  const int64_t j = blockIdx.y;
  for (int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x; r < R; r += int64_t(gridDim.x) * 256)
    dst[j * R + r] = src[(nb - 1 - j) * R + r];
}

// (n_pad, R) row-major right-hand sides = the identity's columns c0 .. c0 + R - 1 (rows >= n_pad never exist)
template <typename T>
__global__ __launch_bounds__(256) void identity_cols_kernel(int64_t npad, int64_t R, int64_t c0, T* __restrict__ out) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= npad * R) return;
  const int64_t i = idx / R, r = idx - i * R;
  out[idx] = (i == c0 + r) ? T(1) : T(0);
}

// local block column (rows from its diagonal block down, ld = npad): ones on the diagonal of the padding columns c >= n2
template <typename T>
__global__ __launch_bounds__(256) void pad_identity_kernel(T* __restrict__ Ab, int64_t npad, int64_t n2, int64_t nb) {
  const int64_t c = n2 + int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (c < nb) Ab[c * npad + c] = T(1);
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

// The priority stream must not overwrite slot k mod 3 before the readers of the panel it held (k - 3) are done:
// that panel's forward step (update stream) and its main-stream updates.  (Its gate, if this rank ran one, was an
// earlier kernel of the priority stream itself.)
inline int wait_slot_free(tgp_dist* h, int64_t k) {
  hipStream_t S1 = h->ctx->panel_stream;
  const int sl = int(k % tgp_dist::NSLOT);
  if (h->ev_solve_set[sl]) TGP_HIP_TRY(hipStreamWaitEvent(S1, h->ev_solve[sl], 0));
  if (h->ev_rest_set[sl]) TGP_HIP_TRY(hipStreamWaitEvent(S1, h->ev_rest[sl], 0));
  return TGP_OK;
}

// Column chunk c of nch of the owned panel k on the priority stream: its blocks of the chain, then the pack of
// its columns (the last chunk also packs the inverses).  Chunk 0 of panel 0 branches off the main stream behind
// the assembly of block column 0; every other chain follows its gate on the priority stream.
template <typename T>
int chain_chunk_and_pack(tgp_dist* h, int64_t k, int64_t c, int64_t nch) {
  tgp_ctx* ctx = h->ctx;
  hipStream_t S0 = ctx->stream, S1 = ctx->panel_stream;
  const int64_t l = k / h->G, rows = rows_of(h, k), ld = h->npad, nblk_p = h->nb / TILE;
  TGP_ARG_CHECK(nch >= 1 && nch <= nblk_p && nblk_p % nch == 0 && c >= 0 && c < nch, "bad panel chunk");
  T* Ap = (T*)h->A + l * h->nb * ld + k * h->nb;
  T* dk = (T*)h->dinv + (k * h->nb / TILE) * 2048;
  const int64_t bpc = nblk_p / nch, cw = bpc * TILE;
  if (c == 0) {
    if (k == 0) {  // (later panels: the gate that precedes this call is already on the priority stream)
      TGP_HIP_TRY(hipEventRecord(ctx->ev_a, S0));
      TGP_HIP_TRY(hipStreamWaitEvent(S1, ctx->ev_a, 0));
    }
    TGP_TRY(wait_slot_free(h, k));
  }
  const std::function<int(hipEvent_t)> no_mid = [](hipEvent_t) { return TGP_OK; };
  // an unchunked panel takes panel_chain's defaults (blk_begin = 0, blk_end = -1): only those reach the
  // one-launch-per-panel / one-launch-per-block forms of the chain (round-3 advisor finding)
  TGP_TRY(panel_chain<T>(ctx, S1, rows, Ap, ld, dk, k * h->nb, 0, h->nb, false, (T*)nullptr, 0, no_mid,
                         nch == 1 ? 0 : c * bpc, nch == 1 ? -1 : (c + 1) * bpc));
  T* slot = (T*)h->ring[k % tgp_dist::NSLOT];
  unsigned gx = (unsigned)((rows / (16 / sizeof(T)) + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL((pack_panel_kernel<T>), dim3(gx, (unsigned)cw), dim3(256), 0, S1, Ap + c * cw * ld, ld,
                     slot + c * cw * rows, rows);
  TGP_HIP_TRY(hipGetLastError());
  if (c == nch - 1)
    TGP_HIP_TRY(hipMemcpyAsync(slot + rows * h->nb, dk, size_t(slot_dinv_elems(h)) * sizeof(T), hipMemcpyDeviceToDevice, S1));
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
  D_TRY(hipEventCreateWithFlags(&h->ev_asm_done, hipEventDisableTiming));
  for (auto& e : h->ev_rest) D_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : h->ev_pre) D_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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
  void* bufs[] = {h->X, h->diag, h->A, h->dinv, h->Xown, h->aown, (void*)h->d_logdet, h->bwd_ws, (void*)h->d_grad,
                  h->kinv_diag};
  for (void* b : bufs)
    if (b) hipFree(b);
  if (h->ev_arrived) hipEventDestroy(h->ev_arrived);
  if (h->ev_asm_done) hipEventDestroy(h->ev_asm_done);
  for (auto e : h->ev_rest)
    if (e) hipEventDestroy(e);
  for (auto e : h->ev_pre)
    if (e) hipEventDestroy(e);
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

// INSTEAD of tgp_dist_assemble (round 6, VERDICT r5 "missing" 2 -- the seam's `covariance=` argument and non-diagonal noise,
// reference solvers/direct.py:36,50-52): this rank's block columns of a matrix the HOST holds, (n, n) row-major, symmetric
// (trusted like the reference trusts it), noise included.  Column j of the lower triangle from row j0 down is row j of the
// host matrix from column j0 on -- contiguous -- so a block column is ONE strided copy; padding = identity.
int tgp_dist_load_matrix(tgp_dist* h, const void* K_host) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(K_host != nullptr, "load_matrix: null matrix");
  tgp_ctx* ctx = h->ctx;
  h->asm_pending = false;
  h->asm_deferred = false;
  h->kp.n = 0;  // (no kernel program behind this matrix: the gradient refuses)
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const size_t es = sizeof(T);
    for (int64_t l = 0; l < h->nloc; ++l) {
      const int64_t j0 = (l * h->G + h->rank) * h->nb;
      const int64_t n1 = std::max<int64_t>(h->n - j0, 0), n2 = std::max<int64_t>(std::min(h->nb, h->n - j0), 0);
      T* Ab = (T*)h->A + l * h->nb * h->npad + j0;
      TGP_HIP_TRY(hipMemset2DAsync(Ab, size_t(h->npad) * es, 0, size_t(h->npad - j0) * es, size_t(h->nb), ctx->stream));
      if (n1 > 0 && n2 > 0)
        TGP_HIP_TRY(hipMemcpy2DAsync(Ab, size_t(h->npad) * es, (const char*)K_host + (size_t(j0) * h->n + size_t(j0)) * es,
                                     size_t(h->n) * es, size_t(n1) * es, size_t(n2), hipMemcpyHostToDevice, ctx->stream));
      if (n2 < h->nb)
        hipLaunchKernelGGL((pad_identity_kernel<T>), dim3((unsigned)((h->nb - n2 + 255) / 256)), dim3(256), 0, ctx->stream, Ab,
                           h->npad, n2, h->nb);
    }
    TGP_HIP_TRY(hipGetLastError());
    TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));  // (pageable source: it may be released when this returns)
    return TGP_OK;
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
  for (bool& b : h->ev_rest_set) b = false;
  for (bool& b : h->ev_pre_set) b = false;
  h->s1_saw_asm = false;
  return TGP_OK;
}

// Every rank: the rest of its block columns is assembled now -- on the main stream, BEHIND the point where the
// first panel's chain branched off (tgp_dist_panel_chunk(0, 0, ..)), so that it hides beside that chain.
int tgp_dist_first_panel(tgp_dist* h) {
  DIST_GUARD(h);
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    if (h->asm_deferred) {
      h->asm_deferred = false;
      TGP_TRY(assemble_columns<T>(h, 1, h->nloc));
    }
    TGP_HIP_TRY(hipEventRecord(h->ev_asm_done, h->ctx->stream));
    return TGP_OK;
  });
}

// Owner of panel k: column chunk c of nch (nch divides nb / 128) of its chain, and the pack of that chunk into
// the ring slot; the host broadcasts the chunk right behind this call.  Panel 0: the first call branches off the
// main stream; k >= 1: call tgp_dist_lookahead(k - 1) first.
int tgp_dist_panel_chunk(tgp_dist* h, int64_t k, int64_t c, int64_t nch) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && owner_of(h, k) == h->rank, "panel_chunk: not the owner of panel %lld",
                (long long)k);
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    return chain_chunk_and_pack<T>(h, k, c, nch);
  });
}

// A rank that RECEIVES panel k: the priority stream (under which the host issues the collective) waits until the
// slot's previous panel has been read.
int tgp_dist_slot_ready(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  return wait_slot_free(h, k);
}

// Owner of panel k+1, on the PRIORITY stream, which the host has made wait for the arrival of panel k: the gate
// -- panel k applied to block column k+1 -- behind the pre-update that brought that block column up to panel k-1
// (tgp_dist_pre_update(k-1) on the main stream).  tgp_dist_panel_chunk(k+1, ..) follows.
int tgp_dist_lookahead(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  const int64_t k1 = k + 1;
  TGP_ARG_CHECK(k >= 0 && k1 < h->nblk && owner_of(h, k1) == h->rank, "lookahead: not the owner of panel %lld",
                (long long)k1);
  tgp_ctx* ctx = h->ctx;
  hipStream_t S1 = ctx->panel_stream;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb;
    const T* P = (const T*)h->ring[k % tgp_dist::NSLOT];  // rows x nb, ld = rows, row 0 = global row k*nb
    if (!h->s1_saw_asm) {
      TGP_HIP_TRY(hipStreamWaitEvent(S1, h->ev_asm_done, 0));
      h->s1_saw_asm = true;
    }
    if (h->ev_pre_set[k1 & 1]) TGP_HIP_TRY(hipStreamWaitEvent(S1, h->ev_pre[k1 & 1], 0));
    const int64_t l1 = k1 / h->G, m = rows_of(h, k1), ld = h->npad;
    T* C = (T*)h->A + l1 * nb * ld + k1 * nb;
    const int64_t tiles = (m / TILE) * (nb / TILE) - (nb / TILE) * (nb / TILE - 1) / 2;
    const int role = tiles <= ctx->first_small_tiles ? 4 : 1;
    return launch_gemm_nt<T>(ctx, S1, m, nb, nb, P + nb, rows, P + nb, rows, C, ld, 1, 0, role);
  });
}

// Panel k sits in its ring slot on this rank and the MAIN stream has been made to wait for its arrival by the
// caller (RCCL work.wait()): the marker the forward step waits for.
int tgp_dist_arrived(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  TGP_HIP_TRY(hipEventRecord(h->ev_arrived, h->ctx->stream));
  return TGP_OK;
}

// Forward-substitution step k of the replicated right-hand side (and sum log L_ii of panel k),
// straight from the received panel, on the update stream (behind the in-panel updates of a chain of this rank;
// the steps have two panels of slack), so that a rank uses three streams of its own.
int tgp_dist_fwd_step(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  tgp_ctx* ctx = h->ctx;
  if (ctx->dist_solve_aux == 0) TGP_TRY(ensure_solve_stream(ctx));
  hipStream_t S2 = ctx->dist_solve_aux != 0 ? ctx->update_stream : ctx->solve_stream;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb;
    const int sl = int(k % tgp_dist::NSLOT);
    const T* P = (const T*)h->ring[sl];
    const T* dv = slot_dinv<T>(h, k);
    TGP_HIP_TRY(hipStreamWaitEvent(S2, h->ev_arrived, 0));
    if (h->solving) {
      T* xk = (T*)h->x + k * nb;
      for (int64_t j = 0; j < nb; j += TILE)
        TGP_TRY(launch_trsv_fwd_step<T>(ctx, S2, rows - (j + TILE), P + j * rows + j, rows,
                                        dv + (j / TILE) * 2048, xk + j));
    }
    TGP_TRY(launch_sum_log_diag_at<T>(ctx, S2, nb, P, rows, h->d_logdet + k));
    TGP_HIP_TRY(hipEventRecord(h->ev_solve[sl], S2));
    h->ev_solve_set[sl] = true;
    return TGP_OK;
  });
}

// Panel k applied to block column k+2 on its owner, FIRST on the main stream: the gate of step k+1 (priority
// stream) waits for this launch only, not for the big update behind it -- the chain pipeline runs two panels
// ahead of the updates.  No-op on the other ranks.
int tgp_dist_pre_update(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  const int64_t k2 = k + 2;
  if (k2 >= h->nblk || owner_of(h, k2) != h->rank) return TGP_OK;
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb;
    const T* P = (const T*)h->ring[k % tgp_dist::NSLOT];
    TGP_TRY(launch_gemm_nt_dist<T>(ctx, ctx->stream, h->npad, nb, nb, P - k * nb, rows, (T*)h->A, h->npad,
                                   h->G, h->rank, k2 / h->G, 1, ctx->chain_reserve));
    TGP_HIP_TRY(hipEventRecord(h->ev_pre[k2 & 1], ctx->stream));
    h->ev_pre_set[k2 & 1] = true;
    return TGP_OK;
  });
}

// The rest of step k: every owned block column right of k+2 (k+1: the gate on the priority stream; k+2: the
// pre-update) in one MFMA launch, which leaves workgroup slots to a chain of this rank that runs beside it.
int tgp_dist_rest(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t rows = rows_of(h, k), nb = h->nb;
    const T* P = (const T*)h->ring[k % tgp_dist::NSLOT];
    const int sl = int(k % tgp_dist::NSLOT);
    // first owned block column j = l*G + rank with j > k+2
    int64_t l0 = (k + 3 - h->rank + h->G - 1) / h->G;
    if (l0 < 0) l0 = 0;
    const int64_t cnt = h->nloc - l0;
    if (cnt > 0) {
      const bool chain_beside = (k + 1 < h->nblk && owner_of(h, k + 1) == h->rank) ||
                                (k + 2 < h->nblk && owner_of(h, k + 2) == h->rank);
      TGP_TRY(launch_gemm_nt_dist<T>(ctx, ctx->stream, h->npad, nb, nb, P - k * nb, rows, (T*)h->A, h->npad,
                                     h->G, h->rank, l0, cnt, chain_beside ? ctx->chain_reserve : 0));
    }
    TGP_HIP_TRY(hipEventRecord(h->ev_rest[sl], ctx->stream));  // the main stream is done with this slot
    h->ev_rest_set[sl] = true;
    return TGP_OK;
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
static int bwd_block_impl(tgp_dist* h, int64_t k, void* x_dev);
int tgp_dist_bwd_step(tgp_dist* h, int64_t k) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk, "panel index out of range");
  return bwd_block_impl(h, k, h->x);
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

// Backward substitution block k on ANY device vector (n_pad entries); tgp_dist_bwd_step is this on the handle's own
// replicated vector.
static int bwd_block_impl(tgp_dist* h, int64_t k, void* x_dev) {
  if (owner_of(h, k) != h->rank) return TGP_OK;
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t l = k / h->G, nb = h->nb, ld = h->npad, below = h->npad - (k + 1) * nb;
    const T* col = (const T*)h->A + l * nb * ld;
    T* xk = (T*)x_dev + k * nb;
    if (below > 0) {
      hipLaunchKernelGGL((gemv_t_sub_kernel<T>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, below,
                         col + (k + 1) * nb, ld, (const T*)x_dev + (k + 1) * nb, xk);
      TGP_HIP_TRY(hipGetLastError());
    }
    return trsv<T>(ctx, nb, col + k * nb, ld, (const T*)h->dinv + (k * nb / TILE) * 2048, 1, xk);
  });
}

int tgp_dist_bwd_block(tgp_dist* h, int64_t k, void* x_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && x_dev != nullptr, "bwd_block: bad argument");
  return bwd_block_impl(h, k, x_dev);
}

// Forward substitution on the RESIDENT factor, block k, fan-in form (reference solvers/direct.py:66-70 for a new
// right-hand side -- north_star's "reduce-scatter of the solve RHS"): the owner of block column k holds the whole
// column, so it alone turns x_k into updates of every row below; what a block row has collected from the columns a
// rank owns sits in that rank's accumulator, and the caller REDUCES block k of the accumulators to the owner of k
// (one nb x nrhs message per block) right before this call.  Owner only (the others return at once):
//   x[block k] = L_kk^-1 (y[block k] + acc[block k]);     acc[rows below] -= L[rows below, k] x[block k]
// (acc holds the NEGATIVE sums).  nrhs == 1: vectors of n_pad entries; nrhs a multiple of 128: (n_pad, nrhs)
// ROW-major buffers -- block k of all right-hand sides is one contiguous chunk, and the chunk read column-major is
// X_k^T (nrhs x nb), exactly the operand the MFMA kernels want: the solve is trsm_right_lt, the update one gemm_nt.
int tgp_dist_fwd_block(tgp_dist* h, int64_t k, int64_t nrhs, const void* y_dev, void* acc_dev, void* x_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && y_dev && acc_dev && x_dev, "fwd_block: bad argument");
  TGP_ARG_CHECK(nrhs == 1 || (nrhs > 0 && nrhs % TILE == 0), "fwd_block: nrhs must be 1 or a multiple of %d", TILE);
  if (owner_of(h, k) != h->rank) return TGP_OK;
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t l = k / h->G, nb = h->nb, ld = h->npad, below = h->npad - (k + 1) * nb;
    const T* Lkk = (const T*)h->A + l * nb * ld + k * nb;
    const T* dk = (const T*)h->dinv + (k * nb / TILE) * 2048;
    const int64_t cnt = nb * nrhs, off = k * nb * nrhs;
    T* xk = (T*)x_dev + off;
    hipLaunchKernelGGL((add_into_kernel<T>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, cnt,
                       (const T*)y_dev + off, (const T*)acc_dev + off, xk);
    TGP_HIP_TRY(hipGetLastError());
    if (nrhs == 1) {
      TGP_TRY(trsv<T>(ctx, nb, Lkk, ld, dk, 0, xk));
      if (below > 0) TGP_TRY(gemv_sub<T>(ctx, below, nb, Lkk + nb, ld, xk, (T*)acc_dev + (k + 1) * nb));
      return TGP_OK;
    }
    TGP_TRY(trsm_right_lt<T>(ctx, nrhs, nb, Lkk, ld, dk, xk, nrhs));
    if (below > 0)
      TGP_TRY(launch_gemm_nt<T>(ctx, ctx->stream, nrhs, below, nb, xk, nrhs, Lkk + nb, ld,
                                (T*)acc_dev + (k + 1) * nb * nrhs, nrhs, 0, 0, 1));
    return TGP_OK;
  });
}

// ---- backward substitution with MANY right-hand sides on the resident factor (round 5; reference solvers/direct.py:66-68:
// solve_triangular(L, y, lower=True, trans=1) with y (N, R)) --------------------------------------------------------------
// Right-looking ("fan-out"): block k from the last to the first --
//   owner(k):  X_k = L_kk^-T Y_k                                    (tgp_dist_bwd_block_multi; Y_k carries every update)
//   the caller broadcasts X_k (ONE nb x R message per block column)
//   everyone:  Y_i -= L[k, i]^T X_k  for its OWN block columns i < k  (tgp_dist_bwd_update_multi: ONE product -- the rank's
//              right-hand-side blocks live side by side in `yloc`, (nloc nb, R) row-major, local column l at rows l nb ..;
//              the left-looking forward solve's xloc IS that layout, tgp_dist_gather_owned makes it from a global buffer)
// -- a rank only ever touches the blocks of L it owns (block row k of its own columns) and the right-hand-side blocks it
// owns: no reduction, and every product has the short dimension (nb) as its k-range, so it fills the chip.  (The
// row-oriented form of the single right-hand side -- x_k from a dot product over all rows below -- would be R x nb
// outputs with a k-range of up to N: a handful of workgroups.)
// Buffers as in tgp_dist_fwd_block: (n_pad, R) ROW-major, R a multiple of 128 -- read column-major that is X^T (R x
// n_pad, leading dimension R), the MFMA kernels' operand layout.  Both products need the TRANSPOSE of a block of L as
// gemm_nt's second operand; blocks are transposed into a scratch buffer first (nb^2 entries each: noise beside the product).
static int bwd_scratch(tgp_dist* h, size_t bytes) {
  if (h->bwd_ws_bytes >= bytes) return TGP_OK;
  if (h->bwd_ws) {
    TGP_HIP_TRY(hipStreamSynchronize(h->ctx->stream));
    TGP_HIP_TRY(hipFree(h->bwd_ws));
    h->bwd_ws = nullptr;
    h->bwd_ws_bytes = 0;
  }
  TGP_HIP_TRY(hipMalloc(&h->bwd_ws, bytes));
  h->bwd_ws_bytes = bytes;
  return TGP_OK;
}

int tgp_dist_bwd_block_multi(tgp_dist* h, int64_t k, int64_t nrhs, void* x_dev, const void* yloc_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && x_dev != nullptr && yloc_dev != nullptr && nrhs > 0 && nrhs % TILE == 0,
                "bwd_block_multi: bad argument (nrhs must be a multiple of %d)", TILE);
  if (owner_of(h, k) != h->rank) return TGP_OK;
  {  // Y_k with every update applied lives in yloc (this rank's blocks side by side): into its place in x, solved there
    const size_t es_ = esz(h->dtype), bytes = size_t(h->nb) * nrhs * es_;
    const char* src = (const char*)yloc_dev + size_t(k / h->G) * bytes;
    char* dst = (char*)x_dev + size_t(k) * bytes;
    if (src != dst) TGP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->ctx->stream));
  }
  tgp_ctx* ctx = h->ctx;
  const int64_t nb = h->nb, nd = (nb / TILE) * 2048;
  const size_t es = esz(h->dtype);
  // scratch: [M nb x nb | dinv(M) | reversed right-hand sides nrhs x nb]
  TGP_TRY(bwd_scratch(h, (size_t(nb) * nb + size_t(nd) + size_t(nrhs) * nb) * es));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t l = k / h->G, ld = h->npad;
    const T* Lkk = (const T*)h->A + l * nb * ld + k * nb;
    T* M = (T*)h->bwd_ws;
    T* dM = M + nb * nb;
    T* Br = dM + nd;
    T* Xk = (T*)x_dev + k * nb * nrhs;
    hipLaunchKernelGGL((reverse_transpose_kernel<T>), dim3((unsigned)(nb / 32), (unsigned)(nb / 32)), dim3(256), 0,
                       ctx->stream, nb, Lkk, ld, M);
    TGP_HIP_TRY(hipGetLastError());
    TGP_TRY(compute_dinv<T>(ctx, nb, M, nb, dM));
    unsigned gx = (unsigned)((nrhs + 255) / 256);
    hipLaunchKernelGGL((reverse_cols_kernel<T>), dim3(gx, (unsigned)nb), dim3(256), 0, ctx->stream, nrhs, nb, (const T*)Xk, Br);
    TGP_HIP_TRY(hipGetLastError());
    TGP_TRY(trsm_right_lt<T>(ctx, nrhs, nb, M, nb, dM, Br, nrhs));  // Br <- (Y_k^T P) M^-T = Y_k^T L_kk^-1 P
    hipLaunchKernelGGL((reverse_cols_kernel<T>), dim3(gx, (unsigned)nb), dim3(256), 0, ctx->stream, nrhs, nb, (const T*)Br, Xk);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

int tgp_dist_bwd_update_multi(tgp_dist* h, int64_t k, int64_t nrhs, const void* x_dev, void* yloc_dev, int64_t stop_block) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && x_dev != nullptr && yloc_dev != nullptr && nrhs > 0 && nrhs % TILE == 0 &&
                    stop_block >= 0, "bwd_update_multi: bad argument");
  tgp_ctx* ctx = h->ctx;
  const int64_t nb = h->nb;
  // owned block columns i in [stop_block, k): local indices l with l * G + rank in that range
  int64_t l_begin = (stop_block - h->rank + h->G - 1) / h->G;
  if (l_begin < 0) l_begin = 0;
  int64_t l_end = (k - h->rank + h->G - 1) / h->G;  // first l with l * G + rank >= k
  if (l_end > h->nloc) l_end = h->nloc;
  if (l_end <= l_begin) return TGP_OK;
  const int64_t cnt = l_end - l_begin;
  TGP_TRY(bwd_scratch(h, std::max<size_t>(h->bwd_ws_bytes, size_t(cnt) * nb * nb * esz(h->dtype))));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t ld = h->npad;
    const T* Xk = (const T*)x_dev + k * nb * nrhs;
    T* Tt = (T*)h->bwd_ws;
    // block row k of the owned columns l_begin .. l_end - 1 (side by side in A_loc) -> its transpose (cnt*nb x nb, leading
    // dimension cnt*nb): gemm_nt's second operand B[c][m] = L[k nb + m, column c of those] at m * (cnt*nb) + c
    const T* src = (const T*)h->A + l_begin * nb * ld + k * nb;
    hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)(nb / 32), (unsigned)(cnt * nb / 32)), dim3(256), 0, ctx->stream,
                       nb, cnt * nb, src, ld, Tt, cnt * nb);
    TGP_HIP_TRY(hipGetLastError());
    // ONE product for all of them: the rank's right-hand-side blocks are CONTIGUOUS in yloc (local column l at rows l nb ..),
    //   Yloc^T[:, l_begin nb .. l_end nb) (nrhs x cnt*nb) -= X_k^T (nrhs x nb) . L[k, those columns]
    // (with the blocks where the global layout has them this was cnt small launches of nrhs/128 x nb/128 tiles each, one
    // k-loop deep: N = 65 536, 64 right-hand sides: 2 080 launches, 336 ms -- profiles/r05_f)
    return launch_gemm_nt<T>(ctx, ctx->stream, nrhs, cnt * nb, nb, Xk, nrhs, Tt, cnt * nb,
                             (T*)yloc_dev + l_begin * nb * nrhs, nrhs, 0, 0, 1);
  });
}

// yloc (nloc nb, nrhs) <- this rank's blocks of x (n_pad, nrhs), side by side (world size 1: the same layout -- pass x itself)
int tgp_dist_gather_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* yloc_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(x_dev && yloc_dev && nrhs > 0, "gather_owned: bad argument");
  const size_t bytes = size_t(h->nb) * nrhs * esz(h->dtype);
  for (int64_t l = 0; l < h->nloc; ++l) {
    const char* src = (const char*)x_dev + size_t(l * h->G + h->rank) * bytes;
    char* dst = (char*)yloc_dev + size_t(l) * bytes;
    if (src != dst) TGP_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->ctx->stream));
  }
  return TGP_OK;
}

// (n_pad, nrhs) row-major device buffer <- columns c0 .. c0 + nrhs - 1 of the identity (the right-hand sides behind
// K^-1[:, c0 : c0 + nrhs] = L^-T L^-1 E: the gradient on the block-column path)
int tgp_dist_identity_cols(tgp_dist* h, int64_t c0, int64_t nrhs, void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(c0 >= 0 && nrhs > 0 && out_dev != nullptr, "identity_cols: bad argument");
  const int64_t cnt = h->npad * nrhs;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((identity_cols_kernel<T>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->ctx->stream, h->npad,
                       nrhs, c0, (T*)out_dev);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

// ---- gradient of log_probability on the block-column path (SURVEY 8f-1 x 8e; what jax.value_and_grad of reference
// gp.py:126-138 gives a caller at any size) ------------------------------------------------------------------------------
//   d ll / d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta),  alpha = K^-1 r replicated in the handle's vector x.
// K^-1 is never held as a whole: the caller solves for it a CHUNK of columns at a time on the resident factor (fan-in
// forward from the chunk's first block, fan-out backward down to it: tinygp_amd/distributed.py), the chunk ends up
// replicated as (n_pad, R) row-major, and every rank adds the lower-triangle terms of ITS block rows to its accumulators:
//   tgp_dist_grad_begin(prog)                       zero the accumulators
//   tgp_dist_grad_chunk(c0, nrhs, Kcols)            += this rank's share for columns c0 .. c0 + nrhs - 1; diag(K^-1) kept
//   tgp_dist_grad_end(out, grad_logscale, kinv_diag) this rank's partial sums (the caller all-reduces 2 nops + d doubles)
// grad layout as tgp_solver_grad: out[2 i + q] = parameter q of op i; grad_logscale[q] per input dimension (or NULL).
int tgp_dist_grad_begin(tgp_dist* h, const tgp_kop* prog, int nops) {
  DIST_GUARD(h);
  TGP_TRY(make_kprog(prog, nops, &h->kp));
  const size_t cnt = size_t(2 * TGP_KPROG_MAX + TGP_MAX_DIM);
  if (!h->d_grad) TGP_HIP_TRY(hipMalloc((void**)&h->d_grad, cnt * sizeof(double)));
  if (!h->kinv_diag) TGP_HIP_TRY(hipMalloc(&h->kinv_diag, size_t(h->npad) * esz(h->dtype)));
  TGP_HIP_TRY(hipMemsetAsync(h->d_grad, 0, cnt * sizeof(double), h->ctx->stream));
  TGP_HIP_TRY(hipMemsetAsync(h->kinv_diag, 0, size_t(h->npad) * esz(h->dtype), h->ctx->stream));
  return TGP_OK;
}

int tgp_dist_grad_chunk(tgp_dist* h, int64_t c0, int64_t nrhs, const void* kcols_dev, int32_t with_logscale) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(h->d_grad != nullptr, "grad_chunk: call tgp_dist_grad_begin first");
  TGP_ARG_CHECK(c0 >= 0 && c0 % TILE == 0 && nrhs > 0 && nrhs % TILE == 0 && kcols_dev != nullptr, "grad_chunk: bad argument");
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const T* alpha = (const T*)h->x;
    const T* Kc = (const T*)kcols_dev;
    for (int i = 0; i < h->kp.n; ++i) {
      const int op = h->kp.op[i];
      if (op >= TGP_K_ADD) continue;
      const int nparam = (op == TGP_K_ESS || op == TGP_K_RQ) ? 2 : 1;
      for (int q = 0; q < nparam; ++q)
        TGP_TRY(launch_kgrad_cols<T>(ctx, h->kp, i, q, h->n, h->d, (const T*)h->X, alpha, Kc, nrhs, c0, h->nb, h->G,
                                     h->rank, h->d_grad + 2 * i + q));
    }
    for (int q = 0; q < (with_logscale ? h->d : 0); ++q)
      TGP_TRY(launch_kgrad_cols<T>(ctx, h->kp, -1 - q, 0, h->n, h->d, (const T*)h->X, alpha, Kc, nrhs, c0, h->nb, h->G,
                                   h->rank, h->d_grad + 2 * TGP_KPROG_MAX + q));
    return launch_kcols_diag<T>(ctx, h->n, Kc, nrhs, c0, (T*)h->kinv_diag);
  });
}

int tgp_dist_grad_end(tgp_dist* h, double* grad_params, double* grad_logscale, void* kinv_diag_host) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(h->d_grad != nullptr && grad_params != nullptr, "grad_end: bad argument");
  tgp_ctx* ctx = h->ctx;
  std::vector<double> g(size_t(2 * TGP_KPROG_MAX + TGP_MAX_DIM), 0.0);
  TGP_HIP_TRY(hipMemcpyAsync(g.data(), h->d_grad, g.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (kinv_diag_host)
    TGP_HIP_TRY(hipMemcpyAsync(kinv_diag_host, h->kinv_diag, size_t(h->n) * esz(h->dtype), hipMemcpyDeviceToHost, ctx->stream));
  TGP_HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 2 * h->kp.n; ++i) grad_params[i] = g[size_t(i)];
  if (grad_logscale)
    for (int q = 0; q < h->d; ++q) grad_logscale[q] = g[size_t(2 * TGP_KPROG_MAX + q)];
  return TGP_OK;
}

// ---- forward substitution, LEFT-looking fan-in (round 5): the form that SCALES over the ranks ---------------------------
// tgp_dist_fwd_block above is right-looking: the owner of block column k alone turns x_k into updates of every row below --
// (N - k nb) x R x nb flops on ONE rank per step while the others wait for the next reduce.  Fine at world size 1 (big
// products, full chip), serial across ranks.  Left-looking, step k:
//   every rank:  acc_k -= X[its columns in [first, k)] . L[k, those columns]^T    (tgp_dist_fwd_partial: ONE product per rank,
//                k-range = its share of the columns left of k -- balanced to within one block over the ranks)
//   the caller reduces acc_k to owner(k)   (the same ONE nb x R message per block column)
//   owner(k):    x_k = L_kk^-1 (y_k + acc_k), kept twice: in x (global rows) and in xloc, the rank's OWN solved blocks side
//                by side (local column l at rows l nb ..) -- the contiguous operand of the next steps' products
// Only owner(k)'s next product depends on x_k; every other rank's does not: the ranks run ahead of the reduce chain.
// The product has a small output (R x nb) and a long k-range: fp64 uses the split-k tail of gemm_nt (partial tiles combined
// in slice order: deterministic).  Buffers: y, acc, x (n_pad, R) row-major, xloc (nloc nb, R) row-major.
int tgp_dist_fwd_partial(tgp_dist* h, int64_t k, int64_t nrhs, const void* xloc_dev, void* acc_dev, int64_t first_block) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && xloc_dev && acc_dev && nrhs > 0 && nrhs % TILE == 0 && first_block >= 0,
                "fwd_partial: bad argument");
  tgp_ctx* ctx = h->ctx;
  const int64_t nb = h->nb;
  int64_t l_begin = (first_block - h->rank + h->G - 1) / h->G;
  if (l_begin < 0) l_begin = 0;
  int64_t l_end = (k - h->rank + h->G - 1) / h->G;  // local columns with global index < k
  if (l_end > h->nloc) l_end = h->nloc;
  if (l_end <= l_begin) return TGP_OK;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t ld = h->npad;
    const T* A = (const T*)xloc_dev + l_begin * nb * nrhs;            // X^T[:, local columns l_begin ..]: (nrhs x K), ld nrhs
    const T* B = (const T*)h->A + l_begin * nb * ld + k * nb;         // L[block row k, local columns l_begin ..]: (nb x K), ld
    T* C = (T*)acc_dev + k * nb * nrhs;                               // acc_k^T (nrhs x nb), ld nrhs
    const int64_t keep = ctx->split_tail;
    ctx->split_tail = 1;  // few output tiles, long k-range: the split-k tail fills the chip (fp64)
    const int st = launch_gemm_nt<T>(ctx, ctx->stream, nrhs, nb, (l_end - l_begin) * nb, A, nrhs, B, ld, C, nrhs, 0, 0, 0);
    ctx->split_tail = keep;
    return st;
  });
}

int tgp_dist_fwd_solve_left(tgp_dist* h, int64_t k, int64_t nrhs, const void* y_dev, const void* acc_dev, void* x_dev,
                            void* xloc_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(k >= 0 && k < h->nblk && y_dev && acc_dev && x_dev && xloc_dev && nrhs > 0 && nrhs % TILE == 0,
                "fwd_solve_left: bad argument");
  if (owner_of(h, k) != h->rank) return TGP_OK;
  tgp_ctx* ctx = h->ctx;
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t l = k / h->G, nb = h->nb, ld = h->npad;
    const T* Lkk = (const T*)h->A + l * nb * ld + k * nb;
    const T* dk = (const T*)h->dinv + (k * nb / TILE) * 2048;
    const int64_t cnt = nb * nrhs, off = k * nb * nrhs;
    T* xk = (T*)x_dev + off;
    hipLaunchKernelGGL((add_into_kernel<T>), dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, cnt,
                       (const T*)y_dev + off, (const T*)acc_dev + off, xk);
    TGP_HIP_TRY(hipGetLastError());
    TGP_TRY(trsm_right_lt<T>(ctx, nrhs, nb, Lkk, ld, dk, xk, nrhs));
    TGP_HIP_TRY(hipMemcpyAsync((T*)xloc_dev + l * nb * nrhs, xk, size_t(cnt) * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    return TGP_OK;
  });
}

// out = sum over the OWNED block columns of L[:, k] y_k  (this rank's share of dot_triangular, reference
// solvers/direct.py:72-73; the caller all-reduces).  Vectors of n_pad entries on the device.
int tgp_dist_trmv_partial(tgp_dist* h, const void* y_dev, void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(y_dev && out_dev, "trmv_partial: null buffer");
  tgp_ctx* ctx = h->ctx;
  TGP_HIP_TRY(hipMemsetAsync(out_dev, 0, size_t(h->npad) * esz(h->dtype), ctx->stream));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t nb = h->nb, ld = h->npad;
    // gemv_sub accumulates out -= P x over the owned panels (the factor's diagonal blocks are stored with zeros above
    // the diagonal); ONE sign change at the end
    for (int64_t l = 0; l < h->nloc; ++l) {
      const int64_t k = l * h->G + h->rank;
      const T* col = (const T*)h->A + l * nb * ld + k * nb;
      TGP_TRY(gemv_sub<T>(ctx, h->npad - k * nb, nb, col, ld, (const T*)y_dev + k * nb, (T*)out_dev + k * nb));
    }
    hipLaunchKernelGGL((negate_kernel<T>), dim3((unsigned)((h->npad + 255) / 256)), dim3(256), 0, ctx->stream,
                       h->npad, (T*)out_dev);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

// K(X, X*) for every data point against m test points, as the right-hand sides of the fan-in forward solve:
// out_dev (n_pad, m_pad) ROW-major (= K(X*, X) column-major with leading dimension m_pad), zero padded.
// Reference solvers/direct.py:87-92 (Ks).
int tgp_dist_cross_cov(tgp_dist* h, const tgp_kop* prog, int nops, int64_t m, const void* Xt_host, int64_t m_pad,
                       void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(m >= 1 && m_pad >= m && m_pad % TILE == 0 && Xt_host && out_dev, "cross_cov: bad argument");
  KProg kp;
  TGP_TRY(make_kprog(prog, nops, &kp));
  tgp_ctx* ctx = h->ctx;
  const size_t es = esz(h->dtype);
  void* xt = nullptr;
  TGP_HIP_TRY(hipMalloc(&xt, size_t(m) * h->d * es));
  int st = TGP_OK;
  if (hipMemcpyAsync(xt, Xt_host, size_t(m) * h->d * es, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = TGP_E_HIP;
  if (st == TGP_OK)
    st = ddispatch(h->dtype, [&](auto tag) {
      using T = decltype(tag);
      // rows = test points (n1 = m), columns = data points (n2 = n), zero padding out to (m_pad, n_pad)
      return launch_kmat<T>(ctx, kp, m, h->n, h->d, (const T*)xt, (const T*)h->X, (const T*)nullptr, (T*)out_dev, m_pad,
                            m_pad, h->npad, 0);
    });
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(xt);
  if (st == TGP_E_HIP) set_error("HIP error in tgp_dist_cross_cov");
  return st;
}

// out[r] (+)= sum over the rows of the OWNED blocks of x[i, r]^2, x (n_pad, nrhs) ROW-major: this rank's share of
// colsum(A o A) behind the conditional variance (reference solvers/direct.py:94-95 without the M x M product).
int tgp_dist_colsumsq_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(nrhs >= 1 && x_dev && out_dev, "colsumsq_owned: bad argument");
  tgp_ctx* ctx = h->ctx;
  const int64_t nb = h->nb, chunks = (nb + 255) / 256;
  TGP_TRY(ensure_work(ctx, size_t(chunks) * size_t(nrhs) * sizeof(double)));
  TGP_HIP_TRY(hipMemsetAsync(out_dev, 0, size_t(nrhs) * esz(h->dtype), ctx->stream));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    for (int64_t l = 0; l < h->nloc; ++l) {  // block by block, in a fixed order
      const int64_t k = l * h->G + h->rank;
      hipLaunchKernelGGL((colsumsq_chunk_kernel<T>), dim3((unsigned)((nrhs + 255) / 256), (unsigned)chunks), dim3(256), 0,
                         ctx->stream, nb, nrhs, (const T*)x_dev + k * nb * nrhs, (double*)ctx->d_work);
      hipLaunchKernelGGL((colsum_final_kernel<T>), dim3((unsigned)((nrhs + 255) / 256)), dim3(256), 0, ctx->stream,
                         chunks, nrhs, (const double*)ctx->d_work, (T*)out_dev, 1);
    }
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

// out (nrhs x nrhs, column-major) = sum over the rows of the OWNED blocks of x_i x_i^T: this rank's share of A^T A
// (reference solvers/direct.py:95); x (n_pad, nrhs) ROW-major, nrhs a multiple of 128.
int tgp_dist_gram_owned(tgp_dist* h, int64_t nrhs, const void* x_dev, void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(nrhs > 0 && nrhs % TILE == 0 && x_dev && out_dev, "gram_owned: nrhs must be a multiple of %d", TILE);
  tgp_ctx* ctx = h->ctx;
  TGP_HIP_TRY(hipMemsetAsync(out_dev, 0, size_t(nrhs) * size_t(nrhs) * esz(h->dtype), ctx->stream));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t nb = h->nb;
    for (int64_t l = 0; l < h->nloc; ++l) {  // out -= X_k^T (nrhs x nb) (-X_k^T)^T is not available: accumulate -G, negate
      const int64_t k = l * h->G + h->rank;
      const T* xk = (const T*)x_dev + k * nb * nrhs;
      TGP_TRY(launch_gemm_nt<T>(ctx, ctx->stream, nrhs, nrhs, nb, xk, nrhs, xk, nrhs, (T*)out_dev, nrhs, 0, 0, 1));
    }
    hipLaunchKernelGGL((negate_kernel<T>), dim3((unsigned)((nrhs * nrhs + 255) / 256)), dim3(256), 0, ctx->stream,
                       nrhs * nrhs, (T*)out_dev);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

// The same for TWO sets of right-hand sides (round 6: the conditional covariance in chunks of test points, so that it is bound
// by ONE (n_pad, M) matrix of solved columns instead of three): out (nrhs_i x nrhs_j, column-major) = sum over the rows of
// the owned blocks of x_i^T x_j, this rank's share of block (i, j) of A^T A.  i == j gives tgp_dist_gram_owned's arithmetic.
int tgp_dist_gram_pair_owned(tgp_dist* h, int64_t nrhs_i, const void* xi_dev, int64_t nrhs_j, const void* xj_dev,
                             void* out_dev) {
  DIST_GUARD(h);
  TGP_ARG_CHECK(nrhs_i > 0 && nrhs_i % TILE == 0 && nrhs_j > 0 && nrhs_j % TILE == 0 && xi_dev && xj_dev && out_dev,
                "gram_pair_owned: the numbers of right-hand sides must be multiples of %d", TILE);
  tgp_ctx* ctx = h->ctx;
  TGP_HIP_TRY(hipMemsetAsync(out_dev, 0, size_t(nrhs_i) * size_t(nrhs_j) * esz(h->dtype), ctx->stream));
  return ddispatch(h->dtype, [&](auto tag) {
    using T = decltype(tag);
    const int64_t nb = h->nb;
    for (int64_t l = 0; l < h->nloc; ++l) {  // (accumulates -G as above, then negates)
      const int64_t k = l * h->G + h->rank;
      const T* xi = (const T*)xi_dev + k * nb * nrhs_i;
      const T* xj = (const T*)xj_dev + k * nb * nrhs_j;
      TGP_TRY(launch_gemm_nt<T>(ctx, ctx->stream, nrhs_i, nrhs_j, nb, xi, nrhs_i, xj, nrhs_j, (T*)out_dev, nrhs_i, 0, 0, 1));
    }
    hipLaunchKernelGGL((negate_kernel<T>), dim3((unsigned)((nrhs_i * nrhs_j + 255) / 256)), dim3(256), 0, ctx->stream,
                       nrhs_i * nrhs_j, (T*)out_dev);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  });
}

// After a rank-local failure: join every stream of the driver and forget the events of the interrupted pass, so that
// a retry starts from a quiet device (round-3 advisor finding).
int tgp_dist_abort(tgp_dist* h) {
  DIST_GUARD(h);
  tgp_ctx* ctx = h->ctx;
  for (hipStream_t q : {ctx->asm_stream, ctx->panel_stream, ctx->update_stream, ctx->solve_stream, ctx->stream})
    if (q) (void)hipStreamSynchronize(q);
  (void)hipGetLastError();
  for (bool& b : h->ev_solve_set) b = false;
  for (bool& b : h->ev_rest_set) b = false;
  for (bool& b : h->ev_pre_set) b = false;
  h->solving = false;
  h->asm_pending = h->asm_deferred = false;
  h->s1_saw_asm = false;
  ctx->chain_polls_pending = false;
  return TGP_OK;
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
