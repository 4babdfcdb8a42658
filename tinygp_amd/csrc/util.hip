// util.hip -- O(N) / O(N^2) helpers around the factor: reductions (K7/K8), L @ y (K11),
// row-major <-> column-major copies for the host-pointer solver layer.
#include "tgp_common.h"

namespace tgp {

namespace {

// deterministic single-workgroup reduction: fixed per-thread strides + fixed LDS tree
template <typename T, typename F>
__device__ __forceinline__ void block_reduce_store(F term, int64_t n, double* out) {
  __shared__ double red[1024];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += term(i);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

template <typename T>
__global__ __launch_bounds__(1024) void sum_log_diag_kernel(int64_t n, const T* __restrict__ L,
                                                            int64_t ld, double* out) {
  block_reduce_store<T>([&](int64_t i) { return log(double(L[i * ld + i])); }, n, out);
}

template <typename T>
__global__ __launch_bounds__(1024) void sum_squares_kernel(int64_t n, const T* __restrict__ y,
                                                           double* out) {
  block_reduce_store<T>([&](int64_t i) { const double v = double(y[i]); return v * v; }, n, out);
}

// out[i] = base[i] - sum_j B[i, j]^2 (B column-major m x n): thread per row, coalesced
template <typename T>
__global__ __launch_bounds__(256) void row_sumsq_kernel(int64_t m, int64_t n,
                                                        const T* __restrict__ B, int64_t ldb,
                                                        const T* __restrict__ base,
                                                        T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= m) return;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int64_t j = 0;
  for (; j + 4 <= n; j += 4) {
    const T v0 = B[j * ldb + i], v1 = B[(j + 1) * ldb + i];
    const T v2 = B[(j + 2) * ldb + i], v3 = B[(j + 3) * ldb + i];
    a0 += v0 * v0; a1 += v1 * v1; a2 += v2 * v2; a3 += v3 * v3;
  }
  for (; j < n; ++j) { const T v = B[j * ldb + i]; a0 += v * v; }
  out[i] = base[i] - ((a0 + a1) + (a2 + a3));
}

// out = L y, L lower column-major: thread per row, y staged through LDS
template <typename T>
__global__ __launch_bounds__(256) void trmv_lower_kernel(int64_t n, const T* __restrict__ L,
                                                         int64_t ld, const T* __restrict__ y,
                                                         T* __restrict__ out) {
  __shared__ T sy[256];
  const int64_t i0 = int64_t(blockIdx.x) * 256, i = i0 + threadIdx.x;
  T acc = 0;
  const int64_t jend = (i0 + 256 < n) ? i0 + 256 : n;
  for (int64_t jb = 0; jb < jend; jb += 256) {
    __syncthreads();
    sy[threadIdx.x] = (jb + threadIdx.x < n) ? y[jb + threadIdx.x] : T(0);
    __syncthreads();
    if (i < n) {
      const int64_t cnt = (jend - jb < 256) ? (jend - jb) : 256;
      for (int64_t jj = 0; jj < cnt; ++jj) {
        const int64_t j = jb + jj;
        if (j <= i) acc += L[j * ld + i] * sy[jj];
      }
    }
  }
  if (i < n) out[i] = acc;
}

// out (n x n, ROW-major, ld n) = lower(L) with the upper triangle zero; 32x32 LDS transpose
template <typename T>
__global__ __launch_bounds__(256) void extract_lower_rowmajor_kernel(int64_t n,
                                                                     const T* __restrict__ L,
                                                                     int64_t ld,
                                                                     T* __restrict__ out) {
  __shared__ T tile[32][33];
  const int64_t bi = int64_t(blockIdx.x) * 32, bj = int64_t(blockIdx.y) * 32;  // rows bi.., cols bj..
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
  for (int q = ty; q < 32; q += 8) {  // read column-major: consecutive tx -> consecutive rows
    const int64_t i = bi + tx, j = bj + q;
    tile[q][tx] = (i < n && j < n && j <= i) ? L[j * ld + i] : T(0);
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {  // write row-major: consecutive tx -> consecutive cols
    const int64_t i = bi + q, j = bj + tx;
    if (i < n && j < n) out[i * n + j] = tile[tx][q];
  }
}

// A (column-major npad x npad, ld) <- symmetric src (n x n, row-major == column-major),
// identity in the padding.  The whole square is written.
template <typename T>
__global__ __launch_bounds__(256) void set_from_rowmajor_kernel(int64_t n, int64_t npad,
                                                                const T* __restrict__ src,
                                                                T* __restrict__ A, int64_t ld) {
  __shared__ T tile[32][33];
  const int64_t bi = int64_t(blockIdx.x) * 32, bj = int64_t(blockIdx.y) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // A[i, j] (i >= j region) = src[i][j] = src row-major at i*n + j: transpose through LDS
  for (int q = ty; q < 32; q += 8) {
    const int64_t i = bi + q, j = bj + tx;
    tile[q][tx] = (i < n && j < n) ? src[i * n + j] : ((i == j) ? T(1) : T(0));
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    const int64_t i = bi + tx, j = bj + q;
    if (i < npad && j < npad) A[j * ld + i] = tile[tx][q];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void add_diag_kernel(int64_t n, T* __restrict__ A, int64_t ld,
                                                       const T* __restrict__ diag) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) A[i * ld + i] += diag[i];
}

}  // namespace

template <typename T>
int launch_sum_log_diag(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, int slot) {
  hipLaunchKernelGGL((sum_log_diag_kernel<T>), dim3(1), dim3(1024), 0, ctx->stream, n, L, ld,
                     ctx->d_scal + slot);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_sum_squares(tgp_ctx* ctx, int64_t n, const T* y, int slot) {
  hipLaunchKernelGGL((sum_squares_kernel<T>), dim3(1), dim3(1024), 0, ctx->stream, n, y,
                     ctx->d_scal + slot);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
// the same two reductions on a given stream into a given device slot (dist.hip)
template <typename T>
int launch_sum_log_diag_at(tgp_ctx* ctx, hipStream_t st, int64_t n, const T* L, int64_t ld, double* out) {
  (void)ctx;
  hipLaunchKernelGGL((sum_log_diag_kernel<T>), dim3(1), dim3(1024), 0, st, n, L, ld, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_sum_squares_at(tgp_ctx* ctx, hipStream_t st, int64_t n, const T* y, double* out) {
  (void)ctx;
  hipLaunchKernelGGL((sum_squares_kernel<T>), dim3(1), dim3(1024), 0, st, n, y, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_row_sumsq(tgp_ctx* ctx, int64_t m, int64_t n, const T* B, int64_t ldb, const T* base,
                     T* out) {
  if (m == 0) return TGP_OK;
  hipLaunchKernelGGL((row_sumsq_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0,
                     ctx->stream, m, n, B, ldb, base, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_trmv_lower(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, const T* y, T* out) {
  if (n == 0) return TGP_OK;
  hipLaunchKernelGGL((trmv_lower_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, n, L, ld, y, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_extract_lower_rowmajor(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* out) {
  if (n == 0) return TGP_OK;
  const unsigned t = (unsigned)((n + 31) / 32);
  TGP_ARG_CHECK(t <= 65535, "matrix too large for a host copy");
  hipLaunchKernelGGL((extract_lower_rowmajor_kernel<T>), dim3(t, t), dim3(256), 0, ctx->stream, n,
                     L, ld, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_set_lower_from_rowmajor(tgp_ctx* ctx, int64_t n, int64_t npad, const T* src, T* A,
                                   int64_t ld) {
  if (npad == 0) return TGP_OK;
  const unsigned t = (unsigned)((npad + 31) / 32);
  TGP_ARG_CHECK(t <= 65535, "matrix too large for a host copy");
  hipLaunchKernelGGL((set_from_rowmajor_kernel<T>), dim3(t, t), dim3(256), 0, ctx->stream, n, npad,
                     src, A, ld);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}
template <typename T>
int launch_add_diag(tgp_ctx* ctx, int64_t n, T* A, int64_t ld, const T* diag) {
  if (n == 0) return TGP_OK;
  hipLaunchKernelGGL((add_diag_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, n, A, ld, diag);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

#define TGP_INST(T)                                                                             \
  template int launch_sum_log_diag<T>(tgp_ctx*, int64_t, const T*, int64_t, int);               \
  template int launch_sum_squares<T>(tgp_ctx*, int64_t, const T*, int);                         \
  template int launch_sum_log_diag_at<T>(tgp_ctx*, hipStream_t, int64_t, const T*, int64_t, double*); \
  template int launch_sum_squares_at<T>(tgp_ctx*, hipStream_t, int64_t, const T*, double*);     \
  template int launch_row_sumsq<T>(tgp_ctx*, int64_t, int64_t, const T*, int64_t, const T*, T*);\
  template int launch_trmv_lower<T>(tgp_ctx*, int64_t, const T*, int64_t, const T*, T*);        \
  template int launch_extract_lower_rowmajor<T>(tgp_ctx*, int64_t, const T*, int64_t, T*);      \
  template int launch_set_lower_from_rowmajor<T>(tgp_ctx*, int64_t, int64_t, const T*, T*,      \
                                                 int64_t);                                      \
  template int launch_add_diag<T>(tgp_ctx*, int64_t, T*, int64_t, const T*);
TGP_INST(float)
TGP_INST(double)
#undef TGP_INST

}  // namespace tgp
