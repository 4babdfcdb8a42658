// Where does the assembly kernel's time go?  A stand-alone copy of kmat_fast_kernel's shape (ExpSquared, 1-D, lower
// tiles of an N x N column-major matrix; lane = 2 rows x 32 columns, 16-byte stores) in variants:
//   0 as shipped   1 stores only (no exp, no quotient)   2 arithmetic only (stores behind a never-true test)
//   3 as 0 on a grid of the lower tiles only (no empty workgroups)   4 as 0 with non-temporal stores
//   5 as 3 + 4
// build + run on the GPU box:  hipcc -O3 -ffp-contract=off --offload-arch=gfx950 scripts/probe_kmat.hip -o /tmp/pk && /tmp/pk
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
constexpr int KT = 128;
typedef double T2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(double p0, double amp, int64_t n, const double* __restrict__ X,
                                             const double* __restrict__ diag, double* __restrict__ out, int64_t ld) {
  constexpr int NCOL = 32;
  const int cq = threadIdx.x >> 6, l = threadIdx.x & 63;
  int tr, tc;
  if constexpr (MODE == 3 || MODE == 5) {  // linear index over the lower tiles, column by column
    const int nt = int(n / KT);
    int b = blockIdx.x, c = 0;
    // column c holds nt - c tiles
    c = int((2.0 * nt + 1 - sqrt((2.0 * nt + 1) * (2.0 * nt + 1) - 8.0 * b)) / 2);
    while (c > 0 && c * nt - c * (c - 1) / 2 > b) --c;
    while ((c + 1) * nt - (c + 1) * c / 2 <= b) ++c;
    tc = c;
    tr = c + (b - (c * nt - c * (c - 1) / 2));
  } else {
    tr = blockIdx.x;
    tc = blockIdx.y;
    if (tr < tc) return;
  }
  __shared__ double s2all[KT];
  for (int t = threadIdx.x; t < KT; t += 256) s2all[t] = X[int64_t(tc) * KT + t];
  __syncthreads();
  const double* s2 = s2all + cq * NCOL;
  const int64_t c0 = int64_t(tc) * KT + cq * NCOL, gi = int64_t(tr) * KT + 2 * l;
  const double xr0 = X[gi], xr1 = X[gi + 1];
  const bool on_diag = tr == tc;
  const double dg0 = on_diag ? diag[gi] : 0.0, dg1 = on_diag ? diag[gi + 1] : 0.0;
  const double b = p0 * p0, y = 1.0 / b;
  const int ldiag = 2 * l - cq * NCOL;
  double* o = out + c0 * ld + gi;
#pragma unroll 4
  for (int c = 0; c < NCOL; ++c) {
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double dx = (h ? xr1 : xr0) - s2[c];
      const double a = dx * dx;
      if constexpr (MODE == 1) {
        v[h] = a;
      } else {
        const double q0 = a * y, r0 = __builtin_fma(-b, q0, a), q1 = __builtin_fma(r0, y, q0),
                     r1 = __builtin_fma(-b, q1, a), q = __builtin_fma(r1, y, q1);
        v[h] = amp * exp(-0.5 * q);
      }
    }
    if (on_diag) {
      if (c == ldiag) v[0] += dg0;
      if (c == ldiag + 1) v[1] += dg1;
    }
    T2 pair;
    pair.x = v[0];
    pair.y = v[1];
    T2* dst = reinterpret_cast<T2*>(o + int64_t(c) * ld);
    if constexpr (MODE == 2) {
      if (v[0] == 12345.678 && v[1] == 9.75) *dst = pair;
    } else if constexpr (MODE == 4 || MODE == 5) {
      __builtin_nontemporal_store(pair, dst);
    } else {
      *dst = pair;
    }
  }
}

template <int MODE>
static void run(const char* what, int64_t n, const double* X, const double* diag, double* out) {
  const int nt = int(n / KT);
  dim3 grid = (MODE == 3 || MODE == 5) ? dim3(unsigned(nt * (nt + 1) / 2)) : dim3(nt, nt);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE>), grid, dim3(256), 0, 0, 2.5, 1.5, n, X, diag, out, n);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE>), grid, dim3(256), 0, 0, 2.5, 1.5, n, X, diag, out, n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = double(nt) * (nt + 1) / 2 * KT * KT * 8;
  printf("N = %lld  mode %d  %-52s %7.3f ms  %5.2f TB/s\n", (long long)n, MODE, what, ms, bytes / ms * 1e-9);
}

int main(int argc, char** argv) {
  for (int64_t n : {16384, 32768}) {
    std::vector<double> hx(n), hd(n, 0.01);
    for (int64_t i = 0; i < n; ++i) hx[i] = 10.0 * double(i) / double(n) + 1e-3 * double((i * 7919) % 13);
    double *X, *diag, *out;
    hipMalloc(&X, n * 8);
    hipMalloc(&diag, n * 8);
    hipMalloc(&out, size_t(n) * n * 8);
    hipMemcpy(X, hx.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(diag, hd.data(), n * 8, hipMemcpyHostToDevice);
    run<0>("as shipped", n, X, diag, out);
    run<1>("stores only", n, X, diag, out);
    run<2>("arithmetic only", n, X, diag, out);
    run<3>("grid of the lower tiles only", n, X, diag, out);
    run<4>("non-temporal stores", n, X, diag, out);
    run<5>("lower tiles only + non-temporal stores", n, X, diag, out);
    hipFree(X);
    hipFree(diag);
    hipFree(out);
  }
  return 0;
}
