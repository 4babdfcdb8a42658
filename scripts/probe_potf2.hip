// Phase timing of potf2_kernel<double> (s_memtime stamps by thread 0).
#define TGP_POTF2_STAMPS 1
#include "../tinygp_amd/csrc/chol.hip"
#include <vector>
#include <cstdio>
namespace tgp {
void set_error(const char*, ...) {}
template <typename T>
int launch_gemm_nt(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*,
                   int64_t, int, int, int) { return 0; }
template int launch_gemm_nt<float>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t, int, int, int);
template int launch_gemm_nt<double>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*, int64_t, int, int, int);
template <typename T>
int launch_gemm_tri(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const T*, int64_t, const T*, int64_t, T*, int64_t, int, int, int,
                    int64_t, int64_t, int64_t) { return 0; }
template int launch_gemm_tri<float>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t, int, int, int, int64_t, int64_t, int64_t);
template int launch_gemm_tri<double>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*, int64_t, int, int, int, int64_t, int64_t, int64_t);
int ensure_solve_stream(tgp_ctx*) { return 0; }
int ensure_work(tgp_ctx*, size_t) { return 0; }
}
int main() {
  const int n = 128;
  std::vector<double> h(n * n);
  for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) h[j * n + i] = (i == j ? n + 1.0 : 1.0 / (1 + abs(i - j)));
  double *dA, *dinv; int* info; long long* st;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dinv, 2048 * 8); hipMalloc(&info, 4);
  long long hs[64];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, h.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 4);
    hipLaunchKernelGGL((tgp::potf2_kernel<double, false>), dim3(1), dim3(512), 0, 0, dA, (int64_t)n, dinv, info, 0, (const double*)nullptr, (int64_t)0, (const int32_t*)nullptr, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(tgp::g_potf2_stamps), sizeof(hs));
    printf("rep %d total %lld cycles; load %lld; store %lld\n", rep, hs[34] - hs[0], hs[1] - hs[0], hs[34] - hs[33]);
    for (int kb = 0; kb < 8; ++kb) {
      long long p1 = hs[2 + 4 * kb] - hs[1 + 4 * kb], b1 = hs[3 + 4 * kb] - hs[2 + 4 * kb], p2 = hs[4 + 4 * kb] - hs[3 + 4 * kb];
      long long nxt = (kb < 7) ? hs[1 + 4 * (kb + 1)] : hs[33];
      printf("  kb %d: P1 %6lld  barrier %5lld  P2+bar %6lld  P3+bar %6lld\n", kb, p1, b1, p2, nxt - hs[4 + 4 * kb]);
    }
  }
  return 0;
}
