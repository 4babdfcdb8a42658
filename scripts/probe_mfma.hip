// Probe the lane -> element mapping of v_mfma_f64_4x4x4_4b_f64 (and 16x16x4 as a control).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe_4x4(unsigned long long* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m;
    }
}
__global__ void probe_16(unsigned long long* out) {  // out[(la*64+lb)*4 + r]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d4{0, 0, 0, 0}, 0, 0, 0);
      for (int r = 0; r < 4; ++r) {
        unsigned long long m = __ballot(d[r] != 0.0);
        if (lane == 0) out[(la * 64 + lb) * 4 + r] = m;
      }
    }
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * 64 * 4 * 8);
  std::vector<unsigned long long> h(64 * 64 * 4);
  hipLaunchKernelGGL(probe_4x4, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, 64 * 64 * 8, hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_4x4x4_4b: rows la (A lane), cols lb (B lane) -> output lane (.. = zero) ==\n");
  for (int la = 0; la < 64; ++la) {
    printf("la=%2d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      unsigned long long m = h[la * 64 + lb];
      if (!m) continue;
      int cnt = __builtin_popcountll(m);
      printf(" lb%d->", lb);
      for (int t = 0; t < 64; ++t) if (m >> t & 1) printf("%d%s", t, cnt > 1 ? "," : "");
    }
    printf("\n");
  }
  hipLaunchKernelGGL(probe_16, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h.data(), d, 64 * 64 * 4 * 8, hipMemcpyDeviceToHost);
  printf("== v_mfma_f64_16x16x4 control: la in {0,1,16,17}, first matches ==\n");
  int las[4] = {0, 1, 16, 17};
  for (int q = 0; q < 4; ++q) {
    int la = las[q];
    printf("la=%2d:", la);
    int shown = 0;
    for (int lb = 0; lb < 64 && shown < 6; ++lb)
      for (int r = 0; r < 4; ++r) {
        unsigned long long m = h[(la * 64 + lb) * 4 + r];
        if (m) { printf(" lb%d->(lane %d,reg %d)", lb, __builtin_ctzll(m), r); ++shown; }
      }
    printf("\n");
  }
  return 0;
}
