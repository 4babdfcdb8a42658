// Does v_mfma_f64_4x4x4_4b honour CBSZ/ABID (A-block broadcast)?  If so, four of them
// (cbsz=2, abid=0..3) reproduce one v_mfma_f64_16x16x4 with identical operand / D layouts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void check(double* out) {
  const int lane = threadIdx.x;
  double a = sin(0.37 * lane + 0.1), b = cos(0.91 * lane + 0.3);
  d4 c = {0.01 * lane, -0.02 * lane, 0.5, 1.0 + lane};
  d4 ref = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  d4 emu;
  emu[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[0], 2, 0, 0);
  emu[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[1], 2, 1, 0);
  emu[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[2], 2, 2, 0);
  emu[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[3], 2, 3, 0);
  for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = ref[r]; out[lane * 8 + 4 + r] = emu[r]; }
}

template <int MODE>
__global__ __launch_bounds__(256) void bench(double* out, int iters) {
  double acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = 0;
  const double a = threadIdx.x * 1e-3, b = (blockIdx.x + 1) * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) {
        acc[4 * i + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * i + 0], 2, 0, 0);
        acc[4 * i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * i + 1], 2, 1, 0);
        acc[4 * i + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * i + 2], 2, 2, 0);
        acc[4 * i + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * i + 3], 2, 3, 0);
      } else {
        d4 c = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        acc[4 * i] = c[0]; acc[4 * i + 1] = c[1]; acc[4 * i + 2] = c[2]; acc[4 * i + 3] = c[3];
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 32; ++i) s += acc[i];
  if (s == -1.2345) out[0] = s;
}

int main() {
  double* d;
  hipMalloc(&d, 64 * 8 * 8);
  double h[64 * 8];
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double maxdiff = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) maxdiff = fmax(maxdiff, fabs(h[l * 8 + r] - h[l * 8 + 4 + r]));
  printf("cbsz/abid emulation vs 16x16x4: max |diff| = %.3e (lane0 ref %.6f %.6f emu %.6f %.6f)\n", maxdiff,
         h[0], h[1], h[4], h[5]);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  for (int mode = 0; mode < 2; ++mode)
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
      const int iters = 2048, blocks = p.multiProcessorCount * bpc;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
        else hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      double tf = double(blocks) * 4 * iters * 8 * 2048.0 / (best * 1e-3) / 1e12;
      printf("%s waves/SIMD=%d : %.2f TFLOP/s\n", mode == 0 ? "4x(4x4x4_4b cbsz=2)" : "16x16x4            ", bpc, tf);
    }
  return 0;
}
