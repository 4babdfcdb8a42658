// Which compute units does a hipExtStreamCreateWithCUMask stream use?  Each workgroup records
// (XCC_ID, HW_ID) and spins ~20 us so that the whole grid is resident at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void who(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, words = (ncu + 31) / 32;
  unsigned* d; hipMalloc(&d, 4096 * 8);
  std::vector<unsigned> h(4096 * 2);
  for (int variant = 0; variant < 5; ++variant) {
    std::vector<uint32_t> mask(words, 0xffffffffu);
    const char* name = "all";
    if (variant == 1) { mask[0] = 0xffffff00u; name = "bits0-7 off"; }
    if (variant == 2) { mask[0] = 0x00000000u; name = "word0 off"; }
    if (variant == 3) { for (int w = 0; w < words; ++w) mask[w] = 0xfffffffeu; name = "bit0 of every word off"; }
    if (variant == 4) { mask[words - 1] = 0x00ffffffu; name = "top 8 bits off"; }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
    if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0xff, 4096 * 8, s);
    hipLaunchKernelGGL(who, dim3(2048), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;  // xcc -> set of (se, cu)
    for (int b = 0; b < 2048; ++b) {
      unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
      cus[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%-26s:", name);
    for (auto& kv : cus) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  total distinct CUs %d\n", total);
    hipStreamDestroy(s);
  }
  return 0;
}
