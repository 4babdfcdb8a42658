// Can a STREAM wait for a device-side counter without a spinning kernel of OURS?  hipStreamWaitValue32 -- which turned out
// to be a one-wave wait kernel of the runtime (__amd_rocclr_streamOpsWait in a kernel trace, profiles/r05_e), not a
// command-processor packet: as fast as our poll kernel, but without a timeout.  Measured here:
//   * is it supported, on signal memory (hipMallocSignalMemory, 8 bytes) and on plain hipMalloc memory;
//   * the hand-off latency producer-kernel atomic -> first instruction of the kernel queued behind the wait,
//     next to the one-wave poll kernel the chain's followers use today (chain_poll_kernel);
//   * what the producer pays for an atomic on signal memory (it may live in host memory).
// Times: s_memrealtime (100 MHz).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/probe_waitvalue scripts/probe_waitvalue.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned* counter, long long* stamps, int delay_ticks, int system_scope) {
  if (threadIdx.x != 0) return;
  long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
  long long ta = (long long)__builtin_amdgcn_s_memrealtime();
  if (system_scope) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  long long tb = (long long)__builtin_amdgcn_s_memrealtime();
  stamps[0] = ta;
  stamps[1] = tb;
  while ((long long)__builtin_amdgcn_s_memrealtime() - tb < delay_ticks) __builtin_amdgcn_s_sleep(8);  // still running
  stamps[3] = (long long)__builtin_amdgcn_s_memrealtime();
}
__global__ void follower(long long* stamps) {
  if (threadIdx.x == 0) stamps[2] = (long long)__builtin_amdgcn_s_memrealtime();
}
__global__ void poller(const unsigned* counter, unsigned target) {
  if (threadIdx.x != 0) return;
  while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) __builtin_amdgcn_s_sleep(16);
}

int main() {
  int can = -1;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  long long* d_st;
  CK(hipMalloc(&d_st, 64));
  unsigned *sig = nullptr, *plain = nullptr;
  hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
  printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(es));
  CK(hipMalloc(&plain, 64));
  struct Mode { const char* name; unsigned* ptr; int wait; int sys; };
  std::vector<Mode> modes = {{"poll kernel, plain memory (today)", plain, 0, 0},
                             {"hipStreamWaitValue32, plain memory", plain, 1, 0},
                             {"hipStreamWaitValue32, plain memory, system-scope atomic", plain, 1, 1}};
  if (es == hipSuccess) {
    modes.push_back({"hipStreamWaitValue32, signal memory", sig, 1, 1});
    modes.push_back({"poll kernel, signal memory", sig, 0, 1});
  }
  for (auto& m : modes) {
    std::vector<double> lat, atom;
    bool failed = false;
    for (int rep = 0; rep < 12 && !failed; ++rep) {
      CK(hipMemset(m.ptr, 0, 4));
      CK(hipMemset(d_st, 0, 64));
      CK(hipDeviceSynchronize());
      if (m.wait) {
        hipError_t e = hipStreamWaitValue32(sb, m.ptr, 1, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("%-58s : %s\n", m.name, hipGetErrorString(e)); failed = true; break; }
      } else {
        hipLaunchKernelGGL(poller, dim3(1), dim3(64), 0, sb, m.ptr, 1u);
      }
      hipLaunchKernelGGL(follower, dim3(1), dim3(64), 0, sb, d_st);
      hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, sa, m.ptr, d_st, 30000 /* 300 us */, m.sys);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("%-58s : sync %s\n", m.name, hipGetErrorString(e)); failed = true; break; }
      long long h[4];
      CK(hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost));
      if (rep >= 2) {  // (h[2] < h[0]: the follower did not wait at all)
        lat.push_back((h[2] - h[1]) * 0.01);
        atom.push_back((h[1] - h[0]) * 0.01);
      }
      if (rep == 2) printf("   [%s] follower started %.2f us after the atomic returned, %.2f us before the producer ended\n",
                           m.name, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01);
    }
    if (failed || lat.empty()) continue;
    std::sort(lat.begin(), lat.end());
    std::sort(atom.begin(), atom.end());
    printf("%-58s : hand-off median %.2f us (min %.2f, max %.2f); atomic median %.2f us\n", m.name, lat[lat.size() / 2],
           lat.front(), lat.back(), atom[atom.size() / 2]);
  }
  // sixteen waits queued in a row on ONE stream against increasing targets (the followers of a 16-column launch)
  if (can == 1) {
    unsigned* p = es == hipSuccess ? sig : plain;
    CK(hipMemset(p, 0, 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sb));
    for (int k = 0; k < 16; ++k) {
      if (hipStreamWaitValue32(sb, p, 0, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { printf("chain of waits failed\n"); return 0; }
      hipLaunchKernelGGL(follower, dim3(1), dim3(64), 0, sb, d_st);
    }
    CK(hipEventRecord(e1, sb));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("16 x (satisfied wait + empty kernel) on one stream: %.1f us each\n", ms * 1000 / 16);
    CK(hipEventRecord(e0, sb));
    for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(follower, dim3(1), dim3(64), 0, sb, d_st);
    CK(hipEventRecord(e1, sb));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("16 x empty kernel on one stream: %.1f us each\n", ms * 1000 / 16);
  }
  return 0;
}
