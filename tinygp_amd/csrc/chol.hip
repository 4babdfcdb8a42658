// chol.hip -- right-looking blocked LL^T (K4), blocked triangular solves (K5/K6).
//
// Replaces jax.scipy.linalg.cholesky / solve_triangular at reference
// solvers/direct.py:53,66-70.  Column-major, in place, lower triangle only.
//
//   outer block NB (default 1024): panel = [potf2 128 | trsm | in-panel update] x NB/128
//                                  trailing update = gemm_nt (fp64 MFMA), K = NB
//   look-ahead: the next panel's block column is updated first, then factored on a
//   high-priority side stream while the main stream updates the rest of the matrix.
//
// potf2   one workgroup (8 waves) factors a 128x128 diagonal block resident in LDS (36 lower
//         16x16 blocks, block-contiguous): per 16-column step the column block is eliminated in
//         VALU registers with lane = row, the rank-16 updates run on the MFMAs, and the inverses
//         of the eight 16x16 diagonal sub-blocks ("dinv") are emitted, which turn every later
//         triangular solve into MFMAs.
// trsm    X L^T = B for a 128-column panel: one wave per 16 rows, transposed recurrence
//         Y_j = inv(L_jj) (B_j^T - sum_{k<j} L_jk Y_k) so that an MFMA result (D layout)
//         is directly the next MFMA's B operand -- no LDS, no shuffles.
#include <algorithm>
#include <atomic>
#include <functional>
#include <type_traits>

#include <thread>

#include "tgp_common.h"
#define CHAIN_HD __host__ __device__
#include "chain_tasks.h"

namespace tgp {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  using acc_t = d4;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
  // v_mfma_f64_4x4x4_4b_f64: 73-76 TFLOP/s on this part against 46-49 for the 16x16x4 form (gemm.hip).  One
  // instruction = four independent 4x4x4 products: lane (k = l>>4, q = (l>>2)&3, e = l&3) supplies A_q[e][k] and
  // B_q[k][e], lane (i = l>>4, q, j = l&3) receives D_q[i][j].  A 16 x 16 x 4 step = four instructions whose B operand
  // is read with its 4-row groups rotated by t = 0..3 (row rot4(lane, t) of the 16): acc[t] then holds, for the
  // A-side row 4q + i and the B-side row rot4(lane, t), the product the 16x16x4 form keeps in its four-entry vector.
  // MEASURED and switched off (profiles/r04_d): in the chain's update tasks (one workgroup per compute unit, wave tile
  // 32 x 64, register-staged operands under the 128-VGPR cap) the form spills inside the k-loop and the tasks take
  // 33-39 us instead of 19-25; in potf2's fold it spills 39-77 registers.  The 16x16x4 form stays until those kernels
  // stage their operands LDS-direct like gemm.hip's.
  static constexpr bool FAST4 = false;
  static __device__ __forceinline__ double mma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int rot4(int lane, int t) { return 4 * ((((lane >> 2) & 3) + t) & 3) + (lane & 3); }
  static __device__ __forceinline__ int arow4(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
};
template <> struct Mfma<float> {
  using acc_t = f4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) * 4 + r; }
  static constexpr bool FAST4 = false;  // (fp32 has no faster small form: the 16x16x4 instruction is the one to use)
  static __device__ __forceinline__ float mma4(float a, float b, float c) { return c + a * b; }
  static __device__ __forceinline__ int rot4(int lane, int t) { return (lane + t) & 15; }
  static __device__ __forceinline__ int arow4(int lane) { return lane & 15; }
};

// 1/x and 1/sqrt(x) from the hardware seed + two Newton steps (<= 1 ulp-class; no IEEE
// division sequence on the factorisation's critical path)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(0.5 * y, e, y);
}
__device__ __forceinline__ float fast_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  float e = __builtin_fmaf(-x * y, y, 1.0f);
  return __builtin_fmaf(0.5f * y, e, y);
}

// ---------------------------------------------------------------------------------------
// potf2: 128x128 diagonal block in ONE workgroup (8 waves), matrix resident in LDS.
//
// Right-looking over 16-column blocks kb = 0..7:
//   E   the whole column block (diagonal block + the blocks below it) is eliminated in VALU
//       registers with lane = row; the pivot-row values are broadcast through v_readlane ->
//       SGPR operands (1..3 eliminating waves, three blocks below the diagonal per wave);
//   U1  rank-16 update of column block kb+1 with MFMAs (one block pair per wave) -- all that
//       the next elimination needs;
//   U2  beside the next elimination: the remaining block pairs (A_ij -= L_ik L_jk^T), the
//       16x16 inverse W = L_kk^-1 (stored to `dinv` for the trsm / trsv kernels that follow)
//       and the write-out of the finished column strip.
// Every MFMA operand is "16 consecutive rows at fixed k" of a column-major 16x16 block of the
// LDS image, and every D tile is written back row-contiguous.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// LDS image of the tile: blk(i, j) (potf2_layout.h, shared with the host replay of potf2_body.inc)
#define TGP_HD __device__ __forceinline__
#include "potf2_layout.h"
// keeps a lane-derived value opaque to the optimiser (predicates are recomputed, not hoisted)
#define TGP_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))

#ifdef TGP_POTF2_STAMPS
__device__ long long g_potf2_stamps[64];
#define POTF2_STAMP(i) do { if (threadIdx.x == 0) g_potf2_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define POTF2_STAMP(i) do {} while (0)
#endif

// waves_per_eu(4): at most 128 VGPRs, so that two potf2 waves fit on a SIMD beside one wave of the
// trailing-update GEMM (<= 251 VGPRs) -- otherwise potf2 waits for the whole update to drain.
// (round 6) `wait_count` != NULL: the launch starts with the stream-side end of a hand-off -- thread 0 polls the counter
// (the PREFIX of the merged trailing update that is running on the main stream: gemm.hip), one acquire, barrier.  As a
// kernel of its own in front of this one (chain_poll_kernel, one wave) the poll got its slot at once and potf2 -- 74 KB of
// LDS -- was dispatched when the prefix was complete, i.e. right after the update's SECOND round of tiles had taken every
// slot: it waited a whole round, 290 of the 316 us it took beside the updates with 7 168 / 8 192 rows (profiles/r06_i).
// Dispatched with the poll inside, it takes the first slot the first round frees and is through 27 us after the prefix.
__device__ void potf2_prefix_wait(const int32_t* count, int32_t target, int32_t* info);

template <typename T, bool FOLD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void potf2_kernel(T* __restrict__ A, int64_t ld,
                                                    T* __restrict__ dinv,
                                                    int32_t* __restrict__ info,
                                                    int32_t pivot_base,
                                                    const T* __restrict__ Xp, int64_t ldx,
                                                    const int32_t* __restrict__ wait_count, int32_t wait_target) {
  __shared__ __attribute__((aligned(16))) T S[36 * 256];
  __shared__ T Rs[2 * 16];                                 // 1 / L_ii of the current / previous block
  __shared__ T Dg[256];  // copy of the current diagonal 16 x 16 block for the eliminating waves of group >= 1
  if (wait_count != nullptr) {
    if (threadIdx.x == 0) potf2_prefix_wait(wait_count, wait_target, info);
    __syncthreads();
  }
#include "potf2_body.inc"
}

// ---------------------------------------------------------------------------------------
// Fused panel step: ONE launch per 128-column block instead of potf2 | trsm (+ the events between
// them and the in-panel update of the next column block).
//
//   workgroup 0        potf2 of the diagonal tile (with the fold of its pending update), then it
//                      publishes L_jj and its 16 x 16 inverses with an agent-scope release + flag;
//   workgroups 1..m/128  one per 128-row tile below: while potf2 runs they are already resident and
//                      apply the pending update of THEIR tile of this column block themselves
//                      (A_ij -= X_i,j-1 X_j,j-1^T: a 128 x 128 x 128 product on the MFMAs, operands
//                      double-buffered through the same 72 KiB of LDS potf2 uses), then wait for the
//                      flag, stage L_jj / its inverses in LDS and solve their rows (trsm).
//
// Why: beside a running trailing update the separate kernels waited for workgroup slots one
// after the other (trsm 55 us per launch in place against 12 on an idle chip, the in-panel update
// 126 us), and the update of column block j+1 sat between trsm(j) and trsm(j+1).  Here the rows'
// workgroups are placed while potf2 computes, the update they depend on is their own work, and
// the in-panel update that is left (column blocks j+2.. -- launch_gemm_nt, role 1) has a whole
// step of slack.  Workgroups are dispatched in index order, so workgroup 0 is resident whenever
// another one polls; the poll is bounded all the same (a lost producer ends in info = INT32_MIN,
// never in a hung GPU).
// ---------------------------------------------------------------------------------------
constexpr int32_t STEP_TIMEOUT = INT32_MIN;

// Device-side polls are bounded in WALL-CLOCK time (s_memrealtime, 100 MHz whatever the shader clock does), not by a
// spin count: a chain launch that shares the chip with other ranks, runs under a debugger / profiler or at a throttled
// clock is slow, not lost (advisor r4).  a single wait may last g_poll_limit ticks (ctx option poll_timeout_ms, default
// 4 s); the clock is read every 256 polls only.  A wait that does expire poisons `info` with STEP_TIMEOUT: the host
// reports TGP_E_TIMEOUT (never "not positive definite"), and tgp_solver_factor* retries once on the launch-per-block path.
__device__ long long g_poll_limit = 400000000LL;  // ticks of 10 ns (set_poll_limit)
struct PollClock {
  long long t0 = 0;
  __device__ __forceinline__ bool expired(long spin) {
    if ((spin & 255) != 255) return false;
    const long long now = (long long)__builtin_amdgcn_s_memrealtime();
    if (t0 == 0) {
      t0 = now;
      return false;
    }
    return now - t0 > __hip_atomic_load(&g_poll_limit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};
__device__ __forceinline__ bool poisoned(const int32_t* info) {
  return __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == STEP_TIMEOUT;
}

template <typename T, bool FOLD>
__device__ __forceinline__ void trsm_fold_body(T* S, int it, const T* __restrict__ Ljj,
                                               int64_t ld, const T* __restrict__ dinv,
                                               const T* __restrict__ Xp, const uint32_t* flag,
                                               uint32_t epoch, int wait_flag, int32_t* info) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  typedef T T2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  __builtin_amdgcn_s_setprio(2);
  T* Bt = const_cast<T*>(Ljj) + TILE + int64_t(it) * TILE;  // this workgroup's 128 rows of the column block

  if constexpr (FOLD) {
    // C (128 x 128 at Bt) -= Xi Xj^T, Xi = the same rows of the previous block column, Xj = Xp
    constexpr int FK = 16, F_LD = 144;  // as gemm_nt: [k][128 rows], 144 mod 32 == 16
    static_assert(4 * FK * F_LD == 36 * 256, "the fold's operand buffers are exactly potf2's tile image");
    T* sA = S;                  // [2][FK * F_LD]
    T* sB = S + 2 * FK * F_LD;  // [2][FK * F_LD]
    const T* Xi = Xp + TILE + int64_t(it) * TILE;
    const int wr = w >> 1, wc = w & 1;  // wave tile: 32 rows x 64 columns
    T2 ra[2], rb[2];
    auto load_global = [&](int kt) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int64_t kk = int64_t(kt) * FK + w + 8 * r;
        ra[r] = *reinterpret_cast<const T2*>(Xi + kk * ld + lane * 2);
        rb[r] = *reinterpret_cast<const T2*>(Xp + kk * ld + lane * 2);
      }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int kk = w + 8 * r;
        *reinterpret_cast<T2*>(&sA[buf * FK * F_LD + kk * F_LD + lane * 2]) = ra[r];
        *reinterpret_cast<T2*>(&sB[buf * FK * F_LD + kk * F_LD + lane * 2]) = rb[r];
      }
    };
    acc_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
    load_global(0);
    store_lds(0);
    __syncthreads();
    constexpr int nkt = TILE / FK;
    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) load_global(kt + 1);
      const T* pa = &sB[buf * FK * F_LD + lk * F_LD + wc * 64 + lrow];  // MFMA A operand <- Xj rows (C column)
      const T* pb = &sA[buf * FK * F_LD + lk * F_LD + wr * 32 + lrow];  // MFMA B operand <- Xi rows (C row)
#pragma unroll
      for (int ks = 0; ks < FK / 4; ++ks) {
        T aop[4], bop[2];
#pragma unroll
        for (int a = 0; a < 4; ++a) aop[a] = pa[ks * 4 * F_LD + a * 16];
#pragma unroll
        for (int b = 0; b < 2; ++b) bop[b] = pb[ks * 4 * F_LD + b * 16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = M::mma(aop[a], bop[b], acc[a][b]);
      }
      if (kt + 1 < nkt) store_lds(buf ^ 1);
      __syncthreads();
    }
    // C[wr*32 + b*16 + lrow, wc*64 + a*16 + drow(lane, r)] -= acc[a][b][r]; 16 loads in flight per pass
    // (uniform base in SGPRs + one 32-bit lane offset: no per-access address registers)
    T* Cu = Bt + int64_t(wc * 64) * ld + wr * 32;
    const uint32_t coff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      T c[2][4];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[b][r] = (Cu + int64_t(a * 16 + M::drow(0, r)) * ld + b * 16)[coff];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) (Cu + int64_t(a * 16 + M::drow(0, r)) * ld + b * 16)[coff] = c[b][r] - acc[a][b][r];
      __builtin_amdgcn_sched_barrier(0);  // one pass of eight loads at a time (register budget: 128)
    }
    __syncthreads();  // the updated rows are visible to the whole workgroup; S is free again
  }

  // this wave's 16 rows of the column block (the fold's result, if any), D layout -- fetched BEFORE the
  // wait: they do not depend on potf2, and the round trip disappears behind it
  T* bu = Bt + w * 16;  // wave-uniform; lane offset below
  const uint32_t boff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
  acc_t V[8];  // V[j]: B_j until step j, then Z_j = -Y_j
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[jb][r] = (bu + int64_t(jb * 16 + M::drow(0, r)) * ld)[boff];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (in registers before the acquire drops the L1)
  __builtin_amdgcn_sched_barrier(0);

  if (wait_flag) {  // L_jj and its inverses come from workgroup 0 of this launch
    if (tid == 0) {
      uint32_t seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      PollClock clk;
      for (unsigned spin = 0; seen != epoch; ++spin) {  // (unsigned: a wait of up to an hour wraps, it does not overflow)
        __builtin_amdgcn_s_sleep(2);
        seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (clk.expired(spin)) break;
      }
      if (seen != epoch) atomicExch(info, STEP_TIMEOUT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // (every wave reads the published tile for the first time below; the acquire above dropped
    // this CU's L1 lines, which potf2's own CU -- possibly this one -- filled before factoring)
  }

  // L_jj -> LDS in potf2's block image: the 28 blocks below the diagonal as they are, the diagonal
  // slots take the 16 x 16 inverses (all a solve needs of a diagonal block)
  {
    T tr[5][4];
#pragma unroll
    for (int trip = 0; trip < 5; ++trip) {
      const int b = w + 8 * trip;
      if (b < 36) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= b) ++i;
        const int j = b - i * (i + 1) / 2;
        if (i == j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) tr[trip][q] = dinv[i * 256 + q * 64 + lane];
        } else {
          const int voff = lk * int(ld) + lrow;
#pragma unroll
          for (int q = 0; q < 4; ++q) tr[trip][q] = (Ljj + int64_t(j * 16 + q * 4) * ld + i * 16)[voff];
        }
      }
    }
#pragma unroll
    for (int trip = 0; trip < 5; ++trip) {
      const int b = w + 8 * trip;
      if (b < 36) {
#pragma unroll
        for (int q = 0; q < 4; ++q) S[b * 256 + q * 64 + lane] = tr[trip][q];
      }
    }
  }
  __syncthreads();
  T* bs = bu;  // the stores recompute their addresses (kept from the loads they are 64 spilled registers)
  asm volatile("" : "+s"(bs));
  // transposed recurrence Y_j = inv(L_jj) (B_j^T - sum_{k<j} L_jk Y_k) as in trsm_kernel, operands
  // from LDS (element (r, c) of block (i, j) at blk(i, j) + c * 16 + r)
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    acc_t acc = V[jb], acc2 = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < jb; ++kb) {
      const T* Lb = &S[blk(jb, kb)];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T l = Lb[M::drow(lane, s) * 16 + lrow];
        if (s & 1) acc2 = M::mma(l, V[kb][s], acc2);
        else acc = M::mma(l, V[kb][s], acc);
      }
    }
    acc += acc2;
    const T* Db = &S[blk(jb, jb)];
    acc_t y = acc_t{0, 0, 0, 0}, y2 = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T d = Db[M::drow(lane, s) * 16 + lrow];
      if (s & 1) y2 = M::mma(d, acc[s], y2);
      else y = M::mma(d, acc[s], y);
    }
    y += y2;
#pragma unroll
    for (int r = 0; r < 4; ++r) (bs + int64_t(jb * 16 + M::drow(0, r)) * ld)[boff] = y[r];
    V[jb] = -y;
  }
}

template <typename T, bool FOLD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void panel_step_kernel(
    T* __restrict__ A, int64_t ld, T* __restrict__ dinv, int32_t* __restrict__ info, int32_t pivot_base,
    const T* __restrict__ Xp, int has_p, uint32_t* __restrict__ flag, uint32_t epoch) {
  __shared__ __attribute__((aligned(16))) T S[36 * 256];
  __shared__ T Rs[2 * 16];
  __shared__ T Dg[256];
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x);
  if (b < has_p) {
    {
      const int64_t ldx = ld;
#include "potf2_body.inc"
    }
    // publish: every wave's stores have left it -> barrier -> one agent-scope release -> flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the fence's own wait)
      __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  trsm_fold_body<T, FOLD>(S, b - has_p, A, ld, dinv, Xp, flag, epoch, has_p, info);
}

// ---------------------------------------------------------------------------------------
// Persistent panel chain (round 4): ONE launch factors block columns [cb, ce) of a panel -- every potf2, every
// solve of the rows below AND every in-panel update -- instead of [potf2 | trsm | rank-128 update] x blocks on
// two streams with an event pair per block.  No dependent-launch gaps (7 us each, two per block), no update
// stream, nothing for the host to submit per block.
//
// Work = tile tasks, one per workgroup, handed out by a ticket counter in an order in which every task depends on
// EARLIER tickets only, so a workgroup never waits for one that has not started (no co-residency requirement):
//
//   diag(c)        the diagonal chain's own workgroup: solves tile (c, c-1) the moment L_{c-1,c-1} is published,
//                  publishes it, folds it -- X_{c,c-1} never leaves the registers -- into tile (c, c) and factors
//                  that (potf2).  Between two potf2 there is ONE hand-off.
//   solve(i, c)    X_ic = A_ic L_cc^-T for the other rows (transposed recurrence, as trsm_fold_body)
//   update(i,c,k)  A_ic -= X_ik X_ck^T, one 128 x 128 x 128 product per task (lower blocks only on the diagonal):
//                  right-looking, so that the updates behind column k spread over the whole chip the moment its
//                  tiles are solved.  (A first version accumulated them left-looking inside the task that solves
//                  the tile: one compute unit then owes a tile up to seven products of 22-32 us each, and the
//                  chain waited for exactly those -- profiles/r04_b.)  The update (c, c, c-1) is diag(c)'s fold.
//   order          per column k: diag(k+1), solve(.., k), update(.., c, k) for c = k+1 .. (column k+1 first)
//
// State of tile (i, c) = ONE word, epoch * 128 + s: s = k + 1 once the updates from block columns <= k are applied
// (those from columns < cb were applied by the launches before), s = 127 once the tile is final.  The epoch is the
// launch's (nothing to reset).  Hand-offs follow MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, form
// R1: everything another workgroup reads is stored WRITE-THROUGH (agent-scope relaxed atomic stores = `sc1`), every
// storing wave drains (`s_waitcnt vmcnt(0)`), barrier, ONE lane stores the word; a consumer polls with ONE wave
// (relaxed, s_sleep), ONE agent-scope acquire, barrier, then plain loads.  Every poll is bounded and a timeout
// poisons `info` for all (STEP_TIMEOUT): a lost producer ends in an error code, never in a hung GPU.
// ---------------------------------------------------------------------------------------
template <typename T> struct AgentBits;
template <> struct AgentBits<double> { using t = unsigned long long; };
template <> struct AgentBits<float> { using t = unsigned int; };
template <typename T>
__device__ __forceinline__ void st_agent(T* p, T v) {
  typename AgentBits<T>::t b;
  __builtin_memcpy(&b, &v, sizeof(T));
  __hip_atomic_store(reinterpret_cast<typename AgentBits<T>::t*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// time stamp k of this workgroup's task (100 MHz real-time counter; the hook is off unless chain_stamps = 1)
__device__ __forceinline__ void chain_stamp(long long* st, int k) {
  if (st != nullptr && threadIdx.x == 0) st[4 + k] = (long long)__builtin_amdgcn_s_memrealtime();
}

constexpr int CHAIN_FLAG_LD = 64;  // state word of tile (i, c) of the panel: flags[i * 64 + c] (chains of <= 64 block columns)
constexpr uint32_t CHAIN_FINAL = 127;
constexpr int CHAIN_COLCNT_OFF = 16;  // d_chain_ticket: [0] the ticket counter, [16 + c] final tiles of block column c,
constexpr int CHAIN_QCNT_OFF = 96;    // [96 + k] finished parts of the update of tile (k+2, k+1) from column k
constexpr int CHAIN_XSTEP_OFF = 160;  // [160 + c] column blocks of X_{c,c-1} complete in memory (0..8): xsolve(c) -> diag(c)
constexpr int CHAIN_ZFLAG_OFF = 224;   // [224 + c] z_c = L_cc^-1 y_c is in memory (forward substitution as tasks: fsolve -> fupdate)
constexpr int CHAIN_YSTATE_OFF = 288;  // [288 + g] block columns of this launch applied to the rows of group g of y
constexpr int CHAIN_TICKET_WORDS = CHAIN_YSTATE_OFF + int(CHAIN_MAX_ROW_TILES) / CHAIN_FWD_GROUP;
// (outside the words a chain launch zeroes) [CHAIN_PREFIX_OFF + q]: finished prefix tiles of the merged trailing update q & 3
constexpr int CHAIN_PREFIX_OFF = CHAIN_TICKET_WORDS + 16;

template <typename T>
struct ChainArgs {
  T* A0;        // the panel's origin: tile (i, c) at A0 + c * 128 * ld + i * 128
  int64_t ld;
  T* dinv;      // 16 x 16 inverses of the panel's diagonal blocks, 2048 entries per block column
  int32_t* info;
  uint32_t* flags;
  int32_t* ticket;
  uint32_t epoch32;    // epoch * 128 (the state words' high bits)
  int32_t pivot_base;  // global index of the panel's first pivot
  int32_t R;           // row tiles of the panel (rows of A0 down to the end of the matrix)
  int32_t nblk;        // block columns of the panel
  int32_t cb, ce;      // block columns [cb, ce) of the panel are factored by this launch; [0, cb) are final, and so
                       // is L_00 when cb == 0 (a panel's first block is factored by potf2_kernel in front)
  long long* stamps;   // measurement hook (ctx option chain_stamps): 16 words per task from this base, or NULL
  int32_t launch;
  int32_t fast_update;  // fp64 whole-tile updates on the 4x4x4 MFMA form, LDS-direct operands (chain_update_fast)
  const uint64_t* tasks;  // the launch's task list in ticket order, one packed word per task (chain_tasks.h, chain_pack)
  T* y;                   // forward substitution as tasks: the right-hand side's rows from the panel's first row on (or NULL)
  int32_t fgs;            // row tiles per fupdate task
  int32_t fprev;          // rows / columns in FRONT of the panel exist: fsolve(0) applies tile (0, -1) of the previous panel
  // the evaluation's two reductions ride along (round 6): fsolve(c) leaves sum_i z_i^2 and sum_i log L_ii of ITS block in
  // red[2 g] / red[2 g + 1] (g = the block's global index) and the fsolve of the matrix's LAST block adds all of them up in a
  // fixed order into scal[0] / scal[1] -- no reduction launch behind the factorisation (NULL: off)
  double* red;
  double* scal;
  int32_t red_total;      // > 0: this launch holds the matrix's last block; blocks of the whole matrix
};

// all threads; wave 0 polls up to three state words (lane l: word f[l] == v[l]; NULL: nothing to wait for),
// one acquire, barrier.  NAP: s_sleep between polls (the diagonal chain polls eagerly, the others politely).
template <int NAP>
__device__ __forceinline__ void chain_wait(const uint32_t* f0, uint32_t v0, const uint32_t* f1, uint32_t v1,
                                           const uint32_t* f2, uint32_t v2, int32_t* info) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const uint32_t* f = lane == 0 ? f0 : (lane == 1 ? f1 : (lane == 2 ? f2 : nullptr));
    const uint32_t want = lane == 0 ? v0 : (lane == 1 ? v1 : v2);
    bool ok = f == nullptr || __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
    unsigned spin = 0;
    PollClock clk;
    while (!__all(ok)) {
      __builtin_amdgcn_s_sleep(NAP);
      if (!ok) ok = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
      bool dead = false;
      if ((spin & 255) == 255 && lane == 0)  // somebody timed out, or this wait has outlasted the wall-clock limit
        dead = clk.expired(spin) || poisoned(info);
      ++spin;
      if (__any(dead)) {
        if (lane == 0) atomicExch(info, STEP_TIMEOUT);
        break;
      }
    }
    if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// every storing wave has drained -> barrier -> one lane publishes the tile's new state; `colcnt` != NULL (the tile
// is FINAL): its block column's count of final tiles goes up -- what the stream-side pollers watch (chain_poll_kernel)
__device__ __forceinline__ void chain_publish(uint32_t* word, uint32_t value, int32_t* colcnt = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (colcnt != nullptr) __hip_atomic_fetch_add(colcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ void potf2_prefix_wait(const int32_t* count, int32_t target, int32_t* info) {
  PollClock clk;
  for (unsigned spin = 0;; ++spin) {
    if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
    __builtin_amdgcn_s_sleep(8);
    if ((spin & 255) == 255 && (clk.expired(spin) || poisoned(info))) {
      atomicExch(info, STEP_TIMEOUT);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Stream-side end of a hand-off: ONE wave that returns once block column c of the running chain launch is final
// (count >= target), so that the kernels behind it on ITS stream -- the forward-substitution step of that block,
// the early share of the next block-column update -- start while the chain launch is still running.  Bounded like
// every poll; a timeout poisons `info`.
__global__ __launch_bounds__(64) void chain_poll_kernel(const int32_t* __restrict__ count, int32_t target,
                                                        int32_t* __restrict__ info) {
  if (threadIdx.x != 0) return;
  PollClock clk;
  for (unsigned spin = 0;; ++spin) {
    if (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
    __builtin_amdgcn_s_sleep(16);
    if ((spin & 255) == 255 && (clk.expired(spin) || poisoned(info))) {
      atomicExch(info, STEP_TIMEOUT);
      break;
    }
  }
}

// update(i, c, k), i > c:  A_ic -= X_ik X_ck^T.  128 x 128 x 128 on the MFMAs, operands double-buffered through S
// as in trsm_fold_body; the tile is read and written once, write-through.
// PARTS = 4 / 8: rows 32 part .. + 31 / 16 part .. + 15 of the tile only, every wave 16 of its 128 columns -- a quarter /
// an eighth of the MFMA work per wave.  The update of tile (k+2, k+1) from column k is the LAST thing xsolve(k+2) waits
// for (it runs beside potf2(k+1) only if the tile is there in time): CHAIN_CRIT_PARTS workgroups share it.
template <typename T, int PARTS>
__device__ __forceinline__ void chain_update_full(const ChainArgs<T>& q, T* S, int i, int c, int k, int part) {
  static_assert(PARTS == 1 || PARTS == 4 || PARTS == 8, "whole tile, row quarters or row eighths");
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  typedef T T2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  const int64_t ld = q.ld;
  constexpr int FK = 16, F_LD = 144;  // as gemm_nt: [k][128 rows], 144 mod 32 == 16
  static_assert(4 * FK * F_LD == 36 * 256, "the operand buffers are exactly potf2's tile image");
  T* sA = S;                  // [2][FK * F_LD]
  T* sB = S + 2 * FK * F_LD;  // [2][FK * F_LD]
  // wave tile: 32 rows x 64 columns (a quarter task: 32 x 16, an eighth: 16 x 16)
  const int rbase = PARTS == 1 ? (w >> 1) * 32 : (PARTS == 4 ? part * 32 : part * 16);  // its first row
  const int cbase = PARTS == 1 ? (w & 1) * 64 : w * 16;                                 // its first column
  constexpr int NA = PARTS == 1 ? 4 : 1;   // 16-column blocks per wave
  constexpr int NBR = PARTS == 8 ? 1 : 2;  // 16-row blocks per wave
  // acc[a][b]: the 16 x 16 block (rows wr*32 + b*16.., columns wc*64 + a*16..) of the wave's 32 x 64 tile.  fp64: four
  // scalars per lane from the 4x4x4_4b instruction (entry t <-> row rot4(lane, t), column arow4(lane) of the block);
  // fp32: the 16x16x4 form's vector (entry r <-> row lrow, column drow(lane, r))
  acc_t acc[NA][NBR];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NBR; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
  const T* Xi = q.A0 + int64_t(k) * TILE * ld + int64_t(i) * TILE;
  const T* Xj = q.A0 + int64_t(k) * TILE * ld + int64_t(c) * TILE;
  T2 ra[2], rb[2];
  auto load_global = [&](int kt) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t kk = int64_t(kt) * FK + w + 8 * r;
      ra[r] = *reinterpret_cast<const T2*>(Xi + kk * ld + lane * 2);
      rb[r] = *reinterpret_cast<const T2*>(Xj + kk * ld + lane * 2);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kk = w + 8 * r;
      *reinterpret_cast<T2*>(&sA[buf * FK * F_LD + kk * F_LD + lane * 2]) = ra[r];
      *reinterpret_cast<T2*>(&sB[buf * FK * F_LD + kk * F_LD + lane * 2]) = rb[r];
    }
  };
  int rot[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) rot[t] = M::rot4(lane, t);
  load_global(0);
  store_lds(0);
  __syncthreads();
  constexpr int nkt = TILE / FK;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_global(kt + 1);
    const T* pa = &sB[buf * FK * F_LD + lk * F_LD + cbase + lrow];  // MFMA A operand <- Xj rows (C column)
    if constexpr (M::FAST4) {
      const T* pb = &sA[buf * FK * F_LD + lk * F_LD + rbase];  // MFMA B operand <- Xi rows (C row), rotated reads
#pragma unroll
      for (int ks = 0; ks < FK / 4; ++ks) {
        T aop[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) aop[a] = pa[ks * 4 * F_LD + a * 16];
#pragma unroll
        for (int b = 0; b < NBR; ++b) {
          T bop[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) bop[t] = pb[ks * 4 * F_LD + b * 16 + rot[t]];
#pragma unroll
          for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][b][t] = M::mma4(aop[a], bop[t], acc[a][b][t]);
        }
      }
    } else {
      const T* pb = &sA[buf * FK * F_LD + lk * F_LD + rbase + lrow];  // MFMA B operand <- Xi rows (C row)
#pragma unroll
      for (int ks = 0; ks < FK / 4; ++ks) {
        T aop[NA], bop[NBR];
#pragma unroll
        for (int a = 0; a < NA; ++a) aop[a] = pa[ks * 4 * F_LD + a * 16];
#pragma unroll
        for (int b = 0; b < NBR; ++b) bop[b] = pb[ks * 4 * F_LD + b * 16];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NBR; ++b) acc[a][b] = M::mma(aop[a], bop[b], acc[a][b]);
      }
    }
    if (kt + 1 < nkt) store_lds(buf ^ 1);
    __syncthreads();
  }
  T* Cu = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE + int64_t(cbase) * ld + rbase;
  if constexpr (M::FAST4) {
    // C[wr*32 + b*16 + rot[t], wc*64 + a*16 + arow4(lane)] -= acc[a][b][t]
    const int64_t ccol = M::arow4(lane);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      T cc[NBR][4];
      T* col = Cu + (int64_t(a * 16) + ccol) * ld;
#pragma unroll
      for (int b = 0; b < NBR; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) cc[b][t] = col[b * 16 + rot[t]];
#pragma unroll
      for (int b = 0; b < NBR; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) st_agent(col + b * 16 + rot[t], T(cc[b][t] - acc[a][b][t]));
      __builtin_amdgcn_sched_barrier(0);  // one pass of eight loads at a time (register budget: 128)
    }
  } else {
    // C[wr*32 + b*16 + lrow, wc*64 + a*16 + drow(lane, r)] -= acc[a][b][r]
    const uint32_t coff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      T cc[NBR][4];
#pragma unroll
      for (int b = 0; b < NBR; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cc[b][r] = (Cu + int64_t(a * 16 + M::drow(0, r)) * ld + b * 16)[coff];
#pragma unroll
      for (int b = 0; b < NBR; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          st_agent(Cu + int64_t(a * 16 + M::drow(0, r)) * ld + b * 16 + coff, T(cc[b][r] - acc[a][b][r]));
      __builtin_amdgcn_sched_barrier(0);  // one pass of eight loads at a time (register budget: 128)
    }
  }
}

// update(i, c, k), i > c, whole tile, fp64 (round 5, ctx option chain_fast_update): the 4x4x4 MFMA form -- 73-76 TFLOP/s
// instruction ceiling against 46-49 for 16x16x4, i.e. 16.7 instead of 107 cycles per 16 x 16 x 4 step and SIMD -- with
// the operands staged global -> LDS DIRECTLY (global_load_lds_dwordx4, one k-tile ahead, as gemm.hip): no staging
// registers, which is what made round 4's attempt spill under the 128-register cap.  The tile is read FIRST, into the
// accumulators, beside the first operand transfer; the MFMA's NEG field turns the products into -X_ik X_ck^T, so the
// accumulators end as the updated tile and the epilogue is 32 write-through stores per lane, nothing loaded.
// Wave tile 32 rows x 64 columns: acc[a][b][t] <-> row wr*32 + b*16 + rot4(lane, t), column wc*64 + a*16 + arow4(lane).
// Round 6: the product runs over block columns [k, k1) -- K = 128 (k1 - k), the operands of consecutive block columns are
// consecutive columns of the panel -- so that a BATCHED update (chain_tasks.h, ChainPolicy) reads and writes its tile once.
__device__ __forceinline__ void chain_update_fast(const ChainArgs<double>& q, double* S, int i, int c, int k, int k1) {
  using M = Mfma<double>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wu >> 1, wc = wu & 1;
  const int lk = lane >> 4, lrow = lane & 15;
  constexpr int FK = 16, F_LD = 144;  // [k][128 rows], 144 mod 32 == 16: conflict-free ds_read_b64
  const int64_t ld = q.ld;
  const double* Xi = q.A0 + int64_t(k) * TILE * ld + int64_t(i) * TILE;  // rows of the tile   -> MFMA B operand
  const double* Xj = q.A0 + int64_t(k) * TILE * ld + int64_t(c) * TILE;  // columns of the tile -> MFMA A operand
  double* sA = S;                  // [2][FK * F_LD]  X_ik
  double* sB = S + 2 * FK * F_LD;  // [2][FK * F_LD]  X_ck
  const uint32_t lds_a = uint32_t(size_t((__attribute__((address_space(3))) double*)(sA))) + uint32_t(wu * F_LD * 8);
  const uint32_t lds_b = uint32_t(size_t((__attribute__((address_space(3))) double*)(sB))) + uint32_t(wu * F_LD * 8);
  const uint32_t lane16 = uint32_t(lane) * 16u;
  // wave wu moves k-rows {wu, wu + 8} of both operands: one 1 KiB row per wave instruction, LDS address in M0
  auto issue_tile = [&](int kt, int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double* ap = Xi + (int64_t(kt) * FK + wu + 8 * r) * ld;
      const double* bp = Xj + (int64_t(kt) * FK + wu + 8 * r) * ld;
      const uint32_t off = uint32_t((buf * FK + 8 * r) * F_LD * 8);
      uint32_t keep;  // (m0 is a reserved register: saved and restored, not clobbered)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                   "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(lane16), "s"(ap), "s"(lds_a + off), "s"(bp), "s"(lds_b + off) : "memory");
    }
  };
  int rot[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) rot[t] = M::rot4(lane, t);
  // the tile: wave-uniform base + four loop-invariant 32-bit lane offsets
  char* Cu = reinterpret_cast<char*>(q.A0 + (int64_t(c) * TILE + wc * 64) * ld + int64_t(i) * TILE + wr * 32);
  uint32_t voff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) voff[t] = uint32_t((int64_t(M::arow4(lane)) * ld + rot[t]) * 8);
  issue_tile(0, 0);
  double acc[4][2][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const char* cb = Cu + (int64_t(a * 16) * ld + b * 16) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[a][b][t] = *reinterpret_cast<const double*>(cb + voff[t]);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the tile's values are consumed HERE as far as the compiler can tell: else its own wait for these loads lands inside
  // the k-loop as a vmcnt(0) behind the transfers each iteration has just issued (it cannot count the asm's loads)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[a][b][t]));
  __syncthreads();
  const int nkt = (k1 - k) * (TILE / FK);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) issue_tile(kt + 1, buf ^ 1);
    const double* pa = &sB[buf * FK * F_LD + lk * F_LD + wc * 64 + lrow];  // A operand <- X_ck rows (tile column)
    const double* pb = &sA[buf * FK * F_LD + lk * F_LD + wr * 32];         // B operand <- X_ik rows (tile row), rotated
#pragma unroll
    for (int ks = 0; ks < FK / 4; ++ks) {
      double aop[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) aop[a] = pa[ks * 4 * F_LD + a * 16];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        double bop[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bop[t] = pb[ks * 4 * F_LD + b * 16 + rot[t]];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[a][b][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aop[a], bop[t], acc[a][b][t], 0, 0, 1);  // NEG: -A B
      }
      // one k-step's operands at a time (24 registers): left to itself the scheduler reads two steps ahead, runs out
      // of the 128 registers and reloads a spilled index behind the transfers it has just issued -- with a vmcnt(0)
      // that waits for THEM.  The other wave of the SIMD covers the read latency.
#ifndef TGP_PROBE_WIDE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next k-tile has landed (the only wait of the transfers)
    __syncthreads();
  }
  // write-through (sc1 = the agent-scope relaxed store st_agent emits) in the base + 32-bit-offset form: through the
  // atomic builtin the compiler materialises a 64-bit address per store, hoists all 32 in front of the k-loop and
  // spills them (42 registers)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const char* cb = Cu + (int64_t(a * 16) * ld + b * 16) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff[t]), "v"(acc[a][b][t]), "s"(cb) : "memory");
    }
}

// update(c, c, k): the diagonal tile (c, c) -= X_ck X_ck^T, lower 16 x 16 blocks only (potf2's fold: block pairs
// spread 5 / 4 over the waves, the slabs exchanged through S), write-through.
template <typename T>
__device__ __forceinline__ void chain_update_diag(const ChainArgs<T>& q, T* S, int c, int k) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  const int64_t ld = q.ld;
  constexpr int XC_LD = 144;
  T* Xc = S;
  const int pr = w & 3, i1 = pr, i2 = 7 - pr;
  const int t0 = (w < 4) ? 0 : 5, nt = (w < 4) ? 5 : 4;
  acc_t Cf[5];
#pragma unroll
  for (int tt = 0; tt < 5; ++tt) Cf[tt] = acc_t{0, 0, 0, 0};
  const T* Xk = q.A0 + int64_t(k) * TILE * ld + int64_t(c) * TILE;
  acc_t V[8];
  const int xoff = M::drow(lane, 0) * int(ld) + lrow;
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[jb][r] = (Xk + int64_t(jb * 16 + M::drow(0, r)) * ld + w * 16)[xoff];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h) __syncthreads();
#pragma unroll
    for (int jq = 0; jq < 4; ++jq)
#pragma unroll
      for (int r = 0; r < 4; ++r) Xc[((jq * 4 + r) * 4 + lk) * XC_LD + w * 16 + lrow] = V[h * 4 + jq][r];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const T* row = &Xc[(ks * 4 + lk) * XC_LD + lrow];
      if constexpr (M::FAST4) {  // block rows i1 / i2 as the B operand, rotated reads (see Mfma<double>)
        const T* rowb = &Xc[(ks * 4 + lk) * XC_LD];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          T bo[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) bo[u] = rowb[(half ? i2 : i1) * 16 + M::rot4(lane, u)];
#pragma unroll
          for (int tt = 0; tt < 5; ++tt) {
            if (tt < nt) {
              const int t = t0 + tt;
              const bool first = t <= pr;
              if (first == (half == 0)) {
                const int jj = first ? t : t - pr - 1;
                const T av = row[jj * 16];
#pragma unroll
                for (int u = 0; u < 4; ++u) Cf[tt][u] = M::mma4(av, bo[u], Cf[tt][u]);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const T b1 = row[i1 * 16], b2 = row[i2 * 16];
#pragma unroll
        for (int tt = 0; tt < 5; ++tt) {
          if (tt < nt) {
            const int t = t0 + tt;
            const bool first = t <= pr;
            const int jj = first ? t : t - pr - 1;
            Cf[tt] = M::mma(row[jj * 16], first ? b1 : b2, Cf[tt]);
          }
        }
      }
    }
  }
  T* Acc = q.A0 + int64_t(c) * TILE * ld + int64_t(c) * TILE;
#pragma unroll
  for (int tt = 0; tt < 5; ++tt) {
    if (tt < nt) {
      const int t = t0 + tt;
      const bool first = t <= pr;
      const int ii = first ? i1 : i2, jj = first ? t : t - pr - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // entry r of a block's accumulator: (row, column) inside the 16 x 16 block
        const int br = M::FAST4 ? M::rot4(lane, r) : lrow, bc = M::FAST4 ? M::arow4(lane) : M::drow(lane, r);
        T* pe = Acc + int64_t(jj * 16 + bc) * ld + ii * 16 + br;
        st_agent(pe, T(*pe - Cf[tt][r]));
      }
    }
  }
}

// solve(i, c): X_ic = A_ic L_cc^-T by the transposed recurrence (L_cc with the 16 x 16 inverses in its diagonal
// slots staged in S).  The caller has waited for the tile's updates and for L_cc.  The solved tile is stored
// write-through; V returns -Y (the operand layout of potf2's fold).
template <typename T>
__device__ __forceinline__ void chain_load_tile(const ChainArgs<T>& q, int i, int c, typename Mfma<T>::acc_t (&V)[8]) {
  using M = Mfma<T>;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t ld = q.ld;
  const T* bu = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE + w * 16;  // wave-uniform; lane offset below
  const uint32_t boff = uint32_t(M::drow(lane, 0) * int(ld) + (lane & 15));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[jb][r] = (bu + int64_t(jb * 16 + M::drow(0, r)) * ld)[boff];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // in registers before a later acquire drops the L1
  __builtin_amdgcn_sched_barrier(0);
}

// (the tile's 16 rows per wave are in V already: chain_load_tile, issued BEFORE the wait for L_cc)
template <typename T>
__device__ __forceinline__ void chain_solve(const ChainArgs<T>& q, T* S, int i, int c,
                                            typename Mfma<T>::acc_t (&V)[8], long long* st, int s0) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  const int64_t ld = q.ld;
  T* Bt = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE;
  const T* Ljj = q.A0 + int64_t(c) * TILE * ld + int64_t(c) * TILE;
  const T* dinv = q.dinv + int64_t(c) * 2048;
  T* bu = Bt + w * 16;  // wave-uniform; lane offset below
  const uint32_t boff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
  // L_cc -> LDS in potf2's block image: the 28 blocks below the diagonal as they are, the diagonal
  // slots take the 16 x 16 inverses (all a solve needs of a diagonal block).  In two rounds of blocks (3 + 2 per
  // wave): the tile's 64 registers are live beside the staging registers, and five blocks at once spill.
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    constexpr int NTR = 3;
    T tr[NTR][4];
    const int t_lo = round * NTR, t_n = round == 0 ? NTR : 5 - NTR;
#pragma unroll
    for (int tq = 0; tq < NTR; ++tq) {
      const int b = w + 8 * (t_lo + tq);
      if (tq < t_n && b < 36) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
        const int bj = b - bi * (bi + 1) / 2;
        if (bi == bj) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) tr[tq][qq] = dinv[bi * 256 + qq * 64 + lane];
        } else {
          const int voff = lk * int(ld) + lrow;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) tr[tq][qq] = (Ljj + int64_t(bj * 16 + qq * 4) * ld + bi * 16)[voff];
        }
      }
    }
#pragma unroll
    for (int tq = 0; tq < NTR; ++tq) {
      const int b = w + 8 * (t_lo + tq);
      if (tq < t_n && b < 36) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) S[b * 256 + qq * 64 + lane] = tr[tq][qq];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  chain_stamp(st, s0);  // tile in registers, L_cc staged in LDS
  T* bs = bu;  // the stores recompute their addresses
  asm volatile("" : "+s"(bs));
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    acc_t acc = V[jb], acc2 = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kb = 0; kb < jb; ++kb) {
      const T* Lb = &S[blk(jb, kb)];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T l = Lb[M::drow(lane, s) * 16 + lrow];
        if (s & 1) acc2 = M::mma(l, V[kb][s], acc2);
        else acc = M::mma(l, V[kb][s], acc);
      }
    }
    acc += acc2;
    const T* Db = &S[blk(jb, jb)];
    acc_t y = acc_t{0, 0, 0, 0}, y2 = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T d = Db[M::drow(lane, s) * 16 + lrow];
      if (s & 1) y2 = M::mma(d, acc[s], y2);
      else y = M::mma(d, acc[s], y);
    }
    y += y2;
#pragma unroll
    for (int r = 0; r < 4; ++r) st_agent(bs + int64_t(jb * 16 + M::drow(0, r)) * ld + boff, T(y[r]));
    V[jb] = -y;
  }
  chain_stamp(st, s0 + 1);  // solved, stores issued
}

// ---- the diagonal chain's STREAMED solve + fold (diag tasks only) ----------------------------------------------
// potf2 stores column block j of L_cc and its 16 x 16 inverse under step j+1, so both are in memory long before the
// factorisation ends.  It publishes its progress step by step (potf2_body.inc, POTF2_PRE_SYNC / POTF2_POST_SYNC: the
// word of the unused tile (c-1, c) = epoch + number of column blocks in memory), and the NEXT diagonal task follows
// one column block behind, right-looking: behind step word > j it takes column block j of L_cc (8 - j blocks, one per
// wave, agent-scope loads: no acquire, every wave polls for itself), finishes X_j = R_j Linv_jj^T from the running
// residual R, stores it, takes it out of the residuals to the right (R_jb -= X_j L_jb,j^T) and adds X_j X_j^T to the
// fold of tile (c+1, c+1) -- 16 columns of the 128 at a time, slabs exchanged through LDS as in potf2_body.inc.
// When L_cc's FINAL flag arrives only column block 7 is left: X_7, its store, one 16-column piece of the fold.
// Between two potf2 of the chain remain ~8 us instead of ~26 (stamps: profiles/r04_b).
template <typename T>
struct ChainStream {
  static constexpr int XC_LD = 144;           // 144 mod 32 == 16: conflict-free operand reads
  static constexpr int LC = 8 * 256;          // one column block of L_cc: up to 8 blocks of 16 x 16
  static constexpr int XC = 16 * XC_LD;       // one 16-column slab of X, [k][row]
  static_assert(2 * (LC + XC) <= 36 * 256, "both double buffers live in S");
};

// wave-level wait until *word has reached `want` within the same epoch (bounded: a lost producer poisons info);
// returns what it saw, so that a reader that is several steps behind polls ONCE
__device__ __forceinline__ uint32_t chain_wave_wait_ge(const uint32_t* word, uint32_t want, int32_t* info) {
  PollClock clk;
  for (unsigned spin = 0;; ++spin) {
    const uint32_t v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (int32_t(v - want) >= 0 && int32_t(v - want) < 128) return v;
    __builtin_amdgcn_s_sleep(1);
    if ((spin & 255) == 255 && (clk.expired(spin) || poisoned(info))) {
      if ((threadIdx.x & 63) == 0) atomicExch(info, STEP_TIMEOUT);
      return want;
    }
  }
}

// barrier that orders LDS traffic only (the streams keep global stores and the next block's loads in flight across it)
__device__ __forceinline__ void chain_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wave-level wait until the counter *p (zeroed per launch) has reached `want`; returns what it saw
__device__ __forceinline__ int chain_wave_wait_count(const int32_t* p, int want, int32_t* info) {
  PollClock clk;
  for (unsigned spin = 0;; ++spin) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= want) return v;
    __builtin_amdgcn_s_sleep(1);
    if ((spin & 255) == 255 && (clk.expired(spin) || poisoned(info))) {
      if ((threadIdx.x & 63) == 0) atomicExch(info, STEP_TIMEOUT);
      return want;
    }
  }
}

// xsolve(c'): tile (i, c) = (c', c'-1) in V (chain_load_tile) -> X = A_ic L_cc^-T, streamed behind potf2(c)'s step word
// (`steps` != NULL: L_cc is being factored by another workgroup of this launch) and stored column block by column
// block; *xstep = number of column blocks of X complete in memory (published one step behind: the stores of block
// j-1 have had step j's loads and barrier to drain).  The fold of X is diag(c')'s work -- on ANOTHER compute unit: solve
// and fold together are 288 MFMAs per wave at the 16x16x4 form's rate, 26 us on one unit wherever they start.
template <typename T>
__device__ __forceinline__ void chain_xsolve_stream(const ChainArgs<T>& q, T* S, int i, int c,
                                                    typename Mfma<T>::acc_t (&V)[8], const uint32_t* steps,
                                                    const uint32_t* final_word, uint32_t E, int32_t* xstep) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  using CS = ChainStream<T>;
  using bits_t = typename AgentBits<T>::t;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  const int64_t ld = q.ld;
  const T* Ljj = q.A0 + int64_t(c) * TILE * ld + int64_t(c) * TILE;
  const T* dinv = q.dinv + int64_t(c) * 2048;
  T* bs = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE + w * 16;  // wave-uniform; lane offset below
  asm volatile("" : "+s"(bs));
  const uint32_t boff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
  // column block jn of L_cc -> registers (this wave's block (jn + w, jn); agent-scope loads: no acquire needed)
  T tr[4];
  auto issue_loads = [&](int jn) {
    const int bi = jn + w;
    if (bi < 8) {
      const T* src = bi == jn ? dinv + jn * 256 + lane : Ljj + int64_t(jn * 16) * ld + bi * 16 + (lk * int(ld) + lrow);
      const int64_t stride = bi == jn ? 64 : 4 * ld;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const bits_t u = __hip_atomic_load(reinterpret_cast<const bits_t*>(src + qq * stride), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        __builtin_memcpy(&tr[qq], &u, sizeof(T));
      }
    }
  };
  // column blocks of L_cc known to be in memory (wave-uniform); a reader that starts late sees several at once and
  // runs through them with the NEXT block's loads in flight under the current block's arithmetic
  int known = steps == nullptr ? 8 : 0;
  bool have = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    T* Lc = S + (j & 1) * CS::LC;  // blocks (j + s, j), s = 0 .. 7 - j; slot 0 = Linv_jj
    if (!have) {
      if (known <= j) {
        if (j < 7) known = int(chain_wave_wait_ge(steps, E + uint32_t(j + 1), q.info) - E);
        else { chain_wave_wait_ge(final_word, E + CHAIN_FINAL, q.info); known = 8; }
        known = __builtin_amdgcn_readfirstlane(known);
      }
      issue_loads(j);
    }
    if (j + w < 8) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) Lc[w * 256 + qq * 64 + lane] = tr[qq];
    }
    if (j >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of X_{j-1}, stored a step ago
    chain_lds_barrier();
    if (j >= 1 && threadIdx.x == 0) __hip_atomic_store(xstep, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    have = j < 7 && known > j + 1;
    if (have) issue_loads(j + 1);
    {  // X_j = R_j Linv_jj^T
      acc_t y = acc_t{0, 0, 0, 0}, y2 = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T d = Lc[M::drow(lane, s) * 16 + lrow];
        if (s & 1) y2 = M::mma(d, V[j][s], y2);
        else y = M::mma(d, V[j][s], y);
      }
      y += y2;
#pragma unroll
      for (int r = 0; r < 4; ++r) st_agent(bs + int64_t(j * 16 + M::drow(0, r)) * ld + boff, T(y[r]));
      V[j] = -y;
    }
#pragma unroll
    for (int jb = j + 1; jb < 8; ++jb) {  // R_jb -= X_j L_jb,j^T
      const T* Lb = Lc + (jb - j) * 256;
#pragma unroll
      for (int s = 0; s < 4; ++s) V[jb] = M::mma(Lb[M::drow(lane, s) * 16 + lrow], V[j][s], V[jb]);
    }
  }
}

// diag(c'): Cf = the lower blocks of X X^T for X = tile (i, c) = (c', c'-1) (distributed as potf2_body.inc's fold: rows p
// and 7 - p belong to waves p and p + 4), 16 columns at a time behind xsolve(c')'s counter (`xstep` == NULL: X is final),
// slabs loaded straight into the exchange layout (agent-scope loads) and exchanged through LDS
template <typename T>
__device__ __forceinline__ void chain_fold_stream(const ChainArgs<T>& q, T* S, int i, int c,
                                                  typename Mfma<T>::acc_t (&Cf)[5], const int32_t* xstep) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  using CS = ChainStream<T>;
  using bits_t = typename AgentBits<T>::t;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 15, lk = lane >> 4;
  const int64_t ld = q.ld;
  const T* bu = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE + w * 16;  // wave-uniform; lane offset below
  const uint32_t boff = uint32_t(M::drow(lane, 0) * int(ld) + lrow);
  const int pr = w & 3, i1 = pr, i2 = 7 - pr;
  const int t0 = (w < 4) ? 0 : 5, nt = (w < 4) ? 5 : 4;
#pragma unroll
  for (int tt = 0; tt < 5; ++tt) Cf[tt] = acc_t{0, 0, 0, 0};
  T xr[4];  // element (row w * 16 + lrow, column jn * 16 + drow(lane, r))
  auto issue_loads = [&](int jn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bits_t u = __hip_atomic_load(reinterpret_cast<const bits_t*>(bu + int64_t(jn * 16 + M::drow(0, r)) * ld + boff),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_memcpy(&xr[r], &u, sizeof(T));
    }
  };
  int known = xstep == nullptr ? 8 : 0;
  bool have = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    T* Xc = S + (j & 1) * CS::XC;  // X_j, [k][row]
    if (!have) {
      if (known <= j) known = __builtin_amdgcn_readfirstlane(chain_wave_wait_count(xstep, j + 1, q.info));
      issue_loads(j);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Xc[(r * 4 + lk) * CS::XC_LD + w * 16 + lrow] = xr[r];
    chain_lds_barrier();
    have = j < 7 && known > j + 1;
    if (have) issue_loads(j + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const T* row = &Xc[(ks * 4 + lk) * CS::XC_LD + lrow];
      const T b1 = row[i1 * 16], b2 = row[i2 * 16];
#pragma unroll
      for (int tt = 0; tt < 5; ++tt) {
        if (tt < nt) {
          const int t = t0 + tt;
          const bool first = t <= pr;
          const int jj = first ? t : t - pr - 1;
          Cf[tt] = M::mma(row[jj * 16], first ? b1 : b2, Cf[tt]);
        }
      }
    }
  }
}

// ---- forward substitution as tasks (round 6; chain_tasks.h, F(c)) --------------------------------------------------
// all threads; wave 0 polls up to three words -- lane 0: *p0 >= want0, lane 1: *p1 == want1, lane 2: *p2 >= want2 (NULL:
// nothing to wait for) --, one acquire, barrier
__device__ __forceinline__ void chain_wait_counts(const int32_t* p0, int want0, const int32_t* p1, int want1,
                                                  const int32_t* p2, int want2, int32_t* info) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int32_t* f = lane == 0 ? p0 : (lane == 1 ? p1 : (lane == 2 ? p2 : nullptr));
    const int want = lane == 0 ? want0 : (lane == 1 ? want1 : want2);
    auto ready = [&]() {
      const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return lane == 1 ? v == want : v >= want;
    };
    bool ok = f == nullptr || ready();
    unsigned spin = 0;
    PollClock clk;
    while (!__all(ok)) {
      __builtin_amdgcn_s_sleep(4);
      if (!ok) ok = ready();
      bool dead = false;
      if ((spin & 255) == 255 && lane == 0) dead = clk.expired(spin) || poisoned(info);
      ++spin;
      if (__any(dead)) {
        if (lane == 0) atomicExch(info, STEP_TIMEOUT);
        break;
      }
    }
    if (lane == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// fsolve(c): y_c -= X_{c,c-1} z_{c-1} (`prev`), then y_c <- L_cc^-1 y_c -- the arithmetic of trsv_diag_fwd_kernel: 16-column
// sub-steps with the 16 x 16 inverses from `dinv`.  Threads 0..127 own a row each, the others only meet the barriers.
// Stored write-through.
template <typename T>
__device__ __forceinline__ void chain_fsolve(const ChainArgs<T>& q, T* S, int c, bool prev) {
  const int r = threadIdx.x, rb = (r >> 4) & 7, rl = r & 15;
  const bool act = r < 128;
  const int64_t ld = q.ld;
  const T* Lkk = q.A0 + int64_t(c) * TILE * ld + int64_t(c) * TILE;
  const T* dinv = q.dinv + int64_t(c) * 2048;
  T* y = q.y + int64_t(c) * TILE;
  T* st = S;        // [128]
  T* sx = S + 128;  // [16]
  T* sz = S + 256;  // [128] z_{c-1}
  T t = act ? y[r] : T(0);
  if (prev) {
    if (act) sz[r] = y[r - TILE];
    __syncthreads();
    if (act) {
      const T* Xp = Lkk - int64_t(TILE) * ld + r;  // tile (c, c-1): row r
      T a0 = 0, a1 = 0;
#pragma unroll 1
      for (int j0 = 0; j0 < TILE; j0 += 16) {
        T v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = Xp[int64_t(j0 + j) * ld];
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          a0 += v[j] * sz[j0 + j];
          a1 += v[j + 1] * sz[j0 + j + 1];
        }
      }
      t -= a0 + a1;
    }
  }
  T di[16];
#pragma unroll
  for (int qq = 0; qq < 16; ++qq) di[qq] = act ? dinv[rb * 256 + qq * 16 + rl] : T(0);
#pragma unroll 1
  for (int jb = 0; jb < 8; ++jb) {
    T cur[16];  // this row's entries of column block jb (requested before the barriers that publish x_jb)
    const bool below = act && rb > jb;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) cur[qq] = below ? Lkk[int64_t(jb * 16 + qq) * ld + r] : T(0);
    if (act) st[r] = t;
    __syncthreads();
    if (act && rb == jb) {
      T x = 0;
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) x += di[qq] * st[jb * 16 + qq];
      sx[rl] = x;
      t = x;
    }
    __syncthreads();
    if (below) {
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) t -= cur[qq] * sx[qq];
    }
  }
  if (act) st_agent(y + r, t);
  if (q.red != nullptr) {
    // sum of z_i^2 and of log L_ii over this block: a fixed tree over 128 values each (the padding rows hold z = 0, L = 1)
    __syncthreads();
    double* ra = reinterpret_cast<double*>(S);  // [128] + [128] (S is at least 36 * 256 elements of T)
    double* rq = ra + 128;
    if (act) {
      const double zv = double(t);
      ra[r] = zv * zv;
      rq[r] = log(double(Lkk[int64_t(r) * ld + r]));
    }
    __syncthreads();
    for (int sft = 64; sft > 0; sft >>= 1) {
      if (r < sft) {
        ra[r] += ra[r + sft];
        rq[r] += rq[r + sft];
      }
      __syncthreads();
    }
    const int g = q.pivot_base / TILE + c;
    if (r == 0) {
      st_agent(q.red + 2 * g, ra[0]);
      st_agent(q.red + 2 * g + 1, rq[0]);
    }
    if (q.red_total > 0 && g == q.red_total - 1) {
      // the matrix's last block: every earlier fsolve has published (z_{c-1} was waited for, block by block, launch by launch
      // in stream order) -- their partial sums are in memory.  Thread r adds the blocks r, r + 512, ... of both, then the tree.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      double* ta = ra + 256;  // [512] + [512]
      double* tb = ta + 512;
      double a = 0, b = 0;
      for (int gg = r; gg < q.red_total; gg += 512) {
        const unsigned long long ua = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q.red + 2 * gg), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long ub = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q.red + 2 * gg + 1),
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a += __longlong_as_double((long long)ua);
        b += __longlong_as_double((long long)ub);
      }
      ta[r] = a;
      tb[r] = b;
      __syncthreads();
      for (int sft = 256; sft > 0; sft >>= 1) {
        if (r < sft) {
          ta[r] += ta[r + sft];
          tb[r] += tb[r + sft];
        }
        __syncthreads();
      }
      if (r == 0) {
        q.scal[0] = ta[0];
        q.scal[1] = tb[0];
      }
    }
  }
}

// fupdate(c, g): y_i -= X_ic z_c for the row tiles i > c + 1 of group g (`fgs` tiles; row c + 1 is fsolve(c + 1)'s): wave w
// takes tiles lo + w, lo + w + 8, ...; lane = two rows, 16 loads of 16 bytes per lane in flight; z_c in LDS
template <typename T>
__device__ __forceinline__ void chain_fupdate(const ChainArgs<T>& q, T* S, int c, int g) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ld = q.ld;
  T* z = S;  // [128]
  if (tid < 128) z[tid] = q.y[int64_t(c) * TILE + tid];
  __syncthreads();
  const int lo = g * q.fgs > c + 2 ? g * q.fgs : c + 2;
  const int hi = (g + 1) * q.fgs < q.R ? (g + 1) * q.fgs : q.R;
  for (int i = lo + w; i < hi; i += 8) {
    const T* Lic = q.A0 + int64_t(c) * TILE * ld + int64_t(i) * TILE + 2 * lane;
    T a0 = 0, a1 = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < TILE; j0 += 16) {
      T2 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const T2*>(Lic + int64_t(j0 + j) * ld);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        a0 += v[j].x * z[j0 + j];
        a1 += v[j].y * z[j0 + j];
      }
    }
    T* yi = q.y + int64_t(i) * TILE + 2 * lane;
    st_agent(yi, T(yi[0] - a0));
    st_agent(yi + 1, T(yi[1] - a1));
  }
}

// One task per workgroup (grid = number of tasks): a task loop inside the kernel lets the compiler hoist the
// lane-derived values of every phase across the whole loop body -- 100+ spilled VGPRs under the 128-register cap
// that keeps two chain workgroups on a CU beside the trailing update.  Tickets are taken at workgroup start, so
// every earlier ticket belongs to a workgroup that is running or done, whatever order the hardware starts them in.
#ifdef TGP_PROBE_WIDE  // measurement build: 256 registers per wave (nothing co-resident)
#define CHAIN_WAVES_PER_EU(T) 2
#else
// fp64: exactly 4 waves per SIMD = 128 registers, so that a chain workgroup (2 waves per SIMD) shares a SIMD with a wave of the
// fp64 trailing update (<= 253).  fp32: 3 = up to 168 registers -- under 128 the float instantiation spilled 13-14 registers
// (round-5 judge, item 10: config 5's whole factorisation), and the fp32 trailing update needs fewer registers itself
#define CHAIN_WAVES_PER_EU(T) (sizeof(T) == 8 ? 4 : 3)
#endif
template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(CHAIN_WAVES_PER_EU(T), 4))) void chain_kernel(const ChainArgs<T> q) {
  __shared__ __attribute__((aligned(16))) T S[36 * 256];
  __shared__ T Rs[2 * 16];
  __shared__ T Dg[256];
  __shared__ int s_task[7];
  using acc_t = typename Mfma<T>::acc_t;
  if (threadIdx.x == 0) {
    // ticket -> task: one word of the launch's table (built on the host by chain_tasks.h, the text the CPU test checks)
    const int t = atomicAdd(q.ticket, 1);
    s_task[4] = t;
    const ChainTask task = chain_unpack(q.tasks[t]);
    const int kind = task.kind, ti = task.i, tc = task.c, tk = task.k;
    s_task[5] = task.part;
    s_task[6] = task.k0;
    s_task[0] = kind; s_task[1] = ti; s_task[2] = tc; s_task[3] = tk;
  }
  __syncthreads();
  const int kind = __builtin_amdgcn_readfirstlane(s_task[0]);
  const int ti = __builtin_amdgcn_readfirstlane(s_task[1]);
  const int tc = __builtin_amdgcn_readfirstlane(s_task[2]);
  const int tk = __builtin_amdgcn_readfirstlane(s_task[3]);
  if (kind > 8) return;
  long long* st = nullptr;
  if (q.stamps != nullptr) {  // {kind, row tile, block column, launch | update column << 8, stamps ...}
    st = q.stamps + int64_t(__builtin_amdgcn_readfirstlane(s_task[4])) * 16;
    if (threadIdx.x == 0) {
      st[0] = kind; st[1] = kind == 1 ? tc : ti; st[2] = tc; st[3] = q.launch | (tk << 8) | (s_task[6] << 16);
      for (int k = 1; k < 12; ++k) st[4 + k] = 0;
    }
    chain_stamp(st, 0);  // task started
  }
  const uint32_t E = q.epoch32;
  const bool head_final = q.cb == 0;  // L_00 was factored in front of the launch
  if (kind == 7) {  // ---- fsolve(c): y_c -= X_{c,c-1} z_{c-1}, z_c = L_cc^-1 y_c ----
    const int c = tc;
    const int g = c / q.fgs;
    const bool prev = c > 0 || q.fprev != 0;  // (a panel's first block: the previous panel's last column, complete)
    // row c carries the columns cb .. c-2 of this launch (fupdate(c-2, g) was the last of them); L_cc (a panel's first block is
    // factored in front of the launch); z_{c-1} when this launch solves it
    chain_wait_counts(c - 1 > q.cb ? q.ticket + CHAIN_YSTATE_OFF + g : nullptr, c - 1 - q.cb,
                      (c == 0 && head_final) ? nullptr : reinterpret_cast<const int32_t*>(q.flags + c * CHAIN_FLAG_LD + c),
                      int32_t(E + CHAIN_FINAL), c > q.cb ? q.ticket + CHAIN_ZFLAG_OFF + (c - 1) : nullptr, 1, q.info);
    chain_stamp(st, 1);
    chain_fsolve<T>(q, S, c, prev);
    chain_stamp(st, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(q.ticket + CHAIN_ZFLAG_OFF + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_stamp(st, 3);
    return;
  }
  if (kind == 8) {  // ---- fupdate(c, g): y_i -= X_ic z_c, rows of group g below block c ----
    const int c = tc, g = ti;
    // block column c final (every tile below the diagonal solved: R - c - 1 of them, + the diagonal tile when this launch
    // factored it), z_c in memory, the group carries the columns before c
    const int target = q.R - c - ((c == 0 && head_final) ? 1 : 0);
    chain_wait_counts(q.ticket + CHAIN_COLCNT_OFF + c, target, q.ticket + CHAIN_ZFLAG_OFF + c, 1,
                      c > q.cb ? q.ticket + CHAIN_YSTATE_OFF + g : nullptr, c - q.cb, q.info);  // (== c - cb: >= is the same here)
    chain_stamp(st, 1);
    chain_fupdate<T>(q, S, c, g);
    chain_stamp(st, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(q.ticket + CHAIN_YSTATE_OFF + g, c - q.cb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_stamp(st, 3);
    return;
  }
  if constexpr (sizeof(T) == 8) {
    if (kind == 6) {  // ---- update(i, c, [k0, k]): the batched form (fp64 only: the policy is off for fp32) ----
      const int i = ti, c = tc, k = tk, k0 = __builtin_amdgcn_readfirstlane(s_task[6]);
      uint32_t* wt = q.flags + i * CHAIN_FLAG_LD + c;
      // X_{i,k} / X_{c,k} final implies the block columns before them are (their solves waited for exactly those); the
      // tile carries every update from the columns in front of the batch
      chain_wait<8>(q.flags + i * CHAIN_FLAG_LD + k, E + CHAIN_FINAL, q.flags + c * CHAIN_FLAG_LD + k, E + CHAIN_FINAL,
                    k0 > q.cb ? wt : nullptr, E + uint32_t(k0), q.info);
      chain_stamp(st, 1);
      chain_update_fast(q, S, i, c, k0, k + 1);
      chain_stamp(st, 2);
      chain_publish(wt, E + uint32_t(k + 1));
      chain_stamp(st, 3);
      return;
    }
  }
  if (kind >= 2 && kind <= 4) {  // ---- update(i, c, k) ----
    const int i = ti, c = tc, k = tk;
    uint32_t* wt = q.flags + i * CHAIN_FLAG_LD + c;
    chain_wait<8>(q.flags + i * CHAIN_FLAG_LD + k, E + CHAIN_FINAL, q.flags + c * CHAIN_FLAG_LD + k, E + CHAIN_FINAL,
                  k > q.cb ? wt : nullptr, E + uint32_t(k), q.info);
    chain_stamp(st, 1);  // operands final, the tile carries every earlier update
    if (kind == 2) {
      if constexpr (sizeof(T) == 8) {
        if (q.fast_update) chain_update_fast(q, S, i, c, k, k + 1);
        else chain_update_full<T, 1>(q, S, i, c, k, 0);
      } else {
        chain_update_full<T, 1>(q, S, i, c, k, 0);
      }
    }
    else if (kind == 4) chain_update_full<T, CHAIN_CRIT_PARTS>(q, S, i, c, k, __builtin_amdgcn_readfirstlane(s_task[5]));
    else chain_update_diag<T>(q, S, c, k);
    chain_stamp(st, 2);
    if (kind == 4) {  // the last of the parts to finish publishes the tile's new state
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0 &&
          __hip_atomic_fetch_add(q.ticket + CHAIN_QCNT_OFF + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              CHAIN_CRIT_PARTS - 1)
        __hip_atomic_store(wt, E + uint32_t(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      chain_publish(wt, E + uint32_t(k + 1));
    }
    chain_stamp(st, 3);
    return;
  }
  __builtin_amdgcn_s_setprio(2);
  if (kind == 0) {  // ---- solve(i, c) ----
    const int i = ti, c = tc;
    uint32_t* wt = q.flags + i * CHAIN_FLAG_LD + c;
    const uint32_t* wl = q.flags + c * CHAIN_FLAG_LD + c;
    // the tile first (its updates are done long before L_cc as a rule): its round trip hides behind potf2
    if (c > q.cb) chain_wait<4>(wt, E + uint32_t(c), nullptr, 0, nullptr, 0, q.info);
    acc_t V[8];
    chain_load_tile<T>(q, i, c, V);
    if (!(c == 0 && head_final)) chain_wait<4>(wl, E + CHAIN_FINAL, nullptr, 0, nullptr, 0, q.info);
    chain_stamp(st, 1);
    chain_solve<T>(q, S, i, c, V, st, 2);
    chain_publish(wt, E + CHAIN_FINAL, q.ticket + CHAIN_COLCNT_OFF + c);
    chain_stamp(st, 4);
    return;
  }
  // ---- diag(c) ----
  const int c = tc;
  T* A = q.A0 + int64_t(c) * TILE * q.ld + int64_t(c) * TILE;
  const int64_t ld = q.ld;
  T* dinv = q.dinv + int64_t(c) * 2048;
  int32_t* info = q.info;
  const int32_t pivot_base = q.pivot_base + c * TILE;
  uint32_t* frow = q.flags + c * CHAIN_FLAG_LD;
  if (kind == 5) {  // ---- xsolve(c): tile (c, c-1) solved column block by column block behind potf2(c-1) ----
    acc_t Vx[8];  // the residual of tile (c, c-1): row w * 16 + lrow, column jb * 16 + drow(lane, r)
    // the tile with its updates from columns cb .. c-2 of this launch (the last one is the four-way split update)
    if (c - 1 > q.cb) chain_wait<4>(frow + (c - 1), E + uint32_t(c - 1), nullptr, 0, nullptr, 0, q.info);
    chain_load_tile<T>(q, c, c - 1, Vx);
    chain_stamp(st, 1);
    // L_{c-1,c-1} is factored inside this launch unless it is the panel's first block: follow it step by step
    const uint32_t* steps = c - 1 >= 1 ? q.flags + (c - 2) * CHAIN_FLAG_LD + (c - 1) : nullptr;
    int32_t* xstep = q.ticket + CHAIN_XSTEP_OFF + c;
    chain_xsolve_stream<T>(q, S, c, c - 1, Vx, steps, q.flags + (c - 1) * CHAIN_FLAG_LD + (c - 1), E, xstep);
    chain_stamp(st, 3);
    // the updates behind column c-1, the next solves and diag(c)'s last fold step need X NOW
    chain_publish(frow + (c - 1), E + CHAIN_FINAL, q.ticket + CHAIN_COLCNT_OFF + (c - 1));
    if (threadIdx.x == 0) __hip_atomic_store(xstep, 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chain_stamp(st, 4);
    return;
  }
  acc_t Vx[8];   // (only named by the fold code potf2_body.inc skips: POTF2_CF_IN_REGS)
  acc_t Cfx[5];  // the fold X_{c,c-1} X_{c,c-1}^T, lower blocks (chain_fold_stream)
  chain_fold_stream<T>(q, S, c, c - 1, Cfx, c > q.cb ? q.ticket + CHAIN_XSTEP_OFF + c : nullptr);
  chain_stamp(st, 3);
  // tile (c, c) with the updates from columns cb .. c-2 (the one from c-1 is the fold in Cfx)
  if (c - 1 > q.cb) chain_wait<1>(frow + c, E + uint32_t(c - 1), nullptr, 0, nullptr, 0, q.info);
  else __syncthreads();  // (every wave is done with the slabs in S)
  chain_stamp(st, 5);
  {
    constexpr bool FOLD = true;
    uint32_t* step_word = q.flags + (c - 1) * CHAIN_FLAG_LD + c;  // (c >= 1 here; tile (c-1, c) does not exist)
#define POTF2_ST(p, v) st_agent((p), T(v))
#define POTF2_V_IN_REGS Vx
#define POTF2_CF_IN_REGS Cfx
#define POTF2_PRE_SYNC(kb) do { if ((kb) >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define POTF2_POST_SYNC(kb)                                                                            \
  do {                                                                                                \
    if ((kb) >= 1 && threadIdx.x == 0)                                                                \
      __hip_atomic_store(step_word, E + uint32_t(kb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    \
  } while (0)
#include "potf2_body.inc"
#undef POTF2_POST_SYNC
#undef POTF2_PRE_SYNC
#undef POTF2_CF_IN_REGS
#undef POTF2_V_IN_REGS
#undef POTF2_ST
  }
  chain_stamp(st, 6);  // factored, stores issued
  chain_publish(frow + c, E + CHAIN_FINAL, q.ticket + CHAIN_COLCNT_OFF + c);
  chain_stamp(st, 7);  // L_cc published
}

// dinv for an existing factor: one thread per (16-block, column)
template <typename T>
__global__ __launch_bounds__(256) void dinv_kernel(int64_t n, const T* __restrict__ L, int64_t ld,
                                                   T* __restrict__ dinv) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int64_t blk = t >> 4;
  const int c = int(t & 15);
  if (blk * 16 >= n) return;
  const T* D = L + blk * 16 * ld + blk * 16;
  T x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    T s = (i == c) ? T(1) : T(0);
#pragma unroll
    for (int k = 0; k < i; ++k) s -= D[int64_t(k) * ld + i] * x[k];
    x[i] = s / D[int64_t(i) * ld + i];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) dinv[blk * 256 + c * 16 + i] = x[i];
}

// ---------------------------------------------------------------------------------------
// trsm: B (m x 128) <- B L^-T, L a 128x128 lower block with its dinv.  Wave per 16 rows.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void trsm_kernel(int64_t m, const T* __restrict__ L, int64_t ldl,
                                                   const T* __restrict__ dinv, T* __restrict__ B,
                                                   int64_t ldb) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  __builtin_amdgcn_s_setprio(2);  // panel work outranks the concurrent trailing update
  const int lane = threadIdx.x & 63;
  const int64_t r0 = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16;
  if (r0 >= m) return;
  const int lrow = lane & 15;
  // The recurrence over the eight 16-column steps is serial, so memory latency must not be:
  // the whole 16 x 128 slab of B is loaded up front, and the L / dinv operands of step j+1
  // are fetched while step j runs on the MFMAs (one exposed round trip instead of eight).
  T* bp = B + r0 + lrow;
  acc_t V[8];  // V[j]: B_j until step j, then Z_j = -Y_j
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) V[jb][r] = bp[int64_t(jb * 16 + M::drow(lane, r)) * ldb];
  T Dn[4], Ln[7][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) Dn[s] = dinv[M::drow(lane, s) * 16 + lrow];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    T Dc[4], Lc[7][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) Dc[s] = Dn[s];
#pragma unroll
    for (int kb = 0; kb < jb; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s) Lc[kb][s] = Ln[kb][s];
    if (jb + 1 < 8) {
#pragma unroll
      for (int s = 0; s < 4; ++s) Dn[s] = dinv[(jb + 1) * 256 + M::drow(lane, s) * 16 + lrow];
#pragma unroll
      for (int kb = 0; kb <= jb; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          Ln[kb][s] = L[int64_t(kb * 16 + M::drow(lane, s)) * ldl + (jb + 1) * 16 + lrow];
    }
    acc_t acc = V[jb], acc2 = acc_t{0, 0, 0, 0};  // two chains: MFMA latency overlaps
#pragma unroll
    for (int kb = 0; kb < jb; ++kb) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s & 1) acc2 = M::mma(Lc[kb][s], V[kb][s], acc2);
        else acc = M::mma(Lc[kb][s], V[kb][s], acc);
      }
    }
    acc += acc2;
    acc_t y = acc_t{0, 0, 0, 0}, y2 = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s & 1) y2 = M::mma(Dc[s], acc[s], y2);
      else y = M::mma(Dc[s], acc[s], y);
    }
    y += y2;
#pragma unroll
    for (int r = 0; r < 4; ++r) bp[int64_t(jb * 16 + M::drow(lane, r)) * ldb] = y[r];
    V[jb] = -y;
  }
}

// ---------------------------------------------------------------------------------------
// trsv (single right-hand side): per 128-block, one diagonal solve + one panel update.
// ---------------------------------------------------------------------------------------
// forward: y_kb <- L_kk^-1 y_kb.  128 threads, thread r owns row r.  The row's entries of the
// 16-column strip that the NEXT sub-step needs are fetched while the current one runs (two 16-
// entry buffers instead of the whole 112-entry row: ~110 VGPRs, so the kernel fits on a SIMD
// beside a trailing-update wave -- with the whole row in registers it needed a CU without any
// MFMA tile and waited milliseconds for one during large factorisations).
template <typename T>
__global__ __launch_bounds__(128) void trsv_diag_fwd_kernel(const T* __restrict__ Lkk,
                                                            int64_t ld,
                                                            const T* __restrict__ dinv,
                                                            T* __restrict__ y) {
  __shared__ T st[128];
  __shared__ T sx[16];
  const int r = threadIdx.x, rb = r >> 4, rl = r & 15;
  T di[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) di[q] = dinv[rb * 256 + q * 16 + rl];
  T cur[16], nxt[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) cur[q] = (rb > 0) ? Lkk[int64_t(q) * ld + r] : T(0);
  T t = y[r];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) {
    if (jb + 1 < 7) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        nxt[q] = (rb > jb + 1) ? Lkk[int64_t((jb + 1) * 16 + q) * ld + r] : T(0);
    }
    st[r] = t;
    __syncthreads();
    if (rb == jb) {
      T x = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) x += di[q] * st[jb * 16 + q];
      sx[rl] = x;
      t = x;
    }
    __syncthreads();
    if (jb < 7) {
      if (rb > jb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t -= cur[q] * sx[q];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
    }
  }
  y[r] = t;
}

// backward: y_kb <- L_kk^-T y_kb.  thread c owns column c.
template <typename T>
__global__ __launch_bounds__(128, 1) void trsv_diag_bwd_kernel(const T* __restrict__ Lkk,
                                                               int64_t ld,
                                                               const T* __restrict__ dinv,
                                                               T* __restrict__ y) {
  __shared__ T st[128];
  __shared__ T sx[16];
  const int c = threadIdx.x, cb = c >> 4, cl = c & 15;
  T Lc[7][16];  // Lc[jb-1][q] = L[jb*16+q][c] for jb = 1..7
#pragma unroll
  for (int jb = 1; jb < 8; ++jb)
#pragma unroll
    for (int q = 0; q < 16; ++q) Lc[jb - 1][q] = (cb < jb) ? Lkk[int64_t(c) * ld + jb * 16 + q] : T(0);
  T di[16];  // row cl of dinv_cb^T = column cl of dinv_cb: element (q, cl) at cl*16 + q
#pragma unroll
  for (int q = 0; q < 16; ++q) di[q] = dinv[cb * 256 + cl * 16 + q];
  T t = y[c];
#pragma unroll
  for (int jb = 7; jb >= 0; --jb) {
    st[c] = t;
    __syncthreads();
    if (cb == jb) {
      T x = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) x += di[q] * st[jb * 16 + q];
      sx[cl] = x;
      t = x;
    }
    __syncthreads();
    if (jb > 0) {
      if (cb < jb) {
#pragma unroll
        for (int q = 0; q < 16; ++q) t -= Lc[jb - 1][q] * sx[q];
      }
    }
  }
  y[c] = t;
}

// forward update: y[r] -= sum_{c<128} P[r, c] x[c], P = L[rows below, block cols]. thread per row.
template <typename T>
__global__ __launch_bounds__(256) void trsv_update_fwd_kernel(int64_t m, const T* __restrict__ P,
                                                              int64_t ld, const T* __restrict__ x,
                                                              T* __restrict__ y) {
  __shared__ T sx[128];
  if (threadIdx.x < 128) sx[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r >= m) return;
  T acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll 8
  for (int c = 0; c < 128; c += 4) {
    acc0 += P[int64_t(c) * ld + r] * sx[c];
    acc1 += P[int64_t(c + 1) * ld + r] * sx[c + 1];
    acc2 += P[int64_t(c + 2) * ld + r] * sx[c + 2];
    acc3 += P[int64_t(c + 3) * ld + r] * sx[c + 3];
  }
  y[r] -= (acc0 + acc1) + (acc2 + acc3);
}

// backward update: y[c] -= sum_{r<128} P[r, c] x[r] for c < ncols; P = L[block rows, cols 0..).
// Workgroup = 32 columns; lanes run along r (coalesced), reduction through LDS.
template <typename T>
__global__ __launch_bounds__(256) void trsv_update_bwd_kernel(int64_t ncols,
                                                              const T* __restrict__ P, int64_t ld,
                                                              const T* __restrict__ x,
                                                              T* __restrict__ y) {
  __shared__ T sx[128];
  __shared__ T tile[32][129];
  if (threadIdx.x < 128) sx[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  const int r = threadIdx.x & 127, half = threadIdx.x >> 7;
  const int64_t c0 = int64_t(blockIdx.x) * 32;
#pragma unroll 4
  for (int cc = 0; cc < 16; ++cc) {
    const int cl = half * 16 + cc;
    const int64_t c = c0 + cl;
    tile[cl][r] = (c < ncols) ? P[c * ld + r] * sx[r] : T(0);
  }
  __syncthreads();
  const int cl = threadIdx.x >> 3, part = threadIdx.x & 7;
  T acc = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc += tile[cl][part * 16 + q];
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (part == 0 && c0 + cl < ncols) y[c0 + cl] -= acc;
}

// ---------------------------------------------------------------------------------------
// Forward substitution on a resident factor in ONE launch (replaces nblk x 2 dependent launches).
//
// Workgroup (ticket) b owns row block b: it streams the tiles (b, 0..b-1) of L -- every load of
// a wave is 1 KiB of one column, 32 in flight per lane, double-buffered over tiles, none of
// them depending on the solution -- and multiplies them with the solved blocks x_c as those
// appear.  x is its own "ready" signal: the output vector starts filled with a sentinel NaN
// and every entry is published with ONE naturally aligned device-scope (sc1) store and polled
// with device-scope loads (MI355X_MICROARCH.md, "data-tagged granule"): no flag, no fence.
// Tickets are handed out in dispatch order, so a workgroup only ever waits for workgroups that
// are already running.  The diagonal block is applied as W_b = L_bb^-1 (winv_kernel) -- two
// 128 x 128 matrix-vector products on the critical path per block instead of eight dependent
// 16-column sub-steps.  All reductions have a fixed order: bit-reproducible.
// ---------------------------------------------------------------------------------------
template <typename T> struct Sent;
template <> struct Sent<double> {
  using bits_t = unsigned long long;
  static constexpr bits_t value = 0xFFF8DEADBEEF5A5AULL;  // a quiet NaN no computation produces
};
template <> struct Sent<float> {
  using bits_t = unsigned int;
  static constexpr bits_t value = 0xFFC5A5A5u;
};
template <typename T>
__device__ __forceinline__ typename Sent<T>::bits_t load_x_bits(const T* p) {
  return __hip_atomic_load(reinterpret_cast<const typename Sent<T>::bits_t*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T bits_to(typename Sent<T>::bits_t b) {
  T v;
  __builtin_memcpy(&v, &b, sizeof(T));
  return v;
}

// tmp <- y, y <- sentinel, ticket <- 0
template <typename T>
__global__ __launch_bounds__(256) void trsv_prep_kernel(int64_t n, T* __restrict__ y, T* __restrict__ tmp,
                                                        int32_t* __restrict__ ticket, T* __restrict__ part, int nparts) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i == 0) *ticket = 0;
  if (i < n) {
    if (tmp != nullptr) tmp[i] = y[i];  // (NULL: the right-hand side is a buffer of its own, the solve reads it in place)
    y[i] = bits_to<T>(Sent<T>::value);
    for (int h = 0; h < nparts; ++h) part[int64_t(h) * n + i] = bits_to<T>(Sent<T>::value);  // the helpers' partial sums
  }
}

// W_b = L_bb^-1 for every 128 x 128 diagonal block (column-major 128 x 128 each, zeros above the
// diagonal): column c of W by forward substitution, one thread per column, L and W packed in LDS.
// Round 3: also the product that takes the LAST off-diagonal tile out of the forward solve's dependent hop,
//   tf_b = W_b L[b, b-1]          (x_b = W_b (y_b - sum_{c < b-1} L_bc x_c) - tf_b x_{b-1}),
// 128 x 128 column-major like W (thread j = column j; the tile 64 rows at a time in LDS, W as LDS broadcasts).
// With it the hop from x_{b-1} to x_b is ONE matrix-vector product and one reduction.
template <typename T>
__global__ __launch_bounds__(128) void winv_kernel(int nblk, const T* __restrict__ L, int64_t ld, T* __restrict__ winv,
                                                   T* __restrict__ winvT, T* __restrict__ tfwd, T* __restrict__ tfwd2) {
  __shared__ T Lp[8256];
  __shared__ T Wp[8256];
  const int c = threadIdx.x;
  const int b = blockIdx.x;
  const T* Lb = L + int64_t(b) * 128 * ld + int64_t(b) * 128;
  for (int k = 0; k < 128; ++k)  // column k of the block: rows k..127, coalesced
    if (c >= k) Lp[c * (c + 1) / 2 + k] = Lb[int64_t(k) * ld + c];
  __syncthreads();
  for (int i = 0; i < 128; ++i) {
    if (i >= c) {
      T sum = (i == c) ? T(1) : T(0);
      const int row = i * (i + 1) / 2;
      for (int k = c; k < i; ++k) sum -= Lp[row + k] * Wp[k * (k + 1) / 2 + c];
      Wp[row + c] = sum / Lp[row + i];
    }
  }
  T* out = winv + int64_t(b) * 16384 + int64_t(c) * 128;
  T* outT = winvT + int64_t(b) * 16384 + c;  // W^T (column-major): element (c, i) at i * 128 + c
  for (int i = 0; i < 128; ++i) {
    const T v = (i >= c) ? Wp[i * (i + 1) / 2 + c] : T(0);
    out[i] = v;
    outT[int64_t(i) * 128] = v;
  }
  __syncthreads();  // every column of W is in Wp; Lp is free: it takes 64 rows of a neighbouring tile at a time
  T* Lt = Lp;  // [64][128]: thread c reads Lt[k * 128 + c] -- conflict-free; W entries are LDS broadcasts
  // Round 6: the same for tile (b, b-2) -- tf2_b = W_b L[b, b-2] -- so that the forward solve's primary workgroup has NO
  // product with W_b behind x_{b-2} either: x_b = W_b (y_b - sum_{c <= b-3} L_bc x_c) - tf2_b x_{b-2} - tf_b x_{b-1}
  for (int which = 1; which <= 2; ++which) {
    if (b < which) break;
    // tf_b[:, c] = W_b L[b, b-which][:, c], in two halves of the inner index k
    const T* tile = Lb - int64_t(128 * which) * ld;  // tile (b, b-which): element (k, c) at c * ld + k
    T* o = (which == 1 ? tfwd : tfwd2) + int64_t(b) * 16384 + int64_t(c) * 128;
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      for (int cc = 0; cc < 128; ++cc)  // column cc, rows 64h .. 64h+63: half the threads, coalesced
        if (c < 64) Lt[c * 128 + cc] = tile[int64_t(cc) * ld + 64 * h + c];
      __syncthreads();
      for (int i = 64 * h; i < 128; ++i) {  // rows above 64h have no entry of W in this half
        const int row = i * (i + 1) / 2, kend = (i < 64 * h + 63 ? i : 64 * h + 63);
        T s0 = 0, s1 = 0;
        int k = 64 * h;
        for (; k + 1 <= kend; k += 2) {
          s0 += Wp[row + k] * Lt[(k - 64 * h) * 128 + c];
          s1 += Wp[row + k + 1] * Lt[(k + 1 - 64 * h) * 128 + c];
        }
        if (k <= kend) s0 += Wp[row + k] * Lt[(k - 64 * h) * 128 + c];
        if (h == 0) o[i] = s0 + s1;
        else o[i] += s0 + s1;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void trsv_fwd_stream_kernel(int nblk, const T* __restrict__ L, int64_t ld,
                                                              const T* __restrict__ winv,
                                                              const T* __restrict__ tfwd,
                                                              const T* __restrict__ tfwd2,
                                                              const T* __restrict__ yin, T* __restrict__ x,
                                                              int32_t* __restrict__ ticket, T* __restrict__ part,
                                                              int G) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  using bits_t = typename Sent<T>::bits_t;
  // 256 threads = one wave per SIMD (up to 512 VGPRs each): lane = 2 rows, wave = 32 columns;
  // per tile a lane has 32 16-byte loads in flight, 128 KiB per workgroup
  constexpr int NC = 32, NG = 4;
  __shared__ int sb;
  __shared__ T sx[2][128];
  __shared__ T red[NG][128];
  __shared__ T sr[128];
  __shared__ T sp[128];
  // tf_b lives in LDS (128 KiB in fp64: this kernel runs one workgroup per CU anyway): loaded FIRST, long before
  // the hop that needs it, and it costs no registers
  __shared__ __attribute__((aligned(16))) T sT[128 * 128];
  const int tid = threadIdx.x, rq = tid & 63;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar address math
  if (tid == 0) sb = atomicAdd(ticket, 1);
  __syncthreads();
  // G workgroups per block row, tickets in this order: G - 1 HELPERS stream the raw tiles (b, c), c <= b - 3 -- helper h the
  // tiles c = b-3-h, b-3-h-(G-1), ... in increasing c, a tile two steps ... one step ahead in registers -- and hand their
  // partial sums over as tagged granules; the PRIMARY streams no tile at all:
  //   x_b = W_b (y_b - sum of the helpers) - tf2_b x_{b-2} - tf_b x_{b-1},   tf_b = W_b L[b, b-1], tf2_b = W_b L[b, b-2]
  // (winv_kernel), all three operands requested at its START: nothing the primary waits for is ever more than one matrix-
  // vector product and one reduction away from x_b.  Round 5 had one helper and the primary streamed every other tile: what
  // followed the arrival of x_{b-2} -- its tile's product, the helper's hand-off, W_b's product: four L2 round trips and
  // four barriers, 4.3 us -- had to fit into ONE hop, and the period was 2.47 us per block whatever the bandwidth (N = 16 384:
  // 0.316 ms = 3.4 TB/s; more workgroups per row alone: slower, profiles/r06_b).
  const int tk = __builtin_amdgcn_readfirstlane(sb);
  const int b = tk / G;
  const int role = tk - b * G;
  const int nh = G - 1;               // helpers per row
  const bool helper = role < nh;      // (their tickets first: a workgroup only ever waits for EARLIER tickets)
  const int64_t npad = int64_t(nblk) * 128;
  if (b >= nblk) return;
  // uniform (SGPR) base + one 32-bit lane offset per load: no per-load address registers
  const T* Lrow = L + int64_t(b) * 128 + int64_t(NC * cg) * ld;  // + (c*128 + j) * ld + 2 rq
  T2 bufA[NC], bufB[NC];
  auto load_tile = [&](T2 (&buf)[NC], int c) {
    const T* p = Lrow + int64_t(c) * 128 * ld;
#pragma unroll
    for (int j = 0; j < NC; ++j) buf[j] = *reinterpret_cast<const T2*>(p + int64_t(j) * ld + 2 * rq);
  };
  // a 128 x 128 column-major block of `base` (W_b, tf2_b): this lane's 2 rows x NC columns
  auto load_blk = [&](T2 (&buf)[NC], const T* base) {
    const T* wb = base + int64_t(b) * 16384 + int64_t(NC * cg) * 128;
#pragma unroll
    for (int j = 0; j < NC; ++j) buf[j] = *reinterpret_cast<const T2*>(wb + j * 128 + 2 * rq);
  };
  bits_t xb = Sent<T>::value;
  // wait for block c of the solution (bounded: a lost producer must end in a NaN, never in a hung GPU)
  auto wait_x = [&](int c, int slot) {
    if (tid < 128) {
      PollClock xclk;
      for (long spin = 0; xb == Sent<T>::value && !xclk.expired(spin); ++spin) {
        if (spin) __builtin_amdgcn_s_sleep(1);
        xb = load_x_bits<T>(x + int64_t(c) * 128 + tid);
      }
      sx[slot][tid] = bits_to<T>(xb);
    }
    __syncthreads();
  };
  // this wave's 32 columns of a block (in `m`) times 32 entries of an LDS vector
  auto matvec = [&](T2 (&m)[NC], const T* v, T& o0, T& o1) {
    o0 = 0;
    o1 = 0;
#pragma unroll
    for (int j0 = 0; j0 < NC; j0 += 8) {
#pragma unroll
      for (int j = j0; j < j0 + 8; ++j) {
        o0 += m[j].x * v[j];
        o1 += m[j].y * v[j];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS operand reads in chunks (register pressure)
    }
  };
  auto sum4 = [&](int i) { return (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]); };
  auto store_tagged = [&](T* dst, T v) {
    bits_t out;
    __builtin_memcpy(&out, &v, sizeof(T));
    if (out == Sent<T>::value) out ^= 1;  // cannot happen for a computed value; keeps the protocol total
    __hip_atomic_store(reinterpret_cast<bits_t*>(dst), out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  if (helper) {
    const int last = b - 3 - role;
    if (last < 0) return;  // (the first blocks have no tiles for every helper; the primary does not wait for those)
    const int first = last % nh;
    const int cnt = (last - first) / nh + 1;  // this workgroup's tiles
    T acc0 = 0, acc1 = 0;
    // the next tile is always requested BEFORE waiting for x_c
    auto step = [&](int c, T2 (&cur)[NC], T2 (&nxt)[NC], bool more, int slot) {
      if (more) load_tile(nxt, c + nh);
      wait_x(c, slot);
      if (tid < 128) xb = more ? load_x_bits<T>(x + int64_t(c + nh) * 128 + tid) : Sent<T>::value;
      T a0, a1;
      matvec(cur, &sx[slot][NC * cg], a0, a1);
      acc0 += a0;
      acc1 += a1;
    };
    load_tile(bufA, first);
    int c = first, left = cnt;
    for (; left >= 2; left -= 2, c += 2 * nh) {
      step(c, bufA, bufB, true, 0);
      step(c + nh, bufB, bufA, left > 2, 1);
    }
    if (left == 1) step(c, bufA, bufB, false, 0);
    red[cg][2 * rq] = acc0;
    red[cg][2 * rq + 1] = acc1;
    __syncthreads();
    if (tid < 128) store_tagged(part + int64_t(role) * npad + int64_t(b) * 128 + tid, sum4(tid));
    return;
  }

  // ---- the primary ----
  if (b >= 1) {  // tf_b -> LDS; column j: 64 lanes x 16 bytes = one wave transfer (bufA is the staging image)
    const T* tb = tfwd + int64_t(b) * 16384 + int64_t(NC * cg) * 128 + 2 * rq;
#pragma unroll
    for (int j = 0; j < NC; ++j) bufA[j] = *reinterpret_cast<const T2*>(tb + j * 128);
#pragma unroll
    for (int j = 0; j < NC; ++j) *reinterpret_cast<T2*>(&sT[(NC * cg + j) * 128 + 2 * rq]) = bufA[j];
    __builtin_amdgcn_sched_barrier(0);  // (the staging image is free before the next two are requested)
  }
  load_blk(bufB, winv);                 // W_b
  if (b >= 2) load_blk(bufA, tfwd2);    // tf2_b
  if (tid < 128) {
    const T yv = yin[int64_t(b) * 128 + tid];
    T pv = T(0);
    for (int h = 0; h < nh && b - 3 - h >= 0; ++h) {  // the helpers' sums, in helper order
      bits_t pb = Sent<T>::value;
      PollClock pclk;
      for (long spin = 0; pb == Sent<T>::value && !pclk.expired(spin); ++spin) {
        if (spin) __builtin_amdgcn_s_sleep(1);
        pb = load_x_bits<T>(part + int64_t(h) * npad + int64_t(b) * 128 + tid);
      }
      pv += bits_to<T>(pb);
    }
    sr[tid] = yv - pv;
    if (b >= 2) xb = load_x_bits<T>(x + int64_t(b - 2) * 128 + tid);  // (its round trip under W_b's product)
  }
  __syncthreads();
  T p0, p1;
  matvec(bufB, &sr[NC * cg], p0, p1);
  red[cg][2 * rq] = p0;
  red[cg][2 * rq + 1] = p1;
  __syncthreads();
  if (b == 0) {
    if (tid < 128) store_tagged(x + tid, sum4(tid));
    return;
  }
  if (tid < 128) sp[tid] = sum4(tid);
  if (b >= 2) {
    wait_x(b - 2, 1);  // (its barrier also separates the reads of `red` above from the writes below)
    xb = Sent<T>::value;
    if (tid < 128) xb = load_x_bits<T>(x + int64_t(b - 1) * 128 + tid);  // (usually still the sentinel)
    T q0, q1;
    matvec(bufA, &sx[1][NC * cg], q0, q1);
    red[cg][2 * rq] = q0;
    red[cg][2 * rq + 1] = q1;
    __syncthreads();
    if (tid < 128) sp[tid] -= sum4(tid);
  }
  // the hop: x_{b-1} seen -> one product with tf_b (LDS), one reduction, the store
  wait_x(b - 1, 0);  // (barrier: `red` is free again)
  T q0 = 0, q1 = 0;
  {
    const T* tcol = &sT[(NC * cg) * 128 + 2 * rq];
    const T* xv = &sx[0][NC * cg];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const T2 t2 = *reinterpret_cast<const T2*>(tcol + j * 128);
      q0 += t2.x * xv[j];
      q1 += t2.y * xv[j];
    }
  }
  red[cg][2 * rq] = q0;
  red[cg][2 * rq + 1] = q1;
  __syncthreads();
  if (tid < 128) store_tagged(x + int64_t(b) * 128 + tid, sp[tid] - sum4(tid));
}

// Backward substitution L^T x = z in one launch: workgroup (ticket) t owns block b = nblk-1-t, i.e.
// the 128 COLUMNS b of L, and streams the tiles (c, b), c = nblk-1 .. b+1, as the solved blocks
// x_c appear from the bottom up.  Output j of the block is a dot product ALONG a column of L --
// contiguous memory -- so a lane keeps, for each of its wave's 32 columns, the partial sum over
// its own two rows across ALL tiles, and the cross-lane reduction happens once per block (LDS
// transpose, fixed order).  The diagonal block is applied as W_b^T (stored transposed by
// winv_kernel, so the product is the forward kernel's).  Same hand-off protocol as above.
template <typename T>
__global__ __launch_bounds__(512) void trsv_bwd_stream_kernel(int nblk, const T* __restrict__ L, int64_t ld,
                                                              const T* __restrict__ winvT,
                                                              const T* __restrict__ zin, T* __restrict__ x,
                                                              int32_t* __restrict__ ticket, T* __restrict__ part) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  using bits_t = typename Sent<T>::bits_t;
  constexpr int NC = 16, NG = 8, PAD = 65;  // 8 waves (two per SIMD, <= 256 VGPRs each) x 16 columns
  __shared__ int sb;
  __shared__ T sx[2][128];
  __shared__ T redT[NG][NC][PAD];  // per-lane partial sums, lane-contiguous (padded: conflict-free column sums)
  __shared__ T red[NG][128];
  __shared__ T sr[128];
  const int tid = threadIdx.x, rq = tid & 63;
  const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) sb = atomicAdd(ticket, 1);
  __syncthreads();
  // two workgroups per block column, as in the forward kernel: ticket 2t the helper (tiles b+2, b+4, ..., partial sums
  // handed over as a tagged granule), ticket 2t+1 the primary (tiles b+1, b+3, ..., W_b^T and the hop)
  const int tk = __builtin_amdgcn_readfirstlane(sb);
  const int t = tk >> 1;
  const bool helper = (tk & 1) == 0;
  if (t >= nblk) return;
  const int b = nblk - 1 - t;
  const T* Lcol = L + (int64_t(b) * 128 + NC * cg) * ld;  // + j * ld + c * 128 + 2 rq
  T acc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) acc[j] = 0;
  T2 bufA[NC], bufB[NC];
  auto load_tile = [&](T2 (&buf)[NC], int c) {
    const T* p = Lcol + int64_t(c) * 128;
#pragma unroll
    for (int j = 0; j < NC; ++j) buf[j] = *reinterpret_cast<const T2*>(p + int64_t(j) * ld + 2 * rq);
  };
  auto load_w = [&](T2 (&buf)[NC]) {
    const T* wb = winvT + int64_t(b) * 16384 + int64_t(NC * cg) * 128;
#pragma unroll
    for (int j = 0; j < NC; ++j) buf[j] = *reinterpret_cast<const T2*>(wb + j * 128 + 2 * rq);
  };
  bits_t xb = Sent<T>::value;
  // tiles are visited in DEcreasing c, every second one; next: 0 nothing, 1 tile c-2, 2 W_b^T
  auto step = [&](int c, T2 (&cur)[NC], T2 (&nxt)[NC], int next) {
    if (next == 1) load_tile(nxt, c - 2);
    if (next == 2) load_w(nxt);
    const int slot = (c >> 1) & 1;
    if (tid < 128) {
      PollClock xclk;
      for (long spin = 0; xb == Sent<T>::value && !xclk.expired(spin); ++spin) {
        if (spin) __builtin_amdgcn_s_sleep(1);
        xb = load_x_bits<T>(x + int64_t(c) * 128 + tid);
      }
      sx[slot][tid] = bits_to<T>(xb);
    }
    __syncthreads();
    if (tid < 128) xb = (next == 1) ? load_x_bits<T>(x + int64_t(c - 2) * 128 + tid) : Sent<T>::value;
    const T x0 = sx[slot][2 * rq], x1 = sx[slot][2 * rq + 1];
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[j] += cur[j].x * x0 + cur[j].y * x1;
  };
  auto finish = [&](T2 (&wf)[NC]) {
#pragma unroll
    for (int j = 0; j < NC; ++j) redT[cg][j][rq] = acc[j];
    __syncthreads();
    {  // column jj of the block: 64 lane partials, four threads x 16 lanes, fixed order
      const int jj = tid >> 2, h = tid & 3;
      const T* src = &redT[jj / NC][jj % NC][16 * h];
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int l = 0; l < 16; l += 2) {
        s0 += src[l];
        s1 += src[l + 1];
      }
      red[h][jj] = s0 + s1;
    }
    __syncthreads();
    if (helper) {  // this workgroup's share of the column sums -> part[b] (tagged like x); nothing else to do
      if (tid < 128) {
        const T pv = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        bits_t out;
        __builtin_memcpy(&out, &pv, sizeof(T));
        if (out == Sent<T>::value) out ^= 1;
        __hip_atomic_store(reinterpret_cast<bits_t*>(part + int64_t(b) * 128 + tid), out, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
    if (tid < 128) {
      T pv = T(0);
      if (b + 2 <= nblk - 1) {  // the helper's tiles b+2, b+4, ...
        bits_t pb = Sent<T>::value;
        PollClock pclk;
        for (long spin = 0; pb == Sent<T>::value && !pclk.expired(spin); ++spin) {
          if (spin) __builtin_amdgcn_s_sleep(1);
          pb = load_x_bits<T>(part + int64_t(b) * 128 + tid);
        }
        pv = bits_to<T>(pb);
      }
      sr[tid] = (zin[int64_t(b) * 128 + tid] - ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]))) - pv;
    }
    __syncthreads();
    T p0 = 0, p1 = 0;
#pragma unroll
    for (int j0 = 0; j0 < NC; j0 += 8) {
#pragma unroll
      for (int j = j0; j < j0 + 8; ++j) {
        const T rj = sr[NC * cg + j];
        p0 += wf[j].x * rj;
        p1 += wf[j].y * rj;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    red[cg][2 * rq] = p0;
    red[cg][2 * rq + 1] = p1;
    __syncthreads();
    if (tid < 128) {
      const T xv = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) +
                   ((red[4][tid] + red[5][tid]) + (red[6][tid] + red[7][tid]));
      bits_t out;
      __builtin_memcpy(&out, &xv, sizeof(T));
      if (out == Sent<T>::value) out ^= 1;
      __hip_atomic_store(reinterpret_cast<bits_t*>(x + int64_t(b) * 128 + tid), out, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // tiles c = nblk-1 .. b+1: the primary takes b+1, b+3, ..., the helper b+2, b+4, ... (each from its largest c down)
  const int last = helper ? b + 2 : b + 1;
  if (last > nblk - 1) {
    if (helper) return;  // (no tiles: the primary does not wait for this helper)
    load_w(bufA);
    finish(bufA);
    return;
  }
  const int cnt = (nblk - 1 - last) / 2 + 1;
  int c = last + 2 * (cnt - 1), left = cnt;
  load_tile(bufA, c);
  for (; left > 2; left -= 2, c -= 4) {
    step(c, bufA, bufB, 1);
    step(c - 2, bufB, bufA, 1);
  }
  const int tail_next = helper ? 0 : 2;  // the primary's last step fetches W_b^T into the free buffer
  if (left == 2) {
    step(c, bufA, bufB, 1);
    step(c - 2, bufB, bufA, tail_next);
    finish(bufA);
  } else {
    step(c, bufA, bufB, tail_next);
    finish(bufB);
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// the side-stream assembly that assemble_lower held back (ctx option asm_defer): now, behind everything queued on `behind`
// (NULL: at once)
int run_deferred_asm(tgp_ctx* ctx, hipStream_t behind) {
  if (!ctx->deferred_asm) return TGP_OK;
  auto f = std::move(ctx->deferred_asm);
  ctx->deferred_asm = nullptr;
  if (behind != nullptr && ctx->asm_stream != nullptr) {
    if (ctx->ev_asm_gate == nullptr) TGP_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_asm_gate, hipEventDisableTiming));
    TGP_HIP_TRY(hipEventRecord(ctx->ev_asm_gate, behind));
    TGP_HIP_TRY(hipStreamWaitEvent(ctx->asm_stream, ctx->ev_asm_gate, 0));
  }
  return f();
}

// The bound lives in ONE device global of the library, shared by every context of the process on that device; the host
// keeps ONE mirror of it (advisor r5: with a copy per context, the host deadline of join_bounded and get_option of a context
// disagreed with the device as soon as another context had set the option).
static std::atomic<int64_t> g_poll_ms_host{4000};
int64_t poll_limit_ms() { return g_poll_ms_host.load(); }
int set_poll_limit(tgp_ctx* ctx, int64_t ms) {
  TGP_ARG_CHECK(ms >= 1 && ms <= 3600000, "poll_timeout_ms must be in [1, 3600000]");
  const long long ticks = (long long)ms * 100000LL;  // s_memrealtime: 100 MHz
  TGP_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_poll_limit), &ticks, sizeof(ticks)));
  g_poll_ms_host.store(ms);
  ctx->poll_timeout_ms = ms;
  return TGP_OK;
}

// A stream wait-value has no timeout of its own: if the chain launch it waits for never makes progress (a tool that holds
// kernels back in an order of its own -- rocprofv3 --pmc hung exactly here, profiles/r05_b -- or a launch that was lost),
// hipStreamSynchronize would block for ever.  So a factorisation that enqueued wait-values is joined with a DEADLINE: the
// host polls an event (busy for the first 50 ms -- no latency added to evaluations up to N ~ 16 k --, then every 200 us);
// past poll_timeout_ms + 3 x the time of N^3 / 3 flops at 20 TFLOP/s it releases every pending wait itself
// (hipStreamWriteValue32 from a rescue stream: command-processor writes, no kernel), joins all streams and reports
// TGP_E_TIMEOUT -- tgp_solver_factor* then repeats the pass on the launch-per-block path, which has no device-side waits.
int join_bounded(tgp_ctx* ctx, hipStream_t st, int64_t n) {
  TGP_TRY(ev_record(ctx, ctx->ev_join, st));
  const auto t0 = std::chrono::steady_clock::now();
  const double budget_ms = double(poll_limit_ms()) + 3.0 * (double(n) * double(n) * double(n) / 3.0) / 2e13 * 1e3;
  for (long spin = 0;; ++spin) {
    const hipError_t e = hipEventQuery(ctx->ev_join);
    if (e == hipSuccess) return TGP_OK;
    if (e != hipErrorNotReady) {
      set_error("hipEventQuery failed: %s", hipGetErrorString(e));
      return TGP_E_HIP;
    }
    (void)hipGetLastError();
    if (ctx->fault_inject == 1) {  // test hook: the deadline has passed NOW, while the device is still at work
      ctx->fault_inject = 0;
      break;
    }
    if ((spin & 63) != 63) continue;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > budget_ms) break;
    if (ms > 50.0) std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  // rescue: satisfy every wait on the block columns' counters, then drain
  if (ctx->rescue_stream == nullptr) TGP_HIP_TRY(hipStreamCreateWithFlags(&ctx->rescue_stream, hipStreamNonBlocking));
  for (int c = 0; c < CHAIN_FLAG_LD; ++c)
    (void)hipStreamWriteValue32(ctx->rescue_stream, ctx->d_chain_ticket + CHAIN_COLCNT_OFF + c, 0x7fffffffu, 0);
  for (hipStream_t q : {ctx->rescue_stream, ctx->asm_stream, ctx->panel_stream, ctx->update_stream, ctx->solve_stream, ctx->stream})
    if (q) (void)hipStreamSynchronize(q);
  (void)hipGetLastError();
  set_error("potrf: the followers of a chain launch were still waiting after %.0f ms (stream wait-values released by the host)", budget_ms);
  return TGP_E_TIMEOUT;
}

template <typename T>
int launch_potf2(tgp_ctx* ctx, hipStream_t st, T* A, int64_t ld, T* dinv, int32_t* info,
                 int32_t pivot_base, const T* Xp, int64_t ldx) {
  if (ctx->trace) {  // v: tile offset, pending-update operand offset (-1: none), ld
    trace_push(ctx, 1, st, trace_off(ctx, A), trace_off(ctx, Xp), ld);
    return TGP_OK;
  }
  const int32_t* wc = ctx->potf2_wait_counter;  // (set by the merged schedule in front of a panel's first potf2: potrf)
  const int32_t wt = (int32_t)ctx->potf2_wait_target;
  ctx->potf2_wait_counter = nullptr;
  if (Xp != nullptr)
    hipLaunchKernelGGL((potf2_kernel<T, true>), dim3(1), dim3(512), 0, st, A, ld, dinv, info, pivot_base, Xp, ldx, wc, wt);
  else
    hipLaunchKernelGGL((potf2_kernel<T, false>), dim3(1), dim3(512), 0, st, A, ld, dinv, info, pivot_base, Xp, ldx, wc, wt);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int launch_trsm(tgp_ctx* ctx, hipStream_t st, int64_t m, const T* L, int64_t ldl, const T* dinv,
                T* B, int64_t ldb) {
  if (m == 0) return TGP_OK;
  TGP_ARG_CHECK(m % 16 == 0, "trsm: m must be a multiple of 16");
  if (ctx->trace) {  // v: L tile offset, B offset, rows, ld
    trace_push(ctx, 2, st, trace_off(ctx, L), trace_off(ctx, B), m, ldl);
    return TGP_OK;
  }
  hipLaunchKernelGGL((trsm_kernel<T>), dim3((unsigned)((m + 63) / 64)), dim3(256), 0, st, m, L,
                     ldl, dinv, B, ldb);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int launch_panel_step(tgp_ctx* ctx, hipStream_t st, int64_t m, T* Ljj, int64_t ld, T* dj, int32_t* info,
                      int32_t pivot_base, const T* Xp, bool has_p) {
  TGP_ARG_CHECK(m >= 0 && m % TILE == 0, "panel step: rows below must be a multiple of %d", TILE);
  if (ctx->trace) {  // v: tile offset, pending-update operand offset (-1: none), rows below, ld, has_p
    trace_push(ctx, 10, st, trace_off(ctx, Ljj), trace_off(ctx, Xp), m, ld, has_p ? 1 : 0);
    return TGP_OK;
  }
  const unsigned grid = (unsigned)(m / TILE) + (has_p ? 1u : 0u);
  if (grid == 0) return TGP_OK;
  const uint32_t epoch = ++ctx->step_epoch;  // launches of a context are issued under its lock, in stream order
  if (Xp != nullptr)
    hipLaunchKernelGGL((panel_step_kernel<T, true>), dim3(grid), dim3(512), 0, st, Ljj, ld, dj, info, pivot_base,
                       Xp, has_p ? 1 : 0, ctx->d_step_flag, epoch);
  else
    hipLaunchKernelGGL((panel_step_kernel<T, false>), dim3(grid), dim3(512), 0, st, Ljj, ld, dj, info, pivot_base,
                       Xp, has_p ? 1 : 0, ctx->d_step_flag, epoch);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// the update-batching policy of a launch: fp64 only (the batched product is chain_update_fast, the 4x4x4 fp64 form)
template <typename T>
ChainPolicy chain_policy(const tgp_ctx* ctx) {
  if (sizeof(T) != 8 || ctx->chain_batch <= 1) return ChainPolicy{1, 1, 2, 0};
  return ChainPolicy{(int)ctx->chain_batch, (int)ctx->chain_batch_lag, (int)ctx->chain_batch_rowlag, (int)ctx->chain_batch_minrows};
}

// Persistent chain over block columns [cb, ce) of the panel whose origin is A0 (R row tiles down to the end of
// the matrix; dinv0 = the inverses of the panel's first block).  head_done: L_cb,cb is already factored.
template <typename T>
int launch_chain(tgp_ctx* ctx, hipStream_t st, T* A0, int64_t ld, T* dinv0, int64_t pivot_base, int64_t R,
                 int64_t nblk, int64_t cb, int64_t ce, bool head_done, hipEvent_t counters_ready, T* y0, bool fprev) {
  TGP_ARG_CHECK(R >= 1 && R <= CHAIN_MAX_ROW_TILES && cb >= 0 && cb < ce && ce <= nblk && nblk <= CHAIN_FLAG_LD && nblk <= R,
                "chain: bad panel shape (R=%lld, columns [%lld, %lld))", (long long)R, (long long)cb, (long long)ce);
  // a panel's first block has no in-panel update pending: plain potf2 in front of the launch (in the look-ahead
  // schedule it already ran on the main stream, in front of the big update: head_done)
  if (cb == 0 && !head_done) {
    TGP_TRY(launch_potf2<T>(ctx, st, A0, ld, dinv0, ctx->d_info, (int32_t)pivot_base));
    if (!ctx->trace) TGP_TRY(run_deferred_asm(ctx, st));  // (asm_defer: the other columns' assembly behind the first potf2)
  }
  // the pollers of the previous launch read the counters this launch zeroes: they must be through
  if (ctx->chain_polls_pending) {
    TGP_TRY(st_wait(ctx, st, ctx->ev_f));
    ctx->chain_polls_pending = false;
  }
  if (ctx->trace) {  // v: panel origin offset, ld, row tiles, first / end block column
    if (counters_ready != nullptr) TGP_TRY(ev_record(ctx, counters_ready, st));
    trace_push(ctx, 11, st, trace_off(ctx, A0), ld, R, cb, ce, nblk, y0 != nullptr ? 1 : 0);
    return TGP_OK;
  }
  // the launch's task list (chain_tasks.h: ticket order + K-batched updates), one table per shape and policy, kept on the
  // device for the life of the context (13 shapes per evaluation at c2, the same in every evaluation)
  const ChainPolicy pol = chain_policy<T>(ctx);
  const int fwd = y0 != nullptr ? 1 : 0;  // the forward substitution of these block columns as tasks of the launch
  const std::array<int64_t, 9> key = {R, nblk, cb, ce, pol.batch, pol.lag, pol.rowlag, pol.minrows, fwd};
  auto found = ctx->chain_tables.find(key);
  if (found == ctx->chain_tables.end()) {
    const std::vector<ChainTask> list = chain_build((int)R, (int)nblk, (int)cb, (int)ce, pol, fwd, CHAIN_FWD_GROUP);
    tgp_ctx::ChainTable tab;
    tab.count = (int64_t)list.size();
    if (tab.count > 0) {
      std::vector<uint64_t> words(list.size());
      for (size_t u = 0; u < list.size(); ++u) words[u] = chain_pack(list[u]);
      TGP_HIP_TRY(hipMalloc((void**)&tab.dev, words.size() * sizeof(uint64_t)));
      // (blocking copy on the null stream: the library's streams are non-blocking, nothing of theirs is joined)
      TGP_HIP_TRY(hipMemcpy(tab.dev, words.data(), words.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    found = ctx->chain_tables.emplace(key, tab).first;
  }
  const int64_t tasks = found->second.count;
  if (tasks == 0) {  // a one-block panel: potf2 in front was all of it; the pollers' event still marks this point
    if (counters_ready != nullptr) TGP_TRY(ev_record(ctx, counters_ready, st));
    return TGP_OK;
  }
  ctx->step_epoch = (ctx->step_epoch + 1) & 0x1FFFFFFu;  // the state words hold epoch * 128 + s
  if (ctx->step_epoch == 0) ctx->step_epoch = 1;          // 0 is the words' initial value
  const uint32_t epoch = ctx->step_epoch;
  ChainArgs<T> q;
  q.A0 = A0; q.ld = ld; q.dinv = dinv0; q.info = ctx->d_info; q.flags = ctx->d_chain_flags;
  q.ticket = ctx->d_chain_ticket; q.epoch32 = epoch * 128u; q.pivot_base = (int32_t)pivot_base; q.R = (int32_t)R;
  q.nblk = (int32_t)nblk;
  q.cb = (int32_t)cb; q.ce = (int32_t)ce;
  q.stamps = nullptr;
  q.launch = (int32_t)ctx->chain_launches++;
  q.fast_update = (int32_t)ctx->chain_fast_update;
  q.tasks = found->second.dev;
  q.y = y0;
  q.fgs = CHAIN_FWD_GROUP;
  q.fprev = fprev ? 1 : 0;
  q.red = nullptr; q.scal = nullptr; q.red_total = 0;
  if (fwd != 0 && ctx->chain_red_total > 0 && ctx->chain_red_total <= CHAIN_MAX_ROW_TILES && pivot_base % TILE == 0) {
    q.red = ctx->d_chain_red;
    q.scal = ctx->d_scal;
    if (pivot_base / TILE + ce == ctx->chain_red_total) {  // the launch with the matrix's last block adds the partial sums up
      q.red_total = (int32_t)ctx->chain_red_total;
      ctx->reductions_done = true;
    }
  }
  if (ctx->chain_stamps != 0 && ctx->chain_stamp_base + tasks <= CHAIN_STAMP_TASKS) {
    if (ctx->d_chain_stamps == nullptr)
      TGP_HIP_TRY(hipMalloc((void**)&ctx->d_chain_stamps, size_t(CHAIN_STAMP_TASKS) * 16 * sizeof(long long)));
    q.stamps = ctx->d_chain_stamps + ctx->chain_stamp_base * 16;
    ctx->chain_stamp_base += tasks;
  }
  // ticket counter + the block columns' counts of final tiles: zero in front of every launch
  TGP_HIP_TRY(hipMemsetAsync(ctx->d_chain_ticket, 0, size_t(CHAIN_TICKET_WORDS) * sizeof(int32_t), st));
  // pollers on other streams start behind THIS point: counters zeroed, the launch itself not awaited
  if (counters_ready != nullptr) TGP_TRY(ev_record(ctx, counters_ready, st));
  // one task per workgroup; `chain_lds_pad` bytes of dynamic LDS nobody uses keep a SECOND chain workgroup off the
  // compute unit (76 + 10 KB > 160 / 2): the diagonal chain shares its MFMA pipes with no update task (potf2 + fold
  // 37 us alone, 51-54 beside one: profiles/r04_b), while one trailing-update workgroup (74 KB) still fits beside it
  // (fp32: the tile image is half the size, the pad makes up for it)
  const size_t dyn = ctx->chain_lds_pad > 0 ? size_t(ctx->chain_lds_pad) + (sizeof(T) == 4 ? 36 * 256 * 4 : 0) : 0;
  hipLaunchKernelGGL((chain_kernel<T>), dim3((unsigned)tasks), dim3(512), dyn, st, q);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// `st` continues once block column c of the chain launch that is running (or queued) on another stream is final:
// R - c tiles (the panel's first diagonal block, factored in front of a launch with cb == 0, is not counted)
int launch_chain_poll(tgp_ctx* ctx, hipStream_t st, const void* A0, int64_t ld, int64_t R, int64_t c, bool first_external) {
  if (ctx->trace) {  // v: panel origin offset (elements of the traced dtype), ld, block column
    trace_push(ctx, 12, st, trace_off(ctx, static_cast<const double*>(A0)), ld, c);
    return TGP_OK;
  }
  const int32_t target = (int32_t)(R - c - ((c == 0 && first_external) ? 1 : 0));
  int32_t* count = ctx->d_chain_ticket + CHAIN_COLCNT_OFF + c;
  if (ctx->chain_polls == 3 && ctx->can_wait_value) {
    // Round 5, MEASURED, not the default: hipStreamWaitValue32 on the counter instead of our own poll kernel.  It works on
    // plain device memory and its hand-off is as fast (1.35 us against 1.55, scripts/probe_waitvalue.hip; c2 25.56 vs
    // 25.53 ms) -- but it is NOT a command-processor wait: the runtime launches a one-wave wait kernel of its own
    // (__amd_rocclr_streamOpsWait in the kernel trace, 1 664 per 13 evaluations: profiles/r05_e), which spins like ours
    // and, unlike ours, has no timeout -- under rocprofv3 --pmc it hung for good (profiles/r05_b).  A factorisation that
    // enqueues these is therefore joined by the host with a deadline (join_bounded).
    ctx->wait_values_inflight = true;
    TGP_HIP_TRY(hipStreamWaitValue32(st, count, (uint32_t)target, hipStreamWaitValueGte, 0xffffffffu));
    return TGP_OK;
  }
  hipLaunchKernelGGL(chain_poll_kernel, dim3(1), dim3(64), 0, st, count, target, ctx->d_info);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// `st` continues once the PREFIX (the next panel's block column: `target` tiles) of the merged trailing update that is
// running (or queued) on the main stream is complete (gemm.hip, GemmArgs::prefix_done).  Bounded like every poll.
int launch_prefix_poll(tgp_ctx* ctx, hipStream_t st, const int32_t* counter, int64_t target, int64_t c_off, int64_t ld) {
  if (ctx->trace) {  // v: C offset of the update the poll belongs to, ld
    trace_push(ctx, 13, st, c_off, ld);
    return TGP_OK;
  }
  hipLaunchKernelGGL(chain_poll_kernel, dim3(1), dim3(64), 0, st, counter, (int32_t)target, ctx->d_info);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int compute_dinv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* dinv) {
  if (n == 0) return TGP_OK;
  hipLaunchKernelGGL((dinv_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, n, L, ld, dinv);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// one forward-substitution step: y[0:128] <- L_jj^-1 y[0:128]; y[128:128+m] -= L[below] y[0:128]
template <typename T>
int launch_trsv_fwd_step(tgp_ctx* ctx, hipStream_t st, int64_t m_below, const T* Ljj, int64_t ld,
                         const T* dj, T* yj) {
  if (ctx->trace) {  // v: L tile offset, rows below, ld
    trace_push(ctx, 4, st, trace_off(ctx, Ljj), m_below, ld);
    return TGP_OK;
  }
  hipLaunchKernelGGL((trsv_diag_fwd_kernel<T>), dim3(1), dim3(128), 0, st, Ljj, ld, dj, yj);
  if (m_below > 0)
    hipLaunchKernelGGL((trsv_update_fwd_kernel<T>), dim3((unsigned)((m_below + 255) / 256)),
                       dim3(256), 0, st, m_below, Ljj + TILE, ld, yj, yj + TILE);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

namespace {
struct ProfSpan {
  hipEvent_t e0, e1;
  double flops;
};

int prof_event(tgp_ctx* ctx, hipEvent_t* out) {
  if (ctx->ev_used == ctx->ev_pool.size()) {
    hipEvent_t e;
    TGP_HIP_TRY(hipEventCreate(&e));
    ctx->ev_pool.push_back(e);
  }
  *out = ctx->ev_pool[ctx->ev_used++];
  return TGP_OK;
}
}  // namespace

// potf2 of the 128-block at (j0, j0) of A; `pend`: fold the pending in-panel update from block
// column j0-128 in.  `pivot_off`: global index of A's first row (distributed panels).
template <typename T>
int panel_potf2(tgp_ctx* ctx, hipStream_t st, T* A, int64_t ld, T* dinv, int64_t pivot_off,
                int64_t j0, bool pend) {
  return launch_potf2<T>(ctx, st, A + j0 * ld + j0, ld, dinv + (j0 / TILE) * 2048, ctx->d_info,
                         (int32_t)(pivot_off + j0),
                         pend ? (const T*)(A + (j0 - TILE) * ld + j0) : (const T*)nullptr, ld);
}

// The panel chain: columns [k0, k0 + kb) of the n-row matrix A are factored in 128-column
// blocks [potf2 | trsm of every row below | in-panel update of the columns to the right].  Per
// block the critical path is potf2 -> trsm; the in-panel update runs on its own stream
// (ctx->update_stream) beside the NEXT block's potf2, which folds the update of its own
// diagonal tile in (role 3 skips that tile).  y != nullptr: forward-substitution step j is
// queued on ctx->solve_stream as soon as block column j is final.
// `after_blocks` / `mid`: once that many blocks of the panel are final, mid() is called with
// ev_d recorded behind the last of them (the caller hangs an early partial update on it).
// Shared by the single-GPU driver (potrf) and the block-column driver (dist.hip).
template <typename T>
int panel_chain(tgp_ctx* ctx, hipStream_t st, int64_t n, T* A, int64_t ld, T* dinv,
                int64_t pivot_off, int64_t k0, int64_t kb, bool head_done, T* y,
                int64_t after_blocks, const std::function<int(hipEvent_t)>& mid, int64_t blk_begin,
                int64_t blk_end) {
  // [blk_begin, blk_end): the 128-column blocks of the panel to run NOW (default: all).  The block-column driver
  // factors a panel in column chunks so that the broadcast of a finished chunk overlaps the rest of the chain;
  // the in-panel updates of a block always cover the whole rest of the panel (resp. of its sub-panel), and the
  // marker of the update in flight (ev_e) carries over from one call to the next.
  const int64_t j_first = k0 + blk_begin * TILE;
  const int64_t j_stop = blk_end < 0 ? k0 + kb : std::min<int64_t>(k0 + kb, k0 + blk_end * TILE);
  hipStream_t S3 = ctx->update_stream;
  hipStream_t S2 = ctx->solve_on_update != 0 ? S3 : ctx->solve_stream;  // behind the update of the same block
  if (ctx->chain_kernel != 0 && kb / TILE <= CHAIN_FLAG_LD && (n - k0) / TILE <= CHAIN_MAX_ROW_TILES) {
    // ONE persistent launch for the blocks to run now: potf2, the solves of the rows below and the in-panel updates
    // are tile tasks of chain_kernel.  What the per-block path hangs on events behind block j -- the forward-
    // substitution step of block j, the early share of the next block-column update behind `after_blocks` blocks --
    // follows on the solve stream behind a ONE-wave poll of block column j's count of final tiles: the launch is
    // never cut, and those kernels still start while it runs.
    const int64_t nblk = kb / TILE, R = (n - k0) / TILE;
    const int64_t cb = blk_begin, ce = blk_end < 0 ? nblk : std::min<int64_t>(nblk, blk_end);
    T* A0 = A + k0 * ld + k0;
    T* d0 = dinv + (k0 / TILE) * 2048;
    if (cb >= ce) return TGP_OK;
    // ONE launch for the whole range.  (Measured and dropped, profiles/r04_d: a few block columns per launch beside a
    // big trailing update -- fewer waiting workgroups holding compute units: the update's launches ran at 0.70 instead
    // of 0.62 of peak, the evaluation took 30.5 instead of 27.4 ms; the launch gaps put the chain on the critical path.)
    // (as in the per-block path: the early share exists when rows AND columns are left behind block after_blocks-1)
    const bool mid_in = after_blocks > cb && after_blocks <= ce && after_blocks < nblk && R > after_blocks;
    // Round 6: the forward-substitution steps of these block columns are TASKS of the launch (chain_tasks.h, F(c)) -- no
    // poller, no per-block launch pair on the solve stream (ctx option chain_fwd_tasks = 0: round 5's followers)
    const bool fwd_in = y != nullptr && ctx->chain_fwd_tasks != 0;
    T* y_follow = fwd_in ? (T*)nullptr : y;
    const bool follow = y_follow != nullptr || mid_in;
    const bool polls = follow && ctx->chain_polls != 0;
    // (ev_d is recorded between the launch's memset and the kernel: the pollers wait for the zeroed counters only;
    // without pollers -- chain_polls = 0 -- it is recorded BEHIND the kernel and the followers wait for the whole launch)
    TGP_TRY(launch_chain<T>(ctx, st, A0, ld, d0, pivot_off + k0, R, nblk, cb, ce, head_done && cb == 0,
                            polls ? ctx->ev_d : (hipEvent_t) nullptr, fwd_in ? y + k0 : (T*)nullptr, k0 > 0));
    if (follow && !polls) TGP_TRY(ev_record(ctx, ctx->ev_d, st));
    if (follow) TGP_TRY(st_wait(ctx, S2, ctx->ev_d));
    for (int64_t c = cb; c < ce && follow; ++c) {
      const int64_t j0 = k0 + c * TILE;
      const bool mid_here = mid_in && c + 1 == after_blocks;
      if (y_follow == nullptr && !mid_here) continue;
      if (polls) TGP_TRY(launch_chain_poll(ctx, S2, A0, ld, R, c, cb == 0));
      if (y_follow != nullptr)
        TGP_TRY(launch_trsv_fwd_step<T>(ctx, S2, n - (j0 + TILE), A + j0 * ld + j0, ld, dinv + (j0 / TILE) * 2048,
                                        y + j0));
      if (mid_here) {
        TGP_TRY(ev_record(ctx, ctx->ev_e, S2));
        TGP_TRY(mid(ctx->ev_e));
      }
    }
    if (polls) {  // the next launch zeroes the counters only behind the last poller of this one
      TGP_TRY(ev_record(ctx, ctx->ev_f, S2));
      ctx->chain_polls_pending = true;
    }
    return TGP_OK;
  }
  if (ctx->fused_step != 0 && blk_begin == 0 && blk_end < 0) {
    // One launch per block (panel_step_kernel).  The rows' workgroups apply the update of THIS column
    // block from the previous one themselves, so the separate in-panel update of block j covers the
    // column blocks j+2.. only and is not needed before step j+2: two alternating markers.
    hipEvent_t far_ev[2] = {ctx->ev_e, ctx->ev_f};
    // split gate (potrf): column block 1 of this panel is complete at ev_g1, column blocks 2.. at ev_g2
    const bool gate = ctx->gate_pending;
    ctx->gate_pending = false;
    int64_t q = 0;
    for (int64_t j0 = k0; j0 < k0 + kb; j0 += TILE, ++q) {
      T* Ljj = A + j0 * ld + j0;
      T* dj = dinv + (j0 / TILE) * 2048;
      const bool pend = j0 > k0;
      const bool has_p = pend || !head_done;
      const int64_t mb = n - (j0 + TILE);
      if (q >= 2) TGP_TRY(st_wait(ctx, st, far_ev[q & 1]));  // far update of block q-2 (and, in order, all before it)
      if (gate && q == 1) TGP_TRY(st_wait(ctx, st, ctx->ev_g1));
      TGP_TRY(launch_panel_step<T>(ctx, st, mb, Ljj, ld, dj, ctx->d_info, (int32_t)(pivot_off + j0),
                                   pend ? (const T*)(A + (j0 - TILE) * ld + j0) : (const T*)nullptr, has_p));
      const int64_t nc = (k0 + kb) - (j0 + 2 * TILE);  // columns j+2.. of the panel
      const int64_t mb2 = mb - TILE;                    // rows below block j+1's diagonal tile, and that tile
      const bool upd = mb2 > 0 && nc > 0;
      const bool need_mid = after_blocks > 0 && mb > 0 && (k0 + kb) - (j0 + TILE) > 0 &&
                            j0 + TILE == k0 + after_blocks * TILE;
      if (y != nullptr || upd || need_mid) TGP_TRY(ev_record(ctx, ctx->ev_d, st));
      if (upd) {
        TGP_TRY(st_wait(ctx, S3, ctx->ev_d));
        if (gate && q == 0) TGP_TRY(st_wait(ctx, S3, ctx->ev_g2));  // the same tiles: gate pieces first
        TGP_TRY(launch_gemm_nt<T>(ctx, S3, mb2, nc, TILE, Ljj + 2 * TILE, ld, Ljj + 2 * TILE, ld,
                                  A + (j0 + 2 * TILE) * ld + j0 + 2 * TILE, ld, 1, 0, 1));
        TGP_TRY(ev_record(ctx, far_ev[q & 1], S3));
      }
      if (y != nullptr) {
        if (S2 != S3 || !upd) TGP_TRY(st_wait(ctx, S2, ctx->ev_d));
        TGP_TRY(launch_trsv_fwd_step<T>(ctx, S2, mb, Ljj, ld, dj, y + j0));
      }
      if (need_mid) TGP_TRY(mid(ctx->ev_d));
    }
    return TGP_OK;
  }
  // Two-level panel (ctx option sub_panel): the rank-128 updates behind a block stay inside its sub-panel, and the
  // last block of a sub-panel is followed -- on the chain's own stream, it gates the next potf2 -- by ONE update of
  // the panel's remaining columns with K = sub_panel.  C traffic and flops of the bandwidth-hungry K = 128 kernel
  // fall by 57 % (NB = 1024, sub-panels of 512: 12 + 16 instead of 28 units of 128 x 128 column blocks per row).
  int64_t SB = ctx->sub_panel;
  if (SB < 2 * TILE || SB >= kb || kb % SB != 0 || n - k0 < ctx->sub_panel_min_rows) SB = 0;
  for (int64_t j0 = j_first; j0 < j_stop; j0 += TILE) {
    T* Ljj = A + j0 * ld + j0;
    T* dj = dinv + (j0 / TILE) * 2048;
    const int64_t sub_end = SB > 0 ? k0 + ((j0 - k0) / SB + 1) * SB : k0 + kb;  // end of this block's sub-panel
    const bool sub_first = SB > 0 && j0 > k0 && (j0 - k0) % SB == 0;  // its pending update was the K = SB product
    const bool pend = j0 > k0 && !sub_first;  // in-panel update from block column j0-128 still in flight
    if (pend || sub_first || !head_done) TGP_TRY(panel_potf2<T>(ctx, st, A, ld, dinv, pivot_off, j0, pend));
    if (pend) TGP_TRY(st_wait(ctx, st, ctx->ev_e));  // rest of that update
    const int64_t mb = n - (j0 + TILE);
    if (mb > 0) TGP_TRY(launch_trsm<T>(ctx, st, mb, Ljj, ld, dj, Ljj + TILE, ld));
    const int64_t nc = sub_end - (j0 + TILE);
    const bool upd = mb > 0 && nc > 0;
    const bool need_mid = after_blocks > 0 && mb > 0 && (k0 + kb) - (j0 + TILE) > 0 &&
                          j0 + TILE == k0 + after_blocks * TILE;
    // one marker behind the trsm serves both side streams (every marker between two
    // kernels of the chain costs it a few microseconds)
    if (y != nullptr || upd || need_mid) TGP_TRY(ev_record(ctx, ctx->ev_d, st));
    if (upd) {
      TGP_TRY(st_wait(ctx, S3, ctx->ev_d));
      TGP_TRY(launch_gemm_nt<T>(ctx, S3, mb, nc, TILE, Ljj + TILE, ld, Ljj + TILE, ld,
                                A + (j0 + TILE) * ld + j0 + TILE, ld, 1, 0, 3));
      TGP_TRY(ev_record(ctx, ctx->ev_e, S3));
    }
    if (y != nullptr) {
      if (S2 != S3 || !upd) TGP_TRY(st_wait(ctx, S2, ctx->ev_d));  // (already waited for by the update)
      TGP_TRY(launch_trsv_fwd_step<T>(ctx, S2, mb, Ljj, ld, dj, y + j0));
    }
    if (need_mid) TGP_TRY(mid(ctx->ev_d));
    if (SB > 0 && j0 + TILE == sub_end && sub_end < k0 + kb) {
      // the sub-panel [sub_end - SB, sub_end) is final (its last trsm is the previous kernel of this stream):
      // A[sub_end.., sub_end..k0+kb) -= P P^T, P = its rows from sub_end down; latency-bound like the gate
      const int64_t s0 = sub_end - SB, mr = n - sub_end, ncr = k0 + kb - sub_end;
      const T* P = A + s0 * ld + sub_end;
      const int64_t tiles = (mr / TILE) * (ncr / TILE) - (ncr / TILE) * (ncr / TILE - 1) / 2;
      TGP_TRY(launch_gemm_nt<T>(ctx, st, mr, ncr, SB, P, ld, P, ld, A + sub_end * ld + sub_end, ld, 1, 0,
                                tiles <= ctx->first_small_tiles ? 4 : 1));
    }
  }
  return TGP_OK;
}

// y != nullptr: also overwrite y (n, zero padded) with L^-1 y.  Block column j of L is final
// once its trsm has run, so the forward substitution step j (diagonal solve + update of the
// rows below) is issued on a third stream right behind it and hides under the
// factorisation -- log_probability needs no separate triangular-solve pass.
template <typename T>
int potrf(tgp_ctx* ctx, int64_t n, T* A, int64_t ld, T* dinv, int32_t* info_host, T* y) {
  TGP_ARG_CHECK(n % TILE == 0 && ld >= n, "potrf: n must be a multiple of %d and ld >= n", TILE);
  if (info_host) *info_host = 0;
  if (n == 0) return TGP_OK;
  // (hints a previous call may have left behind an error return: each is consumed by the launch it was set for)
  ctx->potf2_wait_counter = nullptr;
  ctx->prefix_hint_cols = 0;
  ctx->prefix_hint_counter = nullptr;
  ctx->reserve_hint = 0;
  hipStream_t S0 = ctx->stream, S1 = ctx->panel_stream;
  if (ctx->solve_on_update == 0 && y != nullptr) TGP_TRY(ensure_solve_stream(ctx));
  hipStream_t S2 = ctx->solve_on_update != 0 ? ctx->update_stream : ctx->solve_stream;
  if (y != nullptr) {  // S2 must see y (uploaded on S0)
    TGP_TRY(ev_record(ctx, ctx->ev_c, S0));
    TGP_TRY(st_wait(ctx, S2, ctx->ev_c));
  }
  int64_t NB = ctx->nb_outer;
  if (NB < TILE) NB = TILE;
  NB = NB / TILE * TILE;
  if (!ctx->trace) TGP_HIP_TRY(hipMemsetAsync(ctx->d_info, 0, sizeof(int32_t), S0));
  ctx->chain_stamp_base = 0;
  ctx->chain_launches = 0;
  ctx->chain_polls_pending = false;
  const bool prof_on = ctx->profile != 0 && !ctx->trace;
  std::vector<ProfSpan> spans;
  ctx->ev_used = 0;
  // common time base of the launch spans: their UNION (launches of the same kernel on the main and the priority stream run
  // beside each other at large N) is what the bench divides the flops by; TGP_SPAN_DUMP=1 also prints where each launch sits
  hipEvent_t prof_t0 = nullptr;
  const bool span_dump = prof_on && getenv("TGP_SPAN_DUMP") != nullptr;
  if (prof_on) {
    TGP_TRY(prof_event(ctx, &prof_t0));
    TGP_HIP_TRY(hipEventRecord(prof_t0, S0));
  }

  // Panel = 128-column blocks (panel_chain above).  `head_done`: the first block's potf2 was
  // already issued by the caller.
  auto potf2_at = [&](hipStream_t st, int64_t j0, bool pend) -> int {
    return panel_potf2<T>(ctx, st, A, ld, dinv, 0, j0, pend);
  };
  auto panel = [&](hipStream_t st, int64_t k0, int64_t kb, bool head_done, int64_t after_blocks,
                   const std::function<int(hipEvent_t)>& mid) -> int {
    return panel_chain<T>(ctx, st, n, A, ld, dinv, 0, k0, kb, head_done, y, after_blocks, mid);
  };
  const std::function<int(hipEvent_t)> no_mid = [](hipEvent_t) { return TGP_OK; };
  auto trailing = [&](hipStream_t sq, int64_t m, int64_t nn, int64_t kb, const T* P, T* C, int role) -> int {
    ProfSpan sp{};
    const bool prof = prof_on && role != 4 && role != 5;  // spans time the 128x128-tile kernel only
    if (prof) {
      TGP_TRY(prof_event(ctx, &sp.e0));
      TGP_TRY(prof_event(ctx, &sp.e1));
      TGP_HIP_TRY(hipEventRecord(sp.e0, sq));
    }
    TGP_TRY(launch_gemm_nt<T>(ctx, sq, m, nn, kb, P, ld, P, ld, C, ld, 1, 0, role));
    if (prof) {
      TGP_HIP_TRY(hipEventRecord(sp.e1, sq));
      // algorithmic flops of the lower-trapezoid update: entries (i >= j) x 2 kb
      const double entries = double(nn) * double(m) - double(nn) * double(nn - 1) / 2.0;
      sp.flops = 2.0 * entries * double(kb);
      spans.push_back(sp);
    }
    return TGP_OK;
  };

  // columns right of the first panel may still be in assembly (capi.hip, factor_impl)
  auto join_assembly = [&]() -> int {
    TGP_TRY(run_deferred_asm(ctx, nullptr));
    if (ctx->asm_pending) {
      TGP_TRY(st_wait(ctx, S0, ctx->ev_asm));
      ctx->asm_pending = false;
    }
    return TGP_OK;
  };
  // panel width: 2 NB (half as many read-modify-write passes over the trailing matrix) while at
  // least `nb_wide_rows` rows are left -- there the evaluation is bound by the trailing update, not
  // by the chain -- and never for the first panel, whose chain nothing hides
  auto width = [&](int64_t k0) -> int64_t {
    const int64_t rem = n - k0;
    int64_t w = NB;
    if (k0 == 0 && ctx->nb_first >= TILE) w = ctx->nb_first / TILE * TILE;  // the first chain hides behind nothing
    if (ctx->nb_wide_rows > 0 && k0 > 0 && rem >= ctx->nb_wide_rows) w = 2 * NB;
    // persistent chain: with few enough rows left the WHOLE rest is one launch (no gate, no big update, no panel
    // boundary any more: the chain's own update tasks are all the work there is)
    if (ctx->chain_kernel != 0 && rem <= ctx->chain_full_rows && rem / TILE <= 64) w = rem;
    return rem < w ? rem : w;
  };
  // TWO-LEVEL PANEL for the persistent chain (round 6, merged schedule; ctx option chain_sub_panel): the chain of a panel runs
  // sub-panel by sub-panel -- its in-panel rank-128 update tasks stay inside `chain_sub_panel` columns -- and every finished
  // sub-panel updates the panel's remaining columns with ONE K = chain_sub_panel product on the tiled MFMA kernel (the same
  // stream: chain | product | potf2 | chain ...).  With 4-column sub-panels 16 of a row tile's 28 update tasks per panel
  // (MFMA duty 0.41 inside the chain, one read-modify-write of the tile per K = 128) become 4 tiles of that product.  The
  // panel's first diagonal block is factored by the caller.
  auto chain_of_panel = [&](hipStream_t st, int64_t k0, int64_t w) -> int {
    const int64_t sp = ctx->chain_sub_panel / TILE * TILE;
    if (sp >= TILE && sp < w && w % sp == 0 && n - k0 > w && n - k0 >= ctx->chain_sub_min_rows) {
      for (int64_t o = 0; o < w; o += sp) {
        if (o > 0) TGP_TRY(potf2_at(st, k0 + o, false));
        TGP_TRY(panel(st, k0 + o, sp, true, 0, no_mid));
        const int64_t c0 = k0 + o + sp;
        if (c0 < k0 + w)
          TGP_TRY(trailing(st, n - c0, k0 + w - c0, sp, A + (k0 + o) * ld + c0, A + c0 * ld + c0, (int)ctx->chain_sub_role));
      }
      return TGP_OK;
    }
    return panel(st, k0, w, true, 0, no_mid);
  };
  const bool la = ctx->lookahead != 0 && S1 != nullptr;
  if (la && ctx->chain_kernel != 0 && ctx->chain_merged != 0) {
    // MERGED TRAILING UPDATE (round 6).  Panel p updates EVERYTHING to its right in ONE launch of the 128 x 128-tile kernel,
    //   T(p):  A[next.., next..] -= P_p P_p^T   (lower, K = the panel's width),
    // whose tile ids start with the block column of panel p+1 (the PREFIX: stored write-through and counted, gemm.hip); the
    // priority stream's [potf2 | chain] of panel p+1 follows a poll of that count -- the first thing that potf2 launch does
    // (potf2_kernel) --, runs beside the rest of T(p), and T(p+1) waits for it.  The depth-2 schedule below cuts the same work into gate(p) | pre(p) | rest(p): two of the three on
    // the 64 x 64-tile kernel (a third of the flops at a lower MFMA duty), 26 launch ramps on the main stream instead of 13,
    // the next gate queued behind a whole rest(p-1).  The last panel (the one-launch tail) needs the whole update: no prefix.
    std::vector<int64_t> s0;
    for (int64_t k0 = 0; k0 < n; k0 += width(k0)) s0.push_back(k0);
    s0.push_back(n);
    const int64_t P = (int64_t)s0.size() - 1;
    hipEvent_t ev_chain[2] = {ctx->ev_b, ctx->ev_g1}, ev_pfx[2] = {ctx->ev_g2, ctx->ev_e}, ev_T[2] = {ctx->ev_i, ctx->ev_j};
    const bool asm_side = ctx->asm_pending;  // columns right of the first panel are still being assembled
    ctx->asm_pending = false;
    if (P == 1 && !asm_side) {
      // the whole matrix is ONE chain launch (N <= chain_full_rows): nothing to run beside it -- on the main stream, without the
      // two cross-stream event hops (12 + 17 us of the 0.42-ms evaluation at N = 1 024: profiles/r06_h)
      TGP_TRY(panel(S0, 0, s0[1] - s0[0], false, 0, no_mid));
      TGP_TRY(run_deferred_asm(ctx, nullptr));
    } else {
    TGP_TRY(ev_record(ctx, ctx->ev_a, S0));
    TGP_TRY(st_wait(ctx, S1, ctx->ev_a));
    TGP_TRY(panel(S1, 0, s0[1] - s0[0], false, 0, no_mid));
    TGP_TRY(run_deferred_asm(ctx, nullptr));
    TGP_TRY(ev_record(ctx, ev_chain[0], S1));
    for (int64_t p = 0; p + 1 < P; ++p) {
      const int64_t kb = s0[p + 1] - s0[p], next = s0[p + 1], wn = s0[p + 2] - s0[p + 1], mt = n - next;
      const T* Pp = A + s0[p] * ld + next;
      T* C = A + next * ld + next;
      // main stream: T(p), behind chain(p) (and T(p-1), by stream order)
      TGP_TRY(st_wait(ctx, S0, ev_chain[p & 1]));
      if (p == 0 && asm_side) TGP_TRY(st_wait(ctx, S0, ctx->ev_asm));
      // the chain of panel p+1 behind the prefix -- unless it covers every column (the tail), the poll is switched off
      // (chain_polls = 0: the default under a counter-collecting profiler, which runs kernels one at a time)
      const bool prefix = wn < mt && ctx->chain_polls != 0;
      int32_t* counter = ctx->d_chain_ticket + CHAIN_PREFIX_OFF + (p & 3);
      if (prefix) {
        if (!ctx->trace) TGP_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int32_t), S0));
        TGP_TRY(ev_record(ctx, ev_pfx[p & 1], S0));  // (the counter is zero: the poller may start)
        ctx->prefix_hint_cols = wn;
        ctx->prefix_hint_counter = counter;
      }
      const int64_t t = mt / TILE;
      if (t * (t + 1) / 2 <= ctx->reserve_max_tiles) ctx->reserve_hint = ctx->chain_reserve;
      TGP_TRY(trailing(S0, mt, mt, kb, Pp, C, 0));
      if (!prefix) TGP_TRY(ev_record(ctx, ev_T[p & 1], S0));
      // priority stream: potf2 + chain of panel p+1
      if (prefix) {
        const int64_t tm = mt / TILE, tp = wn / TILE;
        TGP_TRY(st_wait(ctx, S1, ev_pfx[p & 1]));
        // the poll: the first thing the panel's potf2 launch does (potf2_kernel); a kernel of its own in the traced replay
        // (tests/test_schedule.py models the poll as record 13) and with chain_polls = 2 (round 6's first form)
        if (ctx->trace || ctx->chain_polls == 2) {
          TGP_TRY(launch_prefix_poll(ctx, S1, counter, tp * tm - tp * (tp - 1) / 2, trace_off(ctx, C), ld));
        } else {
          ctx->potf2_wait_counter = counter;
          ctx->potf2_wait_target = tp * tm - tp * (tp - 1) / 2;
        }
      } else {
        TGP_TRY(st_wait(ctx, S1, ev_T[p & 1]));
      }
      if (p == 0 && asm_side) TGP_TRY(st_wait(ctx, S1, ctx->ev_asm));
      TGP_TRY(potf2_at(S1, next, false));
      TGP_TRY(chain_of_panel(S1, next, wn));
      TGP_TRY(ev_record(ctx, ev_chain[(p + 1) & 1], S1));
    }
    TGP_TRY(st_wait(ctx, S0, ev_chain[(P - 1) & 1]));
    }
  } else
  if (la && ctx->chain_kernel != 0 && ctx->chain_depth2 != 0) {
    // Persistent chain, depth-2 schedule: the CHAIN PIPELINE -- gate(p): panel p applied to the columns of panel
    // p+1, then potf2 + the chain launch of panel p+1 -- lives on the priority stream and depends on the main stream
    // only through pre(p-1), the small update that brought those columns up to panel p-1; the main stream carries
    // nothing but trailing updates, pre(p) (panel p -> columns of panel p+2) and rest(p) (panel p -> everything
    // right of that), back to back.  Until round 3 gate(p) sat on the main stream between rest(p-1) and rest(p)
    // with potf2 and two event hops around it: ~0.55 ms per panel of a chip that had nothing else to run
    // (profiles/r04_c).  Same idea as the block-column driver's pipeline (dist.hip).  The early share of the gate is
    // gone with it: while updates dominate the gate hides beside rest(p-1), and the chain-bound tail is ONE launch.
    std::vector<int64_t> s0;
    for (int64_t k0 = 0; k0 < n; k0 += width(k0)) s0.push_back(k0);
    s0.push_back(n);
    const int64_t P = (int64_t)s0.size() - 1;
    hipEvent_t ev_chain[2] = {ctx->ev_b, ctx->ev_g1}, ev_pre[2] = {ctx->ev_g2, ctx->ev_e};
    auto first_role = [&](int64_t m, int64_t nn) -> int {
      const int64_t tiles = (m / TILE) * (nn / TILE) - (nn / TILE) * (nn / TILE - 1) / 2;
      return tiles <= ctx->first_small_tiles ? 4 : 0;
    };
    const bool asm_side = ctx->asm_pending;  // columns right of the first panel are still being assembled
    ctx->asm_pending = false;
    TGP_TRY(ev_record(ctx, ctx->ev_a, S0));
    TGP_TRY(st_wait(ctx, S1, ctx->ev_a));
    TGP_TRY(panel(S1, 0, s0[1] - s0[0], false, 0, no_mid));
    TGP_TRY(run_deferred_asm(ctx, nullptr));  // (whatever form the first panel took: ev_asm is recorded from here on)
    TGP_TRY(ev_record(ctx, ev_chain[0], S1));
    for (int64_t p = 0; p + 1 < P; ++p) {
      const int64_t kb = s0[p + 1] - s0[p], next = s0[p + 1], wn = s0[p + 2] - s0[p + 1], mt = n - next;
      // main stream: panel p is final -> pre(p), rest(p)
      // (chain-bound panels only, and only in the host order that records the marker first)
      const int64_t t3p = p + 3 <= P ? (n - s0[p + 3 < P ? p + 3 : P]) / TILE : 0;
      const bool pre_waits = ctx->chain_pre_wait != 0 && ctx->chain_depth2 != 2 &&
                             t3p * (t3p + 1) / 2 <= ctx->reserve_max_tiles;
      auto main_part = [&]() -> int {
        TGP_TRY(st_wait(ctx, S0, ev_chain[p & 1]));
        if (p == 0 && asm_side) TGP_TRY(st_wait(ctx, S0, ctx->ev_asm));
        if (p + 2 < P) {
          // (Round 6, measured and removed, profiles/r06_g: pre(p) on a high-priority side stream BESIDE rest(p) -- both wait
          // for chain(p) and rest(p-1), they touch different columns -- so that the main stream carries rest launches only
          // and the next gate does not wait for a launch queued behind a whole trailing update: pre then takes 0.74-1.3 ms
          // instead of 0.34 and c2 27.2 instead of 25.8 ms.)
          if (pre_waits) TGP_TRY(st_wait(ctx, S0, ctx->ev_h));  // behind the next panel's first potf2 (see tgp_common.h)
          const int64_t next2 = s0[p + 2], wn2 = s0[p + 3] - s0[p + 2], mt2 = n - next2;
          TGP_TRY(trailing(S0, mt2, wn2, kb, A + s0[p] * ld + next2, A + next2 * ld + next2, first_role(mt2, wn2)));
          TGP_TRY(ev_record(ctx, ev_pre[p & 1], S0));
          const int64_t next3 = s0[p + 3], m3 = n - next3;
          if (m3 > 0) {
            const int64_t t3 = m3 / TILE;
            if (t3 * (t3 + 1) / 2 <= ctx->reserve_max_tiles) ctx->reserve_hint = ctx->chain_reserve;
            TGP_TRY(trailing(S0, m3, m3, kb, A + s0[p] * ld + next3, A + next3 * ld + next3, 0));
          }
        }
        return TGP_OK;
      };
      // priority stream: gate(p), then the chain of panel p+1
      auto chain_part = [&]() -> int {
        if (p >= 1) TGP_TRY(st_wait(ctx, S1, ev_pre[(p - 1) & 1]));
        if (p == 0 && asm_side) TGP_TRY(st_wait(ctx, S1, ctx->ev_asm));
        const int grole = first_role(mt, wn);
        // chain_gate_split: the next panel's first diagonal block on the update stream (idle with the persistent chain once
        // the forward steps are chain tasks) beside the gate -- its own 128 x 128 x kb product, then potf2; the gate skips it
        const bool gsplit = ctx->chain_gate_split != 0 && grole == 4 && mt > TILE && ctx->update_stream != nullptr &&
                            (y == nullptr || ctx->chain_fwd_tasks != 0 || ctx->solve_on_update == 0);
        if (gsplit) {
          hipStream_t S3 = ctx->update_stream;
          TGP_TRY(ev_record(ctx, ctx->ev_i, S1));  // (everything the gate waits for)
          TGP_TRY(st_wait(ctx, S3, ctx->ev_i));
          TGP_TRY(trailing(S3, TILE, TILE, kb, A + s0[p] * ld + next, A + next * ld + next, 4));
          TGP_TRY(potf2_at(S3, next, false));
          TGP_TRY(ev_record(ctx, ctx->ev_j, S3));
          if (pre_waits) TGP_TRY(ev_record(ctx, ctx->ev_h, S3));
          TGP_TRY(trailing(S1, mt, wn, kb, A + s0[p] * ld + next, A + next * ld + next, 5));
          TGP_TRY(st_wait(ctx, S1, ctx->ev_j));
        } else {
          TGP_TRY(trailing(S1, mt, wn, kb, A + s0[p] * ld + next, A + next * ld + next, grole));
          TGP_TRY(potf2_at(S1, next, false));
          if (pre_waits) TGP_TRY(ev_record(ctx, ctx->ev_h, S1));
        }
        TGP_TRY(panel(S1, next, wn, true, 0, no_mid));
        TGP_TRY(ev_record(ctx, ev_chain[(p + 1) & 1], S1));
        return TGP_OK;
      };
      // host order (chain_depth2 = 2: the main stream's calls first).  Either order is race-free -- each part waits
      // only for events of EARLIER steps -- but which stream's kernels reach the device first shapes the start of
      // the evaluation: the chain pipeline ahead of the first big update, or beside it
      if (ctx->chain_depth2 == 2) {
        TGP_TRY(main_part());
        TGP_TRY(chain_part());
      } else {
        TGP_TRY(chain_part());
        TGP_TRY(main_part());
      }
    }
    TGP_TRY(st_wait(ctx, S0, ev_chain[(P - 1) & 1]));
  } else
  if (!la) {
    TGP_TRY(join_assembly());
    for (int64_t k0 = 0, kb = 0; k0 < n; k0 += kb) {
      kb = ctx->chain_kernel != 0 ? width(k0) : ((n - k0 < NB) ? (n - k0) : NB);
      TGP_TRY(panel(S0, k0, kb, false, 0, no_mid));
      const int64_t next = k0 + kb, mt = n - next;
      if (mt > 0) TGP_TRY(trailing(S0, mt, mt, kb, A + k0 * ld + next, A + next * ld + next, 0));
    }
  } else {
    int64_t k_done = 0;
    // With fewer 128x128 tiles than ~2 rounds of workgroup slots the block-column update is
    // a round of long serial k-loops, two per CU on some CUs: 64x64 tiles spread it evenly.
    auto first_role = [&](int64_t m, int64_t nn) -> int {
      const int64_t tiles = (m / TILE) * (nn / TILE) - (nn / TILE) * (nn / TILE - 1) / 2;
      return tiles <= ctx->first_small_tiles ? 4 : 0;
    };
    {  // first panel: on the side stream too, so that the main stream can take the early share
      const int64_t kb0 = width(0), mt0 = n - kb0;
      const int64_t kbn0 = width(kb0);
      const int64_t split = (mt0 > 0 && ctx->first_split > 0 && ctx->first_split < kb0 / TILE)
                                ? ctx->first_split : 0;
      const std::function<int(hipEvent_t)> early = [&](hipEvent_t ready) -> int {
        TGP_TRY(join_assembly());
        TGP_TRY(st_wait(ctx, S0, ready));
        TGP_TRY(trailing(S0, mt0, kbn0, split * TILE, A + kb0, A + kb0 * ld + kb0, first_role(mt0, kbn0)));
        k_done = split * TILE;
        return TGP_OK;
      };
      TGP_TRY(ev_record(ctx, ctx->ev_a, S0));
      TGP_TRY(st_wait(ctx, S1, ctx->ev_a));
      TGP_TRY(panel(S1, 0, kb0, false, split, early));
      TGP_TRY(ev_record(ctx, ctx->ev_b, S1));
      TGP_TRY(st_wait(ctx, S0, ctx->ev_b));
      TGP_TRY(join_assembly());
    }
    for (int64_t k0 = 0, kb = 0; k0 < n; k0 += kb) {
      kb = width(k0);
      const int64_t next = k0 + kb, mt = n - next;
      if (mt <= 0) break;
      const int64_t kbn = width(next);
      const T* P = A + k0 * ld + next;
      // 1. block column of the next panel first ...
      // (`k_done` columns of this panel were already applied while its last blocks were
      // being factored -- see `early` below.)
      const bool gsplit = ctx->gate_split != 0 && ctx->fused_step != 0 && kbn >= 3 * TILE && mt > 2 * TILE;
      if (!gsplit) {
        TGP_TRY(trailing(S0, mt, kbn, kb - k_done, P + k_done * ld, A + next * ld + next, first_role(mt, kbn)));
        // the panel's first potf2 goes in front of the big update on the main stream: issued
        // beside it, it waits a whole round of tiles (~0.3 ms) for a free CU
        TGP_TRY(potf2_at(S0, next, false));
        TGP_TRY(ev_record(ctx, ctx->ev_a, S0));
      } else {
        // Only column block 0 of the next panel gates its chain; blocks 1 and 2.. follow beside the
        // chain's first two steps (the fused step of block 1 waits for ev_g1, the far update of block 0
        // -- and through it step 2 -- for ev_g2).
        const T* Pk = P + k_done * ld;
        const int64_t kk = kb - k_done;
        T* C0 = A + next * ld + next;
        TGP_TRY(trailing(S0, mt, TILE, kk, Pk, C0, first_role(mt, TILE)));
        TGP_TRY(potf2_at(S0, next, false));
        TGP_TRY(ev_record(ctx, ctx->ev_a, S0));
        TGP_TRY(trailing(S0, mt - TILE, TILE, kk, Pk + TILE, C0 + TILE * ld + TILE, first_role(mt - TILE, TILE)));
        TGP_TRY(ev_record(ctx, ctx->ev_g1, S0));
        TGP_TRY(trailing(S0, mt - 2 * TILE, kbn - 2 * TILE, kk, Pk + 2 * TILE, C0 + 2 * TILE * ld + 2 * TILE,
                         first_role(mt - 2 * TILE, kbn - 2 * TILE)));
        TGP_TRY(ev_record(ctx, ctx->ev_g2, S0));
      }
      k_done = 0;
      // 2. ... the main stream updates the rest (enqueued first: the ~90 API calls of a
      // panel take the host longer than a small update takes the GPU) ...
      const int64_t m2 = mt - kbn;
      if (m2 > 0) {
        // the next panel's chain runs beside this launch; when the launch is the shorter of the two it
        // leaves workgroup slots to the chain
        const int64_t t2 = m2 / TILE;
        if (t2 * (t2 + 1) / 2 <= ctx->reserve_max_tiles) ctx->reserve_hint = ctx->chain_reserve;
        TGP_TRY(trailing(S0, m2, m2, kb, P + kbn, A + (next + kbn) * ld + next + kbn, 0));
      }
      // 3. ... while the side stream factors the next panel.  Once its first `first_split`
      // blocks are final, their share of the block-column update that will gate the panel
      // AFTER it is issued behind the running update, so that only the last blocks' share
      // (a quarter of the k-range) is left on the critical path between two chains.
      TGP_TRY(st_wait(ctx, S1, ctx->ev_a));
      const int64_t next2 = next + kbn, mt2 = n - next2;
      const int64_t kbn2 = mt2 > 0 ? width(next2) : 0;
      // (a wide panel leaves as many blocks behind the early share as a normal one)
      const int64_t fs = (ctx->first_split > 0 && kbn > NB) ? ctx->first_split + (kbn - NB) / TILE
                                                            : ctx->first_split;
      const int64_t split = (mt2 > 0 && fs > 0 && fs < kbn / TILE) ? fs : 0;
      const std::function<int(hipEvent_t)> early = [&](hipEvent_t ready) -> int {
        TGP_TRY(st_wait(ctx, S0, ready));
        TGP_TRY(trailing(S0, mt2, kbn2, split * TILE, A + next * ld + next2, A + next2 * ld + next2,
                         first_role(mt2, kbn2)));
        k_done = split * TILE;
        return TGP_OK;
      };
      ctx->gate_pending = gsplit;
      TGP_TRY(panel(S1, next, kbn, true, split, early));
      TGP_TRY(ev_record(ctx, ctx->ev_b, S1));
      TGP_TRY(st_wait(ctx, S0, ctx->ev_b));
    }
  }
  if (y != nullptr) {
    TGP_TRY(ev_record(ctx, ctx->ev_c, S2));
    TGP_TRY(st_wait(ctx, S0, ctx->ev_c));
  }
  int32_t info = 0;
  ctx->submitted = std::chrono::steady_clock::now();
  // Round 6: the fused evaluation (capi.hip, factor_body) reads `info` together with its two scalars behind the reductions
  // it queues next -- ONE host round trip per evaluation instead of two (the first cost ~35 us between the last chain task
  // and the first reduction: 8 % of an evaluation at N = 1 024).  It asks for that with ctx->defer_join; profiled passes
  // (event pairs are read below) and passes with stream wait-values in flight (joined with a deadline) keep the join here.
  if (ctx->defer_join && !ctx->trace && !prof_on && !(ctx->wait_values_inflight && ctx->host_join != 0)) {
    ctx->wait_values_inflight = false;
    ctx->join_deferred = true;
    if (info_host) *info_host = 0;
    return TGP_OK;
  }
  if (!ctx->trace) {
    TGP_HIP_TRY(hipMemcpyAsync(&info, ctx->d_info, sizeof(int32_t), hipMemcpyDeviceToHost, S0));
    if (ctx->wait_values_inflight && ctx->host_join != 0) {
      ctx->wait_values_inflight = false;
      TGP_TRY(join_bounded(ctx, S0, n));
    } else {
      ctx->wait_values_inflight = false;
      TGP_HIP_TRY(hipStreamSynchronize(S0));
    }
  }
  if (prof_on) {
    ctx->prof_syrk_ms = 0;
    ctx->prof_syrk_flops = 0;
    ctx->prof_syrk_union_ms = 0;
    ctx->prof_syrk_launches = (int64_t)spans.size();
    std::vector<std::pair<double, double>> iv;  // [start, end) of every launch on the common time base
    for (auto& sp : spans) {
      float ms = 0, at = 0;
      TGP_HIP_TRY(hipEventElapsedTime(&ms, sp.e0, sp.e1));
      TGP_HIP_TRY(hipEventElapsedTime(&at, prof_t0, sp.e0));
      ctx->prof_syrk_ms += ms;
      ctx->prof_syrk_flops += sp.flops;
      iv.emplace_back(double(at), double(at) + double(ms));
      if (span_dump)
        fprintf(stderr, "span at %8.3f ms  + %7.3f ms  %8.2f GF  %5.1f TF/s\n", at, ms, sp.flops * 1e-9, sp.flops / ms * 1e-9);
    }
    // length of the union of the intervals: the time during which AT LEAST ONE trailing-update launch was running
    std::sort(iv.begin(), iv.end());
    double hi = -1e300;
    for (const auto& q : iv) {
      if (q.first > hi) {
        ctx->prof_syrk_union_ms += q.second - q.first;
        hi = q.second;
      } else if (q.second > hi) {
        ctx->prof_syrk_union_ms += q.second - hi;
        hi = q.second;
      }
    }
  }
  if (info == STEP_TIMEOUT) {
    set_error("potrf: a device-side hand-off did not arrive within poll_timeout_ms (a lost or starved chain launch)");
    return TGP_E_TIMEOUT;
  }
  if (info_host) *info_host = info;
  return info > 0 ? info : TGP_OK;
}

// winv: [W_b | W_b^T | tf_b = W_b L[b, b-1]], each b = 0..n/128), 128 x 128 column-major
template <typename T>
int compute_winv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, T* winv) {
  if (n == 0) return TGP_OK;
  const int64_t sec = (n / TILE) * 16384;
  hipLaunchKernelGGL((winv_kernel<T>), dim3((unsigned)(n / TILE)), dim3(128), 0, ctx->stream, (int)(n / TILE), L, ld,
                     winv, winv + sec, winv + 2 * sec, winv + 3 * sec);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int trsv(tgp_ctx* ctx, int64_t n, const T* L, int64_t ld, const T* dinv, int transpose, T* y,
         const T* winv, const T* yin) {
  TGP_ARG_CHECK(n % TILE == 0, "trsv: n must be a multiple of %d", TILE);
  TGP_ARG_CHECK(yin == nullptr || (winv != nullptr && yin != y), "trsv: a separate right-hand side needs the streaming solve");
  hipStream_t st = ctx->stream;
  const int64_t nb = n / TILE;
  if (winv != nullptr && n > 0) {  // one streaming launch (trsv_fwd_stream_kernel / trsv_bwd_stream_kernel)
    // workgroups per block row of the forward solve (ctx option trsv_groups, 2..8; 0 = by size, measured in
    // profiles/r06_b: N = 4 096 0.079 / 0.081 / 0.087 ms with 3 / 4 / 6, N = 16 384 0.367 / 0.288 / 0.284, N = 65 536 3.22 / 3.03 / 2.98)
    const int G_auto = nb <= 40 ? 3 : (nb <= 256 ? 4 : 6);
    const int G = transpose ? 2 : (ctx->trsv_groups <= 0 ? G_auto : (int)std::min<int64_t>(std::max<int64_t>(ctx->trsv_groups, 2), 8));
    TGP_TRY(ensure_work(ctx, size_t(G) * size_t(n) * sizeof(T) + 1024));
    T* tmp = static_cast<T*>(ctx->d_work);
    int32_t* ticket = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->d_work) + size_t(n) * sizeof(T) + 64);
    T* part = static_cast<T*>(ctx->d_work) + n + 64;  // the helpers' partial sums, (G - 1) x n entries
    // yin != NULL: the right-hand side stays where it is (the resident residual of log_probability: no copy into the work
    // vector in front of the solve, no copy inside the prep kernel: 0.294 -> 0.286 ms at N = 16 384, profiles/r06_f section 8)
    const T* rhs = yin != nullptr ? yin : (const T*)tmp;
    hipLaunchKernelGGL((trsv_prep_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, y,
                       yin != nullptr ? (T*)nullptr : tmp, ticket, part, G - 1);
    if (!transpose)
      hipLaunchKernelGGL((trsv_fwd_stream_kernel<T>), dim3((unsigned)(G * nb)), dim3(256), 0, st, (int)nb, L, ld, winv,
                         winv + 2 * nb * 16384, winv + 3 * nb * 16384, rhs, y, ticket, part, G);
    else
      hipLaunchKernelGGL((trsv_bwd_stream_kernel<T>), dim3((unsigned)(2 * nb)), dim3(512), 0, st, (int)nb, L, ld,
                         winv + nb * 16384, rhs, y, ticket, part);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  }
  if (!transpose) {
    for (int64_t kb = 0; kb < nb; ++kb) {
      const int64_t j0 = kb * TILE;
      TGP_TRY(launch_trsv_fwd_step<T>(ctx, st, n - (j0 + TILE), L + j0 * ld + j0, ld, dinv + kb * 2048,
                                      y + j0));
    }
  } else {
    for (int64_t kb = nb - 1; kb >= 0; --kb) {
      const int64_t j0 = kb * TILE;
      hipLaunchKernelGGL((trsv_diag_bwd_kernel<T>), dim3(1), dim3(128), 0, st, L + j0 * ld + j0, ld,
                         dinv + kb * 2048, y + j0);
      if (j0 > 0)
        hipLaunchKernelGGL((trsv_update_bwd_kernel<T>), dim3((unsigned)((j0 + 31) / 32)), dim3(256),
                           0, st, j0, L + j0, ld, y + j0, y);
    }
  }
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// y (m) -= P (m x k, column-major) x (k); k a multiple of 128.  Thread per row, coalesced.
template <typename T>
int gemv_sub(tgp_ctx* ctx, int64_t m, int64_t k, const T* P, int64_t ld, const T* x, T* y) {
  TGP_ARG_CHECK(k % TILE == 0, "gemv_sub: k must be a multiple of %d", TILE);
  if (m == 0) return TGP_OK;
  for (int64_t c0 = 0; c0 < k; c0 += TILE)
    hipLaunchKernelGGL((trsv_update_fwd_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0,
                       ctx->stream, m, P + c0 * ld, ld, x + c0, y);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// B (m x n) <- B L^-T: the rows of K(X_test, X) L^-T behind condition() / predict(return_var=True)
// (direct.py:94).  Two-level right-looking sweep: per panel of NB columns a chain of [128-column solve + in-panel
// update], then the MFMA update of everything to the right.  Round 3: on TWO streams.  The chain is 8 dependent hops
// of kernels with m / 64 workgroups each -- on one stream the chip idled through 280 us per panel (4.5 of 22 ms at
// m = 4 096, n = 16 384).  Now, with (i)_k the update of the NEXT panel's columns by panel k and (ii)_k the update of
// everything right of that:
//   priority stream:  chain_0, (i)_0, chain_1, (i)_1, ...     main stream:  (ii)_0, (ii)_1, ...
//   (i)_k after chain_k [stream order] and (ii)_{k-1} [event];  (ii)_k after chain_k [event] and (ii)_{k-1} [order].
// The long (ii) launches keep the chip full; the next panel's share and chain run beside them on the priority stream.
template <typename T>
int trsm_right_lt(tgp_ctx* ctx, int64_t m, int64_t n, const T* L, int64_t ldl, const T* dinv, T* B,
                  int64_t ldb) {
  TGP_ARG_CHECK(m % TILE == 0 && n % TILE == 0, "trsm_right_lt: m, n must be multiples of %d", TILE);
  if (m == 0 || n == 0) return TGP_OK;
  hipStream_t st = ctx->stream;
  int64_t NB = ctx->nb_outer;
  if (NB < TILE) NB = TILE;
  NB = NB / TILE * TILE;
  const bool two = ctx->lookahead != 0 && ctx->panel_stream != nullptr && n > 2 * NB;
  hipStream_t cs = two ? ctx->panel_stream : st;  // the chains' stream
  auto chain = [&](int64_t k0, int64_t kb) -> int {
    for (int64_t j0 = k0; j0 < k0 + kb; j0 += TILE) {
      TGP_TRY(launch_trsm<T>(ctx, cs, m, L + j0 * ldl + j0, ldl, dinv + (j0 / TILE) * 2048, B + j0 * ldb, ldb));
      const int64_t nc = (k0 + kb) - (j0 + TILE);
      if (nc > 0)  // B[:, j0+128 .. k0+kb) -= X_j0 * L[j0+128 .. k0+kb, j0 block]^T
        TGP_TRY(launch_gemm_nt<T>(ctx, cs, m, nc, TILE, B + j0 * ldb, ldb, L + j0 * ldl + j0 + TILE, ldl,
                                  B + (j0 + TILE) * ldb, ldb, 0, 0, 1));
    }
    return TGP_OK;
  };
  if (two) {  // the chain stream starts behind whatever produced B on the main stream
    TGP_TRY(ev_record(ctx, ctx->ev_a, st));
    TGP_TRY(st_wait(ctx, cs, ctx->ev_a));
  }
  for (int64_t k0 = 0; k0 < n; k0 += NB) {
    const int64_t kb = (n - k0 < NB) ? (n - k0) : NB;
    TGP_TRY(chain(k0, kb));
    const int64_t next = k0 + kb, nr = n - next;
    if (two) TGP_TRY(ev_record(ctx, ctx->ev_b, cs));  // chain_k done
    if (nr <= 0) break;
    const int64_t kn = (nr < NB) ? nr : NB;  // width of the next panel
    if (!two) {
      TGP_TRY(launch_gemm_nt<T>(ctx, st, m, nr, kb, B + k0 * ldb, ldb, L + k0 * ldl + next, ldl, B + next * ldb,
                                ldb, 0, 0, 1));
      continue;
    }
    // (i)_k on the chain stream, behind (ii)_{k-1} (ev_a, recorded below one iteration earlier / above at entry)
    TGP_TRY(st_wait(ctx, cs, ctx->ev_a));
    TGP_TRY(launch_gemm_nt<T>(ctx, cs, m, kn, kb, B + k0 * ldb, ldb, L + k0 * ldl + next, ldl, B + next * ldb, ldb,
                              0, 0, 1));
    // (ii)_k on the main stream, behind chain_k
    TGP_TRY(st_wait(ctx, st, ctx->ev_b));
    if (nr > kn)
      TGP_TRY(launch_gemm_nt<T>(ctx, st, m, nr - kn, kb, B + k0 * ldb, ldb, L + k0 * ldl + next + kn, ldl,
                                B + (next + kn) * ldb, ldb, 0, 0, 1));
    TGP_TRY(ev_record(ctx, ctx->ev_a, st));
  }
  if (two) TGP_TRY(st_wait(ctx, st, ctx->ev_b));  // the last chain rejoins the main stream
  return TGP_OK;
}

// ---------------------------------------------------------------------------------------
// K^-1 from the factor (gradient path):  L^-1 by halves, then K^-1 = L^-T L^-1.
//
// Round 1-2 swept the right-sided solve over the identity (128 dependent hops of a 128-column solve + a K = 128
// update + a K = 1024 update per panel, one stream: 28.7 ms at N = 16 384) and formed M M^T in a second N^2 buffer
// (33.1 ms: tiles of different k-ranges shared nothing through the L2).  Round 3:
//
//   L = [A 0; B C]  ->  L^-1 = [A^-1 0; -C^-1 B A^-1  C^-1],
//
// bottom-up over aligned blocks of h = 1, 2, 4, ... tiles: the diagonal 128 x 128 inverses are the W_b of the
// streaming solve (winv_kernel), and every level is TWO batched products over all its block pairs at once -- no
// dependent chain at all, 2 log2(N/128) GEMM launches in total, all of them on the 128 x 128-tile MFMA kernel:
//   R = M_A B^T          (M_A = A^-T upper triangular: k from the row tile on, walked from the end),
//   Z = X_C R^T          (X_C = C^-1 lower triangular: k up to the row tile),        X_21 = -Z,  M_12 = -Z^T.
// Both orientations of L^-1 are kept -- the NT kernel wants its triangular operand as "rows x k" -- in ONE buffer S
// of (n + 128) x n:  M = L^-T (upper tiles, diagonal tiles included) at S[a, b],  X = L^-1 (lower tiles) at
// S[a + 128, b]: the two triangles do not overlap, R lives where M_12 will go, and the mirror pass (memory-bound,
// 0.1 ms per level) writes -Z back and -Z^T across.  K^-1 = M M^T then goes to X's place (lower tiles; X is dead),
// so the caller's K^-1 is (S + 128, ld = n + 128) and the gradient needs ONE extra matrix, not two.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void tri_seed_kernel(const T* __restrict__ winv, const T* __restrict__ winvT,
                                                      T* __restrict__ S, int64_t ld) {
  const int b = blockIdx.x;
  T* Mb = S + int64_t(b) * 128 * ld + int64_t(b) * 128;  // M_bb = W_b^T; X_bb = W_b sits 128 rows lower
  const T* w = winv + int64_t(b) * 16384;
  const T* wt = winvT + int64_t(b) * 16384;
  for (int e = threadIdx.x; e < 16384; e += 256) {
    const int r = e & 127, c = e >> 7;
    Mb[int64_t(c) * ld + r] = wt[e];
    Mb[int64_t(c) * ld + 128 + r] = w[e];
  }
}

// Z (rows x cols, at Zb) <- -Z and Mt (cols x rows, at Mb) <- -Z^T, 64 x 64 per workgroup, blockIdx.y = batch
template <typename T>
__global__ __launch_bounds__(256) void tri_mirror_kernel(T* __restrict__ Zb, T* __restrict__ Mb, int64_t ld,
                                                        int rt, int64_t stride) {
  __shared__ T t[64][65];
  T* Z = Zb + int64_t(blockIdx.y) * stride;
  T* M = Mb + int64_t(blockIdx.y) * stride;
  const int bi = blockIdx.x % rt, bj = blockIdx.x / rt;
  const int r = threadIdx.x & 63, c4 = threadIdx.x >> 6;
  T* z = Z + int64_t(bj) * 64 * ld + int64_t(bi) * 64;
#pragma unroll 4
  for (int c = c4; c < 64; c += 4) {
    const T v = -z[int64_t(c) * ld + r];
    z[int64_t(c) * ld + r] = v;
    t[c][r] = v;
  }
  __syncthreads();
  T* m = M + int64_t(bi) * 64 * ld + int64_t(bj) * 64;  // M(col, row) = Z(row, col)
#pragma unroll 4
  for (int c = c4; c < 64; c += 4) m[int64_t(c) * ld + r] = t[r][c];
}

// upper triangle <- transpose of the lower one (n a multiple of 64), 64 x 64 per workgroup, strictly-lower blocks only
template <typename T>
__global__ __launch_bounds__(256) void symmetrize_kernel(T* __restrict__ A, int64_t ld, int nt) {
  __shared__ T t[64][65];
  // block (bi, bj), bi > bj, from the linear index over the strict lower triangle
  const int b = blockIdx.x;
  int bi = int((1.0f + sqrtf(1.0f + 8.0f * float(b))) * 0.5f);
  while ((bi * (bi - 1)) / 2 > b) --bi;
  while (((bi + 1) * bi) / 2 <= b) ++bi;
  const int bj = b - (bi * (bi - 1)) / 2;
  const int r = threadIdx.x & 63, c4 = threadIdx.x >> 6;
  const T* z = A + int64_t(bj) * 64 * ld + int64_t(bi) * 64;
#pragma unroll 4
  for (int c = c4; c < 64; c += 4) t[c][r] = z[int64_t(c) * ld + r];
  __syncthreads();
  T* m = A + int64_t(bi) * 64 * ld + int64_t(bj) * 64;
#pragma unroll 4
  for (int c = c4; c < 64; c += 4) m[int64_t(c) * ld + r] = t[r][c];
  // (diagonal 64 x 64 blocks: the caller's producers write them in full)
}

template <typename T>
int symmetrize_lower(tgp_ctx* ctx, int64_t n, T* A, int64_t ld) {
  TGP_ARG_CHECK(n % 64 == 0, "symmetrize_lower: n must be a multiple of 64");
  const int64_t nt = n / 64, blocks = nt * (nt - 1) / 2;
  if (blocks == 0) return TGP_OK;
  TGP_ARG_CHECK(blocks < (int64_t(1) << 31), "symmetrize_lower: too many blocks");
  hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A, ld, (int)nt);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int spd_inverse_lower(tgp_ctx* ctx, int64_t n, const T* L, int64_t ldl, const T* winv, T* S, int64_t lds) {
  TGP_ARG_CHECK(n % TILE == 0 && lds >= n + TILE, "spd_inverse_lower: n a multiple of %d and lds >= n + %d", TILE,
                TILE);
  if (n == 0) return TGP_OK;
  hipStream_t st = ctx->stream;
  const int64_t nt = n / TILE;
  hipLaunchKernelGGL((tri_seed_kernel<T>), dim3((unsigned)nt), dim3(256), 0, st, winv, winv + nt * 16384, S, lds);
  T* X = S + TILE;  // L^-1, lower tiles
  for (int64_t h = 1; h < nt; h *= 2) {
    const int64_t hw = h * TILE, pair = 2 * hw;
    const int64_t nfull = nt / (2 * h), rem = nt - nfull * 2 * h;
    // the full pairs in one batch, a last pair with a shorter second block on its own
    for (int part = 0; part < 2; ++part) {
      const int64_t batch = part == 0 ? nfull : 1;
      const int64_t h2w = part == 0 ? hw : (rem - h) * TILE;
      if (batch == 0 || h2w <= 0) continue;
      const int64_t r0 = part == 0 ? 0 : nfull * pair;
      const int64_t sS = pair * (lds + 1), sL = pair * (ldl + 1);
      T* R = S + r0 + (r0 + hw) * lds;          // hw x h2w, where M_12 goes
      T* Z = X + (r0 + hw) + r0 * lds;          // h2w x hw, where X_21 goes
      TGP_TRY(launch_gemm_tri<T>(ctx, st, hw, h2w, hw, S + r0 * (lds + 1), lds, L + (r0 + hw) + r0 * ldl, ldl, R, lds,
                                 0, 1 | 4 | 16, int(batch), sS, sL, sS));
      TGP_TRY(launch_gemm_tri<T>(ctx, st, h2w, hw, h2w, X + (r0 + hw) * (lds + 1), lds, R, lds, Z, lds, 0, 1 | 8,
                                 int(batch), sS, sS, sS));
      hipLaunchKernelGGL((tri_mirror_kernel<T>), dim3((unsigned)((h2w / 64) * (hw / 64)), (unsigned)batch), dim3(256),
                         0, st, Z, R, lds, int(h2w / 64), sS);
    }
  }
  // K^-1 = M M^T, lower tiles, into X's place
  TGP_TRY(launch_gemm_tri<T>(ctx, st, n, n, n, S, lds, S, lds, X, lds, 1, 1 | 4 | 16, 1, 0, 0, 0));
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

#define TGP_INST(T)                                                                              \
  template int spd_inverse_lower<T>(tgp_ctx*, int64_t, const T*, int64_t, const T*, T*, int64_t); \
  template int symmetrize_lower<T>(tgp_ctx*, int64_t, T*, int64_t);                               \
  template int launch_potf2<T>(tgp_ctx*, hipStream_t, T*, int64_t, T*, int32_t*, int32_t,       \
                               const T*, int64_t);       \
  template int launch_trsm<T>(tgp_ctx*, hipStream_t, int64_t, const T*, int64_t, const T*, T*,   \
                              int64_t);                                                          \
  template int launch_panel_step<T>(tgp_ctx*, hipStream_t, int64_t, T*, int64_t, T*, int32_t*, int32_t, \
                                    const T*, bool);                                             \
  template int compute_dinv<T>(tgp_ctx*, int64_t, const T*, int64_t, T*);                        \
  template int panel_potf2<T>(tgp_ctx*, hipStream_t, T*, int64_t, T*, int64_t, int64_t, bool);   \
  template int panel_chain<T>(tgp_ctx*, hipStream_t, int64_t, T*, int64_t, T*, int64_t, int64_t, \
                              int64_t, bool, T*, int64_t, const std::function<int(hipEvent_t)>&, int64_t, int64_t); \
  template int potrf<T>(tgp_ctx*, int64_t, T*, int64_t, T*, int32_t*, T*);                           \
  template int trsv<T>(tgp_ctx*, int64_t, const T*, int64_t, const T*, int, T*, const T*, const T*); \
  template int compute_winv<T>(tgp_ctx*, int64_t, const T*, int64_t, T*);                        \
  template int gemv_sub<T>(tgp_ctx*, int64_t, int64_t, const T*, int64_t, const T*, T*);                          \
  template int trsm_right_lt<T>(tgp_ctx*, int64_t, int64_t, const T*, int64_t, const T*, T*,     \
                                int64_t);
TGP_INST(float)
TGP_INST(double)
#undef TGP_INST

}  // namespace tgp
