// kmat.hip -- pairwise-distance + stationary-kernel evaluator (K1/K2/K3/K9 of SURVEY 2a).
//
// Replaces reference kernels/base.py:84-103 (nested vmap of Kernel.evaluate) fused with
// noise.py:77-78 (diagonal scatter-add).  HBM-write bound: every lane owns one ROW of a
// 128x128 tile and walks 64 columns, so each wave store is 512 contiguous bytes of the
// column-major output; the X tiles are staged once through LDS.
//
// Built with -ffp-contract=off: the scalar formulas below keep the reference's operation
// order (kernels/stationary.py:76-235, kernels/distance.py:41-59) so entries agree with a
// NumPy evaluation to the last ulp or two (libm-vs-ocml exp is the only difference).
#include <algorithm>

#include "tgp_common.h"

namespace tgp {

namespace {

template <typename T> struct MathC;
template <> struct MathC<double> {
  static constexpr double SQRT3 = 1.7320508075688772;   // np.sqrt(3)
  static constexpr double SQRT5 = 2.23606797749979;     // np.sqrt(5)
  static constexpr double PI = 3.141592653589793;
  static constexpr double TWO_PI = 6.283185307179586;   // 2 * np.pi
};
template <> struct MathC<float> {
  static constexpr float SQRT3 = 1.7320508075688772f;
  static constexpr float SQRT5 = 2.23606797749979f;
  static constexpr float PI = 3.141592653589793f;
  static constexpr float TWO_PI = 6.283185307179586f;
};

// One leaf of the program (a stationary kernel or a constant) at the pair's r1 = sum|d| and
// r2 = sum d^2.  The distance a leaf does not use is not computed: the metric and the op are
// wave-uniform, so e.g. ExpSquared never pays for the fp64 square root of the L2 distance.
// distance.py:51-56: zero-safe sqrt; distance.py:30-38: L1 "squared" = distance^2
// FAM = 1: the program holds only Constant / Exp / ExpSquared / Matern leaves (checked on the
// host): the cosine, sine and power paths -- and their registers -- are compiled out.
template <typename T, int FAM = 0>
__device__ __forceinline__ T leaf_value(const KProg& kp, int i, T r1, T r2) {
  const int op = kp.op[i];
  const bool l2 = kp.metric[i] == TGP_METRIC_L2;
  auto dist = [&]() -> T { return l2 ? ((r2 == T(0)) ? r1 : sqrt(r2)) : r1; };
  auto sq = [&]() -> T { return l2 ? r2 : r1 * r1; };
  const T p0 = T(kp.p0[i]);
  const T p1 = T(kp.p1[i]);
  switch (op) {
    case TGP_K_CONST: return p0;
    case TGP_K_EXP: return exp(-dist() / p0);
    case TGP_K_EXPSQ: return exp(T(-0.5) * (sq() / (p0 * p0)));
    case TGP_K_M32: { const T a = MathC<T>::SQRT3 * (dist() / p0); return (T(1) + a) * exp(-a); }
    case TGP_K_M52: { const T a = MathC<T>::SQRT5 * (dist() / p0);
                      return (T(1) + a + (a * a) / T(3)) * exp(-a); }
    default: break;
  }
  if constexpr (FAM == 0) {
    switch (op) {
      case TGP_K_COS: return cos(MathC<T>::TWO_PI * (dist() / p0));
      case TGP_K_ESS: { const T s = sin(MathC<T>::PI * (dist() / p0)); return exp(-p1 * (s * s)); }
      case TGP_K_RQ: return pow(T(1) + T(0.5) * (sq() / (p0 * p0)) / p1, -p1);
      default: break;
    }
  }
  (void)p1;
  return T(0);
}

// Evaluate the postfix program for one pair given r1 = sum|d| and r2 = sum d^2.
// Programs of one leaf, or of two leaves and one operator (e.g. `amp**2 * ExpSquared(l)`),
// take a direct path; anything else runs the general stack machine, whose evaluation stack
// lives in 8 named registers (no runtime-indexed array -> no scratch).  All branches are
// wave-uniform (the program sits in kernarg / SGPRs).
template <typename T, int FAM = 0>
__device__ __forceinline__ T eval_kprog(const KProg& kp, T r1, T r2) {
  if (kp.n == 1) return leaf_value<T, FAM>(kp, 0, r1, r2);
  if (kp.n == 3 && kp.op[0] < TGP_K_ADD && kp.op[1] < TGP_K_ADD) {
    const T a = leaf_value<T, FAM>(kp, 0, r1, r2);
    const T b = leaf_value<T, FAM>(kp, 1, r1, r2);
    return (kp.op[2] == TGP_K_ADD) ? (a + b) : (a * b);
  }
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  for (int i = 0; i < kp.n; ++i) {
    const int op = kp.op[i];
    if (op >= TGP_K_ADD) {
      const T r = (op == TGP_K_ADD) ? (s1 + s0) : (s1 * s0);
      s0 = r; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
      continue;
    }
    const T v = leaf_value<T, FAM>(kp, i, r1, r2);
    s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
  }
  return s0;
}

// d(leaf)/d(param) at the pair's distances: param 0 = p0 (scale or constant), 1 = p1
// (gamma / alpha).  Derived from the forms of kernels/stationary.py:76-235.
template <typename T>
__device__ __forceinline__ T leaf_deriv(const KProg& kp, int i, int param, T r1, T r2) {
  const int op = kp.op[i];
  const bool l2 = kp.metric[i] == TGP_METRIC_L2;
  // (the derivative passes are not the hot ones: both distances up front)
  const T dist = l2 ? ((r2 == T(0)) ? r1 : sqrt(r2)) : r1;
  const T sq = l2 ? r2 : r1 * r1;
  const T p0 = T(kp.p0[i]);
  const T p1 = T(kp.p1[i]);
  switch (op) {
    case TGP_K_CONST: return param == 0 ? T(1) : T(0);
    case TGP_K_EXP: return param == 0 ? exp(-dist / p0) * dist / (p0 * p0) : T(0);
    case TGP_K_EXPSQ: return param == 0 ? exp(T(-0.5) * (sq / (p0 * p0))) * sq / (p0 * p0 * p0) : T(0);
    case TGP_K_M32: { const T a = MathC<T>::SQRT3 * (dist / p0);
                      return param == 0 ? a * a * exp(-a) / p0 : T(0); }
    case TGP_K_M52: { const T a = MathC<T>::SQRT5 * (dist / p0);
                      return param == 0 ? (a * a / T(3)) * (T(1) + a) * exp(-a) / p0 : T(0); }
    case TGP_K_COS: { const T u = MathC<T>::TWO_PI * (dist / p0);
                      return param == 0 ? sin(u) * u / p0 : T(0); }
    case TGP_K_ESS: { const T u = MathC<T>::PI * (dist / p0);
                      const T sn = sin(u), v = exp(-p1 * (sn * sn));
                      return param == 0 ? v * p1 * T(2) * sn * cos(u) * u / p0 : -(sn * sn) * v; }
    case TGP_K_RQ: { const T q = T(0.5) * (sq / (p0 * p0)) / p1;  // r^2 / (2 alpha l^2)
                     const T u = T(1) + q, v = pow(u, -p1);
                     return param == 0 ? v / u * sq / (p0 * p0 * p0) : v * (q / u - log(u)); }
    default: return T(0);
  }
}

// Forward-mode derivative of the whole program with respect to ONE leaf parameter: the stack
// carries (value, derivative) pairs; only leaf `which_op` seeds a non-zero derivative.
template <typename T>
__device__ __forceinline__ T eval_kprog_deriv(const KProg& kp, int which_op, int which_param, T r1,
                                              T r2) {
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  T d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  for (int i = 0; i < kp.n; ++i) {
    const int op = kp.op[i];
    if (op >= TGP_K_ADD) {
      const T v = (op == TGP_K_ADD) ? (s1 + s0) : (s1 * s0);
      const T dv = (op == TGP_K_ADD) ? (d1 + d0) : (d1 * s0 + s1 * d0);
      s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
      d0 = dv; d1 = d2; d2 = d3; d3 = d4; d4 = d5; d5 = d6; d6 = d7;
      continue;
    }
    const T v = leaf_value<T>(kp, i, r1, r2);
    const T dv = (i == which_op) ? leaf_deriv<T>(kp, i, which_param, r1, r2) : T(0);
    s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
    d7 = d6; d6 = d5; d5 = d4; d4 = d3; d3 = d2; d2 = d1; d1 = d0; d0 = dv;
  }
  return d0;
}

// Forward-mode derivative of the whole program with respect to the LOG-SCALE of one input dimension d (x_d -> s x_d
// for every point): what `transforms.Linear` / `Cholesky` with a per-dimension scale need (reference
// transforms.py:39-133; JAX differentiates through them for free).  Every stationary leaf depends on the
// coordinates through dist / p0 resp. sq / p0^2 only, so
//   d leaf / d log s_d = -p0 * (d leaf / d p0) * w_d,   w_d = dx_d^2 / r2 (L2 metric) or |dx_d| / r1 (L1),
// (the w_d sum to one: scaling every dimension is scaling 1 / p0).  w1 / w2 are this pair's weights.
template <typename T>
__device__ __forceinline__ T eval_kprog_deriv_dim(const KProg& kp, T r1, T r2, T w1, T w2) {
  T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  T d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, d6 = 0, d7 = 0;
  for (int i = 0; i < kp.n; ++i) {
    const int op = kp.op[i];
    if (op >= TGP_K_ADD) {
      const T v = (op == TGP_K_ADD) ? (s1 + s0) : (s1 * s0);
      const T dv = (op == TGP_K_ADD) ? (d1 + d0) : (d1 * s0 + s1 * d0);
      s0 = v; s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7;
      d0 = dv; d1 = d2; d2 = d3; d3 = d4; d4 = d5; d5 = d6; d6 = d7;
      continue;
    }
    const T v = leaf_value<T>(kp, i, r1, r2);
    const T w = kp.metric[i] == TGP_METRIC_L2 ? w2 : w1;
    const T dv = (op == TGP_K_CONST) ? T(0) : -T(kp.p0[i]) * leaf_deriv<T>(kp, i, 0, r1, r2) * w;
    s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = v;
    d7 = d6; d6 = d5; d5 = d4; d4 = d3; d3 = d2; d2 = d1; d1 = d0; d0 = dv;
  }
  return d0;
}

constexpr int KT = 128;  // tile edge

// ---- fast path: the program is one exp-family leaf, optionally times a constant ("amp * leaf") --
// OP and the metric are template parameters, so the inner loops below are straight-line code the
// compiler can unroll and interleave across elements (the general evaluator runs a chain of
// wave-uniform branches per element).  Same expressions, same order: bit-identical values.
struct FastProg {
  int op = -1, l2 = 0;   // op < 0: not a fast program
  double p0 = 1, amp = 1;
  int leaf = -1, konst = -1;  // positions of the leaf and of the constant (-1: none) in the program
};
static FastProg fast_prog(const KProg& kp) {
  FastProg f;
  auto leaf_ok = [](int op) { return op == TGP_K_EXP || op == TGP_K_EXPSQ || op == TGP_K_M32 || op == TGP_K_M52; };
  int li = -1, ci = -1;
  if (kp.n == 1 && leaf_ok(kp.op[0])) li = 0;
  else if (kp.n == 3 && kp.op[2] == TGP_K_MUL) {
    if (kp.op[0] == TGP_K_CONST && leaf_ok(kp.op[1])) { ci = 0; li = 1; }
    else if (kp.op[1] == TGP_K_CONST && leaf_ok(kp.op[0])) { ci = 1; li = 0; }
  }
  if (li < 0) return f;
  f.op = kp.op[li];
  f.l2 = kp.metric[li] == TGP_METRIC_L2;
  f.p0 = kp.p0[li];
  f.amp = ci >= 0 ? kp.p0[ci] : 1.0;  // x * 1 == x exactly
  f.leaf = li;
  f.konst = ci;
  return f;
}

// a / b for a wave-uniform divisor whose reciprocal y = RN(1 / b) was taken once per thread: the IEEE quotient, bit for
// bit, in one multiply and four FMAs instead of the ~11-instruction division sequence (v_div_scale x2, quarter-rate
// v_rcp, 5 FMAs, v_div_fmas, v_div_fixup) -- Markstein's theorem: with y the correctly rounded reciprocal and q1 a
// faithful quotient, RN(q1 + (a - b q1) y) is the correctly rounded a / b, EXCEPT for divisors whose significand is all
// ones (the one case where RN(1 / b) is not close enough): UDiv::exact is false there and for divisors outside
// [2^-200, 2^200], and the caller divides.  q0 = RN(a y) is within 2.5 ulp, q1 = RN(q0 + (a - b q0) y) is faithful
// (both residuals are exact: FMA).  Dividends above 2^600 (infinities) must take the division -- the caller tracks the
// largest distance it saw (`seen` in kmat_fast_kernel) and redoes its entries with FAST = false beyond 2^300; below 2^-600 nothing of
// the fast path can be trusted to the last bit any more (the residuals leave the normal range), but there the
// quotient is below 2^-400 by either route and every leaf that consumes it returns exactly 1 (exp(-q / 2),
// (1 + a) exp(-a), ...: 1 + O(q) rounds to 1) -- the kernel VALUE is the division's.  The entries stay the reference's quotients
// (kernels/stationary.py:104-106: r / scale, r2 / scale^2), which a multiply by the reciprocal alone does not give:
// up to 36 ulp in exp() on config 2.  Checked against the division on 4 x 10^8 structured and random
// pairs on the host (tests/markstein_check.c) and entry for entry on the device (tests/test_gpu_kernels.py).
template <typename T>
struct UDiv {
  T b, y;
  bool exact;
  __device__ __forceinline__ explicit UDiv(T b_) : b(b_), y(T(1) / b_), exact(false) {
    if constexpr (sizeof(T) == 8) {
      unsigned long long u;
      __builtin_memcpy(&u, &b_, 8);
      const unsigned long long frac = u & 0xFFFFFFFFFFFFFull;
      exact = b_ >= 0x1p-200 && b_ <= 0x1p200 && frac != 0xFFFFFFFFFFFFFull;
    }
  }
  template <bool FAST>
  __device__ __forceinline__ T div(T a) const {
    if constexpr (!FAST || sizeof(T) != 8) {
      return a / b;
    } else {
      const T q0 = a * y;
      const T r0 = __builtin_fma(-b, q0, a);
      const T q1 = __builtin_fma(r0, y, q0);
      const T r1 = __builtin_fma(-b, q1, a);
      return __builtin_fma(r1, y, q1);
    }
  }
};

template <typename T, int OP, int L2, bool FAST>
__device__ __forceinline__ T leaf_fast(T r1, T r2, const UDiv<T>& by_p0, const UDiv<T>& by_p0sq, const UDiv<T>& by3) {
  if constexpr (OP == TGP_K_EXPSQ) {
    const T sq = L2 ? r2 : r1 * r1;
    return exp(T(-0.5) * by_p0sq.template div<FAST>(sq));
  } else {
    const T dist = L2 ? ((r2 == T(0)) ? r1 : sqrt(r2)) : r1;
    if constexpr (OP == TGP_K_EXP) {
      return exp(-by_p0.template div<FAST>(dist));
    } else if constexpr (OP == TGP_K_M32) {
      const T a = MathC<T>::SQRT3 * by_p0.template div<FAST>(dist);
      return (T(1) + a) * exp(-a);
    } else {
      const T a = MathC<T>::SQRT5 * by_p0.template div<FAST>(dist);
      return (T(1) + a + by3.template div<FAST>(a * a)) * exp(-a);
    }
  }
}

// (the gradient and matrix-vector kernels: the division itself)
template <typename T, int OP, int L2>
__device__ __forceinline__ T leaf_fast(T r1, T r2, T p0, T p0sq) {
  const UDiv<T> by_p0(p0), by_p0sq(p0sq), by3(T(3));
  return leaf_fast<T, OP, L2, false>(r1, r2, by_p0, by_p0sq, by3);
}

// Full interior tiles only (every row and column inside n1 x n2): no bounds checks in the loop.
// One 128 x 128 tile per workgroup: wave w its columns 32w .. 32w + 31, lane l its rows 2l and 2l + 1 -- one 16-byte
// store per lane and column, a wave store is the tile's whole 1-KiB column.  (Rounds 1-2: lane = one row, 64 columns,
// 8-byte stores: 4.48 / 4.20 / 4.49 TB/s at N = 16 384 / 32 768 / 65 536; this shape 4.54 / 4.48 / 4.78.  A 512 x 32
// shape -- 4 KiB of one matrix column per workgroup and column -- measured 3.18 / 3.97 / 4.49: profiles/r03_j.)
template <typename T, int D, int OP, int L2>
__global__ __launch_bounds__(256) void kmat_fast_kernel(T p0, T amp, int64_t n1, int64_t n2,
                                                        const T* __restrict__ X1,
                                                        const T* __restrict__ X2,
                                                        const T* __restrict__ diag, T* __restrict__ out,
                                                        int64_t ld, int flags, int tc0, int tri_h) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  constexpr int NCOL = 32;
  const int cq = threadIdx.x >> 6, l = threadIdx.x & 63;
  int tr, tc;
  if (flags & KMAT_LOWER) {
    // 1-D grid over the tiles on and below the diagonal, column by column (local column j = 0.. holds tri_h - j tiles):
    // a 2-D grid whose upper half returns at once spent 0.07 of 0.27 ms at N = 16 384 dispatching empty workgroups
    // (scripts/probe_kmat.hip, profiles/r04_d)
    const int b = blockIdx.x;
    const double h2 = 2.0 * tri_h + 1.0;
    int j = int((h2 - sqrt(h2 * h2 - 8.0 * double(b))) * 0.5);
    while (j > 0 && j * tri_h - j * (j - 1) / 2 > b) --j;
    while ((j + 1) * tri_h - (j + 1) * j / 2 <= b) ++j;
    tc = tc0 + j;
    tr = tc + (b - (j * tri_h - j * (j - 1) / 2));
  } else {
    tr = blockIdx.x;
    tc = blockIdx.y + tc0;
  }
  __shared__ T s2all[KT * D];
  for (int t = threadIdx.x; t < KT * D; t += 256) s2all[t] = X2[int64_t(tc) * KT * D + t];
  __syncthreads();
  const T* s2 = s2all + cq * NCOL * D;
  const int64_t c0 = int64_t(tc) * KT + cq * NCOL;
  const int64_t gi = int64_t(tr) * KT + 2 * l;
  T xr[2][D];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < D; ++t) xr[h][t] = X1[(gi + h) * D + t];
  const bool on_diag = diag != nullptr && tr == tc;
  const T dg0 = on_diag ? diag[gi] : T(0), dg1 = on_diag ? diag[gi + 1] : T(0);
  const UDiv<T> by_p0(p0), by_p0sq(p0 * p0), by3(T(3));
  const int ldiag = 2 * l - cq * NCOL;  // column index (within the wave's 32) of this lane's first diagonal entry
  T* o = out + c0 * ld + gi;
  auto columns = [&](auto fast) -> T {
    constexpr bool FAST = decltype(fast)::value;
    T seen = 0;  // the largest distance of this lane's entries (a NaN passes through both routes alike)
#pragma unroll 4
    for (int c = 0; c < NCOL; ++c) {
      T v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        T r1 = 0, r2 = 0;
#pragma unroll
        for (int t = 0; t < D; ++t) {
          const T dx = xr[h][t] - s2[c * D + t];
          r1 += fabs(dx);
          r2 += dx * dx;
        }
        v[h] = amp * leaf_fast<T, OP, L2, FAST>(r1, r2, by_p0, by_p0sq, by3);
        if constexpr (FAST) seen = fmax(seen, L2 ? r2 : r1);
      }
      if (on_diag) {  // noise.py:77-78 fused
        if (c == ldiag) v[0] += dg0;
        if (c == ldiag + 1) v[1] += dg1;
      }
      T2 pair;
      pair.x = v[0];
      pair.y = v[1];
      // (non-temporal stores: 6 % in the stand-alone probe at N = 16 384, nothing at 32 768 and nothing in the library)
      *reinterpret_cast<T2*>(o + int64_t(c) * ld) = pair;
    }
    return seen;
  };
  // (wave-uniform: the divisors are kernel arguments; KMAT_PLAIN_DIV is the tests' switch to the division itself)
  bool plain = true;
  if (sizeof(T) == 8 && !(flags & KMAT_PLAIN_DIV) && by_p0.exact && by_p0sq.exact) {
    // every dividend -- r, r^2, (sqrt(5) r / l)^2 -- stays below 2^1010 while r <= 2^300
    plain = !(columns(std::true_type{}) <= T(0x1p300));
  }
  if (plain) columns(std::false_type{});
}

// D = 0: dynamic dimension (coordinates re-read from LDS); D > 0: row point in registers.
template <typename T, int D, int FAM>
__global__ __launch_bounds__(256) void kmat_kernel(KProg kp, int64_t n1, int64_t n2, int d,
                                                   const T* __restrict__ X1,
                                                   const T* __restrict__ X2,
                                                   const T* __restrict__ diag, T* __restrict__ out,
                                                   int64_t ld, int64_t rows_out, int64_t cols_out,
                                                   int flags, int tc0, int ftr, int ftc) {
  const int tr = blockIdx.x, tc = blockIdx.y + tc0;
  if ((flags & KMAT_LOWER) && tr < tc) return;
  if (tr < ftr && tc < ftc) return;  // full tiles already written by kmat_fast_kernel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s1 = reinterpret_cast<T*>(smem);  // [KT][d]
  T* s2 = s1 + KT * d;                 // [KT][d]
  const int64_t r0 = int64_t(tr) * KT, c0 = int64_t(tc) * KT;
  for (int t = threadIdx.x; t < KT * d; t += 256) {
    const int64_t gi = r0 + t / d, gj = c0 + t / d;
    s1[t] = (gi < n1) ? X1[gi * d + t % d] : T(0);
    s2[t] = (gj < n2) ? X2[gj * d + t % d] : T(0);
  }
  __syncthreads();

  const int il = threadIdx.x & (KT - 1);
  const int g = threadIdx.x >> 7;  // column half
  const int64_t gi = r0 + il;
  T xr[D > 0 ? D : 1];
  if constexpr (D > 0) {
#pragma unroll
    for (int t = 0; t < D; ++t) xr[t] = s1[il * D + t];
  }
  const T dg = (diag != nullptr && gi < n1) ? diag[gi] : T(0);
  if (gi >= rows_out) return;
  for (int c = 0; c < KT / 2; ++c) {
    const int jl = g * (KT / 2) + c;
    const int64_t gj = c0 + jl;
    if (gj >= cols_out) break;
    T v;
    if (gi < n1 && gj < n2) {
      T r1 = 0, r2 = 0;
      if constexpr (D > 0) {
#pragma unroll
        for (int t = 0; t < D; ++t) {
          const T dx = xr[t] - s2[jl * D + t];
          r1 += fabs(dx);
          r2 += dx * dx;
        }
      } else {
        for (int t = 0; t < d; ++t) {
          const T dx = s1[il * d + t] - s2[jl * d + t];
          r1 += fabs(dx);
          r2 += dx * dx;
        }
      }
      v = eval_kprog<T, FAM>(kp, r1, r2);
      if (diag != nullptr && gi == gj) v += dg;  // noise.py:77-78 fused
    } else {
      v = ((flags & KMAT_PAD_IDENTITY) && gi == gj) ? T(1) : T(0);
    }
    out[gj * ld + gi] = v;
  }
}

// One lower 128x128 tile: sum of w_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta, w = 1/2 on the
// diagonal, 1 strictly below it, 0 above (the 1/2 of 1/2 tr(G dK) folded with symmetry).
// partial[tile] gets the tile's sum; fixed LDS tree -> deterministic.
template <typename T>
__global__ __launch_bounds__(256) void kgrad_kernel(KProg kp, int which_op, int which_param,
                                                    int64_t n, int d, const T* __restrict__ X,
                                                    const T* __restrict__ alpha,
                                                    const T* __restrict__ Kinv, int64_t ld,
                                                    double* __restrict__ partial) {
  const int tr = blockIdx.x, tc = blockIdx.y;
  const int tile_id = tr * gridDim.y + tc;
  if (tr < tc) {
    if (threadIdx.x == 0) partial[tile_id] = 0.0;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s1 = reinterpret_cast<T*>(smem);  // [KT][d]
  T* s2 = s1 + KT * d;                 // [KT][d]
  __shared__ double red[256];
  const int64_t r0 = int64_t(tr) * KT, c0 = int64_t(tc) * KT;
  for (int t = threadIdx.x; t < KT * d; t += 256) {
    const int64_t gi = r0 + t / d, gj = c0 + t / d;
    s1[t] = (gi < n) ? X[gi * d + t % d] : T(0);
    s2[t] = (gj < n) ? X[gj * d + t % d] : T(0);
  }
  __syncthreads();
  const int il = threadIdx.x & (KT - 1), g = threadIdx.x >> 7;
  const int64_t gi = r0 + il;
  const T ai = (gi < n) ? alpha[gi] : T(0);
  double acc = 0.0;
  if (gi < n) {
    for (int c = 0; c < KT / 2; ++c) {
      const int jl = g * (KT / 2) + c;
      const int64_t gj = c0 + jl;
      if (gj >= n || gj > gi) continue;
      T r1 = 0, r2 = 0, dxd = 0;
      for (int t = 0; t < d; ++t) {
        const T dx = s1[il * d + t] - s2[jl * d + t];
        r1 += fabs(dx);
        r2 += dx * dx;
        if (t == -1 - which_op) dxd = dx;  // (which_op < 0: the log-scale of input dimension -1 - which_op)
      }
      const T dk = which_op >= 0 ? eval_kprog_deriv<T>(kp, which_op, which_param, r1, r2)
                                 : eval_kprog_deriv_dim<T>(kp, r1, r2, r1 > T(0) ? fabs(dxd) / r1 : T(0),
                                                           r2 > T(0) ? dxd * dxd / r2 : T(0));
      const T gij = ai * alpha[gj] - Kinv[gj * ld + gi];
      acc += double(gij) * double(dk) * (gi == gj ? 0.5 : 1.0);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[tile_id] = red[0];
}

// The same sum for a CHUNK OF COLUMNS of K^-1 held as rows (round 5: the gradient on the block-column path).  `Kc` is
// (n_pad, R) ROW-major -- K^-1[i, c0 + r] at i * R + r, the layout of the distributed solves' right-hand sides --, and
// only row tiles inside block rows THIS RANK owns contribute (block row b = row / nb belongs to rank b mod G): the ranks'
// partial sums add up to the lower triangle of the chunk.  Tile (tr, tc): rows tr * 128 .., columns c0 + tc * 128 ...
template <typename T>
__global__ __launch_bounds__(256) void kgrad_cols_kernel(KProg kp, int which_op, int which_param, int64_t n, int d,
                                                         const T* __restrict__ X, const T* __restrict__ alpha,
                                                         const T* __restrict__ Kc, int64_t R, int64_t c0, int64_t nb,
                                                         int G, int rank, double* __restrict__ partial) {
  const int tr = blockIdx.x, tc = blockIdx.y;
  const int tile_id = tr * gridDim.y + tc;
  const int64_t r0 = int64_t(tr) * KT, cc0 = c0 + int64_t(tc) * KT;
  const bool mine = int((r0 / nb) % G) == rank;
  if (!mine || r0 + KT <= cc0 || cc0 >= n || r0 >= n) {  // not this rank's rows, or strictly above the diagonal
    if (threadIdx.x == 0) partial[tile_id] = 0.0;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s1 = reinterpret_cast<T*>(smem);  // [KT][d]
  T* s2 = s1 + KT * d;                 // [KT][d]
  __shared__ double red[256];
  for (int t = threadIdx.x; t < KT * d; t += 256) {
    const int64_t gi = r0 + t / d, gj = cc0 + t / d;
    s1[t] = (gi < n) ? X[gi * d + t % d] : T(0);
    s2[t] = (gj < n) ? X[gj * d + t % d] : T(0);
  }
  __syncthreads();
  const int il = threadIdx.x & (KT - 1), g = threadIdx.x >> 7;
  const int64_t gi = r0 + il;
  const T ai = (gi < n) ? alpha[gi] : T(0);
  double acc = 0.0;
  if (gi < n) {
    for (int c = 0; c < KT / 2; ++c) {
      const int jl = g * (KT / 2) + c;
      const int64_t gj = cc0 + jl;
      if (gj >= n || gj > gi) continue;
      T r1 = 0, r2 = 0, dxd = 0;
      for (int t = 0; t < d; ++t) {
        const T dx = s1[il * d + t] - s2[jl * d + t];
        r1 += fabs(dx);
        r2 += dx * dx;
        if (t == -1 - which_op) dxd = dx;
      }
      const T dk = which_op >= 0 ? eval_kprog_deriv<T>(kp, which_op, which_param, r1, r2)
                                 : eval_kprog_deriv_dim<T>(kp, r1, r2, r1 > T(0) ? fabs(dxd) / r1 : T(0),
                                                           r2 > T(0) ? dxd * dxd / r2 : T(0));
      const T gij = ai * alpha[gj] - Kc[gi * R + (gj - c0)];
      acc += double(gij) * double(dk) * (gi == gj ? 0.5 : 1.0);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[tile_id] = red[0];
}

// *out += sum(partial) (one workgroup, fixed tree: deterministic)
__global__ __launch_bounds__(1024) void add_partials_kernel(int64_t count, const double* __restrict__ partial,
                                                            double* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < count; i += 1024) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 512; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out += red[0];
}

// diag[c0 + r] = Kc[c0 + r, r] for the chunk's columns (K^-1_jj: the noise gradient's second term)
template <typename T>
__global__ __launch_bounds__(256) void kcols_diag_kernel(int64_t n, const T* __restrict__ Kc, int64_t R, int64_t c0,
                                                         T* __restrict__ diag) {
  const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (r < R && c0 + r < n) diag[c0 + r] = Kc[(c0 + r) * R + r];
}

__global__ __launch_bounds__(1024) void sum_partials_kernel(int64_t count, const double* __restrict__ partial,
                                                            double* __restrict__ out) {
  __shared__ double red[1024];
  double acc = 0;
  for (int64_t i = threadIdx.x; i < count; i += 1024) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 512; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// d loglik / d noise_i = 1/2 (alpha_i^2 - Kinv_ii)
template <typename T>
__global__ __launch_bounds__(256) void noise_grad_kernel(int64_t n, const T* __restrict__ alpha,
                                                         const T* __restrict__ Kinv, int64_t ld,
                                                         T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) out[i] = T(0.5) * (alpha[i] * alpha[i] - Kinv[i * ld + i]);
}

template <typename T>
__global__ __launch_bounds__(256) void kdiag_kernel(KProg kp, int64_t n, const T* __restrict__ add,
                                                    T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  // evaluate_diag = evaluate(x, x) (base.py:59-66): every distance is exactly zero
  T v = eval_kprog<T>(kp, T(0), T(0));
  if (add != nullptr) v += add[i];
  out[i] = v;
}

// Fused K9: partial[chunk][r][i] = sum_{j in chunk} k(X1[i], X2[j]) v[r][j] for up to GV_NV
// right-hand sides at once (every kernel value is evaluated ONCE per pass, whatever the number
// of columns of `y` in Kernel.matmul).  One lane per row i, X2 / v chunks staged through LDS and
// broadcast-read.
constexpr int GV_ROWS = 256, GV_JB = 256, GV_NV = 8;
template <typename T, int FAM>
__global__ __launch_bounds__(256) void kmat_gemv_kernel(KProg kp, int64_t n1, int64_t n2, int d,
                                                        const T* __restrict__ X1,
                                                        const T* __restrict__ X2,
                                                        const T* __restrict__ v, int nv,
                                                        T* __restrict__ partial, int64_t jchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [GV_JB][d]
  T* sv = sx + GV_JB * d;              // [GV_JB][GV_NV]
  const int64_t i = int64_t(blockIdx.x) * GV_ROWS + threadIdx.x;
  const int64_t j0 = int64_t(blockIdx.y) * jchunk;
  const int64_t j1 = (j0 + jchunk < n2) ? j0 + jchunk : n2;
  T xi[TGP_MAX_DIM];
#pragma unroll
  for (int t = 0; t < TGP_MAX_DIM; ++t) xi[t] = (t < d && i < n1) ? X1[i * d + t] : T(0);
  T acc[GV_NV];
#pragma unroll
  for (int r = 0; r < GV_NV; ++r) acc[r] = 0;
  for (int64_t jb = j0; jb < j1; jb += GV_JB) {
    const int cnt = int((j1 - jb < GV_JB) ? (j1 - jb) : GV_JB);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * d; t += 256) sx[t] = X2[jb * d + t];
    for (int t = threadIdx.x; t < cnt * GV_NV; t += 256) {
      const int jj = t / GV_NV, r = t % GV_NV;
      sv[t] = (r < nv) ? v[int64_t(r) * n2 + jb + jj] : T(0);
    }
    __syncthreads();
    for (int jj = 0; jj < cnt; ++jj) {
      T r1 = 0, r2 = 0;
#pragma unroll
      for (int t = 0; t < TGP_MAX_DIM; ++t) {
        if (t < d) {
          const T dx = xi[t] - sx[jj * d + t];
          r1 += fabs(dx);
          r2 += dx * dx;
        }
      }
      const T kv = eval_kprog<T, FAM>(kp, r1, r2);
#pragma unroll
      for (int r = 0; r < GV_NV; ++r) acc[r] += kv * sv[jj * GV_NV + r];
    }
  }
  if (i < n1)
    for (int r = 0; r < nv; ++r) partial[(int64_t(blockIdx.y) * nv + r) * n1 + i] = acc[r];
}



// d(leaf)/d(scale), the expressions of leaf_deriv with op and metric fixed at compile time
template <typename T, int OP, int L2>
__device__ __forceinline__ T leaf_deriv_fast(T r1, T r2, T p0) {
  if constexpr (OP == TGP_K_EXPSQ) {
    const T sq = L2 ? r2 : r1 * r1;
    return exp(T(-0.5) * (sq / (p0 * p0))) * sq / (p0 * p0 * p0);
  } else {
    const T dist = L2 ? ((r2 == T(0)) ? r1 : sqrt(r2)) : r1;
    if constexpr (OP == TGP_K_EXP) {
      return exp(-dist / p0) * dist / (p0 * p0);
    } else if constexpr (OP == TGP_K_M32) {
      const T a = MathC<T>::SQRT3 * (dist / p0);
      return a * a * exp(-a) / p0;
    } else {
      const T a = MathC<T>::SQRT5 * (dist / p0);
      return (a * a / T(3)) * (T(1) + a) * exp(-a) / p0;
    }
  }
}

// kgrad_kernel for "leaf" / "amp * leaf" programs: BOTH derivatives (d/d amp = leaf,
// d/d scale = amp * d leaf) in one pass over K^-1.  partial[tile] and partial[ntiles + tile].
template <typename T, int OP, int L2>
__global__ __launch_bounds__(256) void kgrad_fast_kernel(T p0, T amp, int64_t n, int d,
                                                         const T* __restrict__ X,
                                                         const T* __restrict__ alpha,
                                                         const T* __restrict__ Kinv, int64_t ld,
                                                         double* __restrict__ partial) {
  const int tr = blockIdx.x, tc = blockIdx.y;
  const int tile_id = tr * gridDim.y + tc, ntiles = gridDim.x * gridDim.y;
  if (tr < tc) {
    if (threadIdx.x == 0) partial[tile_id] = partial[ntiles + tile_id] = 0.0;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s1 = reinterpret_cast<T*>(smem);  // [KT][d]
  T* s2 = s1 + KT * d;                 // [KT][d]
  __shared__ double red[2][256];
  const int64_t r0 = int64_t(tr) * KT, c0 = int64_t(tc) * KT;
  for (int t = threadIdx.x; t < KT * d; t += 256) {
    const int64_t gi = r0 + t / d, gj = c0 + t / d;
    s1[t] = (gi < n) ? X[gi * d + t % d] : T(0);
    s2[t] = (gj < n) ? X[gj * d + t % d] : T(0);
  }
  __syncthreads();
  const int il = threadIdx.x & (KT - 1), g = threadIdx.x >> 7;
  const int64_t gi = r0 + il;
  const T ai = (gi < n) ? alpha[gi] : T(0);
  const T p0sq = p0 * p0;
  double acc_a = 0.0, acc_s = 0.0;
  if (gi < n) {
    for (int c = 0; c < KT / 2; ++c) {
      const int jl = g * (KT / 2) + c;
      const int64_t gj = c0 + jl;
      if (gj >= n || gj > gi) continue;
      T r1 = 0, r2 = 0;
      for (int t = 0; t < d; ++t) {
        const T dx = s1[il * d + t] - s2[jl * d + t];
        r1 += fabs(dx);
        r2 += dx * dx;
      }
      const T leaf = leaf_fast<T, OP, L2>(r1, r2, p0, p0sq);
      const T dk = amp * leaf_deriv_fast<T, OP, L2>(r1, r2, p0);
      const double w = double(ai * alpha[gj] - Kinv[gj * ld + gi]) * (gi == gj ? 0.5 : 1.0);
      acc_a += w * double(leaf);
      acc_s += w * double(dk);
    }
  }
  red[0][threadIdx.x] = acc_a;
  red[1][threadIdx.x] = acc_s;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) {
      red[0][threadIdx.x] += red[0][threadIdx.x + sft];
      red[1][threadIdx.x] += red[1][threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[tile_id] = red[0][0];
    partial[ntiles + tile_id] = red[1][0];
  }
}

// the same for "leaf" / "amp * leaf" programs: straight-line inner loop (see kmat_fast_kernel)
template <typename T, int OP, int L2>
__global__ __launch_bounds__(256) void kmat_gemv_fast_kernel(T p0, T amp, int64_t n1, int64_t n2, int d,
                                                             const T* __restrict__ X1,
                                                             const T* __restrict__ X2,
                                                             const T* __restrict__ v, int nv,
                                                             T* __restrict__ partial, int64_t jchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sx = reinterpret_cast<T*>(smem);  // [GV_JB][d]
  T* sv = sx + GV_JB * d;              // [GV_JB][GV_NV]
  const int64_t i = int64_t(blockIdx.x) * GV_ROWS + threadIdx.x;
  const int64_t j0 = int64_t(blockIdx.y) * jchunk;
  const int64_t j1 = (j0 + jchunk < n2) ? j0 + jchunk : n2;
  T xi[TGP_MAX_DIM];
#pragma unroll
  for (int t = 0; t < TGP_MAX_DIM; ++t) xi[t] = (t < d && i < n1) ? X1[i * d + t] : T(0);
  const T p0sq = p0 * p0;
  T acc[GV_NV];
#pragma unroll
  for (int r = 0; r < GV_NV; ++r) acc[r] = 0;
  for (int64_t jb = j0; jb < j1; jb += GV_JB) {
    const int cnt = int((j1 - jb < GV_JB) ? (j1 - jb) : GV_JB);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * d; t += 256) sx[t] = X2[jb * d + t];
    for (int t = threadIdx.x; t < cnt * GV_NV; t += 256) {
      const int jj = t / GV_NV, r = t % GV_NV;
      sv[t] = (r < nv) ? v[int64_t(r) * n2 + jb + jj] : T(0);
    }
    __syncthreads();
#pragma unroll 2
    for (int jj = 0; jj < cnt; ++jj) {
      T r1 = 0, r2 = 0;
#pragma unroll
      for (int t = 0; t < TGP_MAX_DIM; ++t) {
        if (t < d) {
          const T dx = xi[t] - sx[jj * d + t];
          r1 += fabs(dx);
          r2 += dx * dx;
        }
      }
      const T kv = amp * leaf_fast<T, OP, L2>(r1, r2, p0, p0sq);
#pragma unroll
      for (int r = 0; r < GV_NV; ++r) acc[r] += kv * sv[jj * GV_NV + r];
    }
  }
  if (i < n1)
    for (int r = 0; r < nv; ++r) partial[(int64_t(blockIdx.y) * nv + r) * n1 + i] = acc[r];
}

// out[r][i] = sum over chunks, fixed order
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(int64_t n1, int nchunks, int nv,
                                                              const T* __restrict__ partial,
                                                              T* __restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= n1) return;
  T acc = 0;
  for (int c = 0; c < nchunks; ++c) acc += partial[(int64_t(c) * nv + r) * n1 + i];
  out[int64_t(r) * n1 + i] = acc;
}

}  // namespace

int make_kprog(const tgp_kop* prog, int nops, KProg* out) {
  TGP_ARG_CHECK(prog != nullptr && nops >= 1 && nops <= TGP_KPROG_MAX,
                "kernel program must have 1..%d ops (got %d)", TGP_KPROG_MAX, nops);
  int depth = 0;
  out->n = nops;
  for (int i = 0; i < nops; ++i) {
    const int op = prog[i].op;
    if (op == TGP_K_ADD || op == TGP_K_MUL) {
      TGP_ARG_CHECK(depth >= 2, "kernel program: stack underflow at op %d", i);
      depth -= 1;
    } else {
      TGP_ARG_CHECK(op >= TGP_K_CONST && op <= TGP_K_RQ, "kernel program: bad opcode %d at %d", op, i);
      TGP_ARG_CHECK(prog[i].metric == TGP_METRIC_L1 || prog[i].metric == TGP_METRIC_L2,
                    "kernel program: bad metric at op %d", i);
      depth += 1;
      TGP_ARG_CHECK(depth <= TGP_KSTACK_MAX, "kernel program: stack deeper than %d", TGP_KSTACK_MAX);
    }
    out->op[i] = op;
    out->metric[i] = prog[i].metric;
    out->p0[i] = prog[i].p0;
    out->p1[i] = prog[i].p1;
  }
  TGP_ARG_CHECK(depth == 1, "kernel program leaves %d values on the stack", depth);
  return TGP_OK;
}

// only Constant / Exp / ExpSquared / Matern leaves (+ sums and products): the FAM = 1 kernels
static bool exp_family(const KProg& kp) {
  for (int i = 0; i < kp.n; ++i) {
    const int op = kp.op[i];
    if (!(op == TGP_K_CONST || op == TGP_K_EXP || op == TGP_K_EXPSQ || op == TGP_K_M32 ||
          op == TGP_K_M52 || op >= TGP_K_ADD))
      return false;
  }
  return true;
}

template <typename T>
int launch_kmat_cols(tgp_ctx* ctx, hipStream_t st, const KProg& kp, int64_t n1, int64_t n2, int d,
                     const T* X1, const T* X2, const T* diag, T* out, int64_t ld, int64_t rows_out,
                     int64_t cols_out, int flags, int64_t tc0, int64_t ntc) {
  TGP_ARG_CHECK(d >= 1 && d <= TGP_MAX_DIM, "input dimension must be 1..%d (got %d)", TGP_MAX_DIM, d);
  TGP_ARG_CHECK(rows_out >= n1 && cols_out >= n2 && ld >= rows_out, "kmat: bad output extents");
  if (rows_out == 0 || cols_out == 0 || ntc <= 0) return TGP_OK;
  const int64_t tr = (rows_out + KT - 1) / KT, tc = (cols_out + KT - 1) / KT;
  TGP_ARG_CHECK(tc0 >= 0 && tc0 + ntc <= tc, "kmat: column tile range outside the matrix");
  if (ctx->trace) {  // v: first column tile, number of column tiles, ld, flags
    trace_push(ctx, 7, st, tc0, ntc, ld, flags);
    return TGP_OK;
  }
  TGP_ARG_CHECK(ntc <= 65535, "kmat: too many column tiles");
  // full tiles of "leaf" / "amp * leaf" programs: the straight-line kernel
  int ftr = 0, ftc = 0;
  const FastProg fp = fast_prog(kp);
  // (its 16-byte stores need an even leading dimension and a 16-byte aligned matrix)
  if (fp.op >= 0 && d <= 3 && ld % 2 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0) {
    ftr = int(n1 / KT);
    ftc = int(n2 / KT);
    const int64_t fc = std::min<int64_t>(tc0 + ntc, ftc) - tc0;  // column tiles of this call that are full
    // lower-only: column tile c holds the row tiles c .. ftr-1 (the matrix is square there)
    const int64_t tri_h = (flags & KMAT_LOWER) ? ftr - tc0 : 0;
    const int64_t tri_c = std::min<int64_t>(fc, tri_h);  // columns of this call with a tile on or below the diagonal
    const bool any = (flags & KMAT_LOWER) ? (tri_c > 0) : (ftr > 0 && fc > 0);
    if (any) {
      dim3 fgrid((unsigned)ftr, (unsigned)fc);
      if (flags & KMAT_LOWER) fgrid = dim3((unsigned)(tri_c * tri_h - tri_c * (tri_c - 1) / 2));
#define TGP_FAST3(DD, OP, L2)                                                                     \
  hipLaunchKernelGGL((kmat_fast_kernel<T, DD, OP, L2>), fgrid, dim3(256), 0, st, T(fp.p0),        \
                     T(fp.amp), n1, n2, X1, X2, diag, out, ld, flags | (ctx->kmat_plain_div != 0 ? KMAT_PLAIN_DIV : 0), (int)tc0,  \
                     (int)tri_h)
#define TGP_FAST2(DD, OP)                                                                         \
  do {                                                                                           \
    if (fp.l2) TGP_FAST3(DD, OP, 1); else TGP_FAST3(DD, OP, 0);                                   \
  } while (0)
#define TGP_FAST1(DD)                                                                             \
  do {                                                                                           \
    switch (fp.op) {                                                                             \
      case TGP_K_EXP: TGP_FAST2(DD, TGP_K_EXP); break;                                            \
      case TGP_K_EXPSQ: TGP_FAST2(DD, TGP_K_EXPSQ); break;                                        \
      case TGP_K_M32: TGP_FAST2(DD, TGP_K_M32); break;                                            \
      default: TGP_FAST2(DD, TGP_K_M52); break;                                                   \
    }                                                                                            \
  } while (0)
      if (d == 1) TGP_FAST1(1);
      else if (d == 2) TGP_FAST1(2);
      else TGP_FAST1(3);
#undef TGP_FAST1
#undef TGP_FAST2
#undef TGP_FAST3
    } else {
      ftr = ftc = 0;
    }
    if (ftr == tr && ftc >= tc0 + ntc) {  // nothing ragged left
      TGP_HIP_TRY(hipGetLastError());
      return TGP_OK;
    }
  }
  dim3 grid((unsigned)tr, (unsigned)ntc);
  const size_t shmem = 2 * size_t(KT) * d * sizeof(T);
  const bool fam = exp_family(kp);
#define TGP_KMAT_LAUNCH(DD)                                                                      \
  do {                                                                                           \
    if (fam)                                                                                     \
      hipLaunchKernelGGL((kmat_kernel<T, DD, 1>), grid, dim3(256), shmem, st, kp, n1, n2, d, X1, \
                         X2, diag, out, ld, rows_out, cols_out, flags, (int)tc0, ftr, ftc);      \
    else                                                                                         \
      hipLaunchKernelGGL((kmat_kernel<T, DD, 0>), grid, dim3(256), shmem, st, kp, n1, n2, d, X1, \
                         X2, diag, out, ld, rows_out, cols_out, flags, (int)tc0, ftr, ftc);      \
  } while (0)
  switch (d) {
    case 1: TGP_KMAT_LAUNCH(1); break;
    case 2: TGP_KMAT_LAUNCH(2); break;
    case 3: TGP_KMAT_LAUNCH(3); break;
    case 4: TGP_KMAT_LAUNCH(4); break;
    default: TGP_KMAT_LAUNCH(0); break;
  }
#undef TGP_KMAT_LAUNCH
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int launch_kmat(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                const T* X2, const T* diag, T* out, int64_t ld, int64_t rows_out, int64_t cols_out,
                int flags) {
  return launch_kmat_cols<T>(ctx, ctx->stream, kp, n1, n2, d, X1, X2, diag, out, ld, rows_out,
                             cols_out, flags, 0, (cols_out + KT - 1) / KT);
}

template <typename T>
int launch_kdiag(tgp_ctx* ctx, const KProg& kp, int64_t n, int d, const T* X, const T* add, T* out) {
  (void)X; (void)d;
  if (n == 0) return TGP_OK;
  hipLaunchKernelGGL((kdiag_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, kp, n, add, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// out (nv x n1, row-major) = [K(X1, X2) v_r]_r for the nv vectors v (nv x n2, row-major)
template <typename T>
int launch_kmat_gemv_multi(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                           const T* X2, const T* v, int64_t nv_total, T* out) {
  TGP_ARG_CHECK(d >= 1 && d <= TGP_MAX_DIM, "input dimension must be 1..%d (got %d)", TGP_MAX_DIM, d);
  if (n1 == 0 || nv_total == 0) return TGP_OK;
  const int64_t rb = (n1 + GV_ROWS - 1) / GV_ROWS;
  // enough column chunks to fill the chip when there are few rows of test points
  int64_t nch = (2048 + rb - 1) / rb;
  const int64_t max_ch = (n2 + GV_JB - 1) / GV_JB;
  if (nch > max_ch) nch = max_ch;
  if (nch < 1) nch = 1;
  if (nch > 65535) nch = 65535;
  int64_t jchunk = round_up((n2 + nch - 1) / nch, GV_JB);
  if (jchunk < GV_JB) jchunk = GV_JB;
  nch = (n2 + jchunk - 1) / jchunk;
  if (nch < 1) nch = 1;
  TGP_TRY(ensure_work(ctx, size_t(nch) * GV_NV * n1 * sizeof(T)));
  T* partial = static_cast<T*>(ctx->d_work);
  const size_t shmem = size_t(GV_JB) * (d + GV_NV) * sizeof(T);
  for (int64_t r0 = 0; r0 < nv_total; r0 += GV_NV) {
    const int nv = int(std::min<int64_t>(GV_NV, nv_total - r0));
    const FastProg fp = fast_prog(kp);
#define TGP_GV3(OP, L2)                                                                             \
  hipLaunchKernelGGL((kmat_gemv_fast_kernel<T, OP, L2>), dim3((unsigned)rb, (unsigned)nch), dim3(256), shmem, \
                     ctx->stream, T(fp.p0), T(fp.amp), n1, n2, d, X1, X2, v + r0 * n2, nv, partial, jchunk)
#define TGP_GV2(OP)                                                                                 \
  do {                                                                                             \
    if (fp.l2) TGP_GV3(OP, 1); else TGP_GV3(OP, 0);                                                 \
  } while (0)
    if (fp.op == TGP_K_EXP) TGP_GV2(TGP_K_EXP);
    else if (fp.op == TGP_K_EXPSQ) TGP_GV2(TGP_K_EXPSQ);
    else if (fp.op == TGP_K_M32) TGP_GV2(TGP_K_M32);
    else if (fp.op == TGP_K_M52) TGP_GV2(TGP_K_M52);
#undef TGP_GV2
#undef TGP_GV3
    else if (exp_family(kp))
      hipLaunchKernelGGL((kmat_gemv_kernel<T, 1>), dim3((unsigned)rb, (unsigned)nch), dim3(256), shmem,
                         ctx->stream, kp, n1, n2, d, X1, X2, v + r0 * n2, nv, partial, jchunk);
    else
      hipLaunchKernelGGL((kmat_gemv_kernel<T, 0>), dim3((unsigned)rb, (unsigned)nch), dim3(256), shmem,
                         ctx->stream, kp, n1, n2, d, X1, X2, v + r0 * n2, nv, partial, jchunk);
    hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((unsigned)((n1 + 255) / 256), (unsigned)nv), dim3(256), 0,
                       ctx->stream, n1, (int)nch, nv, partial, out + r0 * n1);
  }
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int launch_kmat_gemv(tgp_ctx* ctx, const KProg& kp, int64_t n1, int64_t n2, int d, const T* X1,
                     const T* X2, const T* v, T* out) {
  return launch_kmat_gemv_multi<T>(ctx, kp, n1, n2, d, X1, X2, v, 1, out);
}

template <typename T>
int launch_kgrad(tgp_ctx* ctx, const KProg& kp, int which_op, int which_param, int64_t n, int d,
                 const T* X, const T* alpha, const T* Kinv, int64_t ld, double* out_dev) {
  const int64_t tiles = (n + KT - 1) / KT;
  TGP_ARG_CHECK(tiles <= 65535, "kgrad: too many tiles");
  TGP_TRY(ensure_work(ctx, size_t(tiles) * tiles * sizeof(double)));
  double* partial = static_cast<double*>(ctx->d_work);
  const size_t shmem = 2 * size_t(KT) * d * sizeof(T);
  hipLaunchKernelGGL((kgrad_kernel<T>), dim3((unsigned)tiles, (unsigned)tiles), dim3(256), shmem,
                     ctx->stream, kp, which_op, which_param, n, d, X, alpha, Kinv, ld, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles * tiles, partial,
                     out_dev);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// out_accum[0] += this rank's share of  sum_{i >= j, j in chunk} w_ij (alpha_i alpha_j - Kinv_ij) dK_ij / dtheta  over the
// chunk of R columns from c0 (Kc: (n_pad, R) row-major); diag_out (or NULL) receives the chunk's K^-1_jj
template <typename T>
int launch_kgrad_cols(tgp_ctx* ctx, const KProg& kp, int which_op, int which_param, int64_t n, int d, const T* X,
                      const T* alpha, const T* Kc, int64_t R, int64_t c0, int64_t nb, int G, int rank,
                      double* out_accum) {
  const int64_t tr = (n + KT - 1) / KT, tc = (std::min<int64_t>(R, n - c0) + KT - 1) / KT;
  TGP_ARG_CHECK(tr <= 65535 && tc >= 1 && tc <= 65535 && c0 % KT == 0, "kgrad_cols: bad chunk");
  TGP_TRY(ensure_work(ctx, size_t(tr) * tc * sizeof(double)));
  double* partial = static_cast<double*>(ctx->d_work);
  const size_t shmem = 2 * size_t(KT) * d * sizeof(T);
  hipLaunchKernelGGL((kgrad_cols_kernel<T>), dim3((unsigned)tr, (unsigned)tc), dim3(256), shmem, ctx->stream, kp,
                     which_op, which_param, n, d, X, alpha, Kc, R, c0, nb, G, rank, partial);
  hipLaunchKernelGGL(add_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, tr * tc, partial, out_accum);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

template <typename T>
int launch_kcols_diag(tgp_ctx* ctx, int64_t n, const T* Kc, int64_t R, int64_t c0, T* diag) {
  hipLaunchKernelGGL((kcols_diag_kernel<T>), dim3((unsigned)((R + 255) / 256)), dim3(256), 0, ctx->stream, n, Kc, R, c0,
                     diag);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// Gradient sums of a "leaf" / "amp * leaf" program in one pass: out_dev[0] = d/d(constant) (only
// meaningful when the program has one), out_dev[1] = d/d(scale).  Returns 1 when the program is
// of that shape (*leaf / *konst = their positions, -1: none), 0 when the caller must use launch_kgrad.
template <typename T>
int launch_kgrad_fast(tgp_ctx* ctx, const KProg& kp, int64_t n, int d, const T* X, const T* alpha,
                      const T* Kinv, int64_t ld, double* out_dev, int* leaf, int* konst) {
  const FastProg fp = fast_prog(kp);
  if (fp.op < 0) return 0;
  const int64_t tiles = (n + KT - 1) / KT;
  TGP_ARG_CHECK(tiles <= 65535, "kgrad: too many tiles");
  TGP_TRY(ensure_work(ctx, 2 * size_t(tiles) * tiles * sizeof(double)));
  double* partial = static_cast<double*>(ctx->d_work);
  const size_t shmem = 2 * size_t(KT) * d * sizeof(T);
  const dim3 grid((unsigned)tiles, (unsigned)tiles);
#define TGP_KG3(OP, L2)                                                                            \
  hipLaunchKernelGGL((kgrad_fast_kernel<T, OP, L2>), grid, dim3(256), shmem, ctx->stream, T(fp.p0), \
                     T(fp.amp), n, d, X, alpha, Kinv, ld, partial)
#define TGP_KG2(OP)                                                                                \
  do {                                                                                            \
    if (fp.l2) TGP_KG3(OP, 1); else TGP_KG3(OP, 0);                                                \
  } while (0)
  if (fp.op == TGP_K_EXP) TGP_KG2(TGP_K_EXP);
  else if (fp.op == TGP_K_EXPSQ) TGP_KG2(TGP_K_EXPSQ);
  else if (fp.op == TGP_K_M32) TGP_KG2(TGP_K_M32);
  else TGP_KG2(TGP_K_M52);
#undef TGP_KG2
#undef TGP_KG3
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles * tiles, partial, out_dev);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles * tiles,
                     partial + tiles * tiles, out_dev + 1);
  TGP_HIP_TRY(hipGetLastError());
  *leaf = fp.leaf;
  *konst = fp.konst;
  return 1;
}

template <typename T>
int launch_noise_grad(tgp_ctx* ctx, int64_t n, const T* alpha, const T* Kinv, int64_t ld, T* out) {
  hipLaunchKernelGGL((noise_grad_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, n, alpha, Kinv, ld, out);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

#define TGP_INST(T)                                                                               \
  template int launch_kgrad_cols<T>(tgp_ctx*, const KProg&, int, int, int64_t, int, const T*, const T*, const T*,   \
                                    int64_t, int64_t, int64_t, int, int, double*);                                  \
  template int launch_kcols_diag<T>(tgp_ctx*, int64_t, const T*, int64_t, int64_t, T*);                            \
  template int launch_kgrad<T>(tgp_ctx*, const KProg&, int, int, int64_t, int, const T*, const T*, \
                               const T*, int64_t, double*);                                       \
  template int launch_noise_grad<T>(tgp_ctx*, int64_t, const T*, const T*, int64_t, T*);          \
  template int launch_kgrad_fast<T>(tgp_ctx*, const KProg&, int64_t, int, const T*, const T*,     \
                                    const T*, int64_t, double*, int*, int*);                      \
  template int launch_kmat<T>(tgp_ctx*, const KProg&, int64_t, int64_t, int, const T*, const T*,  \
                              const T*, T*, int64_t, int64_t, int64_t, int);                      \
  template int launch_kmat_cols<T>(tgp_ctx*, hipStream_t, const KProg&, int64_t, int64_t, int,    \
                                   const T*, const T*, const T*, T*, int64_t, int64_t, int64_t,   \
                                   int, int64_t, int64_t);                                        \
  template int launch_kdiag<T>(tgp_ctx*, const KProg&, int64_t, int, const T*, const T*, T*);     \
  template int launch_kmat_gemv<T>(tgp_ctx*, const KProg&, int64_t, int64_t, int, const T*,       \
                                   const T*, const T*, T*);                                       \
  template int launch_kmat_gemv_multi<T>(tgp_ctx*, const KProg&, int64_t, int64_t, int, const T*, \
                                         const T*, const T*, int64_t, T*);
TGP_INST(float)
TGP_INST(double)
#undef TGP_INST

}  // namespace tgp
