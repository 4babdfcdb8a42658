/*
 * ref_c.c -- plain-C restatement of the tinygp dense DirectSolver path (CPU, scalar, libm).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/tinygp_np.py header; PARITY UNPINNED against
 * reference outputs -- the reference cannot run in this container).  This is the second,
 * independent implementation the NumPy/SciPy oracle is cross-checked against, the way the
 * reference's own tests cross-check DirectSolver against george's C++ (tests/
 * test_george_compat.py:105-154).  It shares nothing with SciPy/LAPACK: textbook unblocked
 * Cholesky (row-major, lower), forward/back substitution, scalar kernel evaluation.
 * It consumes the same `tgp_kop` kernel program as the HIP library so the host-side
 * kernel-tree compiler is checked too.
 *
 * Reference lines followed (relative to /root/reference/src/tinygp):
 *   kernels/distance.py:41-59, kernels/stationary.py:76-235, kernels/base.py:170-209,
 *   noise.py:77-78, solvers/direct.py:49-70, gp.py:313-320.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tgp_hip.h"

/* kernels/distance.py + kernels/stationary.py: one pair of points */
static double eval_prog(const tgp_kop* prog, int nops, const double* x1, const double* x2, int d) {
  double r1 = 0.0, r2 = 0.0; /* L1Distance.distance, L2Distance.squared_distance */
  for (int t = 0; t < d; ++t) {
    double dx = x1[t] - x2[t];
    r1 += fabs(dx);
    r2 += dx * dx;
  }
  double st[TGP_KSTACK_MAX];
  int sp = 0;
  for (int i = 0; i < nops; ++i) {
    const tgp_kop* k = &prog[i];
    if (k->op == TGP_K_ADD) { st[sp - 2] = st[sp - 2] + st[sp - 1]; --sp; continue; }
    if (k->op == TGP_K_MUL) { st[sp - 2] = st[sp - 2] * st[sp - 1]; --sp; continue; }
    /* distance.py:51-56: zero-safe sqrt for L2; distance.py:30-38: L1 squared = dist^2 */
    double dist = (k->metric == TGP_METRIC_L2) ? (r2 == 0.0 ? r1 : sqrt(r2)) : r1;
    double sq = (k->metric == TGP_METRIC_L2) ? r2 : r1 * r1;
    double v = 0.0;
    switch (k->op) {
      case TGP_K_CONST: v = k->p0; break;
      case TGP_K_EXP: v = exp(-dist / k->p0); break;
      case TGP_K_EXPSQ: v = exp(-0.5 * (sq / (k->p0 * k->p0))); break;
      case TGP_K_M32: { double a = sqrt(3.0) * (dist / k->p0); v = (1.0 + a) * exp(-a); } break;
      case TGP_K_M52: { double a = sqrt(5.0) * (dist / k->p0);
                        v = (1.0 + a + (a * a) / 3.0) * exp(-a); } break;
      case TGP_K_COS: v = cos(2.0 * M_PI * (dist / k->p0)); break;
      case TGP_K_ESS: { double s = sin(M_PI * (dist / k->p0)); v = exp(-k->p1 * (s * s)); } break;
      case TGP_K_RQ: v = pow(1.0 + 0.5 * (sq / (k->p0 * k->p0)) / k->p1, -k->p1); break;
      default: v = NAN;
    }
    st[sp++] = v;
  }
  return st[0];
}

/* kernels/base.py:94-96 (+ noise.py:77-78 when diag != NULL): out (n1,n2) row-major */
void ref_kmat(const tgp_kop* prog, int nops, int64_t n1, int64_t n2, int d, const double* X1,
              const double* X2, const double* diag, double* out) {
  for (int64_t i = 0; i < n1; ++i)
    for (int64_t j = 0; j < n2; ++j) {
      double v = eval_prog(prog, nops, X1 + i * d, X2 + j * d, d);
      if (diag && i == j) v += diag[i];
      out[i * n2 + j] = v;
    }
}

/* solvers/direct.py:53 -- unblocked LL^T, row-major, in place, upper zeroed.
 * Returns 0 or the 1-based failing pivot (then the factor is NaN-filled like JAX's). */
int ref_cholesky_lower(int64_t n, double* A) {
  for (int64_t j = 0; j < n; ++j) {
    double s = A[j * n + j];
    for (int64_t k = 0; k < j; ++k) s -= A[j * n + k] * A[j * n + k];
    if (!(s > 0.0)) {
      for (int64_t t = 0; t < n * n; ++t) A[t] = NAN;
      return (int)(j + 1);
    }
    double ljj = sqrt(s);
    A[j * n + j] = ljj;
    for (int64_t i = j + 1; i < n; ++i) {
      double t = A[i * n + j];
      for (int64_t k = 0; k < j; ++k) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / ljj;
    }
    for (int64_t c = j + 1; c < n; ++c) A[j * n + c] = 0.0;
  }
  return 0;
}

/* solvers/direct.py:66-70 -- single right-hand side, in place */
void ref_trsv(int64_t n, const double* L, int transpose, double* y) {
  if (!transpose) {
    for (int64_t i = 0; i < n; ++i) {
      double t = y[i];
      for (int64_t k = 0; k < i; ++k) t -= L[i * n + k] * y[k];
      y[i] = t / L[i * n + i];
    }
  } else {
    for (int64_t i = n - 1; i >= 0; --i) {
      double t = y[i];
      for (int64_t k = i + 1; k < n; ++k) t -= L[k * n + i] * y[k];
      y[i] = t / L[i * n + i];
    }
  }
}

/* gp.py:313-320 + solvers/direct.py:61-64: the whole log_probability, resid = y - loc.
 * work is an n*n scratch matrix.  Returns the log-probability (-inf when not finite). */
double ref_log_probability(const tgp_kop* prog, int nops, int64_t n, int d, const double* X,
                           const double* diag, const double* resid, double* work) {
  ref_kmat(prog, nops, n, n, d, X, X, diag, work);
  ref_cholesky_lower(n, work);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)n);
  memcpy(alpha, resid, sizeof(double) * (size_t)n);
  ref_trsv(n, work, 0, alpha);
  double ss = 0.0, ld = 0.0;
  for (int64_t i = 0; i < n; ++i) { ss += alpha[i] * alpha[i]; ld += log(work[i * n + i]); }
  free(alpha);
  double ll = -0.5 * ss - (ld + 0.5 * (double)n * log(2.0 * M_PI));
  return isfinite(ll) ? ll : -INFINITY;
}
