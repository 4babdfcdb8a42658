/* TEST INFRASTRUCTURE (host, gcc): the division-free quotient of csrc/kmat.hip (UDiv: y = RN(1 / b) once, then one
 * multiply and four FMAs per dividend) against the division itself, bit for bit, on structured and random pairs:
 * divisors next to the all-ones significand (the one excluded case: reported separately), short significands,
 * dividends that are products b * q (quotients next to representable numbers) and b * (q + half an ulp) (next to
 * rounding boundaries).  usage: markstein_check [pairs]; exit status 1 on any mismatch outside the excluded case. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double mk(double a, double b, double y) {
  double q0 = a * y, r0 = fma(-b, q0, a), q1 = fma(r0, y, q0), r1 = fma(-b, q1, a);
  return fma(r1, y, q1);
}
static inline double frombits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 100000000L, bad = 0, bad1 = 0;
  for (long i = 0; i < n; ++i) {
    uint64_t mb = rnd() & 0xFFFFFFFFFFFFFull, ma = rnd() & 0xFFFFFFFFFFFFFull;
    int kind = i & 7;
    if (kind == 1) mb = 0xFFFFFFFFFFFFFull - (rnd() & 7);       // near all-ones
    if (kind == 2) mb = rnd() & 0xFFFF;                          // short significands
    if (kind == 3) mb = (rnd() & 0xFF) << 44;
    if (kind == 4) ma = 0xFFFFFFFFFFFFFull - (rnd() & 7);
    double b = frombits((uint64_t)(1023 + (int)(rnd() % 41) - 20) << 52 | mb);
    double a = frombits((uint64_t)(1023 + (int)(rnd() % 801) - 400) << 52 | ma);
    if (kind == 5) { double q = frombits((uint64_t)1023 << 52 | ma); a = b * q; }  // quotient near representable
    if (kind == 6) { double q = frombits((uint64_t)1023 << 52 | ma); a = fma(b, q, 0.5 * b * 0x1p-52); }  // near midpoint
    double y = 1.0 / b, t = a / b, m = mk(a, b, y);
    int allones = (mb == 0xFFFFFFFFFFFFFull);
    if (m != t) { if (allones) ++bad1; else { if (++bad < 10) printf("MISMATCH a=%a b=%a true=%a mk=%a\n", a, b, t, m); } }
  }
  printf("n=%ld mismatches=%ld (all-ones divisor mismatches=%ld)\n", n, bad, bad1);
  return bad != 0;
}
