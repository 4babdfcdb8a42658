// potf2_lds_check.cpp -- host replay of tinygp_amd/csrc/potf2_body.inc (test infrastructure, g++ only).
//
// The device text of potf2 is #included below UNCHANGED, with a tracking scalar type in place of
// double / float: every read or write of an element of the LDS arrays (S, Rs, Dg) is logged with the
// barrier phase (number of __syncthreads() the thread has crossed) and the wave of the thread.  The 512
// threads run one after the other (values are meaningless in that order -- control flow in potf2 does
// not depend on them), and at the end the log is checked:
//
//   * every thread crossed the same number of barriers;
//   * no LDS element is written by one wave and read or written by ANOTHER wave inside one barrier
//     phase (i.e. every inter-wave hand-off is ordered by a barrier, none by timing).
//
// Accesses of different lanes of the SAME wave are ordered by program order (a wave's LDS requests are
// served in order) and are not checked.  Exit code 0: clean; 1: conflicts (printed, first 20).
//
// Build variants (tests/test_potf2_lds.py): -DCHK_FOLD=0|1, -DCHK_FLOAT=0|1, and
// -DTGP_POTF2_UNORDERED_WRITEBACK, the round-2 order of the diagonal block, which MUST be reported.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <type_traits>
#include <vector>

#ifndef CHK_FOLD
#define CHK_FOLD 1
#endif
#ifndef CHK_FLOAT
#define CHK_FLOAT 0
#endif

namespace chk {

struct Access {
  int phase, wave, array, index;
  bool write;
};
std::vector<Access> g_log;
int g_phase = 0, g_tid = 0;
const char* g_lo[3] = {nullptr, nullptr, nullptr};
const char* g_hi[3] = {nullptr, nullptr, nullptr};
size_t g_elem = 8;

inline void touch(const void* p, bool write) {
  const char* c = static_cast<const char*>(p);
  for (int a = 0; a < 3; ++a)
    if (c >= g_lo[a] && c < g_hi[a]) {
      g_log.push_back(Access{g_phase, g_tid >> 6, a, int((c - g_lo[a]) / g_elem), write});
      return;
    }
}

// scalar whose every use as an operand / destination is visible
template <typename F>
struct Trk {
  F v;
  Trk() : v(0) {}
  Trk(F x) : v(x) {}
  Trk(int x) : v(F(x)) {}
  template <typename G, typename = typename std::enable_if<!std::is_same<F, G>::value && std::is_floating_point<G>::value>::type>
  Trk(G x) : v(F(x)) {}
  Trk(const Trk& o) : v(o.v) { touch(&o, false); }
  Trk& operator=(const Trk& o) {
    touch(&o, false);
    touch(this, true);
    v = o.v;
    return *this;
  }
#define CHK_COMPOUND(OP)                 \
  Trk& operator OP(const Trk& o) {       \
    touch(&o, false);                    \
    touch(this, false);                  \
    touch(this, true);                   \
    v OP o.v;                            \
    return *this;                        \
  }
  CHK_COMPOUND(+=) CHK_COMPOUND(-=) CHK_COMPOUND(*=)
#undef CHK_COMPOUND
  Trk operator-() const {
    touch(this, false);
    return Trk(-v);
  }
};
#define CHK_BIN(OP)                                             \
  template <typename F>                                         \
  Trk<F> operator OP(const Trk<F>& a, const Trk<F>& b) {        \
    touch(&a, false);                                           \
    touch(&b, false);                                           \
    return Trk<F>(a.v OP b.v);                                  \
  }
CHK_BIN(+) CHK_BIN(-) CHK_BIN(*)
#undef CHK_BIN
template <typename F>
bool operator>(const Trk<F>& a, const Trk<F>& b) {
  touch(&a, false);
  touch(&b, false);
  return a.v > b.v;
}

}  // namespace chk

#if CHK_FLOAT
using T = chk::Trk<float>;
#else
using T = chk::Trk<double>;
#endif

// ---- the device environment of potf2_body.inc -------------------------------------------------
struct {
  int x;
} threadIdx;
#define __syncthreads() (++chk::g_phase)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define POTF2_STAMP(i) ((void)0)
#define TGP_OPAQUE_VGPR(x) ((void)(x))
#define TGP_HD inline
#include "../tinygp_amd/csrc/potf2_layout.h"

struct acc_host {
  T v[4];
  acc_host() {}
  acc_host(int a, int b, int c, int d) {
    v[0] = T(a); v[1] = T(b); v[2] = T(c); v[3] = T(d);
  }
  T& operator[](int i) { return v[i]; }
  acc_host& operator+=(const acc_host& o) {
    for (int i = 0; i < 4; ++i) v[i] += o.v[i];
    return *this;
  }
};
template <typename X>
struct Mfma {
  using acc_t = acc_host;
  static acc_t mma(X a, X b, acc_t c) {
    (void)a; (void)b;
    return c;
  }
#if CHK_FLOAT
  static int drow(int lane, int r) { return (lane >> 4) * 4 + r; }
  static constexpr bool FAST4 = false;
  static int rot4(int lane, int t) { return (lane + t) & 15; }
  static int arow4(int lane) { return lane & 15; }
#else
  static int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
  static constexpr bool FAST4 = false;
  static int rot4(int lane, int t) { return 4 * ((((lane >> 2) & 3) + t) & 3) + (lane & 3); }
  static int arow4(int lane) { return 4 * ((lane >> 2) & 3) + (lane >> 4); }
#endif
  static X mma4(X a, X b, X c) {
    (void)a; (void)b;
    return c;
  }
};
static T readlane(T v, int) { return v; }
static T fast_rcp(T x) { return x; }
static T fast_rsqrt(T x) { return x; }
static int atomicCAS(int32_t* p, int cmp, int val) {
  const int old = *p;
  if (old == cmp) *p = val;
  return old;
}

static T g_S[36 * 256];
static T g_Rs[2 * 16];
static T g_Dg[256];

static void run_thread(T* A, int64_t ld, T* dinv, int32_t* info, int32_t pivot_base, const T* Xp, int64_t ldx) {
  constexpr bool FOLD = CHK_FOLD != 0;
  T* S = g_S;
  T* Rs = g_Rs;
  T* Dg = g_Dg;
#include "../tinygp_amd/csrc/potf2_body.inc"
}

int main() {
  chk::g_elem = sizeof(T);
  chk::g_lo[0] = reinterpret_cast<const char*>(g_S);
  chk::g_hi[0] = reinterpret_cast<const char*>(g_S + 36 * 256);
  chk::g_lo[1] = reinterpret_cast<const char*>(g_Rs);
  chk::g_hi[1] = reinterpret_cast<const char*>(g_Rs + 32);
  chk::g_lo[2] = reinterpret_cast<const char*>(g_Dg);
  chk::g_hi[2] = reinterpret_cast<const char*>(g_Dg + 256);
  const int64_t ld = 256;
  std::vector<T> A(size_t(ld) * 256, T(1.0)), dinv(8 * 256);
  int32_t info = 0;
  // the tile at (128, 128) of a 256 x 256 matrix; its pending-update operand is block column 0
  T* tile = A.data() + 128 * ld + 128;
  const T* Xp = CHK_FOLD ? A.data() + 128 : nullptr;
  int phases = -1;
  for (int tid = 0; tid < 512; ++tid) {
    threadIdx.x = tid;
    chk::g_tid = tid;
    chk::g_phase = 0;
    run_thread(tile, ld, dinv.data(), &info, 0, Xp, ld);
    if (phases < 0) phases = chk::g_phase;
    if (chk::g_phase != phases) {
      std::printf("thread %d crossed %d barriers, thread 0 crossed %d\n", tid, chk::g_phase, phases);
      return 1;
    }
  }
  // (phase, array, index) -> waves that wrote / read
  struct Cell {
    std::set<int> writers, readers;
  };
  std::map<std::tuple<int, int, int>, Cell> cells;
  for (const auto& a : chk::g_log) {
    Cell& c = cells[{a.phase, a.array, a.index}];
    (a.write ? c.writers : c.readers).insert(a.wave);
  }
  const char* names[3] = {"S", "Rs", "Dg"};
  int conflicts = 0;
  for (const auto& kv : cells) {
    const Cell& c = kv.second;
    if (c.writers.empty()) continue;
    std::set<int> others = c.readers;
    others.insert(c.writers.begin(), c.writers.end());
    if (c.writers.size() > 1 || others.size() > 1) {
      if (conflicts < 20) {
        const int idx = std::get<2>(kv.first);
        std::printf("conflict: phase %d %s[%d]", std::get<0>(kv.first), names[std::get<1>(kv.first)], idx);
        if (std::get<1>(kv.first) == 0) {
          const int b = idx / 256;
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= b) ++i;
          std::printf(" = block (%d, %d) element %d", i, b - i * (i + 1) / 2, idx % 256);
        }
        std::printf(" written by wave(s)");
        for (int w : c.writers) std::printf(" %d", w);
        std::printf(", read by wave(s)");
        for (int w : c.readers) std::printf(" %d", w);
        std::printf("\n");
      }
      ++conflicts;
    }
  }
  std::printf("potf2 LDS replay: fold=%d float=%d phases=%d accesses=%zu conflicts=%d\n", int(CHK_FOLD), int(CHK_FLOAT),
              phases, chk::g_log.size(), conflicts);
  return conflicts ? 1 : 0;
}
