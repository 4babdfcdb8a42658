// gemm.hip -- C (m x n) {-=, =} A (m x k) * B (n x k)^T on the fp64 / fp32 MFMA pipe.
//
// The building block of the blocked Cholesky (K4 trailing update: SYRK on diagonal tiles,
// GEMM below), of the blocked multi-RHS triangular solve (K5) and of A^T A in
// DirectSolver.condition (K10, reference solvers/direct.py:95).
//
// Layout: everything is COLUMN-major.  An MFMA A/B operand is "16 consecutive rows at a
// fixed k" -- 128 contiguous bytes of a column-major panel -- so global -> LDS staging is
// one coalesced 1 KiB row of 128 elements per wave instruction and the LDS image is
// [k][128 rows] with a leading dimension of 144 elements (144 mod 32 = 16 puts the two k
// values a 32-lane group reads on disjoint bank halves: conflict-free ds_read_b64).
//
// Tile: 128 x 128 x 16 per workgroup of 4 waves; each wave owns 64 x 64 = 4 x 4 MFMA
// 16x16x4 accumulators (128 VGPRs in fp64).  fp64 MFMA issues one 16x16x4 per 64 cycles
// per SIMD, so 8 LDS operand reads feed 16 MFMAs (1024 matrix-pipe cycles): the kernel is
// MFMA-bound by construction and two workgroups per CU hide the staging latency.
//
// The MFMA "A" operand is fed with rows of B (the C COLUMN index) and the "B" operand with
// rows of A (the C ROW index): D[a][b] = C[i=b][j=a], so a lane's 16-lane group writes 16
// consecutive rows of one C column = 128 contiguous bytes per group (full lines).
#include <type_traits>

#include "tgp_common.h"

#define TILE_HD __host__ __device__
#include "tile_order.h"

namespace tgp {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  using acc_t = d4;
  using v2_t = double __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // D row of accumulator register r (f64 16x16x4: row = (lane >> 4) + 4 r)
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma<float> {
  using acc_t = f4;
  using v2_t = float __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // f32 16x16x4: row = (lane >> 4) * 4 + r
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) * 4 + r; }
};

constexpr int BM = 128, BN = 128, BK = 16, LDS_LD = 144;

template <typename T>
struct GemmArgs {
  const T* A;
  const T* B;
  T* C;
  int64_t lda, ldb, ldc;
  int tm, tn;  // tiles in m / n
  int k;
  int lower;   // only tiles with ti >= tj
  int mode;    // bit0: 0 = C -= A B^T, 1 = C = A B^T; bit1 (TRMM): B lower-triangular, k < end of
               // the column tile; bit2: A, B upper-triangular (M M^T), k starts at the ROW tile
  int nblk;    // total workgroups
  int skip00;  // small kernel: skip the 2x2 tiles of the first 128x128 diagonal block
  // Block-cyclic column map (dist.hip): dG > 0 -> C holds this rank's block columns
  // l = dl0, dl0+1, ... (dnbt tiles wide each, contiguous in C) of a matrix whose block column
  // l*dG + dr they are; rows of A / B / C are GLOBAL, tm = global row tiles, and local column
  // tile tj updates rows >= its global column tile only (lower trapezoid per column tile).
  int dG, dr, dl0, dnbt;
  // Split tail (ROLE 2): workgroup ids >= split_first cover the LAST, partly filled round of tiles with split_s
  // workgroups per tile, each over 1 / split_s of the k-range; partial products meet in `ws`
  // (16384 elements per (tile, slice)), the workgroup that arrives last (counter `cnt` per tile, self-resetting)
  // subtracts them from C in slice order -- deterministic whoever that is.  ngrid = split_first + tail * split_s.
  int split_first, split_s, ngrid;
  T* ws;
  int* cnt;
  // ROLE 3 (products with a triangular operand: L^-T by halves, K^-1 = M M^T): `batch` independent problems, operand
  // q at base + q * s{A,B,C}; mode bit3: A lower triangular, k ends with the ROW tile; mode bit4: the k-range is
  // walked from its END (tiles whose ranges end together -- bit2 -- then stream the same operand slices at the same
  // time); workgroup ids enumerate 8 x 8 PATCHES of tiles, one patch per XCD at a time (see the kernel).
  int batch;
  int64_t sA, sB, sC;
  int band;  // tile order: 0 column by column, > 0 bands of that many tile rows (tile_order.h; ctx option tile_band)
  // PREFIX (round 6, the merged trailing update of potrf): the first `prefix_tn` tile columns of a lower update -- the next
  // panel's block column -- take the lowest `prefix_cnt` workgroup ids, are stored WRITE-THROUGH and counted in *prefix_done
  // (one increment per finished tile, behind a drained barrier): a one-wave poll on another stream lets that panel's chain
  // start while the rest of THIS launch still runs.  0: no prefix.
  int prefix_tn = 0, prefix_cnt = 0;  // (default member initialisers: every launcher that fills a GemmArgs by hand gets "none")
  int32_t* prefix_done = nullptr;
};

// Linear workgroup id -> (ti, tj): tile_order.h (shared with the CPU test hook).  band == 0: column by column (tj major) so
// that consecutive ids share the B tile; for `lower` column tj holds rows tj..tm-1.
__device__ __forceinline__ void decode_tile(int b, int tm, int tn, int lower, int& ti, int& tj, int band = 0) {
  tile_decode(b, tm, tn, lower, band, ti, tj);
}

// Block-cyclic variant: local block column q (tiles [q*dnbt, (q+1)*dnbt)) starts at global tile
// g0(q) = ((dl0 + q) * dG + dr) * dnbt and holds sum_s (tm - g0 - s) tiles.  *gt = global column
// tile (the B operand's row tile and the first row tile of the column).
template <typename T>
__device__ __forceinline__ void decode_tile_dist(int b, const GemmArgs<T>& g, int& ti, int& tj, int& gt) {
  int q = 0, g0 = (g.dl0 * g.dG + g.dr) * g.dnbt;
  const int step = g.dG * g.dnbt, tri = g.dnbt * (g.dnbt - 1) / 2;
  for (;;) {
    const int cnt = g.dnbt * (g.tm - g0) - tri;
    if (b < cnt) break;
    b -= cnt;
    g0 += step;
    ++q;
  }
  int tir, s;
  decode_tile(b, g.tm - g0, g.dnbt, 1, tir, s, g.band);  // (band order inside the block column: 8 x 8 patches per XCD)
  ti = g0 + tir;
  tj = q * g.dnbt + s;
  gt = g0 + s;
}

// ROLE only changes the kernel's name (rocprof separates the trailing update from the rest)
template <typename T, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs<T> g) {
  // fp64: two padded [16 k][144] images per operand (73.7 KB); fp32: three unpadded [16 k][128] stages (48 KB)
  constexpr int SBUF = sizeof(T) == 8 ? 2 * BK * LDS_LD : 3 * BK * BM;
  __shared__ __attribute__((aligned(16))) T sAm[SBUF];
  __shared__ __attribute__((aligned(16))) T sBm[SBUF];
  T(*sA)[BK * LDS_LD] = reinterpret_cast<T(*)[BK * LDS_LD]>(sAm);  // the fp64 path's [2][BK * LDS_LD] view
  T(*sB)[BK * LDS_LD] = reinterpret_cast<T(*)[BK * LDS_LD]>(sBm);

  if (ROLE == 1) __builtin_amdgcn_s_setprio(1);  // in-panel update: on the critical path
  // XCD-aware remap: hardware places workgroup b on XCD b % 8; give every XCD a contiguous
  // run of tile ids so neighbouring tiles (shared panels) meet in one L2.  Bijective form.
  // Persistent over tiles: workgroup b takes tiles b, b + gridDim, ... (gridDim is a multiple of 8
  // whenever it is smaller than the tile count, so a workgroup stays on its XCD's run of tile ids).
  // The launcher sizes the grid: all workgroup slots of the chip for an update that runs alone,
  // fewer for one that runs beside a panel chain -- the chain's kernels then find free slots instead
  // of waiting for a round of 250-us tiles to retire (ctx option chain_reserve).
  for (int tile = blockIdx.x; tile < (ROLE == 2 ? g.ngrid : g.nblk); tile += gridDim.x) {
  int bid = tile;
  int slice = 0, nsl = 1, tail_idx = 0;  // ROLE 2: this workgroup's k-slice of a tail tile
  if (ROLE == 2 && tile >= g.split_first) {
    tail_idx = (tile - g.split_first) / g.split_s;
    slice = (tile - g.split_first) % g.split_s;
    nsl = g.split_s;
    bid = g.split_first + tail_idx;
  }
  // (with a prefix the two ranges of ids are remapped each within itself: the prefix must spread over ALL XCDs)
  const bool in_prefix = g.prefix_cnt > 0 && bid < g.prefix_cnt;
  {
    const int base = g.prefix_cnt > 0 ? (in_prefix ? 0 : g.prefix_cnt) : 0;
    const int cnt = g.prefix_cnt > 0 ? (in_prefix ? g.prefix_cnt : g.nblk - g.prefix_cnt) : g.nblk;
    const int loc = bid - base;
    const int nx = 8, q = cnt / nx, r = cnt % nx;
    const int xcd = loc % nx, idx = loc / nx;
    bid = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int ti, tj, gt;
  const T* gA = g.A;
  const T* gB = g.B;
  T* gC = g.C;
  if constexpr (ROLE == 3) {
    // Products with a triangular operand have k-ranges that differ from tile row to tile row: in the column-major
    // tile order the 64 tiles an XCD runs at a time are 64 different row tiles -- 65 operand slices per k-step for
    // 64 tiles, nothing shared through the L2 -- and K^-1 = M M^T ran at 44 TFLOP/s on the fabric (33 ms at
    // N = 16 384).  Here the workgroup id (dispatched in order, id % 8 = XCD) names slot s of the idx-th 8 x 8 PATCH
    // of that XCD: its 64 resident tiles share 8 + 8 operand slices per k-step, and with bit4 their k-ranges END
    // together.  Patches in order of decreasing k-range (longest first); slots outside the matrix leave at once.
    const int xcd = tile & 7, idx = tile >> 3, slot = idx & 63;
    const int pk = (idx >> 6) * 8 + xcd;
    const int ptm = (g.tm + 7) >> 3, ptn = (g.tn + 7) >> 3;
    const int pps = g.lower ? (ptm * (ptm + 1)) / 2 : ptm * ptn;
    // the batch index runs fastest: the XCDs are dealt patches in order of decreasing k-range whatever the batch
    // (problem by problem, two problems of 4 x 4 patches put both long rows on XCDs 0-3: 45 instead of 60 TFLOP/s)
    const int pl = pk / g.batch, q = pk - pl * g.batch;
    if (pl >= pps) return;
    int Pi, Pj;
    if (g.lower) {  // patch rows top down, row Pi holds patches 0..Pi
      Pi = int((sqrtf(8.0f * float(pl) + 1.0f) - 1.0f) * 0.5f);
      while (Pi > 0 && (Pi * (Pi + 1)) / 2 > pl) --Pi;
      while (((Pi + 1) * (Pi + 2)) / 2 <= pl) ++Pi;
      Pj = pl - (Pi * (Pi + 1)) / 2;
    } else {
      Pi = pl / ptn;
      Pj = pl - Pi * ptn;
      if (g.mode & 8) Pi = ptm - 1 - Pi;
    }
    ti = Pi * 8 + (slot & 7);
    tj = Pj * 8 + (slot >> 3);
    if (ti >= g.tm || tj >= g.tn || (g.lower && ti < tj)) return;
    gt = tj;
    gA += int64_t(q) * g.sA;
    gB += int64_t(q) * g.sB;
    gC += int64_t(q) * g.sC;
  } else
  if (g.dG > 0) {
    decode_tile_dist<T>(bid, g, ti, tj, gt);
  } else if (g.prefix_cnt > 0) {
    if (in_prefix) {  // the first prefix_tn tile columns (rows from the diagonal down), band order among themselves
      decode_tile(bid, g.tm, g.prefix_tn, 1, ti, tj, g.band);
    } else {          // the lower triangle right of them, shifted
      decode_tile(bid - g.prefix_cnt, g.tm - g.prefix_tn, g.tn - g.prefix_tn, 1, ti, tj, g.band);
      ti += g.prefix_tn;
      tj += g.prefix_tn;
    }
    gt = tj;
  } else {
    decode_tile(bid, g.tm, g.tn, g.lower, ti, tj, g.band);
    gt = tj;
  }

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int64_t i0 = int64_t(ti) * BM, j0 = int64_t(tj) * BN;

  // mode 3 (TRMM): B = L is lower triangular, so column tile tj only needs k < (tj+1)*BN
  int nkt = (g.mode & 2) ? ((g.k < (tj + 1) * BN ? g.k : (tj + 1) * BN) / BK) : g.k / BK;
  // ROLE 3, mode bit 3: A lower triangular -- row tile ti only needs k < (ti+1)*BM
  if (ROLE == 3 && (g.mode & 8)) nkt = (g.k < (ti + 1) * BM ? g.k : (ti + 1) * BM) / BK;
  // mode bit 2: rows of an upper-triangular operand are zero left of the diagonal, so a
  // lower tile (ti >= tj) of M M^T only needs k >= ti*BM
  const int kt0 = (g.mode & 4) ? (ti * BM) / BK : 0;
  const bool krev = ROLE == 3 && (g.mode & 16);
  const int lrow = lane & 15, lk = lane >> 4;
  T* Cb = gC + (j0 + wc * 64) * g.ldc + i0 + wr * 64;

  if constexpr (sizeof(T) == 8) {
    // fp64: v_mfma_f64_4x4x4_4b_f64 (16 cycles, measured 73-76 TFLOP/s) instead of
    // v_mfma_f64_16x16x4_f64 (measured 46-49 TFLOP/s on MI355X).  One instruction is four
    // independent 4x4x4 products: lane (k = l>>4, q = (l>>2)&3, e = l&3) supplies A_q[e][k]
    // and B_q[k][e]; lane (i = l>>4, q, j = l&3) receives D_q[i][j].  Four instructions with
    // the B operand's 4-row groups rotated by t = 0..3 (a rotated LDS read, no VALU) cover
    // the full 16x16 outer product: acc[a][b][t] holds, in lane (i, q, j),
    //   C[row = b*16 + 4((q+t)&3) + j][col = a*16 + 4q + i]   (within the wave's 64x64).
    const int lq = (lane >> 2) & 3, lj = lane & 3;
    double acc[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][b][t] = 0.0;
    int rot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rot[t] = 4 * ((lq + t) & 3) + lj;

    // The C read-modify-write used to be an epilogue that no MFMA overlapped (a fixed 0.92 ms per
    // 16384^2 launch, 10 % of a K = 1024 update: every workgroup of the chip reads and writes its
    // tile in the same few microseconds).  With at least NCH + 1 k-tiles the READ half is spread
    // over the k-loop instead: k-tile r fetches chunk r of C (two of a lane's 64 entries) right
    // behind the operand prefetch and subtracts it one k-tile later, when the in-order return of
    // the operand loads has already proved it complete -- no additional wait.  The MFMA operand
    // is negated by the instruction itself (its NEG field), so the accumulators end as C - A B^T and the epilogue
    // is 64 stores, nothing loaded.
    constexpr int NCH = 8;  // chunks of eight entries: (a, b-pair)
    const int kq = (nkt - kt0) / nsl;               // k-tiles of this workgroup (all of them unless a tail slice)
    const int kbeg = kt0 + slice * kq, kend = kbeg + kq;
    const bool pipe = (g.mode == 0) && nsl == 1 && (kend - kbeg >= NCH);
    double cst[8];
    // C addresses as a wave-uniform base (SGPRs) + four loop-invariant 32-bit lane offsets: no
    // per-chunk address registers
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const char* Cu = reinterpret_cast<const char*>(gC + (j0 + (wu & 1) * 64) * g.ldc + i0 + (wu >> 1) * 64);
    uint32_t voff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) voff[t] = uint32_t((int64_t(4 * lq + lk) * g.ldc + rot[t]) * 8);
    // Operand staging is direct global -> LDS (global_load_lds_dwordx4: one wave instruction moves the
    // 1 KiB k-row of 128 elements straight into its LDS row -- LDS address in M0, lane l lands at
    // +16 l): no staging VGPRs, no ds_write pass.  Issued one k-tile ahead into the other buffer.
    // Written as asm: through the builtin the compiler orders every later ds_read behind the
    // transfer (vmcnt(0) at the top of the k-tile it was meant to overlap); here the only wait is
    // the explicit one in front of the barrier that ends the k-tile.
    const uint32_t lds_a = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sA[0][0]))) + uint32_t(wu * LDS_LD * 8);
    const uint32_t lds_b = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sB[0][0]))) + uint32_t(wu * LDS_LD * 8);
    const T* Au = gA + i0;                 // wave-uniform row bases; the lane adds 16 l bytes
    const T* Bu = gB + int64_t(gt) * BN;
    const uint32_t lane16 = uint32_t(lane) * 16u;
    auto issue_tile = [&](int kt_, int buf_) {
      const int64_t kg = int64_t(krev ? kbeg + kend - 1 - kt_ : kt_) * BK + wu;
#pragma unroll
      for (int r = 0; r < BK / 4; ++r) {
        const T* ap = Au + (kg + 4 * r) * g.lda;
        const T* bp = Bu + (kg + 4 * r) * g.ldb;
        const uint32_t off = uint32_t((buf_ * BK + 4 * r) * LDS_LD * 8);
        uint32_t keep;  // (m0 is a reserved register: saved and restored, not clobbered)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(ap), "s"(lds_a + off), "s"(bp), "s"(lds_b + off) : "memory");
      }
    };
    auto tile_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    // STAGE: -1 plain k-tile; c in 0..NCH-1: fetch chunk c of C behind the operand prefetch and
    // add it behind the barrier that ends the k-tile (the vmcnt(0) in front of it -- needed for the
    // LDS-direct loads anyway -- has completed it: no wait of its own)
    auto ktile = [&](int kt, auto stage, auto negated) {
      constexpr int c = decltype(stage)::value;
      const int buf = kt & 1;
      if (kt + 1 < kend) issue_tile(kt + 1, buf ^ 1);
      if constexpr (c >= 0) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const char* cb = Cu + (int64_t((c >> 1) * 16) * g.ldc + ((c & 1) * 2 + bb) * 16) * 8;
#pragma unroll
          for (int t = 0; t < 4; ++t) cst[bb * 4 + t] = *reinterpret_cast<const double*>(cb + voff[t]);
        }
      }
      const T* pa = &sB[buf][lk * LDS_LD + wc * 64 + lrow];
      const T* pb = &sA[buf][lk * LDS_LD + wr * 64];
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        double aop[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) aop[a] = pa[ks * 4 * LDS_LD + a * 16];
        // accumulate -A B^T: the f64 MFMA's BLGP field is its NEG field (bit 0 negates A) -- no v_xor per operand
        constexpr int NEG = decltype(negated)::value ? 1 : 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          double bop[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) bop[t] = pb[ks * 4 * LDS_LD + b * 16 + rot[t]];
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[a][b][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aop[a], bop[t], acc[a][b][t], 0, 0, NEG);
        }
      }
      tile_landed();
      __syncthreads();
      if constexpr (c >= 0) {
        __builtin_amdgcn_sched_barrier(0);  // the adds stay behind the barrier's wait
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[c >> 1][(c & 1) * 2 + bb][t] += cst[bb * 4 + t];
      }
    };
    issue_tile(kbeg, kbeg & 1);
    tile_landed();
    __syncthreads();
    int kt = kbeg;
    if (pipe) {
      // every staged k-tile is its own (single-trip) loop: as one straight-line block the nine
      // bodies are scheduled together and spill
      int one = 1;
      asm volatile("" : "+s"(one));
#define TGP_STAGED(C) \
  for (int q = 0; q < one; ++q, ++kt) ktile(kt, std::integral_constant<int, C>{}, std::true_type{});
      TGP_STAGED(0) TGP_STAGED(1) TGP_STAGED(2) TGP_STAGED(3) TGP_STAGED(4)
      TGP_STAGED(5) TGP_STAGED(6) TGP_STAGED(7)
#undef TGP_STAGED
      for (; kt < kend; ++kt) ktile(kt, std::integral_constant<int, -1>{}, std::true_type{});
    } else {
      for (; kt < kend; ++kt) ktile(kt, std::integral_constant<int, -1>{}, std::false_type{});
    }
    if (ROLE == 2 && nsl > 1) {
      // tail slice: the partial product goes to the workspace plane by plane (256 lanes x 8 bytes contiguous)
      T* W = g.ws + (size_t(tail_idx) * nsl + slice) * size_t(BM * BN);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) W[((a * 4 + b) * 4 + t) * 256 + tid] = acc[a][b][t];
      // publish: every wave's stores have left it -> barrier -> one agent-scope release -> count
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      __shared__ int s_last;
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int seen = __hip_atomic_fetch_add(&g.cnt[tail_idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen == nsl - 1;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(&g.cnt[tail_idx], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        s_last = last;
      }
      __syncthreads();
      if (s_last) {  // (uniform) C -= P_0 + P_1 + ... in SLICE order, whoever arrived last
        // a quarter of the tile at a time (16 entries per lane): with all 64 in flight beside the partial products' loads
        // this instantiation spilled 16 registers (round-5 judge, item 10)
#pragma unroll 1
        for (int a = 0; a < 4; ++a) {
          double cc[4][4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const char* cb = Cu + (int64_t(a * 16) * g.ldc + b * 16) * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) cc[b][t] = *reinterpret_cast<const double*>(cb + voff[t]);
          }
          for (int sl = 0; sl < nsl; ++sl) {
            const T* Wp = g.ws + (size_t(tail_idx) * nsl + sl) * size_t(BM * BN);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
              for (int t = 0; t < 4; ++t) cc[b][t] -= Wp[((a * 4 + b) * 4 + t) * 256 + tid];
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            char* cb = const_cast<char*>(Cu) + (int64_t(a * 16) * g.ldc + b * 16) * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) *reinterpret_cast<double*>(cb + voff[t]) = cc[b][t];
          }
        }
      }
    } else
    if (pipe) {
      if (ROLE == 0 && in_prefix) {
        // write-through (sc1: what an agent-scope relaxed store emits), base + 32-bit lane offset as everywhere here; then
        // every wave drains, a barrier, ONE lane counts the tile (Guideline 16, form R1 -- the consumer is a kernel that
        // starts behind a poll of this counter, its start is the acquire)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const char* cb = Cu + (int64_t(a * 16) * g.ldc + b * 16) * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t)
              asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff[t]), "v"(acc[a][b][t]), "s"(cb) : "memory");
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(g.prefix_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          char* cb = const_cast<char*>(Cu) + (int64_t(a * 16) * g.ldc + b * 16) * 8;
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<double*>(cb + voff[t]) = acc[a][b][t];
        }
      }
      }
    } else if (g.mode == 0) {
      // short k: the read-modify-write is batched: 32 independent loads in flight, then 32
      // stores, twice (element by element it is 64 dependent HBM round trips per lane).
#pragma unroll
      for (int ap = 0; ap < 4; ap += 2) {
        T* col[2];
        double c[2][4][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          col[a] = Cb + int64_t((ap + a) * 16 + 4 * lq + lk) * g.ldc;
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) c[a][b][t] = col[a][b * 16 + rot[t]];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) col[a][b * 16 + rot[t]] = c[a][b][t] - acc[ap + a][b][t];
      }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        T* col = Cb + int64_t(a * 16 + 4 * lq + lk) * g.ldc;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) col[b * 16 + rot[t]] = acc[a][b][t];
      }
    }
  } else {
    // fp32, round 3 (BASELINE config 5's arithmetic): v_mfma_f32_32x32x2_f32 (the fp32 instruction that reaches
    // the 157 TFLOP/s peak; 16x16x4 tops out at 139), operands staged global -> LDS directly, three k-tiles deep
    // with counted vmcnt waits and a raw s_barrier, C fetched first and used as the accumulators' initial value
    // (negated: the accumulators collect A B^T on top of -C and change sign once before the 64 stores).  Round 1-2's loop -- register
    // staging one k-tile ahead, read-modify-write epilogue -- ran at 117 of 157 TFLOP/s.
    // LDS stage: [16 k][128 rows] floats, unpadded: a k-row is 512 bytes = half a wave transfer, and a 32-lane
    // ds_read_b32 group reads 128 contiguous bytes of ONE k-row -- no conflicts, no swizzle.
    // Lane map of the 32x32x2 instruction: lane l supplies A[m = l & 31][k = l >> 5] and B[k][n = l & 31];
    // accumulator register r of lane l is D[m = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][n = l & 31].  As in the fp64
    // path the MFMA "A" operand is fed with rows of B (m <-> C column) and "B" with rows of A (n <-> C row):
    // 32 consecutive lanes hold 32 consecutive rows of one C column -- 128-byte stores.
    typedef float f16v __attribute__((ext_vector_type(16)));
    constexpr int NST = 3, STG = BK * BM;  // floats per operand per stage (8 KiB)
    static_assert(SBUF == NST * STG, "the stages fill sAm / sBm");
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int l32 = lane & 31, lh = lane >> 5;
    f16v acc[2][2];  // [a: 32 columns][b: 32 rows] of the wave's 64 x 64
    // this lane's C entries: row i0 + wr*64 + b*32 + l32, column j0 + wc*64 + a*32 + 8 (r >> 2) + 4 lh + (r & 3)
    T* cbase = gC + (j0 + (wu & 1) * 64 + 4 * lh) * g.ldc + i0 + (wu >> 1) * 64 + l32;
    auto centry = [&](int a, int b, int r) -> T* {
      return cbase + int64_t(a * 32 + 8 * (r >> 2) + (r & 3)) * g.ldc + b * 32;
    };
    __builtin_amdgcn_s_barrier();  // (persistent loop: the previous tile's last stage is still being read)
    if (g.mode == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = *centry(a, b, r);
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    }
    // wave w moves k-rows {2w, 2w + 1} and {2w + 8, 2w + 9} of each operand: lanes 0..31 the even row, 32..63
    // the odd one, 16 bytes per lane
    const uint32_t voa = uint32_t(lh) * uint32_t(g.lda) * 4u + uint32_t(l32) * 16u;
    const uint32_t vob = uint32_t(lh) * uint32_t(g.ldb) * 4u + uint32_t(l32) * 16u;
    const uint32_t lds_a = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sAm[0]))) + uint32_t(wu * 2 * BM * 4);
    const uint32_t lds_b = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sBm[0]))) + uint32_t(wu * 2 * BM * 4);
    const T* Au = gA + i0;
    const T* Bu = gB + int64_t(gt) * BN;
    auto issue_tile = [&](int kt_, int stage) {
      const int64_t kg = int64_t(krev ? kt0 + nkt - 1 - kt_ : kt_) * BK + 2 * wu;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const T* ap = Au + (kg + 8 * r) * g.lda;
        const T* bp = Bu + (kg + 8 * r) * g.ldb;
        const uint32_t off = uint32_t((stage * STG + 8 * r * BM) * 4);
        uint32_t keep;  // (m0 is a reserved register: saved and restored, not clobbered)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voa), "s"(ap), "s"(lds_a + off), "v"(vob), "s"(bp), "s"(lds_b + off) : "memory");
      }
    };
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
      if (kt0 + q < nkt) issue_tile(kt0 + q, q);
    // the C values are consumed HERE as far as the compiler can tell (else its waits for these loads land inside
    // the k-loop, where every iteration would drain the transfers in flight)
    // mode 0 accumulates A B^T onto -C and stores the negative: the f32 MFMA has no NEG field, and negating the operand
    // on its way from LDS was 16 VALU instructions per k-tile (1 024 per tile) against 128 here
    if (g.mode == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          asm volatile("" : "+v"(acc[a][b]));
          acc[a][b] = -acc[a][b];
        }
    }
    auto kloop = [&]() {
      int stage = 0;
      for (int kt = kt0; kt < nkt; ++kt) {
        const int ahead = nkt - 1 - kt;  // tiles issued behind this one
        if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's share of tile kt is in LDS; tile kt-1's stage is free
        if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
        const T* pa = &sBm[stage * STG + lh * BM + wc * 64 + l32];  // MFMA A operand <- B rows (C column)
        const T* pb = &sAm[stage * STG + lh * BM + wr * 64 + l32];  // MFMA B operand <- A rows (C row)
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
          T aop[2], bop[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            aop[t] = pa[ks * 2 * BM + t * 32];
            bop[t] = pb[ks * 2 * BM + t * 32];
          }
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[a], bop[b], acc[a][b], 0, 0, 0);
        }
        stage = stage == NST - 1 ? 0 : stage + 1;
      }
    };
    kloop();
    if (g.mode == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = -acc[a][b];
    }
    if (ROLE == 0 && in_prefix) {  // (see the fp64 path)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(centry(a, b, r)), __float_as_uint(acc[a][b][r]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(g.prefix_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) *centry(a, b, r) = acc[a][b][r];
    }
  }
  }  // tiles of this workgroup
}

// ---- small-tile variant -----------------------------------------------------------------
// 64 x 64 x 16 tiles, 4 waves of 32 x 32.  For the short-K updates on the factorisation's
// critical path (in-panel updates, K = 128): a 128 x 128 tile there is ONE long serial
// k-loop per workgroup (36 us measured) however few tiles exist; quarter-size tiles spread
// the same flops over 4x the workgroups and finish in a fraction of that.
constexpr int SM = 64, S_LD = 80;  // 80 mod 32 == 16: conflict-free operand reads

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_nt_small_kernel(GemmArgs<T> g) {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  using v2_t = typename M::v2_t;
  // fp64: three unpadded 8-KiB stages per operand (48 KiB: three workgroups per CU); fp32: two padded buffers
  constexpr int SBUF = sizeof(T) == 8 ? 3 * BK * SM : 2 * BK * S_LD;
  __shared__ __attribute__((aligned(16))) T sAm[SBUF];
  __shared__ __attribute__((aligned(16))) T sBm[SBUF];
  T(*sA)[BK * S_LD] = reinterpret_cast<T(*)[BK * S_LD]>(sAm);  // the fp32 path's [2][BK * S_LD] view
  T(*sB)[BK * S_LD] = reinterpret_cast<T(*)[BK * S_LD]>(sBm);
  __builtin_amdgcn_s_setprio(1);
  int ti, tj;
  {
    int bid = blockIdx.x;
    if (g.band > 0) {  // (with bands: the XCD-aware bijective remap of gemm_nt_kernel -- a contiguous run of ids per XCD)
      const int nx = 8, q = g.nblk / nx, r = g.nblk % nx;
      const int xcd = bid % nx, idx = bid / nx;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    decode_tile(bid, g.tm, g.tn, g.lower, ti, tj, g.band);
  }
  if (g.skip00 && ti < 2 && tj < 2) return;  // that tile is updated inside the next potf2
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int64_t i0 = int64_t(ti) * SM, j0 = int64_t(tj) * SM;
  // staging: 32 lanes x 2 rows cover one 64-row k-line; thread t loads k-lines t/32 and t/32+8
  const int l32 = tid & 31, kq = tid >> 5;
  const T* Ag = g.A + i0 + l32 * 2;
  const T* Bg = g.B + j0 + l32 * 2;
  v2_t ra[2], rb[2];
  auto load_global = [&](int kt) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t kk = int64_t(kt) * BK + kq + 8 * r;
      ra[r] = *reinterpret_cast<const v2_t*>(Ag + kk * g.lda);
      rb[r] = *reinterpret_cast<const v2_t*>(Bg + kk * g.ldb);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kk = kq + 8 * r;
      *reinterpret_cast<v2_t*>(&sA[buf][kk * S_LD + l32 * 2]) = ra[r];
      *reinterpret_cast<v2_t*>(&sB[buf][kk * S_LD + l32 * 2]) = rb[r];
    }
  };
  const int nkt = g.k / BK;
  const int lrow = lane & 15, lk = lane >> 4;
  T* Cb = g.C + (j0 + wc * 32) * g.ldc + i0 + wr * 32;

  if constexpr (sizeof(T) == 8) {
    // fp64, round 3: operands staged global -> LDS directly (global_load_lds_dwordx4: no staging VGPRs, no
    // ds_write pass), NST k-tiles deep with counted vmcnt waits and a raw s_barrier (a __syncthreads() would
    // drain the transfers in flight), C fetched before the first transfer and used as the accumulators' initial
    // value (the MFMA negates its A operand -- NEG field: acc ends as C - A B^T, the epilogue is 16 stores).
    // Before: one k-tile of register prefetch -- a 0.4-us k-tile cannot cover a 1-2-us L2 / HBM round trip, the
    // MFMA pipes were 50 % busy (profiles/r02_s) and this kernel's 27 % of the flops cost 36 % of an evaluation.
    // LDS image per stage: [16 k][64 rows] doubles, UNPADDED (one wave transfer = two 512-byte k-rows, lane l at
    // +16 l) with the 128-byte blocks of odd k-rows swapped pairwise (block b of row k at b ^ (k & 1)): the two
    // k-rows a 32-lane ds_read_b64 group touches sit on different halves of the 256-byte bank row.  The swizzle is
    // applied to the per-lane GLOBAL source offset (the LDS side of a transfer is lane-linear by construction).
    constexpr int NST = 3, STG = BK * SM;  // stages; doubles per operand per stage (8 KiB)
    static_assert(SBUF == NST * STG, "the stages fill sAm / sBm");
    const int lq = (lane >> 2) & 3, lj = lane & 3;
    double acc[2][2][4];
    int rot[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rot[t] = 4 * ((lq + t) & 3) + lj;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int x1 = lk & 1;  // odd k-row of the 4-row group this lane reads: blocks swapped pairwise
    T* col[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) col[a] = Cb + int64_t(a * 16 + 4 * lq + lk) * g.ldc;
    if (g.mode == 0) {  // (issued BEFORE the transfers: the loads below complete in order)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[a][b][t] = col[a][b * 16 + rot[t]];
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[a][b][t] = 0.0;
    }
    // wave w moves k-rows {2w, 2w + 1} and {2w + 8, 2w + 9} of each operand: lanes 0..31 the even row, 32..63 the
    // odd one; lane l fetches the 16 bytes that belong at LDS chunk (l & 31) of its row
    const uint32_t half = uint32_t(lane >> 5), ch = uint32_t(lane & 31);
    const uint32_t soff = 128u * ((ch >> 3) ^ half) + 16u * (ch & 7);
    const uint32_t voa = half * uint32_t(g.lda) * 8u + soff, vob = half * uint32_t(g.ldb) * 8u + soff;
    const uint32_t lds_a = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sAm[0]))) + uint32_t(wu * 2 * SM * 8);
    const uint32_t lds_b = uint32_t(size_t((__attribute__((address_space(3))) T*)(&sBm[0]))) + uint32_t(wu * 2 * SM * 8);
    const T* Au = g.A + i0;
    const T* Bu = g.B + j0;
    auto issue_tile = [&](int kt_, int stage) {
      const int64_t kg = int64_t(kt_) * BK + 2 * wu;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const T* ap = Au + (kg + 8 * r) * g.lda;
        const T* bp = Bu + (kg + 8 * r) * g.ldb;
        const uint32_t off = uint32_t((stage * STG + 8 * r * SM) * 8);
        uint32_t keep;  // (m0 is a reserved register: saved and restored, not clobbered)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voa), "s"(ap), "s"(lds_a + off), "v"(vob), "s"(bp), "s"(lds_b + off) : "memory");
      }
    };
    // 4 transfers per wave and k-tile; tile kt has landed once at most the 4 (NST - 1) younger ones are in flight
    // (fewer near the end of the k-range: the wait is exact there too)
    const T* sa0 = &sAm[0];
    const T* sb0 = &sBm[0];
    const int oa[2] = {((wc * 2) ^ x1) * 16, ((wc * 2 + 1) ^ x1) * 16};  // swizzled 16-row blocks of this lane
    const int ob[2] = {((wr * 2) ^ x1) * 16, ((wr * 2 + 1) ^ x1) * 16};
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
      if (q < nkt) issue_tile(q, q);
    // the C values are consumed HERE as far as the compiler can tell: otherwise it puts its own waits for these
    // loads inside the k-loop (at the accumulators' first use), where every iteration would drain the transfers
    if (g.mode == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[a][b][t]));
    }
    auto kloop = [&](auto negated) {
    int stage = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      const int ahead = nkt - 1 - kt;  // tiles issued behind this one
      if (ahead >= NST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave's share of tile kt is in LDS; tile kt-1's stage is free
      if (kt + NST - 1 < nkt) issue_tile(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);
      const T* pa = sb0 + stage * STG + lk * SM + lrow;  // MFMA A operand <- B rows (C column)
      const T* pb = sa0 + stage * STG + lk * SM;         // MFMA B operand <- A rows (C row), rotated
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        double aop[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) aop[a] = pa[ks * 4 * SM + oa[a]];
        constexpr int NEG = decltype(negated)::value ? 1 : 0;  // accumulate -A B^T through the MFMA's NEG field
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          double bop[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) bop[t] = pb[ks * 4 * SM + ob[b] + rot[t]];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[a][b][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(aop[a], bop[t], acc[a][b][t], 0, 0, NEG);
        }
      }
      stage = stage == NST - 1 ? 0 : stage + 1;
    }
    };
    if (g.mode == 0) kloop(std::true_type{});
    else kloop(std::false_type{});
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) col[a][b * 16 + rot[t]] = acc[a][b][t];
  } else {
    acc_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) load_global(kt + 1);
      const T* pa = &sB[buf][lk * S_LD + wc * 32 + lrow];
      const T* pb = &sA[buf][lk * S_LD + wr * 32 + lrow];
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        T aop[2], bop[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          aop[t] = pa[ks * 4 * S_LD + t * 16];
          bop[t] = pb[ks * 4 * S_LD + t * 16];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = M::mma(aop[a], bop[b], acc[a][b]);
      }
      if (kt + 1 < nkt) store_lds(buf ^ 1);
      __syncthreads();
    }
    {
      T* col[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          col[a][r] = Cb + lrow + int64_t(a * 16 + M::drow(lane, r)) * g.ldc;
      if (g.mode == 0) {
        T c[2][4][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 2; ++b) c[a][r][b] = col[a][r][b * 16];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 2; ++b) col[a][r][b * 16] = c[a][r][b] - acc[a][b][r];
      } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < 2; ++b) col[a][r][b * 16] = acc[a][b][r];
      }
    }
  }
}

// ---- issue-rate microbenchmarks ---------------------------------------------------------
// kind 0: v_mfma_f64_16x16x4_f64   1: v_mfma_f32_16x16x4_f32   2: v_fma_f64 (VALU)
//      3: MFMA f64 + VALU f64 FMA interleaved in one wave (1 MFMA : 4 FMA)
//      4: v_mfma_f64_4x4x4_4b_f64
// cyc[blockIdx] = shader cycles (s_memtime) spent by wave 0 of the block in the loop.
template <int KIND>
__global__ __launch_bounds__(256) void ubench_kernel(double* out, long long* cyc, int iters) {
  d4 acc[8];
  f4 accf[8];
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = d4{0, 0, 0, 0};
    accf[i] = f4{0, 0, 0, 0};
    v[i] = 1e-3 * i;
  }
  const double a = double(threadIdx.x) * 1e-3, b = double(blockIdx.x + 1) * 1e-3;
  const float af = float(a), bf = float(b);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if constexpr (KIND == 1) accf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, accf[i], 0, 0, 0);
      if constexpr (KIND == 2) v[i] = __builtin_fma(v[i], a, b);
      if constexpr (KIND == 3) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        v[i] = __builtin_fma(v[i], a, b);
        v[(i + 1) & 7] = __builtin_fma(v[(i + 1) & 7], a, b);
        v[(i + 2) & 7] = __builtin_fma(v[(i + 2) & 7], a, b);
        v[(i + 3) & 7] = __builtin_fma(v[(i + 3) & 7], a, b);
      }
      if constexpr (KIND == 4) {
        double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
        acc[i][0] = r;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + double(accf[i][0] + accf[i][3]) + v[i];
  if (s == -1.2345) out[0] = s;  // never true; keeps the chains live
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

}  // namespace

// Workgroups of a persistent 128x128-tile launch: two per CU fill the chip (234 VGPRs, 72 KiB of LDS
// each); `reserve` of those slots are left to the kernels of a panel chain that runs beside the
// update.  A multiple of 8 whenever tiles are left over, so a workgroup's tiles stay on one XCD.
static unsigned persistent_grid(const tgp_ctx* ctx, int nblk, int64_t reserve) {
  // an update that has the chip to itself is launched one workgroup per tile: the hardware's dynamic
  // assignment beats the static round-robin by 2-3 % (profiles/r02_r_persistent_reserve.txt)
  if (reserve <= 0) return unsigned(nblk);
  int64_t slots = 2 * int64_t(ctx->cus > 0 ? ctx->cus : 256) - reserve;
  slots = slots / 8 * 8;
  if (slots < 8) slots = 8;
  return unsigned(nblk <= slots ? nblk : slots);
}

template <typename T>
int launch_gemm_nt(tgp_ctx* ctx, hipStream_t st, int64_t m, int64_t n, int64_t k, const T* A,
                   int64_t lda, const T* B, int64_t ldb, T* C, int64_t ldc, int lower, int mode,
                   int role) {
  TGP_ARG_CHECK(m % BM == 0 && n % BN == 0 && k % BK == 0 && k > 0,
                "gemm_nt: m,n must be multiples of %d and k of %d (got %lld,%lld,%lld)", BM, BK,
                (long long)m, (long long)n, (long long)k);
  if (m == 0 || n == 0) return TGP_OK;
  if (ctx->trace) {  // v: A, B, C offsets, m, n, k, lower | role << 8, ld (all operands share it)
    // (bits 16..: tile columns of the PREFIX of a merged trailing update -- the hints are consumed here as in a real launch)
    const int64_t pfx = role == 0 ? ctx->prefix_hint_cols / BN : 0;
    ctx->prefix_hint_cols = 0;
    ctx->prefix_hint_counter = nullptr;
    if (role == 0) ctx->reserve_hint = 0;
    trace_push(ctx, 3, st, trace_off(ctx, A), trace_off(ctx, B), trace_off(ctx, C), m, n, k,
               int64_t(lower) | (int64_t(role) << 8) | (pfx << 16), ldc);
    return TGP_OK;
  }
  GemmArgs<T> g;
  g.skip00 = 0;
  g.dG = g.dr = g.dl0 = g.dnbt = 0;
  g.split_first = 0; g.split_s = 1; g.ngrid = 0; g.ws = nullptr; g.cnt = nullptr;
  g.batch = 1; g.sA = g.sB = g.sC = 0;
  g.band = (int)ctx->tile_band;
  g.prefix_tn = g.prefix_cnt = 0;
  g.prefix_done = nullptr;
  TGP_ARG_CHECK(role != 3 || (k <= 256 && mode == 0), "role 3 needs the small-tile path");
  // short-K updates, and (role 4) a latency-bound update with too few big tiles to fill the chip
  // role 5 (round 6): role 4 without the first 128 x 128 diagonal block -- the gate of the next panel while that block is
  // updated and factored on a side stream (potrf, chain_gate_split)
  if ((((role == 1 || role == 3) && k <= 256) || role == 4 || role == 5) && (mode == 0 || mode == 1)) {
    g.skip00 = (role == 3 || role == 5);
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.tm = int(m / SM); g.tn = int(n / SM);
    g.k = int(k); g.lower = lower; g.mode = mode;
    if (lower) {
      if (g.tn > g.tm) g.tn = g.tm;
      g.nblk = g.tn * g.tm - (g.tn * (g.tn - 1)) / 2;
    } else {
      g.nblk = g.tm * g.tn;
    }
    hipLaunchKernelGGL((gemm_nt_small_kernel<T>), dim3((unsigned)g.nblk), dim3(256), 0, st, g);
    TGP_HIP_TRY(hipGetLastError());
    return TGP_OK;
  }
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tm = int(m / BM); g.tn = int(n / BN);
  g.k = int(k); g.lower = lower; g.mode = mode;
  if (lower) {
    if (g.tn > g.tm) g.tn = g.tm;
    g.nblk = g.tn * g.tm - (g.tn * (g.tn - 1)) / 2;
  } else {
    g.nblk = g.tm * g.tn;
  }
  const int64_t reserve = role == 0 ? ctx->reserve_hint : 0;
  ctx->reserve_hint = 0;
  if (role == 0 && ctx->prefix_hint_cols > 0) {  // the merged trailing update: the next panel's block column first (see GemmArgs)
    TGP_ARG_CHECK(lower && mode == 0 && m == n && ctx->prefix_hint_cols % BN == 0 && ctx->prefix_hint_cols < n &&
                      k / BK >= 8 && ctx->prefix_hint_counter != nullptr,
                  "gemm_nt: a prefix needs a lower square update with at least 128 k-columns");
    g.prefix_tn = int(ctx->prefix_hint_cols / BN);
    g.prefix_cnt = tile_count(g.tm, g.prefix_tn, 1);
    g.prefix_done = ctx->prefix_hint_counter;
  }
  ctx->prefix_hint_cols = 0;
  ctx->prefix_hint_counter = nullptr;
  g.split_first = g.nblk; g.split_s = 1; g.ngrid = g.nblk; g.ws = nullptr; g.cnt = nullptr;
  // Split tail (option split_tail, fp64 trailing updates on the main stream that run one workgroup per tile): the
  // launch's last round of tiles fills R < slots workgroup slots for the duration of a full k-loop (250 us at
  // K = 1024) -- up to 24 % of a 3-round launch at N = 16 384.  Those R tiles are given S = 2, 4 or 8 workgroups
  // each, over 1 / S of the k-range, so that the last round is R S <= slots short workgroups.
  if (ctx->split_tail != 0 && role == 0 && mode == 0 && reserve <= 0 && sizeof(T) == 8 && st == ctx->stream && g.prefix_cnt == 0) {
    const int slots = 2 * (ctx->cus > 0 ? ctx->cus : 256);
    const int R = g.nblk % slots, nkt = g.k / BK;
    int S = 1;
    while (R > 0 && 2 * S * R <= slots && 2 * S <= 8 && nkt % (2 * S) == 0 && nkt / (2 * S) >= 8) S *= 2;
    if (S > 1) {
      const size_t need = size_t(slots) * BM * BN * sizeof(T) + size_t(slots) * sizeof(int);
      if (ctx->gemm_ws_bytes < need) {
        if (ctx->d_gemm_ws) TGP_HIP_TRY(hipFree(ctx->d_gemm_ws));
        ctx->d_gemm_ws = nullptr;
        ctx->gemm_ws_bytes = 0;
        TGP_HIP_TRY(hipMalloc(&ctx->d_gemm_ws, need));
        TGP_HIP_TRY(hipMemset(ctx->d_gemm_ws, 0, need));  // (the per-tile counters reset themselves afterwards)
        ctx->gemm_ws_bytes = need;
      }
      g.split_first = g.nblk - R;
      g.split_s = S;
      g.ngrid = g.split_first + R * S;
      g.ws = static_cast<T*>(ctx->d_gemm_ws);
      g.cnt = reinterpret_cast<int*>(static_cast<char*>(ctx->d_gemm_ws) + size_t(slots) * BM * BN * sizeof(T));
      hipLaunchKernelGGL((gemm_nt_kernel<T, 2>), dim3(unsigned(g.ngrid)), dim3(256), 0, st, g);
      TGP_HIP_TRY(hipGetLastError());
      return TGP_OK;
    }
  }
  // (Measured and removed, profiles/r02_r: the last, partly filled round of tiles on the 64x64-tile
  // kernel -- four workgroups per tile behind the full rounds.  Its k-loop of 64 short k-tiles takes
  // as long as the 245-us round it replaces: 4.43 vs 4.40 ms on a 16384^2 lower update.)
  // (Measured and removed, profiles/r03_j: the 8 x 8 patch order of ROLE 3 on the trailing update -- FETCH_SIZE per
  // launch -32 %, the factorisation 4 % SLOWER at N = 16 384 and 1 % at 65 536: half-empty diagonal patches unbalance
  // the XCDs; and the run-time flag alone cost the fp32 kernel 1.3 %.)
  const unsigned grid = persistent_grid(ctx, g.nblk, reserve);
  if (role == 0)
    hipLaunchKernelGGL((gemm_nt_kernel<T, 0>), dim3(grid), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<T, 1>), dim3(grid), dim3(256), 0, st, g);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// `batch` independent products C_q = A_q B_q^T (operand q at base + q * s{A,B,C}) with a triangular A, on the
// 128 x 128-tile kernel in its patch order (ROLE 3):
//   mode bit2: A upper triangular (k from the row tile on), bit3: A lower triangular (k up to the row tile),
//   bit4: k walked from the end; always C = A B^T (bit0).  lower: only tiles ti >= tj (m == n).
template <typename T>
int launch_gemm_tri(tgp_ctx* ctx, hipStream_t st, int64_t m, int64_t n, int64_t k, const T* A, int64_t lda,
                    const T* B, int64_t ldb, T* C, int64_t ldc, int lower, int mode, int batch, int64_t sA,
                    int64_t sB, int64_t sC) {
  TGP_ARG_CHECK(m % BM == 0 && n % BN == 0 && k % BK == 0 && k > 0 && batch >= 1 && (mode & 1) && !(mode & 2) &&
                    (!lower || m == n),
                "gemm_tri: bad shape or mode (m, n, k = %lld, %lld, %lld)", (long long)m, (long long)n, (long long)k);
  TGP_ARG_CHECK(!ctx->trace, "gemm_tri is not part of the traced schedule");
  if (m == 0 || n == 0) return TGP_OK;
  GemmArgs<T> g;
  g.skip00 = 0;
  g.dG = g.dr = g.dl0 = g.dnbt = 0;
  g.split_first = 0; g.split_s = 1; g.ngrid = 0; g.ws = nullptr; g.cnt = nullptr;
  g.A = A; g.B = B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.tm = int(m / BM); g.tn = int(n / BN);
  g.k = int(k); g.lower = lower; g.mode = mode;
  g.batch = batch; g.sA = sA; g.sB = sB; g.sC = sC;
  g.band = 0;
  const int64_t ptm = (g.tm + 7) / 8, ptn = (g.tn + 7) / 8;
  const int64_t patches = int64_t(batch) * (lower ? ptm * (ptm + 1) / 2 : ptm * ptn);
  const int64_t grid = (patches + 7) / 8 * 8 * 64;
  TGP_ARG_CHECK(grid < (int64_t(1) << 31), "gemm_tri: too many tiles");
  g.nblk = int(grid);
  hipLaunchKernelGGL((gemm_nt_kernel<T, 3>), dim3(unsigned(grid)), dim3(256), 0, st, g);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

// Trailing update of a rank's block columns l >= l0 (nloc of them, nb wide) by a panel:
//   C_loc[i, c] -= sum_k P[i, k] P[col(c), k]   for i >= col(c),
// with P addressed by GLOBAL row (caller passes the panel pointer minus its first row) and
// col(c) the global column of local column c under the cyclic map j = l * G + rank.
template <typename T>
int launch_gemm_nt_dist(tgp_ctx* ctx, hipStream_t st, int64_t n_rows, int64_t nb, int64_t k,
                        const T* P, int64_t ldp, T* Cloc, int64_t ldc, int G, int rank, int64_t l0,
                        int64_t nloc, int64_t reserve) {
  TGP_ARG_CHECK(n_rows % BM == 0 && nb % BN == 0 && k % BK == 0 && k > 0 && G >= 1 && rank >= 0 &&
                    rank < G && l0 >= 0,
                "gemm_nt_dist: bad shape");
  if (nloc <= 0) return TGP_OK;
  GemmArgs<T> g;
  g.skip00 = 0;
  g.A = P; g.B = P; g.C = Cloc + l0 * nb * ldc;
  g.lda = ldp; g.ldb = ldp; g.ldc = ldc;
  g.tm = int(n_rows / BM); g.tn = int(nloc * (nb / BN));
  g.k = int(k); g.lower = 1; g.mode = 0;
  g.dG = G; g.dr = rank; g.dl0 = int(l0); g.dnbt = int(nb / BN);
  g.split_first = 0; g.split_s = 1; g.ngrid = 0; g.ws = nullptr; g.cnt = nullptr;
  g.batch = 1; g.sA = g.sB = g.sC = 0;
  g.band = (int)ctx->tile_band;
  int64_t total = 0;
  for (int64_t q = 0; q < nloc; ++q) {
    const int64_t g0 = ((l0 + q) * G + rank) * g.dnbt;
    TGP_ARG_CHECK(g0 + g.dnbt <= g.tm, "gemm_nt_dist: block column outside the matrix");
    total += int64_t(g.dnbt) * (g.tm - g0) - int64_t(g.dnbt) * (g.dnbt - 1) / 2;
  }
  TGP_ARG_CHECK(total < (int64_t(1) << 31), "gemm_nt_dist: too many tiles");
  g.nblk = int(total);
  // `reserve`: workgroup slots left to a panel chain of this rank that runs beside the update (ctx chain_reserve)
  // -- as in the single-GPU driver only when the update is the shorter of the two (a long update on 3/4 of the
  // slots loses more than the chain gains: N = 65 536 at world size 1 was 27 % slower with the reserve always on)
  if (g.nblk > ctx->reserve_max_tiles) reserve = 0;
  hipLaunchKernelGGL((gemm_nt_kernel<T, 0>), dim3(persistent_grid(ctx, g.nblk, reserve)), dim3(256), 0, st, g);
  TGP_HIP_TRY(hipGetLastError());
  return TGP_OK;
}

int ubench(tgp_ctx* ctx, int kind, int blocks_per_cu, double* tflops, double* cycles_per_op) {
  TGP_ARG_CHECK(kind >= 0 && kind <= 4 && blocks_per_cu >= 1 && blocks_per_cu <= 8, "ubench: bad kind");
  const int iters = 2048, blocks = ctx->cus * blocks_per_cu;
  TGP_TRY(ensure_work(ctx, size_t(blocks) * sizeof(long long)));
  long long* cyc = static_cast<long long*>(ctx->d_work);
  hipEvent_t e0, e1;
  TGP_HIP_TRY(hipEventCreate(&e0));
  TGP_HIP_TRY(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    TGP_HIP_TRY(hipEventRecord(e0, ctx->stream));
#define TGP_UB(K) hipLaunchKernelGGL((ubench_kernel<K>), dim3(blocks), dim3(256), 0, ctx->stream, \
                                     ctx->d_scal, cyc, iters)
    switch (kind) {
      case 0: TGP_UB(0); break;
      case 1: TGP_UB(1); break;
      case 2: TGP_UB(2); break;
      case 3: TGP_UB(3); break;
      default: TGP_UB(4); break;
    }
#undef TGP_UB
    TGP_HIP_TRY(hipEventRecord(e1, ctx->stream));
    TGP_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    TGP_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  long long c0 = 0;
  TGP_HIP_TRY(hipMemcpy(&c0, cyc, sizeof(long long), hipMemcpyDeviceToHost));
  // flops per wave-iteration of the unrolled-by-8 body
  const double per_op[5] = {2048.0, 2048.0, 128.0, 2048.0 + 4 * 128.0, 512.0};
  const double flops = double(blocks) * 4.0 /*waves*/ * iters * 8.0 * per_op[kind];
  *tflops = flops / (double(best) * 1e-3) / 1e12;
  if (cycles_per_op) *cycles_per_op = double(c0) / (double(iters) * 8.0);
  return TGP_OK;
}

int ubench_mfma(tgp_ctx* ctx, int dtype, double* tflops) {
  return ubench(ctx, dtype == TGP_F64 ? 0 : 1, 4, tflops, nullptr);
}

#define TGP_INST(T)                                                                              \
  template int launch_gemm_nt_dist<T>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const T*, \
                                      int64_t, T*, int64_t, int, int, int64_t, int64_t, int64_t); \
  template int launch_gemm_nt<T>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const T*,     \
                                 int64_t, const T*, int64_t, T*, int64_t, int, int, int);         \
  template int launch_gemm_tri<T>(tgp_ctx*, hipStream_t, int64_t, int64_t, int64_t, const T*, int64_t, const T*, \
                                  int64_t, T*, int64_t, int, int, int, int64_t, int64_t, int64_t);
TGP_INST(float)
TGP_INST(double)
#undef TGP_INST

}  // namespace tgp
