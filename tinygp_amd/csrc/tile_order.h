// tile_order.h -- linear workgroup id -> output tile of the MFMA products, shared by the kernels (gemm.hip) and the C ABI's
// test hook (tgp_tile_order: tests/test_host_logic.py enumerates a launch on the CPU and checks that the map is a bijection).
// TILE_HD is defined by the includer (__host__ __device__ in HIP code, empty in a host-only translation unit).
//
// band == 0: column by column (tj major; consecutive ids share the B operand's tile; `lower`: column tj holds rows tj .. tm-1).
// band  > 0 (round 5, ctx option "tile_band", default 8): BANDS of `band` tile rows, top down; inside a band column by column.
//   The 64 workgroups an XCD has resident then cover a patch of ~band rows x 64/band columns -- band + 64/band operand panels
//   instead of 64 + 1 -- with the same contiguous, equal-length run of ids per XCD as before (no half-empty diagonal
//   patches: the 8 x 8 patch order of profiles/r03_j cut the fabric traffic by a third and lost 4 % to that imbalance).
#pragma once

TILE_HD inline void tile_decode_columns(int b, int tm, int tn, int lower, int& ti, int& tj) {
  if (!lower) {
    tj = b / tm;
    ti = b - tj * tm;
    return;
  }
  // offset(tj) = tj*tm - tj*(tj-1)/2
  const float fm = 2.0f * float(tm) + 1.0f;
  float disc = fm * fm - 8.0f * float(b);
  if (disc < 0.0f) disc = 0.0f;
  int t = int((fm - sqrtf(disc)) * 0.5f);
  if (t < 0) t = 0;
  if (t > tn - 1) t = tn - 1;
  while (t > 0 && t * tm - (t * (t - 1)) / 2 > b) --t;
  while (t + 1 < tn && (t + 1) * tm - ((t + 1) * t) / 2 <= b) ++t;
  tj = t;
  ti = tj + (b - (t * tm - (t * (t - 1)) / 2));
}

TILE_HD inline void tile_decode(int b, int tm, int tn, int lower, int band, int& ti, int& tj) {
  if (band <= 0) {
    tile_decode_columns(b, tm, tn, lower, ti, tj);
    return;
  }
  for (int r0 = 0; r0 < tm; r0 += band) {
    const int r1 = r0 + band < tm ? r0 + band : tm, h = r1 - r0;
    int cfull, ctri;
    if (!lower) {
      cfull = tn;
      ctri = 0;
    } else {
      cfull = tn < r0 ? tn : r0;                    // columns left of the band's rows: full height
      ctri = (tn < r1 ? tn : r1) - cfull;           // columns tj = r0 + s that cross the diagonal: rows tj .. r1-1
    }
    const int cnt = cfull * h + ctri * h - (ctri * (ctri - 1)) / 2;
    if (b >= cnt) {
      b -= cnt;
      continue;
    }
    if (b < cfull * h) {
      tj = b / h;
      ti = r0 + (b - tj * h);
      return;
    }
    b -= cfull * h;
    int s = 0;
    while (s + 1 < ctri && (s + 1) * h - ((s + 1) * s) / 2 <= b) ++s;
    tj = cfull + s;
    ti = tj + (b - (s * h - (s * (s - 1)) / 2));
    return;
  }
  ti = tm - 1;  // (unreachable for b < number of tiles)
  tj = 0;
}

TILE_HD inline int tile_count(int tm, int tn, int lower) {
  if (!lower) return tm * tn;
  const int c = tn < tm ? tn : tm;
  return c * tm - (c * (c - 1)) / 2;
}
