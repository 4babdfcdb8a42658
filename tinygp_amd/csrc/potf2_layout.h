// potf2_layout.h -- the LDS image of potf2's 128 x 128 tile; shared by chol.hip (device code) and
// tests/potf2_lds_check.cpp (host replay of potf2_body.inc).  TGP_HD is defined by the includer.
#pragma once
// Only the 36 lower 16x16 blocks, each contiguous and column-major (element (r, c) of block (i, j) at
// blk(i, j) + c * 16 + r).  72 KiB in fp64: small enough to share a CU with one 74 KiB GEMM workgroup
// during look-ahead, and a 32-lane operand read (16 rows x 2 k) is 256 contiguous bytes:
// conflict-free without padding.
TGP_HD constexpr int blk(int i, int j) { return (i * (i + 1) / 2 + j) * 256; }
