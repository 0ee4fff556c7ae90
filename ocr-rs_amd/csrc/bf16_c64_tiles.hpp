// Shared by conv3x3_bf16_c64.hip and basic_block_bf16_c64.hip: the LDS image layout of the bf16 3x3 64 -> 64 kernels and the
// software-pipelined (tile, tap) sequence that reads it.
#pragma once

#include "common.hpp"

namespace ocr {
namespace bf16_c64 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// compile-time ablations (tools/build_abl_bb.sh; results are wrong, only the time matters): 1 no fragment reads, 2 no MFMAs, 4 no patch
// DMA, 8 no stores of the intermediate, 16 no conv2 epilogue, 32 conv1 waves idle, 64 conv2 waves idle
#ifndef BB_ABL
#define BB_ABL 0
#endif

// LDS images (patch of x, intermediate): one 128-byte row per pixel, pixel (row, col) in LDS row row * PITCH + col (PITCH even), its
// 16-byte chunk c in slot c ^ sw, sw = ((col >> 1) & 7) ^ 4 (row & 1).  A ds_read_b128 is served in the lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} (+ 32), one LDS cycle per group when its sixteen lanes touch sixteen different 16-byte bank slots.
// An MFMA row tile is 2 image rows x 16 columns: lane l31 < 16 -> (row, l31), lane l31 >= 16 -> (row + 1, (l31 & 15) ^ 8): with the
// second row's column halves swapped the sixteen lanes of a group see sixteen different (col & 15, row & 1) for every tap - 4 LDS
// cycles per read (conv3x3_bf16_c64.hip's first layout, swizzled by linear pixel index with a plain second row, was 2-way on every
// read).  conv1's sixth tile - the two columns of the intermediate left of the five 2 x 16 tiles, ten rows of them - reads one
// patch column over many rows and is 4-way (72 of a block's 720 fragment reads; the LDS is a third booked).
// The swizzle is an XOR, so a fragment address is (pixel row + immediate tap offset) + (xb(tx, row parity) ^ 32 s): six lane
// offsets xb per wave and block, one v_xor per read (the first layout: three instructions per read - and a wave issues one
// instruction per four cycles, which is what this kernel is bound by: profiles/r06_bf16_block_ablations.txt).
__device__ __forceinline__ int tile_col8(int i) { return i < 16 ? i : ((i & 15) ^ 8); }

// the six in-row offsets of a lane whose pixel is (row parity par, column col) for tap (0, 0): [tx][parity of ty] -> ((half ^ sw) << 4)
__device__ __forceinline__ void xor_bases(int col, int par, int half, int (&xb)[3][2]) {
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int x = (half ^ ((col + tx) >> 1) ^ (par << 2)) & 7;
    xb[tx][0] = x << 4;
    xb[tx][1] = (x ^ 4) << 4;
  }
}

// NT row tiles, one after the other, nine taps x four k-steps each: A = shifted 16-byte reads of an image (pixb[j] = byte offset of the
// LDS row of the lane's pixel of tile j for tap (0, 0); xbase(j, tx, p) = the lane's in-row offset), B = the wave's
// register-resident weights.  The (tile, tap) steps form ONE software-pipelined sequence: the four fragment reads of step n + 1 are
// issued before the four MFMAs of step n.  One accumulator (16 registers) at a time: epi(j, acc) takes tile j's sums when its
// ninth tap is done.
template <int NT, int PITCH, typename XBase, typename Epi>
__device__ __forceinline__ void conv_tiles(const unsigned char* img, const int (&pixb)[NT], XBase&& xbase, const bf16x8 (&wreg)[9][4], Epi&& epi) {
  bf16x8 a[2][4];
  int R[3][2];
  auto fetch = [&](int step, bf16x8(&dst)[4]) {
    const int j = step / 9, t = step % 9;
    if (t == 0) {
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int q = 0; q < 2; ++q) R[tx][q] = pixb[j] + xbase(j, tx, q);
    }
    const int r = R[t % 3][(t / 3) & 1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (BB_ABL & 1) dst[s] = wreg[(step + 1) % 9][s];
      else dst[s] = *reinterpret_cast<const bf16x8*>(img + (r ^ (s << 5)) + ((t / 3) * PITCH + (t % 3)) * 128);
    }
  };
  fetch(0, a[0]);
  f32x16 acc;
#pragma unroll
  for (int step = 0; step < 9 * NT; ++step) {
    if (step + 1 < 9 * NT) fetch(step + 1, a[(step + 1) & 1]);
    if (step % 9 == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (BB_ABL & 2) acc[s] += __builtin_bit_cast(f32x4, a[step & 1][s])[0];
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[step & 1][s], wreg[step % 9][s], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (step % 9 == 8) epi(step / 9, acc);
  }
}

}  // namespace bf16_c64
}  // namespace ocr
