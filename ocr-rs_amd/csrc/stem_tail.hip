// Memory-bound ends of the detection graph (HBM roofline, not MFMA):
//   stem : conv1 7x7 s2 p3 (1->64) + bn1 + ReLU + max_pool2d(3,2,1), one pass
//          /root/reference/src/text_detection/model.rs:108-112
//   tail : bin_conv_tr2 convT 2x2 s2 (64->1) + bias + sigmoid, optional fused
//          binarize(pred, thresh)      model.rs:149-150, metrics.rs:129-131
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TPH = 4, TPW = 6;                 // pooled output tile per workgroup
constexpr int CR = 2 * TPH + 1, CC = 2 * TPW + 1;  // conv rows/cols feeding it (3x3 s2 window): 9 x 13
constexpr int NPIX = CR * CC;                   // 117 conv pixels per workgroup
constexpr int MTILES = (NPIX + 31) / 32;        // as 4 MFMA row tiles of 32 pixels: one per wave
constexpr int IR = 2 * (CR - 1) + 7;            // input rows feeding the conv tile (7x7 s2): 23
constexpr int ICP = 2 * (CC - 1) + 7 + 1;       // input columns (31), padded to 32
static_assert(MTILES == 4, "one pixel tile per wave");
// LDS image of the input tile: even and odd columns in separate planes, so that the stride-2 walk of the
// conv pixels along x becomes a unit-stride walk (lane i -> word i: no bank conflicts on the A operand);
// 22 words per plane row keeps the 2-3 pixel rows that a 32-pixel MFMA tile spans on different banks.
constexpr int PW = 22;
static_assert(2 * PW >= ICP, "plane row holds half of the tile's columns");
constexpr int PLANE = IR * PW;
// K order: MFMA step s = 4 kh + jp multiplies taps (kh, 2 jp) on lanes 0-31 and (kh, 2 jp + 1) on lanes 32-63;
// in the plane layout those two samples are exactly PLANE words apart for every step, so one per-lane base
// address plus a compile-time immediate serves all steps.  kw = 7 is a zero-weight pad: 28 steps for 49 taps.
constexpr int KSTEPS = 28;
constexpr int TL = 4;                           // tiles per workgroup along x                      // K = 49 taps padded to 50 = 25 MFMA steps of 2

typedef float f32x16 __attribute__((ext_vector_type(16)));

// conv1 as a GEMM on the matrix cores: rows = conv pixels of the tile, columns = 64 channels,
// K = 49 taps.  The A operand is read straight from the LDS input tile - lane l supplies pixel
// (l & 31) and tap 2s + (l >> 5), one ds_read_b32 per MFMA whose address is a per-lane base plus
// a compile-time tap offset - and the 49x64 weights sit in registers (50 per lane).
// grid (Wp/(TPW*TL), Hp/TPH, N), 256 threads = 4 waves: wave w owns pixel tile w for both channel halves (one
// A read feeds two MFMAs, every wave does the same amount of work).  BN + ReLU on the accumulators, conv tile
// to LDS, then the 3x3 s2 max pool.
constexpr int NT = 64 * MTILES;
// TX: element type of the frames - float, or uint8_t (the reference's image IS u8, image_ops.rs:350-364: raw 0..255 luma
// converted to f32 without scaling; the cast here is that conversion, exact)
template <typename TO, typename TX>
__global__ __launch_bounds__(NT) void stem_kernel(const TX* __restrict__ x, const float* __restrict__ w49x64,
                                                   const float* __restrict__ scale, const float* __restrict__ bias,
                                                   TO* __restrict__ out, int H, int W) {
  __shared__ __attribute__((aligned(16))) float in_s[2 * PLANE];
  __shared__ __attribute__((aligned(16))) float conv_s[32 * MTILES][64];  // padded to whole MFMA tiles: stores need no guard
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.z;
  const int ph0 = blockIdx.y * TPH;
  const int Hc = H >> 1, Wc = W >> 1, Hp = H >> 2, Wp = W >> 2;
  const int cr0 = 2 * ph0 - 1;
  const int ir0 = 2 * cr0 - 3;
  const TX* xin = x + (size_t)n * H * W;
  const int half = lane >> 5, l31 = lane & 31;
  // B operand: weight of tap 2s + half for channel 32 ct + (lane & 31); the padded tap 49 is zero
  float wreg[2][KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int kh = s >> 2, kw = 2 * (s & 3) + half;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) wreg[ct][s] = kw < 7 ? w49x64[(kh * 7 + kw) * 64 + 32 * ct + l31] : 0.f;
  }
  const float sc0 = scale[l31], bi0 = bias[l31], sc1 = scale[32 + l31], bi1 = bias[32 + l31];
  // a workgroup walks TL tiles along x with the weights resident in registers
  for (int tl = 0; tl < TL; ++tl) {
  const int pw0 = (blockIdx.x * TL + tl) * TPW;
  if (pw0 >= Wp) break;
  const int cc0 = 2 * pw0 - 1;
  const int ic0 = 2 * cc0 - 3;
  for (int i = tid; i < IR * ICP; i += NT) {
    const int rr = i / ICP, cc = i - rr * ICP;
    const int ih = ir0 + rr, iw = ic0 + cc;
    float v = 0.f;  // zero padding of conv1 (and the pad column)
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = (float)xin[(size_t)ih * W + iw];
    in_s[(cc & 1) * PLANE + rr * PW + (cc >> 1)] = v;
  }
  __syncthreads();  // also: every wave has left the previous tile's pool phase (conv_s is free)

  {
    const int mt = wv;
    const int pix = min(32 * mt + l31, NPIX - 1);  // rows past the tile repeat its last pixel (never stored)
    const int pr = pix / CC, pc = pix - pr * CC;
    const int base = 2 * pr * PW + pc + half * PLANE;  // top-left sample of this pixel's window, this lane's plane
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
#ifndef STEM_NO_MFMA
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const float a = in_s[base + (s >> 2) * PW + (s & 3)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[0][s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wreg[1][s], acc1, 0, 0, 0);
    }
#else
    acc0[0] = in_s[base] * wreg[0][0];
    acc1[0] = in_s[base] * wreg[1][0];
#endif
    // C/D map: column (channel) = lane & 31, row (pixel) = (e&3) + 8 (e>>2) + 4 (lane>>5)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int p = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * half;
      {
        const int qr = p / CC, qc = p - qr * CC;
        // conv positions outside the conv grid are max-pool padding: they never win against the
        // always-valid window centre because ReLU output is >= 0.
        const bool valid = (unsigned)(cr0 + qr) < (unsigned)Hc && (unsigned)(cc0 + qc) < (unsigned)Wc;
        conv_s[p][l31] = valid ? fmaxf(acc0[e] * sc0 + bi0, 0.f) : 0.f;
        conv_s[p][32 + l31] = valid ? fmaxf(acc1[e] * sc1 + bi1, 0.f) : 0.f;
      }
    }
  }
  __syncthreads();
  // 3x3 s2 max pool: one thread = one pooled pixel x 4 channels (b128 LDS reads, 16-byte stores)
#ifdef STEM_NO_POOL
  if (H < 0)
#endif
  for (int o = tid; o < TPH * TPW * 16; o += NT) {
    const int c4 = (o & 15) * 4, pp = o >> 4;
    const int py = pp / TPW, px = pp - py * TPW;
    const int ph = ph0 + py, pw = pw0 + px;
    if (ph < Hp && pw < Wp) {
      f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&conv_s[(2 * py + dy) * CC + 2 * px + dx][c4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
      TO* dst = out + (((size_t)n * Hp + ph) * Wp + pw) * 64 + c4;
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<f32x4*>(dst) = m;
      } else {
        typedef TO to4 __attribute__((ext_vector_type(4)));
        to4 h = {(TO)m[0], (TO)m[1], (TO)m[2], (TO)m[3]};
        *reinterpret_cast<to4*>(dst) = h;
      }
    }
  }
  }  // tl
}

// The same stem with its convolution on the bf16 matrix cores (OCR_PRECISION_BF16; f32 accumulate, f32 batch norm,
// bf16 output).  K = 8 kh + kw over an 8 x 8 window whose last row and column carry zero weights: one
// v_mfma_f32_32x32x16_bf16 step covers window rows 2 s and 2 s + 1, a lane's eight operands are eight consecutive
// pixels of ONE input row - four 4-byte LDS reads from a plain row-major bf16 image of the input tile (raw 0..255
// luma is exact in bf16).  8 MFMAs per wave and tile instead of 56: the kernel is bound by its staging / pooling
// phases and by HBM, not by the matrix cores.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Tile shapes of the bf16 / split-bf16 stem: TPH x TPW pooled pixels per workgroup step <- (2 TPH + 1) x (2 TPW + 1) conv pixels as
// MTILES MFMA row tiles of 32 (MTILES / 4 per wave) <- IR x ICP input pixels kept as bf16 images of PWB pixels per row (+ one row:
// window row 7 of the last conv row - zero weight, must be finite).
//   4 x 6 (round 2): 117 of 128 MFMA rows used, 24 pooled pixels per pair of barriers
//   8 x 7 (round 5): 255 of 256 rows, 56 pooled pixels per pair of barriers, 1.91 x the staging for 2.33 x the output; 74 KB of LDS,
//                    still two workgroups per CU
template <int TPH_, int TPW_, int PWB_>
struct StemTile {
  static constexpr int TPH = TPH_, TPW = TPW_;
  static constexpr int CR = 2 * TPH + 1, CC = 2 * TPW + 1, NPIX = CR * CC, MTILES = (NPIX + 31) / 32;
  static constexpr int IR = 2 * (CR - 1) + 7, ICP = 2 * (CC - 1) + 7 + 1, IRB = IR + 1, PWB = PWB_;
  static_assert(MTILES % 4 == 0, "whole row tiles per wave");
  static_assert(ICP % 2 == 0 && PWB % 2 == 0 && PWB >= ICP && PWB / 2 >= CC - 1 + 4, "a window's four words lie inside the image row");
};
using StemTileS = StemTile<4, 6, 36>;
using StemTileL = StemTile<8, 7, 40>;
// X3: the f32 stem on the bf16 matrix cores - the input tile as three bf16 images (pixel = hi + mid + lo exactly), the
// weights as three fragment sets, six partial products per window (mid.lo, lo.mid, lo.lo are below 2^-23 of a product),
// f32 accumulate, f32 output: 48 MFMAs of 32 cycles per wave and tile where the exact-f32 kernel above issues 56 of 64.
#ifdef STEM_STAMPS
// diagnostic build only (make EXTRA=-DSTEM_STAMPS): s_memtime of one workgroup in the middle of the grid, waves 0 and 3, at the phase
// boundaries of its four tiles (tools/stem_stamps.py)
__device__ long long g_stem_stamps[2 * 4 * 8];
#define STEM_STAMP(k)                                                                                                  \
  do {                                                                                                                 \
    if (blockIdx.x == 3 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 5 && (wv == 0 || wv == 3) && lane == 0)                   \
      g_stem_stamps[((wv ? 1 : 0) * 4 + tl) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime();                      \
  } while (0)
#else
#define STEM_STAMP(k) do {} while (0)
#endif
template <bool X3, typename TX, typename TO, typename TC>
__global__ __launch_bounds__(NT) void stem_bf16_kernel(const TX* __restrict__ x, const u32x4* __restrict__ wfrag,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       TO* __restrict__ out, int H, int W) {
  static_assert(std::is_same<TO, std::conditional_t<X3, float, __bf16>>::value, "X3 writes f32, the bf16 precision bf16");
  constexpr int TPH = TC::TPH, TPW = TC::TPW, CR = TC::CR, CC = TC::CC, NPIX = TC::NPIX, MTILES = TC::MTILES;
  constexpr int IR = TC::IR, ICP = TC::ICP, IRB = TC::IRB, PWB = TC::PWB;
  constexpr int NPL = X3 ? 3 : 1;                       // bf16 images of the input tile / weight fragment sets
  constexpr int IMG = IRB * PWB / 2;                    // words per image (two bf16 per word)
  __shared__ __attribute__((aligned(16))) unsigned in_s[NPL * IMG];
  __shared__ __attribute__((aligned(16))) float conv_s[32 * MTILES][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.z;
  const int ph0 = blockIdx.y * TPH;
  const int Hc = H >> 1, Wc = W >> 1, Hp = H >> 2, Wp = W >> 2;
  const int cr0 = 2 * ph0 - 1;
  const int ir0 = 2 * cr0 - 3;
  const TX* xin = x + (size_t)n * H * W;
  const int half = lane >> 5, l31 = lane & 31;
  // B operand of step s, channel tile ct: weights of window row 2 s + half, columns 0..7, channel 32 ct + (lane & 31)
  bf16x8 wreg[NPL][2][4];
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int s = 0; s < 4; ++s) wreg[pl][ct][s] = __builtin_bit_cast(bf16x8, wfrag[((pl * 2 + ct) * 4 + s) * 64 + lane]);
  const float sc0 = scale[l31], bi0 = bias[l31], sc1 = scale[32 + l31], bi1 = bias[32 + l31];
  for (int i = tid; i < NPL * (PWB / 2); i += NT) in_s[(i / (PWB / 2)) * IMG + (IRB - 1) * (PWB / 2) + i % (PWB / 2)] = 0u;  // the extra row
  // the input pixels of a tile are requested one tile ahead (two pixel pairs per thread, in registers): their HBM latency
  // runs under the previous tile's MFMAs and pooling instead of in front of every tile
  constexpr int NLD = (IR * (ICP / 2) + NT - 1) / NT;
  float pre[NLD][2];
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<TX*>(xin), 0, (unsigned)(H * W * (int)sizeof(TX)), 0x00020000);
  auto request = [&](int tl) {
    const int pw0 = (blockIdx.x * TL + tl) * TPW;
    const int ic0 = 2 * (2 * pw0 - 1) - 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = tid + k * NT;
      const int rr = i / (ICP / 2), cp = i - rr * (ICP / 2);
      const int ih = ir0 + rr, iw = ic0 + 2 * cp;
      // zero padding of conv1 = an out-of-range offset into this frame's buffer descriptor (no branch per pixel)
      const bool rowok = i < IR * (ICP / 2) && pw0 < Wp && (unsigned)ih < (unsigned)H;
      const unsigned o0 = rowok && (unsigned)iw < (unsigned)W ? (unsigned)((ih * W + iw) * (int)sizeof(TX)) : 0x80000000u;
      const unsigned o1 = rowok && (unsigned)(iw + 1) < (unsigned)W ? (unsigned)((ih * W + iw + 1) * (int)sizeof(TX)) : 0x80000000u;
      if constexpr (sizeof(TX) == 4) {
        pre[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o0, 0, 0));
        pre[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, o1, 0, 0));
      } else {
        pre[k][0] = (float)__builtin_amdgcn_raw_buffer_load_b8(x_rsrc, o0, 0, 0);
        pre[k][1] = (float)__builtin_amdgcn_raw_buffer_load_b8(x_rsrc, o1, 0, 0);
      }
    }
  };
  request(0);
  bool inexact = false;
  for (int tl = 0; tl < TL; ++tl) {
    const int pw0 = (blockIdx.x * TL + tl) * TPW;
    if (pw0 >= Wp) break;
    const int cc0 = 2 * pw0 - 1;
    STEM_STAMP(0);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {   // one word = two neighbouring pixels
      const int i = tid + k * NT;
      if (i >= IR * (ICP / 2)) break;
      const int rr = i / (ICP / 2), cp = i - rr * (ICP / 2);
      const float v0 = pre[k][0], v1 = pre[k][1];
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bf16x2 pk = {(__bf16)v0, (__bf16)v1};
      in_s[rr * (PWB / 2) + cp] = __builtin_bit_cast(unsigned, pk);
      if constexpr (X3 && sizeof(TX) == 4) {   // remainders are exact in f32; round to nearest even at every level
        const float r0 = v0 - (float)pk[0], r1 = v1 - (float)pk[1];
        const bf16x2 pm = {(__bf16)r0, (__bf16)r1};
        const bf16x2 pl = {(__bf16)(r0 - (float)pm[0]), (__bf16)(r1 - (float)pm[1])};
        in_s[IMG + rr * (PWB / 2) + cp] = __builtin_bit_cast(unsigned, pm);
        in_s[2 * IMG + rr * (PWB / 2) + cp] = __builtin_bit_cast(unsigned, pl);
        inexact |= (r0 != 0.f) || (r1 != 0.f);
      }
    }
    if (tl + 1 < TL) request(tl + 1);
    STEM_STAMP(1);
    // Raw luma (integers 0 .. 255, what the reference feeds: image_ops.rs:350-364) IS its bf16 value: the mid and lo terms of every
    // pixel of the tile are zero and the three products they enter are exactly zero - they are skipped, the sum is bit for bit
    // the same.  u8 frames: statically so; f32 frames: decided per tile with the barrier that publishes it.
    const bool six = X3 && sizeof(TX) == 4 && __builtin_amdgcn_readfirstlane(__syncthreads_or(inexact)) != 0;
    if constexpr (!(X3 && sizeof(TX) == 4)) __syncthreads();
    inexact = false;
    STEM_STAMP(2);
#pragma unroll 1   // (unrolled, the two tiles' accumulators and fragments take the split-bf16 form past 256 registers: one workgroup per CU)
    for (int u = 0; u < MTILES / 4; ++u) {   // this wave's row tiles
      const int mt = wv + 4 * u;
      const int pix = min(32 * mt + l31, NPIX - 1);
      const int pr = pix / CC, pc = pix - pr * CC;
      const unsigned* row = in_s + (2 * pr + half) * (PWB / 2) + pc;  // window row `half`, first pixel pair
      f32x16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc0[e] = acc1[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          u32x4 a;
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] = row[pl * IMG + 2 * s * (PWB / 2) + k];
          af[pl] = __builtin_bit_cast(bf16x8, a);
        }
        if constexpr (X3) {   // small terms first: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi
          constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            if (PA[q] != 0 && !six) continue;   // a zero operand plane (workgroup-uniform)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[q]], wreg[PB[q]][0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[q]], wreg[PB[q]][1][s], acc1, 0, 0, 0);
          }
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wreg[0][0][s], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wreg[0][1][s], acc1, 0, 0, 0);
        }
      }
      STEM_STAMP(3);
      // a tile whose 9 x 13 conv pixels all lie inside the conv map (every tile but the first row / column of tiles and a
      // ragged last column) needs no validity test: per element two FMA, two max, one paired LDS write - the general
      // form's per-element division, compares and selects were a fifth of this kernel's instructions
      if (cr0 >= 0 && cr0 + CR <= Hc && cc0 >= 0 && cc0 + CC <= Wc) {   // workgroup-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int p = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * half;
          conv_s[p][l31] = fmaxf(acc0[e] * sc0 + bi0, 0.f);
          conv_s[p][32 + l31] = fmaxf(acc1[e] * sc1 + bi1, 0.f);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int p = 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * half;
          const int qr = p / CC, qc = p - qr * CC;
          const bool valid = (unsigned)(cr0 + qr) < (unsigned)Hc && (unsigned)(cc0 + qc) < (unsigned)Wc;
          conv_s[p][l31] = valid ? fmaxf(acc0[e] * sc0 + bi0, 0.f) : 0.f;
          conv_s[p][32 + l31] = valid ? fmaxf(acc1[e] * sc1 + bi1, 0.f) : 0.f;
        }
      }
    }
    STEM_STAMP(4);
    __syncthreads();
    STEM_STAMP(5);
    for (int o = tid; o < TPH * TPW * 16; o += NT) {
      const int c4 = (o & 15) * 4, pp = o >> 4;
      const int py = pp / TPW, px = pp - py * TPW;
      const int ph = ph0 + py, pw = pw0 + px;
      if (ph < Hp && pw < Wp) {
        f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&conv_s[(2 * py + dy) * CC + 2 * px + dx][c4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
          }
        auto* dst = out + (((size_t)n * Hp + ph) * Wp + pw) * 64 + c4;
        if constexpr (X3) {
          *reinterpret_cast<f32x4*>(dst) = m;
        } else {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          const bf16x4 hv = {(__bf16)m[0], (__bf16)m[1], (__bf16)m[2], (__bf16)m[3]};
          *reinterpret_cast<bf16x4*>(dst) = hv;
        }
      }
    }
    STEM_STAMP(6);
  }  // tl
}

// 16 lanes per input pixel (64 channels as 16 x float4, coalesced), xor-reduce, 4 taps out.
__global__ __launch_bounds__(256) void convt2_sigmoid_kernel(const float* __restrict__ in, const float* __restrict__ w4x64,
                                                             float bias, float* __restrict__ prob,
                                                             uint8_t* __restrict__ bitmap, float thresh, int H2,
                                                             int W2, long long P) {
  const int l16 = threadIdx.x & 15;
  f32x4 wt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wt[t] = *reinterpret_cast<const f32x4*>(w4x64 + t * 64 + 4 * l16);
  const long long stride = (long long)gridDim.x * 16;
  for (long long pix = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4; pix < P; pix += stride) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + pix * 64 + 4 * l16);
    float s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = v[0] * wt[t][0] + v[1] * wt[t][1] + v[2] * wt[t][2] + v[3] * wt[t][3];
#pragma unroll
    for (int k = 8; k >= 1; k >>= 1)
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] += __shfl_xor(s[t], k, 16);
    if (l16 < 4) {
      const float z = (l16 == 0 ? s[0] : l16 == 1 ? s[1] : l16 == 2 ? s[2] : s[3]) + bias;
      const float pr = 1.0f / (1.0f + expf(-z));
      const long long hw = (long long)H2 * W2;
      const long long n = pix / hw;
      const long long rem = pix - n * hw;
      const int i = (int)(rem / W2), j = (int)(rem - (long long)i * W2);
      const size_t o = ((size_t)n * (2 * H2) + 2 * i + (l16 >> 1)) * (size_t)(2 * W2) + 2 * j + (l16 & 1);
      prob[o] = pr;
      if (bitmap) bitmap[o] = pr > thresh ? 1 : 0;
    }
  }
}

__global__ __launch_bounds__(256) void binarize_kernel(const float* __restrict__ prob, uint8_t* __restrict__ bitmap,
                                                       float thresh, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) bitmap[i] = prob[i] > thresh ? 1 : 0;
}

// binarize(pred, thresh) (metrics.rs:129-131) as a packed image for the host contour tracer: bit (i & 31) of word
// i >> 5 for the row-major pixel index i; one 64-bit ballot per wave = two words, 5 bytes per pixel moved instead of 8
// Image blockIdx.y is packed on its own (its first pixel is bit 0 of its first word), padded to whole 64-bit words.
__global__ __launch_bounds__(256) void binarize_pack_kernel(const float* __restrict__ prob, unsigned long long* __restrict__ bits64,
                                                            float thresh, size_t px_per_image, size_t words64_per_image) {
  const float* img = prob + (size_t)blockIdx.y * px_per_image;
  unsigned long long* out = bits64 + (size_t)blockIdx.y * words64_per_image;
  const size_t n64 = words64_per_image * 64, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += stride) {  // whole waves: every lane reaches the ballot
    const unsigned long long b = __ballot(i < px_per_image && img[i] > thresh);
    if ((threadIdx.x & 63) == 0) out[i >> 6] = b;
  }
}

}  // namespace

size_t binarize_pack_words(size_t px_per_image) { return (px_per_image + 63) / 64 * 2; }

void launch_binarize_pack(const float* prob, uint32_t* bits, float thresh, int n_images, size_t px_per_image, hipStream_t s) {
  if (n_images <= 0 || px_per_image == 0) return;
  if (n_images > 65535) fail(OCR_ERR_INVALID, "binarize_pack: %d images in one call", n_images);
  const size_t w64 = (px_per_image + 63) / 64;
  size_t blocks = (w64 * 64 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(binarize_pack_kernel, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, s, prob,
                     reinterpret_cast<unsigned long long*>(bits), thresh, px_per_image, w64);
  OCR_HIP(hipGetLastError());
}

void launch_stem(const void* x, int x_u8, const float* w49x64, const float* scale, const float* bias, void* out, int out_bf16, int N,
                 int H, int W, hipStream_t s) {
  if (H % 32 || W % 32 || N <= 0 || N > 65535) fail(OCR_ERR_INVALID, "stem: bad shape N=%d H=%d W=%d", N, H, W);
  const int Hp = H / 4, Wp = W / 4;
  dim3 grid(((Wp + TPW - 1) / TPW + TL - 1) / TL, (Hp + TPH - 1) / TPH, N);
  const float* xf = static_cast<const float*>(x);
  const uint8_t* xb = static_cast<const uint8_t*>(x);
  if (out_bf16 && x_u8) hipLaunchKernelGGL((stem_kernel<__bf16, uint8_t>), grid, dim3(NT), 0, s, xb, w49x64, scale, bias, static_cast<__bf16*>(out), H, W);
  else if (out_bf16) hipLaunchKernelGGL((stem_kernel<__bf16, float>), grid, dim3(NT), 0, s, xf, w49x64, scale, bias, static_cast<__bf16*>(out), H, W);
  else if (x_u8) hipLaunchKernelGGL((stem_kernel<float, uint8_t>), grid, dim3(NT), 0, s, xb, w49x64, scale, bias, static_cast<float*>(out), H, W);
  else hipLaunchKernelGGL((stem_kernel<float, float>), grid, dim3(NT), 0, s, xf, w49x64, scale, bias, static_cast<float*>(out), H, W);
  OCR_HIP(hipGetLastError());
}

// conv1 [64][1][7][7] (f32) -> bf16 MFMA B fragments [2 channel tiles][4 steps][64 lanes][8]: element j of lane
// (co = 32 ct + (l & 31), half = l >> 5) is w[co][kh = 2 s + half][kw = j], zero for kh = 7 or kw = 7
// the same as three fragment sets hi / mid / lo (w = hi + mid + lo exactly) for the split-bf16 stem
std::vector<uint16_t> stem_x3_fragments(const float* w64x49) {
  auto bf = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  auto up = [](uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  const size_t plane = (size_t)2 * 4 * 64 * 8;
  std::vector<uint16_t> fr(3 * plane, 0);
  for (int ct = 0; ct < 2; ++ct)
    for (int s = 0; s < 4; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int co = 32 * ct + (l & 31), kh = 2 * s + (l >> 5);
          if (kh >= 7 || j >= 7) continue;
          const float w = w64x49[co * 49 + kh * 7 + j];
          const uint16_t h = bf(w);
          const float r1 = w - up(h);
          const uint16_t m = bf(r1);
          const size_t o = (((size_t)ct * 4 + s) * 64 + l) * 8 + j;
          fr[o] = h;
          fr[plane + o] = m;
          fr[2 * plane + o] = bf(r1 - up(m));
        }
  return fr;
}

std::vector<uint16_t> stem_bf16_fragments(const float* w64x49) {
  auto bf = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);  // finite weights: round to nearest even
  };
  std::vector<uint16_t> fr((size_t)2 * 4 * 64 * 8, 0);
  for (int ct = 0; ct < 2; ++ct)
    for (int s = 0; s < 4; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int co = 32 * ct + (l & 31), kh = 2 * s + (l >> 5);
          if (kh < 7 && j < 7) fr[(((size_t)ct * 4 + s) * 64 + l) * 8 + j] = bf(w64x49[co * 49 + kh * 7 + j]);
        }
  return fr;
}

void launch_stem_bf16(const void* x, int x_u8, const void* wfrag, const float* scale, const float* bias, void* out, int N, int H, int W,
                      hipStream_t s) {
  if (H % 32 || W % 32 || N <= 0 || N > 65535) fail(OCR_ERR_INVALID, "stem: bad shape N=%d H=%d W=%d", N, H, W);
  const int Hp = H / 4, Wp = W / 4;
  using TC = StemTileS;   // (8 x 7: 0.126 against 0.118 ms - with one image and eight MFMAs per wave and tile this form is bound by staging and pooling)
  dim3 grid(((Wp + TC::TPW - 1) / TC::TPW + TL - 1) / TL, (Hp + TC::TPH - 1) / TC::TPH, N);
  if (x_u8) hipLaunchKernelGGL((stem_bf16_kernel<false, uint8_t, __bf16, TC>), grid, dim3(NT), 0, s, static_cast<const uint8_t*>(x), static_cast<const u32x4*>(wfrag),
                               scale, bias, static_cast<__bf16*>(out), H, W);
  else hipLaunchKernelGGL((stem_bf16_kernel<false, float, __bf16, TC>), grid, dim3(NT), 0, s, static_cast<const float*>(x), static_cast<const u32x4*>(wfrag), scale, bias,
                          static_cast<__bf16*>(out), H, W);
  OCR_HIP(hipGetLastError());
}

void launch_stem_x3(const void* x, int x_u8, const void* wfrag3, const float* scale, const float* bias, float* out, int N, int H, int W,
                    hipStream_t s) {
  if (H % 32 || W % 32 || N <= 0 || N > 65535) fail(OCR_ERR_INVALID, "stem: bad shape N=%d H=%d W=%d", N, H, W);
  const int Hp = H / 4, Wp = W / 4;
  using TC = StemTileL;   // (4 x 6: 0.227 against 0.177 ms at 32 x 640 x 640)
  dim3 grid(((Wp + TC::TPW - 1) / TC::TPW + TL - 1) / TL, (Hp + TC::TPH - 1) / TC::TPH, N);
  if (x_u8) hipLaunchKernelGGL((stem_bf16_kernel<true, uint8_t, float, TC>), grid, dim3(NT), 0, s, static_cast<const uint8_t*>(x), static_cast<const u32x4*>(wfrag3),
                               scale, bias, out, H, W);
  else hipLaunchKernelGGL((stem_bf16_kernel<true, float, float, TC>), grid, dim3(NT), 0, s, static_cast<const float*>(x), static_cast<const u32x4*>(wfrag3), scale, bias,
                          out, H, W);
  OCR_HIP(hipGetLastError());
}

#ifdef STEM_STAMPS
void stem_read_stamps(long long* out) { OCR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stem_stamps), sizeof(g_stem_stamps))); }
#endif

void launch_convt2_sigmoid(const float* in, const float* w4x64, float bias, float* prob, uint8_t* bitmap,
                           float thresh, int N, int H2, int W2, hipStream_t s) {
  if (W2 % 4) fail(OCR_ERR_INVALID, "convt2: W/2 = %d must be a multiple of 4", W2);
  const long long P = (long long)N * H2 * W2;
  long long blocks = (P * 16 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(convt2_sigmoid_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, w4x64, bias, prob, bitmap,
                     thresh, H2, W2, P);
  OCR_HIP(hipGetLastError());
}

void launch_binarize(const float* prob, uint8_t* bitmap, float thresh, size_t n, hipStream_t s) {
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks == 0) return;
  hipLaunchKernelGGL(binarize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, prob, bitmap, thresh, n);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
