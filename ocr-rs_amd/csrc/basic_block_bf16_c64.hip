// One ResNet BasicBlock of layer1 in the bf16 precision as ONE launch (model.rs:40-55: conv3x3 + BN + ReLU, conv3x3 + BN, + x, ReLU;
// 64 -> 64 channels, stride 1): the activation between the two convs never reaches HBM.  conv3x3_bf16_c64.hip runs the same two convs
// as two launches bound by HBM (read 105 MB + residual 105 MB + write 105 MB per launch at 32 x 160 x 160); this kernel reads x once
// and writes the block's output once.  Both results are bit-identical: the same fragments, the same order of taps and k-steps in the
// accumulators, the same folded-BN fma, the intermediate rounded to bf16 exactly where the first launch stores it.
//
//   * a workgroup of EIGHT waves owns an 8 x 16 pixel block of the output; it needs the 10 x 18 block of the intermediate and the
//     12 x 20 patch of x (30 KB, one 128-byte LDS row per pixel, by LDS-DMA, double-buffered);
//   * waves 0-3 hold conv1's weights in registers (144 VGPRs, as conv3x3_bf16_c64.hip), waves 4-7 conv2's: a SIMD hosts wave w and
//     wave w + 4, one of each, so both weight sets are register-resident on every SIMD and the matrix pipe is shared by a producer and
//     a consumer.  In iteration i the conv1 waves turn patch i + 1 into intermediate i + 1 (180 pixels = six 32-pixel MFMA row tiles,
//     three per wave, one after the other) while the conv2 waves turn intermediate i into
//     output i.  One s_barrier per iteration; the intermediate (2 x 24 KB) and the patches (2 x 30 KB) are double-buffered;
//   * intermediate pixels outside the image are conv2's zero padding, not conv1 evaluated there: written as zeros;
//   * conv2's epilogue is wave-local (64 pixels x 32 channels through 9 KB of LDS per wave, no workgroup barrier): folded BN, + x
//     (16-byte loads issued before the matrix phase), ReLU, bf16, 16-byte stores.
// Matrix work is 1.25 x the two launches' (the halo of the intermediate is computed by every block that needs it): 180 MFMAs of
// 32 cycles per SIMD and block.
#include <cstdio>
#include <cstring>

#include "common.hpp"
#include "bf16_c64_tiles.hpp"

namespace ocr {
namespace {
using namespace bf16_c64;


template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
// 16-byte buffer load into registers as inline asm (not awaited by the compiler): valid after an explicit s_waitcnt + settle()
template <typename R>
__device__ __forceinline__ u32x4 load16_async(R rsrc, unsigned voff) {
  u32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
  return v;
}
__device__ __forceinline__ void settle(u32x4& v) { asm volatile("" : "+v"(v)::"memory"); }

#if defined(BB_STAMPS)
// diagnostic build: s_memtime at the phases of block iterations 4 and 5 of workgroups 0 .. 3, lane 0 of every wave (profiles/r06_bf16_block_stamps.txt)
__device__ unsigned long long bb_stamp_buf[4 * 8 * 2 * 16];
#define BB_STAMP(k) do { if (blockIdx.x < 4 && (bb_it == 4 || bb_it == 5) && lane == 0) bb_stamp_buf[((blockIdx.x * 8 + wave) * 2 + (bb_it - 4)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_STAMP(k) do { } while (0)
#endif

struct BlockArgs {
  const __bf16* x;        // [N][H][W][64]
  const u32x4* wfrag1;    // conv1 / conv2: [2 ct][9 taps][4 s][64 lanes] x 8 bf16 (conv3x3_bf16_c64_fragments)
  const u32x4* wfrag2;
  const float* scale1;    // folded BN (may be null)
  const float* bias1;
  const float* scale2;
  const float* bias2;
  __bf16* y;              // [N][H][W][64]
  unsigned x_bytes;
  int H, W, bh, bw;
  int nblocks;
};

constexpr int XW = 20;                      // patch of x: 12 x 20 pixels
constexpr int MW = 18;                      // intermediate: 10 x 18 pixels
constexpr int PATCH_BYTES = 12 * XW * 128;  // 30 720 = 30 DMA instructions
constexpr int MID_BYTES = 10 * MW * 128 + 1536;   // 180 pixels, a multiple of 1 KB
constexpr int EXROW = 36;                   // floats per exchange row (32 channels + pad)
constexpr int EX_WAVE_BYTES = 64 * EXROW * 4;
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 2 * MID_BYTES + 4 * EX_WAVE_BYTES;   // 147 456: one workgroup per CU
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;

__global__ __launch_bounds__(512, 1) void basic_block_bf16_c64_kernel(BlockArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool second = wave >= 4;                  // waves 4-7: conv2 (wave w shares its SIMD with wave w - 4)
  const int w4 = wave & 3, ct = w4 & 1, rp = w4 >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  [[maybe_unused]] int bb_it = 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  unsigned char* const patch0 = lds;
  unsigned char* const mid0 = lds + 2 * PATCH_BYTES;
  float* const ex = reinterpret_cast<float*>(lds + 2 * PATCH_BYTES + 2 * MID_BYTES + w4 * EX_WAVE_BYTES);

  // this wave's 32 output channels of its conv: resident for the whole launch
  bf16x8 wreg[9][4];
  {
    const u32x4* wf = second ? p.wfrag2 : p.wfrag1;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) wreg[t][s] = __builtin_bit_cast(bf16x8, wf[((ct * 9 + t) * 4 + s) * 64 + lane]);
  }
  const float* scp = second ? p.scale2 : p.scale1;
  const float* bip = second ? p.bias2 : p.bias1;
  const float sc = scp ? scp[32 * ct + l31] : 1.f;
  const float bi = bip ? bip[32 * ct + l31] : 0.f;

  // the blocks of this workgroup: workgroup ids go round the eight XCDs, so each XCD takes one contiguous eighth of the blocks and
  // its workgroups walk it side by side - the halo a block shares with its neighbours is in that XCD's L2
  int first, step, end;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    first = (int)((long long)p.nblocks * xcd / 8) + (int)(blockIdx.x >> 3);
    end = (int)((long long)p.nblocks * (xcd + 1) / 8);
    step = per;
  } else {
    first = blockIdx.x;
    step = gridDim.x;
    end = p.nblocks;
  }
  auto coords = [&](int bb, int& n_, int& y0_, int& x0_) {
    x0_ = 16 * (bb % p.bw);
    bb /= p.bw;
    y0_ = 8 * (bb % p.bh);
    n_ = bb / p.bh;
  };
  // (block-invariant lane values are computed per block behind an opaque zero: the compiler would otherwise keep them in registers -
  // or spill them - across the loop beside the 144 weight registers)
  auto opaque_zero = [] {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
  };
  // the 12 x 20 patch of block bb: 30 DMA instructions of 8 pixels (zero padding = out-of-range lanes), instructions w, w + 8, ... by
  // wave w - an LDS-DMA instruction costs its wave 100 - 300 cycles of issue, so all eight waves share them.  A lane's pixel is
  // (py, pxx) = divmod(8 k + lane / 8, 20): the quotient and remainder of 8 k are scalar
  auto issue_patch = [&](int bb, int buf) {
    int pn, py0, px0;
    coords(bb, pn, py0, px0);
    const int lz = lane + opaque_zero();
    const int sub = lz >> 3, slot = lz & 7;
    const bool interior = py0 >= 2 && py0 + 10 <= p.H && px0 >= 2 && px0 + 18 <= p.W;   // (uniform) no lane of this patch is padding
    const int base = ((pn * p.H + py0 - 2) * p.W + px0 - 2) * 128;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = wave + 8 * m;
      if (k >= 30) break;   // (uniform) waves 6 and 7 have three
      const int ka = (8 * k) / XW, kb = (8 * k) % XW;
      const int t = kb + sub;
      const int py = ka + (t >= XW ? 1 : 0), pxx = t >= XW ? t - XW : t;
      const int chunk = slot ^ ((pxx >> 1) & 7) ^ ((py & 1) << 2);
      unsigned off = (unsigned)(base + (py * p.W + pxx) * 128 + chunk * 16);
      if (!interior) {
        const int yy = py0 - 2 + py, xx = px0 - 2 + pxx;
        if (!((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)) off = OOB;
      }
      if (!(BB_ABL & 4)) dma16(x_rsrc, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * PATCH_BYTES + k * 1024)), off, 0);
    }
  };

  // conv1: the 10 x 18 intermediate in six row tiles: T = 0 .. 4 rows 2 T, 2 T + 1 x columns 0 .. 15; T = 5 the two columns
  // left over, row i >> 1 (of ten), column 16 + (i & 1) - rows i = 20 .. 31 of that tile compute pixels eight rows up again and are
  // not stored.  Wave rp takes tiles 3 rp .. 3 rp + 2
  auto mid_pixel = [&](int T, int i, int& mrow, int& mcol, bool& valid) {
    if (T < 5) {
      mrow = 2 * T + (i >> 4);
      mcol = tile_col8(i);
      valid = true;
    } else {
      const int r = i >> 1;
      valid = r < 10;
      mrow = valid ? r : r - 8;
      mcol = 16 + (i & 1);
    }
  };
  // folded BN + ReLU + bf16 of a row tile of the intermediate: 2-byte stores, pixel-major.  Element e of a lane is tile row
  // i = u + 4 half, u = (e & 3) + 8 (e >> 2), channel ch1 = 32 ct + l31; its LDS address is one of four lane offsets (the XOR of the
  // channel's chunk with the pixel's swizzle, which depends on e only through (e >> 1) & 1 and (e >> 2) & 1) + an immediate:
  //   full tile, i < 16:  pixel (2 T, u + 4 half),                  sw = (u >> 1) ^ (half << 1)
  //   full tile, i >= 16: pixel (2 T + 1, ((u - 16) ^ 8) + 4 half), sw = ((u - 16) >> 1) ^ (half << 1)
  //   sixth tile:         pixel ((u >> 1) + 2 half, 16 + (u & 1)),  sw = 4 ((u >> 1) & 1); stored while i < 20
  const int ch1 = 32 * ct + l31, cg1 = ch1 >> 3, cb1 = (ch1 & 7) * 2;
  auto store_full = [&](const f32x16& acc, int j, unsigned char* mid) {
    int offa[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cc = (c & 1) | ((c & 2) << 1);   // u >> 1 (mod 8) of the element: 0, 1, 4, 5
      offa[c] = ((cg1 ^ (half << 1) ^ cc) << 4) + cb1 + half * 4 * 128;
    }
    unsigned char* base = mid + (3 * rp + j) * (2 * MW * 128);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = ((e >> 1) & 1) | (((e >> 2) & 1) << 1);
      const int u = (e & 3) + 8 * (e >> 2);
      const float v = fmaxf(__builtin_fmaf(acc[e], sc, bi), 0.f);
      *reinterpret_cast<__bf16*>(base + offa[c] + (e < 8 ? u : MW + ((u - 16) ^ 8)) * 128) = (__bf16)v;
    }
  };
  auto store_left = [&](const f32x16& acc, unsigned char* mid) {
    int offl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) offl[k] = ((cg1 ^ (k << 2)) << 4) + cb1 + half * 2 * MW * 128;
#pragma unroll
    for (int e = 0; e < 12; ++e) {   // tile rows 24 .. 31 are never pixels
      const int u = (e & 3) + 8 * (e >> 2);
      const float v = fmaxf(__builtin_fmaf(acc[e], sc, bi), 0.f);
      __bf16* dst = reinterpret_cast<__bf16*>(mid + offl[(e >> 1) & 1] + (u >> 1) * MW * 128 + (16 + (e & 1)) * 128);
      if (e < 8 || half == 0) *dst = (__bf16)v;
    }
  };
  // a block at the edge of the image: the intermediate's pixels outside the image are conv2's zero padding.  Each wave clears, among
  // the pixels x 32 channels it has just written, the ones outside (after its own stores: LDS runs in order)
  auto clear_outside = [&](unsigned char* mid, int y0, int x0) {
    if (y0 >= 1 && y0 + 8 < p.H && x0 >= 1 && x0 + 16 < p.W) return;   // (uniform)
    const int lz = lane + opaque_zero();
#pragma unroll 1
    for (int it = 0; it < 6; ++it) {
      int mrow, mcol;
      bool valid;
      mid_pixel(3 * rp + (it >> 1), 16 * (it & 1) + (lz >> 2), mrow, mcol, valid);
      const int chunk = 4 * ct + (lz & 3);
      const int yy = y0 - 1 + mrow, xx = x0 - 1 + mcol;
      const bool inside = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const int sw = ((mcol >> 1) & 7) ^ ((mrow & 1) << 2);
      if (valid && !inside) *reinterpret_cast<u32x4*>(mid + (mrow * MW + mcol) * 128 + ((chunk ^ sw) << 4)) = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto conv1 = [&](int bb, int buf) {
    int n, y0, x0;
    coords(bb, n, y0, x0);
    const unsigned char* patch = patch0 + buf * PATCH_BYTES;
    unsigned char* mid = mid0 + buf * MID_BYTES;
    // this lane's pixels: tiles 3 rp, + 1 (, + 2 for rp = 0) are 2 x 16 tiles - one column map and row parity for all of them;
    // the sixth tile (rp = 1, j = 2) has its own
    const int lz = l31 + opaque_zero();
    int pixb[3], xb[3][2], xbl[3][2];
    xor_bases(tile_col8(lz), lz >> 4, half, xb);
    xor_bases(16 + (lz & 1), (lz >> 1) & 1, half, xbl);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int mrow, mcol;
      bool valid;
      mid_pixel(3 * rp + j, lz, mrow, mcol, valid);
      pixb[j] = (mrow * XW + mcol) * 128;   // tap (ty, tx) of intermediate pixel (mrow, mcol) = patch pixel (mrow + ty, mcol + tx)
    }
    const bool left = rp == 1;
    conv_tiles<3, XW>(patch, pixb, [&](int j, int tx, int q) { return j == 2 && left ? xbl[tx][q] : xb[tx][q]; }, wreg,
                      [&](int j, const f32x16& acc) {
      BB_STAMP(2 + 2 * j);
      if (BB_ABL & 8) {
        if (acc[0] == 123.456f) p.y[j] = (__bf16)1.f;
      } else if (j < 2 || rp == 0) {
        store_full(acc, j, mid);
      } else {
        store_left(acc, mid);
      }
      BB_STAMP(3 + 2 * j);
    });
    if (!(BB_ABL & 8)) clear_outside(mid, y0, x0);
    BB_STAMP(8);
  };
  auto conv2 = [&](int bb, int buf) {
    int n, y0, x0;
    coords(bb, n, y0, x0);
    // x at this wave's 64 pixels x 32 channels (the block's residual): in flight under the matrix phase
    u32x4 res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int item = k * 64 + lane;
      const int px = item >> 2, c8 = 32 * ct + (item & 3) * 8;
      const int yy = y0 + 4 * rp + (px >> 4), xx = x0 + (px & 15);
      if (BB_ABL & 16) res[k] = u32x4{0u, 0u, 0u, 0u};
      else res[k] = load16_async(x_rsrc, (yy < p.H && xx < p.W) ? (unsigned)((((n * p.H + yy) * p.W + xx) * 64 + c8) * 2) : OOB);
    }
    BB_STAMP(2);
    // row tile r of the wave = output rows 4 rp + 2 r, + 1 of the block
    const int lz = l31 + opaque_zero();
    int pixb[2], xb[3][2];
    xor_bases(tile_col8(lz), lz >> 4, half, xb);
#pragma unroll
    for (int r = 0; r < 2; ++r) pixb[r] = ((4 * rp + 2 * r + (lz >> 4)) * MW + tile_col8(lz)) * 128;
    // folded BN, then pixel-major through this wave's own LDS rows: row = 32 r + tile row, column = channel l31
    conv_tiles<2, MW>(mid0 + buf * MID_BYTES, pixb, [&](int, int tx, int q) { return xb[tx][q]; }, wreg, [&](int r, const f32x16& acc) {
      BB_STAMP(3 + 2 * r);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
        if (!(BB_ABL & 16)) ex[(32 * r + i) * EXROW + l31] = __builtin_fmaf(acc[e], sc, bi);
        else if (acc[e] == 123.456f) p.y[i] = (__bf16)1.f;
      }
    });
    if (BB_ABL & 16) return;
    BB_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the residual loads (and the exchange rows of this wave)
    BB_STAMP(8);
#pragma unroll
    for (int k = 0; k < 4; ++k) settle(res[k]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int item = k * 64 + lane;
      const int px = item >> 2, c8 = (item & 3) * 8;
      const int prow = px >> 4, pcol = px & 15;
      const int yy = y0 + 4 * rp + prow, xx = x0 + pcol;
      const int er = 32 * (prow >> 1) + ((prow & 1) ? 16 + (pcol ^ 8) : pcol);   // tile row of this pixel (inverse of tile_col8)
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ex[er * EXROW + c8]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ex[er * EXROW + c8 + 4]);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const bf16x8 rr = __builtin_bit_cast(bf16x8, res[k]);
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (__bf16)fmaxf(v[e] + (float)rr[e], 0.f);
      if (yy < p.H && xx < p.W) *reinterpret_cast<bf16x8*>(p.y + ((((size_t)n * p.H + yy) * p.W + xx) * 64 + 32 * ct + c8)) = h;
    }
    BB_STAMP(9);
    // the next iteration's exchange rows are written after its matrix phase, by this wave, after these reads (LDS runs in order)
  };

  if (first >= end) return;   // (uniform over the workgroup)
#ifndef BB_PRIO
#define BB_PRIO 1
#endif
  if ((BB_PRIO == 1 && !second) || (BB_PRIO == 2 && second)) __builtin_amdgcn_s_setprio(3);
  issue_patch(first, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (first + step < end) issue_patch(first + step, 1);
  if (!second) conv1(first, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int blk = first; blk < end; blk += step, buf ^= 1, ++bb_it) {
    BB_STAMP(0);
    // patch i + 2 into the buffer conv1 read in the previous iteration
    if (blk + 2 * step < end) issue_patch(blk + 2 * step, buf);
    BB_STAMP(1);
    if (!second) {
      // intermediate i + 1 from the patch that landed before the barrier
      if (blk + step < end && !(BB_ABL & 32)) conv1(blk + step, buf ^ 1);
    } else {
      if (!(BB_ABL & 64)) conv2(blk, buf);
    }
    BB_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BB_STAMP(11);
    __builtin_amdgcn_s_barrier();   // intermediate i + 1 and patch i + 2 complete; intermediate i free
  }
#endif
}

}  // namespace

#if defined(BB_STAMPS)
static void basic_block_dump_stamps() {
  static unsigned long long h[4 * 8 * 2 * 16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(bb_stamp_buf), sizeof(h)) != hipSuccess) return;
  for (int wg = 0; wg < 4; ++wg)
    for (int w = 0; w < 8; ++w)
      for (int it = 0; it < 2; ++it) {
        const unsigned long long* q = &h[((wg * 8 + w) * 2 + it) * 16];
        fprintf(stderr, "STAMP wg %d wave %d (%s) it %d:", wg, w, w < 4 ? "conv1" : "conv2", it);
        for (int k = 1; k <= 11; ++k) fprintf(stderr, " %lld", q[k] ? (long long)(q[k] - q[0]) : -1ll);
        if (it == 1) fprintf(stderr, "  | block period %lld", (long long)(q[0] - h[((wg * 8 + w) * 2) * 16]));
        fprintf(stderr, "\n");
      }
}
#endif

bool basic_block_bf16_c64_applicable(int N, int H, int W) {
  return N > 0 && H > 0 && W > 0 && (long long)N * H * W * 128 < (1ll << 31);
}

void launch_basic_block_bf16_c64(const void* x, const void* wfrag1, const float* scale1, const float* bias1, const void* wfrag2, const float* scale2,
                                 const float* bias2, void* y, int N, int H, int W, int num_cus, hipStream_t s) {
  if (!basic_block_bf16_c64_applicable(N, H, W)) fail(OCR_ERR_INVALID, "basic_block_bf16_c64: bad shape N=%d H=%d W=%d (tensors must be < 2^31 bytes)", N, H, W);
  BlockArgs a{};
  a.x = static_cast<const __bf16*>(x);
  a.wfrag1 = static_cast<const u32x4*>(wfrag1);
  a.wfrag2 = static_cast<const u32x4*>(wfrag2);
  a.scale1 = scale1;
  a.bias1 = bias1;
  a.scale2 = scale2;
  a.bias2 = bias2;
  a.y = static_cast<__bf16*>(y);
  a.x_bytes = (unsigned)((long long)N * H * W * 64 * 2);
  a.H = H;
  a.W = W;
  a.bh = (H + 7) / 8;
  a.bw = (W + 15) / 16;
  const long long blocks = (long long)N * a.bh * a.bw;
  if (blocks >= (1ll << 31)) fail(OCR_ERR_INVALID, "basic_block_bf16_c64: grid too large");
  a.nblocks = (int)blocks;
  const long long resident = num_cus > 0 ? num_cus : 256;   // one workgroup per CU
  const unsigned grid = blocks > resident ? (unsigned)resident : (unsigned)blocks;
  hipLaunchKernelGGL(basic_block_bf16_c64_kernel, dim3(grid), dim3(512), 0, s, a);
  OCR_HIP(hipGetLastError());
#if defined(BB_STAMPS)
  static int launches = 0;
  if (++launches == 15) {
    (void)hipStreamSynchronize(s);
    basic_block_dump_stamps();
  }
#endif
}

}  // namespace ocr
