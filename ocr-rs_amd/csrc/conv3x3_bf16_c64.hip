// 3x3 s1 p1 convolution, 64 -> 64 channels, bf16 operands on v_mfma_f32_32x32x16_bf16 (f32 accumulate), for the
// opt-in bf16 precision: layer1's four convs and the p2 lateral term (model.rs:40-55, :126-129) - the layers
// where conv_igemm's im2col staging re-reads every input pixel nine times through L2 -> LDS (64 FLOP per staged byte
// at best for 64 output channels: 460-620 TFLOP/s measured, neither roofline).  Here the input is staged ONCE:
//   * a workgroup owns an 8 x 16 pixel block and all 64 output channels; its 10 x 18 x 64 bf16 patch (23 KB, one
//     128-byte LDS row per pixel) arrives by LDS-DMA, double-buffered across the blocks of a persistent workgroup;
//   * the nine taps are nine shifted reads of that patch (ds_read_b128): layout, swizzle and the lane -> pixel map that keep every
//     read free of bank conflicts, and the software-pipelined (tile, tap) sequence with one v_xor of address arithmetic per read, are
//     bf16_c64_tiles.hpp's (shared with basic_block_bf16_c64.hip);
//   * the weights live in REGISTERS for the whole launch: wave w owns output channels 32 (w & 1) .. + 31 and pixel
//     rows 4 (w >> 1) .. + 3 of the block (two 32-pixel MFMA row tiles): 9 taps x 4 k-steps x 4 VGPRs = 144 VGPRs.
// HBM traffic is the compulsory one (+ 40 % halo from L2): the kernel is bound by HBM, not by operand staging.
// Epilogue: folded batch norm on the accumulators, through LDS into pixel-major order, residual add, ReLU, bf16, 16-byte stores.
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

// 16-byte buffer load into registers as inline asm (invisible to the compiler's wait insertion, like the LDS-DMA: a
// compiler-visible load would be awaited with a vmcnt that also covers the patch DMA issued after it).  Valid after an
// explicit s_waitcnt + settle(); out-of-range offsets read as zero.
template <typename R>
__device__ __forceinline__ u32x4 load16_async(R rsrc, unsigned voff) {
  u32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
  return v;
}
__device__ __forceinline__ void settle(u32x4& v) { asm volatile("" : "+v"(v)::"memory"); }

#if defined(C64_STAMPS)
// diagnostic build: s_memtime at the phases of block iterations 4 and 5 of workgroups 0 .. 7, every wave's lane 0 (tools/bf16_stamps.py)
__device__ unsigned long long c64_stamp_buf[8 * 4 * 2 * 16];
#define C64_STAMP(k) do { if (blockIdx.x < 8 && (it == 4 || it == 5) && lane == 0) c64_stamp_buf[((blockIdx.x * 4 + wave) * 2 + (it - 4)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C64_STAMP(k) do { } while (0)
#endif

struct C64Args {
  const __bf16* x;         // [N][H][W][64]
  const u32x4* wfrag;      // [2 ct][9 taps][4 s][64 lanes] x 8 bf16
  const float* scale;      // folded BN (may be null)
  const float* bias;
  const __bf16* residual;  // [N][H][W][64], may be null
  __bf16* y;               // [N][H][W][64]
  unsigned x_bytes;
  int H, W, bh, bw;
  int relu;
  int nblocks;
};

[[maybe_unused]] constexpr int PH = 10, PWD = 18;
constexpr int PATCH_BYTES = 23 * 1024;   // 180 pixels x 128 B, rounded up to whole DMA instructions
constexpr int EXROW = 68;                // floats per exchange row (64 + pad)
[[maybe_unused]] constexpr int EX_BYTES = 128 * EXROW * 4;
[[maybe_unused]] constexpr int LDS_BYTES = 2 * PATCH_BYTES + EX_BYTES;   // 81 920: two workgroups per CU
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;

__global__ __launch_bounds__(256, 2) void conv3x3_bf16_c64_kernel(C64Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 1, rp = wave >> 1;       // channel tile, pair of row tiles (pixel rows 4 rp .. 4 rp + 3)
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.residual ? p.residual : p.x), 0, p.x_bytes, 0x00020000);

  // weights of this wave's 32 output channels: resident for the whole launch
  bf16x8 wreg[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) wreg[t][s] = __builtin_bit_cast(bf16x8, p.wfrag[((ct * 9 + t) * 4 + s) * 64 + lane]);
  const float sc = p.scale ? p.scale[32 * ct + l31] : 1.f;
  const float bi = p.bias ? p.bias[32 * ct + l31] : 0.f;

  auto coords = [&](int bb, int& n_, int& y0_, int& x0_) {
    x0_ = 16 * (bb % p.bw);
    bb /= p.bw;
    y0_ = 8 * (bb % p.bh);
    n_ = bb / p.bh;
  };
  // the patch of block bb into buffer `buf`: 23 DMA instructions of 8 pixels (zero padding = out-of-range lanes);
  // pixel px = (py, pxx) of the patch lands in row px, its 16-byte chunk c in slot c ^ ((pxx >> 1) & 7) ^ 4 (py & 1)
  auto issue_patch = [&](int bb, int buf) {
    int pn, py0, px0;
    coords(bb, pn, py0, px0);
    const int sub = lane >> 3, slot = lane & 7;
    for (int k = wave; k < 23; k += 4) {
      const int px = 8 * k + sub;
      const int py = px / PWD, pxx = px - py * PWD;
      const int yy = py0 - 1 + py, xx = px0 - 1 + pxx;
      const bool inside = px < PH * PWD && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const int chunk = slot ^ ((pxx >> 1) & 7) ^ ((py & 1) << 2);
      const unsigned off = inside ? (unsigned)((((pn * p.H + yy) * p.W + xx) * 64 + chunk * 8) * 2) : OOB;
      dma16(x_rsrc, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * PATCH_BYTES + k * 1024)), off, 0);
    }
  };

  float* ex = reinterpret_cast<float*>(lds + 2 * PATCH_BYTES);
  issue_patch(blockIdx.x, 0);
  int buf = 0;
  [[maybe_unused]] int it = 0;
  for (int blk = blockIdx.x; blk < p.nblocks; blk += gridDim.x, buf ^= 1, ++it) {
    int n, y0, x0;
    coords(blk, n, y0, x0);
    C64_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    C64_STAMP(1);
    __builtin_amdgcn_s_barrier();
    C64_STAMP(2);  // this block's patch is complete; every wave has left the previous block's store phase
    // residual rows of this thread's store items first, then the NEXT block's patch: both fly under this block's MFMAs
    const bool has_next = blk + (int)gridDim.x < p.nblocks;
    u32x4 res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      res[k] = u32x4{0u, 0u, 0u, 0u};
      if (p.residual) {
        const int item = k * 256 + tid;
        const int px = item >> 3, c8 = (item & 7) * 8;
        const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
        res[k] = load16_async(r_rsrc, (yy < p.H && xx < p.W) ? (unsigned)((((n * p.H + yy) * p.W + xx) * 64 + c8) * 2) : OOB);
      }
    }
    if (has_next) issue_patch(blk + (int)gridDim.x, buf ^ 1);
    C64_STAMP(3);

    const unsigned char* patch = lds + buf * PATCH_BYTES;
    // this lane's pixels: row tile r of the wave covers block rows 4 rp + 2 r, + 1; lane -> (row l31 >> 4, column l31 & 15, XOR 8 in
    // the second row).  Computed per block behind an opaque zero: block-invariant values the compiler would otherwise keep in
    // registers - or spill - beside the 144 weight registers
    int lz = l31;
    asm volatile("" : "+v"(lz));
    int pixb[2], xb[3][2];
    xor_bases(tile_col8(lz), lz >> 4, half, xb);
#pragma unroll
    for (int r = 0; r < 2; ++r) pixb[r] = ((4 * rp + 2 * r + (lz >> 4)) * PWD + tile_col8(lz)) * 128;
    // the two row tiles one after the other, (tile, tap) steps software-pipelined (bf16_c64_tiles.hpp); folded BN of a tile's sums,
    // then through LDS: row = 32 x row tile + tile row, column = output channel
    conv_tiles<2, PWD>(patch, pixb, [&](int, int tx, int q) { return xb[tx][q]; }, wreg, [&](int r, const f32x16& acc) {
      C64_STAMP(4 + 2 * r);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * half;   // pixel of the row tile
        ex[(32 * (2 * rp + r) + i) * EXROW + 32 * ct + l31] = __builtin_fmaf(acc[e], sc, bi);
      }
      C64_STAMP(5 + 2 * r);
    });
    // the residual loads are older than the (at most six) patch DMA instructions of this wave: a counted wait leaves
    // those in flight
    if (!p.residual) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (has_next) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    C64_STAMP(8);
#pragma unroll
    for (int k = 0; k < 4; ++k) settle(res[k]);
    __builtin_amdgcn_s_barrier();
    C64_STAMP(9);
    // 128 pixels x 8 chunks of 8 channels: residual, ReLU, bf16, 16-byte stores
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int item = k * 256 + tid;
      const int px = item >> 3, c8 = (item & 7) * 8;
      const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
      if (yy < p.H && xx < p.W) {
        const size_t o = (((size_t)n * p.H + yy) * p.W + xx) * 64 + c8;
        const int prow = px >> 4, pcol = px & 15;
        const int er = 32 * (prow >> 1) + ((prow & 1) ? 16 + (pcol ^ 8) : pcol);   // the exchange row of this pixel
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&ex[er * EXROW + c8]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&ex[er * EXROW + c8 + 4]);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.residual) {
          const bf16x8 rr = __builtin_bit_cast(bf16x8, res[k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rr[e];
        }
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (__bf16)(p.relu ? fmaxf(v[e], 0.f) : v[e]);
        *reinterpret_cast<bf16x8*>(p.y + o) = h;
      }
    }
    C64_STAMP(10);
  }
#endif
}

}  // namespace

#if defined(C64_STAMPS)
void conv3x3_bf16_c64_dump_stamps() {
  static unsigned long long h[8 * 4 * 2 * 16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(c64_stamp_buf), sizeof(h)) != hipSuccess) return;
  for (int wg = 0; wg < 8; ++wg)
    for (int w = 0; w < 4; ++w)
      for (int it = 0; it < 2; ++it) {
        const unsigned long long* q = &h[((wg * 4 + w) * 2 + it) * 16];
        fprintf(stderr, "STAMP wg %d wave %d it %d:", wg, w, it);
        for (int k = 1; k <= 10; ++k) fprintf(stderr, " %lld", (long long)(q[k] - q[0]));
        if (it == 1) fprintf(stderr, "  | block period %lld", (long long)(q[0] - h[((wg * 4 + w) * 2) * 16]));
        fprintf(stderr, "\n");
      }
}
#endif

// [Cout 64][9][Cin 64] f32 -> bf16 MFMA B fragments [2 ct][9 taps][4 s][64 lanes][8]: element j of lane (n = l & 31,
// h = l >> 5) is w[32 ct + n][tap][16 s + 8 h + j]
std::vector<uint16_t> conv3x3_bf16_c64_fragments(const float* ohwi) {
  auto bf = [](float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  std::vector<uint16_t> fr((size_t)2 * 9 * 4 * 64 * 8);
  for (int ct = 0; ct < 2; ++ct)
    for (int t = 0; t < 9; ++t)
      for (int s = 0; s < 4; ++s)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j)
            fr[((((size_t)ct * 9 + t) * 4 + s) * 64 + l) * 8 + j] = bf(ohwi[((size_t)(32 * ct + (l & 31)) * 9 + t) * 64 + 16 * s + 8 * (l >> 5) + j]);
  return fr;
}

void launch_conv3x3_bf16_c64(const void* x, const void* wfrag, const float* scale, const float* bias, const void* residual, int relu,
                             void* y, int N, int H, int W, int num_cus, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0) fail(OCR_ERR_INVALID, "conv3x3_bf16_c64: bad shape N=%d H=%d W=%d", N, H, W);
  const long long bytes = (long long)N * H * W * 64 * 2;
  if (bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "conv3x3_bf16_c64: tensors of %lld bytes must be < 2^31; split the batch", bytes);
  C64Args a{};
  a.x = static_cast<const __bf16*>(x);
  a.wfrag = static_cast<const u32x4*>(wfrag);
  a.scale = scale;
  a.bias = bias;
  a.residual = static_cast<const __bf16*>(residual);
  a.y = static_cast<__bf16*>(y);
  a.x_bytes = (unsigned)bytes;
  a.H = H;
  a.W = W;
  a.bh = (H + 7) / 8;
  a.bw = (W + 15) / 16;
  a.relu = relu;
  const long long blocks = (long long)N * a.bh * a.bw;
  if (blocks >= (1ll << 31)) fail(OCR_ERR_INVALID, "conv3x3_bf16_c64: grid too large");
  a.nblocks = (int)blocks;
  const long long resident = 2ll * (num_cus > 0 ? num_cus : 256);
  const unsigned grid = blocks > resident ? (unsigned)resident : (unsigned)blocks;
  hipLaunchKernelGGL(conv3x3_bf16_c64_kernel, dim3(grid), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
#if defined(C64_STAMPS)
  static int launches = 0;
  if (++launches == 30) {
    (void)hipStreamSynchronize(s);
    conv3x3_bf16_c64_dump_stamps();
  }
#endif
}

}  // namespace ocr
