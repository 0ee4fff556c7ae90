// Fused Winograd F(4x4, 3x3) with its thirty-six element-wise GEMMs on the bf16 matrix cores (mfma=split_bf16): the 3x3 s1 p1
// convs of the large grids (layer1, layer2, the FPN's lateral terms; /root/reference/src/text_detection/model.rs:40-55,
// :126-133) with f32 operands split EXACTLY into three bf16 terms each and six of the nine partial products accumulated
// in f32 by v_mfma_f32_16x16x32_bf16 (the arithmetic of conv_igemm's X3 form).  Re-proportioned from
// winograd43_fused.hip for a matrix pipe that is 2.67 x faster and wants 6 bytes per operand element:
//
//   * the weights U (36 x C x K x 6 bytes) no longer fit the L2 -> CU path once per 16 x 16 pixel block (0.86 MB for
//     C = K = 64): a workgroup (8 waves) owns TWO pixel blocks (32 tiles) x 64 output channels and every B fragment fetched
//     from L2 feeds both blocks.  A wave owns 16 output channels of both blocks and HALF of the components - three of the six
//     rows i of M[i][j]: 18 x 2 accumulator tiles = 144 registers, two waves per SIMD;
//   * V = B^T d B is written to LDS ALREADY SPLIT (three bf16 planes in the MFMA's A-fragment order: one ds_read_b128 per
//     plane, tile and component), 12 components at a time (73.7 KB), channels in chunks of 32 = one MFMA's K;
//   * per chunk three phases { rows (0,5) | (1,2) | (3,4) of B^T d; the first row of a pair belongs to the waves of half 0 }:
//     transform (one (tile, channel pair) item per thread, packed f32 math on the pair, v_cvt_pk_bf16_f32 for the split)
//     -> barrier -> per wave 6 components x (6 A fragments from LDS, 3 B fragments from a three-deep register ring, 12 MFMAs)
//     -> barrier;
//   * B fragments stream through the ring all the time (two components ahead, also across the transform phases), the
//     next patch is requested right behind the last ring load of a chunk so that no ring wait ever has to force it;
//   * output transform: A^T along j in registers (every j of a row sits in one lane), then the two halves exchange their
//     three rows of the OTHER block's partial result through LDS (lane to lane, ds_write_b128 / ds_read_b128) and each
//     finishes A^T along i for one block; results staged through LDS so that folded BN, residual, ReLU and the stores
//     run on whole 256-byte pixel rows.
//
// LDS: 4 patch buffers (2 blocks x 2 halves of 16 channels) + V = 158 KB, one workgroup per CU.  Inline-asm loads with
// hand-counted s_waitcnt as in winograd43_fused.hip: every load's issue, wait and use sit in one straight-line chunk body.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <typename R>
__device__ __forceinline__ void load16(f32x4& v, R rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void settle(f32x4& v) { asm volatile("" : "+v"(v)::"memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct W43XArgs {
  const float* x;         // [N][H][W][C]
  const void* ufrag;      // winograd43_x3_fragments()
  const float* scale;     // folded BN, may be null
  const float* bias;
  const float* residual;  // [N][H][W][K], may be null
  float* y;               // [N][H][W][K]
  unsigned x_bytes, u_bytes, y_bytes;
  int H, W, bh, bw;       // block grid: bh x bw blocks of 16 x 16 pixels per image
  int C, K, kblocks;      // channels in / out, K / 64
  int relu;
  int npb;                // pixel blocks N * bh * bw
  int nunits;             // work units: ceil(npb / 2) pairs of pixel blocks x kblocks
  int xcd_chunks;         // 1: every XCD (blockIdx & 7) walks its own contiguous run of units
  int debug;              // -DW43_DEBUG builds: 1 skip B loads, 2 skip the transform, 4 skip patch DMA, 8 skip stores, 32 skip MFMAs
};

constexpr int PP = 18;                          // patch rows / columns
constexpr int PITCH = PP * 64 + 16;             // bytes per patch row (16 channels): tile rows 4 patch rows apart sit 64 B apart mod 256
constexpr int PATCH_STRIDE = 21120;             // >= 18 * PITCH, = 128 mod 256: the two channel halves of a pair item read different bank halves
constexpr int V_OFF = 4 * PATCH_STRIDE;         // V[12 components][2 blocks][3 planes][1 KB = 16 tiles x 32 channels bf16]
constexpr int V_BYTES = 12 * 2 * 3 * 1024;
constexpr int LDS_BYTES = V_OFF + V_BYTES;      // 158208 of 163840
constexpr int STAGE_BYTES = 128 * 64 * 4;       // half a result block (8 pixel rows x 16 x 64 channels f32)
constexpr unsigned OOB = 0x80000000u;
constexpr int RING = 3;                         // B ring: components in flight per wave (3 x 16 B per lane each); divides 18
static_assert(18 * PITCH <= PATCH_STRIDE && PATCH_STRIDE % 256 == 128, "patch buffer stride");
static_assert(2 * STAGE_BYTES <= V_BYTES && 8 * 6 * 1024 <= V_BYTES, "both blocks' staged halves and a round of the exchange fit the V space");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
#ifdef W43_DEBUG
#define W43X_DBG(p, bit) ((p).debug & (bit))
#else
#define W43X_DBG(p, bit) false
#endif

#ifdef W43_STAMPS
// diagnostic build only (make EXTRA=-DW43_STAMPS): s_memtime of workgroup 0, waves 0 and 4 (one of each half), at the phase
// boundaries of its first three units; kept in LDS during the kernel and copied out at the end (tools/w43x_stamps.py)
__device__ long long g_w43x_stamps[2 * 3 * 64];
#define W43X_STAMP(k)                                                                                          \
  do {                                                                                                         \
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && unit_count < 3)                                     \
      stamp_lds[((wave >> 2) * 3 + unit_count) * 64 + (k)] = (long long)__builtin_amdgcn_s_memtime();          \
  } while (0)
#else
#define W43X_STAMP(k) do {} while (0)
#endif

// rows of B^T d produced by phase ph: the pairs share their loads (rows 0 / 5 read patch rows 0, 2, 4 / 1, 3, 5; the others 1 .. 4)
__host__ __device__ constexpr int phase_row(int ph, int r) { return ph == 0 ? (r == 0 ? 0 : 5) : ph == 1 ? (r == 0 ? 1 : 2) : (r == 0 ? 3 : 4); }
// processing order o = 12 ph + 6 r + j  ->  component 6 i + j
__host__ __device__ constexpr int order_comp(int o) { return 6 * phase_row(o / 12, (o % 12) / 6) + o % 6; }

// B^T (x) for six values of two channels at once (same arithmetic as winograd43_fused.hip's bt6)
__device__ __forceinline__ void bt6x2(const f32x2 d0, const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4, const f32x2 d5,
                                      f32x2* t) {
  const f32x2 a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}
// A^T for two tiles at once (registers r, r + 1 of an accumulator)
__device__ __forceinline__ void at6x2(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                      f32x2* y) {
  const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}
// x = hi + mid + lo exactly (round to nearest even at every level; the remainders are exact in f32), the pair's three
// dwords to the three planes of V
__device__ __forceinline__ void split_store(const f32x2 x, unsigned char* dst) {
  const bf16x2 h = __builtin_convertvector(x, bf16x2);
  const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  *reinterpret_cast<bf16x2*>(dst) = h;
  *reinterpret_cast<bf16x2*>(dst + 1024) = m;
  *reinterpret_cast<bf16x2*>(dst + 2048) = l;
}

// NCH = C / 32 channel chunks; a workgroup produces 64 of the K output channels of two 16 x 16 pixel blocks
template <int NCH>
__global__ __launch_bounds__(512, 2) void winograd43_x3_kernel(W43XArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
#ifdef W43_STAMPS
  __shared__ long long stamp_lds[2 * 3 * 64];
  int unit_count = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, cg = wave & 3;   // rows of M this wave accumulates (first / second of a phase's pair), its 16 output channels
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.ufrag), 0, p.u_bytes, 0x00020000);
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.x), 0,
                                                        p.residual ? p.y_bytes : 0u, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;

  // unit -> (pair of pixel blocks, output-channel block); kb innermost: the K / 64 units of a pair share its patches in L2
  auto coords = [&](int pb, int& n_, int& y0_, int& x0_) {
    x0_ = 16 * (pb % p.bw);
    pb /= p.bw;
    y0_ = 16 * (pb % p.bh);
    n_ = pb / p.bh;
  };
  // every XCD (blockIdx & 7) walks ONE contiguous run of units (winograd43_fused.hip): placement is a speed matter only
  int first = 0, count = p.nunits, q = blockIdx.x, stride = gridDim.x;
  if (p.xcd_chunks) {
    const int j = blockIdx.x & 7, sb = p.nunits / p.kblocks, c = sb >> 3, rem = sb & 7;
    q = blockIdx.x >> 3;
    stride = gridDim.x >> 3;
    first = (j * c + min(j, rem)) * p.kblocks;
    count = (c + (j < rem ? 1 : 0)) * p.kblocks;
  }
  const int kb = q % p.kblocks;   // constant over a workgroup's units: stride is a multiple of kblocks

  // ---- the two pixel blocks of a unit.  Everything here is wave-uniform except the lane's patch column offset.
  struct Blocks {
    int n[2], y0[2], x0[2];
    bool ok[2];
  };
  auto locate = [&](int unit, Blocks& b) {
    const int pair = unit / p.kblocks;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int pb = 2 * pair + g;
      b.ok[g] = pb < p.npb;
      coords(min(pb, p.npb - 1), b.n[g], b.y0[g], b.x0[g]);
    }
  };
  // ---- patch DMA: per pixel block and channel half one patch row (18 px x 64 B) in two pieces - 16 pixels (1 KB) and 2 pixels
  // (lanes 0..7).  Waves 0-3 fetch block 0, waves 4-7 block 1; inside a block as winograd43_fused.hip: piece = wave & 1,
  // rows ((wave >> 1) & 1) + 2 m.  18 instructions per wave and chunk.
  const int dg = half, part = wave & 1, row0 = (wave >> 1) & 1;
  auto issue_patch = [&](const Blocks& b, int c) {
    const int bn = dg ? b.n[1] : b.n[0], by0 = dg ? b.y0[1] : b.y0[0], bx0 = dg ? b.x0[1] : b.x0[0];
    const bool bok = dg ? b.ok[1] : b.ok[0];
    const int xx = bx0 - 1 + 16 * part + (lane >> 2);
    const unsigned pv = (bok && (unsigned)xx < (unsigned)p.W) ? (unsigned)((xx * p.C + (lane & 3) * 4) * 4) : OOB;
    // (one exec-mask change around all eighteen, not one per instruction)
    if (W43X_DBG(p, 4) || (part != 0 && lane >= 8)) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int row = row0 + 2 * m;
        const int yy = by0 - 1 + row;
        const bool row_ok = (unsigned)yy < (unsigned)p.H;
        const unsigned soff = row_ok ? (unsigned)(((bn * p.H + yy) * p.W * p.C + (2 * c + hf) * 16) * 4) : 0u;
        const unsigned dst = lds0 + (unsigned)((2 * dg + hf) * PATCH_STRIDE + row * PITCH + part * 1024);
        // (the select sits right in front of its use: hoisted, the eighteen offsets live in registers - or scratch - all chunk long)
        unsigned voff = pv;
        asm volatile("" : "+v"(voff));
        voff = row_ok ? voff : OOB;
        dma16(x_rsrc, __builtin_amdgcn_readfirstlane(dst), voff, __builtin_amdgcn_readfirstlane(soff));
      }
  };
  // ---- B fragments: ring of RING components x three planes, one 16-byte load per lane and plane.  Component o (processing
  // order 12 ph + 6 half + j) of chunk c sits at ((kb NCH + c) 36 + o) x 3 planes x 4 KB, this wave's 16 channels at 1 KB x cg.
  f32x4 ring[RING][3];
  const unsigned b_voff = (unsigned)((cg * 64 + lane) * 16);
  const unsigned b_base = ((unsigned)(kb * NCH) * 36u + 6u * half) * 12288u;
  // t = 6 ph + j: this wave's t-th component of the chunk
  auto issue_b = [&](int c, int t, f32x4 (&dst)[3]) {
    const int o = 12 * (t / 6) + t % 6;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      if (W43X_DBG(p, 1)) asm volatile("s_nop 0" : "=v"(dst[pl]));
      else load16(dst[pl], u_rsrc, b_voff, b_base + (unsigned)(((c * 36 + o) * 3 + pl) * 4096));
    }
  };

  // transform item of this thread: pixel block tid >> 8, tile m = 4 tx + ty (tile column tx = wave & 3, tile row ty), channels
  // 2 cp, 2 cp + 1 of the chunk
  const int cp = tid & 15, t_ty = (tid >> 4) & 3, t_g = half;
  const int t_src = (2 * t_g + (cp >> 3)) * PATCH_STRIDE + 4 * t_ty * PITCH + 4 * cg * 64 + (cp & 7) * 8;
  // V plane of a (component, block): [channel group kg = ch >> 3][tile m ^ 2 kg][8 channels] bf16 - lane (m = lane & 15,
  // kg = lane >> 4) reads its A fragment with one ds_read_b128, the wave 1 KB contiguous per 16-lane group (conflict-free);
  // the transform's dword stores of a half-wave (2 tiles x 4 kg x 4 pairs) hit 32 different banks
  const int t_m = 4 * cg + t_ty;
  const int v_dst = V_OFF + t_g * 3072 + (cp >> 2) * 256 + ((t_m ^ (2 * (cp >> 2))) & 15) * 16 + (cp & 3) * 4;
  // this wave's components of a phase: slots 6 half .. 6 half + 5
  const unsigned char* a_ptr = lds + V_OFF + half * 6 * 6144 + (lane >> 4) * 256 + (((lane & 15) ^ (2 * (lane >> 4))) & 15) * 16;

  f32x4 acc[3][6][2];   // [phase = this half's row of the pair][j][pixel block]

  // ---- one phase of a chunk: two rows of B^T d B for both pixel blocks -> V, then their twelve components' MFMAs
  auto transform = [&](auto phc) {
    constexpr int PH = decltype(phc)::value;
    if (W43X_DBG(p, 2)) return;
    const unsigned char* pb = lds + t_src;
    f32x2 ta[6], tb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      auto d = [&](int r) { return *reinterpret_cast<const f32x2*>(pb + r * PITCH + j * 64); };
      if constexpr (PH == 0) {
        ta[j] = 4.f * d(0) - 5.f * d(2) + d(4);
        tb[j] = 4.f * d(1) - 5.f * d(3) + d(5);
      } else if constexpr (PH == 1) {
        const f32x2 a = d(4) - 4.f * d(2), b = d(3) - 4.f * d(1);
        ta[j] = a + b;
        tb[j] = a - b;
      } else {
        const f32x2 c = d(4) - d(2), e = 2.f * (d(3) - d(1));
        ta[j] = c + e;
        tb[j] = c - e;
      }
    }
    f32x2 o[6];
    bt6x2(ta[0], ta[1], ta[2], ta[3], ta[4], ta[5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) split_store(o[j], lds + v_dst + j * 6144);
    bt6x2(tb[0], tb[1], tb[2], tb[3], tb[4], tb[5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) split_store(o[j], lds + v_dst + (6 + j) * 6144);
  };

  auto mfma_phase = [&](auto phc, const int c, const bool has_patch, const Blocks& nb, const int nc) {
    constexpr int PH = decltype(phc)::value;
    bf16x8 acur[2][3], anext[2][3];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        if (W43X_DBG(p, 128)) asm volatile("" : "=v"(acur[g][pl]));
        else acur[g][pl] = *reinterpret_cast<const bf16x8*>(a_ptr + (g * 3 + pl) * 1024);
      }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int t = 6 * PH + j;
      // ring protocol: at step t the loads of component t + RING - 1 go out (into the registers component t - 1 has just left),
      // then everything but the RING - 1 youngest components must have landed.  Behind the last ring load of the chunk
      // (step 18 - RING) the next patch is requested: 18 more loads younger than everything still awaited.
      if (t + RING - 1 < 18) issue_b(c, t + RING - 1, ring[(t + RING - 1) % RING]);
#ifdef W43_SAFE
      wait_vm<0>();
      if (t == 18 - RING && has_patch) issue_patch(nb, nc);
#else
      if (t <= 18 - RING) {
        wait_vm<3 * (RING - 1)>();
        if (t == 18 - RING && has_patch) issue_patch(nb, nc);
      } else if (has_patch) {
        if (17 - t == 1) wait_vm<3 + 18>();
        else wait_vm<18>();
      } else {
        if (17 - t == 1) wait_vm<3>();
        else wait_vm<0>();
      }
#endif
      f32x4(&b)[3] = ring[t % RING];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) settle(b[pl]);
      if (j < 5) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            if (W43X_DBG(p, 128)) asm volatile("" : "=v"(anext[g][pl]));
            else anext[g][pl] = *reinterpret_cast<const bf16x8*>(a_ptr + (((j + 1) * 2 + g) * 3 + pl) * 1024);
          }
      }
      if (!W43X_DBG(p, 32)) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]), bl = __builtin_bit_cast(bf16x8, b[2]);
        // six of the nine partial products, small terms first (mid.lo, lo.mid, lo.lo are below 2^-23 of the product); the two
        // pixel blocks' chains alternate
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][2], bh, acc[PH][j][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][0], bl, acc[PH][j][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][1], bm, acc[PH][j][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][1], bh, acc[PH][j][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][0], bm, acc[PH][j][g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[PH][j][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[g][0], bh, acc[PH][j][g], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) acur[g][pl] = anext[g][pl];
    }
  };

  // ---- output transform, first step: R[ri][b] = sum_j M[row ri][j] A[j][b] of pixel block G, every register (tile row) at once
  auto row_transform = [&](auto gc, f32x4 (&R)[3][4]) {
    constexpr int G = decltype(gc)::value;
#pragma unroll
    for (int ri = 0; ri < 3; ++ri) {
      f32x2 lo[4], hi[4];
      at6x2(acc[ri][0][G].xy, acc[ri][1][G].xy, acc[ri][2][G].xy, acc[ri][3][G].xy, acc[ri][4][G].xy, acc[ri][5][G].xy, lo);
      at6x2(acc[ri][0][G].zw, acc[ri][1][G].zw, acc[ri][2][G].zw, acc[ri][3][G].zw, acc[ri][4][G].zw, acc[ri][5][G].zw, hi);
#pragma unroll
      for (int b = 0; b < 4; ++b) R[ri][b] = f32x4{lo[b].x, lo[b].y, hi[b].x, hi[b].y};
    }
  };
  // exchange area (inside the V space): [output-channel group][sending half][6 values of a round][lane] x 16 B; a lane's partner
  // in the other half holds the same (tile column, output channel)
  const int x_wr = V_OFF + ((cg * 2 + half) * 6) * 1024 + lane * 16;
  const int x_rd = V_OFF + ((cg * 2 + (half ^ 1)) * 6) * 1024 + lane * 16;

  bool patch_in_flight = false;  // chunk 0 of this unit's patches was requested during the previous unit
  Blocks cur, nxt;
  if (q < count) locate(first + q, cur);
  nxt = cur;
  for (int lu = q; lu < count; lu += stride) {
    const bool has_next_unit = lu + stride < count;
    W43X_STAMP(0);
    if (has_next_unit) locate(first + lu + stride, nxt);
    if (!patch_in_flight) issue_patch(cur, 0);
    patch_in_flight = false;

#pragma unroll
    for (int ri = 0; ri < 3; ++ri)
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[ri][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < NCH; ++c) {
      const bool last_chunk = c + 1 == NCH;
      const bool has_patch = !last_chunk || has_next_unit;
      // the B stream of the chunk starts here and drains inside it (no inline-asm load is in flight across the back edge);
      // with its first RING - 1 components requested, everything older - this wave's share of the chunk's patches - has landed
#pragma unroll
      for (int t = 0; t < RING - 1; ++t) issue_b(c, t, ring[t]);
      W43X_STAMP(1 + 14 * c);
      wait_vm<3 * (RING - 1)>();
      W43X_STAMP(2 + 14 * c);
      __syncthreads();   // every wave's share; and nobody still reads the V of the previous chunk
      W43X_STAMP(3 + 14 * c);
      transform(std::integral_constant<int, 0>{});
      W43X_STAMP(4 + 14 * c);
      __syncthreads();
      W43X_STAMP(5 + 14 * c);
      mfma_phase(std::integral_constant<int, 0>{}, c, false, cur, 0);
      W43X_STAMP(6 + 14 * c);
      __syncthreads();
      W43X_STAMP(7 + 14 * c);
      transform(std::integral_constant<int, 1>{});
      W43X_STAMP(8 + 14 * c);
      __syncthreads();
      W43X_STAMP(9 + 14 * c);
      mfma_phase(std::integral_constant<int, 1>{}, c, false, cur, 0);
      W43X_STAMP(10 + 14 * c);
      __syncthreads();
      W43X_STAMP(11 + 14 * c);
      transform(std::integral_constant<int, 2>{});
      W43X_STAMP(12 + 14 * c);
      __syncthreads();   // the patches are free from here on
      W43X_STAMP(13 + 14 * c);
      // the next patch: this unit's next chunk, or chunk 0 of the next unit
      Blocks tgt = cur;
      if (last_chunk) tgt = nxt;
      mfma_phase(std::integral_constant<int, 2>{}, c, has_patch, tgt, last_chunk ? 0 : c + 1);
      W43X_STAMP(14 + 14 * c);
    }
    patch_in_flight = has_next_unit;

    // ---- output transform.  M's rows are split over the two halves (half 0: rows 0, 1, 3; half 1: rows 5, 2, 4).  Every wave
    // applies A^T along j to its rows of BOTH blocks, hands the other block's three partial rows to its partner (same lane, other
    // half) and finishes block `half`: A^T along i over its own three rows and the partner's.
    {
      f32x4 Y[4][4];   // [a][b], register = tile row ty
      auto finish = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        f32x4 Rs[3][4], Ro[3][4];   // partial rows of the partner's block (sent) and of this wave's own
        row_transform(std::integral_constant<int, 1 - HH>{}, Rs);
        __syncthreads();   // every wave is done with V
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          if (rd) __syncthreads();   // round 0 has been read
#pragma unroll
          for (int ri = 0; ri < 3; ++ri)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) *reinterpret_cast<f32x4*>(lds + x_wr + (ri * 2 + bb) * 1024) = Rs[ri][2 * rd + bb];
          __syncthreads();
          if (rd == 0) row_transform(std::integral_constant<int, HH>{}, Ro);
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int b = 2 * rd + bb;
            f32x4 m[6];
#pragma unroll
            for (int ri = 0; ri < 3; ++ri) {
              m[phase_row(ri, HH)] = Ro[ri][b];
              m[phase_row(ri, 1 - HH)] = *reinterpret_cast<const f32x4*>(lds + x_rd + (ri * 2 + bb) * 1024);
            }
            f32x2 lo[4], hi[4];
            at6x2(m[0].xy, m[1].xy, m[2].xy, m[3].xy, m[4].xy, m[5].xy, lo);
            at6x2(m[0].zw, m[1].zw, m[2].zw, m[3].zw, m[4].zw, m[5].zw, hi);
#pragma unroll
            for (int a = 0; a < 4; ++a) Y[a][b] = f32x4{lo[a].x, lo[a].y, hi[a].x, hi[a].y};
          }
        }
      };
      W43X_STAMP(57);
      if (W43X_DBG(p, 256)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) Y[a][b] = acc[a % 3][b][0] + acc[b % 3][a][1];
      } else if (half == 0) finish(std::integral_constant<int, 0>{});
      else finish(std::integral_constant<int, 1>{});

      W43X_STAMP(58);
      // Y -> pixel rows: lane = output channel 16 cg + (lane & 15), tile column tx = lane >> 4, register = tile row ty; both blocks'
      // halves of 8 pixel rows staged through the V space (block `half` by this wave), then row-major epilogue by all threads
      int etid = tid;                          // (made opaque: everything derived from it here is computed here, not carried through the unit)
      asm volatile("" : "+v"(etid));
      const int elane = etid & 63;
      const int tx = elane >> 4;
      const int col = ((cg * 16 + (elane & 15)) + 16 * tx) & 63;  // rotated by the tile column: the four tiles of a store hit different banks
      const int c4 = (etid & 15) * 4;
      const int eg = half;                     // pixel block of this thread's epilogue rows
      const int et = etid & 255;               // ... and its place in it: pixel column et >> 4, channels c4 .. c4 + 3
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + kb * 64 + c4);
      if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + kb * 64 + c4);
      const int en = eg ? cur.n[1] : cur.n[0], ey0 = eg ? cur.y0[1] : cur.y0[0], ex0 = eg ? cur.x0[1] : cur.x0[0];
      const bool eok = eg ? cur.ok[1] : cur.ok[0];
      const int ep_xx = ex0 + (et >> 4);
      const unsigned ep_v = (eok && ep_xx < p.W) ? (unsigned)((ep_xx * p.K + kb * 64 + c4) * 4) : OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        auto row_offset = [&](int k) -> unsigned { return (unsigned)(((en * p.H + min(ey0 + 8 * h + k, p.H - 1)) * p.W * p.K) * 4); };
        auto row_voff = [&](int k) -> unsigned { return ey0 + 8 * h + k < p.H ? ep_v : OOB; };
        f32x4 res[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          res[k] = W43X_DBG(p, 64) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                   : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, row_voff(k), __builtin_amdgcn_readfirstlane(row_offset(k)), 0));
        W43X_STAMP(59 + 2 * h);
        __syncthreads();  // h = 0: the exchange area has been read; h = 1: the first halves have been read
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const f32x2 o = h == 0 ? Y[a][b].xy : Y[a][b].zw;
            *reinterpret_cast<float*>(lds + V_OFF + eg * STAGE_BYTES + ((a * 16 + 4 * tx + b) * 64 + col) * 4) = o.x;
            *reinterpret_cast<float*>(lds + V_OFF + eg * STAGE_BYTES + (((4 + a) * 16 + 4 * tx + b) * 64 + col) * 4) = o.y;
          }
        __syncthreads();
        f32x4 vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int pl = k * 16 + (et >> 4);
          vr[k] = *reinterpret_cast<const f32x4*>(lds + V_OFF + eg * STAGE_BYTES + (pl * 64 + ((c4 + 16 * (et >> 6)) & 63)) * 4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) vr[k] = vr[k] * sc + bi + res[k];
        if (p.relu) {   // one uniform branch per half
#pragma unroll
          for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) vr[k][e] = fmaxf(vr[k][e], 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (!W43X_DBG(p, 8))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vr[k]), y_rsrc, row_voff(k), __builtin_amdgcn_readfirstlane(row_offset(k)), 0);
      }
    }
    cur = nxt;
    W43X_STAMP(63);
#ifdef W43_STAMPS
    ++unit_count;
#endif
  }
#ifdef W43_STAMPS
  __syncthreads();
  if (blockIdx.x == 0 && tid < 2 * 3 * 64) g_w43x_stamps[tid] = stamp_lds[tid];
#endif
#endif
}

// f32 -> bf16, round to nearest even (weights are finite)
inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace

// u: winograd_weights(..., 4) = [36][Cout][Cin] -> [Cout / 64][Cin / 32][36 components in processing order][3 planes hi / mid /
// lo][wave 4][lane 64][8 bf16]: element j of lane l of wave w is U[comp][cout = 64 kb + 16 w + (l & 15)][cin = 32 c + 8 (l >> 4) + j],
// the B fragment of v_mfma_f32_16x16x32_bf16 (k = 8 (l >> 4) + j, the order the A fragments read from V have)
std::vector<uint16_t> winograd43_x3_fragments(const std::vector<float>& u, int cout, int cin) {
  if (cout % 64 || cin % 32 || u.size() != (size_t)36 * cout * cin) fail(OCR_ERR_INTERNAL, "winograd43_x3_fragments: bad shape");
  const int nch = cin / 32;
  std::vector<uint16_t> f(3 * u.size());
  size_t o = 0;
  for (int kb = 0; kb < cout / 64; ++kb)
    for (int c = 0; c < nch; ++c)
      for (int ord = 0; ord < 36; ++ord) {
        const int comp = order_comp(ord);
        for (int w = 0; w < 4; ++w)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const float x = u[((size_t)comp * cout + kb * 64 + 16 * w + (l & 15)) * cin + 32 * c + 8 * (l >> 4) + j];
              const uint16_t h = bf16_rne(x);
              const float r1 = x - bf16_f32(h);      // exact
              const uint16_t m = bf16_rne(r1);
              const float r2 = r1 - bf16_f32(m);     // exact
              const size_t at = o + ((size_t)w * 64 + l) * 8 + j;
              f[at] = h;
              f[at + 2048] = m;
              f[at + 4096] = bf16_rne(r2);
            }
        o += 3 * 2048;
      }
  return f;
}

void winograd43_set_debug(int d);
int winograd43_get_debug();
#ifdef W43_STAMPS
void winograd43_x3_read_stamps(long long* out) { OCR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w43x_stamps), sizeof(g_w43x_stamps))); }
#endif

void launch_winograd43_x3(const float* x, const void* ufrag, const float* scale, const float* bias, const float* residual,
                          int relu, float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || (C != 64 && C != 128 && C != 256) || K % 64)
    fail(OCR_ERR_INVALID, "fused Winograd F(4x4) x3: bad shape N=%d H=%d W=%d C=%d K=%d", N, H, W, C, K);
  const long long xb = (long long)N * H * W * C * 4, ub = (long long)36 * C * K * 6;
  if (xb >= (1ll << 31) || (long long)N * H * W * K * 4 >= (1ll << 31)) fail(OCR_ERR_INVALID, "fused Winograd F(4x4) x3: tensor too large");
  W43XArgs a{};
  a.x = x;
  a.ufrag = ufrag;
  a.scale = scale;
  a.bias = bias;
  a.residual = residual;
  a.y = y;
  a.x_bytes = (unsigned)xb;
  a.y_bytes = (unsigned)((long long)N * H * W * K * 4);
  a.u_bytes = (unsigned)ub;
  a.H = H;
  a.W = W;
  a.bh = (H + 15) / 16;
  a.bw = (W + 15) / 16;
  a.C = C;
  a.K = K;
  a.kblocks = K / 64;
  a.relu = relu;
  const long long npb = (long long)N * a.bh * a.bw;
  const long long units = (npb + 1) / 2 * a.kblocks;
  if (npb >= (1ll << 30)) fail(OCR_ERR_INVALID, "fused Winograd F(4x4) x3: too many blocks");
  a.npb = (int)npb;
  a.nunits = (int)units;
  a.debug = winograd43_get_debug();
  // persistent workgroups, one per CU; a multiple of kblocks so that each keeps its output-channel block, and of 8 kblocks
  // (when there are that many units) so that the eight XCDs can each walk a contiguous run
  long long grid = std::min<long long>(units, (num_cus > 0 ? num_cus : 256));
  const long long unit = 8ll * a.kblocks;
  if (grid >= unit && !(a.debug & 16)) {
    grid = grid / unit * unit;
    a.xcd_chunks = 1;
  } else {
    grid = std::max<long long>(a.kblocks, grid / a.kblocks * a.kblocks);
  }
  if (C == 64) hipLaunchKernelGGL(winograd43_x3_kernel<2>, dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (C == 128) hipLaunchKernelGGL(winograd43_x3_kernel<4>, dim3((unsigned)grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(winograd43_x3_kernel<8>, dim3((unsigned)grid), dim3(512), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
