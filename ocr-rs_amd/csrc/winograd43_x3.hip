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
typedef float f32x16 __attribute__((ext_vector_type(16)));
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
// the same with a count that is constant only after unrolling
__device__ __forceinline__ void wait_vm_n(int n) {
  switch (n) {
    case 0: wait_vm<0>(); break;
    case 3: wait_vm<3>(); break;
    case 6: wait_vm<6>(); break;
    case 9: wait_vm<9>(); break;
    default: wait_vm<0>(); break;
  }
}

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
constexpr int PATCH_STRIDE = 21120;             // >= 18 * PITCH; four patch buffers: [chunk parity][pixel block]
constexpr int V_OFF = 4 * PATCH_STRIDE;         // V[12 components][3 planes][1 KB = 32 tiles x 16 channels bf16]
constexpr int V_BYTES = 12 * 3 * 1024;
constexpr int E_OFF = V_OFF;                    // the epilogue's exchange / staging area: V and the space behind it
constexpr int E_BYTES = 8 * 9216;               // a round of the first exchange: nine 1 KB vectors per wave
constexpr int LDS_BYTES = E_OFF + E_BYTES;      // 158208 of 163840
constexpr int STAGE_BYTES = 128 * 64 * 4;       // half a result block (8 pixel rows x 16 x 64 channels f32)
constexpr unsigned OOB = 0x80000000u;
constexpr int RING = 3;                         // B ring: components in flight per wave (3 x 16 B per lane each); divides 9
static_assert(18 * PITCH <= PATCH_STRIDE && PATCH_STRIDE % 16 == 0, "patch buffer stride");
static_assert(V_BYTES <= E_BYTES && 2 * STAGE_BYTES <= E_BYTES && 8 * 6144 <= E_BYTES, "V, both blocks' staged halves and the exchanges fit");
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

// NCH = C / 16 channel chunks; a workgroup produces 64 of the K output channels of two 16 x 16 pixel blocks
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
  // wave = (g, ch, hj): the row of a phase's pair it accumulates, its 32 output channels, its three of the six columns j
  const int g = wave >> 2, ch = (wave >> 1) & 1, hj = wave & 1;
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

  // ---- the two pixel blocks of a unit (wave-uniform)
  struct Blocks {
    int n[2], y0[2], x0[2];
    bool ok[2];
  };
  auto locate = [&](int unit, Blocks& b) {
    const int pair = unit / p.kblocks;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pb = 2 * pair + k;
      b.ok[k] = pb < p.npb;
      coords(min(pb, p.npb - 1), b.n[k], b.y0[k], b.x0[k]);
    }
  };
  // ---- patch DMA: per pixel block and chunk of 16 channels, one patch row (18 px x 64 B) in two pieces - 16 pixels (1 KB) and 2
  // pixels (lanes 0..7).  Waves 0-3 fetch block 0, waves 4-7 block 1; piece = wave & 1, rows ((wave >> 1) & 1) + 2 m: nine
  // instructions per wave and chunk, spread over the matrix steps of the chunk BEFORE (double-buffered patches).
  const int part = wave & 1, row0 = (wave >> 1) & 1;
  struct PatchSrc {   // this wave's share of one chunk's patches
    int n, y0, c;
    unsigned pv;      // this lane's column offset inside an image row of x (out of range = reads zero)
    bool on;
  };
  auto patch_src = [&](const Blocks& b, int c, bool on) {
    PatchSrc s;
    s.n = g ? b.n[1] : b.n[0];
    s.y0 = g ? b.y0[1] : b.y0[0];
    const int bx0 = g ? b.x0[1] : b.x0[0];
    const bool bok = g ? b.ok[1] : b.ok[0];
    const int xx = bx0 - 1 + 16 * part + (lane >> 2);
    s.pv = (bok && (unsigned)xx < (unsigned)p.W) ? (unsigned)((xx * p.C + (lane & 3) * 4) * 4) : OOB;
    s.c = c;
    s.on = on;
    return s;
  };
  auto issue_piece = [&](const PatchSrc& s, int m) {
    if (W43X_DBG(p, 4) || !s.on || (part != 0 && lane >= 8)) return;
    const int row = row0 + 2 * m;
    const int yy = s.y0 - 1 + row;
    const bool row_ok = (unsigned)yy < (unsigned)p.H;
    const unsigned soff = row_ok ? (unsigned)(((s.n * p.H + yy) * p.W * p.C + s.c * 16) * 4) : 0u;
    const unsigned dst = lds0 + (unsigned)((2 * (s.c & 1) + g) * PATCH_STRIDE + row * PITCH + part * 1024);
    unsigned voff = s.pv;
    asm volatile("" : "+v"(voff));   // (keeps the select next to its use)
    voff = row_ok ? voff : OOB;
    dma16(x_rsrc, __builtin_amdgcn_readfirstlane(dst), voff, __builtin_amdgcn_readfirstlane(soff));
  };
  // ---- B fragments: ring of RING components x three planes, one 16-byte load per lane and plane.  Component o (processing
  // order 12 ph + 6 g + 3 hj + jj) of chunk c sits at ((kb NCH + c) 36 + o) x 6 KB: [plane][ch][lane] x 16 B
  f32x4 ring[RING][3];
  const unsigned b_voff = (unsigned)((ch * 64 + lane) * 16);
  const unsigned b_base = ((unsigned)(kb * NCH) * 36u + 6u * g + 3u * hj) * 6144u;
  // t = 3 ph + jj: this wave's t-th component of the chunk
  auto issue_b = [&](int c, int t, f32x4 (&dst)[3]) {
    const int o = 12 * (t / 3) + t % 3;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      if (W43X_DBG(p, 1)) asm volatile("s_nop 0" : "=v"(dst[pl]));
      else load16(dst[pl], u_rsrc, b_voff, b_base + (unsigned)((c * 36 + o) * 6144 + pl * 2048));
    }
  };

  // transform item of this thread: row g of the phase's pair; tile t32 = (tid & 255) >> 3 of the 32 (block t32 >> 4, tile m = t32 & 15 =
  // 4 tx + ty), channels 2 cp, 2 cp + 1 of the chunk.  A quarter-wave's four tiles sit four patch rows apart: conflict-free ds_read_b64.
  const int cp = tid & 7, t32 = (tid & 255) >> 3;
  const int t_src = (t32 >> 4) * PATCH_STRIDE + 4 * (t32 & 3) * PITCH + 4 * ((t32 >> 2) & 3) * 64 + cp * 8;
  // V plane of a component: the A operand of v_mfma_f32_32x32x16_bf16, [channel group kg = ch >> 3][tile 32][8 channels] bf16 = 1 KB that
  // the 64 lanes read with one ds_read_b128 (lane = 32 kg + tile: lane-linear)
  const int v_dst = V_OFF + g * 6 * 3072 + (cp >> 2) * 512 + t32 * 16 + (cp & 3) * 4;
  const unsigned char* a_ptr = lds + V_OFF + (6 * g + 3 * hj) * 3072 + lane * 16;
  const float sgn = g ? -1.f : 1.f;

  f32x16 acc[3][3];   // [phase = this wave's row of the pair][jj]; rows = the 32 tiles, columns = 32 output channels

  // ---- one phase of a chunk: the two rows of B^T d B (one per half of the workgroup) -> V, then their twelve components' MFMAs
  auto transform = [&](auto phc, int c, const PatchSrc& nps) {
    constexpr int PH = decltype(phc)::value;
    // the next chunk's patches: three DMA pieces per wave and phase, requested HERE - in front of the loads of the phase, where a full
    // request queue stalls nobody's MFMAs - and two matrix steps old when the third step's wait forces them
    issue_piece(nps, 3 * PH);
    issue_piece(nps, 3 * PH + 1);
    issue_piece(nps, 3 * PH + 2);
    if (W43X_DBG(p, 2)) return;
    const unsigned char* pb = lds + (c & 1) * 2 * PATCH_STRIDE + t_src;
    f32x2 t[6];
    if constexpr (PH == 0) {        // rows 0 / 5: 4 d0 - 5 d2 + d4 one patch row further down for the second
      const unsigned char* pr = pb + g * PITCH;
      f32x2 d[3][6];
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r) d[r][j] = *reinterpret_cast<const f32x2*>(pr + 2 * r * PITCH + j * 64);
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = 4.f * d[0][j] - 5.f * d[1][j] + d[2][j];
    } else {                         // rows 1 / 2: (d4 - 4 d2) +- (d3 - 4 d1); rows 3 / 4: (d4 - d2) +- 2 (d3 - d1)
      f32x2 d[4][6];
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r][j] = *reinterpret_cast<const f32x2*>(pb + (r + 1) * PITCH + j * 64);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if constexpr (PH == 1) {
          const f32x2 a = d[3][j] - 4.f * d[1][j], b = d[2][j] - 4.f * d[0][j];
          t[j] = a + sgn * b;
        } else {
          const f32x2 cc = d[3][j] - d[1][j], e = 2.f * (d[2][j] - d[0][j]);
          t[j] = cc + sgn * e;
        }
      }
    }
    f32x2 o[6];
    bt6x2(t[0], t[1], t[2], t[3], t[4], t[5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) split_store(o[j], lds + v_dst + j * 3072);
  };

  // DMA pieces per matrix step of a chunk (the next chunk's patches): early, so that the chunk's last pieces are a transform and
  // two steps old when the next chunk's first wait forces them
  auto mfma_phase = [&](auto phc, const int c, const PatchSrc& nps) {
    constexpr int PH = decltype(phc)::value;
    bf16x8 acur[3], anext[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      if (W43X_DBG(p, 128)) asm volatile("" : "=v"(acur[pl]));
      else acur[pl] = *reinterpret_cast<const bf16x8*>(a_ptr + pl * 1024);
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int t = 3 * PH + jj;
      // ring protocol: at step t the loads of component t + 2 go out, then everything but what was issued after component t's loads
      // must have landed: components t + 1, t + 2 and, in the first two steps of a phase, the three DMA pieces of its transform
      if (t + RING - 1 < 9) issue_b(c, t + RING - 1, ring[(t + RING - 1) % RING]);
#ifdef W43_SAFE
      wait_vm<0>();
#else
      if (nps.on) {
        constexpr int NW[9] = {9, 9, 6, 9, 9, 6, 9, 6, 0};
        wait_vm_n(NW[t]);
      } else {
        constexpr int NW0[9] = {6, 6, 6, 6, 6, 6, 6, 3, 0};
        wait_vm_n(NW0[t]);
      }
#endif
      f32x4(&b)[3] = ring[t % RING];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) settle(b[pl]);
      if (jj < 2) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          if (W43X_DBG(p, 128)) asm volatile("" : "=v"(anext[pl]));
          else anext[pl] = *reinterpret_cast<const bf16x8*>(a_ptr + ((jj + 1) * 3 + pl) * 1024);
        }
      }
      if (!W43X_DBG(p, 32)) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]), bl = __builtin_bit_cast(bf16x8, b[2]);
        // six of the nine partial products, small terms first (mid.lo, lo.mid, lo.lo are below 2^-23 of the product)
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], bh, acc[PH][jj], 0, 0, 0);
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[0], bl, acc[PH][jj], 0, 0, 0);
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], bm, acc[PH][jj], 0, 0, 0);
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], bh, acc[PH][jj], 0, 0, 0);
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[0], bm, acc[PH][jj], 0, 0, 0);
        acc[PH][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[0], bh, acc[PH][jj], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) acur[pl] = anext[pl];
    }
  };

  bool patch_in_flight = false;  // chunk 0 of this unit's patches was requested during the previous unit
  Blocks cur, nxt;
  if (q < count) locate(first + q, cur);
  nxt = cur;
  for (int lu = q; lu < count; lu += stride) {
    W43X_STAMP(0);
    const bool has_next_unit = lu + stride < count;
    if (has_next_unit) locate(first + lu + stride, nxt);
    if (!patch_in_flight) {
      const PatchSrc s0 = patch_src(cur, 0, true);
#pragma unroll
      for (int m = 0; m < 9; ++m) issue_piece(s0, m);
    }
    patch_in_flight = false;

#pragma unroll
    for (int ph = 0; ph < 3; ++ph)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ph][jj][e] = 0.f;

    for (int c = 0; c < NCH; ++c) {
      const bool last_chunk = c + 1 == NCH;
      // the patches requested during this chunk: the unit's next chunk, or chunk 0 of the next unit
      Blocks tgt = cur;
      if (last_chunk) tgt = nxt;
      const PatchSrc nps = patch_src(tgt, last_chunk ? 0 : c + 1, !last_chunk || has_next_unit);
      // the B stream of the chunk starts here and drains inside it (no inline-asm load is in flight across the back edge);
      // with its first RING - 1 components requested, everything older - this wave's share of the chunk's patches - has landed
#pragma unroll
      for (int t = 0; t < RING - 1; ++t) issue_b(c, t, ring[t]);
      W43X_STAMP(1 + 14 * (c & 3));
      wait_vm<3 * (RING - 1)>();
      W43X_STAMP(2 + 14 * (c & 3));
      __syncthreads();   // every wave's share; and nobody still reads the V of the previous chunk
      W43X_STAMP(3 + 14 * (c & 3));
      transform(std::integral_constant<int, 0>{}, c, nps);
      W43X_STAMP(4 + 14 * (c & 3));
      __syncthreads();
      W43X_STAMP(5 + 14 * (c & 3));
      mfma_phase(std::integral_constant<int, 0>{}, c, nps);
      W43X_STAMP(6 + 14 * (c & 3));
      __syncthreads();
      W43X_STAMP(7 + 14 * (c & 3));
      transform(std::integral_constant<int, 1>{}, c, nps);
      W43X_STAMP(8 + 14 * (c & 3));
      __syncthreads();
      W43X_STAMP(9 + 14 * (c & 3));
      mfma_phase(std::integral_constant<int, 1>{}, c, nps);
      W43X_STAMP(10 + 14 * (c & 3));
      __syncthreads();
      W43X_STAMP(11 + 14 * (c & 3));
      transform(std::integral_constant<int, 2>{}, c, nps);
      W43X_STAMP(12 + 14 * (c & 3));
      __syncthreads();
      W43X_STAMP(13 + 14 * (c & 3));
      mfma_phase(std::integral_constant<int, 2>{}, c, nps);
      W43X_STAMP(14 + 14 * (c & 3));
    }
    patch_in_flight = has_next_unit;

    // ---- output transform.  A wave holds M[i][j] for three rows i (its phases) x three columns j of 32 tiles x 32 output channels;
    // Y = A^T M A needs all 36.  Two exchanges through LDS, lane to lane (partners hold the same (tile, channel) per lane and
    // register): (1) over the column halves - the partner's three partial sums along j for half of the tile quads (register
    // quads; a wave finishes R[i][b] = sum_j M[i][j] A[j][b] for two quads), (2) over the row halves - the partner's R of one quad
    // (a wave finishes Y for one quad = 8 tiles x 32 channels).  Quad Q = 2 hj + g: pixel block hj, tile columns 2 g, 2 g + 1.
    W43X_STAMP(57);
    // ---- epilogue addressing (this thread's pixel column and channels of the staged result rows).  (Requesting the first half's
    // residual here, in front of the exchanges, costs 12 spilled registers and buys nothing: the unit is the SUM of its HBM
    // streams and its arithmetic - eight waves in lockstep have nothing to run while a request queue is full.)
    int etid = tid;                          // (made opaque: everything derived from it here is computed here, not carried through the unit)
    asm volatile("" : "+v"(etid));
    const int c4 = (etid & 15) * 4;
    const int eg = etid >> 8;                // pixel block of this thread's epilogue rows
    const int et = etid & 255;               // ... and its place in it: pixel column et >> 4, channels c4 .. c4 + 3
    const int en = eg ? cur.n[1] : cur.n[0], ey0 = eg ? cur.y0[1] : cur.y0[0], ex0 = eg ? cur.x0[1] : cur.x0[0];
    const bool eok = eg ? cur.ok[1] : cur.ok[0];
    const int ep_xx = ex0 + (et >> 4);
    const unsigned ep_v = (eok && ep_xx < p.W) ? (unsigned)((ep_xx * p.K + kb * 64 + c4) * 4) : OOB;
    auto row_offset = [&](int h, int k) -> unsigned { return (unsigned)(((en * p.H + min(ey0 + 8 * h + k, p.H - 1)) * p.W * p.K) * 4); };
    auto row_voff = [&](int h, int k) -> unsigned { return ey0 + 8 * h + k < p.H ? ep_v : OOB; };
    auto load_residual = [&](int h, f32x4 (&res)[8]) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        res[k] = W43X_DBG(p, 64) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                 : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, row_voff(h, k), __builtin_amdgcn_readfirstlane(row_offset(h, k)), 0));
    };
    f32x4 res0[8], res1[8];
    f32x4 Y[4][4];   // [a][b], element = tile row ty
    {
      auto quad = [](const f32x16& v, int qd) { return f32x4{v[4 * qd], v[4 * qd + 1], v[4 * qd + 2], v[4 * qd + 3]}; };
      auto pick = [](bool c, const f32x4 a, const f32x4 b) { return c ? a : b; };
      const bool H = hj != 0, G = g != 0;
      const int e1_wr = E_OFF + wave * 9216 + lane * 16, e1_rd = E_OFF + (wave ^ 1) * 9216 + lane * 16;
      f32x4 R[2][3][4];   // [final quad 2 hj + k][phase][b]
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f32x4 mine[3][3];   // partial sums along this wave's three columns, of the quad it finishes
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          f32x4 xm[3], xs[3];
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) {
            xm[jj] = pick(H, quad(acc[ph][jj], 2 + k), quad(acc[ph][jj], k));
            xs[jj] = pick(H, quad(acc[ph][jj], k), quad(acc[ph][jj], 2 + k));
          }
          // columns 0..2 (hj = 0): m0, m1 + m2, m1 - m2; columns 3..5 (hj = 1): m3 + m4, m3 - m4, m5
          mine[ph][0] = pick(H, xm[0] + xm[1], xm[0]);
          mine[ph][1] = pick(H, xm[0] - xm[1], xm[1] + xm[2]);
          mine[ph][2] = pick(H, xm[2], xm[1] - xm[2]);
          const f32x4 s0 = pick(H, xs[0] + xs[1], xs[0]), s1 = pick(H, xs[0] - xs[1], xs[1] + xs[2]), s2 = pick(H, xs[2], xs[1] - xs[2]);
          *reinterpret_cast<f32x4*>(lds + e1_wr + (ph * 3 + 0) * 1024) = s0;
          *reinterpret_cast<f32x4*>(lds + e1_wr + (ph * 3 + 1) * 1024) = s1;
          *reinterpret_cast<f32x4*>(lds + e1_wr + (ph * 3 + 2) * 1024) = s2;
        }
        __syncthreads();
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
          f32x4 rc[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) rc[i] = *reinterpret_cast<const f32x4*>(lds + e1_rd + (ph * 3 + i) * 1024);
          const f32x4 m0 = pick(H, rc[0], mine[ph][0]), s12 = pick(H, rc[1], mine[ph][1]), d12 = pick(H, rc[2], mine[ph][2]);
          const f32x4 s34 = pick(H, mine[ph][0], rc[0]), d34 = pick(H, mine[ph][1], rc[1]), m5 = pick(H, mine[ph][2], rc[2]);
          R[k][ph][0] = m0 + s12 + s34;
          R[k][ph][1] = d12 + 2.f * d34;
          R[k][ph][2] = s12 + 4.f * s34;
          R[k][ph][3] = d12 + 8.f * d34 + m5;
        }
        __syncthreads();   // the area is written again (round 1, then the second exchange)
      }
      const int e2_wr = E_OFF + wave * 6144 + lane * 16, e2_rd = E_OFF + (wave ^ 4) * 6144 + lane * 16;
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int ph = 0; ph < 3; ++ph)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
            *reinterpret_cast<f32x4*>(lds + e2_wr + (ph * 2 + bb) * 1024) = pick(G, R[0][ph][2 * rd + bb], R[1][ph][2 * rd + bb]);
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int b = 2 * rd + bb;
          f32x4 own[3], rcv[3];
#pragma unroll
          for (int ph = 0; ph < 3; ++ph) {
            own[ph] = pick(G, R[1][ph][b], R[0][ph][b]);
            rcv[ph] = *reinterpret_cast<const f32x4*>(lds + e2_rd + (ph * 2 + bb) * 1024);
          }
          // rows of half 0: 0, 1, 3; of half 1: 5, 2, 4
          const f32x4 m0 = pick(G, rcv[0], own[0]), m5 = pick(G, own[0], rcv[0]);
          const f32x4 m1 = pick(G, rcv[1], own[1]), m2 = pick(G, own[1], rcv[1]);
          const f32x4 m3 = pick(G, rcv[2], own[2]), m4 = pick(G, own[2], rcv[2]);
          f32x2 lo[4], hi[4];
          at6x2(m0.xy, m1.xy, m2.xy, m3.xy, m4.xy, m5.xy, lo);
          at6x2(m0.zw, m1.zw, m2.zw, m3.zw, m4.zw, m5.zw, hi);
#pragma unroll
          for (int a = 0; a < 4; ++a) Y[a][b] = f32x4{lo[a].x, lo[a].y, hi[a].x, hi[a].y};
        }
        __syncthreads();   // read before the area is written again (round 1, then the staging)
      }
    }
    W43X_STAMP(58);
    // ---- Y -> pixel rows: this wave's quad = pixel block hj, tile columns tx = 2 g + (lane >> 5), channel 32 ch + (lane & 31), element =
    // tile row ty.  Both blocks' halves of 8 pixel rows staged through LDS, then the row-major epilogue by all threads.
    {
      const int elane = etid & 63;
      const int tx = 2 * g + (elane >> 5);
      const int col = ch * 32 + (elane & 31);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + kb * 64 + c4);
      if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + kb * 64 + c4);
      load_residual(0, res0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        W43X_STAMP(59 + 2 * h);
        if (h) __syncthreads();  // the first halves have been read
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const f32x2 o = h == 0 ? Y[a][b].xy : Y[a][b].zw;
            *reinterpret_cast<float*>(lds + E_OFF + hj * STAGE_BYTES + ((a * 16 + 4 * tx + b) * 64 + col) * 4) = o.x;
            *reinterpret_cast<float*>(lds + E_OFF + hj * STAGE_BYTES + (((4 + a) * 16 + 4 * tx + b) * 64 + col) * 4) = o.y;
          }
        __syncthreads();
        f32x4 vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) vr[k] = *reinterpret_cast<const f32x4*>(lds + E_OFF + eg * STAGE_BYTES + ((k * 16 + (et >> 4)) * 64 + c4) * 4);
        if (h == 0) load_residual(1, res1);   // the second half's residual streams in behind the first half's stores
#pragma unroll
        for (int k = 0; k < 8; ++k) vr[k] = vr[k] * sc + bi + (h == 0 ? res0[k] : res1[k]);
        if (p.relu) {   // one uniform branch per half
#pragma unroll
          for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) vr[k][e] = fmaxf(vr[k][e], 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (!W43X_DBG(p, 8))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vr[k]), y_rsrc, row_voff(h, k), __builtin_amdgcn_readfirstlane(row_offset(h, k)), 0);
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

// u: winograd_weights(..., 4) = [36][Cout][Cin] -> [Cout / 64][Cin / 16][36 components in processing order][3 planes hi / mid /
// lo][2 halves of 32 output channels][lane 64][8 bf16]: element e of lane l is U[comp][cout = 64 kb + 32 ch + (l & 31)][cin = 16 c +
// 8 (l >> 5) + e], the B fragment of v_mfma_f32_32x32x16_bf16 (k = 8 (l >> 5) + e, the order the A fragments read from V have)
std::vector<uint16_t> winograd43_x3_fragments(const std::vector<float>& u, int cout, int cin) {
  if (cout % 64 || cin % 16 || u.size() != (size_t)36 * cout * cin) fail(OCR_ERR_INTERNAL, "winograd43_x3_fragments: bad shape");
  const int nch = cin / 16;
  std::vector<uint16_t> f(3 * u.size());
  size_t o = 0;
  for (int kb = 0; kb < cout / 64; ++kb)
    for (int c = 0; c < nch; ++c)
      for (int ord = 0; ord < 36; ++ord) {
        const int comp = order_comp(ord);
        for (int chh = 0; chh < 2; ++chh)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
              const float x = u[((size_t)comp * cout + kb * 64 + 32 * chh + (l & 31)) * cin + 16 * c + 8 * (l >> 5) + e];
              const uint16_t h = bf16_rne(x);
              const float r1 = x - bf16_f32(h);      // exact
              const uint16_t m = bf16_rne(r1);
              const float r2 = r1 - bf16_f32(m);     // exact
              const size_t at = o + ((size_t)chh * 64 + l) * 8 + e;
              f[at] = h;
              f[at + 1024] = m;
              f[at + 2048] = bf16_rne(r2);
            }
        o += 3 * 1024;
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
  if (C == 64) hipLaunchKernelGGL(winograd43_x3_kernel<4>, dim3((unsigned)grid), dim3(512), 0, s, a);
  else if (C == 128) hipLaunchKernelGGL(winograd43_x3_kernel<8>, dim3((unsigned)grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(winograd43_x3_kernel<16>, dim3((unsigned)grid), dim3(512), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
