// Fused Winograd F(4x4, 3x3) for the 3x3 s1 p1 convs of the large grids (64 -> 64 at H/4: layer1 and the FPN's p2 terms,
// /root/reference/src/text_detection/model.rs:40-55, :126-133): input transform, the thirty-six element-wise GEMMs and the
// output transform in ONE kernel - 36 multiplies per 4x4 outputs (2.25 per output) where F(2x2,3x3) spends 4 and the
// direct form 9, with none of the 36-component tensors ever leaving the CU.
//
// A workgroup (4 waves) owns a 16 x 16 pixel output block = 4 x 4 Winograd tiles (the 16 rows of every GEMM) and 64 output
// channels; wave w owns the output channels 16 w .. 16 w + 15 of ALL 36 components, so its accumulators
// (36 x v_mfma_f32_16x16x4_f32 tiles = 144 registers) hold every component of a (tile, channel) pair in ONE lane:
// the output transform Y = A^T M A needs no exchange between lanes or waves.  The input channels are walked in
// chunks of 16:
//     patch  18 x 18 px x 16 ch (20.5 KB with the row padding) by LDS-DMA, double-buffered: chunk c + 1 (or the next block's
//            chunk 0) streams in while chunk c is used
//     V = B^T d B   one (tile, channel) item per thread: 36 LDS reads, ~110 VALU, 36 LDS writes -> V[36][16][16] in LDS (36 KB)
//     M_xi += V_xi [16 x 16] * U_xi [16 x 16]   4 MFMAs per component; the A operand is one ds_read_b128 of V, the B operand
//                                               comes straight from global memory (L2-resident, host-side fragment order:
//                                               one 16-byte load per lane and component), prefetched nine components ahead
// then the output transform in registers, the 16 x 16 x 64 result block staged through LDS (reusing the patch / V space)
// so that folded BN, residual, ReLU and the stores run on whole 256-byte pixel rows.
// LDS: 2 x 20.5 KB + 36 KB = 77 KB, two workgroups per CU - one multiplies while the other transforms or stores.
//
// The B loads are inline asm with hand-counted s_waitcnt (a compiler-visible load next to the LDS-DMA makes hipcc wait for
// the DMA too).  The compiler believes such a register is valid as soon as the asm statement has run, so the code keeps every
// load's issue, wait and use inside one straight-line chunk body; tests/test_gpu_conv_kernel.py (eight shapes, all 36
// components, every chunk) and the end-to-end parity tests run the shipped binary and fail on any stale operand.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// 16-byte buffer load into registers as inline asm: invisible to the compiler's wait insertion (a compiler-visible load
// would make it wait for the LDS-DMA as well); completion = explicit s_waitcnt + settle()
template <typename R>
__device__ __forceinline__ void load16(f32x4& v, R rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void settle(f32x4& v) { asm volatile("" : "+v"(v)::"memory"); }

struct W43Args {
  const float* x;         // [N][H][W][C]
  const float* ufrag;     // winograd43_fragments()
  const float* scale;     // folded BN, may be null
  const float* bias;
  const float* residual;  // [N][H][W][K], may be null
  float* y;               // [N][H][W][K]
  unsigned x_bytes, u_bytes, y_bytes;
  int H, W, bh, bw;       // block grid: bh x bw blocks of 16 x 16 pixels per image
  int C, K, kblocks;      // channels in / out, K / 64
  int relu;
  int nblocks;
  int xcd_chunks;         // 1: every XCD (blockIdx & 7) walks its own contiguous run of blocks (launcher: grid % (8 kblocks) == 0)
  int debug;              // builds with -DW43_DEBUG only (ocr_test_w43_debug): 1 skip B loads, 2 skip the input transform, 4 skip patch DMA, 8 skip stores
};

constexpr int PP = 18;                          // patch rows / columns
constexpr int PITCH = PP * 64 + 16;             // bytes per patch row: 18 px x 16 channels, + 16 so that tile rows (4 patch rows apart)
                                                // alternate between the two halves of the banks (4 * PITCH = 64 mod 128)
constexpr int PATCH_BYTES = PP * PITCH;         // 21024 B
constexpr int V_OFF = 2 * PATCH_BYTES;          // V[36][16 tiles][16 ch] f32
constexpr int V_BYTES = 36 * 1024;
constexpr int LDS_BYTES = V_OFF + V_BYTES;      // 78912: two workgroups per CU
constexpr unsigned OOB = 0x80000000u;
#ifdef W43_DEBUG
#define W43_DBG(p, bit) ((p).debug & (bit))
#else
#define W43_DBG(p, bit) false
#endif
#ifdef W43_STAMPS
// diagnostic build only (make EXTRA=-DW43_STAMPS): s_memtime of workgroup 0, waves 0 and 1, at the phase boundaries of its
// first four blocks; kept in LDS during the kernel and copied out at the end (tools/w43_stamps.py)
__device__ long long g_w43_stamps[2 * 4 * 32];
#define W43_STAMP(k)                                                                                     \
  do {                                                                                                   \
    if (blockIdx.x == 0 && wave < 2 && lane == 0 && blk_count < 4)                                       \
      stamp_lds[(wave * 4 + blk_count) * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime();            \
  } while (0)
// ... and inside the matrix phases of its third block: before / after the B wait of every step (W43_STEP(c, 2 t [+ 1]))
__device__ long long g_w43_steps[2 * 4 * 26];
#define W43_STEP(c, k)                                                                                   \
  do {                                                                                                   \
    if (blockIdx.x == 0 && wave < 2 && lane == 0 && blk_count == 2 && (c) < 4)                           \
      step_lds[(wave * 4 + (c)) * 26 + (k)] = (long long)__builtin_amdgcn_s_memtime();                   \
  } while (0)
#else
#define W43_STAMP(k) do {} while (0)
#define W43_STEP(c, k) do {} while (0)
#endif
static_assert(128 * 64 * 4 <= V_BYTES, "half of the staged result block fits the V space");

// one step of B^T (x) for six values
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5,
                                    float* t) {
  const float a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// B^T for two independent columns (or rows) at once: v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32
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
// the same for two tiles at once (v_pk_add_f32 / v_pk_fma_f32: the two halves are registers r, r + 1 of an accumulator)
__device__ __forceinline__ void at6x2(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                      f32x2* y) {
  const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}
__device__ __forceinline__ void at6(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                    float* y) {
  const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

// NCH = C / 16 channel chunks; a workgroup produces 64 of the K output channels
template <int NCH>
__global__ __launch_bounds__(256, 2) void winograd43_fused_kernel(W43Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
#ifdef W43_STAMPS
  __shared__ long long stamp_lds[2 * 4 * 32];
  __shared__ long long step_lds[2 * 4 * 26];
  int blk_count = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ufrag), 0, p.u_bytes, 0x00020000);
  // result and residual through buffer descriptors as well: the row part of an address is scalar, the column / channel part
  // one register per lane (out-of-range columns carry an out-of-range voffset: loads return zero, stores are dropped), no
  // per-access VALU address arithmetic and no divergent branches.  No residual = an empty descriptor: every load reads zero.
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.x), 0,
                                                        p.residual ? p.y_bytes : 0u, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;

  // block -> (image, pixel block, output-channel block); innermost kb: the K / 64 workgroups of a pixel block share its
  // patch in L2 (the grid is a multiple of kblocks, so a persistent workgroup keeps its kb)
  auto coords = [&](int bb, int& n_, int& y0_, int& x0_) {
    bb /= p.kblocks;
    x0_ = 16 * (bb % p.bw);
    bb /= p.bw;
    y0_ = 16 * (bb % p.bh);
    n_ = bb / p.bh;
  };
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx & 7 labels the L2 a workgroup sits behind).  Each XCD gets
  // ONE contiguous run of blocks - whole images, or bands of one - and its workgroups walk that run together, `stride`
  // blocks per round in raster order: the 18 x 18 patches of neighbouring blocks (27 % halo) and the K / 64 siblings of a
  // pixel block meet in the same L2 instead of being fetched by eight.  Placement is a speed matter only.
  int first = 0, count = p.nblocks, q = blockIdx.x, stride = gridDim.x;
  if (p.xcd_chunks) {
    const int j = blockIdx.x & 7, sb = p.nblocks / p.kblocks, c = sb >> 3, rem = sb & 7;
    q = blockIdx.x >> 3;
    stride = gridDim.x >> 3;
    first = (j * c + min(j, rem)) * p.kblocks;
    count = (c + (j < rem ? 1 : 0)) * p.kblocks;
  }
  const int kb = q % p.kblocks;   // constant over a workgroup's blocks: stride is a multiple of kblocks

  // ---- patch DMA, one patch row (18 pixels x 64 B = 1152 B) in two pieces: 16 pixels (a full 1 KB instruction; waves 0, 2)
  // and 2 pixels (lanes 0..7 only; waves 1, 3).  36 pieces, nine per wave = rows (wave >> 1) + 2 m.  The row part of the
  // address is scalar (soffset), the column part one register per lane: out-of-range columns / rows read as zero through
  // an out-of-range voffset (the range check covers voffset only, so the soffset part must stay inside the tensor).
  const int part = wave & 1;
  unsigned pvoff = OOB;
  auto patch_columns = [&](int x0_) {
    const int xx = x0_ - 1 + 16 * part + (lane >> 2);
    pvoff = (unsigned)xx < (unsigned)p.W ? (unsigned)((xx * p.C + (lane & 3) * 4) * 4) : OOB;
  };
  auto issue_patch = [&](int n_, int y0_, int c, int buf) {
#pragma unroll
    for (int m = 0; m < 9; ++m) {
      const int row = (wave >> 1) + 2 * m;
      const int yy = y0_ - 1 + row;
      const bool row_ok = (unsigned)yy < (unsigned)p.H;
      const unsigned soff = row_ok ? (unsigned)(((n_ * p.H + yy) * p.W * p.C + c * 16) * 4) : 0u;
      const unsigned dst = lds0 + (unsigned)(buf * PATCH_BYTES + row * PITCH + part * 1024);
      const unsigned voff = row_ok ? pvoff : OOB;
      if (!W43_DBG(p, 4) && (part == 0 || lane < 8)) dma16(x_rsrc, __builtin_amdgcn_readfirstlane(dst), voff, __builtin_amdgcn_readfirstlane(soff));
    }
  };
  // ---- B fragments: a ring of four triples of components, one 16-byte load per lane and component; triple tg of the
  // block = components 3 (tg % 12) .. + 2 of chunk tg / 12
  f32x4 ring[4][3];
  const unsigned b_voff = (unsigned)(tid * 16);
  const unsigned b_base = (unsigned)(kb * NCH) * 36u * 4096u;
  auto issue_b = [&](int tg, f32x4 (&dst)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (W43_DBG(p, 1)) asm volatile("s_nop 0" : "=v"(dst[q]));
      else load16(dst[q], u_rsrc, b_voff, b_base + (unsigned)((tg * 3 + q) * 4096));
    }
  };

  // transform item of this thread: tile column tx = wave, tile row ty, channel ch of the chunk - the four tile rows of a wave
  // read patch rows 4 apart (two bank halves: the natural two cycles of a 64-lane ds_read_b32) and write 256 contiguous
  // bytes of V.  GEMM row of a tile = 4 tx + ty.
  const int t_ch = tid & 15, t_ty = (tid >> 4) & 3;
  const int t_src = 4 * t_ty * PITCH + 4 * wave * 64 + t_ch * 4;
  // V layout: component (1 KB) x channel group g = ch >> 2 (256 B) x GEMM row (16 B = the group's four channels), the row
  // XORed with 4 (g & 1): the A operand of lane (row = lane & 15, g = lane >> 4) is one ds_read_b128 and eight consecutive
  // lanes read 128 contiguous bytes (conflict-free; rows 64 B apart would be a 4-way conflict on every read), while the
  // transform's ds_write_b32 of a wave (4 rows x 16 channels) spreads over both bank halves.
  const int v_dst = V_OFF + (t_ch >> 2) * 256 + ((4 * wave + t_ty) ^ (((t_ch >> 2) & 1) << 2)) * 16 + (t_ch & 3) * 4;
  const unsigned char* a_ptr = lds + V_OFF + (lane >> 4) * 256 + ((lane & 15) ^ (((lane >> 4) & 1) << 2)) * 16;

  bool patch_in_flight = false;  // chunk 0 of this block's patch was requested during the previous block
  for (int lb = q; lb < count; lb += stride) {
    const int blk = first + lb;
    int n, y0, x0;
    coords(blk, n, y0, x0);
    W43_STAMP(0);
    const bool has_next_block = lb + stride < count;
    if (!patch_in_flight) {
      patch_columns(x0);
      issue_patch(n, y0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    patch_in_flight = false;

    f32x4 acc[36];

    auto chunk = [&](const int c, auto first) {
      // this wave's share of patch c has landed: it went out at step 0 of the previous chunk and the wait of step 4 there
      // covered it (first block: the wait above)
      if (c < 4) W43_STAMP(1 + 4 * c);   // (the first four chunks: c128 / c256 have more)
      __syncthreads();  // ... every wave's; and nobody still reads the V of the previous chunk
      if (c < 4) W43_STAMP(2 + 4 * c);   // (the first four chunks: c128 / c256 have more)
      // the B stream of a chunk starts here (the transform covers its latency) and drains inside the chunk: a value loaded by
      // inline asm must not be in flight across the loop's back edge, where the compiler may copy registers it believes ready
      issue_b(c * 12 + 0, ring[0]);
      issue_b(c * 12 + 1, ring[1]);
      issue_b(c * 12 + 2, ring[2]);
      if (!W43_DBG(p, 2)) {
        const unsigned char* pb = lds + (c & 1) * PATCH_BYTES + t_src;
        // packed math on register pairs: the column step on pairs of patch columns, the halves regrouped into pairs of rows
        // (v_pk_mov_b32), the row step on those - 36 + 18 + 36 VALU instead of 144.  Every VALU instruction here competes
        // with the other workgroup's MFMAs for the same pipe and waits up to a whole MFMA (32 cycles) for its slot.
        f32x2 tc[6][3];  // (B^T d)[i][2 jp], [i][2 jp + 1]
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
          f32x2 d[6], t[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            d[i].x = *reinterpret_cast<const float*>(pb + i * PITCH + (2 * jp) * 64);
            d[i].y = *reinterpret_cast<const float*>(pb + i * PITCH + (2 * jp + 1) * 64);
          }
          bt6x2(d[0], d[1], d[2], d[3], d[4], d[5], t);
#pragma unroll
          for (int i = 0; i < 6; ++i) tc[i][jp] = t[i];
        }
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
          f32x2 r[6], o[6];  // rows 2 ip, 2 ip + 1 of B^T d, column j
#pragma unroll
          for (int j = 0; j < 6; ++j)
            r[j] = (j & 1) ? __builtin_shufflevector(tc[2 * ip][j >> 1], tc[2 * ip + 1][j >> 1], 1, 3)
                           : __builtin_shufflevector(tc[2 * ip][j >> 1], tc[2 * ip + 1][j >> 1], 0, 2);
          bt6x2(r[0], r[1], r[2], r[3], r[4], r[5], o);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            *reinterpret_cast<float*>(lds + v_dst + (6 * (2 * ip) + j) * 1024) = o[j].x;
            *reinterpret_cast<float*>(lds + v_dst + (6 * (2 * ip + 1) + j) * 1024) = o[j].y;
          }
        }
      }
      if (c < 4) W43_STAMP(3 + 4 * c);   // (the first four chunks: c128 / c256 have more)
      __syncthreads();  // V of chunk c is complete
      if (c < 4) W43_STAMP(4 + 4 * c);   // (the first four chunks: c128 / c256 have more)
      const bool last_chunk = c + 1 == NCH;
      const bool has_patch = !last_chunk || has_next_block;
      f32x4 acur[3], anext[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) acur[q] = *reinterpret_cast<const f32x4*>(a_ptr + q * 1024);
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        // issue order of a chunk: ... T(t+1) T(t+2) | step t: T(t+3) [, the next patch at t = 0].  Behind triple t are the
        // three triples after it (9 loads) and, at t = 0..3, the nine patch loads; the stream drains at the chunk's end.
        // (Requesting the patch two chunks ahead at the END of the matrix phase instead - legal, its buffer is free once the
        // transform is done - forces it at the next chunk's first B wait: 0.218 vs 0.208 ms.)
        if (t < 9) issue_b(c * 12 + t + 3, ring[(t + 3) & 3]);
        if (t == 0 && has_patch) {
          if (last_chunk) {
            int nn, ny0, nx0;
            coords(blk + stride, nn, ny0, nx0);
            patch_columns(nx0);
            issue_patch(nn, ny0, 0, 0);
            patch_in_flight = true;
          } else {
            issue_patch(n, y0, c + 1, (c + 1) & 1);
          }
        }
        W43_STEP(c, 2 * t);
#ifdef W43_SAFE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        if (t < 4) {
          if (has_patch) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else if (t < 9) {
          asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else if (t == 9) {
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (t == 10) {
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#endif
        W43_STEP(c, 2 * t + 1);
        f32x4(&b)[3] = ring[t & 3];
#pragma unroll
        for (int q = 0; q < 3; ++q) settle(b[q]);
        if (t < 11) {
#pragma unroll
          for (int q = 0; q < 3; ++q) anext[q] = *reinterpret_cast<const f32x4*>(a_ptr + (3 * (t + 1) + q) * 1024);
        }
        // three accumulator chains interleaved: a dependent v_mfma_f32_16x16x4_f32 issued back to back waits for its predecessor
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            acc[3 * t + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[q][e], b[q][e],
                                                                  (decltype(first)::value && e == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[3 * t + q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) acur[q] = anext[q];
      }
    };
    // (Touching the residual block's lines one chunk ahead - 4 bytes per line into a dump area - made the epilogue's loads L2
    // hits but cost more than it gave: 0.212 vs 0.207 ms.)
    // (starting the accumulators from a constant-zero C operand in chunk 0 - peeled, or as a branch inside the loop - saves
    // 144 v_mov per block but gave wrong results with this compiler and no speed-up: 0.211 vs 0.208 ms; kept as a loop)
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NCH; ++c) chunk(c, std::false_type{});

    // ---- output transform in registers: lane = (output channel wave * 16 + (lane & 15), tile column tx = lane >> 4), register
    // r of a component = tile row ty.  The result block goes through LDS in two halves of 8 pixel rows (32 KB inside the V
    // space: patch buffer 0 may already hold the next block's first chunk) so that folded BN, residual, ReLU and the stores
    // run on whole 256-byte pixel rows.
    {
      const int tx = lane >> 4;
      const int col = ((wave * 16 + (lane & 15)) + 16 * tx) & 63;  // rotated by the tile column: the four tiles of a store hit different banks
      const int c4 = (tid & 15) * 4;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + kb * 64 + c4);
      if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + kb * 64 + c4);
      // this lane's pixel column (tid >> 4) and channels c4 .. c4 + 3 of the block, as a byte offset inside an image row
      const int ep_xx = x0 + (tid >> 4);
      const unsigned ep_v = ep_xx < p.W ? (unsigned)((ep_xx * p.K + kb * 64 + c4) * 4) : OOB;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // residual of this half, requested before the transform so that its latency hides behind it
        f32x4 res[8];
        // image row y0 + 8 h + k: its byte offset (scalar; y < 2^31 bytes) and this lane's offset inside it - out of the
        // descriptor's range for rows below the image, so that loads (zeros) and stores (dropped) need no branch
        auto row_offset = [&](int k) -> unsigned { return (unsigned)(((n * p.H + min(y0 + 8 * h + k, p.H - 1)) * p.W * p.K) * 4); };
        auto row_voff = [&](int k) -> unsigned { return y0 + 8 * h + k < p.H ? ep_v : OOB; };
#pragma unroll
        for (int k = 0; k < 8; ++k)
          res[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, row_voff(k), __builtin_amdgcn_readfirstlane(row_offset(k)), 0));
        W43_STAMP(17 + 4 * h);
        __syncthreads();  // h = 0: every wave is done with V; h = 1: the first half has been read
        W43_STAMP(18 + 4 * h);
        {
          // tile rows 2 h and 2 h + 1 together: registers 2 h, 2 h + 1 of every accumulator as one packed pair
          auto pr = [&](int comp) { return h == 0 ? acc[comp].xy : acc[comp].zw; };
          f32x2 u[4][6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            f32x2 t[4];
            at6x2(pr(j), pr(6 + j), pr(12 + j), pr(18 + j), pr(24 + j), pr(30 + j), t);
#pragma unroll
            for (int a = 0; a < 4; ++a) u[a][j] = t[a];
          }
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            f32x2 o[4];
            at6x2(u[a][0], u[a][1], u[a][2], u[a][3], u[a][4], u[a][5], o);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              *reinterpret_cast<float*>(lds + V_OFF + ((a * 16 + 4 * tx + b) * 64 + col) * 4) = o[b].x;
              *reinterpret_cast<float*>(lds + V_OFF + (((4 + a) * 16 + 4 * tx + b) * 64 + col) * 4) = o[b].y;
            }
          }
        }
        W43_STAMP(19 + 4 * h);
        __syncthreads();
        W43_STAMP(20 + 4 * h);
        // all eight pixel rows of the half out of LDS first (one wait, not eight), then arithmetic and stores with one uniform
        // branch (ReLU) instead of two per row; rows below the image are dropped by the descriptor's range check
        f32x4 vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int pl = k * 16 + (tid >> 4);
          vr[k] = *reinterpret_cast<const f32x4*>(lds + V_OFF + (pl * 64 + ((c4 + 16 * (tid >> 6)) & 63)) * 4);
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
          if (!W43_DBG(p, 8))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vr[k]), y_rsrc, row_voff(k), __builtin_amdgcn_readfirstlane(row_offset(k)), 0);
      }
    }
    W43_STAMP(25);
#ifdef W43_STAMPS
    ++blk_count;
#endif
  }
#ifdef W43_STAMPS
  __syncthreads();
  if (blockIdx.x == 0 && tid < 2 * 4 * 32) g_w43_stamps[tid] = stamp_lds[tid];
  if (blockIdx.x == 0 && tid < 2 * 4 * 26) g_w43_steps[tid] = step_lds[tid];
#endif
#endif
}

}  // namespace

#ifdef W43_STAMPS
void winograd43_read_stamps(long long* out) {
  OCR_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w43_stamps), sizeof(g_w43_stamps)));
  OCR_HIP(hipMemcpyFromSymbol(out + 2 * 4 * 32, HIP_SYMBOL(g_w43_steps), sizeof(g_w43_steps)));
}
#endif

// u: winograd_weights(..., 4) = [36][Cout][Cin] -> [Cout / 64][Cin / 16][36][wave 4][lane 64][4]: element e of lane l of
// wave w is U[comp][cout = kb * 64 + 16 w + (l & 15)][cin = 16 c + 4 (l >> 4) + e] - what MFMA e of the chunk reads as its B
// operand (k index l >> 4 <-> channel 4 (l >> 4) + e, the same permutation the A operand's ds_read_b128 applies)
std::vector<float> winograd43_fragments(const std::vector<float>& u, int cout, int cin) {
  if (cout % 64 || cin % 16 || u.size() != (size_t)36 * cout * cin) fail(OCR_ERR_INTERNAL, "winograd43_fragments: bad shape");
  const int nch = cin / 16;
  std::vector<float> f(u.size());
  size_t o = 0;
  for (int kb = 0; kb < cout / 64; ++kb)
    for (int c = 0; c < nch; ++c)
      for (int comp = 0; comp < 36; ++comp)
        for (int w = 0; w < 4; ++w)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e)
              f[o++] = u[((size_t)comp * cout + kb * 64 + 16 * w + (l & 15)) * cin + 16 * c + 4 * (l >> 4) + e];
  return f;
}

static int g_w43_debug = 0;
void winograd43_set_debug(int d) { g_w43_debug = d; }
int winograd43_get_debug() { return g_w43_debug; }

void launch_winograd43_fused(const float* x, const float* ufrag, const float* scale, const float* bias, const float* residual,
                             int relu, float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || (C != 64 && C != 128 && C != 256) || K % 64)
    fail(OCR_ERR_INVALID, "fused Winograd F(4x4): bad shape N=%d H=%d W=%d C=%d K=%d", N, H, W, C, K);
  const long long xb = (long long)N * H * W * C * 4, ub = (long long)36 * C * K * 4;
  if (xb >= (1ll << 31) || (long long)N * H * W * K * 4 >= (1ll << 31)) fail(OCR_ERR_INVALID, "fused Winograd F(4x4): tensor too large");
  W43Args a{};
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
  const long long blocks = (long long)N * a.bh * a.bw * a.kblocks;
  if (blocks >= (1ll << 31)) fail(OCR_ERR_INVALID, "fused Winograd F(4x4): too many blocks");
  a.nblocks = (int)blocks;
  a.debug = g_w43_debug;
  // persistent workgroups, two per CU; a multiple of kblocks so that each keeps its output-channel block, and of 8 kblocks
  // (when there are that many blocks) so that the eight XCDs can each walk a contiguous run of blocks
  long long grid = std::min<long long>(blocks, 2ll * (num_cus > 0 ? num_cus : 256));
  const long long unit = 8ll * a.kblocks;
  if (grid >= unit && !(g_w43_debug & 16)) {   // (bit 16 of the test hook's word: plain linear order, for A/B timing)
    grid = grid / unit * unit;
    a.xcd_chunks = 1;
  } else {
    grid = std::max<long long>(a.kblocks, grid / a.kblocks * a.kblocks);
  }
  if (C == 64) hipLaunchKernelGGL(winograd43_fused_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (C == 128) hipLaunchKernelGGL(winograd43_fused_kernel<8>, dim3((unsigned)grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(winograd43_fused_kernel<16>, dim3((unsigned)grid), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
