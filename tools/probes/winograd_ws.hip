// Fused Winograd F(2x2, 3x3), wave-specialised: the 3x3 s1 p1 convs of the large grids
//   /root/reference/src/text_detection/model.rs:40-55 (basic_block convs), :126-133 (FPN lateral terms), :143 (p2's term of bin_conv1)
// Same mathematics and the same 8 x 16 pixel block / 64 output channel decomposition as winograd_fused.hip, but the
// two kinds of work no longer take turns inside one wave.  A 512-thread workgroup (one per CU, persistent) holds
//   waves 0-3  MULTIPLIERS, one per SIMD: wave j owns Winograd column j.  Per step (channel chunk hc, component row i)
//              32 x v_mfma_f32_32x32x2_f32: M_ij += V_ij [32 tiles x 32 ch] * U_ij [32 ch x 64 cout]; A from the V ring in
//              LDS, B straight from global memory in host-arranged fragment order (1 KiB per wave load, L2-resident),
//              fetched one step ahead.  Four accumulator sets M_0j .. M_3j for the whole block (the row step of the
//              output transform, Z_j[0] = M_0j + M_1j + M_2j, Z_j[1] = M_1j - M_2j - M_3j, is applied once per block).
//   waves 4-7  HELPERS, sharing the SIMDs: helper j computes V_(i+1)j = (B^T d B)_(i+1)j for the NEXT step from the input
//              patch while the multipliers run the current one, issues the LDS-DMA of the patch chunks two chunks
//              ahead (double-buffered), and runs the previous block's epilogue (column step of the output transform
//              across the four columns through LDS, folded BN, residual, ReLU, stores) in slices spread over steps.
// One s_barrier per step orders the hand-offs; the matrix pipes wait only for that barrier and one LDS read.
#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

// 16-byte buffer load into registers as inline asm: like the LDS-DMA above it is invisible to the compiler's wait
// insertion, so the helper waves' vmcnt discipline is entirely explicit (a compiler-visible load would make hipcc
// wait vmcnt(0) at its first use and at loop heads - i.e. for the patch DMA in flight - and stall the step barrier).
// The value is valid only after the issuing wave's next s_waitcnt vmcnt(0) + settle() of the registers.
template <typename R>
__device__ __forceinline__ f32x4 load16_async(R rsrc, unsigned voff) {
  f32x4 v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
  return v;
}
__device__ __forceinline__ void settle(f32x4& v) { asm volatile("" : "+v"(v)::"memory"); }
// 16-byte buffer store; a lane whose offset is out of range (>= 2^31) writes nothing, so every wave issues the same
// instructions whatever part of its block lies outside the image
template <typename R>
__device__ __forceinline__ void store16(R rsrc, unsigned voff, f32x4 v) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rsrc) : "memory");
}

#ifdef WS_STAMPS
// diagnostic build only (make EXTRA=-DWS_STAMPS): s_memtime of workgroup 0, wave 0 (multiplier) and wave 4 (helper),
// at up to four points of each step; kept in LDS during the kernel (a global store per stamp would itself be waited
// for) and copied out at the end.  Read back by tools/ws_stamps.py.
__device__ long long g_ws_stamps[2 * 128 * 4];
#define WS_STAMP(role, step, k)                                                                          \
  do {                                                                                                   \
    if (blockIdx.x == 0 && col == 0 && lane == 0 && (step) < 128)                                        \
      ws_stamp_lds[((role) * 128 + (step)) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime();         \
  } while (0)
#else
#define WS_STAMP(role, step, k) do {} while (0)
#endif

struct WsArgs {
  const float* x;         // [N][H][W][C]
  const float* uf;        // U = G g G^T as MFMA B fragments: [16 xi][K/64][C/32][2 nt][4 g][64 lanes][4]
  const float* scale;     // folded BN, may be null
  const float* bias;
  const float* residual;  // [N][H][W][K], may be null
  float* y;               // [N][H][W][K]
  unsigned x_bytes, y_bytes;
  int H, W, bh, bw;       // block grid: bh x bw blocks of 8 x 16 pixels per image
  int C, K, kblocks;
  int relu;
  int nblocks;
};

[[maybe_unused]] constexpr int PH = 10, PWD = 18;    // patch rows / columns
constexpr int PATCH_BYTES = 23 * 1024;               // one 32-channel chunk of the patch (180 px x 128 B, whole DMA instructions)
constexpr int V_BYTES = 32 * 128;                    // one component, one chunk: 32 tiles x 32 channels
constexpr int ZROW = 68;
constexpr int Z_BYTES = 2 * 4 * 32 * ZROW * 4;       // [a][column][tile][cout (padded)]
[[maybe_unused]] constexpr int OFF_V = 2 * PATCH_BYTES;
[[maybe_unused]] constexpr int OFF_Z = OFF_V + 2 * 4 * V_BYTES;
[[maybe_unused]] constexpr int LDS_BYTES = OFF_Z + Z_BYTES;           // 149 504
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;

#ifndef WS_SKIP
#define WS_SKIP 0   // timing experiments only (wrong results): 1 no epilogue units, 2 no transforms, 4 no Z combine, 8 no patch DMA
#endif

template <int NCH>
__global__ __launch_bounds__(512, 2) void winograd_ws_kernel(WsArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
#ifdef WS_STAMPS
  __shared__ long long ws_stamp_lds[2 * 128 * 4];
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mul = wave < 4;          // multiplier or helper
  const int col = wave & 3;           // Winograd column of both
  const int htid = tid & 255;         // thread index inside its role
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.y), 0, p.y_bytes, 0x00020000);
  const auto sc_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale ? p.scale : p.y), 0, (unsigned)(p.K * 4), 0x00020000);
  const auto bi_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.y), 0, (unsigned)(p.K * 4), 0x00020000);

  const int nloc = (p.nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // blocks of this workgroup (>= 1)
  const int Q = nloc * NCH;                                                              // its chunks, in order

  auto coords = [&](int t, int& n_, int& y0_, int& x0_, int& kb_) {
    int bb = (int)blockIdx.x + t * (int)gridDim.x;
    kb_ = bb % p.kblocks;
    bb /= p.kblocks;
    x0_ = 16 * (bb % p.bw);
    bb /= p.bw;
    y0_ = 8 * (bb % p.bh);
    n_ = bb / p.bh;
  };

  // Both roles run the SAME loop nest (block t, chunk hc, row i) with exactly one s_barrier per step and two in the
  // prologue; they live in separate branches so that the register allocation is the larger of the two, not the sum.
  if (!mul) {
    // ================================================================ helper waves
    // On gfx950 the f32-input MFMA runs on the vector ALU itself: VALU work of another wave of the SIMD does not
    // overlap it, it is ADDED to it (tools/probes/mfma_valu_coissue.hip: MFMA 1.92 ms, VALU 1.05 ms, both 2.91 ms).
    // What specialisation buys is that the helper's instructions are few and never wait on the matrix stream: with
    // priority they issue between two MFMAs instead of queueing behind the multiplier's back-to-back issue.
    __builtin_amdgcn_s_setprio(3);
    // patch chunk q (block q / NCH, channels 32 (q % NCH) ..) -> patch buffer q & 1: 23 DMA instructions of 8 pixels
    // this lane's patch pixels (its up to six DMA instructions k = col, col + 4, ...): row / column inside the patch
    // and the byte offset relative to the patch origin, computed once
    int dpy[6], dpx[6];
    unsigned drel[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int px = 8 * (col + 4 * j) + (lane >> 3);
      dpy[j] = px < PH * PWD ? px / PWD : (1 << 20);      // past the patch (the 23rd instruction's tail): never inside
      dpx[j] = px - (px / PWD) * PWD;
      drel[j] = (unsigned)(((dpy[j] * p.W + dpx[j]) * p.C + (lane & 7) * 4) * 4);
    }
    auto issue_patch = [&](int q) {
      int pn, py0, px0, pkb;
      coords(q / NCH, pn, py0, px0, pkb);
      const int hc = q % NCH;
      // scalar origin (pixel (py0 - 1, px0 - 1), channel chunk hc): may be "negative" - the sum with drel is what counts
      const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((((pn * p.H + py0 - 1) * p.W + px0 - 1) * p.C + hc * 32) * 4);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int k = col + 4 * j;
        if (k < 23) {  // wave-uniform
          const bool inside = (unsigned)(py0 - 1 + dpy[j]) < (unsigned)p.H && (unsigned)(px0 - 1 + dpx[j]) < (unsigned)p.W;
          dma16(x_rsrc, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((q & 1) * PATCH_BYTES + k * 1024)), inside ? base + drel[j] : OOB, 0);
        }
      }
    };
    // column pair and sign of B^T row `col`:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
    const int ca = col == 0 ? 0 : col == 2 ? 2 : 1;
    const int cb = col == 0 ? 2 : col == 1 ? 2 : col == 2 ? 1 : 3;
    const float cs = col == 1 ? 1.f : -1.f;
    f32x4 c1[4], c2[4];  // column-combined patch rows 1 and 2 of the chunk being transformed (live across its four steps)
#pragma unroll
    for (int k = 0; k < 4; ++k) c1[k] = c2[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // V_(i)col of chunk q -> V slot i & 1: 32 tiles x 32 channels, 4 (tile, 4 channels) items per lane
    auto transform = [&](int q, int i) {
      const unsigned char* patch = lds + (q & 1) * PATCH_BYTES;
      auto colsum = [&](int r, f32x4 (&c)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int item = k * 64 + lane;
          const int tile = item >> 3, c4 = item & 7;
          const unsigned char* base = patch + ((2 * (tile >> 3) + r) * PWD + 2 * (tile & 7)) * 128 + c4 * 16;
          c[k] = *reinterpret_cast<const f32x4*>(base + ca * 128) + cs * *reinterpret_cast<const f32x4*>(base + cb * 128);
        }
      };
      f32x4 v[4];
      if (i == 0) {         // d0 - d2
        f32x4 c0[4];
        colsum(0, c0);
        colsum(2, c2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c0[k] - c2[k];
      } else if (i == 1) {  // d1 + d2
        colsum(1, c1);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c1[k] + c2[k];
      } else if (i == 2) {  // d2 - d1
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c2[k] - c1[k];
      } else {              // d1 - d3
        f32x4 c3[4];
        colsum(3, c3);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c1[k] - c3[k];
      }
      unsigned char* vbuf = lds + OFF_V + ((i & 1) * 4 + col) * V_BYTES;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int item = k * 64 + lane;
        const int tile = item >> 3, c4 = item & 7;
        *reinterpret_cast<f32x4*>(vbuf + tile * 128 + ((c4 ^ ((tile >> 1) & 7)) * 16)) = v[k];
      }
    };
    // epilogue of the block at (en, ey0, ex0, ekb), unit u = 2 a + k: output row a of the tiles, items k * 256 + htid.
    // Its global operands (residual rows, folded BN of the block's 64 channels) are requested by load_block_operands()
    // in the block's last step and are valid after the vmcnt(0) of the next chunk's step i = 1 (or of the tail).
    const float* zb = reinterpret_cast<const float*>(lds + OFF_Z);
    f32x4 res[4][2], sc, bi;            // operands of the block being multiplied (requested in its step (hc 1, i 0))
    f32x4 pres[4][2], psc, pbi;         // ... of the block whose epilogue is pending
    sc = psc = f32x4{1.f, 1.f, 1.f, 1.f};
    bi = pbi = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) res[u][0] = res[u][1] = pres[u][0] = pres[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // byte offset of output pixel (unit u, column qq) of this thread inside a block at (en, ey0, ex0, ekb): everything
    // that depends on the thread alone is computed once (integer multiplies are slow and every VALU instruction of a
    // helper is taken from the multiplier of its SIMD); per block only a scalar base and the bounds checks remain
    int rel_y[4], rel_x[4][2];
    unsigned rel_off[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int a = u >> 1, item = (u & 1) * 256 + htid;
      const int tile = item >> 4, c4 = (item & 15) * 4;
      rel_y[u] = 2 * (tile >> 3) + a;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        rel_x[u][qq] = 2 * (tile & 7) + qq;
        rel_off[u][qq] = (unsigned)(((rel_y[u] * p.W + rel_x[u][qq]) * p.K + c4) * 4);
      }
    }
    auto out_offset = [&](int en, int ey0, int ex0, int ekb, int u, int qq) -> unsigned {
      const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((((en * p.H + ey0) * p.W + ex0) * p.K + ekb * 64) * 4);
      return (ey0 + rel_y[u] < p.H && ex0 + rel_x[u][qq] < p.W) ? base + rel_off[u][qq] : OOB;
    };
    auto load_block_operands = [&](int en, int ey0, int ex0, int ekb) {
      const int c4 = (htid & 15) * 4;
      if (p.scale) sc = load16_async(sc_rsrc, (unsigned)((ekb * 64 + c4) * 4));
      if (p.bias) bi = load16_async(bi_rsrc, (unsigned)((ekb * 64 + c4) * 4));
      if (p.residual) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) res[u][qq] = load16_async(r_rsrc, out_offset(en, ey0, ex0, ekb, u, qq));  // outside the image: zeros
      }
    };
    auto settle_block_operands = [&]() {
      settle(sc);
      settle(bi);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        settle(res[u][0]);
        settle(res[u][1]);
      }
    };
    auto epilogue_unit = [&](int u, int en, int ey0, int ex0, int ekb) {
      const int a = u >> 1, item = (u & 1) * 256 + htid;
      const int tile = item >> 4, c4 = (item & 15) * 4;
      f32x4 z[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) z[w] = *reinterpret_cast<const f32x4*>(&zb[((a * 4 + w) * 32 + tile) * ZROW + c4]);
      const f32x4 o0 = z[0] + z[1] + z[2];
      const f32x4 o1 = z[1] - z[2] - z[3];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        f32x4 val = (qq ? o1 : o0) * psc + pbi + pres[u][qq];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
        }
        store16(y_rsrc, out_offset(en, ey0, ex0, ekb, u, qq), val);
      }
    };

    int n, y0, x0, kb;                       // block being multiplied
    coords(0, n, y0, x0, kb);
    int pn = 0, py0 = 0, px0 = 0, pkb = 0;   // block whose epilogue is pending
    issue_patch(0);
    if (Q > 1) issue_patch(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // prologue barrier 1: chunks 0 and 1 are resident
    transform(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // prologue barrier 2: V of step 0 is written
#pragma unroll 1
    for (int t = 0; t < nloc; ++t) {
#pragma unroll 1
      for (int hc = 0; hc < NCH; ++hc) {
        const int q = t * NCH + hc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool last_of_block = hc == NCH - 1 && i == 3;
          const bool last_step = last_of_block && t == nloc - 1;
          WS_STAMP(1, 4 * q + i, 0);
          // (1) the V operand of the next step
          if (!last_step && !(WS_SKIP & 2)) transform(i == 3 ? q + 1 : q, (i + 1) & 3);
          WS_STAMP(1, 4 * q + i, 2);
          // (2) a slice of the previous block's epilogue, in the steps whose transform is light (never i = 3, which
          // carries two column sums and the DMA issue): local steps 0, 1, 2 and 5; its Z is overwritten in local step
          // 4 NCH - 1 >= 7, its operands were settled in step (hc 1, i 2) of its own block
          if (t > 0 && !(WS_SKIP & 1)) {
            if (hc == 0 && i == 0) epilogue_unit(0, pn, py0, px0, pkb);
            if (hc == 0 && i == 1) epilogue_unit(1, pn, py0, px0, pkb);
            if (hc == 0 && i == 2) epilogue_unit(2, pn, py0, px0, pkb);
            if (hc == 1 && i == 1) epilogue_unit(3, pn, py0, px0, pkb);
          }
          // (3) patch chunk q + 2 into the buffer chunk q has just finished with; (4) in step (hc 1, i 0) this block's
          // residual rows and folded BN; (5) three steps after a DMA issue, at the end of step i = 2 of the next chunk
          // (one barrier before that chunk's first reader), all of it is awaited - the only vector memory wait of the
          // helper waves (nothing here is visible to the compiler's own wait insertion).  A patch chunk takes 5-6 k
          // cycles to land under load.
          WS_STAMP(1, 4 * q + i, 3);
          if (i == 3 && q + 2 < Q && !(WS_SKIP & 8)) issue_patch(q + 2);
          if (hc == 1 && i == 0 && !(WS_SKIP & 1)) load_block_operands(n, y0, x0, kb);
          if (i == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (hc == 1) settle_block_operands();
          }
          if (last_of_block) {
            pn = n, py0 = y0, px0 = x0, pkb = kb;
            psc = sc, pbi = bi;
#pragma unroll
            for (int u = 0; u < 4; ++u) pres[u][0] = res[u][0], pres[u][1] = res[u][1];
            if (!last_step) coords(t + 1, n, y0, x0, kb);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's V writes have landed
          WS_STAMP(1, 4 * q + i, 1);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    // the last block's epilogue (its Z was published by the last step's barrier)
#pragma unroll
    for (int u = 0; u < 4; ++u) epilogue_unit(u, pn, py0, px0, pkb);
  } else {
    // ================================================================ multiplier waves
    const int frow = lane & 31, half = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    int xoff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xoff[g] = ((2 * g + half) ^ fsw) * 16;
    f32x16 acc[4][2];  // M_i,col for the two 32-cout halves
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][nt][e] = 0.f;
    // B fragments of step (block kb, chunk hc, row i): 8 x 16 bytes per lane, contiguous 8 KiB per wave
    auto load_b = [&](int kb_, int hc_, int i_, f32x4 (&b)[2][4]) {
      const f32x4* bp = reinterpret_cast<const f32x4*>(p.uf) + ((size_t)(((4 * i_ + col) * p.kblocks + kb_) * NCH + hc_) * 8) * 64 + lane;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) b[nt][g] = bp[(nt * 4 + g) * 64];
    };
    f32x4 bcur[2][4], bnext[2][4];
    int n, y0, x0, kb;
    coords(0, n, y0, x0, kb);
    load_b(kb, 0, 0, bcur);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) bnext[nt][g] = bcur[nt][g];
    __builtin_amdgcn_s_barrier();            // prologue barrier 1
    __builtin_amdgcn_s_barrier();            // prologue barrier 2
#pragma unroll 1
    for (int t = 0; t < nloc; ++t) {
#pragma unroll 1
      for (int hc = 0; hc < NCH; ++hc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool last_of_block = hc == NCH - 1 && i == 3;
          const bool last_step = last_of_block && t == nloc - 1;
          WS_STAMP(0, 4 * (t * NCH + hc) + i, 0);
          if (!last_step) {
            if (last_of_block) coords(t + 1, n, y0, x0, kb);
            load_b(kb, i == 3 ? (hc + 1) % NCH : hc, (i + 1) & 3, bnext);  // next step's weights fly during this step's MFMAs
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the loads HERE: sunk below the MFMAs (to share registers with bcur)
                                              // they would be awaited right after the barrier, latency exposed
          const unsigned char* vbuf = lds + OFF_V + ((i & 1) * 4 + col) * V_BYTES;
          f32x4 af[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const f32x4*>(vbuf + frow * 128 + xoff[g]);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
                acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][e], bcur[nt][g][e], acc[i][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bcur[nt][g] = bnext[nt][g];
          if (last_of_block && !(WS_SKIP & 4)) {
            // row step of the output transform, then the exchange buffer [a][column][tile][cout]
            float* zw = reinterpret_cast<float*>(lds + OFF_Z);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int tile = (e & 3) + 8 * (e >> 2) + 4 * half, co = nt * 32 + frow;
                zw[((0 * 4 + col) * 32 + tile) * ZROW + co] = acc[0][nt][e] + acc[1][nt][e] + acc[2][nt][e];
                zw[((1 * 4 + col) * 32 + tile) * ZROW + co] = acc[1][nt][e] - acc[2][nt][e] - acc[3][nt][e];
              }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ii][nt][e] = 0.f;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the Z writes of a block's last step have landed
          WS_STAMP(0, 4 * (t * NCH + hc) + i, 1);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
  }
#ifdef WS_STAMPS
  if (blockIdx.x == 0 && col == 0 && lane == 0)
    for (int k = 0; k < 128 * 4; ++k) g_ws_stamps[(mul ? 0 : 1) * 128 * 4 + k] = ws_stamp_lds[(mul ? 0 : 1) * 128 * 4 + k];
#endif
#endif
}

}  // namespace

// U = G g G^T ([16][K][C], winograd_weights()) -> the B fragments the multiplier waves load:
// [xi][K/64][C/32][nt][g][lane][e] = U_xi[64 kb + 32 nt + (lane & 31)][32 hc + 8 g + 4 (lane >> 5) + e]
std::vector<float> winograd_ws_fragments(const std::vector<float>& u, int cout, int cin) {
  const int kblocks = cout / 64, nch = cin / 32;
  std::vector<float> f(u.size());
  for (int xi = 0; xi < 16; ++xi)
    for (int kb = 0; kb < kblocks; ++kb)
      for (int hc = 0; hc < nch; ++hc)
        for (int nt = 0; nt < 2; ++nt)
          for (int g = 0; g < 4; ++g)
            for (int l = 0; l < 64; ++l)
              for (int e = 0; e < 4; ++e) {
                const int co = 64 * kb + 32 * nt + (l & 31), ci = 32 * hc + 8 * g + 4 * (l >> 5) + e;
                f[((((((size_t)xi * kblocks + kb) * nch + hc) * 2 + nt) * 4 + g) * 64 + l) * 4 + e] = u[((size_t)xi * cout + co) * cin + ci];
              }
  return f;
}

#ifdef WS_STAMPS
void winograd_ws_read_stamps(long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_stamps), sizeof(long long) * 2 * 128 * 4); }
#endif

void launch_winograd_ws(const float* x, const float* ufrag, const float* scale, const float* bias, const float* residual, int relu,
                        float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || (C != 64 && C != 128 && C != 256) || K % 64 || K <= 0)
    fail(OCR_ERR_INVALID, "winograd ws: bad shape N=%d H=%d W=%d C=%d K=%d", N, H, W, C, K);
  const long long bytes = (long long)N * H * W * C * 4;
  if (bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "winograd ws: input of %lld bytes must be < 2^31; split the batch", bytes);
  if ((long long)N * H * W * K * 4 >= (1ll << 31)) fail(OCR_ERR_INVALID, "winograd ws: output must be < 2^31 bytes; split the batch");
  WsArgs a{};
  a.x = x;
  a.uf = ufrag;
  a.scale = scale;
  a.bias = bias;
  a.residual = residual;
  a.y = y;
  a.x_bytes = (unsigned)bytes;
  a.y_bytes = (unsigned)((long long)N * H * W * K * 4);
  a.H = H;
  a.W = W;
  a.bh = (H + 7) / 8;
  a.bw = (W + 15) / 16;
  a.C = C;
  a.K = K;
  a.kblocks = K / 64;
  a.relu = relu;
  const long long blocks = (long long)N * a.bh * a.bw * a.kblocks;
  if (blocks >= (1ll << 31)) fail(OCR_ERR_INVALID, "winograd ws: grid too large");
  a.nblocks = (int)blocks;
  const long long resident = num_cus > 0 ? num_cus : 256;  // persistent: one 8-wave workgroup per CU (149.5 KB of LDS)
  const unsigned grid = blocks > resident ? (unsigned)resident : (unsigned)blocks;
  if (C == 64) hipLaunchKernelGGL(winograd_ws_kernel<2>, dim3(grid), dim3(512), 0, s, a);
  else if (C == 128) hipLaunchKernelGGL(winograd_ws_kernel<4>, dim3(grid), dim3(512), 0, s, a);
  else hipLaunchKernelGGL(winograd_ws_kernel<8>, dim3(grid), dim3(512), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
