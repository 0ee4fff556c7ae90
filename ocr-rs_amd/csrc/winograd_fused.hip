// Fused Winograd F(2x2, 3x3) for the 3x3 s1 p1 convs of the large grids: 64 -> 64 at H/4 (layer1's four convs and
// the FPN's p2 lateral term) and 128 -> 128 / 128 -> 64 at H/8 (layer2, the p3 lateral term),
// /root/reference/src/text_detection/model.rs:40-55, :126-133: input transform, the sixteen
// element-wise GEMMs and the output transform in ONE kernel, so that none of the 16-component tensors that make
// the unfused form (winograd.hip) HBM-bound at this resolution ever leaves the CU.
//
// A workgroup owns an 8 x 16 pixel output block = 4 x 8 Winograd tiles (32 rows of the GEMMs) and all 64 output
// channels.  Its 10 x 18 x 64 input patch is brought into LDS once by LDS-DMA (zero padding = out-of-range
// lanes).  Wave w owns the Winograd column j = w, i.e. the four components xi = (i, w), i = 0..3:
//     V_iw = (B^T d B)_iw                         VALU, from the LDS patch into this wave's own operand buffer
//     M_iw = V_iw [32 x 64] * U_iw [64 x 64]      64 v_mfma_f32_32x32x2_f32, U_iw streamed by LDS-DMA
//     Z_w[a] = sum_i A^T[a][i] M_iw               i = 0 accumulates into Z_w[0], i = 3 into Z_w[1] (its weights are
//                                                 stored negated), i = 1, 2 go through a temporary and VALU adds
// and the column step of the output transform, Y[a][b] = sum_w A^T[b][w] Z_w[a], crosses the waves through LDS,
// followed by folded BN, residual, ReLU and the store.  16 multiplies per 2x2 outputs instead of 36.
// The 64 input channels are walked as two halves of 32 (everything above is linear in the channels, z keeps
// accumulating), which halves every buffer: 22.5 KB patch + 4 x 4 KB V + 4 x 8 KB U = 70.5 KB, two workgroups per
// CU - one multiplies while the other loads its patch, transforms, or runs its epilogue.
#include <cstdlib>

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

struct WfArgs {
  const float* x;         // [N][H][W][C]
  const float* u;         // [16][K cout][C cin], components 12..15 (i = 3) negated
  const float* scale;     // folded BN, may be null
  const float* bias;
  const float* residual;  // [N][H][W][K], may be null
  float* y;               // [N][H][W][K]
  unsigned x_bytes, u_bytes;
  int H, W, bh, bw;       // block grid: bh x bw blocks of 8 x 16 pixels per image
  int C, K, kblocks;      // channels in / out, K / 64
  int relu;
  int nblocks;
};

[[maybe_unused]] constexpr int PH = 10, PWD = 18;                 // patch rows / columns
constexpr int RAW_BYTES = 23 * 1024;             // one 32-channel half of the patch: 180 px x 128 B = 23040, rounded up to
                                                 // whole DMA instructions (the last one zero-fills 512 bytes past the patch)
constexpr int V_BYTES = 32 * 128;                // one component, one channel half: 32 tiles x 32 channels
constexpr int U_BYTES = 64 * 128;                // one component, one channel half: 64 couts x 32 channels
constexpr int ZROW = 68;                         // exchange row of 64 couts, padded: the two lane halves of a C-layout store
                                                 // sit 4 rows apart and would otherwise hit the same banks
constexpr int Z_BYTES = 4 * 32 * ZROW * 4;       // output-transform exchange of one output row a: [wave][tile][cout]
constexpr int WORK_BYTES = RAW_BYTES + 4 * V_BYTES + 4 * U_BYTES;   // 72192: two workgroups per CU
[[maybe_unused]] constexpr int LDS_BYTES = WORK_BYTES > RAW_BYTES + Z_BYTES ? WORK_BYTES : RAW_BYTES + Z_BYTES;
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;

// NCH = C / 32: the input channels are walked in chunks of 32; a workgroup produces 64 of the K output channels
template <int NCH>
__global__ __launch_bounds__(256, 2) void winograd_fused_kernel(WfArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // persistent workgroups (two per CU) walking the blocks: 3 % faster than one workgroup per block; delaying the
  // start of every second one to de-phase the pairs did not help.  A variant with 16-channel chunks (44 KB of LDS,
  // a second V buffer per wave so that the next transform overlaps the MFMAs, three workgroups per CU at 168
  // VGPRs) was 25 % slower: twice the steps, half the MFMA burst per step, register spills.
  // epilogue items of this thread (k = 0, 1: tile = (k * 256 + tid) >> 4, four output channels): position inside a
  // block and element offset from the block's origin, computed once - per block only a scalar base and the bounds
  // checks remain (integer multiplies are slow, and every VALU instruction is time taken from the matrix stream)
  int ep_y[2], ep_x[2], ep_off[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int item = k * 256 + tid;
    const int tile = item >> 4, c4 = (item & 15) * 4;
    ep_y[k] = 2 * (tile >> 3);
    ep_x[k] = 2 * (tile & 7);
    ep_off[k] = (ep_y[k] * p.W + ep_x[k]) * p.K + c4;
  }
  const int row_el = p.W * p.K;  // elements per image row
  bool patch_in_flight = false;  // chunk 0 of this block's patch was requested during the previous block
  for (int blk = blockIdx.x; blk < p.nblocks; blk += gridDim.x) {
  // block -> (image, pixel block, output-channel block); innermost kb: the K / 64 workgroups of a pixel block share
  // its patch in L2
  auto coords = [&](int bb, int& n_, int& y0_, int& x0_, int& kb_) {
    kb_ = bb % p.kblocks;
    bb /= p.kblocks;
    x0_ = 16 * (bb % p.bw);
    bb /= p.bw;
    y0_ = 8 * (bb % p.bh);
    n_ = bb / p.bh;
  };
  int n, y0, x0, kb;
  coords(blk, n, y0, x0, kb);

  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  unsigned char* vbuf = lds + RAW_BYTES + wave * V_BYTES;
  unsigned char* ubuf = lds + RAW_BYTES + 4 * V_BYTES + wave * U_BYTES;
  const unsigned u_lds = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(RAW_BYTES + 4 * V_BYTES + wave * U_BYTES));

  // ---- input patch, channel half hc: 180 pixels x 128 B = 22.5 DMA instructions of 8 pixels, dealt to the waves
  auto issue_patch_at = [&](int pn, int py0, int px0, int hc) {
    const int sub = lane >> 3, chunk = lane & 7;
    for (int k = wave; k < 23; k += 4) {
      const int px = 8 * k + sub;
      const int py = px / PWD, pxx = px - py * PWD;
      const int yy = py0 - 1 + py, xx = px0 - 1 + pxx;
      const bool inside = px < PH * PWD && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const unsigned off = inside ? (unsigned)((((pn * p.H + yy) * p.W + xx) * p.C + hc * 32 + chunk * 4) * 4) : OOB;
      dma16(x_rsrc, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(k * 1024)), off, 0);
    }
  };
  // ---- weights of component (i, wave), channel half hc: 8 DMA instructions of 8 rows x 128 B, swizzled like conv_igemm
  // (measured slower: staging them through registers a whole step ahead, 0.35 vs 0.33 ms; and feeding the MFMA B
  // operand straight from global memory in a lane-contiguous K order with no U in LDS at all, 0.32 vs 0.31 ms -
  // both push the kernel to the 256-VGPR cap of two waves per SIMD)
  const int urow = lane >> 3, uq = lane & 7;
  auto issue_u = [&](int step) {  // step = 4 hc + i
    const int hc = step >> 2, xi = 4 * (step & 3) + wave;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int row = m * 8 + urow;
      const int gq = uq ^ ((row >> 1) & 7);
      dma16(u_rsrc, u_lds + (unsigned)(m * 1024), (unsigned)((row * p.C + hc * 32 + gq * 4) * 4), (xi * p.K + kb * 64) * p.C * 4);
    }
  };
  static_assert(23 * 1024 <= RAW_BYTES && PH * PWD * 128 <= RAW_BYTES, "the patch DMA stays inside its buffer");
  auto issue_patch = [&](int hc) { issue_patch_at(n, y0, x0, hc); };
  if (!patch_in_flight) issue_patch(0);
  patch_in_flight = false;
  issue_u(0);

  // column pair and sign of B^T row `wave` (the wave's Winograd column), row pair and sign of B^T row i:
  //   0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3
  const int ca = wave == 0 ? 0 : wave == 2 ? 2 : 1;
  const int cb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
  const float cs = wave == 1 ? 1.f : -1.f;

  const int frow = lane & 31, half = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  int xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = ((2 * g + half) ^ fsw) * 16;

  // M_iw accumulated per component row i over all channel chunks; the row step of the output transform
  // (Z_w[0] = M_0w + M_1w + M_2w, Z_w[1] = M_1w - M_2w - M_3w; i = 3 is stored negated) is applied once per block.
  // On gfx950 the f32-input MFMA runs on the vector ALU, so every VALU instruction is time taken from the matrix
  // stream (tools/probes/mfma_valu_coissue.hip): adding a temporary into z0 / z1 after every i = 1, 2 step cost 64
  // v_add per step, this costs 128 per block.
  f32x16 m4[4][2];  // started by the first chunk's MFMAs from a constant-zero C operand: no zeroing instructions

  auto gemm = [&](f32x16 (&acc)[2], bool first) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(vbuf + frow * 128 + xoff[g]);
      f32x4 bf[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bf[nt] = *reinterpret_cast<const f32x4*>(ubuf + (nt * 32 + frow) * 128 + xoff[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const f32x16 zero = {};
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[nt][e], (first && g == 0 && e == 0) ? zero : acc[nt], 0, 0, 0);
        }
    }
  };

#pragma unroll
  for (int hc = 0; hc < NCH; ++hc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // this channel half of the patch is complete (every wave's share has landed)
    f32x4 c1[4], c2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // ---- V_iw for 32 tiles x 32 channels: 4 (tile, 4 channels) items per lane.  The column step of the transform
      // (patch row r -> c[r] = d[r][ca] + cs d[r][cb]) is shared by the components: rows 1 and 2 stay in registers
      // from i = 0 / 1 on, so a chunk costs 32 LDS reads per lane instead of 64.
      auto col = [&](int r, f32x4 (&c)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int item = k * 64 + lane;
          const int tile = item >> 3, c4 = item & 7;
          const unsigned char* base = lds + ((2 * (tile >> 3) + r) * PWD + 2 * (tile & 7)) * 128 + c4 * 16;
          c[k] = *reinterpret_cast<const f32x4*>(base + ca * 128) + cs * *reinterpret_cast<const f32x4*>(base + cb * 128);
        }
      };
      f32x4 v[4];
      if (i == 0) {        // d0 - d2
        f32x4 c0[4];
        col(0, c0);
        col(2, c2);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c0[k] - c2[k];
      } else if (i == 1) { // d1 + d2
        col(1, c1);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c1[k] + c2[k];
      } else if (i == 2) { // d2 - d1
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c2[k] - c1[k];
      } else {             // d1 - d3
        f32x4 c3[4];
        col(3, c3);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = c1[k] - c3[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int item = k * 64 + lane;
        const int tile = item >> 3, c4 = item & 7;
        *reinterpret_cast<f32x4*>(vbuf + tile * 128 + ((c4 ^ ((tile >> 1) & 7)) * 16)) = v[k];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's U has landed, its V is written
      if (i == 3) {
        // the last transform of this channel chunk is done in every wave: fetch the next chunk of the patch - or
        // the first chunk of the NEXT block's patch (the exchange buffer of the epilogue lies behind the patch
        // buffer) - now, under the MFMAs of this step
        if (hc + 1 < NCH) {
          __syncthreads();
          issue_patch(hc + 1);
        } else if (blk + (int)gridDim.x < p.nblocks) {
          __syncthreads();
          int nn, ny0, nx0, nkb;
          coords(blk + (int)gridDim.x, nn, ny0, nx0, nkb);
          issue_patch_at(nn, ny0, nx0, 0);
          patch_in_flight = true;
        }
      }
      gemm(m4[i], hc == 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read of U / V has returned
      if (4 * hc + i + 1 < 4 * NCH) issue_u(4 * hc + i + 1);  // the buffer is free: the next step's weights fly
    }
  }

  // ---- column step of the output transform across the waves, through LDS, one output row a at a time: the
  // exchange buffer [wave][tile 32][ZROW] lies behind the patch buffer (over the V / U buffers, free now)
  float* zb = reinterpret_cast<float*>(lds + RAW_BYTES);
  f32x4 res[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) res[k][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();  // a = 0: every wave is done with V / U;  a = 1: row 0 has been combined
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tile = (e & 3) + 8 * (e >> 2) + 4 * half, co = nt * 32 + frow;
        zb[(wave * 32 + tile) * ZROW + co] = a == 0 ? m4[0][nt][e] + m4[1][nt][e] + m4[2][nt][e] : m4[1][nt][e] - m4[2][nt][e] + m4[3][nt][e];
      }
    const size_t blk_el = (((size_t)n * p.H + y0) * p.W + x0) * p.K + kb * 64;  // wave-uniform: scalar arithmetic
    if (a == 0 && p.residual) {
      // residual rows of this thread's epilogue items (tile, 4 couts): in flight during the exchange
      const float* rbase = p.residual + blk_el;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (y0 + ep_y[k] + (q >> 1) < p.H && x0 + ep_x[k] + (q & 1) < p.W)
            res[k][q] = *reinterpret_cast<const f32x4*>(rbase + ep_off[k] + (q >> 1) * row_el + (q & 1) * p.K);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int item = k * 256 + tid;
      const int tile = item >> 4, c4 = (item & 15) * 4;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
      if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + kb * 64 + c4);
      if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + kb * 64 + c4);
      f32x4 z[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) z[w] = *reinterpret_cast<const f32x4*>(&zb[(w * 32 + tile) * ZROW + c4]);
      const f32x4 o0 = z[0] + z[1] + z[2];
      const f32x4 o1 = z[1] - z[2] - z[3];
      if (y0 + ep_y[k] + a >= p.H) continue;
      float* ybase = p.y + blk_el;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (x0 + ep_x[k] + q >= p.W) continue;
        f32x4 val = (q ? o1 : o0) * sc + bi + res[k][2 * a + q];
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(ybase + ep_off[k] + a * row_el + q * p.K) = val;
      }
    }
  }
  __syncthreads();  // the exchange buffer has been read: the next block's V / U may be written
  }  // blk
#endif
}

}  // namespace

void launch_winograd_fused(const float* x, const float* u_neg3, const float* scale, const float* bias, const float* residual,
                           int relu, float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || (C != 64 && C != 128 && C != 256) || K % 64 || K <= 0)
    fail(OCR_ERR_INVALID, "winograd fused: bad shape N=%d H=%d W=%d C=%d K=%d", N, H, W, C, K);
  const long long bytes = (long long)N * H * W * C * 4;
  if (bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "winograd fused: input of %lld bytes must be < 2^31; split the batch", bytes);
  if ((long long)N * H * W * K * 4 >= (1ll << 40)) fail(OCR_ERR_INVALID, "winograd fused: output too large");
  WfArgs a{};
  a.x = x;
  a.u = u_neg3;
  a.scale = scale;
  a.bias = bias;
  a.residual = residual;
  a.y = y;
  a.x_bytes = (unsigned)bytes;
  a.u_bytes = (unsigned)(16ll * K * C * 4);
  a.H = H;
  a.W = W;
  a.bh = (H + 7) / 8;
  a.bw = (W + 15) / 16;
  a.C = C;
  a.K = K;
  a.kblocks = K / 64;
  a.relu = relu;
  const long long blocks = (long long)N * a.bh * a.bw * a.kblocks;
  if (blocks >= (1ll << 31)) fail(OCR_ERR_INVALID, "winograd fused: grid too large");
  a.nblocks = (int)blocks;
  const long long resident = 2ll * (num_cus > 0 ? num_cus : 256);  // persistent: two workgroups per CU of THIS device
  const unsigned grid = blocks > resident ? (unsigned)resident : (unsigned)blocks;
  if (C == 64) hipLaunchKernelGGL(winograd_fused_kernel<2>, dim3(grid), dim3(256), 0, s, a);
  else if (C == 128) hipLaunchKernelGGL(winograd_fused_kernel<4>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(winograd_fused_kernel<8>, dim3(grid), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
