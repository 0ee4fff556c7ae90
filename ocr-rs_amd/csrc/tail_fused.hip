// Fused probability head: bin_conv_tr1 (convT 2x2 s2, 64->64, +bias) + bin_bn2 + ReLU
//                       + bin_conv_tr2 (convT 2x2 s2, 64->1, +bias) + sigmoid (+ binarize)
//   /root/reference/src/text_detection/model.rs:146-150, metrics.rs:129-131
// Both transposed convs have kernel = stride = 2, so one input pixel (i,j) of bin_conv1's output
// owns a 4x4 block of the probability map: no halo, no overlap.  Unfused, the 64-channel
// intermediate at H/2 x W/2 (839 MB per 32 frames) is written and read back; here it lives in
// the accumulators.
//
// Per workgroup: 128 pixels.  For each of the 4 taps t=(a,b) of bin_conv_tr1 a wave computes the
// TRANSPOSED product  Z_t^T[co][px] = W_t[co][ci] * Y^T[ci][px]  with v_mfma_f32_32x32x2_f32
// (weights as the row operand, pixels as the column operand): channels land on registers, pixels
// on lanes.  The second transposed conv contracts over channels, i.e. over REGISTERS of a lane:
// 4 x 32 FMAs per lane and tap plus one cross-half add - no LDS round trip, no transpose.
// Operands move by LDS-DMA exactly as in conv_igemm.hip (same swizzled 128-byte rows).
#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA load as inline asm (see conv_igemm.hip::dma16: the builtin form makes the compiler wait for the
// DMA in front of every following ds_read)
template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

struct TailArgs {
  const void* y;       // bin_conv1 output, NHWC [M][64]            (f32, or bf16 in the bf16 precision)
  const void* wt1;     // [4 taps][64 co][64 ci]                    (same element type as y)
  const float* s4;     // [256] folded bin_bn2 scale, index t*64+co
  const float* b4;     // [256] folded bias (conv bias and BN)
  const float* w2t;    // [64 co][4 u] bin_conv_tr2 weights, u = c'*2+d'
  float* prob;
  uint8_t* bitmap;
  unsigned y_bytes;
  float bias2, thresh;
  int M, h4, w4;       // pixels, grid of y
  unsigned mg_hw, sh_hw, mg_w, sh_w;
};

__device__ __forceinline__ int fdiv(int x, unsigned magic, unsigned shift) {
  if (shift == 0xFFFFFFFFu) return x;
  const unsigned t = __umulhi((unsigned)x, magic);
  return (int)((t + (((unsigned)x - t) >> 1)) >> shift);
}

constexpr int TP = 128;                  // pixels per workgroup
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// TI = float: v_mfma_f32_32x32x2_f32, a pixel's 64 channels are two 128-byte LDS rows (KC = 2 chunks of 32);
// TI = __bf16 (OCR_PRECISION_BF16): v_mfma_f32_32x32x16_bf16, one 128-byte row holds all 64 channels (KC = 1) and a
// 16-byte fragment is one MFMA operand.  Everything after the accumulators (bias, bin_bn2, ReLU, the second transposed
// conv, sigmoid) is f32 in both.
// X3 (TI = float): the f32 head on the bf16 matrix cores - pixels stay f32 in HBM / LDS and are split into three bf16 terms in
// registers once per workgroup (the four taps share them), the weights arrive as three bf16 planes (split3_weights of
// [4 taps][64 co][64 ci]); six partial products per pair, f32 accumulate: 48 MFMAs of 32 cycles per tap instead of 64 of 64.
// f(integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

template <typename TI, bool X3 = false>
__global__ __launch_bounds__(256, 2) void tail_fused_kernel(TailArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr bool BF = sizeof(TI) == 2;
  constexpr int KC = BF ? 1 : 2;                 // 128-byte chunks per pixel row
  constexpr int EB = sizeof(TI);
  constexpr int EBW = X3 ? 2 : EB;               // bytes per weight element
  constexpr int WKC = (BF || X3) ? 1 : 2;        // 128-byte chunks per weight row
  constexpr int WPL = X3 ? 3 : 1;                // weight planes
  constexpr int A_FLOATS = KC * TP * 32;         // LDS words: KC chunks of the pixel rows
  constexpr int W_FLOATS = WPL * WKC * 64 * 32;  // one tap: planes x chunks of 64 output-channel rows
  // X3: the pixels are only read once (split into registers), after which their 32 KB serve as the second weight buffer:
  // 56 KB instead of 80, two workgroups per CU
  static_assert(!X3 || W_FLOATS <= A_FLOATS, "the weight planes of a tap fit the pixel rows' space");
  constexpr int W_STRIDE = X3 ? -A_FLOATS : W_FLOATS;   // buffer 1 relative to buffer 0 (floats)
  __shared__ __attribute__((aligned(1024))) float lds[A_FLOATS + (X3 ? 1 : 2) * W_FLOATS];
  __shared__ __attribute__((aligned(16))) float tab_s[256], tab_b[256], tab_w2[256];
  float* As = lds;
  float* Ws = lds + A_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * TP;
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.y), 0, p.y_bytes, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt1), 0, WPL * 4 * 64 * 64 * EBW, 0x00020000);
  tab_s[tid] = p.s4[tid];
  tab_b[tid] = p.b4[tid];
  tab_w2[tid] = p.w2t[tid];

  const unsigned lds_a = (unsigned)(size_t)(lds_void*)As, lds_w = (unsigned)(size_t)(lds_void*)Ws;
  // DMA: lane fills slot q of rows r + 32 i with global chunk q ^ f(r) (see conv_igemm.hip)
  const int r = tid >> 3, q = tid & 7;
  const int gq = q ^ ((r >> 1) & 7);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc)
#pragma unroll
    for (int i = 0; i < TP / 32; ++i) {
      const int m = m0 + r + 32 * i;
      const unsigned off = m < p.M ? (unsigned)(m * 64 * EB + kc * 128 + gq * 16) : 0x80000000u;  // rows past M read as zeros
      dma16(y_rsrc, __builtin_amdgcn_readfirstlane(lds_a + (unsigned)((kc * TP * 32 + (32 * i + 8 * wave) * 32) * 4)), off, 0);
    }
  auto issue_w = [&](int t, int buf) {
#pragma unroll
    for (int pl = 0; pl < WPL; ++pl)
#pragma unroll
      for (int kc = 0; kc < WKC; ++kc)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          dma16(w_rsrc, __builtin_amdgcn_readfirstlane(lds_w + (unsigned)((buf * W_STRIDE + (pl * WKC + kc) * 64 * 32 + (32 * i + 8 * wave) * 32) * 4)),
                (unsigned)((r + 32 * i) * 64 * EBW + kc * 128 + gq * 16), (pl * 4 + t) * 64 * 64 * EBW);
  };
  issue_w(0, 0);

  const int frow = lane & 31, half = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  int xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = ((2 * g + half) ^ fsw) * 4;

  float o[4][4];  // [tap][u] of this lane's pixel
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // bin_conv_tr2's weights of this lane's 32 channels (co = 32 ct + (e&3) + 8 (e>>2) + 4 half) do not depend on the
  // tap: into registers once.  Read from LDS inside the tap loop they were 32 more dependent round trips per tap.
  // (X3: its pixel fragments need 48 registers; it reads these from LDS per tap and channel half instead - the reads
  // ride under bf16 MFMAs, which leave the vector and LDS pipes free)
  f32x4 w2r[X3 ? 1 : 2][16];   // (X3 reads them from LDS slice by slice instead)
  if constexpr (!X3) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) w2r[ct][e] = *reinterpret_cast<const f32x4*>(&tab_w2[(32 * ct + (e & 3) + 8 * (e >> 2) + 4 * half) * 4]);
  }
  // X3: this lane's pixel operand as three bf16 fragments per 16-channel group kk = 2 kc + gp (chunks g = 2 gp, 2 gp + 1 of
  // row chunk kc: k = 32 kc + 16 gp + 4 half + e and + 8; the weight planes carry the same order inside a 16-group)
  bf16x8 pxh[X3 ? 4 : 1], pxm[X3 ? 4 : 1], pxl[X3 ? 4 : 1];
  if constexpr (X3) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float* row = As + (kk >> 1) * TP * 32 + (32 * wave + frow) * 32;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(row + xoff[2 * (kk & 1)]);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + xoff[2 * (kk & 1) + 1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? a0[e] : a1[e - 4];
        const __bf16 h = (__bf16)x;          // round to nearest even at every level: the remainders are exact in f32
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        pxh[kk][e] = h;
        pxm[kk][e] = m;
        pxl[kk][e] = (__bf16)(r1 - (float)m);
      }
    }
    __syncthreads();   // every wave has its pixels in registers: their LDS rows become weight buffer 1
  }
  if constexpr (X3) {
    // Software-pipelined over the four taps: the MFMAs of tap t (48 of 32 cycles, the two channel halves as ALTERNATING
    // accumulator chains - a dependent MFMA issued back to back waits for its predecessor) carry the epilogue of tap t - 1
    // (bias / bin_bn2 / ReLU / bin_conv_tr2 / sigmoid: ~230 VALU and 48 LDS reads per lane) between them; two accumulator
    // sets alternate.  The empty asm statements pin that epilogue to its region (the compiler would otherwise sink it to
    // the stores at the end of the kernel, where no MFMA covers it).
    f32x16 accs[2][2];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x8 wfr[2][2][3];     // [buffer][channel half][hi, mid, lo]: weight fragments of a 16-channel group
    f32x4 svq[2], bvq[2], w2e[8];
    auto load_w = [&](auto t_c, auto kk_c) {
      constexpr int t = decltype(t_c)::value, kk = decltype(kk_c)::value, buf = kk & 1;
      const float* wb = Ws + (t & 1) * W_STRIDE;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float* wr = wb + (32 * ct + frow) * 32 + xoff[kk];   // chunk 2 kk + half of the 64-channel bf16 row
        wfr[buf][ct][0] = *reinterpret_cast<const bf16x8*>(wr);
        wfr[buf][ct][1] = *reinterpret_cast<const bf16x8*>(wr + 64 * 32);
        wfr[buf][ct][2] = *reinterpret_cast<const bf16x8*>(wr + 2 * 64 * 32);
      }
    };
    // MFMA idx (0..11) of 16-channel group kk: product idx / 2 (small terms first: mid.lo, lo.mid, lo.lo are below 2^-23
    // of the product) of channel half idx % 2 - consecutive MFMAs go to different accumulators, per accumulator the order
    // is the one of the un-pipelined form (bit-identical sums)
    auto mfma_one = [&](auto t_c, auto kk_c, auto idx_c) {
      constexpr int t = decltype(t_c)::value, kk = decltype(kk_c)::value, idx = decltype(idx_c)::value;
      constexpr int ct = idx & 1, pr = idx >> 1, buf = kk & 1;
      constexpr int wsel = pr == 0 ? 2 : (pr == 2 || pr == 3) ? 1 : 0;
      if constexpr (pr == 1) accs[t & 1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[buf][ct][wsel], pxl[kk], accs[t & 1][ct], 0, 0, 0);
      else if constexpr (pr == 2 || pr == 4) accs[t & 1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[buf][ct][wsel], pxm[kk], accs[t & 1][ct], 0, 0, 0);
      else accs[t & 1][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[buf][ct][wsel], pxh[kk], accs[t & 1][ct], 0, 0, 0);
    };
    // epilogue of tap t in 32 elements: quarter s4 = (ct, q pair), element j = 4 qq + r4 of the quarter; channel
    // co = 32 ct + (e&3) + 8 (e>>2) + 4 half with e = 4 q + r4; accs[t & 1][ct][e] = Z_t^T[co][pixel = lane&31]
    auto finish_reads = [&](auto t_c, auto s4_c, auto first_c) {   // the LDS operands of quarter s4: first / second half of them
      constexpr int t = decltype(t_c)::value, s4 = decltype(s4_c)::value, first = decltype(first_c)::value, ct = s4 >> 1;
      if constexpr (first) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int co0 = t * 64 + 32 * ct + 8 * (2 * (s4 & 1) + qq) + 4 * half;
          svq[qq] = *reinterpret_cast<const f32x4*>(&tab_s[co0]);   // folded bias / bin_bn2 of this tap
          bvq[qq] = *reinterpret_cast<const f32x4*>(&tab_b[co0]);
        }
      }
      static_for<(first ? 0 : 4), (first ? 4 : 8)>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value, e = 4 * (2 * (s4 & 1) + (j >> 2)) + (j & 3);
        w2e[j] = *reinterpret_cast<const f32x4*>(&tab_w2[(32 * ct + (e & 3) + 8 * (e >> 2) + 4 * half) * 4]);
      });
    };
    auto finish_elem = [&](auto t_c, auto s4_c, auto j_c) {
      constexpr int t = decltype(t_c)::value, s4 = decltype(s4_c)::value, j = decltype(j_c)::value;
      constexpr int ct = s4 >> 1, qq = j >> 2, r4 = j & 3, e = 4 * (2 * (s4 & 1) + qq) + r4;
      const float z = fmaxf(accs[t & 1][ct][e] * svq[qq][r4] + bvq[qq][r4], 0.f);  // + bias, bin_bn2, ReLU
#pragma unroll
      for (int u = 0; u < 4; ++u) part[u] = fmaf(z, w2e[j][u], part[u]);
    };
    auto finish_end = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sm = part[u] + __shfl_xor(part[u], 32, 64) + p.bias2;  // the other half holds the other 32 channels
        o[t][u] = 1.0f / (1.0f + expf(-sm));
        part[u] = 0.f;
      }
    };
    using std::integral_constant;
    // one slice = one MFMA of tap t plus its share of the epilogue of tap t - 1; nothing moves across a slice boundary
    static_for<0, 4>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      if constexpr (t + 1 < 4) issue_w(t + 1, (t + 1) & 1);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[t & 1][ct][e] = 0.f;
      load_w(t_c, integral_constant<int, 0>{});
      static_for<0, 4>([&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        static_for<0, 12>([&](auto i_c) {
          constexpr int i = decltype(i_c)::value;
          mfma_one(t_c, kk_c, i_c);
          if constexpr (t > 0) {
            constexpr integral_constant<int, (t > 0 ? t - 1 : 0)> tp{};
            if constexpr (i == 0) finish_reads(tp, kk_c, integral_constant<int, 1>{});
            if constexpr (i == 1) finish_reads(tp, kk_c, integral_constant<int, 0>{});
            if constexpr (i >= 2 && i < 10) finish_elem(tp, kk_c, integral_constant<int, (i >= 2 && i < 10 ? i - 2 : 0)>{});
            if constexpr (i == 11 && kk == 3) finish_end(tp);
          }
          if constexpr (i == 10 && kk < 3) load_w(t_c, integral_constant<int, (kk < 3 ? kk + 1 : 0)>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      if constexpr (t + 1 < 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    });
    static_for<0, 4>([&](auto s4_c) {
      constexpr integral_constant<int, 3> t3{};
      finish_reads(t3, s4_c, integral_constant<int, 1>{});
      finish_reads(t3, s4_c, integral_constant<int, 0>{});
      static_for<0, 8>([&](auto j_c) { finish_elem(t3, s4_c, j_c); });
    });
    finish_end(integral_constant<int, 3>{});
  } else {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t + 1 < 4) issue_w(t + 1, (t + 1) & 1);
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
    const float* wb = Ws + (t & 1) * W_STRIDE;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 px = *reinterpret_cast<const f32x4*>(As + kc * TP * 32 + (32 * wave + frow) * 32 + xoff[g]);
        f32x4 wf[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wf[ct] = *reinterpret_cast<const f32x4*>(wb + kc * 64 * 32 + (32 * ct + frow) * 32 + xoff[g]);
        if constexpr (BF) {  // the 16-byte fragment is 8 bf16 of K = 16 g + 8 half + j: one MFMA
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ct]), __builtin_bit_cast(bf16x8, px), acc[ct], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ct][e], px[e], acc[ct], 0, 0, 0);  // rows = co, cols = pixels
        }
      }
    // acc[ct][e] = Z_t^T[co][pixel = lane&31] with co = 32 ct + (e&3) + 8 (e>>2) + 4 half
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 sv[2][4], bv[2][4];  // folded bias / bin_bn2 of this tap: all sixteen reads first, then the arithmetic
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co0 = t * 64 + 32 * ct + 8 * q + 4 * half;
        sv[ct][q] = *reinterpret_cast<const f32x4*>(&tab_s[co0]);
        bv[ct][q] = *reinterpret_cast<const f32x4*>(&tab_b[co0]);
      }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float z = fmaxf(acc[ct][e] * sv[ct][e >> 2][e & 3] + bv[ct][e >> 2][e & 3], 0.f);  // + bias, bin_bn2, ReLU
#pragma unroll
        for (int u = 0; u < 4; ++u) part[u] = fmaf(z, w2r[ct][e][u], part[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = part[u] + __shfl_xor(part[u], 32, 64) + p.bias2;  // the other half holds the other 32 channels
      o[t][u] = 1.0f / (1.0f + expf(-s));
    }
    if (t + 1 < 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  }
  // pixel (n,i,j) owns prob[n][4i .. 4i+3][4j .. 4j+3]: row 2a+c', column 2b+d' = o[a*2+b][c'*2+d'].
  // Lanes 0-31 write rows 0,1, lanes 32-63 rows 2,3 (both halves hold the sums): float4 per row,
  // 512 contiguous bytes per 32 lanes.
  const int m = m0 + 32 * wave + frow;
  if (m < p.M) {
    const int hw = p.h4 * p.w4;
    const int n = fdiv(m, p.mg_hw, p.sh_hw);
    const int rem = m - n * hw;
    const int i = fdiv(rem, p.mg_w, p.sh_w);
    const int j = rem - i * p.w4;
    const int W = 4 * p.w4;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const int rr = 2 * half + rs;  // output row inside the 4x4 block: a = rr>>1 = half, c' = rs
      f32x4 v;
      v[0] = half ? o[2][2 * rs] : o[0][2 * rs];
      v[1] = half ? o[2][2 * rs + 1] : o[0][2 * rs + 1];
      v[2] = half ? o[3][2 * rs] : o[1][2 * rs];
      v[3] = half ? o[3][2 * rs + 1] : o[1][2 * rs + 1];
      const size_t off = ((size_t)n * (4 * p.h4) + 4 * i + rr) * W + 4 * j;
      *reinterpret_cast<f32x4*>(p.prob + off) = v;
      if (p.bitmap) {
        const unsigned bits = (v[0] > p.thresh ? 1u : 0u) | (v[1] > p.thresh ? 0x100u : 0u) | (v[2] > p.thresh ? 0x10000u : 0u) |
                              (v[3] > p.thresh ? 0x1000000u : 0u);
        *reinterpret_cast<unsigned*>(p.bitmap + off) = bits;
      }
    }
  }
#endif
}

void make_magic(unsigned d, unsigned* magic, unsigned* shift) {
  unsigned L = 0;
  while ((1ull << L) < d) ++L;
  *magic = L == 0 ? 0u : (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
  *shift = L == 0 ? 0xFFFFFFFFu : L - 1;
}

}  // namespace

// wt1: f32 [4][64][64]; bf16 != 0: the same in bf16 (y bf16 too); bf16 == 2: y f32, wt1 = split3_weights planes (X3)
void launch_tail_fused(const void* y, const void* wt1, int bf16, const float* s4, const float* b4, const float* w2t, float bias2,
                       float* prob, uint8_t* bitmap, float thresh, int N, int h4, int w4, hipStream_t s) {
  const long long M = (long long)N * h4 * w4;
  const int eb = bf16 == 1 ? 2 : 4;
  if (M * 64 * eb >= (1ll << 31)) fail(OCR_ERR_INVALID, "tail: input exceeds 2^31 bytes; split the batch");
  if (w4 < 1 || h4 < 1) fail(OCR_ERR_INVALID, "tail: bad grid");
  TailArgs a{};
  a.y = y;
  a.wt1 = wt1;
  a.s4 = s4;
  a.b4 = b4;
  a.w2t = w2t;
  a.prob = prob;
  a.bitmap = bitmap;
  a.y_bytes = (unsigned)(M * 64 * eb);
  a.bias2 = bias2;
  a.thresh = thresh;
  a.M = (int)M;
  a.h4 = h4;
  a.w4 = w4;
  make_magic((unsigned)(h4 * w4), &a.mg_hw, &a.sh_hw);
  make_magic((unsigned)w4, &a.mg_w, &a.sh_w);
  if (bf16 == 2) hipLaunchKernelGGL((tail_fused_kernel<float, true>), dim3((unsigned)((M + TP - 1) / TP)), dim3(256), 0, s, a);
  else if (bf16) hipLaunchKernelGGL(tail_fused_kernel<__bf16>, dim3((unsigned)((M + TP - 1) / TP)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(tail_fused_kernel<float>, dim3((unsigned)((M + TP - 1) / TP)), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
