// NHWC f32 convolution as an implicit GEMM on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 FMA chain, 157 TF/s chip peak).
//
// Implements every conv of /root/reference/src/text_detection/model.rs:107-151:
//   3x3 s1/s2 p1 (basic_block :40-55, out2..5 :80-98, bin_conv1 :100), 1x1 s1/s2
//   (downsample :30-38, in2..in5 :75-78) and, as a 1x1 GEMM with a pixel-shuffle
//   store, conv_transpose2d k=2 s=2 (bin_conv_tr1 :103).
// Fused into the A-operand gather: nearest-upsample(x2) + add (:126-137) and the
// channel concat of the four FPN outputs with their x8/x4/x2 upsamples (:140).
// Fused into the epilogue: eval-mode batch norm as scale/bias, residual add, ReLU.
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile BM x BN x 32; each wave owns
// (BM/2) x (BN/2) as 32x32 MFMA tiles.  Operands are staged global -> VGPR -> LDS
// (K-contiguous rows, stride 36 floats: conflict-free ds_read_b128), next K-step's
// global loads are in flight while the current step's MFMAs run.
#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* src[4];
  const float* wgt;
  const float* scale;
  const float* bias;
  const float* residual;
  float* out;
  int N, Hin, Win, Cin;
  int Ho, Wo, Cout;
  int M;        // N*Ho*Wo
  int pad;
  int relu;
  int nblk_n;   // Cout / BN
  int nblk;     // total blocks
};

// Consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its
// own L2).  Remap so that every XCD walks a contiguous run of tiles: neighbouring
// tiles share input halos and weight panels (bijective for any block count).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

#ifndef PIPE_AT
#define PIPE_AT 0      // K-group after whose fragment reads the staging registers are recycled
#endif
#ifndef PIPE_FENCE
#define PIPE_FENCE 1   // scheduling fence after that group's MFMAs
#endif
constexpr int BK = 32;
constexpr int LDSK = 36;  // row stride in floats (144 B): 16-lane groups hit 16 distinct bank quads

template <int BM, int BN, int KS, int STRIDE, int SRC, int STORE>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvArgs p) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int AI = BM / 32, BI = BN / 32;
  constexpr int TILE = (BM + BN) * LDSK;  // floats per LDS stage
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int bid = xcd_remap(blockIdx.x, p.nblk);
  const int tile_n = bid % p.nblk_n;
  const int tile_m = bid / p.nblk_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread gather coordinates: row r (+32 i), 16-byte chunk q of the K slab.
  // All offsets are 32-bit element offsets (the host checks every tensor < 2^31 elements).
  const int r = tid >> 3;
  const int q = tid & 7;
  int ih0[AI], iw0[AI], abase[AI], img[AI];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r + 32 * i;
    if (m < p.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int oh = rem / p.Wo;
      const int ow = rem - oh * p.Wo;
      img[i] = n;
      ih0[i] = oh * STRIDE - p.pad;
      iw0[i] = ow * STRIDE - p.pad;
      if constexpr (SRC == SRC_CAT4) abase[i] = n;  // image index; pixel offsets depend on the source
      else abase[i] = ((n * p.Hin + ih0[i]) * p.Win + iw0[i]) * p.Cin + q * 4;
    } else {
      ih0[i] = -(1 << 20);  // every tap out of range -> zeros
      iw0[i] = 0;
      abase[i] = 0;
      img[i] = 0;
    }
  }
  int wbase[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) wbase[i] = (n0 + r + 32 * i) * (KS * KS) * p.Cin + q * 4;

  // staging registers of the K-step in flight: raw loads only; zero padding (and the
  // upsample+add of UPADD) are applied when the registers go to LDS, one K-step later,
  // so nothing waits on a load right after issuing it.
  f32x4 areg[AI], breg[BI];
  f32x4 areg1[SRC == SRC_UPADD ? AI : 1];
  unsigned okmask = 0;

  auto load_tiles = [&](int tap, int c0) {
    const int kh = tap / KS, kw = tap - kh * KS;
    okmask = 0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int ih = ih0[i] + kh, iw = iw0[i] + kw;
      const bool ok = (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      okmask |= (ok ? 1u : 0u) << i;
      if constexpr (SRC == SRC_PLAIN) {
        const int off = abase[i] + (kh * p.Win + kw) * p.Cin + c0;
        areg[i] = *reinterpret_cast<const f32x4*>(p.src[0] + (ok ? off : 0));
      } else if constexpr (SRC == SRC_UPADD) {
        const int off = abase[i] + (kh * p.Win + kw) * p.Cin + c0;
        // src[1] is the half-resolution lateral: nearest upsample = index >> 1
        const int off1 = ((img[i] * (p.Hin >> 1) + (ih >> 1)) * (p.Win >> 1) + (iw >> 1)) * p.Cin + c0 + q * 4;
        areg[i] = *reinterpret_cast<const f32x4*>(p.src[0] + (ok ? off : 0));
        areg1[i] = *reinterpret_cast<const f32x4*>(p.src[1] + (ok ? off1 : 0));
      } else {  // SRC_CAT4: channels [0,64) p5 (x8), [64,128) p4 (x4), [128,192) p3 (x2), [192,256) p2
        const int s = c0 >> 6;
        const int sh = 3 - s;
        const int off = ((abase[i] * (p.Hin >> sh) + (ih >> sh)) * (p.Win >> sh) + (iw >> sh)) * 64 + (c0 & 63) + q * 4;
        areg[i] = *reinterpret_cast<const f32x4*>(p.src[s] + (ok ? off : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) breg[i] = *reinterpret_cast<const f32x4*>(p.wgt + wbase[i] + tap * p.Cin + c0);
  };

  auto store_tiles = [&](float* stage) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      f32x4 v = areg[i];
      if constexpr (SRC == SRC_UPADD) v = areg1[i] + v;  // reference order: upsample(x_in{k+1}) + x_in{k}
      *reinterpret_cast<f32x4*>(&stage[(r + 32 * i) * LDSK + q * 4]) = ((okmask >> i) & 1u) ? v : z;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&stage[(BM + r + 32 * i) * LDSK + q * 4]) = breg[i];
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int csteps = p.Cin / BK;
  const int nsteps = KS * KS * csteps;

  // step -> (tap, c0), clamped to the last step so that the loop body needs no branch
  auto load_step = [&](int step) {
    step = min(step, nsteps - 1);
    const int tap = step / csteps;
    load_tiles(tap, (step - tap * csteps) * BK);
  };

  // Pipeline: LDS stage k&1 holds K-step k; registers hold K-step k+1 (in flight); inside the
  // MFMA sequence of step k the registers are written to the other stage and the loads of step
  // k+2 are issued.  One barrier per K-step.
  load_step(0);
  store_tiles(lds);
  load_step(1);
  __syncthreads();

  // MFMA operand fetch: lane l supplies row (l & 31); lanes 0-31 hold k = 8g+j,
  // lanes 32-63 hold k = 8g+4+j for the j-th MFMA of K-group g (same map for A and B).
  const int a_off = (wm * WM + (lane & 31)) * LDSK + (lane >> 5) * 4;
  const int b_off = (BM + wn * WN + (lane & 31)) * LDSK + (lane >> 5) * 4;

  for (int step = 0; step < nsteps; ++step) {
    const float* cur = lds + (step & 1) * TILE;
    float* nxt = lds + ((step + 1) & 1) * TILE;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(cur + a_off + i * 32 * LDSK + g * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(cur + b_off + j * 32 * LDSK + g * 8);
      if (g == PIPE_AT) {
        store_tiles(nxt);      // K-step k+1 (loaded one iteration ago): registers -> other LDS stage
        load_step(step + 2);   // K-step k+2: global -> registers, most of an iteration ahead of its use
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      // Region fence after the first K-group: hipcc otherwise sinks the global loads to the end
      // of the body, next to their consumer.  Pinned here they are issued within the first
      // quarter of the MFMAs and have the other three quarters (plus the next iteration's first
      // group) to land.
      if (PIPE_FENCE && g == PIPE_AT) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  // Per store instruction a wave writes 2 rows x 128 contiguous bytes.  A tile that lies fully
  // inside M takes a branch-free path: all residual loads of a 32x32 tile are issued before
  // they are consumed.
  const int colq = lane & 31;
  const int rowq = (lane >> 5) * 4;
  const bool full_tile = m0 + BM <= p.M;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + wn * WN + j * 32 + colq;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mbase = m0 + wm * WM + i * 32 + rowq;
      if constexpr (STORE == STORE_NHWC) {
        float res[16];
        if (p.residual) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = min(mbase + (e & 3) + 8 * (e >> 2), p.M - 1);
            res[e] = p.residual[(size_t)m * p.Cout + col];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) res[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + (e & 3) + 8 * (e >> 2);
          float v = acc[i][j][e] * sc + bi + res[e];
          if (p.relu) v = fmaxf(v, 0.f);
          if (full_tile || m < p.M) p.out[(size_t)m * p.Cout + col] = v;
        }
      } else {
        // conv_transpose2d k=2 s=2: column = (a*2+b)*64 + co -> out[n][2i+a][2j+b][co]
        const int t = col >> 6, co = col & 63;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mbase + (e & 3) + 8 * (e >> 2);
          float v = acc[i][j][e] * sc + bi;
          if (p.relu) v = fmaxf(v, 0.f);
          const int mm = min(m, p.M - 1);
          const int n = mm / HoWo;
          const int rem = mm - n * HoWo;
          const int oh = rem / p.Wo;
          const int ow = rem - oh * p.Wo;
          const size_t o = (((size_t)n * (2 * p.Ho) + 2 * oh + (t >> 1)) * (2 * p.Wo) + 2 * ow + (t & 1)) * 64 + co;
          if (full_tile || m < p.M) p.out[o] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int KS, int STRIDE, int SRC, int STORE>
void launch_inst(const ConvDesc& d, hipStream_t s) {
  ConvArgs a;
  for (int i = 0; i < 4; ++i) a.src[i] = d.src[i];
  a.wgt = d.wgt;
  a.scale = d.scale;
  a.bias = d.bias;
  a.residual = d.residual;
  a.out = d.out;
  a.N = d.N;
  a.Hin = d.Hin;
  a.Win = d.Win;
  a.Cin = d.Cin;
  a.Ho = d.Ho;
  a.Wo = d.Wo;
  a.Cout = d.Cout;
  a.M = d.N * d.Ho * d.Wo;
  a.pad = d.pad;
  a.relu = d.relu;
  a.nblk_n = d.Cout / BN;
  const int nblk_m = (a.M + BM - 1) / BM;
  a.nblk = nblk_m * a.nblk_n;
  hipLaunchKernelGGL((conv_igemm_f32<BM, BN, KS, STRIDE, SRC, STORE>), dim3(a.nblk), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace

// Host-side shape checks: the kernel assumes exactly these, and an out-of-bounds
// access on the GPU can take the whole node down.
static void check(const ConvDesc& d) {
  if (d.Cin % BK != 0) fail(OCR_ERR_INVALID, "%s: Cin %d not a multiple of %d", d.name, d.Cin, BK);
  if (d.Cout % 64 != 0) fail(OCR_ERR_INVALID, "%s: Cout %d not a multiple of 64", d.name, d.Cout);
  if (d.ks != 1 && d.ks != 3) fail(OCR_ERR_INVALID, "%s: kernel size %d", d.name, d.ks);
  if (d.stride != 1 && d.stride != 2) fail(OCR_ERR_INVALID, "%s: stride %d", d.name, d.stride);
  if (d.pad != (d.ks - 1) / 2) fail(OCR_ERR_INVALID, "%s: pad %d", d.name, d.pad);
  if (d.Ho != (d.Hin + 2 * d.pad - d.ks) / d.stride + 1 || d.Wo != (d.Win + 2 * d.pad - d.ks) / d.stride + 1)
    fail(OCR_ERR_INVALID, "%s: output grid %dx%d does not follow from input %dx%d", d.name, d.Ho, d.Wo, d.Hin, d.Win);
  if ((long long)d.N * d.Ho * d.Wo >= (1ll << 31)) fail(OCR_ERR_INVALID, "%s: M overflows int", d.name);
  // the kernel addresses with 32-bit element offsets
  if ((long long)d.N * d.Hin * d.Win * d.Cin >= (1ll << 31) || (long long)d.N * d.Ho * d.Wo * d.Cout * (d.store_mode == STORE_SHUFFLE2 ? 1 : 1) >= (1ll << 31) * 4)
    fail(OCR_ERR_INVALID, "%s: tensor exceeds 2^31 elements; split the batch", d.name);
  if (d.src_mode == SRC_UPADD && ((d.Hin | d.Win) & 1)) fail(OCR_ERR_INVALID, "%s: UPADD needs even grid", d.name);
  if (d.src_mode == SRC_CAT4 && (d.Cin != 256 || ((d.Hin | d.Win) & 7)))
    fail(OCR_ERR_INVALID, "%s: CAT4 needs Cin 256 and a grid divisible by 8", d.name);
  if (d.store_mode == STORE_SHUFFLE2 && (d.Cout != 256 || d.ks != 1 || d.residual))
    fail(OCR_ERR_INVALID, "%s: SHUFFLE2 store needs a 1x1 conv with Cout 4*64", d.name);
  if (!d.src[0] || !d.wgt || !d.out) fail(OCR_ERR_INVALID, "%s: null operand", d.name);
}

enum Variant { V_3x3_S1_64, V_3x3_S1_128, V_3x3_S2_128, V_1x1_S1_128, V_1x1_S2_128, V_UPADD_64, V_CAT4_64, V_SHUFFLE_128, V_NONE };

static Variant pick(const ConvDesc& d) {
  if (d.store_mode == STORE_SHUFFLE2) return V_SHUFFLE_128;
  if (d.src_mode == SRC_UPADD) return (d.ks == 3 && d.stride == 1 && d.Cout == 64) ? V_UPADD_64 : V_NONE;
  if (d.src_mode == SRC_CAT4) return (d.ks == 3 && d.stride == 1 && d.Cout == 64) ? V_CAT4_64 : V_NONE;
  if (d.ks == 3 && d.stride == 1) return d.Cout == 64 ? V_3x3_S1_64 : (d.Cout % 128 == 0 ? V_3x3_S1_128 : V_NONE);
  if (d.ks == 3 && d.stride == 2) return d.Cout % 128 == 0 ? V_3x3_S2_128 : V_NONE;
  if (d.ks == 1 && d.stride == 1) return d.Cout % 128 == 0 ? V_1x1_S1_128 : V_NONE;
  if (d.ks == 1 && d.stride == 2) return d.Cout % 128 == 0 ? V_1x1_S2_128 : V_NONE;
  return V_NONE;
}

const char* conv_igemm_kernel_name(const ConvDesc& d) {
  switch (pick(d)) {
    case V_3x3_S1_64: return "conv_igemm_f32<128,64,3,1,PLAIN>";
    case V_3x3_S1_128: return "conv_igemm_f32<128,128,3,1,PLAIN>";
    case V_3x3_S2_128: return "conv_igemm_f32<128,128,3,2,PLAIN>";
    case V_1x1_S1_128: return "conv_igemm_f32<128,128,1,1,PLAIN>";
    case V_1x1_S2_128: return "conv_igemm_f32<128,128,1,2,PLAIN>";
    case V_UPADD_64: return "conv_igemm_f32<128,64,3,1,UPADD>";
    case V_CAT4_64: return "conv_igemm_f32<128,64,3,1,CAT4>";
    case V_SHUFFLE_128: return "conv_igemm_f32<128,128,1,1,PLAIN,SHUFFLE2>";
    default: return "conv_igemm_f32<?>";
  }
}

void launch_conv_igemm(const ConvDesc& d, hipStream_t s) {
  check(d);
  switch (pick(d)) {
    case V_3x3_S1_64: launch_inst<128, 64, 3, 1, SRC_PLAIN, STORE_NHWC>(d, s); break;
    case V_3x3_S1_128: launch_inst<128, 128, 3, 1, SRC_PLAIN, STORE_NHWC>(d, s); break;
    case V_3x3_S2_128: launch_inst<128, 128, 3, 2, SRC_PLAIN, STORE_NHWC>(d, s); break;
    case V_1x1_S1_128: launch_inst<128, 128, 1, 1, SRC_PLAIN, STORE_NHWC>(d, s); break;
    case V_1x1_S2_128: launch_inst<128, 128, 1, 2, SRC_PLAIN, STORE_NHWC>(d, s); break;
    case V_UPADD_64: launch_inst<128, 64, 3, 1, SRC_UPADD, STORE_NHWC>(d, s); break;
    case V_CAT4_64: launch_inst<128, 64, 3, 1, SRC_CAT4, STORE_NHWC>(d, s); break;
    case V_SHUFFLE_128: launch_inst<128, 128, 1, 1, SRC_PLAIN, STORE_SHUFFLE2>(d, s); break;
    default: fail(OCR_ERR_INVALID, "%s: no conv_igemm variant for ks=%d stride=%d Cout=%d mode=%d", d.name, d.ks, d.stride, d.Cout, d.src_mode);
  }
}

}  // namespace ocr
