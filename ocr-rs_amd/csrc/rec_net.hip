// Recognition network on the matrix cores, weights shared across crops.
//   /root/reference/src/char_recognition/model.rs:27-39
//     view[-1,1,28,28] -> conv5x5(1->32)+b -> maxpool2 -> conv5x5(32->64)+b -> maxpool2
//     -> view[-1,1024] -> fc1(1024->512)+b -> ReLU -> (dropout: identity in eval) -> fc2(512->62)+b
//   /root/reference/src/char_recognition/mod.rs:53-56 + utils.rs:28-43
//     softmax(-1, Kind::Double) and top-1 (label index into "A-Za-z0-9").
//
// Three stages, every multiply on v_mfma_f32_32x32x2_f32 (exact f32 FMA chains):
//   rec_conv_kernel<T>   conv1 + pool + conv2 + pool of T crops per workgroup, intermediates in LDS, -> feat [B][1024]
//                        conv1: M = 576 pixels, N = 32, K = 25 (+1 zero) - weights live in 13 registers per lane;
//                        conv2: implicit GEMM M = 64 T pixels, N = 64, K = 800 (25 taps x 32 channels) - the weights of
//                        a tap (16 registers per lane, host-arranged in MFMA fragment order so that a wave's load is
//                        1 KiB contiguous) are fetched from L2 once per tap and serve T row tiles.
//                        Rows of an MFMA tile are (pool window, position in window): the 2x2 max pool is a max over
//                        the four accumulator registers of a lane, and the pooled pixels a lane holds are contiguous.
//   fc1                  the batched GEMM [B x 1024] x [1024 x 512] (+b, ReLU) through conv_igemm's 1x1 form - LDS-DMA
//                        staged, weights shared by the 64 / 128 crops of a tile
//   rec_fc2_softmax_kernel  [B x 512] x [512 x 64] (+b; 62 real columns), softmax over the 62 logits in f64, top-1
#include <cstring>

#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IMG = 784;            // 28 x 28 input pixels
constexpr int P1_PIX = 144;         // 12 x 12 pooled pixels after conv1
constexpr int P1_ROW = 32;          // channels per pooled pixel (one 128-byte LDS row)

// float index of channel c of pooled pixel `pix` in a crop's [144][32] LDS image: the 16-byte chunk c / 4 sits in
// slot (c / 4) ^ ((pix >> 1) & 7), which spreads the 32 pixel rows a wave reads over the LDS banks
__device__ __forceinline__ int p1_index(int pix, int c) { return pix * P1_ROW + ((((c >> 2) ^ ((pix >> 1) & 7)) << 2) | (c & 3)); }

// T = crops per workgroup.  256 threads = 4 waves; 80 000 bytes of LDS at T = 4: two workgroups per CU.
template <int T>
__global__ __launch_bounds__(256, 2) void rec_conv_kernel(const float* __restrict__ crops, int n, const float* __restrict__ w1f,
                                                          const float* __restrict__ b1, const float* __restrict__ w2f,
                                                          const float* __restrict__ b2, float* __restrict__ feat) {
  constexpr int NIMG = T < 2 ? T : 2;  // crops staged at a time (conv1 runs over the crops two by two)
  __shared__ __attribute__((aligned(16))) float p1[T * P1_PIX * P1_ROW];
  __shared__ __attribute__((aligned(16))) float img[NIMG * IMG];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int crop0 = blockIdx.x * T;

  // conv1 weights as MFMA B fragments: step s multiplies taps k = 2 s + h (k = 25 is the zero pad)
  float w1r[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) w1r[s] = w1f[s * 64 + lane];
  const float bias1 = b1[j];

  // ---- staging: crop c of the batch -> img slot; crops past the end of the batch re-read the last one (never stored)
  auto load_pair = [&](int first, f32x4 (&r)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;  // float4 index inside the NIMG * 196 float4 of the pair
      if (idx < NIMG * (IMG / 4)) {
        const int c = idx / (IMG / 4), o = idx - c * (IMG / 4);
        const int g = min(crop0 + first + c, n - 1);
        r[u] = *reinterpret_cast<const f32x4*>(crops + (size_t)g * IMG + o * 4);
      }
    }
  };
  auto store_pair = [&](const f32x4 (&r)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u;
      if (idx < NIMG * (IMG / 4)) *reinterpret_cast<f32x4*>(img + idx * 4) = r[u];
    }
  };

  // ---- conv1 (valid 5x5, 1 -> 32) + bias + 2x2 max pool of one 8-window tile: rows i = 4 * window slot + (dy, dx)
  auto conv1_tile = [&](int slot_img, int crop_local, int tile) {
    const int i = lane & 31;
    const int wi = tile * 8 + (i >> 2);            // pooled pixel (window) 0..143
    const int qy = wi / 12, qx = wi - qy * 12;
    const int base = (2 * qy + ((i >> 1) & 1)) * 28 + 2 * qx + (i & 1);
    const float* im = img + slot_img * IMG + base;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = (k0 / 5) * 28 + k0 % 5;
      const int o1 = k1 < 25 ? (k1 / 5) * 28 + k1 % 5 : 0;  // pad tap: weight 0, any in-range pixel
      const float a = im[h ? o1 : o0];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1r[s], acc, 0, 0, 0);
    }
    // C/D map: column = lane & 31 (channel), rows 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3 = window slot 2 g + h
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float m = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3])) + bias1;
      p1[crop_local * (P1_PIX * P1_ROW) + p1_index(tile * 8 + 2 * g + h, j)] = m;
    }
  };

  f32x4 stage[2];
  load_pair(0, stage);
  store_pair(stage);
  if constexpr (T > 2) load_pair(2, stage);  // in flight during conv1 of the first pair
  __syncthreads();
  if constexpr (T == 1) {
    for (int t = wave; t < 18; t += 4) conv1_tile(0, 0, t);
  } else {
#pragma unroll 1
    for (int pr = 0; pr < T / 2; ++pr) {
      if (pr > 0) {
        __syncthreads();  // every wave is done reading the previous pair
        store_pair(stage);
        if (2 * pr + 2 < T) load_pair(2 * pr + 2, stage);
        __syncthreads();
      }
      const int c = wave >> 1;  // this wave's crop of the pair, half of its 18 tiles
#pragma unroll 1
      for (int t = 9 * (wave & 1); t < 9 * (wave & 1) + 9; ++t) conv1_tile(c, 2 * pr + c, t);
    }
  }
  __syncthreads();

  // ---- conv2 (valid 5x5, 32 -> 64) + bias + 2x2 max pool.  Wave w: output channels 32 (w & 1) .. + 31 and T of the 2 T
  // row tiles (a crop's 8 x 8 outputs = 16 windows = 2 tiles of 8 windows).  Row i of tile t' = window slot i >> 2
  // -> pooled pixel p = 8 t' + 4 (slot & 1) + (slot >> 1), so that after the pool lane half h holds p = 8 t' + 4 h + g.
  const int ct = wave & 1;
  int pix0[T], crow[T];
#pragma unroll
  for (int r = 0; r < T; ++r) {
    const int rt = (wave >> 1) * T + r;
    const int crop_local = rt >> 1, tp = rt & 1;
    const int i = lane & 31, slot = i >> 2;
    const int p = 8 * tp + 4 * (slot & 1) + (slot >> 1);
    const int y = 2 * (p >> 2) + ((i >> 1) & 1), x = 2 * (p & 3) + (i & 1);
    pix0[r] = y * 12 + x;
    crow[r] = crop_local * (P1_PIX * P1_ROW);
  }
  f32x16 acc[T];
#pragma unroll
  for (int r = 0; r < T; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

  const f32x4* wf = reinterpret_cast<const f32x4*>(w2f) + ct * 4 * 64 + lane;  // [tap][ct][g][lane] float4
  f32x4 bcur[4], bnext[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bcur[g] = wf[g * 64];
#pragma unroll 1
  for (int tap = 0; tap < 25; ++tap) {
    const int tn = tap + 1 < 25 ? tap + 1 : tap;
#pragma unroll
    for (int g = 0; g < 4; ++g) bnext[g] = wf[(tn * 2 * 4 + g) * 64];  // next tap's weights fly during this tap's MFMAs
    const int ky = tap / 5, kx = tap - ky * 5;
    const int toff = ky * 12 + kx;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[T];
#pragma unroll
      for (int r = 0; r < T; ++r) {
        const int pix = pix0[r] + toff;
        a[r] = *reinterpret_cast<const f32x4*>(p1 + crow[r] + pix * P1_ROW + (((2 * g + h) ^ ((pix >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < T; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r][e], bcur[g][e], acc[r], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) bcur[g] = bnext[g];
  }

  // pooled + bias -> feat[crop][co * 16 + p] (the C,H,W flatten of view[-1,1024]): a lane's four p are contiguous
  const float bias2 = b2[32 * ct + j];
#pragma unroll
  for (int r = 0; r < T; ++r) {
    const int rt = (wave >> 1) * T + r;
    const int crop = crop0 + (rt >> 1), tp = rt & 1;
    f32x4 v;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      v[g] = fmaxf(fmaxf(acc[r][4 * g], acc[r][4 * g + 1]), fmaxf(acc[r][4 * g + 2], acc[r][4 * g + 3])) + bias2;
    if (crop < n) *reinterpret_cast<f32x4*>(feat + (size_t)crop * 1024 + (32 * ct + j) * 16 + 8 * tp + 4 * h) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Low-latency forms for small batches (BASELINE configs[2] is 256 crops: one launch wave, nothing to amortise).  The
// time of the kernels above is then the LENGTH OF ONE DEPENDENT MFMA CHAIN (conv2 accumulates K = 800 as 400
// v_mfma_f32_32x32x2_f32 of 64 cycles each, fc1 K = 1024 as 512) on a machine that is mostly idle.
// Three launches built for LATENCY, one crop per workgroup in the conv stage:
// rec_conv_small_x3_kernel: conv1 as above; the pooled map goes to LDS as three bf16 images (p = hi + mid + lo exactly), and
// conv2 runs on v_mfma_f32_16x16x32_bf16 - K = 32 input channels of one tap per instruction, six partial products per
// tap (weights as three bf16 fragment sets): 150 MFMAs of 16 cycles per wave where the f32 form issues 200 of 32.
// Results are f32-accurate (dropped terms <= 2^-23 of a product), not bit-identical to the large-batch path - the
// contract is logits within 1e-4 and labels exact.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int P1X_PLANE = P1_PIX * 64;   // bytes of one bf16 image [144 pixels][32 channels]
// byte offset of channel c of pooled pixel pix inside a plane: 16-byte chunk c / 8 in slot (c / 8) ^ ((pix >> 1) & 3)
// (conflict-free for the conv2 fragment reads of every tap: checked exhaustively)
__device__ __forceinline__ int p1x_offset(int pix, int c) { return pix * 64 + ((((c >> 3) ^ ((pix >> 1) & 3)) << 4) | ((c & 7) << 1)); }

__global__ __launch_bounds__(1024) void rec_conv_small_x3_kernel(const float* __restrict__ crops, int n, const float* __restrict__ w1f,
                                                                 const float* __restrict__ b1, const uint4* __restrict__ w2x,
                                                                 const float* __restrict__ b2, float* __restrict__ feat) {
  __shared__ __attribute__((aligned(16))) unsigned char p1x[3 * P1X_PLANE];
  __shared__ __attribute__((aligned(16))) float img[IMG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crop = blockIdx.x;
  if (tid < IMG / 4) *reinterpret_cast<f32x4*>(img + tid * 4) = *reinterpret_cast<const f32x4*>(crops + (size_t)crop * IMG + tid * 4);
  // conv2 coordinates and the first weights: requested before conv1 so that their L2 latency runs under it
  const int r = lane & 15, q = lane >> 4;
  const int rt = wave >> 2, ct = wave & 3;
  const int slot = r >> 2, sub = r & 3;
  const int pix0 = (2 * rt + (sub >> 1)) * 12 + 2 * slot + (sub & 1);      // window (py = rt, px = slot)
  const uint4* wf = w2x + (size_t)ct * 3 * 64 + lane;                      // [tap][ct][plane][lane] 16 bytes
  constexpr int DEPTH = 4;
  uint4 ring[DEPTH][3];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) ring[d][pl] = wf[(size_t)(d * 4 * 3 + pl) * 64];
  float w1r[13];
#pragma unroll
  for (int s = 0; s < 13; ++s) w1r[s] = w1f[s * 64 + lane];
  const float bias1 = b1[lane & 31];
  __syncthreads();
  for (int tile = wave; tile < 18; tile += 16) {
    const int i = lane & 31, h = lane >> 5, j = lane & 31;
    const int wi = tile * 8 + (i >> 2);
    const int qy = wi / 12, qx = wi - qy * 12;
    const float* im = img + (2 * qy + ((i >> 1) & 1)) * 28 + 2 * qx + (i & 1);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = (k0 / 5) * 28 + k0 % 5;
      const int o1 = k1 < 25 ? (k1 / 5) * 28 + k1 % 5 : 0;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(im[h ? o1 : o0], w1r[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3])) + bias1;
      const __bf16 vh = (__bf16)v;            // round to nearest even at every level: the remainders are exact in f32
      const float r1 = v - (float)vh;
      const __bf16 vm = (__bf16)r1;
      const __bf16 vl = (__bf16)(r1 - (float)vm);
      const int o = p1x_offset(tile * 8 + 2 * g + h, j);
      *reinterpret_cast<__bf16*>(p1x + o) = vh;
      *reinterpret_cast<__bf16*>(p1x + P1X_PLANE + o) = vm;
      *reinterpret_cast<__bf16*>(p1x + 2 * P1X_PLANE + o) = vl;
    }
  }
  __syncthreads();
  // conv2: one v_mfma_f32_16x16x32_bf16 group per tap; lane (row r, channels 8 q .. 8 q + 7)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 25; ++tap) {
    const bf16x8 wh = __builtin_bit_cast(bf16x8, ring[tap % DEPTH][0]), wm = __builtin_bit_cast(bf16x8, ring[tap % DEPTH][1]),
                 wl = __builtin_bit_cast(bf16x8, ring[tap % DEPTH][2]);
    if (tap + DEPTH < 25) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ring[tap % DEPTH][pl] = wf[(size_t)((tap + DEPTH) * 4 * 3 + pl) * 64];
    }
    const int ky = tap / 5, kx = tap - ky * 5;
    const int pix = pix0 + ky * 12 + kx;
    const unsigned char* row = p1x + pix * 64 + ((q ^ ((pix >> 1) & 3)) << 4);
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(row), am = *reinterpret_cast<const bf16x8*>(row + P1X_PLANE),
                 al = *reinterpret_cast<const bf16x8*>(row + 2 * P1X_PLANE);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, acc, 0, 0, 0);   // small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, wh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, acc, 0, 0, 0);
  }
  // C/D map of 16x16: column = lane & 15, rows 4 q .. 4 q + 3 in the four registers = window q of this tile.
  // feat goes out in the operand order of the fc1 kernels: k = co * 16 + p of crop `crop`
  const int co = 16 * ct + r;
  const float v = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])) + b2[co];
  const int kk = co * 16 + 4 * rt + q;
  feat[((((size_t)(crop >> 4) * 64 + (kk >> 4)) * 4 + ((kk >> 2) & 3)) * 16 + (crop & 15)) * 4 + (kk & 3)] = v;
}

// fc1 + bias + ReLU, K split over the waves: a workgroup = 16 crops x 64 outputs, wave (column tile ct = w & 3, K quarter
// kq = w >> 2) accumulates 64 dependent v_mfma_f32_16x16x4_f32 (a 256-long chain cut in four);
// the four partial tiles meet in LDS, summed in a fixed order.  grid (ceil(n / 16), 8), 1024 threads.
__global__ __launch_bounds__(1024) void rec_fc1_ksplit_kernel(const float* __restrict__ feat_t, int n, const float* __restrict__ w_t,
                                                              const float* __restrict__ bias, float* __restrict__ hid) {
  __shared__ float part[4][16][65];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ctw = wave & 3, kq = wave >> 2;
  const int row0 = blockIdx.x * 16, ct = blockIdx.y * 4 + ctw;
  // both operands in OPERAND ORDER [tile][g of sixteen k][q][r][e]: lane (r, q) of group g holds k = 16 g + 4 q + e in element
  // e and multiplies it in its e-th instruction - every loaded byte is used, a wave load is 1 KB contiguous
  const f32x4* ap = reinterpret_cast<const f32x4*>(feat_t) + ((size_t)blockIdx.x * 64 + 16 * kq) * 64 + lane;
  const f32x4* bp = reinterpret_cast<const f32x4*>(w_t) + ((size_t)ct * 64 + 16 * kq) * 64 + lane;
  f32x4 a[16], b[16];   // the whole K quarter in flight (128 registers), then the 64-long chain
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    a[u] = ap[u * 64];
    b[u] = bp[u * 64];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) part[kq][4 * q + e][16 * ctw + r] = acc[e];
  __syncthreads();
  const int rr = tid >> 6, cc = tid & 63;
  const int row = row0 + rr, col = blockIdx.y * 64 + cc;
  const float v = (part[0][rr][cc] + part[1][rr][cc]) + (part[2][rr][cc] + part[3][rr][cc]) + bias[col];
  if (row < n) hid[(size_t)row * 512 + col] = fmaxf(v, 0.f);
}

// fc2 + bias + softmax(f64) + top-1 for small batches: 16 crops per workgroup, wave (column tile ct = w & 3 of the 64 padded
// columns, K quarter kq = w >> 2) runs a 32-long v_mfma_f32_16x16x4_f32 chain; partial tiles summed in LDS in a fixed
// order; then 16 lanes per crop as in rec_fc2_softmax_kernel.  w2s: rec_fc2_small_fragments.
__global__ __launch_bounds__(1024) void rec_fc2_small_kernel(const float* __restrict__ hid, int n, const float* __restrict__ w2s,
                                                             const float* __restrict__ b2, float* __restrict__ logits_out,
                                                             int32_t* __restrict__ labels, double* __restrict__ probs) {
  __shared__ __attribute__((aligned(16))) float part[4][16][68];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int ct = wave & 3, kq = wave >> 2;
  const int m0 = blockIdx.x * 16;
  {
    const int row = min(m0 + r, n - 1);   // rows past the batch re-read the last crop (never stored)
    // lane (r, q) of group g multiplies k = 128 kq + 16 g + 4 q + e in its e-th instruction (both operands alike)
    const f32x4* ap = reinterpret_cast<const f32x4*>(hid + (size_t)row * 512 + 128 * kq + 4 * q);
    const f32x4* bp = reinterpret_cast<const f32x4*>(w2s) + ((size_t)(ct * 4 + kq) * 8) * 64 + lane;   // [ct][kq][g][lane] float4
    f32x4 a[8], b[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      a[g] = ap[4 * g];
      b[g] = bp[g * 64];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][e], b[g][e], acc, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) part[kq][4 * q + e][16 * ct + r] = acc[e];
  }
  __syncthreads();
  if (tid >= 256) return;
  const int rr = tid >> 4, l16 = tid & 15;  // crop row of the tile, its columns 4 l16 .. 4 l16 + 3
  const int crop = m0 + rr;
  f32x4 v = (*reinterpret_cast<const f32x4*>(&part[0][rr][4 * l16]) + *reinterpret_cast<const f32x4*>(&part[1][rr][4 * l16])) +
            (*reinterpret_cast<const f32x4*>(&part[2][rr][4 * l16]) + *reinterpret_cast<const f32x4*>(&part[3][rr][4 * l16]));
  v += *reinterpret_cast<const f32x4*>(b2 + 4 * l16);
  if (l16 == 15) v[2] = v[3] = -INFINITY;  // columns 62, 63 are padding
  if (crop < n && logits_out) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * l16 + e < 62) logits_out[(size_t)crop * 62 + 4 * l16 + e] = v[e];
  }
  if (!labels && !probs) return;
  float mx = v[0];  // first index of the maximum
  int arg = 4 * l16;
#pragma unroll
  for (int e = 1; e < 4; ++e)
    if (v[e] > mx) {
      mx = v[e];
      arg = 4 * l16 + e;
    }
#pragma unroll
  for (int k = 1; k <= 8; k <<= 1) {
    const float ov = __shfl_xor(mx, k, 64);
    const int oa = __shfl_xor(arg, k, 64);
    if (ov > mx || (ov == mx && oa < arg)) {
      mx = ov;
      arg = oa;
    }
  }
  double e = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) e += exp((double)v[c] - (double)mx);  // exp(-inf) = 0 for the padding
#pragma unroll
  for (int k = 1; k <= 8; k <<= 1) e += __shfl_xor(e, k, 64);
  if (l16 == 0 && crop < n) {
    if (labels) labels[crop] = arg;
    if (probs) probs[crop] = 1.0 / e;
  }
}

// fc2 (512 -> 62, as 64 columns with two zero ones) + bias, then softmax(-1, f64) and its top-1: 64 crops per
// 1024-thread workgroup.  Wave w multiplies the 32 x 32 tile (crops 32 ((w >> 1) & 1) .., columns 32 (w & 1) ..) over
// the K quarter w >> 2 (16 waves: the dependent MFMA chain is 64 long instead of 256 - this kernel is latency-bound at
// every batch size); the four partial tiles are added in a fixed order ((q0 + q1) + (q2 + q3)) + bias, so a crop's
// logits do not depend on the batch around it.  Both operands come straight from global memory: the hidden rows as
// 16-byte pieces, the weights host-arranged in fragment order (1 KiB contiguous per wave load).  Tail: 16 lanes per
// crop, 4 columns each: first index of the maximum, exp in f64 (mod.rs:55 softmax(-1, Kind::Double), utils.rs:28-43).
__global__ __launch_bounds__(1024) void rec_fc2_softmax_kernel(const float* __restrict__ hid, int n, const float* __restrict__ w2f,
                                                               const float* __restrict__ b2, float* __restrict__ logits_out,
                                                               int32_t* __restrict__ labels, double* __restrict__ probs) {
  __shared__ __attribute__((aligned(16))) float part[4][64][68];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int kq = wave >> 2, rt = (wave >> 1) & 1, ct = wave & 1;
  const int m0 = blockIdx.x * 64;
  {
    const int row = min(m0 + 32 * rt + j, n - 1);  // rows past the batch re-read the last crop (never stored)
    const f32x4* ap = reinterpret_cast<const f32x4*>(hid + (size_t)row * 512 + 128 * kq + 4 * h);
    const f32x4* bp = reinterpret_cast<const f32x4*>(w2f) + ((size_t)ct * 64 + 16 * kq) * 64 + lane;  // [ct][g][lane] float4
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += 8) {
      f32x4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = ap[2 * (g0 + u)];
        b[u] = bp[(g0 + u) * 64];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[kq][32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h][32 * ct + j] = acc[e];
  }
  __syncthreads();
  const int r = tid >> 4, l16 = tid & 15;  // crop row of the tile, its columns 4 l16 .. 4 l16 + 3
  const int crop = m0 + r;
  f32x4 v = (*reinterpret_cast<const f32x4*>(&part[0][r][4 * l16]) + *reinterpret_cast<const f32x4*>(&part[1][r][4 * l16])) +
            (*reinterpret_cast<const f32x4*>(&part[2][r][4 * l16]) + *reinterpret_cast<const f32x4*>(&part[3][r][4 * l16]));
  v += *reinterpret_cast<const f32x4*>(b2 + 4 * l16);
  if (l16 == 15) v[2] = v[3] = -INFINITY;  // columns 62, 63 are padding
  if (crop < n && logits_out) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * l16 + e < 62) logits_out[(size_t)crop * 62 + 4 * l16 + e] = v[e];
  }
  if (!labels && !probs) return;
  float mx = v[0];  // first index of the maximum
  int arg = 4 * l16;
#pragma unroll
  for (int e = 1; e < 4; ++e)
    if (v[e] > mx) {
      mx = v[e];
      arg = 4 * l16 + e;
    }
#pragma unroll
  for (int k = 1; k <= 8; k <<= 1) {
    const float ov = __shfl_xor(mx, k, 64);
    const int oa = __shfl_xor(arg, k, 64);
    if (ov > mx || (ov == mx && oa < arg)) {
      mx = ov;
      arg = oa;
    }
  }
  double e = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) e += exp((double)v[c] - (double)mx);  // exp(-inf) = 0 for the padding
#pragma unroll
  for (int k = 1; k <= 8; k <<= 1) e += __shfl_xor(e, k, 64);
  if (l16 == 0 && crop < n) {
    if (labels) labels[crop] = arg;
    if (probs) probs[crop] = 1.0 / e;
  }
}

}  // namespace


// conv1 [32][1][5][5] -> [13 steps][64 lanes]: lane (j = l & 31, h = l >> 5) of step s holds w[j][k = 2 s + h] (0 for k = 25)
std::vector<float> rec_conv1_fragments(const float* w) {
  std::vector<float> f(13 * 64, 0.f);
  for (int s = 0; s < 13; ++s)
    for (int l = 0; l < 64; ++l) {
      const int k = 2 * s + (l >> 5);
      if (k < 25) f[s * 64 + l] = w[(l & 31) * 25 + k];
    }
  return f;
}

// conv2 [64][32][5][5] -> [25 taps][2 column tiles][4 g][64 lanes][4]: element e of lane (j, h) is
// w[co = 32 ct + j][ci = 8 g + 4 h + e][tap] - the B operand of the e-th of four MFMAs of K group g
std::vector<float> rec_conv2_fragments(const float* w) {
  std::vector<float> f((size_t)25 * 2 * 4 * 64 * 4);
  for (int tap = 0; tap < 25; ++tap)
    for (int ct = 0; ct < 2; ++ct)
      for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int co = 32 * ct + (l & 31), ci = 8 * g + 4 * (l >> 5) + e;
            f[((((size_t)tap * 2 + ct) * 4 + g) * 64 + l) * 4 + e] = w[((size_t)co * 32 + ci) * 25 + tap];
          }
  return f;
}

// fc1.weight [512][1024] in the operand order of rec_fc1_ksplit_kernel: [32 column tiles][64 g][4 q][16 c][4 e] holds
// w[16 ct + c][16 g + 4 q + e]
std::vector<float> rec_fc1_small_weights(const float* w) {
  std::vector<float> f((size_t)512 * 1024);
  for (int ct = 0; ct < 32; ++ct)
    for (int g = 0; g < 64; ++g)
      for (int q = 0; q < 4; ++q)
        for (int c = 0; c < 16; ++c)
          for (int e = 0; e < 4; ++e)
            f[((((size_t)ct * 64 + g) * 4 + q) * 16 + c) * 4 + e] = w[(size_t)(16 * ct + c) * 1024 + 16 * g + 4 * q + e];
  return f;
}

// conv2 [64][32][5][5] for the split-bf16 16x16x32 form: [25 taps][4 ct][3 planes hi / mid / lo][64 lanes][8 bf16]; lane
// (c = l & 15, q = l >> 4) holds w[16 ct + c][ci = 8 q + j][tap] in element j
std::vector<uint16_t> rec_conv2_small_x3_fragments(const float* w) {
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
  std::vector<uint16_t> f((size_t)25 * 4 * 3 * 64 * 8);
  for (int tap = 0; tap < 25; ++tap)
    for (int ct = 0; ct < 4; ++ct)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const float x = w[((size_t)(16 * ct + (l & 15)) * 32 + 8 * (l >> 4) + j) * 25 + tap];
          const uint16_t h = bf(x);
          const float r1 = x - up(h);
          const uint16_t m = bf(r1);
          const size_t o = ((((size_t)tap * 4 + ct) * 3) * 64 + l) * 8 + j;
          f[o] = h;
          f[o + 64 * 8] = m;
          f[o + 2 * 64 * 8] = bf(r1 - up(m));
        }
  return f;
}

// fc2 [62][512] for rec_fc2_small_kernel: [4 ct][4 kq][8 g][64 lanes][4]; element e of lane (c = l & 15, q = l >> 4) is
// w[16 ct + c][128 kq + 16 g + 4 q + e] (0 for rows 62, 63)
std::vector<float> rec_fc2_small_fragments(const float* w) {
  std::vector<float> f((size_t)4 * 4 * 8 * 64 * 4, 0.f);
  for (int ct = 0; ct < 4; ++ct)
    for (int kq = 0; kq < 4; ++kq)
      for (int g = 0; g < 8; ++g)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            const int o = 16 * ct + (l & 15), k = 128 * kq + 16 * g + 4 * (l >> 4) + e;
            if (o < 62) f[((((size_t)ct * 4 + kq) * 8 + g) * 64 + l) * 4 + e] = w[(size_t)o * 512 + k];
          }
  return f;
}

void launch_rec_small(const RecWeights& w, const float* crops, int n, float* feat, float* hid, float* logits, int32_t* labels,
                      double* probs, hipStream_t s, int stage) {
  if (n <= 0) return;
  if (stage == 0) hipLaunchKernelGGL(rec_conv_small_x3_kernel, dim3(n), dim3(1024), 0, s, crops, n, w.c1f, w.c1b, static_cast<const uint4*>(w.c2x), w.c2b, feat);
  else if (stage == 1) hipLaunchKernelGGL(rec_fc1_ksplit_kernel, dim3((n + 15) / 16, 8), dim3(1024), 0, s, feat, n, w.f1s, w.f1b, hid);
  else if (logits || labels || probs) hipLaunchKernelGGL(rec_fc2_small_kernel, dim3((n + 15) / 16), dim3(1024), 0, s, hid, n, w.f2s, w.f2b, logits, labels, probs);
  OCR_HIP(hipGetLastError());
}

constexpr int kRecSmallBatch = 1024;  // up to here the chain-latency-optimised kernels; beyond, the throughput ones

bool rec_small_batch(int n) { return n <= kRecSmallBatch; }

int rec_crops_per_block(int n) { return n <= 768 ? 1 : n <= 3072 ? 2 : 4; }

void launch_rec_conv(const RecWeights& w, const float* crops, int n, float* feat, hipStream_t s) {
  if (n <= 0) return;
  // few crops: one per workgroup so that every CU gets work; many: four, so that a tap's weights serve four row tiles
  const int t = rec_crops_per_block(n);
  const dim3 grid((n + t - 1) / t), block(256);
  if (t == 1) hipLaunchKernelGGL(rec_conv_kernel<1>, grid, block, 0, s, crops, n, w.c1f, w.c1b, w.c2f, w.c2b, feat);
  else if (t == 2) hipLaunchKernelGGL(rec_conv_kernel<2>, grid, block, 0, s, crops, n, w.c1f, w.c1b, w.c2f, w.c2b, feat);
  else hipLaunchKernelGGL(rec_conv_kernel<4>, grid, block, 0, s, crops, n, w.c1f, w.c1b, w.c2f, w.c2b, feat);
  OCR_HIP(hipGetLastError());
}

// fc2 [62][512] -> [2 column tiles][64 g][64 lanes][4]: element e of lane (j, h) is w[32 ct + j][8 g + 4 h + e] (0 for rows 62, 63)
std::vector<float> rec_fc2_fragments(const float* w) {
  std::vector<float> f((size_t)2 * 64 * 64 * 4, 0.f);
  for (int ct = 0; ct < 2; ++ct)
    for (int g = 0; g < 64; ++g)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
          const int o = 32 * ct + (l & 31), k = 8 * g + 4 * (l >> 5) + e;
          if (o < 62) f[(((size_t)ct * 64 + g) * 64 + l) * 4 + e] = w[(size_t)o * 512 + k];
        }
  return f;
}

void launch_rec_fc2_softmax(const RecWeights& w, const float* hid, int n, float* logits, int32_t* labels, double* probs, hipStream_t s) {
  if (n <= 0 || (!logits && !labels && !probs)) return;
  hipLaunchKernelGGL(rec_fc2_softmax_kernel, dim3((n + 63) / 64), dim3(1024), 0, s, hid, n, w.f2f, w.f2b, logits, labels, probs);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
