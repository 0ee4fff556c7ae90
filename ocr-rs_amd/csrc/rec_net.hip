// Recognition network, one fused launch: one 1024-thread workgroup per 28x28 crop.
//   /root/reference/src/char_recognition/model.rs:27-39
//     view[-1,1,28,28] -> conv5x5(1->32)+b -> maxpool2 -> conv5x5(32->64)+b -> maxpool2
//     -> view[-1,1024] -> fc1(1024->512)+b -> ReLU -> (dropout: identity in eval) -> fc2(512->62)+b
//   /root/reference/src/char_recognition/mod.rs:53-56 + utils.rs:28-43
//     softmax(-1, Kind::Double) and top-1 (label index into "A-Za-z0-9").
// All intermediates stay in LDS; weights (2.4 MB) are served from L2 / Infinity Cache.
#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
  return v;
}

constexpr int REC_THREADS = 1024;  // 16 waves per crop: at 256 crops every CU holds one crop

__global__ __launch_bounds__(REC_THREADS) void rec_forward_kernel(RecWeights w, const float* __restrict__ crops,
                                                                  float* __restrict__ logits_out,
                                                                  int32_t* __restrict__ labels, double* __restrict__ probs) {
  __shared__ __attribute__((aligned(16))) float img[28 * 28];
  __shared__ __attribute__((aligned(16))) float w1[32 * 25];
  __shared__ __attribute__((aligned(16))) float p1[32][12][12];   // after conv1 + pool
  __shared__ __attribute__((aligned(16))) float feat[1024];       // [c][h][w] flatten (view[-1,1024])
  __shared__ __attribute__((aligned(16))) float hid[512];
  __shared__ float lg[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int crop = blockIdx.x;
  const float* src = crops + (size_t)crop * 784;
  if (tid < 784) img[tid] = src[tid];
  if (tid < 800) w1[tid] = w.c1w[tid];
  // conv2 weights of this thread's output channel, first input channel: requested now,
  // consumed after conv1 (their L2 latency hides behind it)
  const int co2 = tid >> 4;                   // conv2: thread = (output channel, pooled position)
  const int pp = tid & 15, py = pp >> 2, px = pp & 3;
  const float* wc = w.c2w + (size_t)co2 * 800;
  float wcur[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) wcur[k] = wc[k];
  __syncthreads();

  // conv1 (valid 5x5) fused with 2x2 max pool: 32 x 12 x 12 pooled outputs
  for (int o = tid; o < 32 * 144; o += REC_THREADS) {
    const int co = o / 144, rem = o - co * 144, qy = rem / 12, qx = rem - qy * 12;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const float wv = w1[co * 25 + kh * 5 + kw];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx)
            acc[dy][dx] = fmaf(img[(2 * qy + dy + kh) * 28 + 2 * qx + dx + kw], wv, acc[dy][dx]);
      }
    const float b = w.c1b[co];
    p1[co][qy][qx] = fmaxf(fmaxf(acc[0][0] + b, acc[0][1] + b), fmaxf(acc[1][0] + b, acc[1][1] + b));
  }
  __syncthreads();

  // conv2 (valid 5x5, 32->64) fused with its 2x2 max pool: the thread owns the 2x2 conv outputs
  // under one pooled pixel; per input channel it reads a 6x6 patch and does 100 FMAs while the
  // next channel's 25 weights are in flight.
  {
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int ci = 0; ci < 32; ++ci) {
      float wnext[25];
      const float* wn = wc + (ci + 1 < 32 ? ci + 1 : ci) * 25;
#pragma unroll
      for (int k = 0; k < 25; ++k) wnext[k] = wn[k];
      float patch[6][6];
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) patch[a][b] = p1[ci][2 * py + a][2 * px + b];
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
          const float wv = wcur[kh * 5 + kw];
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) acc[dy][dx] = fmaf(patch[dy + kh][dx + kw], wv, acc[dy][dx]);
        }
#pragma unroll
      for (int k = 0; k < 25; ++k) wcur[k] = wnext[k];
    }
    const float bb = w.c2b[co2];
    feat[co2 * 16 + pp] = fmaxf(fmaxf(acc[0][0] + bb, acc[0][1] + bb), fmaxf(acc[1][0] + bb, acc[1][1] + bb));
  }
  __syncthreads();

  // fc1 + ReLU: one wave per output row (coalesced 4 KB weight rows), four rows in flight
  f32x4 fv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) fv[k] = *reinterpret_cast<const f32x4*>(feat + k * 256 + lane * 4);
  for (int o0 = wave * 4; o0 < 512; o0 += 64) {
    f32x4 a[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) a[u][k] = *reinterpret_cast<const f32x4*>(w.f1w + (size_t)(o0 + u) * 1024 + k * 256 + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[u][k][0] * fv[k][0] + a[u][k][1] * fv[k][1] + a[u][k][2] * fv[k][2] + a[u][k][3] * fv[k][3];
      s = wave_sum(s);
      if (lane == 0) hid[o0 + u] = fmaxf(s + w.f1b[o0 + u], 0.f);
    }
  }
  __syncthreads();
  for (int o = wave; o < 62; o += 16) {
    const float* wr = w.f2w + (size_t)o * 512;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(wr + k * 256 + lane * 4);
      const f32x4 f = *reinterpret_cast<const f32x4*>(hid + k * 256 + lane * 4);
      s += a[0] * f[0] + a[1] * f[1] + a[2] * f[2] + a[3] * f[3];
    }
    s = wave_sum(s);
    if (lane == 0) lg[o] = s + w.f2b[o];
  }
  __syncthreads();
  if (wave == 0) {
    const float v = lane < 62 ? lg[lane] : -INFINITY;
    if (logits_out && lane < 62) logits_out[(size_t)crop * 62 + lane] = v;
    if (labels || probs) {
      // top-1 of softmax(-1, f64): first index of the maximum
      float mx = v;
      int arg = lane;
#pragma unroll
      for (int k = 32; k >= 1; k >>= 1) {
        const float ov = __shfl_xor(mx, k, 64);
        const int oa = __shfl_xor(arg, k, 64);
        if (ov > mx || (ov == mx && oa < arg)) {
          mx = ov;
          arg = oa;
        }
      }
      double e = lane < 62 ? exp((double)v - (double)mx) : 0.0;
#pragma unroll
      for (int k = 32; k >= 1; k >>= 1) e += __shfl_xor(e, k, 64);
      if (lane == 0) {
        if (labels) labels[crop] = arg;
        if (probs) probs[crop] = 1.0 / e;
      }
    }
  }
}

}  // namespace

void launch_rec_forward(const RecWeights& w, const float* crops, int n, float* logits, int32_t* labels,
                        double* probs, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(rec_forward_kernel, dim3(n), dim3(REC_THREADS), 0, s, w, crops, logits, labels, probs);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
