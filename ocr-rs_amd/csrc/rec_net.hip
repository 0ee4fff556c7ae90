// Recognition network, one fused launch: one workgroup per 28x28 crop.
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

__global__ __launch_bounds__(256) void rec_forward_kernel(RecWeights w, const float* __restrict__ crops,
                                                          float* __restrict__ logits_out,
                                                          int32_t* __restrict__ labels, double* __restrict__ probs) {
  __shared__ __attribute__((aligned(16))) float img[28 * 28];
  __shared__ __attribute__((aligned(16))) float w1[32 * 25];
  __shared__ __attribute__((aligned(16))) float p1[32][12][12];   // after conv1 + pool
  __shared__ __attribute__((aligned(16))) float c2[64][8][8];     // conv2 output (pre-pool)
  __shared__ __attribute__((aligned(16))) float feat[1024];       // [c][h][w] flatten (view[-1,1024])
  __shared__ __attribute__((aligned(16))) float hid[512];
  __shared__ float lg[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int crop = blockIdx.x;
  const float* src = crops + (size_t)crop * 784;
  for (int i = tid; i < 784; i += 256) img[i] = src[i];
  for (int i = tid; i < 800; i += 256) w1[i] = w.c1w[i];
  __syncthreads();

  // conv1 (valid 5x5) fused with 2x2 max pool: 32 x 12 x 12 pooled outputs
  for (int o = tid; o < 32 * 144; o += 256) {
    const int co = o / 144, rem = o - co * 144, py = rem / 12, px = rem - py * 12;
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
            acc[dy][dx] = fmaf(img[(2 * py + dy + kh) * 28 + 2 * px + dx + kw], wv, acc[dy][dx]);
      }
    const float b = w.c1b[co];
    p1[co][py][px] = fmaxf(fmaxf(acc[0][0] + b, acc[0][1] + b), fmaxf(acc[1][0] + b, acc[1][1] + b));
  }
  __syncthreads();

  // conv2 (valid 5x5, 32->64): thread = (co, 4x4 block of the 8x8 output)
  {
    const int co = tid >> 2, blk = tid & 3;
    const int oy0 = (blk >> 1) * 4, ox0 = (blk & 1) * 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const float* wc = w.c2w + (size_t)co * 800;
    for (int ci = 0; ci < 32; ++ci) {
#pragma unroll
      for (int kh = 0; kh < 5; ++kh) {
        float wv[5];
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) wv[kw] = wc[ci * 25 + kh * 5 + kw];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float row[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) row[k] = p1[ci][oy0 + a + kh][ox0 + k];
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) acc[a][b] = fmaf(row[b + kw], wv[kw], acc[a][b]);
        }
      }
    }
    const float bb = w.c2b[co];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) c2[co][oy0 + a][ox0 + b] = acc[a][b] + bb;
  }
  __syncthreads();
  for (int o = tid; o < 1024; o += 256) {
    const int co = o >> 4, py = (o >> 2) & 3, px = o & 3;
    feat[o] = fmaxf(fmaxf(c2[co][2 * py][2 * px], c2[co][2 * py][2 * px + 1]),
                    fmaxf(c2[co][2 * py + 1][2 * px], c2[co][2 * py + 1][2 * px + 1]));
  }
  __syncthreads();

  // fc1 + ReLU: one wave per output, K split over the lanes (coalesced weight rows)
  for (int o = wave; o < 512; o += 4) {
    const float* wr = w.f1w + (size_t)o * 1024;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(wr + k * 256 + lane * 4);
      const f32x4 f = *reinterpret_cast<const f32x4*>(feat + k * 256 + lane * 4);
      s += a[0] * f[0] + a[1] * f[1] + a[2] * f[2] + a[3] * f[3];
    }
    s = wave_sum(s);
    if (lane == 0) hid[o] = fmaxf(s + w.f1b[o], 0.f);
  }
  __syncthreads();
  for (int o = wave; o < 62; o += 4) {
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
  hipLaunchKernelGGL(rec_forward_kernel, dim3(n), dim3(256), 0, s, w, crops, logits, labels, probs);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
