// preprocess_image as HIP image kernels.
//   /root/reference/src/image_ops.rs:188-220: decode -> RGBA -> DynamicImage::resize(W, H, Triangle)
//   (aspect preserving) -> to_luma -> zero-pad to W x H; adjust = resized / original.
// The sampling arithmetic is that of the un-vendored crate image 0.23.11 (imageops::resize =
// vertical_sample then horizontal_sample, f32 weights normalised per output index, every pass
// truncated to u8; luma = trunc(0.2126 R + 0.7152 G + 0.0722 B)).  Weight tables are built on the
// host with the same f32 operation order; the kernels accumulate with separately rounded multiply
// and add (no FMA contraction), so the result equals oracle/preprocess_oracle.py bit for bit.
// HBM-bound: one pass reads the RGBA source, the second writes the padded gray frame (+ f32 copy).
#include <cmath>
#include <vector>

#include "common.hpp"

namespace ocr {
namespace {

struct AxisTable {       // per output index: first source index and up to maxk normalised weights
  std::vector<int> left;
  std::vector<int> count;
  std::vector<float> w;  // [out][maxk]
  int maxk = 0;
};

// image 0.23.11 sample.rs, Triangle filter (support 1.0); every step in f32 like the crate
AxisTable build_axis(int in_size, int out_size) {
  AxisTable t;
  const float ratio = (float)in_size / (float)out_size;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float support = 1.0f * sratio;
  std::vector<std::vector<float>> ws(out_size);
  t.left.resize(out_size);
  t.count.resize(out_size);
  for (int o = 0; o < out_size; ++o) {
    float inp = ((float)o + 0.5f) * ratio;
    long left = (long)std::floor(inp - support);
    left = std::min<long>(std::max<long>(left, 0), in_size - 1);
    long right = (long)std::ceil(inp + support);
    right = std::min<long>(std::max<long>(right, left + 1), in_size);
    inp = inp - 0.5f;
    float sum = 0.0f;
    for (long i = left; i < right; ++i) {
      const float x = ((float)i - inp) / sratio;
      const float ax = std::fabs(x);
      const float w = ax < 1.0f ? 1.0f - ax : 0.0f;
      ws[o].push_back(w);
      sum += w;
    }
    for (float& w : ws[o]) w /= sum;
    t.left[o] = (int)left;
    t.count[o] = (int)ws[o].size();
    t.maxk = std::max(t.maxk, t.count[o]);
  }
  t.w.assign((size_t)out_size * t.maxk, 0.0f);
  for (int o = 0; o < out_size; ++o)
    for (int k = 0; k < t.count[o]; ++k) t.w[(size_t)o * t.maxk + k] = ws[o][k];
  return t;
}

__device__ __forceinline__ unsigned char to_u8(float t) { return (unsigned char)fminf(fmaxf(t, 0.f), 255.f); }  // clamp, truncate

// vertical_sample: src h x w RGBA -> tmp nh x w RGBA; one thread per output pixel (4 channels)
__global__ __launch_bounds__(256) void resize_vertical_kernel(const uchar4* __restrict__ src, uchar4* __restrict__ tmp, int w,
                                                              int nh, const int* __restrict__ left, const int* __restrict__ cnt,
                                                              const float* __restrict__ wts, int maxk) {
  const int x = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (x >= w || oy >= nh) return;
  const int l = left[oy], c = cnt[oy];
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  for (int k = 0; k < c; ++k) {
    const uchar4 p = src[(size_t)(l + k) * w + x];
    const float wk = wts[oy * maxk + k];
    t0 = __fadd_rn(t0, __fmul_rn((float)p.x, wk));
    t1 = __fadd_rn(t1, __fmul_rn((float)p.y, wk));
    t2 = __fadd_rn(t2, __fmul_rn((float)p.z, wk));
    t3 = __fadd_rn(t3, __fmul_rn((float)p.w, wk));
  }
  tmp[(size_t)oy * w + x] = make_uchar4(to_u8(t0), to_u8(t1), to_u8(t2), to_u8(t3));
}

// horizontal_sample + to_luma + zero padding: tmp nh x w RGBA -> gray H x W u8 (and f32 copy)
__global__ __launch_bounds__(256) void resize_horizontal_luma_kernel(const uchar4* __restrict__ tmp, unsigned char* __restrict__ gray,
                                                                     float* __restrict__ gray_f32, int w, int nw, int nh, int W,
                                                                     int H, const int* __restrict__ left, const int* __restrict__ cnt,
                                                                     const float* __restrict__ wts, int maxk) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (ox >= W || oy >= H) return;
  unsigned char g = 0;  // zero padding right of / below the resized image (image_ops.rs:204-214)
  if (ox < nw && oy < nh) {
    const int l = left[ox], c = cnt[ox];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < c; ++k) {
      const uchar4 p = tmp[(size_t)oy * w + l + k];
      const float wk = wts[ox * maxk + k];
      t0 = __fadd_rn(t0, __fmul_rn((float)p.x, wk));
      t1 = __fadd_rn(t1, __fmul_rn((float)p.y, wk));
      t2 = __fadd_rn(t2, __fmul_rn((float)p.z, wk));
    }
    const float r = (float)to_u8(t0), gg = (float)to_u8(t1), b = (float)to_u8(t2);
    const float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.2126f, r), __fmul_rn(0.7152f, gg)), __fmul_rn(0.0722f, b));
    g = (unsigned char)lum;  // NumCast: truncation
  }
  gray[(size_t)oy * W + ox] = g;
  if (gray_f32) gray_f32[(size_t)oy * W + ox] = (float)g;  // convert_image_to_tensor(..).to_kind(Float): raw 0..255
}

}  // namespace

void resize_dimensions(int width, int height, int nwidth, int nheight, int* ow, int* oh) {
  const unsigned long long ratio = (unsigned long long)width * nheight, nratio = (unsigned long long)nwidth * height;
  const bool use_width = nratio <= ratio;
  unsigned long long inter = use_width ? (unsigned long long)height * nwidth / width : (unsigned long long)width * nheight / height;
  if (inter < 1) inter = 1;
  *ow = use_width ? nwidth : (int)inter;
  *oh = use_width ? (int)inter : nheight;
}

// rgba_dev: h x w x 4 u8 on the device.  scratch must hold nh*w*4 bytes (tmp) + the tables.
void launch_preprocess(const unsigned char* rgba_dev, int w, int h, int W, int H, unsigned char* gray_dev, float* gray_f32_dev,
                       void* scratch, size_t scratch_bytes, double* adj_xy, hipStream_t s) {
  if (w < 1 || h < 1 || W < 1 || H < 1) fail(OCR_ERR_INVALID, "preprocess: bad dimensions");
  int nw, nh;
  resize_dimensions(w, h, W, H, &nw, &nh);
  const AxisTable ty = build_axis(h, nh), tx = build_axis(w, nw);
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t o_tmp = 0, o_ly = al((size_t)nh * w * 4), o_cy = o_ly + al((size_t)nh * 4), o_wy = o_cy + al((size_t)nh * 4);
  const size_t o_lx = o_wy + al(ty.w.size() * 4), o_cx = o_lx + al((size_t)nw * 4), o_wx = o_cx + al((size_t)nw * 4);
  const size_t total = o_wx + al(tx.w.size() * 4);
  if (total > scratch_bytes) fail(OCR_ERR_INTERNAL, "preprocess: scratch of %zu bytes needed, %zu given", total, scratch_bytes);
  char* sc = static_cast<char*>(scratch);
  OCR_HIP(hipMemcpyAsync(sc + o_ly, ty.left.data(), (size_t)nh * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipMemcpyAsync(sc + o_cy, ty.count.data(), (size_t)nh * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipMemcpyAsync(sc + o_wy, ty.w.data(), ty.w.size() * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipMemcpyAsync(sc + o_lx, tx.left.data(), (size_t)nw * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipMemcpyAsync(sc + o_cx, tx.count.data(), (size_t)nw * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipMemcpyAsync(sc + o_wx, tx.w.data(), tx.w.size() * 4, hipMemcpyHostToDevice, s));
  OCR_HIP(hipStreamSynchronize(s));  // the host tables go out of scope
  hipLaunchKernelGGL(resize_vertical_kernel, dim3((w + 255) / 256, nh), dim3(256), 0, s, reinterpret_cast<const uchar4*>(rgba_dev),
                     reinterpret_cast<uchar4*>(sc + o_tmp), w, nh, reinterpret_cast<const int*>(sc + o_ly),
                     reinterpret_cast<const int*>(sc + o_cy), reinterpret_cast<const float*>(sc + o_wy), ty.maxk);
  OCR_HIP(hipGetLastError());
  hipLaunchKernelGGL(resize_horizontal_luma_kernel, dim3((W + 255) / 256, H), dim3(256), 0, s,
                     reinterpret_cast<const uchar4*>(sc + o_tmp), gray_dev, gray_f32_dev, w, nw, nh, W, H,
                     reinterpret_cast<const int*>(sc + o_lx), reinterpret_cast<const int*>(sc + o_cx),
                     reinterpret_cast<const float*>(sc + o_wx), tx.maxk);
  OCR_HIP(hipGetLastError());
  if (adj_xy) {
    adj_xy[0] = (double)nw / (double)w;  // image_ops.rs:200-202
    adj_xy[1] = (double)nh / (double)h;
  }
}

size_t preprocess_scratch_bytes(int w, int h, int W, int H) {
  int nw, nh;
  resize_dimensions(w, h, W, H, &nw, &nh);
  // tmp + generous room for the weight tables (support grows with the down-scaling ratio)
  const size_t ky = (size_t)(2.0 * std::max(1.0, (double)h / nh) + 3), kx = (size_t)(2.0 * std::max(1.0, (double)w / nw) + 3);
  return (size_t)nh * w * 4 + (size_t)nh * (ky + 2) * 4 + (size_t)nw * (kx + 2) * 4 + 16 * 256;
}

}  // namespace ocr
