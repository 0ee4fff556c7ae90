// Winograd F(2x2, 3x3) transforms for the deep 3x3 s1 p1 convs of the detector trunk (layer3 / layer4,
// /root/reference/src/text_detection/model.rs:40-55): y = A^T [ (G g G^T) . (B^T d B) ] A per 2x2 output tile,
// 16 multiplies instead of 36 per (tile, cin, cout).  The 16 element-wise products over all tiles and channels
// are 16 independent GEMMs [T x C] x [C x K] and run on the matrix cores through conv_igemm's batched mode;
// the kernels here are the HBM-bound ends: d -> V = B^T d B and M -> Y = A^T M A (+ folded BN, residual, ReLU).
//
//   B^T = | 1  0 -1  0 |     G = | 1    0    0  |     A^T = | 1 1  1  0 |
//         | 0  1  1  0 |         | 1/2  1/2  1/2|           | 0 1 -1 -1 |
//         | 0 -1  1  0 |         | 1/2 -1/2  1/2|
//         | 0  1  0 -1 |         | 0    0    1  |
//
// Only additions and subtractions of f32 values happen here (the 1/2 factors live in the host-side weight
// transform, engine.hip), so the results differ from the direct convolution by summation order / a few ulp.
#include "common.hpp"

namespace ocr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// one thread = one tile x 4 channels; consecutive threads walk the channels of a tile (coalesced 16-byte
// accesses on both sides).  v layout [16][T][C].
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ x, float* __restrict__ v, int H, int W,
                                                             int C, int th, int tw, long long T) {
  const int c4n = C >> 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * c4n) return;
  const long long t = gid / c4n;
  const int c = (int)(gid - t * c4n) * 4;
  const int tx = (int)(t % tw);
  const long long r = t / tw;
  const int ty = (int)(r % th);
  const long long n = r / th;
  const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
  f32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = y0 + i, xx = x0 + j;
      const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      d[i][j] = ok ? *reinterpret_cast<const f32x4*>(x + ((n * H + yy) * W + xx) * C + c) : zero;
    }
  f32x4 r4[4][4];  // B^T d
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r4[0][j] = d[0][j] - d[2][j];
    r4[1][j] = d[1][j] + d[2][j];
    r4[2][j] = d[2][j] - d[1][j];
    r4[3][j] = d[1][j] - d[3][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (B^T d) B
    const f32x4 v0 = r4[i][0] - r4[i][2];
    const f32x4 v1 = r4[i][1] + r4[i][2];
    const f32x4 v2 = r4[i][2] - r4[i][1];
    const f32x4 v3 = r4[i][1] - r4[i][3];
    float* o = v + ((size_t)(4 * i) * T + t) * C + c;
    const size_t step = (size_t)T * C;
    *reinterpret_cast<f32x4*>(o) = v0;
    *reinterpret_cast<f32x4*>(o + step) = v1;
    *reinterpret_cast<f32x4*>(o + 2 * step) = v2;
    *reinterpret_cast<f32x4*>(o + 3 * step) = v3;
  }
}

// one thread = one tile x 4 output channels.  m layout [16][T][K]; y NHWC.
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                              const float* __restrict__ bias, const float* __restrict__ residual,
                                                              int relu, float* __restrict__ y, int H, int W, int K, int th,
                                                              int tw, long long T) {
  const int k4n = K >> 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * k4n) return;
  const long long t = gid / k4n;
  const int k = (int)(gid - t * k4n) * 4;
  const int tx = (int)(t % tw);
  const long long r = t / tw;
  const int ty = (int)(r % th);
  const long long n = r / th;
  const size_t step = (size_t)T * K;
  const float* src = m + (size_t)t * K + k;
  f32x4 a[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[i][j] = *reinterpret_cast<const f32x4*>(src + (size_t)(4 * i + j) * step);
  f32x4 u[2][4];  // A^T M
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[0][j] = a[0][j] + a[1][j] + a[2][j];
    u[1][j] = a[1][j] - a[2][j] - a[3][j];
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4*>(scale + k);
  if (bias) bi = *reinterpret_cast<const f32x4*>(bias + k);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x4 o0 = u[p][0] + u[p][1] + u[p][2];
    const f32x4 o1 = u[p][1] - u[p][2] - u[p][3];
    const int yy = 2 * ty + p;
    if (yy >= H) continue;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int xx = 2 * tx + q;
      if (xx >= W) continue;
      const size_t o = ((n * H + yy) * W + xx) * (size_t)K + k;
      f32x4 val = (q ? o1 : o0) * sc + bi;
      if (residual) val += *reinterpret_cast<const f32x4*>(residual + o);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f);
      }
      *reinterpret_cast<f32x4*>(y + o) = val;
    }
  }
}

// ---- F(4x4, 3x3): 36 multiplies per 4x4 output tile (2.25 per output instead of 4) for the grids where the transforms are
// cheap beside the GEMMs (layer4's 512 channels at H/32).  Lavin & Gray's matrices; the weight transform carries the
// fractions of G (engine.hip, f64 rounded once), so the device side multiplies by 2, 4, 5 and 8 only:
//
//   B^T = | 4  0 -5  0  1  0 |     A^T = | 1  1  1  1  1  0 |
//         | 0 -4 -4  1  1  0 |           | 0  1 -1  2 -2  0 |
//         | 0  4 -4 -1  1  0 |           | 0  1  1  4  4  0 |
//         | 0 -2 -1  2  1  0 |           | 0  1 -1  8 -8  1 |
//         | 0  2 -1 -2  1  0 |
//         | 0  4  0 -5  0  1 |
//
// The products are larger and cancel more than F(2x2)'s: ~1e-6 relative error of a conv output instead of ~1e-7.
template <typename V>
__device__ __forceinline__ void bt6(const V d0, const V d1, const V d2, const V d3, const V d4, const V d5, V* t) {
  const V a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
  t[0] = 4.f * d0 - 5.f * d2 + d4;
  t[1] = a + b;
  t[2] = a - b;
  t[3] = c + e;
  t[4] = c - e;
  t[5] = 4.f * d1 - 5.f * d3 + d5;
}

// one thread = one tile x 2 channels (36 loads of 8 bytes; consecutive threads walk the channels).  v layout [36][T][C].
// Addresses through buffer descriptors: lane part (the tile's first pixel, this channel pair) in the voffset, tap / component
// part a scalar - no 64-bit index arithmetic and no branch per element (the first form spent ~10 integer instructions and
// an exec-mask branch on each of its 36 loads); taps outside the image get an out-of-range voffset and read as zero.
// The input descriptor starts one row and one pixel BEFORE the tensor so that the tap offsets (i W + j) C are non-negative.
constexpr unsigned WOOB = 0x80000000u;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void winograd43_input_kernel(const float* __restrict__ x, float* __restrict__ v, int H, int W,
                                                               int C, int th, int tw, long long T, unsigned x_bytes, unsigned v_bytes) {
  const int c2n = C >> 1;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * c2n) return;
  const long long t = gid / c2n;
  const int c = (int)(gid - t * c2n) * 2;
  const int tx = (int)(t % tw);
  const long long r = t / tw;
  const int ty = (int)(r % th);
  const int n = (int)(r / th);
  const unsigned lead = (unsigned)(W + 1) * C * 4;
  const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) - (size_t)(W + 1) * C, 0, x_bytes + lead, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(v, 0, v_bytes, 0x00020000);
  const unsigned voff0 = (unsigned)((((n * H + 4 * ty) * W + 4 * tx) * C + c) * 4);
  bool row_ok[6], col_ok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    row_ok[i] = (unsigned)(4 * ty - 1 + i) < (unsigned)H;
    col_ok[i] = (unsigned)(4 * tx - 1 + i) < (unsigned)W;
  }
  f32x2 rt[6][6];  // B^T d, column by column
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    f32x2 d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
      d[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rsrc, (row_ok[i] && col_ok[j]) ? voff0 : WOOB,
                                                                            (i * W + j) * C * 4, 0));
    f32x2 tcol[6];
    bt6(d[0], d[1], d[2], d[3], d[4], d[5], tcol);
#pragma unroll
    for (int i = 0; i < 6; ++i) rt[i][j] = tcol[i];
  }
  const unsigned step = (unsigned)(T * C) * 4u;              // bytes between components (36 x step < 2^31: checked by the launcher)
  const unsigned voff_v = (unsigned)((t * C + c) * 4);
#pragma unroll
  for (int i = 0; i < 6; ++i) {  // (B^T d) B
    f32x2 o[6];
    bt6(rt[i][0], rt[i][1], rt[i][2], rt[i][3], rt[i][4], rt[i][5], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[j]), v_rsrc, voff_v, (6 * i + j) * step, 0);
  }
}

template <typename V>
__device__ __forceinline__ void at6(const V m0, const V m1, const V m2, const V m3, const V m4, const V m5, V* y) {
  const V s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = m0 + s12 + s34;
  y[1] = d12 + 2.f * d34;
  y[2] = s12 + 4.f * s34;
  y[3] = d12 + 8.f * d34 + m5;
}

// one thread = one tile x 2 output channels.  m layout [36][T][K]; y NHWC.  Buffer addressing as in the input kernel; the
// residual block is requested first: its 16 loads are in flight under the 36 loads and the arithmetic of the transform
// (requested where they are consumed, behind per-pixel bounds checks, they cost 0.024 ms per layer3 conv for 52 MB - half
// the kernel); pixels beyond the image read as zero and are not stored (out-of-range voffset).
__global__ __launch_bounds__(256) void winograd43_output_kernel(const float* __restrict__ m, const float* __restrict__ scale,
                                                                const float* __restrict__ bias, const float* __restrict__ residual,
                                                                int relu, float* __restrict__ y, int H, int W, int K, int th,
                                                                int tw, long long T, unsigned m_bytes, unsigned y_bytes) {
  const int k2n = K >> 1;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= T * k2n) return;
  const long long t = gid / k2n;
  const int k = (int)(gid - t * k2n) * 2;
  const int tx = (int)(t % tw);
  const long long r = t / tw;
  const int ty = (int)(r % th);
  const int n = (int)(r / th);
  const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m), 0, m_bytes, 0x00020000);
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, y_bytes, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual ? residual : y), 0, residual ? y_bytes : 0u, 0x00020000);
  const unsigned voff0 = (unsigned)((((n * H + 4 * ty) * W + 4 * tx) * K + k) * 4);
  unsigned pv[4][4];   // this lane's offset of pixel (p, q) of the tile: the tile origin, or out of range beyond the image
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) pv[p][q] = (4 * ty + p < H && 4 * tx + q < W) ? voff0 : WOOB;
  f32x2 res[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      res[p][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, pv[p][q], (p * W + q) * K * 4, 0));   // (no residual: 0 records, zeros)
  const unsigned step = (unsigned)(T * K) * 4u;
  const unsigned voff_m = (unsigned)((t * K + k) * 4);
  f32x2 u[4][6];  // A^T M, column by column
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    f32x2 a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) a[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(m_rsrc, voff_m, (6 * i + j) * step, 0));
    f32x2 col[4];
    at6(a[0], a[1], a[2], a[3], a[4], a[5], col);
#pragma unroll
    for (int p = 0; p < 4; ++p) u[p][j] = col[p];
  }
  f32x2 sc = {1.f, 1.f}, bi = {0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x2*>(scale + k);
  if (bias) bi = *reinterpret_cast<const f32x2*>(bias + k);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    f32x2 o4[4];
    at6(u[p][0], u[p][1], u[p][2], u[p][3], u[p][4], u[p][5], o4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x2 val = o4[q] * sc + bi + res[p][q];
      if (relu) {
        val[0] = fmaxf(val[0], 0.f);
        val[1] = fmaxf(val[1], 0.f);
      }
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), y_rsrc, pv[p][q], (p * W + q) * K * 4, 0);
    }
  }
}


// ---- output transform of conv k (+ folded BN, residual, ReLU) and input transform of conv k + 1 in ONE launch, for the layers whose
// image fits a workgroup's LDS (layer3 / layer4 at 640 x 640: 40 x 40 and 20 x 20): the activation between two 3x3 convs of a basic block
// (/root/reference/src/text_detection/model.rs:40-55) goes M -> y -> V without y's round trip through HBM - and, where nothing else
// reads it (conv1's output inside a block), without y ever existing.  One workgroup = one image x 16 channels: phase A, per tile and
// channel pair, A^T M A + epilogue into an LDS image with a zero ring (the next conv's padding); phase B, per tile and pair, the 6 x 6
// patch out of that image, B^T d B, 36 stores into V.  Same arithmetic, operation for operation, as the two kernels above: bit-identical.
constexpr int OI_CG = 16;                 // channels per workgroup (64 contiguous bytes per tile and component on the global side)
constexpr int OI_PS = 20;                 // floats per LDS pixel: 16 + 4 of padding (tile stride 320 B: neighbouring tiles on different banks)
__global__ __launch_bounds__(256) void winograd43_out_in_kernel(const float* __restrict__ m, const float* __restrict__ scale, const float* __restrict__ bias,
                                                                const float* __restrict__ residual, int relu, float* __restrict__ y, float* __restrict__ v,
                                                                int H, int W, int K, int th, int tw, long long T, unsigned m_bytes, unsigned y_bytes) {
  extern __shared__ __attribute__((aligned(16))) float img[];   // [(4 th + 2)][(4 tw + 2)][OI_PS]
  const int groups = K / OI_CG;
  const int n = blockIdx.x / groups, k0 = (blockIdx.x - n * groups) * OI_CG;
  const int tid = threadIdx.x;
  const int LW = 4 * tw + 2, LH = 4 * th + 2;
  // zero ring (and nothing else: every interior pixel is written in phase A - pixels beyond the image as zeros)
  for (int i = tid; i < 2 * (LW + LH) * (OI_CG / 4); i += 256) {
    const int c4 = (i % (OI_CG / 4)) * 4, q = i / (OI_CG / 4);
    int yy, xx;
    if (q < LW) { yy = 0; xx = q; }
    else if (q < 2 * LW) { yy = LH - 1; xx = q - LW; }
    else if (q < 2 * LW + LH) { yy = q - 2 * LW; xx = 0; }
    else { yy = q - 2 * LW - LH; xx = LW - 1; }
    *reinterpret_cast<f32x4*>(&img[(yy * LW + xx) * OI_PS + c4]) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const auto m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m), 0, m_bytes, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(v, 0, m_bytes, 0x00020000);
  const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(y ? y : v, 0, y ? y_bytes : 0u, 0x00020000);
  const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual ? residual : m), 0, residual ? y_bytes : 0u, 0x00020000);
  const unsigned step = (unsigned)(T * K) * 4u;
  const int items = th * tw * (OI_CG / 2);
  // ---- phase A
  for (int it = tid; it < items; it += 256) {
    const int kp = it & (OI_CG / 2 - 1), tl = it / (OI_CG / 2);
    const int ty = tl / tw, tx = tl - ty * tw;
    const int k = k0 + 2 * kp;
    const long long t = ((long long)n * th + ty) * tw + tx;
    const unsigned voff0 = (unsigned)((((n * H + 4 * ty) * W + 4 * tx) * K + k) * 4);
    unsigned pv[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[p][q] = (4 * ty + p < H && 4 * tx + q < W) ? voff0 : WOOB;
    f32x2 res[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        res[p][q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, pv[p][q], (p * W + q) * K * 4, 0));
    const unsigned voff_m = (unsigned)((t * K + k) * 4);
    f32x2 u[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x2 a[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) a[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(m_rsrc, voff_m, (6 * i + j) * step, 0));
      f32x2 col[4];
      at6(a[0], a[1], a[2], a[3], a[4], a[5], col);
#pragma unroll
      for (int p = 0; p < 4; ++p) u[p][j] = col[p];
    }
    f32x2 sc = {1.f, 1.f}, bi = {0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x2*>(scale + k);
    if (bias) bi = *reinterpret_cast<const f32x2*>(bias + k);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      f32x2 o4[4];
      at6(u[p][0], u[p][1], u[p][2], u[p][3], u[p][4], u[p][5], o4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x2 val = o4[q] * sc + bi + res[p][q];
        if (relu) {
          val[0] = fmaxf(val[0], 0.f);
          val[1] = fmaxf(val[1], 0.f);
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, val), y_rsrc, pv[p][q], (p * W + q) * K * 4, 0);   // (no y: an empty descriptor drops it)
        if (pv[p][q] == WOOB) val = f32x2{0.f, 0.f};   // beyond the image: the next conv's zero padding
        *reinterpret_cast<f32x2*>(&img[((4 * ty + p + 1) * LW + 4 * tx + q + 1) * OI_PS + 2 * kp]) = val;
      }
    }
  }
  __syncthreads();
  // ---- phase B
  for (int it = tid; it < items; it += 256) {
    const int kp = it & (OI_CG / 2 - 1), tl = it / (OI_CG / 2);
    const int ty = tl / tw, tx = tl - ty * tw;
    const int k = k0 + 2 * kp;
    const long long t = ((long long)n * th + ty) * tw + tx;
    const float* src = &img[((4 * ty) * LW + 4 * tx) * OI_PS + 2 * kp];   // patch origin = pixel (4 ty - 1, 4 tx - 1) of the image
    f32x2 rt[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x2 d[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = *reinterpret_cast<const f32x2*>(src + (i * LW + j) * OI_PS);
      f32x2 tcol[6];
      bt6(d[0], d[1], d[2], d[3], d[4], d[5], tcol);
#pragma unroll
      for (int i = 0; i < 6; ++i) rt[i][j] = tcol[i];
    }
    const unsigned voff_v = (unsigned)((t * K + k) * 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f32x2 o[6];
      bt6(rt[i][0], rt[i][1], rt[i][2], rt[i][3], rt[i][4], rt[i][5], o);
#pragma unroll
      for (int j = 0; j < 6; ++j) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[j]), v_rsrc, voff_v, (6 * i + j) * step, 0);
    }
  }
}

}  // namespace

size_t winograd43_out_in_lds_bytes(int H, int W) { return (size_t)(4 * ((H + 3) / 4) + 2) * (4 * ((W + 3) / 4) + 2) * OI_PS * 4; }
bool winograd43_out_in_fits(int H, int W, int K) { return K % OI_CG == 0 && winograd43_out_in_lds_bytes(H, W) <= 160 * 1024; }

void launch_winograd43_out_in(const float* m, const float* scale, const float* bias, const float* residual, int relu, float* y, float* v, int N, int H, int W,
                              int K, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || !winograd43_out_in_fits(H, W, K)) fail(OCR_ERR_INVALID, "winograd out+in: bad shape N=%d H=%d W=%d K=%d", N, H, W, K);
  const int th = (H + 3) / 4, tw = (W + 3) / 4;
  const long long T = (long long)N * th * tw;
  const unsigned long long yb = (unsigned long long)N * H * W * K * 4, mb = 36ull * T * K * 4;
  if (yb >= (1ull << 31) || mb >= (1ull << 31)) fail(OCR_ERR_INVALID, "winograd out+in: tensors beyond 2 GB");
  const size_t lds = winograd43_out_in_lds_bytes(H, W);
  // (per launch: the attribute belongs to the device the calling thread has current, and handles of several devices may share a process)
  OCR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(winograd43_out_in_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(winograd43_out_in_kernel, dim3((unsigned)(N * (K / OI_CG))), dim3(256), lds, s, m, scale, bias, residual, relu, y, v, H, W, K, th, tw, T,
                     (unsigned)mb, (unsigned)yb);
  OCR_HIP(hipGetLastError());
}

namespace {
}  // namespace

void launch_winograd_input(const float* x, float* v, int N, int H, int W, int C, int m, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || C % 4 || (m != 2 && m != 4)) fail(OCR_ERR_INVALID, "winograd input: bad shape N=%d H=%d W=%d C=%d m=%d", N, H, W, C, m);
  if (m == 4) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const long long threads = T * (C / 2);
    if (threads >= (1ll << 31) * 256) fail(OCR_ERR_INVALID, "winograd input: too large");
    const unsigned long long xb = (unsigned long long)N * H * W * C * 4, vb = 36ull * T * C * 4;
    if (xb + (unsigned long long)(W + 1) * C * 4 >= (1ull << 31) || vb >= (1ull << 31)) fail(OCR_ERR_INVALID, "winograd input: tensors beyond 2 GB");
    hipLaunchKernelGGL(winograd43_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, v, H, W, C, th, tw, T, (unsigned)xb, (unsigned)vb);
    OCR_HIP(hipGetLastError());
    return;
  }
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const long long T = (long long)N * th * tw;
  const long long threads = T * (C / 4);
  if (threads >= (1ll << 31) * 256) fail(OCR_ERR_INVALID, "winograd input: too large");
  hipLaunchKernelGGL(winograd_input_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, v, H, W, C, th, tw, T);
  OCR_HIP(hipGetLastError());
}

void launch_winograd_output(const float* m, const float* scale, const float* bias, const float* residual, int relu, float* y,
                            int N, int H, int W, int K, int m_tile, hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || K % 4 || (m_tile != 2 && m_tile != 4)) fail(OCR_ERR_INVALID, "winograd output: bad shape N=%d H=%d W=%d K=%d m=%d", N, H, W, K, m_tile);
  if (m_tile == 4) {
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)N * th * tw;
    const long long threads = T * (K / 2);
    if (threads >= (1ll << 31) * 256) fail(OCR_ERR_INVALID, "winograd output: too large");
    const unsigned long long yb = (unsigned long long)N * H * W * K * 4, mb = 36ull * T * K * 4;
    if (yb >= (1ull << 31) || mb >= (1ull << 31)) fail(OCR_ERR_INVALID, "winograd output: tensors beyond 2 GB");
    hipLaunchKernelGGL(winograd43_output_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, scale, bias, residual,
                       relu, y, H, W, K, th, tw, T, (unsigned)mb, (unsigned)yb);
    OCR_HIP(hipGetLastError());
    return;
  }
  const int th = (H + 1) / 2, tw = (W + 1) / 2;
  const long long T = (long long)N * th * tw;
  const long long threads = T * (K / 4);
  if (threads >= (1ll << 31) * 256) fail(OCR_ERR_INVALID, "winograd output: too large");
  hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, m, scale, bias, residual,
                     relu, y, H, W, K, th, tw, T);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
