// Probe (round 4): can the three-way bf16 split of an f32 operand (x = hi + mid + lo, round to nearest even at every level)
// take its remainders from v_dot2c_f32_bf16 instead of expanding the bf16 term back to f32 and subtracting?
//   r = x - float(h)        : shift/and to rebuild float(h) (1 VALU per element) + v_sub_f32 (1)
//   r = dot2c(x; P, (-1, 0)): ONE VALU per element on the packed pair P = cvt_pk(x0, x1) itself
// The remainder is exactly representable, so any product-sum that keeps f32 alignment returns it exactly; what has to be
// MEASURED is whether gfx950's dot2 datapath does (flushes, truncated alignment) and whether it co-issues with bf16 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 split_dot2.hip -o split_dot2 && ./split_dot2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split_ref(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  bf16x2 h, m, l;
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  const float s0 = r0 - (float)m[0], s1 = r1 - (float)m[1];
  l[0] = (__bf16)s0; l[1] = (__bf16)s1;
  H = __builtin_bit_cast(unsigned, h); M = __builtin_bit_cast(unsigned, m); L = __builtin_bit_cast(unsigned, l);
}
__device__ inline void split_dot2(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  bf16x2 sel0, sel1;
  sel0[0] = (__bf16)-1.0f; sel0[1] = (__bf16)0.0f;
  sel1[0] = (__bf16)0.0f; sel1[1] = (__bf16)-1.0f;
  bf16x2 h; h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, sel0, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(h, sel1, x1, false);
  bf16x2 m; m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(m, sel0, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(m, sel1, r1, false);
  bf16x2 l; l[0] = (__bf16)s0; l[1] = (__bf16)s1;
  H = __builtin_bit_cast(unsigned, h); M = __builtin_bit_cast(unsigned, m); L = __builtin_bit_cast(unsigned, l);
}
__global__ void check(const float* x, int n, unsigned* out_ref, unsigned* out_d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  split_ref(x[2 * i], x[2 * i + 1], out_ref[3 * i], out_ref[3 * i + 1], out_ref[3 * i + 2]);
  split_dot2(x[2 * i], x[2 * i + 1], out_d2[3 * i], out_d2[3 * i + 1], out_d2[3 * i + 2]);
}

// timing: waves 0-3 issue 32x32x16 bf16 MFMAs back to back (mode bit 0), waves 4-7 split 8 pairs per iteration with the
// reference arithmetic (bit 1) or with dot2 (bit 2); bit 3: ONE wave does both, 3 MFMAs then one pair's split (the igemm's shape)
template <int mode>
__global__ __launch_bounds__(512) void timing(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_amdgcn_s_memtime();
  float v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = threadIdx.x * 1.37f + u * 0.11f;
  unsigned acc = 0;
  auto pair = [&](int u) {
    unsigned H, M, L;
    if (mode & 4) split_dot2(v[2 * u], v[2 * u + 1], H, M, L); else split_ref(v[2 * u], v[2 * u + 1], H, M, L);
    acc ^= H ^ M ^ L;
    v[2 * u] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v[2 * u]) ^ (L & 0x10000u)));   // keeps the chain data dependent
  };
  if (mode & 8) {
    if (wave >= 4) return;
    f32x16 b0 = {}, b1 = {}, b2 = {};
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + e * 0.01f); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, b2, 0, 0, 0);
        if (mode & 6) pair(u);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = b0[0] + b1[5] + b2[7] + (float)acc;
  } else if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 b0 = {}, b1 = {};
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + e * 0.01f); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, b1, 0, 0, 0);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = b0[0] + b1[5];
  } else {
    if (!(mode & 6)) return;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pair(u);
    }
    float s = (float)acc;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
static void launch(int mode, float* out, long long* cyc, int iters) {
  switch (mode) {
#define C(m) case m: hipLaunchKernelGGL(timing<m>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
    C(1) C(2) C(4) C(3) C(5) C(8) C(10) C(12)
  }
}

int main() {
  // ---- exactness
  std::vector<float> x;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (int i = 0; i < (1 << 22); ++i) {
    const float scales[] = {1.f, 1e-3f, 37.f, 1e4f, 1e-6f, 1e12f, 1e-20f, 1e30f, 1e-36f, 3e38f, 1e-38f, 1e-41f};
    x.push_back(u(rng) * scales[i % 12]);
  }
  for (uint32_t b = 0; b < (1u << 16); ++b) {   // every bf16 pattern widened, and its neighbours (ties of the first rounding)
    for (uint32_t lo : {0u, 1u, 0x7fffu, 0x8000u, 0x8001u, 0xffffu}) {
      const uint32_t w = (b << 16) | lo;
      float f; memcpy(&f, &w, 4);
      if (f == f && f - f == 0.f) x.push_back(f);
    }
  }
  if (x.size() & 1) x.push_back(0.f);
  const int n = (int)x.size();
  float* dx; unsigned *dr, *dd;
  hipMalloc(&dx, n * 4); hipMalloc(&dr, n / 2 * 12); hipMalloc(&dd, n / 2 * 12);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, dx, n, dr, dd);
  std::vector<unsigned> hr(n / 2 * 3), hd(n / 2 * 3);
  hipMemcpy(hr.data(), dr, hr.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hd.data(), dd, hd.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0;
  for (int i = 0; i < n / 2; ++i) {
    bool diff = false;
    for (int j = 0; j < 3; ++j) diff |= hr[3 * i + j] != hd[3 * i + j];
    if (!diff) continue;
    ++bad;
    const float a = x[2 * i], b = x[2 * i + 1];
    const bool tiny = (a != 0 && fabsf(a) < 1e-30f) || (b != 0 && fabsf(b) < 1e-30f);
    const bool huge = fabsf(a) > 3.3e38f || fabsf(b) > 3.3e38f;
    if (!tiny && !huge) {
      if (bad_normal < 10) printf("  differs: x = (%a, %a)  ref %08x %08x %08x  dot2 %08x %08x %08x\n", a, b, hr[3 * i], hr[3 * i + 1], hr[3 * i + 2], hd[3 * i], hd[3 * i + 1], hd[3 * i + 2]);
      ++bad_normal;
    }
  }
  printf("exactness: %d pairs, %ld differ (%ld of them with both |x| in [1e-30, 3.3e38])\n", n / 2, bad, bad_normal);
  // ---- timing
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  const int modes[] = {1, 2, 4, 3, 5, 8, 10, 12};
  const char* names[] = {"MFMA alone", "ref split alone", "dot2 split alone", "MFMA | ref split partner", "MFMA | dot2 split partner", "one wave: 3 MFMA", "one wave: 3 MFMA + ref pair",
                         "one wave: 3 MFMA + dot2 pair"};
  for (int mi = 0; mi < 8; ++mi) {
    hipMemset(cyc, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) launch(modes[mi], out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256 * 8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0, v = 0; for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; v += h[b * 8 + 4]; }
    m /= 256; v /= 256;
    printf("%-30s wave0 %.0f clk (%.1f per 24 MFMA group / per 8 x (3 MFMA + pair))   wave4 %.0f clk (%.1f per pair)\n", names[mi], m, m / iters, v, v / (iters * 8.0));
  }
  return 0;
}
