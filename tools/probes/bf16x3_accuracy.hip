// How accurate is an f32 product sum computed on the bf16 matrix cores from operands split into three bf16 terms
// (a = hi + mid + lo exactly: 3 x 8 significant bits), six products per pair (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi;
// the dropped mid.lo, lo.mid, lo.lo are <= 2^-23 of the product), accumulated by v_mfma_f32_32x32x16_bf16 in f32?
// Compared with the exact-f32 MFMA chain (v_mfma_f32_32x32x2_f32) and a three-product variant (hi.hi, hi.mid, mid.hi)
// against an f64 reference, for K = 576 / 2304 / 4608 (the reduction lengths of the detector's 3x3 convs), on normal data
// and on data with a wide exponent spread.  One wave per 32 x 32 tile, 256 tiles.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bf16x3_accuracy.hip -o /tmp/bf16x3 && /tmp/bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_trunc_f32(float a) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u); }
__device__ __forceinline__ __bf16 top16(float a) { return __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, a) >> 16)); }

// mode 0: round to nearest even at each level; mode 1: truncation at each level (remainders stay exact either way)
template <int MODE>
__device__ __forceinline__ void split3(float a, __bf16& hi, __bf16& mid, __bf16& lo) {
  if (MODE == 0) {
    hi = (__bf16)a;
    const float r1 = a - (float)hi;
    mid = (__bf16)r1;
    const float r2 = r1 - (float)mid;
    lo = (__bf16)r2;
  } else {
    const float h = bf16_trunc_f32(a);
    hi = top16(h);
    const float r1 = a - h;
    const float m = bf16_trunc_f32(r1);
    mid = top16(m);
    lo = top16(r1 - m);
  }
}

template <int MODE>
__global__ void k(const float* A, const float* B, int K, float* d_f32, float* d_x6, float* d_x3, float* d_x6s) {  // A [T][32][K], B [T][32][K]
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  A += (size_t)blockIdx.x * 32 * K;
  B += (size_t)blockIdx.x * 32 * K;
  f32x16 acc = {}, x6 = {}, x3 = {}, s_hi = {}, s_mid = {}, s_lo = {};
  for (int g = 0; g < K / 8; ++g)
    for (int e = 0; e < 4; ++e) {
      const int kk = 8 * g + 4 * h + e;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + kk], B[r * K + kk], acc, 0, 0, 0);
    }
  for (int g = 0; g < K / 16; ++g) {
    bf16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      const int kk = 16 * g + 8 * h + j;
      __bf16 x, y, z;
      split3<MODE>(A[r * K + kk], x, y, z);
      ah[j] = x; am[j] = y; al[j] = z;
      split3<MODE>(B[r * K + kk], x, y, z);
      bh[j] = x; bm[j] = y; bl[j] = z;
    }
    // one accumulator, small terms first inside a K group
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, x6, 0, 0, 0);
    x6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, x6, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, x3, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, x3, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, x3, 0, 0, 0);
    // three accumulators by magnitude class, summed at the end
    s_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, s_lo, 0, 0, 0);
    s_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, s_lo, 0, 0, 0);
    s_lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, s_lo, 0, 0, 0);
    s_mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, s_mid, 0, 0, 0);
    s_mid = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, s_mid, 0, 0, 0);
    s_hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, s_hi, 0, 0, 0);
  }
  const size_t o = (size_t)blockIdx.x * 1024;
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
    d_f32[o + row * 32 + r] = acc[e];
    d_x6[o + row * 32 + r] = x6[e];
    d_x3[o + row * 32 + r] = x3[e];
    d_x6s[o + row * 32 + r] = s_hi[e] + (s_mid[e] + s_lo[e]);
  }
}

int main() {
  const int T = 256;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (int wide = 0; wide < 3; ++wide)
    for (int K : {576, 2304, 4608}) {
      std::vector<float> A((size_t)T * 32 * K), B(A.size());
      for (auto& v : A) {
        v = nd(rng);
        if (wide == 1) v = ldexpf(v, (int)(rng() % 24) - 12);
        if (wide == 2) v = fabsf(v);  // post-ReLU-like: no cancellation in the sum
      }
      for (auto& v : B) {
        v = nd(rng) * 0.05f;
        if (wide == 1) v = ldexpf(v, (int)(rng() % 24) - 12);
        if (wide == 2) v = fabsf(v);
      }
      float *dA, *dB, *d[4];
      (void)hipMalloc(&dA, A.size() * 4);
      (void)hipMalloc(&dB, B.size() * 4);
      for (auto& p : d) (void)hipMalloc(&p, (size_t)T * 1024 * 4);
      (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
      (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      std::vector<double> ref((size_t)T * 1024);
      double rms = 0;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double s = 0;
            const float* a = &A[((size_t)t * 32 + i) * K];
            const float* b = &B[((size_t)t * 32 + j) * K];
            for (int kk = 0; kk < K; ++kk) s += (double)a[kk] * (double)b[kk];
            ref[(size_t)t * 1024 + i * 32 + j] = s;
            rms += s * s;
          }
      rms = sqrt(rms / ref.size());
      for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) k<0><<<T, 64>>>(dA, dB, K, d[0], d[1], d[2], d[3]);
        else k<1><<<T, 64>>>(dA, dB, K, d[0], d[1], d[2], d[3]);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        const char* names[4] = {"f32 mfma chain", "bf16 x6 one acc", "bf16 x3", "bf16 x6 three acc"};
        printf("data %s K %4d split %s  (rms of C %.3g)\n", wide == 0 ? "normal" : wide == 1 ? "wide-exponent" : "non-negative", K, mode ? "trunc" : "rne", rms);
        for (int v = 0; v < 4; ++v) {
          std::vector<float> h((size_t)T * 1024);
          (void)hipMemcpy(h.data(), d[v], h.size() * 4, hipMemcpyDeviceToHost);
          double mx = 0, se = 0, bias = 0;
          for (size_t i = 0; i < h.size(); ++i) {
            const double e = (double)h[i] - ref[i];
            mx = fmax(mx, fabs(e));
            se += e * e;
            bias += e;
          }
          printf("   %-18s max|err|/rms %.3e   rms err/rms %.3e   mean err/rms %+.3e\n", names[v], mx / rms, sqrt(se / h.size()) / rms, bias / h.size() / rms);
        }
      }
      (void)hipFree(dA); (void)hipFree(dB);
      for (auto& p : d) (void)hipFree(p);
    }
  return 0;
}
