// Is v_mfma_f32_16x16x4_f32 fed with k = (0,4,1,5),(2,6,3,7),... bit-identical to v_mfma_f32_32x32x2_f32 fed with
// (0,4),(1,5),(2,6),(3,7),... (the k order of conv_igemm / rec_conv)?  One wave, K = 64, random data with a wide
// exponent spread so that the rounding order shows.  Prints the number of differing outputs of the common 16 x 16 corner.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;
__global__ void k(const float* A, const float* B, float* d32, float* d16) {  // A [32][K], B [32][K]
  const int lane = threadIdx.x;
  {
    f32x16 acc = {};
    const int r = lane & 31, h = lane >> 5;
    for (int g = 0; g < K / 8; ++g)
      for (int e = 0; e < 4; ++e) {
        const int kk = 8 * g + 4 * h + e;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + kk], B[r * K + kk], acc, 0, 0, 0);
      }
    for (int e = 0; e < 16; ++e) d32[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
  }
  {
    f32x4 acc = {};
    const int r = lane & 15, q = lane >> 4;
    for (int g = 0; g < K / 8; ++g)
      for (int half = 0; half < 2; ++half) {
        // instruction `half` of group g: k slots q = 0..3 -> 8 g + {0,4,1,5} (+2 for the second instruction)
        const int kk = 8 * g + 2 * half + (q >> 1) + 4 * (q & 1);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + kk], B[r * K + kk], acc, 0, 0, 0);
      }
    for (int e = 0; e < 4; ++e) d16[(q * 4 + e) * 16 + r] = acc[e];
  }
}
int main() {
  std::vector<float> A(32 * K), B(32 * K);
  srand(1);
  for (auto& v : A) v = ldexpf((rand() / (float)RAND_MAX) - 0.5f, rand() % 12 - 6);
  for (auto& v : B) v = ldexpf((rand() / (float)RAND_MAX) - 0.5f, rand() % 12 - 6);
  float *dA, *dB, *d32, *d16;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d32, 32 * 32 * 4); (void)hipMalloc(&d16, 16 * 16 * 4);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, d32, d16);
  std::vector<float> h32(32 * 32), h16(16 * 16);
  (void)hipMemcpy(h32.data(), d32, h32.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h16.data(), d16, h16.size() * 4, hipMemcpyDeviceToHost);
  int diff = 0, diff_chain = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      if (memcmp(&h32[i * 32 + j], &h16[i * 16 + j], 4)) ++diff;
      float c = 0.f;  // the k-ordered fmaf chain in conv_igemm's order
      for (int g = 0; g < K / 8; ++g)
        for (int e = 0; e < 4; ++e) {
          c = fmaf(A[i * K + 8 * g + e], B[j * K + 8 * g + e], c);
          c = fmaf(A[i * K + 8 * g + 4 + e], B[j * K + 8 * g + 4 + e], c);
        }
      if (memcmp(&c, &h32[i * 32 + j], 4)) ++diff_chain;
    }
  printf("16x16x4 vs 32x32x2: %d of 256 outputs differ; 32x32x2 vs host fmaf chain: %d differ\n", diff, diff_chain);
  return 0;
}
