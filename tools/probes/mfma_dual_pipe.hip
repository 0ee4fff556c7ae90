// Do f32 MFMA (v_mfma_f32_16x16x4_f32, the fused Winograd kernel's instruction) and bf16 MFMA (v_mfma_f32_32x32x16_bf16, the
// split-bf16 conv family's) from TWO WAVES OF ONE SIMD run at the same time on gfx950, or do they share one matrix pipe?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue the f32 instruction, waves 4-7 (same SIMDs) the bf16 one.
// mode bit 0: f32 waves active, bit 1: bf16 waves active.  mode 4/5: ONE wave per SIMD alternating the two instructions
// (4 = 1:1, 5 = two f32 per bf16).  Reports ms and cycles per instruction (s_memtime-free: from the wall clock at the reported
// shader clock), alone and together.
//   hipcc --offload-arch=gfx950 -O3 mfma_dual_pipe.hip -o mfma_dual_pipe && ./mfma_dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define F32_MFMA(acc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0)
#define BF_MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  const float x = threadIdx.x * 1e-3f, y = 1.0001f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(1.0f + 0.01f * i); }
  if (mode >= 4) {  // one wave per SIMD issues both kinds, interleaved
    if (wave >= 4) return;
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x16 d0 = {}, d1 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        F32_MFMA(c0); BF_MFMA(d0);
        if (mode == 5) F32_MFMA(c2);
        F32_MFMA(c1); BF_MFMA(d1);
        if (mode == 5) F32_MFMA(c3);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5];
    return;
  }
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { F32_MFMA(c0); F32_MFMA(c1); F32_MFMA(c2); F32_MFMA(c3); }   // 32 per iteration
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    if (!(mode & 2)) return;
    f32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { BF_MFMA(d0); BF_MFMA(d1); BF_MFMA(d2); BF_MFMA(d3); }       // 16 per iteration
    }
    out[blockIdx.x * 512 + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
  }
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double mhz = p.clockRate / 1000.0;
  const int iters = 4000;
  struct { int mode; const char* name; double nf32, nbf; } runs[] = {
    {1, "f32 16x16x4 alone (32/iter)", 32, 0},
    {2, "bf16 32x32x16 alone (16/iter)", 0, 16},
    {3, "both, separate waves of a SIMD", 32, 16},
    {4, "one wave, 1 f32 : 1 bf16 (16+16/iter)", 16, 16},
    {5, "one wave, 2 f32 : 1 bf16 (32+16/iter)", 32, 16},
  };
  printf("shader clock reported %.0f MHz\n", mhz);
  for (auto& r : runs) {
    k<<<256, 512>>>(out, 10, r.mode);
    hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, r.mode);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * mhz * 1e6 / iters;   // cycles per loop iteration
    printf("%-42s %8.3f ms  %7.1f cycles/iter", r.name, ms, cyc);
    if (r.nf32 && !r.nbf) printf("  = %.1f cycles per f32 MFMA", cyc / r.nf32);
    if (r.nbf && !r.nf32) printf("  = %.1f cycles per bf16 MFMA", cyc / r.nbf);
    if (r.nf32 && r.nbf) printf("  (serial sum would be %.0f f32 x 8 + %.0f bf16 x 16(8 pass) cycles)", r.nf32, r.nbf);
    printf("\n");
  }
  return 0;
}
