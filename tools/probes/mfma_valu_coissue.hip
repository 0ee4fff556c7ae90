// Do f32 MFMA (v_mfma_f32_32x32x2_f32) and f32 VALU from ANOTHER wave of the same SIMD overlap on gfx950?
// 512-thread workgroups, one per CU: waves 0-3 issue MFMAs, waves 4-7 (same SIMDs) issue v_fma_f32 / ds_read.
// mode bit 0: MFMA waves active, bit 1: VALU waves active, bit 2: LDS-read waves instead of VALU.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_coissue.hip -o mfma_valu_coissue && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  __shared__ f32x4 buf[1024];
  const int wave = threadIdx.x >> 6;
  buf[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  buf[threadIdx.x + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 a0 = {}, a1 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[3];
  } else {
    if (!(mode & 2)) return;
    if (mode & 4) {
      f32x4 s = {};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) s += buf[(threadIdx.x + 64 * u + i) & 1023];
      }
      out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = threadIdx.x + u;
      const float m = 1.0001f, c = 0.5f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = fmaf(v[u], m, c);   // 512 independent-ish v_fma per iteration
      }
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  }
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  const char* names[] = {"", "MFMA only (32/iter)", "VALU only (512 v_fma/iter)", "MFMA + VALU", "", "", "LDS only (32 ds_read_b128/iter)", "MFMA + LDS"};
  for (int mode : {1, 2, 3, 6, 7}) {
    k<<<256, 512>>>(out, 10, mode);
    hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, mode);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-32s %8.3f ms\n", names[mode], ms);
  }
  return 0;
}
