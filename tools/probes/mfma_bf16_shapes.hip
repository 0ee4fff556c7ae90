// Probe (round 4): (1) cycles per v_mfma_f32_16x16x32_bf16 vs the K = 16 form v_mfma_f32_16x16x16_bf16 on gfx950, one wave per SIMD,
// four independent accumulators; (2) how much packed-f32 VALU (v_pk_fma_f32 + v_cvt_pk_bf16_f32, the Winograd input transform's
// mix) a SECOND wave of the same SIMD gets through beside a wave issuing 16x16x32 MFMAs back to back.
//   hipcc --offload-arch=gfx950 -O3 mfma_bf16_shapes.hip -o mfma_bf16_shapes && ./mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
// mode bit 0: MFMA waves (0-3) active; bit 1: VALU waves (4-7) active; bit 2: K = 16 form
template <int mode>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f + e); y[e] = (__bf16)(1.0f + e * 0.01f); }
    if (mode & 4) {
      s16x4 xs = __builtin_bit_cast(s16x4, __builtin_shufflevector(x, x, 0, 1, 2, 3)), ys = __builtin_bit_cast(s16x4, __builtin_shufflevector(y, y, 0, 1, 2, 3));
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xs, ys, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ys, xs, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xs, xs, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ys, ys, a3, 0, 0, 0);
        }
      }
    } else if (mode & 8) {   // 32x32x16: half as many instructions for the same FLOPs
      typedef float f32x16 __attribute__((ext_vector_type(16)));
      f32x16 b0 = {}, b1 = {};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b0, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 11"); else if (mode & 32) asm volatile("s_nop 15\n\ts_nop 3");
          b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, b1, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 11"); else if (mode & 32) asm volatile("s_nop 15\n\ts_nop 3");
        }
      }
      a0[0] = b0[0] + b1[5];
    } else {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 2"); else if (mode & 32) asm volatile("s_nop 5");
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 2"); else if (mode & 32) asm volatile("s_nop 5");
          a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 2"); else if (mode & 32) asm volatile("s_nop 5");
          a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
          if (mode & 64) asm volatile("s_nop 2"); else if (mode & 32) asm volatile("s_nop 5");
        }
      }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  } else {
    if (!(mode & 2)) return;
    if (mode & 16) __builtin_amdgcn_s_setprio(3);   // the VALU wave outranks its (older) MFMA partner
    f32x2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = f32x2{threadIdx.x + u * 1.f, u * 0.5f};
    const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned acc = 0;
    if (mode & 128) {   // the same arithmetic without packed f32 instructions: v_fma_f32 / v_sub_f32 on the halves (asm: hipcc would re-pack them)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(v[u].x), "+v"(v[u].y) : "v"(m.x), "v"(c.x), "v"(m.y), "v"(c.y));
          }
#pragma unroll
          for (int u = 0; u < 8; u += 2) {
            const bf16x2 h = __builtin_convertvector(v[u], bf16x2);
            acc ^= __builtin_bit_cast(unsigned, h);
            const f32x2 hf = __builtin_convertvector(h, f32x2);
            asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %3" : "+v"(v[u + 1].x), "+v"(v[u + 1].y) : "v"(hf.x), "v"(hf.y));
          }
        }
      }
    } else {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = v[u] * m + c;           // 8 v_pk_fma_f32
#pragma unroll
          for (int u = 0; u < 8; u += 2) {                            // 4 cvt_pk + 4 sub-like ops
            const bf16x2 h = __builtin_convertvector(v[u], bf16x2);
            acc ^= __builtin_bit_cast(unsigned, h);
            v[u + 1] = v[u + 1] - __builtin_convertvector(h, f32x2);
          }
        }
      }
    }
    float s = (float)acc;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x + v[u].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
static void launch(int mode, float* out, long long* cyc, int iters) {
  switch (mode) {
#define C(m) case m: hipLaunchKernelGGL(k<m>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
    C(1) C(2) C(3) C(9) C(11) C(130) C(131) C(139) C(97) C(227) C(105) C(235)
  }
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  const int modes[] = {1, 2, 3, 9, 11, 130, 131, 139, 97, 227, 105, 235};
  const char* names[] = {"16x16x32 alone", "VALU (packed) alone", "16x16x32 + packed VALU partner", "32x32x16 alone", "32x32x16 + packed VALU partner",
                         "VALU (unpacked) alone", "16x16x32 + unpacked VALU partner", "32x32x16 + unpacked VALU partner", "16x16x32 lightly paced alone",
                         "16x16x32 lightly paced + unpacked VALU", "32x32x16 lightly paced alone", "32x32x16 lightly paced + unpacked VALU"};
  for (int mi = 0; mi < 12; ++mi) {
    hipMemset(cyc, 0, 256 * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) launch(modes[mi], out, cyc, iters);
    hipDeviceSynchronize();
    long long h[256 * 8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0, v = 0; for (int b = 0; b < 256; ++b) { m += h[b * 8 + 0]; v += h[b * 8 + 4]; }
    m /= 256; v /= 256;
    printf("%-28s MFMA wave %.0f cycles (%.2f per MFMA of %d)   VALU wave %.0f cycles (%.2f per instruction of ~%d)\n", names[mi], m, m / (iters * 32.0), iters * 32,
           v, v / (iters * 4.0 * 20), iters * 80);
  }
  return 0;
}
