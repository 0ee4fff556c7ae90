// Probe (round 4): what an LDS-DMA patch stream costs as a function of the SEGMENT each pixel contributes.  The fused Winograd
// kernel stages an 18 x 18 pixel patch of a [N][H][W][64] f32 tensor per 16 x 16 output block, 16 channels (64 B per pixel = half
// a cache line) at a time; the alternative is 32 channels (128 B = a whole line) half as often.  Same bytes, same number of
// buffer_load_dwordx4 ... lds instructions per byte, half as many distinct lines per instruction.
//   mode 0: 4 loads per block of 16 channels (18 rows x [16 px + 2 px] instructions, as winograd43_fused.hip)
//   mode 1: 2 loads per block of 32 channels (41 linear instructions of 8 px)
// Two workgroups per CU; each requests 41 KB (two 16-channel loads, or one 32-channel load), waits, touches the buffer, repeats:
// equal bytes in flight, nothing else running - the cost of the request pattern itself.
//   hipcc --offload-arch=gfx950 -O3 lds_dma_gather.hip -o lds_dma_gather && ./lds_dma_gather
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned OOB = 0x80000000u;
template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
constexpr int H = 160, W = 160, C = 64, N = 32;
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* x, unsigned bytes, int nblocks, float* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[42 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  auto issue = [&](int blk, int part) {
    const int bx = blk % 10, by = (blk / 10) % 10, n = blk / 100;
    const int y0 = 16 * by, x0 = 16 * bx;
    const unsigned buf = lds0 + (unsigned)(MODE == 0 ? (part & 1) * 21 * 1024 : 0);
    if (MODE == 0) {
      const int piece = wave & 1;
      const int xx = x0 - 1 + 16 * piece + (lane >> 2);
      const unsigned pv = (unsigned)xx < (unsigned)W ? (unsigned)((xx * C + (lane & 3) * 4) * 4) : OOB;
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int row = (wave >> 1) + 2 * m, yy = y0 - 1 + row;
        const bool ok = (unsigned)yy < (unsigned)H;
        const unsigned soff = ok ? (unsigned)(((n * H + yy) * W * C + part * 16) * 4) : 0u;
        if (piece == 0 || lane < 8) dma16(rs, __builtin_amdgcn_readfirstlane(buf + (unsigned)(row * 1168 + piece * 1024)), ok ? pv : OOB, __builtin_amdgcn_readfirstlane(soff));
      }
    } else {
#pragma unroll
      for (int m = 0; m < 11; ++m) {
        const int kk = min(wave + 4 * m, 40);   // 41 instructions; the spare slots of waves 1-3 repeat the last one
        const int q = 8 * kk + (lane >> 3), row = q / 18, col = q - row * 18;
        const int yy = y0 - 1 + row, xx = x0 - 1 + col;
        const bool ok = q < 324 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        const int chunk = (lane & 7) ^ (((row >> 2) & 1) << 2);
        const unsigned v = ok ? (unsigned)((((n * H + yy) * W + xx) * C + part * 32 + chunk * 4) * 4) : OOB;
        dma16(rs, __builtin_amdgcn_readfirstlane(buf + (unsigned)(kk * 1024)), v, 0u);
      }
    }
  };
  float acc = 0.f;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    for (int h = 0; h < 2; ++h) {
      if (MODE == 0) {
        issue(blk, 2 * h);
        issue(blk, 2 * h + 1);
      } else {
        issue(blk, h);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      acc += *reinterpret_cast<const float*>(lds + tid * 16);   // touch the finished buffer
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + tid] = acc;
}
int main() {
  const size_t bytes = (size_t)N * H * W * C * 4;
  float *x, *out;
  hipMalloc(&x, bytes); hipMalloc(&out, 512 * 256 * 4);
  hipMemset(x, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nblocks = N * 100;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, x, (unsigned)bytes, nblocks, out);
        else hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, x, (unsigned)bytes, nblocks, out);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      const double staged = (double)nblocks * 324 * 256;   // bytes staged into LDS per launch
      printf("mode %d (%s per pixel): %.4f ms per pass over the tensor, %.2f TB/s staged (%.2f x the tensor's %.0f MB)\n", mode, mode ? "128 B" : " 64 B", ms,
             staged / ms * 1e-9, staged / bytes, bytes * 1e-6);
    }
  }
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
