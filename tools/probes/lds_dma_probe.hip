// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write zeros to LDS
// (or leave the old bytes)?  Also checks the lane -> LDS placement (M0 base + 16*lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

__global__ void probe(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = -7.0f;  // sentinel
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // even lanes: valid (reverse order), odd lanes: out of range
  int voff = (lane & 1) ? 0x40000000 : (63 - lane) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds, 16, voff, 0, 0, 0);
  // second instruction into the second KiB with an SGPR offset of 16 bytes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + 256), 16, lane * 16, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}

int main() {
  const int n = 64 * 4 + 8;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  (void)hipMalloc(&d, n * 4);
  (void)hipMalloc(&o, 512 * 4);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n * 4, o);
  std::vector<float> r(512);
  (void)hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  for (int l = 0; l < 6; ++l) printf("lane %d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
  printf("second: lane0 %g %g, lane63 %g %g\n", r[256], r[257], r[256 + 63 * 4], r[256 + 63 * 4 + 3]);
  int zeros = 0, sentinels = 0;
  for (int l = 1; l < 64; l += 2) { zeros += r[l * 4] == 0.0f; sentinels += r[l * 4] == -7.0f; }
  printf("oob lanes: %d zero, %d sentinel (of 32)\n", zeros, sentinels);
  return 0;
}
