// CTC greedy (best-path) decode, one wave per crop - an EXTENSION: BASELINE.json's north_star and configs[2] name "32x128 recognition
// crops + CTC greedy decode", the reference has no sequence recogniser (its Net classifies single 28 x 28 glyphs,
// /root/reference/src/char_recognition/model.rs:27-39; alphabet of 62 characters, src/utils.rs:7-9 - a CTC head over it has 63 classes).
// Nothing of the reference is replaced; the exact integer oracle is oracle/ctc_oracle.py.
//   logits [N][T][C] f32 (any monotone transform of them: log-probabilities, probabilities)
//   per crop: a[t] = the FIRST class attaining the maximum of column t; keep a[t] where a[t] != blank and (t == 0 or a[t] != a[t - 1]);
//   labels [N][T] = the kept classes in order, padded with -1; lengths [N].
// Lane t of the crop's wave takes column t (T > 64: chunks of 64 columns, the last class of a chunk carried into the next): the argmax
// is a serial scan of C consecutive floats per lane, the repeat test one lane shuffle, the compaction one ballot + popcount.  HBM / latency-
// bound: N T C 4 bytes in, N T 4 out.
#include "common.hpp"

namespace ocr {
namespace {

constexpr int kCtcWaves = 4;   // crops per workgroup

__global__ __launch_bounds__(64 * kCtcWaves) void ctc_greedy_kernel(const float* __restrict__ logits, int n, int t_len, int c, int blank,
                                                                    int32_t* __restrict__ labels, int32_t* __restrict__ lengths) {
  const int lane = threadIdx.x & 63;
  const int crop = blockIdx.x * kCtcWaves + (threadIdx.x >> 6);
  if (crop >= n) return;   // (whole waves: a wave is one crop)
  const float* x = logits + (size_t)crop * t_len * c;
  int32_t* out = labels + (size_t)crop * t_len;
  int kept = 0;            // wave-uniform
  int carry = -1;          // class of the last column of the previous chunk (-1: there is none)
  for (int t0 = 0; t0 < t_len; t0 += 64) {
    const int t = t0 + lane;
    int a = -1;
    if (t < t_len) {
      const float* col = x + (size_t)t * c;
      float best = col[0];
      a = 0;
      for (int k = 1; k < c; ++k) {
        const float v = col[k];
        if (v > best) {   // strict: the first maximum wins
          best = v;
          a = k;
        }
      }
    }
    int prev = __shfl_up(a, 1, 64);
    if (lane == 0) prev = carry;
    const bool keep = t < t_len && a != blank && a != prev;
    const unsigned long long m = __ballot(keep);
    if (keep) out[kept + __popcll(m & ((1ull << lane) - 1ull))] = a;
    kept += __popcll(m);
    carry = __shfl(a, 63, 64);   // (only read when another chunk follows: then lane 63 held a column)
  }
  for (int i = kept + lane; i < t_len; i += 64) out[i] = -1;
  if (lane == 0) lengths[crop] = kept;
}

}  // namespace

void launch_ctc_greedy(const float* logits_dev, int n, int t, int c, int blank, int32_t* labels_dev, int32_t* lengths_dev, hipStream_t s) {
  if (n <= 0) return;
  if (t <= 0 || c <= 0 || blank < 0 || blank >= c) fail(OCR_ERR_INVALID, "ctc_greedy_decode: T=%d C=%d blank=%d", t, c, blank);
  if ((long long)n * t * c >= (1ll << 40)) fail(OCR_ERR_INVALID, "ctc_greedy_decode: logits too large");
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3((unsigned)((n + kCtcWaves - 1) / kCtcWaves)), dim3(64 * kCtcWaves), 0, s, logits_dev, n, t, c, blank, labels_dev,
                     lengths_dev);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
