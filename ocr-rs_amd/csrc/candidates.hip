// From device contours to box-score jobs without the host: arc length, Douglas-Peucker and the >= 4 points filter per contour
// (one wave each), then the candidates of the whole batch compacted - in image and contour order, which is the order of the
// reference's result lists - into the job / point arrays that box_score.hip and unclip.hip read.
//   /root/reference/src/text_detection/metrics.rs:86-98   epsilon = 0.01 * arc_length(contour, closed), approximate_polygon_dp, pop a
//                                                          repeated end point, skip polygons with fewer than 4 points
//   metrics.rs:151-166                                     the clamped bounding box of box_score_fast (x by H, y by W: the reference's quirk)
// imageproc 0.22.0's arc_length / approximate_polygon_dp are restated in postproc_geom.cpp (pinned to the reference's known answers);
// this file is that code's arithmetic on the device, compiled with -ffp-contract=off:
//   * arc length: the f64 sum of the segment lengths IN ORDER (sqrt of an exact integer; every lane adds the same 64 values one after
//     the other, read with v_readlane - a parallel sum would round differently);
//   * Douglas-Peucker: distance to the infinite line through the end points, |a x + b y + c| / sqrt(a^2 + b^2) with an exact integer
//     numerator; the FIRST index attaining the maximum wins (the sequential scan's strict >): per-lane strided scans keep their first
//     maximum, the cross-lane reduction prefers the smaller index among equal distances; split while dmax > epsilon.  The recursion
//     is an explicit stack (smaller half first: depth <= 16), the result "all kept indices in increasing order", as on the host.
#include "common.hpp"

namespace ocr {
namespace {

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
constexpr int kDpWaves = 4;
constexpr int kKeepWords = 1024;   // contour points per wave: 32 768 (the tracer's per-image capacity)

__device__ __forceinline__ double readlane_f64(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int px(uint32_t p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ int py(uint32_t p) { return (int)(p >> 16); }
__device__ __forceinline__ double seg_len(uint32_t p, uint32_t q) {
  const double dx = (double)px(p) - (double)px(q), dy = (double)py(p) - (double)py(q);
  return sqrt(dx * dx + dy * dy);
}

// hdr [n][4] = {contours, points, status, -}, pts [n][cap] (y << 16 | x), starts [n][maxc + 1]: contours.hip.
// cand_pts [n][cap]: the polygon of contour c at its contour's offset; cand_len [n][maxc]: its length, 0 = no candidate.
__global__ __launch_bounds__(64 * kDpWaves) void dp_kernel(const int* __restrict__ hdr_all, const uint32_t* __restrict__ pts_all, int cap,
                                                           const int* __restrict__ starts_all, int maxc, uint32_t* __restrict__ cand_pts,
                                                           int* __restrict__ cand_len, int wgs_per_image) {
  __shared__ uint32_t keep_s[kDpWaves][kKeepWords];
  __shared__ int stack_s[kDpWaves][2 * 64];
  const int img = blockIdx.x / wgs_per_image, wg = blockIdx.x % wgs_per_image;
  const int* hdr = hdr_all + 4 * img;
  if (hdr[2] != 0) return;   // the tracer gave this image up: the host takes all of it
  const int nc = hdr[0];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t* keep = keep_s[wave];
  int* stack = stack_s[wave];
  const int* st = starts_all + (size_t)img * (maxc + 1);
  for (int c = wg * kDpWaves + wave; c < nc; c += wgs_per_image * kDpWaves) {
    const int s0 = st[c], L = st[c + 1] - s0;
    const uint32_t* P = pts_all + (size_t)img * cap + s0;
    int m = 0;   // points of the candidate
    if (L >= 4 && L <= 32 * kKeepWords) {   // (fewer than 4 contour points cannot leave 4 polygon points)
      // ---- arc_length(contour, closed = true)
      double len = 0.0;
      for (int base = 0; base < L - 1; base += 64) {
        const int i = base + lane;
        const double d = i + 1 < L ? seg_len(P[i], P[i + 1]) : 0.0;
        const int cnt = min(64, L - 1 - base);
        for (int k = 0; k < cnt; ++k) len += readlane_f64(d, k);
      }
      if (L > 2) len += seg_len(P[0], P[L - 1]);
      double eps = 0.01 * len;   // metrics.rs:87-90
      if (eps == 0.0) eps = 0.01;
      // ---- approximate_polygon_dp(contour, eps, closed = true)
      const int nwords = (L + 31) >> 5;
      for (int w = lane; w < nwords; w += 64) keep[w] = 0u;
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        atomicOr(&keep[0], 1u);
        atomicOr(&keep[(L - 1) >> 5], 1u << ((L - 1) & 31));
        stack[0] = 0;
        stack[1] = L - 1;
      }
      int sp = 1;
      while (sp > 0) {
        __builtin_amdgcn_wave_barrier();
        --sp;
        const int lo = stack[2 * sp], hi = stack[2 * sp + 1];
        __builtin_amdgcn_wave_barrier();
        if (hi - lo < 2) continue;   // the end point itself is at distance 0: nothing can exceed eps
        const uint32_t p0 = P[lo], p1 = P[hi];
        const double x0 = px(p0), y0 = py(p0), x1 = px(p1), y1 = py(p1);
        const double a = y0 - y1, b = x1 - x0, cc = x0 * y1 - x1 * y0;
        const double norm = sqrt(a * a + b * b);
        double best = 0.0;
        int bidx = lo;
        for (int i = lo + 1 + lane; i < hi; i += 64) {
          const uint32_t q = P[i];
          const double d = fabs(a * (double)px(q) + b * (double)py(q) + cc) / norm;   // NaN when both ends are one point: never greater
          if (d > best) {
            best = d;
            bidx = i;
          }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const double od = __shfl_xor(best, off, 64);
          const int oi = __shfl_xor(bidx, off, 64);
          if (od > best || (od == best && oi < bidx)) {
            best = od;
            bidx = oi;
          }
        }
        if (best > eps) {
          if (lane == 0) {
            atomicOr(&keep[bidx >> 5], 1u << (bidx & 31));
            // the larger half below the smaller one: the stack never holds more than log2(L) + 1 ranges
            const bool left_larger = bidx - lo > hi - bidx;
            stack[2 * sp] = left_larger ? lo : bidx;
            stack[2 * sp + 1] = left_larger ? bidx : hi;
            stack[2 * sp + 2] = left_larger ? bidx : lo;
            stack[2 * sp + 3] = left_larger ? hi : bidx;
          }
          sp += 2;
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- kept points in order; closed: the last point (index L - 1) goes; a repeated end point goes too (metrics.rs:92-94)
      uint32_t* out = cand_pts + (size_t)img * cap + s0;
      int total = 0;
      for (int wb = 0; wb < nwords; wb += 64) {
        const int w = wb + lane;
        uint32_t bits = w < nwords ? keep[w] : 0u;
        if (w == ((L - 1) >> 5)) bits &= ~(1u << ((L - 1) & 31));
        const int cnt = __popc(bits);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(incl, off, 64);
          if (lane >= off) incl += v;
        }
        int pos = total + incl - cnt;
        while (bits) {
          const int bpos = __ffs(bits) - 1;
          bits &= bits - 1;
          out[pos++] = P[(w << 5) + bpos];
        }
        total += __shfl(incl, 63, 64);
      }
      m = total;
      if (m > 1) {
        // index of the last kept point below L - 1
        int last = 0;
        for (int w = (L - 2) >> 5; w >= 0; --w) {
          uint32_t bits = keep[w];
          if (w == ((L - 2) >> 5) && ((L - 2) & 31) != 31) bits &= (2u << ((L - 2) & 31)) - 1u;
          if (bits) {
            last = (w << 5) + 31 - __clz(bits);
            break;
          }
        }
        if (P[last] == P[0]) --m;
      }
      if (m < 4) m = 0;
    } else if (L > 32 * kKeepWords) {
      m = -1;   // cannot happen with the tracer's capacity; marks the image for the host all the same
    }
    if (lane == 0) cand_len[(size_t)img * maxc + c] = m;
  }
}

// per image: candidates and candidate points, exclusive offsets of every contour inside its image.  tot [n][2] = {candidates, points}
// ( -1, -1 for an image the host must take: tracer status, or a contour this path does not hold)
__global__ __launch_bounds__(256) void cand_count_kernel(const int* __restrict__ hdr_all, const int* __restrict__ cand_len, int maxc, int* __restrict__ cand_idx,
                                                         int* __restrict__ cand_off, int* __restrict__ tot) {
  __shared__ int s_cnt[256], s_pts[256];
  const int img = blockIdx.x, t = threadIdx.x;
  const int* hdr = hdr_all + 4 * img;
  if (hdr[2] != 0) {
    if (t == 0) tot[2 * img] = tot[2 * img + 1] = -1;
    return;
  }
  const int nc = hdr[0];
  const int* len = cand_len + (size_t)img * maxc;
  int run_c = 0, run_p = 0, bad = 0;
  for (int base = 0; base < nc; base += 256) {
    const int c = base + t;
    const int L = c < nc ? len[c] : 0;
    bad |= L < 0 || L > kBoxScoreMaxPts;
    const int has = L > 0 ? 1 : 0, np = L > 0 ? L : 0;
    s_cnt[t] = has;
    s_pts[t] = np;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive scan
      const int a = t >= off ? s_cnt[t - off] : 0, b = t >= off ? s_pts[t - off] : 0;
      __syncthreads();
      s_cnt[t] += a;
      s_pts[t] += b;
      __syncthreads();
    }
    if (c < nc) {
      cand_idx[(size_t)img * maxc + c] = run_c + s_cnt[t] - has;
      cand_off[(size_t)img * maxc + c] = run_p + s_pts[t] - np;
    }
    run_c += s_cnt[255];
    run_p += s_pts[255];
    __syncthreads();
  }
  bad = __syncthreads_or(bad);
  if (t == 0) {
    tot[2 * img] = bad ? -1 : run_c;
    tot[2 * img + 1] = bad ? -1 : run_p;
  }
}

// the batch's job list: jobs [sum candidates], pts_xy [sum points][2]; totals = {jobs, points, overflow}
__global__ __launch_bounds__(256) void cand_fill_kernel(const int* __restrict__ hdr_all, const uint32_t* __restrict__ cand_pts, int cap,
                                                        const int* __restrict__ starts_all, const int* __restrict__ cand_len, int maxc,
                                                        const int* __restrict__ cand_idx, const int* __restrict__ cand_off, const int* __restrict__ tot, int n,
                                                        int H, int W, BoxScoreJob* __restrict__ jobs, int max_jobs, int32_t* __restrict__ pts_xy, int max_pts,
                                                        int* __restrict__ totals) {
  __shared__ int s_a[256], s_b[256];
  const int img = blockIdx.x, t = threadIdx.x;
  // offsets of this image in the batch's lists (and, in the last block, the totals)
  int a = 0, b = 0;
  for (int j = t; j < img; j += 256)
    if (tot[2 * j] > 0) {
      a += tot[2 * j];
      b += tot[2 * j + 1];
    }
  s_a[t] = a;
  s_b[t] = b;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (t < off) {
      s_a[t] += s_a[t + off];
      s_b[t] += s_b[t + off];
    }
    __syncthreads();
  }
  const int job0 = s_a[0], pt0 = s_b[0];
  const int my_c = max(tot[2 * img], 0), my_p = max(tot[2 * img + 1], 0);
  const bool fits = job0 + my_c <= max_jobs && pt0 + my_p <= max_pts;
  if (img == n - 1 && t == 0) {
    // the lists are prefixes: if the last image fits, all do.  An overflowing batch reports ZERO jobs - the kernels behind this one
    // (box scores, unclip) walk the list by this count and must not run past what was written - and the host path takes the batch
    totals[0] = fits ? job0 + my_c : 0;
    totals[1] = fits ? pt0 + my_p : 0;
    totals[2] = fits ? 0 : 1;
  }
  if (tot[2 * img] <= 0 || !fits) return;
  const int nc = hdr_all[4 * img];
  const int* st = starts_all + (size_t)img * (maxc + 1);
  for (int c = t; c < nc; c += 256) {
    const int L = cand_len[(size_t)img * maxc + c];
    if (L <= 0) continue;
    const uint32_t* src = cand_pts + (size_t)img * cap + st[c];
    const int off = pt0 + cand_off[(size_t)img * maxc + c];
    int mnx = INT32_MAX, mxx = 0, mny = INT32_MAX, mxy = 0;
    for (int i = 0; i < L; ++i) {
      const int x = px(src[i]), y = py(src[i]);
      pts_xy[2 * (off + i)] = x;
      pts_xy[2 * (off + i) + 1] = y;
      mnx = min(mnx, x);
      mxx = max(mxx, x);
      mny = min(mny, y);
      mxy = max(mxy, y);
    }
    // the reference clamps x by size[-2] (= H) and y by size[-1] (= W): metrics.rs:151-166 (api.hip refuses non-square maps where that matters)
    mnx = min(max(mnx, 0), H - 1);
    mxx = min(max(mxx, 0), H - 1);
    mny = min(max(mny, 0), W - 1);
    mxy = min(max(mxy, 0), W - 1);
    jobs[job0 + cand_idx[(size_t)img * maxc + c]] = BoxScoreJob{img, off, L, mnx, mny, mxx - mnx + 1, mxy - mny + 1};
  }
}

}  // namespace

size_t candidates_scratch_bytes(int n, int cap, int maxc) { return align256((size_t)n * cap * 4) + 3 * align256((size_t)n * maxc * 4) + 256; }

void launch_candidates(const int* hdr, const uint32_t* pts, int cap, const int* starts, int maxc, int n, int h, int w, void* scratch, BoxScoreJob* jobs,
                       int max_jobs, int32_t* pts_xy, int max_pts, int* tot, int* totals, hipStream_t s) {
  if (n <= 0) return;
  char* base = static_cast<char*>(scratch);
  uint32_t* cand_pts = reinterpret_cast<uint32_t*>(base);
  base += align256((size_t)n * cap * 4);
  int* cand_len = reinterpret_cast<int*>(base);
  base += align256((size_t)n * maxc * 4);
  int* cand_idx = reinterpret_cast<int*>(base);
  base += align256((size_t)n * maxc * 4);
  int* cand_off = reinterpret_cast<int*>(base);
  const int wgs = 8;   // 32 waves per image: a dense page's 60 - 130 contours in two to four rounds
  hipLaunchKernelGGL(dp_kernel, dim3((unsigned)(n * wgs)), dim3(64 * kDpWaves), 0, s, hdr, pts, cap, starts, maxc, cand_pts, cand_len, wgs);
  OCR_HIP(hipGetLastError());
  hipLaunchKernelGGL(cand_count_kernel, dim3((unsigned)n), dim3(256), 0, s, hdr, cand_len, maxc, cand_idx, cand_off, tot);
  OCR_HIP(hipGetLastError());
  hipLaunchKernelGGL(cand_fill_kernel, dim3((unsigned)n), dim3(256), 0, s, hdr, cand_pts, cap, starts, cand_len, maxc, cand_idx, cand_off, tot, n, h, w, jobs,
                     max_jobs, pts_xy, max_pts, totals);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
