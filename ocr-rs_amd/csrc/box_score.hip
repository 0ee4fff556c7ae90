// box_score_fast on the GPU: masked mean of the probability map over a rasterised
// polygon, one workgroup per candidate polygon.
//   /root/reference/src/text_detection/metrics.rs:150-184
// The mask is imageproc 0.22.0 `draw_polygon_mut` (not vendored in the reference):
// scanline fill between sorted, f32-rounded edge intersections, then a Bresenham
// outline of every edge.  Both are evaluated in closed form per pixel so that no
// sorting or serial line walking is needed:
//   fill   : with s = multiset of a row's intersections, x is filled iff some s == x
//            or #{s < x} is odd (pairs (s0,s1),(s2,s3).. of the sorted list)
//   outline: the k-th pixel of imageproc's BresenhamLineIter sits at
//            y0 + ystep * floor((2*k*dy + dx - 1) / (2*dx))   (error starts at dx/2)
// The mask lives in an LDS bit image, processed in row bands; values are summed in
// f64 like `sum(Kind::Double)`.
#include "common.hpp"

namespace ocr {
namespace {

constexpr int BAND_WORDS = 8192;  // 32 KiB of mask bits per band

__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) v += __shfl_xor(v, k, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void box_score_kernel(const float* __restrict__ prob, int H, int W,
                                                        const BoxScoreJob* __restrict__ jobs,
                                                        const int32_t* __restrict__ pts_xy,
                                                        double* __restrict__ sums, double* __restrict__ counts, int n_jobs,
                                                        const int* __restrict__ n_jobs_dev) {
  __shared__ int px[kBoxScoreMaxPts], py[kBoxScoreMaxPts];
  __shared__ unsigned mask[BAND_WORDS];
  __shared__ double red[4];
  // one workgroup per job - or, when the job count only exists on the device (candidates.hip), a fixed grid walking the list
  if (n_jobs_dev) n_jobs = *n_jobs_dev;
  for (int jb = blockIdx.x; jb < n_jobs; jb += gridDim.x) {
  const BoxScoreJob job = jobs[jb];
  const int tid = threadIdx.x;
  const int np = job.n_pts;
  for (int i = tid; i < np; i += 256) {  // moved_points: relative to the canvas origin
    px[i] = pts_xy[2 * (job.pt_offset + i)] - job.min_x;
    py[i] = pts_xy[2 * (job.pt_offset + i) + 1] - job.min_y;
  }
  const int bw = job.bw, bh = job.bh;
  const int wpr = (bw + 31) >> 5;  // mask words per row
  const int band_rows = max(1, BAND_WORDS / wpr);
  const float* pmap = prob + (size_t)job.image * H * W;
  double sum = 0.0, cnt = 0.0;
  __syncthreads();

  for (int y0 = 0; y0 < bh; y0 += band_rows) {
    const int rows = min(band_rows, bh - y0);
    for (int i = tid; i < rows * wpr; i += 256) mask[i] = 0u;
    __syncthreads();
    // ---- scanline fill
    for (int i = tid; i < rows * bw; i += 256) {
      const int ry = i / bw, x = i - ry * bw;
      const int y = y0 + ry;
      int c_lt = 0, c_le = 0;
      for (int e = 0; e < np; ++e) {
        const int ax = px[e], ay = py[e];
        const int e1 = e + 1 == np ? 0 : e + 1;
        const int bx = px[e1], by = py[e1];
        if ((ay <= y && by >= y) || (by <= y && ay >= y)) {
          if (ay == by) {
            c_lt += (ax < x) + (bx < x);
            c_le += (ax <= x) + (bx <= x);
          } else if (ay == y || by == y) {
            if (by > y) {
              c_lt += ax < x;
              c_le += ax <= x;
            }
            if (ay > y) {
              c_lt += bx < x;
              c_le += bx <= x;
            }
          } else {
            // f32, separately rounded divide / multiply / add (no FMA contraction), round half away
            const float frac = __fdiv_rn((float)(y - ay), (float)(by - ay));
            const float inter = __fadd_rn((float)ax, __fmul_rn(frac, (float)(bx - ax)));
            const int s = (int)roundf(inter);
            c_lt += s < x;
            c_le += s <= x;
          }
        }
      }
      if (c_le > c_lt || (c_lt & 1)) atomicOr(&mask[ry * wpr + (x >> 5)], 1u << (x & 31));
    }
    // ---- Bresenham outline of every edge
    for (int e = 0; e < np; ++e) {
      const int e1 = e + 1 == np ? 0 : e + 1;
      int ax = px[e], ay = py[e], bx = px[e1], by = py[e1];
      const bool steep = abs(by - ay) > abs(bx - ax);
      if (steep) {
        int t = ax; ax = ay; ay = t;
        t = bx; bx = by; by = t;
      }
      if (ax > bx) {
        int t = ax; ax = bx; bx = t;
        t = ay; ay = by; by = t;
      }
      const int dx = bx - ax, dy = abs(by - ay), ystep = ay < by ? 1 : -1;
      for (int k = tid; k <= dx; k += 256) {
        const int m = dx == 0 ? 0 : (int)((2ll * k * dy + dx - 1) / (2ll * dx));
        const int X = ax + k, Y = ay + ystep * m;
        const int cx = steep ? Y : X, cy = steep ? X : Y;
        if (cx >= 0 && cx < bw && cy >= y0 && cy < y0 + rows)
          atomicOr(&mask[(cy - y0) * wpr + (cx >> 5)], 1u << (cx & 31));
      }
    }
    __syncthreads();
    // ---- masked sum
    for (int i = tid; i < rows * wpr; i += 256) {
      unsigned bits = mask[i];
      const int ry = i / wpr, wx = i - ry * wpr;
      const float* row = pmap + (size_t)(job.min_y + y0 + ry) * W + job.min_x + (wx << 5);
      while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        sum += (double)row[b];
        cnt += 1.0;
      }
    }
    __syncthreads();
  }
  const double ts = block_sum(sum, red);
  const double tc = block_sum(cnt, red);
  if (tid == 0) {
    sums[jb] = ts;
    counts[jb] = tc;
  }
  __syncthreads();
  }
}

}  // namespace

void launch_box_scores(const float* prob, int H, int W, const BoxScoreJob* jobs_dev, const int32_t* pts_xy_dev,
                       int n_jobs, double* sums_dev, double* counts_dev, hipStream_t s) {
  if (n_jobs <= 0) return;
  hipLaunchKernelGGL(box_score_kernel, dim3(n_jobs), dim3(256), 0, s, prob, H, W, jobs_dev, pts_xy_dev, sums_dev,
                     counts_dev, n_jobs, static_cast<const int*>(nullptr));
  OCR_HIP(hipGetLastError());
}

void launch_box_scores_counted(const float* prob, int H, int W, const BoxScoreJob* jobs_dev, const int32_t* pts_xy_dev, const int* n_jobs_dev, int grid,
                               double* sums_dev, double* counts_dev, hipStream_t s) {
  if (grid <= 0) return;
  hipLaunchKernelGGL(box_score_kernel, dim3(grid), dim3(256), 0, s, prob, H, W, jobs_dev, pts_xy_dev, sums_dev, counts_dev, 0, n_jobs_dev);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
