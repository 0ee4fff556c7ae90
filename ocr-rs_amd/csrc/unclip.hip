// Unclip on the GPU: for every candidate polygon of a batch - behind its box score - the score threshold, the Clipper-style miter
// offset, the union's simple-ring case, the min-size test and the coordinate adjustment, one lane per polygon.
//   /root/reference/src/text_detection/metrics.rs:100-123   (box_thresh, expand_polygon(.., 2.0), sside < min_size, round(p / adj) as u32)
//   /root/reference/src/polygon.rs:13-42                    (d = area * factor / perimeter, Clipper offset: miter 2.0, closed polygon)
// The arithmetic is postproc_geom.cpp's, operation for operation (f64, separately rounded: this file is compiled with
// -ffp-contract=off; f64 sqrt and divide are correctly rounded on gfx950), including glibc 2.35's hypot kernel for the perimeter
// (geo's euclidean_length calls libm's hypot, which is NOT sqrt(dx^2 + dy^2) to the last bit: 0.6 % of integer pairs differ).
// What is not reproducible bit for bit on the device - libm's atan2 / tan / sin / cos - is never decided here: a lane that
//   * meets a squared-off corner (tan(atan2(..) / 4)),
//   * gets a ring whose self-intersections are anything but the expected ones - at a concave vertex the offset emits (p + n_k d, p,
//     p + n_j d): the two neighbouring offset edges cross once, at X, and the loop X .. p .. X has winding 2; the union's outer
//     boundary takes X (rounded) for the three points.  Exactly these crossings, strictly inside both edges, in order along an edge
//     that has one at either end, are resolved here with exact integers; any other contact between two edges of the ring is the
//     business of postproc_geom.cpp's exact-rational arrangement,
//   * or whose min-area rectangle has a short side within 3 px of min_size (the reference rounds the rectangle's corners outwards
//     after a rotation by an angle from atan2 / fmod / sin / cos: the integerised side can move by < 2 sqrt 2)
// returns UNCLIP_HOST and the host finishes exactly that polygon with postproc_geom.cpp.  Everything else is final here.
#include "common.hpp"
#include "hypot_glibc.hpp"

namespace ocr {
namespace {

struct I2 {
  int x, y;
};

__device__ __forceinline__ long long cround(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

__device__ __forceinline__ unsigned as_u32(double v) {  // Rust `as u32`: saturating, NaN -> 0
  if (!(v > 0.0)) return 0u;
  if (v >= 4294967295.0) return 4294967295u;
  return (unsigned)v;
}

__device__ long long shoelace2(const I2* r, int n) {
  long long s = 0;
  for (int i = 0; i < n; ++i) {
    const I2 a = r[i], b = r[i + 1 == n ? 0 : i + 1];
    s += (long long)a.x * b.y - (long long)b.x * a.y;
  }
  return s;
}

// why a lane hands its polygon back: out_len[j] = -(reason), for tools/unclip_stats.py (the host only looks at the status)
#define PUNT(r)        \
  do {                 \
    out_len[j] = -(r); \
    return;            \
  } while (0)
__global__ __launch_bounds__(64) void unclip_kernel(const BoxScoreJob* __restrict__ jobs, const int32_t* __restrict__ pts_xy, const int* __restrict__ n_jobs_dev,
                                                    int n_jobs, const double* __restrict__ sums, const double* __restrict__ counts,
                                                    const double* __restrict__ adj, UnclipParams prm, I2* __restrict__ work, size_t work_stride,
                                                    uint32_t* __restrict__ out_xy, int32_t* __restrict__ out_len, int32_t* __restrict__ status) {
  constexpr int kLdsPts = 16, kLdsCap = 3 * kLdsPts + 1;   // 49 entries of 8 bytes: an odd stride, lanes spread over the banks
  __shared__ I2 lds_i2[3][64][kLdsCap];
  __shared__ double lds_d[2][64][kLdsCap];
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (n_jobs_dev) n_jobs = min(n_jobs, *n_jobs_dev);
  if (j >= n_jobs) return;
  const BoxScoreJob job = jobs[j];
  out_len[j] = 0;
  const double score = sums[j] / counts[j];
  if (prm.box_thresh > score) {  // metrics.rs:100 (NaN passes, as in the reference)
    status[j] = UNCLIP_DROP;
    return;
  }
  status[j] = UNCLIP_HOST;  // until proven final
  const int np = job.n_pts;
  if (np > kUnclipMaxPts) PUNT(1);
  // five arrays of 3 np (+ 1) entries each, private to the job: source points / crossing points / sorted points, raw ring, vertex
  // kinds / hull, and per ring edge the parameters of the crossings at its two ends
  // The usual polygon (<= 16 points: <= 49 entries per array) keeps them in LDS - the loops below are chains of dependent reads and
  // writes of these arrays (insertion sort, duplicate removal, the ring rebuilt in place), which through global memory cost a round
  // trip each (0.25 ms per batch of 2 000 polygons; in LDS 0.06) - a larger one in the job's global scratch.
  I2* src = work + 3 * (size_t)job.pt_offset + j;
  I2* raw = src + work_stride;
  I2* hull = raw + work_stride;
  double* lo = reinterpret_cast<double*>(hull + work_stride);
  double* hi = lo + work_stride;
  if (np <= kLdsPts) {
    const int t = threadIdx.x;
    src = lds_i2[0][t];
    raw = lds_i2[1][t];
    hull = lds_i2[2][t];
    lo = lds_d[0][t];
    hi = lds_d[1][t];
  }
  I2* kind = hull;
  const int32_t* p = pts_xy + 2 * (size_t)job.pt_offset;

  // ---- raw_offset_ring: closing duplicate and repeated points out, orientation fixed
  int n = 0;
  for (int i = 0; i < np; ++i) {
    const I2 q = {p[2 * i], p[2 * i + 1]};
    if (n == 0 || src[n - 1].x != q.x || src[n - 1].y != q.y) src[n++] = q;
  }
  // (offset_distance works on the polygon as given: repeated points add nothing to the area and exact zeros to the perimeter)
  while (n > 1 && src[0].x == src[n - 1].x && src[0].y == src[n - 1].y) --n;
  if (n < 3) PUNT(2);  // no ring -> no polygon: the reference unwraps None (host: OCR_ERR_DEGENERATE or skip)
  double per = 0.0;
  long long a2 = 0;
  for (int i = 0; i < np; ++i) {
    const int i1 = i + 1 == np ? 0 : i + 1;
    const long long ax = p[2 * i], ay = p[2 * i + 1], bx = p[2 * i1], by = p[2 * i1 + 1];
    a2 += ax * by - bx * ay;
    per += hypot_glibc((double)(bx - ax), (double)(by - ay));
  }
  const double area = fabs((double)a2) / 2.0;
  const double delta = area * prm.unclip_ratio / per;   // polygon.rs:27
  if (shoelace2(src, n) < 0)
    for (int i = 0; i < n / 2; ++i) {
      const I2 t = src[i];
      src[i] = src[n - 1 - i];
      src[n - 1 - i] = t;
    }
  int m = 0, n_concave = 0;
  for (int i = 0; i < 3 * n; ++i) kind[i] = {0, 0};
  {
    auto normal = [&](int i, double& nx, double& ny) {
      const I2 a = src[i], b = src[i + 1 == n ? 0 : i + 1];
      double dx = (double)(b.x - a.x), dy = (double)(b.y - a.y);
      const double f = 1.0 / sqrt(dx * dx + dy * dy);
      dx *= f;
      dy *= f;
      nx = dy;
      ny = -dx;
    };
    double nkx, nky;
    normal(n - 1, nkx, nky);
    for (int jv = 0; jv < n; ++jv) {
      double njx, njy;
      normal(jv, njx, njy);
      const double sx = src[jv].x, sy = src[jv].y;
      double sin_a = nkx * njy - njx * nky;
      bool done = false;
      if (fabs(sin_a * delta) < 1.0) {
        const double cos_a = nkx * njx + njy * nky;
        if (cos_a > 0) {
          raw[m++] = {(int)cround(sx + nkx * delta), (int)cround(sy + nky * delta)};
          done = true;
        }
      } else if (sin_a > 1.0) sin_a = 1.0;
      else if (sin_a < -1.0) sin_a = -1.0;
      if (!done) {
        if (sin_a * delta < 0) {
          kind[m].x = 1;   // first of a concave vertex's three points
          ++n_concave;
          raw[m++] = {(int)cround(sx + nkx * delta), (int)cround(sy + nky * delta)};
          raw[m++] = src[jv];
          raw[m++] = {(int)cround(sx + njx * delta), (int)cround(sy + njy * delta)};
        } else {
          const double r = 1.0 + (njx * nkx + njy * nky);
          if (r >= 0.5) {   // 2 / MiterLimit^2
            const double q = delta / r;
            raw[m++] = {(int)cround(sx + (nkx + njx) * q), (int)cround(sy + (nky + njy) * q)};
          } else {
            PUNT(3);   // squared-off corner: tan(atan2(..) / 4) is libm's -> host
          }
        }
      }
      nkx = njx;
      nky = njy;
    }
  }
  // ---- positive_union_outer: repeated points out; the ring must be simple but for the expected crossing at every concave vertex
  {
    int k = 0;
    for (int i = 0; i < m; ++i)
      if (k == 0 || raw[k - 1].x != raw[i].x || raw[k - 1].y != raw[i].y) raw[k++] = raw[i];
    while (k > 1 && raw[0].x == raw[k - 1].x && raw[0].y == raw[k - 1].y) --k;
    if (k != m && n_concave) PUNT(4);   // (a zero-length piece next to a concave vertex: the host's)
    m = k;
  }
  if (m < 3) PUNT(5);
  for (int i = 0; i < m; ++i) {
    lo[i] = -1.0;
    hi[i] = 2.0;
  }
  for (int i = 0; i < m; ++i) {
    const I2 a = raw[i], b = raw[i + 1 == m ? 0 : i + 1];
    const long long d1x = b.x - a.x, d1y = b.y - a.y;
    for (int q = i + 1; q < m; ++q) {
      const I2 c = raw[q], d = raw[q + 1 == m ? 0 : q + 1];
      const long long d2x = d.x - c.x, d2y = d.y - c.y;
      long long den = d1x * d2y - d1y * d2x;
      const bool adjacent = q == i + 1 || (i == 0 && q == m - 1);
      if (adjacent) {
        if (den == 0) PUNT(6);   // collinear neighbours (a spike or a straight vertex)
        continue;
      }
      const long long wx = c.x - a.x, wy = c.y - a.y;
      if (den == 0) {
        if (wx * d1y - wy * d1x == 0) {
          // on one line (two stretches of a word's long side with a dent between them): harmless when strictly apart - the
          // arrangement ignores parallel pairs as well -, the host's when they overlap or touch
          const long long l2 = d1x * d1x + d1y * d1y;
          const long long tc = wx * d1x + wy * d1y, td = (long long)(d.x - a.x) * d1x + (long long)(d.y - a.y) * d1y;
          if (!((tc < 0 && td < 0) || (tc > l2 && td > l2))) PUNT(7);
        }
        continue;
      }
      long long tn = wx * d2y - wy * d2x, un = wx * d1y - wy * d1x;
      if (den < 0) {
        den = -den;
        tn = -tn;
        un = -un;
      }
      if (tn < 0 || tn > den || un < 0 || un > den) continue;
      // the two edges meet.  Expected only as (edge into a concave vertex's first point, edge out of its third point).
      const int t_fwd = i + 1, t_rev = q + 1 == m ? 0 : q + 1;   // the concave triple would start here
      const bool fwd = t_fwd < m && kind[t_fwd].x == 1 && q == i + 3;
      const bool rev = !fwd && kind[t_rev].x == 1 && t_rev + 2 == i;   // (only the triple at the ring's start: edge m - 1 into it, edge 2 out of it)
      if (!(fwd || rev) || tn == 0 || tn == den || un == 0 || un == den) PUNT(8);
      const int t = fwd ? t_fwd : t_rev;
      if (kind[t].y == 1) PUNT(9);   // (twice: cannot be)
      kind[t].y = 1;
      // X = a + tn / den (b - a), rounded half away from zero like cround((double)((long double)xn / den)): exact for these magnitudes
      const long long xn = (long long)a.x * den + tn * d1x, yn = (long long)a.y * den + tn * d1y;
      auto rnd = [](long long num, long long dd) { return num >= 0 ? (2 * num + dd) / (2 * dd) : -((-2 * num + dd) / (2 * dd)); };
      src[t] = {(int)rnd(xn, den), (int)rnd(yn, den)};
      if (fwd) {   // edge i ends at the crossing, edge q starts from it
        hi[i] = (double)tn / (double)den;
        lo[q] = (double)un / (double)den;
      } else {
        hi[q] = (double)un / (double)den;
        lo[i] = (double)tn / (double)den;
      }
    }
  }
  {
    int k = 0;
    for (int i = 0; i < m; ++i) {
      if (!(lo[i] + 1e-9 < hi[i])) PUNT(10);   // the crossings at an edge's two ends out of order (or too close to call)
      if (kind[i].x == 1) {
        if (kind[i].y != 1) PUNT(11);          // a concave vertex whose neighbours do not cross: not the simple picture
        raw[k++] = src[i];
        i += 2;
      } else {
        raw[k++] = raw[i];
      }
    }
    m = k;
  }
  // FixupOutPolygon: repeated and collinear vertices out (rounded crossings can make them), one at a time from the front, as on the host
  for (bool changed = true; changed && m >= 3;) {
    changed = false;
    for (int i = 0; i < m; ++i) {
      const I2 pv = raw[i == 0 ? m - 1 : i - 1], c = raw[i], nx = raw[i + 1 == m ? 0 : i + 1];
      if ((c.x == nx.x && c.y == nx.y) || (c.x == pv.x && c.y == pv.y) ||
          (long long)(c.y - pv.y) * (nx.x - c.x) == (long long)(c.x - pv.x) * (nx.y - c.y)) {
        for (int k = i; k + 1 < m; ++k) raw[k] = raw[k + 1];
        --m;
        changed = true;
        break;
      }
    }
  }
  if (m < 3) PUNT(12);
  if (shoelace2(raw, m) <= 0) PUNT(13);   // nothing of positive winding, or a ring the rounding turned over: the host decides what that means

  // ---- min-area rectangle, conservatively: convex hull (monotone chain on the lexicographically sorted points), then per hull edge
  // the extents along and across it.  The reference's short side is that of the integerised rectangle: < 2 sqrt 2 away.
  for (int i = 0; i < m; ++i) {   // insertion sort of a copy (src is free now; m <= 3 n fits its 3 np + 1 entries)
    const I2 q = raw[i];
    int k = i;
    while (k > 0 && (src[k - 1].x > q.x || (src[k - 1].x == q.x && src[k - 1].y > q.y))) {
      src[k] = src[k - 1];
      --k;
    }
    src[k] = q;
  }
  auto cross = [](const I2 o, const I2 a, const I2 b) { return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x); };
  int hn = 0;
  for (int i = 0; i < m; ++i) {
    while (hn >= 2 && cross(hull[hn - 2], hull[hn - 1], src[i]) <= 0) --hn;
    hull[hn++] = src[i];
  }
  for (int i = m - 2, lo = hn + 1; i >= 0; --i) {
    while (hn >= lo && cross(hull[hn - 2], hull[hn - 1], src[i]) <= 0) --hn;
    hull[hn++] = src[i];
  }
  --hn;   // the last point repeats the first
  if (hn < 3) PUNT(14);
  double best_area = INFINITY;
  for (int pass = 0; pass < 2; ++pass) {
    double short_min = INFINITY;
    for (int e = 0; e < hn; ++e) {
      const I2 a = hull[e], b = hull[e + 1 == hn ? 0 : e + 1];
      const long long ex = b.x - a.x, ey = b.y - a.y;
      long long amin = 0, amax = 0, cmin = 0, cmax = 0;
      for (int i = 0; i < hn; ++i) {
        const long long px = hull[i].x - a.x, py = hull[i].y - a.y;
        const long long al = ex * px + ey * py, ac = ex * py - ey * px;
        amin = min(amin, al);
        amax = max(amax, al);
        cmin = min(cmin, ac);
        cmax = max(cmax, ac);
      }
      const double l2 = (double)(ex * ex + ey * ey);
      const double da = (double)(amax - amin), dc = (double)(cmax - cmin);
      const double ar = da * dc / l2;
      if (pass == 0) best_area = fmin(best_area, ar);
      else if (ar <= best_area * (1.0 + 1e-9) + 1e-9) short_min = fmin(short_min, fmin(da, dc) / sqrt(l2));   // every orientation the reference may settle on
    }
    if (pass == 1 && !(short_min > prm.min_size + 3.0)) PUNT(15);   // too close to call (or too small): the host decides
  }
  // ---- metrics.rs:108-121: BuildResult order (ends at the top-most vertex, ties: right-most), round(p / adj) as u32
  int top = 0;
  for (int i = 1; i < m; ++i)
    if (raw[i].y < raw[top].y || (raw[i].y == raw[top].y && raw[i].x > raw[top].x)) top = i;
  const double adj_x = adj[2 * job.image], adj_y = adj[2 * job.image + 1];
  uint32_t* o = out_xy + 2 * (3 * (size_t)job.pt_offset);
  for (int i = 0; i < m; ++i) {
    int k = top + 1 + i;
    if (k >= m) k -= m;
    o[2 * i] = as_u32(round((double)raw[k].x / adj_x));
    o[2 * i + 1] = as_u32(round((double)raw[k].y / adj_y));
  }
  out_len[j] = m;
  status[j] = UNCLIP_KEEP;
}

}  // namespace

size_t unclip_work_bytes(size_t total_pts, int n_jobs) { return 5 * (3 * total_pts + (size_t)n_jobs + 1) * sizeof(I2); }

void launch_unclip(const BoxScoreJob* jobs_dev, const int32_t* pts_xy_dev, const int* n_jobs_dev, int n_jobs, size_t total_pts, const double* sums_dev,
                   const double* counts_dev, const double* adj_dev, const UnclipParams& prm, void* work_dev, uint32_t* out_xy_dev,
                   int32_t* out_len_dev, int32_t* status_dev, hipStream_t s) {
  if (n_jobs <= 0) return;
  const size_t stride = 3 * total_pts + (size_t)n_jobs + 1;
  hipLaunchKernelGGL(unclip_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, s, jobs_dev, pts_xy_dev, n_jobs_dev, n_jobs, sums_dev, counts_dev, adj_dev, prm,
                     static_cast<I2*>(work_dev), stride, out_xy_dev, out_len_dev, status_dev);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
