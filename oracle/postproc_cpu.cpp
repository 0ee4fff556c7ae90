// CPU post-processing baseline (TEST INFRASTRUCTURE: timed by bench.py's cpu_baseline leg, pinned by tests/ - never
// linked into or called by the product).  get_boxes_and_box_scores of /root/reference/src/text_detection/metrics.rs:37-127
// entirely on host threads, compiled: binarize (:129-131) into a bit image, the library's host geometry for contours,
// Douglas-Peucker, unclip and filters (ocr-rs_amd/csrc/postproc_geom.cpp, itself pinned to the reference's known answers
// through oracle/postproc_oracle.py), and box_score_fast (:150-184) as a scalar restatement of imageproc 0.22.0's
// draw_polygon_mut: scanline fill between sorted, f32-rounded edge intersections plus a Bresenham outline of every edge.
// One image per task, `threads` std::threads.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../ocr-rs_amd/csrc/postproc_geom.hpp"

namespace {
using ocr::geom::Pt;

// masked mean over the rasterised polygon; x clamped by H and y by W as the reference does (metrics.rs:151-166)
double box_score_cpu(const float* map, int H, int W, const std::vector<Pt>& c) {
  int mnx = INT_MAX, mxx = 0, mny = INT_MAX, mxy = 0;
  for (const Pt& p : c) {
    mnx = std::min(mnx, p.x);
    mxx = std::max(mxx, p.x);
    mny = std::min(mny, p.y);
    mxy = std::max(mxy, p.y);
  }
  mnx = std::clamp(mnx, 0, H - 1);
  mxx = std::clamp(mxx, 0, H - 1);
  mny = std::clamp(mny, 0, W - 1);
  mxy = std::clamp(mxy, 0, W - 1);
  if (mxx >= W || mxy >= H) return -1.0;  // non-square map: the reference fails in narrow()
  const int bw = mxx - mnx + 1, bh = mxy - mny + 1, np = (int)c.size();
  std::vector<uint8_t> mask((size_t)bw * bh, 0);
  std::vector<int> px(np), py(np), xs;
  for (int i = 0; i < np; ++i) {
    px[i] = c[i].x - mnx;
    py[i] = c[i].y - mny;
  }
  for (int y = 0; y < bh; ++y) {   // scanline fill: pairs of the sorted intersections, both ends inclusive
    xs.clear();
    for (int e = 0; e < np; ++e) {
      const int e1 = e + 1 == np ? 0 : e + 1;
      const int ax = px[e], ay = py[e], bx = px[e1], by = py[e1];
      if (!((ay <= y && by >= y) || (by <= y && ay >= y))) continue;
      if (ay == by) {
        xs.push_back(ax);
        xs.push_back(bx);
      } else if (ay == y || by == y) {
        if (by > y) xs.push_back(ax);
        if (ay > y) xs.push_back(bx);
      } else {
        const float frac = (float)(y - ay) / (float)(by - ay);
        const float prod = frac * (float)(bx - ax);   // separately rounded (built with -ffp-contract=off)
        xs.push_back((int)std::round((float)ax + prod));
      }
    }
    std::sort(xs.begin(), xs.end());
    for (size_t i = 0; i + 1 < xs.size(); i += 2)
      for (int x = std::max(xs[i], 0); x <= std::min(xs[i + 1], bw - 1); ++x) mask[(size_t)y * bw + x] = 1;
  }
  for (int e = 0; e < np; ++e) {   // Bresenham outline (imageproc BresenhamLineIter: error starts at dx / 2)
    const int e1 = e + 1 == np ? 0 : e + 1;
    int ax = px[e], ay = py[e], bx = px[e1], by = py[e1];
    const bool steep = std::abs(by - ay) > std::abs(bx - ax);
    if (steep) {
      std::swap(ax, ay);
      std::swap(bx, by);
    }
    if (ax > bx) {
      std::swap(ax, bx);
      std::swap(ay, by);
    }
    const int dx = bx - ax, dy = std::abs(by - ay), ystep = ay < by ? 1 : -1;
    for (int k = 0; k <= dx; ++k) {
      const int m = dx == 0 ? 0 : (int)((2ll * k * dy + dx - 1) / (2ll * dx));
      const int X = ax + k, Y = ay + ystep * m;
      const int cx = steep ? Y : X, cy = steep ? X : Y;
      if (cx >= 0 && cx < bw && cy >= 0 && cy < bh) mask[(size_t)cy * bw + cx] = 1;
    }
  }
  double sum = 0.0, cnt = 0.0;
  for (int y = 0; y < bh; ++y)
    for (int x = 0; x < bw; ++x)
      if (mask[(size_t)y * bw + x]) {
        sum += (double)map[(size_t)(mny + y) * W + mnx + x];
        cnt += 1.0;
      }
  return sum / cnt;
}
}  // namespace

// prob: N x 1 x H x W f32; adj: N x 2.  Polygons come back image by image: img_polys[N] counts, lens (vertices per polygon),
// xy pairs, scores - up to the given capacities (the counts are exact even when the arrays are too small or null).
// Returns 0, or 6 when a zero-area candidate made expand_polygon fail and skip_degenerate is 0 (the reference aborts).
extern "C" int postproc_cpu(const float* prob, int n, int h, int w, const double* adj, double thresh, double box_thresh,
                            double min_size, double unclip_ratio, int skip_degenerate, int threads, int* img_polys,
                            int* total_polys, int* total_vertices, int32_t* lens, int lens_cap, uint32_t* xy, int xy_cap,
                            double* scores, int scores_cap) {
  struct PerImage {
    std::vector<uint32_t> xy;
    std::vector<int32_t> lens;
    std::vector<double> scores;
  };
  std::vector<PerImage> per(n);
  ocr_postproc_params_t prm{};
  prm.thresh = thresh;
  prm.box_thresh = box_thresh;
  prm.min_size = min_size;
  prm.unclip_ratio = unclip_ratio;
  prm.skip_degenerate = skip_degenerate;
  std::atomic<int> next{0}, failed{0};
  auto work = [&] {
    const size_t hw = (size_t)h * w, wpi = (hw + 63) / 64 * 2;
    std::vector<uint32_t> bits(wpi);
    std::vector<std::vector<Pt>> cands;
    for (int b = next.fetch_add(1); b < n; b = next.fetch_add(1)) {
      const float* map = prob + (size_t)b * hw;
      std::fill(bits.begin(), bits.end(), 0u);
      for (size_t i = 0; i < hw; ++i)
        if (map[i] > (float)thresh) bits[i >> 5] |= 1u << (i & 31);   // binarize, metrics.rs:129-131
      cands.clear();
      ocr::geom::contour_candidates_bits(bits.data(), h, w, cands);
      PerImage& r = per[b];
      try {
        for (const auto& c : cands) {
          const double score = box_score_cpu(map, h, w, c);
          const size_t before = r.xy.size();
          if (ocr::geom::finish_polygon(c, score, adj[2 * b], adj[2 * b + 1], prm, r.xy)) {
            r.lens.push_back((int32_t)((r.xy.size() - before) / 2));
            r.scores.push_back(score);
          }
        }
      } catch (const ocr::geom::DegeneratePolygon&) {
        failed.store(1);
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < std::max(1, threads); ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (failed.load()) return 6;
  int np = 0, nv = 0;
  for (int b = 0; b < n; ++b) {
    if (img_polys) img_polys[b] = (int)per[b].scores.size();
    for (size_t i = 0; i < per[b].scores.size(); ++i) {
      if (lens && np < lens_cap) lens[np] = per[b].lens[i];
      if (scores && np < scores_cap) scores[np] = per[b].scores[i];
      ++np;
    }
    for (uint32_t v : per[b].xy) {
      if (xy && nv < xy_cap) xy[nv] = v;
      ++nv;
    }
  }
  if (total_polys) *total_polys = np;
  if (total_vertices) *total_vertices = nv / 2;
  return 0;
}
