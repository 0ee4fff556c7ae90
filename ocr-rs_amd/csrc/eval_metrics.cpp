// Detection quality metrics, host code: consumers of the polygon lists.
//   evaluate_image    /root/reference/src/text_detection/metrics.rs:255-380
//   combine_results   metrics.rs:229-253        (validate_measure / gather_measure: the mirrors)
//   get_intersection / get_union / IoU  metrics.rs:382-394 -> geo-clipper boolean ops (not vendored)
// Polygon intersection / union AREAS are computed from the arrangement of the two rings: every
// sub-segment whose two sides differ in "inside" contributes its shoelace term, oriented with the
// region on its left.  (Clipper rounds intersection vertices to integers at factor 1.0; the exact
// area differs from its area by far less than the 0.5 thresholds' margins in the reference KATs.)
#include <algorithm>
#include <cmath>
#include <vector>

#include "postproc_geom.hpp"

namespace ocr {
namespace geom {
namespace {

typedef long double ld;

static long long shoelace2i(const std::vector<Pt>& r) {
  long long s = 0;
  const size_t n = r.size();
  for (size_t i = 0; i < n; ++i) s += (long long)r[i].x * r[(i + 1) % n].y - (long long)r[(i + 1) % n].x * r[i].y;
  return s;
}

static int winding(const std::vector<Pt>& ring, ld qx, ld qy) {
  int wn = 0;
  const size_t n = ring.size();
  for (size_t i = 0; i < n; ++i) {
    const Pt &a = ring[i], &b = ring[(i + 1) % n];
    const ld cr = (ld)(b.x - a.x) * (qy - a.y) - (qx - a.x) * (ld)(b.y - a.y);
    if ((ld)a.y <= qy) {
      if ((ld)b.y > qy && cr > 0) ++wn;
    } else if ((ld)b.y <= qy && cr < 0) --wn;
  }
  return wn;
}

struct Seg {
  Pt a, b;
};

// area of {inside A} AND / OR {inside B}
static double boolean_area(std::vector<Pt> A, std::vector<Pt> B, bool is_union) {
  if (A.size() < 3 || B.size() < 3) {
    if (!is_union) return 0.0;
    return (A.size() >= 3 ? std::fabs((double)shoelace2i(A)) : 0.0) / 2.0 + (B.size() >= 3 ? std::fabs((double)shoelace2i(B)) : 0.0) / 2.0;
  }
  std::vector<Seg> segs;
  for (const auto* r : {&A, &B})
    for (size_t i = 0; i < r->size(); ++i)
      if (!((*r)[i] == (*r)[(i + 1) % r->size()])) segs.push_back({(*r)[i], (*r)[(i + 1) % r->size()]});
  const ld eps = 1e-7L;
  ld area2 = 0;
  for (size_t i = 0; i < segs.size(); ++i) {
    const Seg& s = segs[i];
    const long long d1x = s.b.x - s.a.x, d1y = s.b.y - s.a.y;
    std::vector<ld> ts = {0.0L, 1.0L};
    for (size_t j = 0; j < segs.size(); ++j) {
      if (j == i) continue;
      const Seg& o = segs[j];
      const long long d2x = o.b.x - o.a.x, d2y = o.b.y - o.a.y;
      const long long den = d1x * d2y - d1y * d2x;
      if (den == 0) continue;
      const long long wx = o.a.x - s.a.x, wy = o.a.y - s.a.y;
      const ld t = (ld)(wx * d2y - wy * d2x) / (ld)den, u = (ld)(wx * d1y - wy * d1x) / (ld)den;
      if (t > 0 && t < 1 && u >= 0 && u <= 1) ts.push_back(t);
    }
    std::sort(ts.begin(), ts.end());
    const ld len = std::sqrt((double)(d1x * d1x + d1y * d1y));
    const ld ux = d1x / len, uy = d1y / len;
    for (size_t k = 0; k + 1 < ts.size(); ++k) {
      if (ts[k + 1] - ts[k] < 1e-15L) continue;
      const ld x0 = s.a.x + ts[k] * d1x, y0 = s.a.y + ts[k] * d1y, x1 = s.a.x + ts[k + 1] * d1x, y1 = s.a.y + ts[k + 1] * d1y;
      const ld mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
      auto inside = [&](ld qx, ld qy) {
        const bool ia = winding(A, qx, qy) != 0, ib = winding(B, qx, qy) != 0;
        return is_union ? (ia || ib) : (ia && ib);
      };
      const bool left = inside(mx - eps * uy, my + eps * ux), right = inside(mx + eps * uy, my - eps * ux);
      if (left == right) continue;
      // a boundary piece shared by both rings (collinear, overlapping edges) appears once per ring:
      // count it only for the lowest-numbered segment that carries it
      bool duplicate = false;
      for (size_t j = 0; j < i && !duplicate; ++j) {
        const Seg& o = segs[j];
        const ld cr = (ld)(o.b.x - o.a.x) * (my - o.a.y) - (mx - o.a.x) * (ld)(o.b.y - o.a.y);
        if (fabsl(cr) > 1e-9L) continue;
        const ld dot = (mx - o.a.x) * (ld)(o.b.x - o.a.x) + (my - o.a.y) * (ld)(o.b.y - o.a.y);
        const ld l2 = (ld)(o.b.x - o.a.x) * (o.b.x - o.a.x) + (ld)(o.b.y - o.a.y) * (o.b.y - o.a.y);
        duplicate = dot > 0 && dot < l2;
      }
      if (duplicate) continue;
      const ld term = x0 * y1 - x1 * y0;
      area2 += left ? term : -term;
    }
  }
  return std::fabs((double)area2) / 2.0;
}

}  // namespace

double polygon_area(const std::vector<Pt>& p) { return p.size() < 3 ? 0.0 : std::fabs((double)shoelace2i(p)) / 2.0; }

double intersection_area(const std::vector<Pt>& a, const std::vector<Pt>& b) { return boolean_area(a, b, false); }

double union_area(const std::vector<Pt>& a, const std::vector<Pt>& b) {
  // inclusion-exclusion is exact for areas and immune to coincident edges
  return polygon_area(a) + polygon_area(b) - intersection_area(a, b);
}

// metrics.rs:255-380, including its quirk of testing `gt_dont_care.contains(det_num)` for detections
MetricsItem evaluate_image(const std::vector<std::vector<Pt>>& gt, const std::vector<bool>& ignore,
                           const std::vector<std::vector<Pt>>& pred) {
  const double area_precision_constraint = 0.5, iou_constraint = 0.5;
  std::vector<size_t> gt_dont_care, det_dont_care;
  for (size_t n = 0; n < gt.size(); ++n)
    if (ignore[n]) gt_dont_care.push_back(n);
  for (size_t d = 0; d < pred.size(); ++d) {
    for (size_t g : gt_dont_care) {
      const double inter = intersection_area(gt[g], pred[d]);
      const double pd = polygon_area(pred[d]);
      const double precision = pd == 0.0 ? 0.0 : inter / pd;
      if (precision > area_precision_constraint) {
        det_dont_care.push_back(d);
        break;
      }
    }
  }
  int det_matched = 0;
  if (!gt.empty() && !pred.empty()) {
    std::vector<int> gt_used(gt.size(), 0), det_used(pred.size(), 0);
    auto in_gt_dc = [&](size_t v) { return std::find(gt_dont_care.begin(), gt_dont_care.end(), v) != gt_dont_care.end(); };
    for (size_t g = 0; g < gt.size(); ++g)
      for (size_t d = 0; d < pred.size(); ++d) {
        const double iou = intersection_area(pred[d], gt[g]) / union_area(pred[d], gt[g]);
        if (gt_used[g] == 0 && det_used[d] == 0 && !in_gt_dc(g) && !in_gt_dc(d) && iou > iou_constraint) {
          gt_used[g] = det_used[d] = 1;
          ++det_matched;
        }
      }
  }
  MetricsItem m{};
  m.gt_care = (int)(gt.size() - gt_dont_care.size());
  m.det_care = (int)(pred.size() - det_dont_care.size());
  m.det_matched = det_matched;
  if (m.gt_care == 0) {
    m.recall = 1.0;
    m.precision = m.det_care > 0 ? 0.0 : 1.0;
  } else {
    m.recall = (double)det_matched / m.gt_care;
    m.precision = m.det_care == 0 ? 0.0 : (double)det_matched / m.det_care;
  }
  m.hmean = m.precision + m.recall == 0.0 ? 0.0 : 2.0 * m.precision * m.recall / (m.precision + m.recall);
  return m;
}

// metrics.rs:229-253 -> (precision, recall, hmean)
void combine_results(const MetricsItem* r, int n, double* precision, double* recall, double* hmean) {
  long long gt = 0, det = 0, matched = 0;
  for (int i = 0; i < n; ++i) {
    gt += r[i].gt_care;
    det += r[i].det_care;
    matched += r[i].det_matched;
  }
  *recall = gt != 0 ? (double)matched / gt : 0.0;
  *precision = det != 0 ? (double)matched / det : 0.0;
  *hmean = *recall + *precision != 0.0 ? 2.0 * (*recall * *precision) / (*recall + *precision) : 0.0;
}

}  // namespace geom
}  // namespace ocr
