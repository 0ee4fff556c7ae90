// Host-side geometry of the detection post-processing (irregular, pointer-chasing
// work that stays on CPU threads; the dense parts - binarize and the masked box
// score - are HIP kernels, see box_score.hip / stem_tail.hip).
//
//   get_polygons_from_bitmap   /root/reference/src/text_detection/metrics.rs:58-127
//   get_min_area_bounding_box  metrics.rs:133-148
//   expand_polygon             /root/reference/src/polygon.rs:13-56
//
// The reference delegates to crates that are not vendored in /root/reference:
// imageproc 0.22.0 (find_contours, arc_length, approximate_polygon_dp,
// min_area_rect), geo 0.15.0 (area, length) and Clipper 6.x through geo-clipper
// (offset).  Their published algorithms are implemented here from their
// descriptions; the reference's known-answer tests
// (metrics.rs:406-646) are reproduced by tests/test_capi_cpu.py.
#include "postproc_geom.hpp"
#include "hypot_glibc.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <numeric>
#include <tuple>

namespace ocr {
namespace geom {

// ---------------------------------------------------------------------------
// Suzuki-Abe border following (imageproc::contours::find_contours, threshold 0).
// Outer and hole borders are both reported, in raster discovery order.
// ---------------------------------------------------------------------------
static const int kDx[8] = {-1, -1, 0, 1, 1, 1, 0, -1};  // W NW N NE E SE S SW (clockwise, y down)
static const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

static inline int dir_index(int dx, int dy) {
  for (int k = 0; k < 8; ++k)
    if (kDx[k] == dx && kDy[k] == dy) return k;
  return -1;
}

// bits s, s + 1, s + 2 of the bit image (the word behind is touched only when the window reaches into it)
static inline unsigned get3(const uint32_t* bits, size_t s) {
  const size_t q = s >> 5;
  const unsigned sh = (unsigned)(s & 31);
  unsigned v = bits[q] >> sh;
  if (sh > 29) v |= bits[q + 1] << (32 - sh);
  return v & 7u;
}
// one step of the border following as a table: [direction of the previous pixel][neighbour byte, bit d = neighbour kDx/kDy[d] is
// foreground] -> direction of the next border pixel (the first set neighbour counter-clockwise from just after the previous
// one) | 8 if the E neighbour was examined and found zero before it
struct StepTable {
  uint8_t t[8][256];
  StepTable() {
    for (int base = 0; base < 8; ++base)
      for (int m = 0; m < 256; ++m) {
        int found = 0, right_edge = 0;
        for (int k = 1; k <= 8; ++k) {
          const int d = (base - k) & 7;
          if (m >> d & 1) {
            found = d;
            break;
          }
          if (d == 4) right_edge = 8;
        }
        t[base][m] = (uint8_t)(found | right_edge);
      }
  }
};
static const StepTable kStep;

// The working grid of the algorithm (0 background, 1 foreground, +-id once a border has passed) is kept as the
// immutable foreground BIT image plus two label bit planes that only border pixels ever touch: value(x, y) = fg ? (seen ?
// (neg ? -id : id) : 1) : 0 - the id itself is never read, only its sign and whether it is there.  A pixel can only start a border at the first or last pixel of a horizontal foreground run (its
// left resp. right neighbour must be zero), so the raster scan walks run boundaries found with word-wide bit tricks
// instead of testing every pixel - the cost is O(runs + border pixels), not O(H W).
// bits: row-major, bit (x & 31) of word (y * w + x) >> 5 (w is a multiple of 32 on the product path; any w works).
void find_contours_bits(const uint32_t* bits, int h, int w, std::vector<std::vector<Pt>>& out) {
  out.clear();
  const size_t npx = (size_t)h * w;
  // The algorithm only ever asks two things of a border pixel's label: has a border passed here (label != 0), and did one leave it
  // negative (label < 0: it was a right edge).  Two bit planes beside the foreground bits - 2 x H W / 8 bytes per thread, cache
  // resident, cleared per image - instead of an int32 per pixel that only the scattered border pixels touch.
  struct Labels {
    std::vector<uint32_t> seen, neg;
    std::vector<Pt> pts;
  };
  thread_local Labels tl;
  const size_t nwords = (npx + 31) / 32;
  if (tl.seen.size() < nwords) {
    tl.seen.resize(nwords);
    tl.neg.resize(nwords);
  }
  uint32_t* seen = tl.seen.data();
  uint32_t* neg = tl.neg.data();
  std::memset(seen, 0, nwords * sizeof(uint32_t));
  std::memset(neg, 0, nwords * sizeof(uint32_t));
  auto fg = [&](size_t i) { return (bits[i >> 5] >> (i & 31)) & 1u; };
  auto nz = [&](int x, int y) { return x >= 0 && x < w && y >= 0 && y < h && fg((size_t)y * w + x); };
  auto trace = [&](int x, int y, int adjx) {
    std::vector<Pt>& pts = tl.pts;   // grown once per thread; the contour leaves as an exactly sized copy
    pts.clear();
    const int start = dir_index(adjx - x, 0);
    int p1x = 0, p1y = 0;
    bool found = false;
    for (int k = 0; k < 8 && !found; ++k) {  // clockwise from the adjacent zero pixel
      const int d = (start + k) & 7;
      if (nz(x + kDx[d], y + kDy[d])) {
        p1x = x + kDx[d];
        p1y = y + kDy[d];
        found = true;
      }
    }
    if (!found) {
      pts.push_back({x, y});
      const size_t i0 = (size_t)y * w + x;
      seen[i0 >> 5] |= 1u << (i0 & 31);
      neg[i0 >> 5] |= 1u << (i0 & 31);
    } else {
      int p3x = x, p3y = y;
      int base = dir_index(p1x - x, p1y - y);   // direction of the previous pixel as seen from the current one
      for (;;) {
        pts.push_back({p3x, p3y});
        int p4x = 0, p4y = 0;
        bool right_edge = false;
        int dn = 0;
        if (p3x > 0 && p3y > 0 && p3x + 1 < w && p3y + 1 < h) {
          // interior pixel: the eight neighbours as one byte (three 3-bit windows of the bit image), the search as a table look-up
          const size_t i3 = (size_t)p3y * w + p3x;
          const unsigned top = get3(bits, i3 - w - 1), mid = get3(bits, i3 - 1), bot = get3(bits, i3 + w - 1);
          const unsigned m = (mid & 1u) | (top & 1u) << 1 | (top & 2u) << 1 | (top & 4u) << 1 | (mid & 4u) << 2 | (bot & 4u) << 3 | (bot & 2u) << 5 |
                             (bot & 1u) << 7;
          const unsigned st = kStep.t[base][m];   // m != 0: the pixel we came from is a neighbour
          dn = st & 7;
          right_edge = (st & 8) != 0;
        } else {
          for (int k = 1; k <= 8; ++k) {  // counter-clockwise, starting just after the previous pixel's direction
            const int d = (base - k) & 7;
            if (nz(p3x + kDx[d], p3y + kDy[d])) {
              dn = d;
              break;
            }
            if (d == 4) right_edge = true;  // the E neighbour was examined and is zero
          }
        }
        p4x = p3x + kDx[dn];
        p4y = p3y + kDy[dn];
        const size_t ic = (size_t)p3y * w + p3x;
        const uint32_t bit = 1u << (ic & 31);
        seen[ic >> 5] |= bit;                                 // label = border id (or -id below): only "non-zero" is ever read
        if (p3x + 1 == w || right_edge) neg[ic >> 5] |= bit;  // label = -id, whatever it was
        if (p4x == x && p4y == y && p3x == p1x && p3y == p1y) break;
        p3x = p4x;
        p3y = p4y;
        base = (dn + 4) & 7;
      }
    }
    out.emplace_back(pts.begin(), pts.end());
  };
  // one pixel of the raster scan: outer border start if value == 1 and the W neighbour is 0 (x > 0), else hole border
  // start if value > 0 and the E neighbour is 0 (x + 1 < w)
  auto visit = [&](int x, int y) {
    const size_t i = (size_t)y * w + x;
    const uint32_t bit = 1u << (i & 31);
    if (!(seen[i >> 5] & bit) && x > 0 && !fg(i - 1)) trace(x, y, x - 1);            // value == 1
    else if (!(neg[i >> 5] & bit) && x + 1 < w && !fg(i + 1)) trace(x, y, x + 1);    // value > 0
  };
  for (int y = 0; y < h; ++y) {
    const size_t r0 = (size_t)y * w;
    int x = 0;
    while (x < w) {
      // next foreground pixel at or after x
      size_t i = r0 + x;
      const size_t rend = r0 + w;
      unsigned cur = bits[i >> 5] >> (i & 31);
      while (!cur) {
        i = ((i >> 5) + 1) << 5;
        if (i >= rend) break;
        cur = bits[i >> 5];
      }
      if (i >= rend) break;
      i += __builtin_ctz(cur);
      if (i >= rend) break;
      const int x0 = (int)(i - r0);
      // end of this run: last foreground pixel before the next zero (or the row end)
      size_t j = i;
      unsigned inv = ~bits[j >> 5] >> (j & 31);   // set bits = zero pixels from j on (the shift fills with "no zero here")
      while (!inv) {
        j = ((j >> 5) + 1) << 5;
        if (j >= rend) break;
        inv = ~bits[j >> 5];
      }
      size_t z = j >= rend ? rend : j + __builtin_ctz(inv);  // first zero pixel after the run
      if (z > rend) z = rend;
      const int x1 = (int)(z - r0) - 1;
      visit(x0, y);
      if (x1 != x0) visit(x1, y);
      x = x1 + 1;
    }
  }
}

void find_contours(const uint8_t* bitmap, int h, int w, std::vector<std::vector<Pt>>& out) {
  const size_t npx = (size_t)h * w;
  std::vector<uint32_t> bits((npx + 31) / 32 + 1, 0u);
  for (size_t i = 0; i < npx; ++i)
    if (bitmap[i]) bits[i >> 5] |= 1u << (i & 31);
  find_contours_bits(bits.data(), h, w, out);
}

// ---------------------------------------------------------------------------
// arc_length / approximate_polygon_dp (imageproc::geometry)
// ---------------------------------------------------------------------------
static inline double dist(const Pt& a, const Pt& b) {
  const double dx = (double)a.x - (double)b.x, dy = (double)a.y - (double)b.y;
  return std::sqrt(dx * dx + dy * dy);
}

double arc_length(const std::vector<Pt>& p, bool closed) {
  double len = 0.0;
  for (size_t i = 0; i + 1 < p.size(); ++i) len += dist(p[i], p[i + 1]);
  if (p.size() > 2 && closed) len += dist(p[0], p[p.size() - 1]);
  return len;
}

// Douglas-Peucker with the distance to the infinite line through the two end
// points; recursion replaced by an explicit stack of [first,last] ranges.  The
// recursive form concatenates left part (minus its last point) and right part, so
// the result is exactly "all kept indices in increasing order".
void approximate_polygon_dp(const std::vector<Pt>& curve, double eps, bool closed, std::vector<Pt>& out) {
  out.clear();
  const int n = (int)curve.size();
  if (n == 0) return;
  std::vector<char> keep(n, 0);
  keep[0] = keep[n - 1] = 1;
  std::vector<std::pair<int, int>> st;
  st.push_back({0, n - 1});
  while (!st.empty()) {
    const auto [lo, hi] = st.back();
    st.pop_back();
    const double x0 = curve[lo].x, y0 = curve[lo].y, x1 = curve[hi].x, y1 = curve[hi].y;
    const double a = y0 - y1, b = x1 - x0, c = x0 * y1 - x1 * y0;
    const double norm = std::sqrt(a * a + b * b);
    double dmax = 0.0;
    int idx = lo;
    for (int i = lo + 1; i <= hi; ++i) {
      const double d = std::fabs(a * (double)curve[i].x + b * (double)curve[i].y + c) / norm;  // NaN when lo==hi point
      if (d > dmax) {
        dmax = d;
        idx = i;
      }
    }
    if (dmax > eps) {
      keep[idx] = 1;
      st.push_back({lo, idx});
      st.push_back({idx, hi});
    }
  }
  for (int i = 0; i < n; ++i)
    if (keep[i]) out.push_back(curve[i]);
  if (n == 1) out.push_back(curve[0]);  // [first, last] of a one-point curve
  if (closed) out.pop_back();
}

// ---------------------------------------------------------------------------
// min_area_rect (imageproc::geometry) + the reference's ordering / short side
// ---------------------------------------------------------------------------
static inline int orient(const Pt& p, const Pt& q, const Pt& r) {
  const long long v = (long long)(q.y - p.y) * (r.x - q.x) - (long long)(q.x - p.x) * (r.y - q.y);
  return v == 0 ? 0 : (v > 0 ? 1 : 2);  // 1 clockwise, 2 counter-clockwise
}

static void convex_hull(const std::vector<Pt>& in, std::vector<Pt>& hull) {
  hull.clear();
  if (in.empty()) return;
  std::vector<Pt> pts = in;
  size_t sp = 0;
  for (size_t i = 1; i < pts.size(); ++i)
    if (pts[i].y < pts[sp].y || (pts[i].y == pts[sp].y && pts[i].x < pts[sp].x)) sp = i;
  const Pt start = pts[sp];
  std::swap(pts[0], pts[sp]);
  pts.erase(pts.begin());
  std::stable_sort(pts.begin(), pts.end(), [&](const Pt& a, const Pt& b) {
    const int o = orient(start, a, b);
    if (o == 0) return dist(start, a) < dist(start, b);
    return o == 2;
  });
  std::vector<Pt> rem;
  for (size_t i = 0; i < pts.size(); ++i) {
    Pt p = pts[i];
    while (i + 1 < pts.size() && orient(start, p, pts[i + 1]) == 0) p = pts[++i];
    rem.push_back(p);
  }
  hull.push_back(start);
  for (const Pt& p : rem) {
    while (hull.size() > 1 && orient(hull[hull.size() - 2], hull[hull.size() - 1], p) != 2) hull.pop_back();
    hull.push_back(p);
  }
}

static void min_area_rect(const std::vector<Pt>& pts, Pt box[4]) {
  std::vector<Pt> hull;
  convex_hull(pts, hull);
  if (hull.size() == 1) {
    box[0] = box[1] = box[2] = box[3] = hull[0];
    return;
  }
  if (hull.size() == 2) {
    box[0] = hull[0];
    box[1] = hull[1];
    box[2] = hull[1];
    box[3] = hull[0];
    return;
  }
  const double kPi = 3.14159265358979323846;
  std::vector<double> angles;
  for (size_t i = 0; i + 1 < hull.size(); ++i) {
    const double ex = (double)hull[i + 1].x - (double)hull[i].x, ey = (double)hull[i + 1].y - (double)hull[i].y;
    const double ang = std::fabs(std::fmod(std::atan2(ey, ex) + kPi, kPi / 2.0));
    if (angles.empty() || angles.back() != ang) angles.push_back(ang);
  }
  double min_area = INFINITY;
  double rx[4] = {0, 0, 0, 0}, ry[4] = {0, 0, 0, 0};
  for (double ang : angles) {
    const double s = std::sin(ang), c = std::cos(ang);
    double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (const Pt& p : hull) {  // rotate by -angle
      const double x = p.x * c + p.y * s, y = p.y * c - p.x * s;
      mnx = std::min(mnx, x);
      mxx = std::max(mxx, x);
      mny = std::min(mny, y);
      mxy = std::max(mxy, y);
    }
    const double area = (mxx - mnx) * (mxy - mny);
    if (area < min_area) {
      min_area = area;
      const double cx[4] = {mxx, mnx, mnx, mxx}, cy[4] = {mny, mny, mxy, mxy};
      for (int k = 0; k < 4; ++k) {  // rotate back by +angle
        rx[k] = cx[k] * c - cy[k] * s;
        ry[k] = cy[k] * c + cx[k] * s;
      }
    }
  }
  int order[4] = {0, 1, 2, 3};
  std::stable_sort(order, order + 4, [&](int a, int b) { return rx[a] < rx[b]; });
  const double sx[4] = {rx[order[0]], rx[order[1]], rx[order[2]], rx[order[3]]};
  const double sy[4] = {ry[order[0]], ry[order[1]], ry[order[2]], ry[order[3]]};
  const int i1 = sy[1] > sy[0] ? 0 : 1, i2 = sy[3] > sy[2] ? 2 : 3, i3 = sy[3] > sy[2] ? 3 : 2, i4 = sy[1] > sy[0] ? 1 : 0;
  box[0] = {(int)std::floor(sx[i1]), (int)std::floor(sy[i1])};
  box[1] = {(int)std::ceil(sx[i2]), (int)std::floor(sy[i2])};
  box[2] = {(int)std::ceil(sx[i3]), (int)std::ceil(sy[i3])};
  box[3] = {(int)std::floor(sx[i4]), (int)std::ceil(sy[i4])};
}

double min_area_bounding_box(const std::vector<Pt>& pts, Pt res[4]) {
  Pt b[4];
  min_area_rect(pts, b);
  std::stable_sort(b, b + 4, [](const Pt& p, const Pt& q) { return p.x < q.x; });  // metrics.rs:138
  const int i1 = b[1].y > b[0].y ? 0 : 1, i2 = b[3].y > b[2].y ? 2 : 3, i3 = b[3].y > b[2].y ? 3 : 2, i4 = b[1].y > b[0].y ? 1 : 0;
  res[0] = b[i1];
  res[1] = b[i2];
  res[2] = b[i3];
  res[3] = b[i4];
  // geo's euclidean_length = libm's hypot (metrics.rs:145-146): the restatement pinned to the reference's environment
  // (hypot_glibc.hpp, glibc 2.35 x86-64), the one definition host and device code share - not whatever libm is loaded here
  const double wl = hypot_glibc((double)res[0].x - res[1].x, (double)res[0].y - res[1].y);
  const double hl = hypot_glibc((double)res[0].x - res[3].x, (double)res[0].y - res[3].y);
  return std::min(wl, hl);
}

// ---------------------------------------------------------------------------
// Clipper offset (JoinType::Miter(2.0), EndType::ClosedPolygon, scale 1.0)
// ---------------------------------------------------------------------------
static inline long long cround(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

static long long shoelace2(const std::vector<Pt>& r) {
  long long s = 0;
  const size_t n = r.size();
  for (size_t i = 0; i < n; ++i) {
    const Pt &a = r[i], &b = r[(i + 1) % n];
    s += (long long)a.x * b.y - (long long)b.x * a.y;
  }
  return s;
}

double offset_distance(const std::vector<Pt>& poly, double factor) {  // polygon.rs:27
  const size_t n = poly.size();
  const double area = std::fabs((double)shoelace2(poly)) / 2.0;
  double per = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const Pt &a = poly[i], &b = poly[(i + 1) % n];
    per += hypot_glibc((double)(b.x - a.x), (double)(b.y - a.y));   // the same definition as unclip.hip: bit for bit on any host libm
  }
  return area * factor / per;
}

void raw_offset_ring(const std::vector<Pt>& poly, double delta, std::vector<Pt>& out) {
  out.clear();
  std::vector<Pt> pts = poly, src;
  while (pts.size() > 1 && pts.front() == pts.back()) pts.pop_back();
  for (const Pt& p : pts)
    if (src.empty() || !(src.back() == p)) src.push_back(p);
  const int n = (int)src.size();
  if (n < 3) return;
  if (shoelace2(src) < 0) std::reverse(src.begin(), src.end());  // FixOrientations
  const double miter_lim = 0.5;  // 2 / (MiterLimit^2), MiterLimit = 2.0
  std::vector<double> nx(n), ny(n);
  for (int i = 0; i < n; ++i) {
    const Pt &a = src[i], &b = src[(i + 1) % n];
    double dx = (double)(b.x - a.x), dy = (double)(b.y - a.y);
    const double f = 1.0 / std::sqrt(dx * dx + dy * dy);
    dx *= f;
    dy *= f;
    nx[i] = dy;
    ny[i] = -dx;
  }
  int k = n - 1;
  for (int j = 0; j < n; ++j) {
    const double sx = src[j].x, sy = src[j].y;
    double sin_a = nx[k] * ny[j] - nx[j] * ny[k];
    bool done = false;
    if (std::fabs(sin_a * delta) < 1.0) {
      const double cos_a = nx[k] * nx[j] + ny[j] * ny[k];
      if (cos_a > 0) {
        out.push_back({(int)cround(sx + nx[k] * delta), (int)cround(sy + ny[k] * delta)});
        done = true;
      }
    } else if (sin_a > 1.0) sin_a = 1.0;
    else if (sin_a < -1.0) sin_a = -1.0;
    if (!done) {
      if (sin_a * delta < 0) {
        out.push_back({(int)cround(sx + nx[k] * delta), (int)cround(sy + ny[k] * delta)});
        out.push_back(src[j]);
        out.push_back({(int)cround(sx + nx[j] * delta), (int)cround(sy + ny[j] * delta)});
      } else {
        const double r = 1.0 + (nx[j] * nx[k] + ny[j] * ny[k]);
        if (r >= miter_lim) {
          const double q = delta / r;
          out.push_back({(int)cround(sx + (nx[k] + nx[j]) * q), (int)cround(sy + (ny[k] + ny[j]) * q)});
        } else {  // squared-off corner
          const double dxx = std::tan(std::atan2(sin_a, nx[k] * nx[j] + ny[k] * ny[j]) / 4.0);
          out.push_back({(int)cround(sx + delta * (nx[k] - ny[k] * dxx)), (int)cround(sy + delta * (ny[k] + nx[k] * dxx))});
          out.push_back({(int)cround(sx + delta * (nx[j] + ny[j] * dxx)), (int)cround(sy + delta * (ny[j] - nx[j] * dxx))});
        }
      }
    }
    k = j;
  }
}

// --- union with positive fill of one closed ring: outer boundary of {winding > 0}
namespace {
typedef __int128 i128;

static i128 gcd128(i128 a, i128 b) {
  if (a < 0) a = -a;
  if (b < 0) b = -b;
  while (b != 0) {
    const i128 t = a % b;
    a = b;
    b = t;
  }
  return a;
}

struct NodeKey {  // exact rational point (xn/den, yn/den), reduced, den > 0
  i128 xn, yn, den;
  bool operator<(const NodeKey& o) const { return std::tie(xn, yn, den) < std::tie(o.xn, o.yn, o.den); }
};

static NodeKey make_key(i128 xn, i128 yn, i128 den) {
  if (den < 0) {
    xn = -xn;
    yn = -yn;
    den = -den;
  }
  if (den == 1) return {xn, yn, 1};   // a ring vertex: reduced as it stands (the 128-bit gcd loop was most of the general path's time)
  constexpr i128 kLim = (i128)1 << 62;
  if (xn > -kLim && xn < kLim && yn > -kLim && yn < kLim && den < kLim) {   // the usual case: 64-bit arithmetic, the same reduced key
    const long long a = (long long)xn, b = (long long)yn, c = (long long)den;
    long long g = std::gcd(std::gcd(a < 0 ? -a : a, b < 0 ? -b : b), c);
    if (g == 0) g = 1;
    return {(i128)(a / g), (i128)(b / g), (i128)(c / g)};
  }
  i128 g = gcd128(gcd128(xn, yn), den);
  if (g == 0) g = 1;
  return {xn / g, yn / g, den / g};
}

struct Split {
  i128 tn, td;  // parameter tn/td on the segment, td > 0
  int node;
};

static int winding(const std::vector<Pt>& ring, long double qx, long double qy) {
  int wn = 0;
  const size_t n = ring.size();
  for (size_t i = 0; i < n; ++i) {
    const Pt &a = ring[i], &b = ring[i + 1 == n ? 0 : i + 1];
    const bool a_below = (long double)a.y <= qy, b_below = (long double)b.y <= qy;
    if (a_below == b_below) continue;   // only an edge that straddles the ray's height can count (the same two tests as below)
    const long double cr = (long double)(b.x - a.x) * (qy - a.y) - (qx - a.x) * (long double)(b.y - a.y);
    if (a_below) {
      if (cr > 0) ++wn;
    } else if (cr < 0) --wn;
  }
  return wn;
}
}  // namespace

void positive_union_outer(const std::vector<Pt>& ring_in, std::vector<Pt>& out) {
  out.clear();
  std::vector<Pt> ring;
  for (const Pt& p : ring_in)
    if (ring.empty() || !(ring.back() == p)) ring.push_back(p);
  while (ring.size() > 1 && ring.front() == ring.back()) ring.pop_back();
  const int n = (int)ring.size();
  if (n < 3) return;

  std::vector<Pt> pts;
  // Fast path for what an offset word box almost always is - a SIMPLE ring: no two non-adjacent edges meet (not even in a
  // point), no two adjacent edges are collinear.  Its arrangement is the ring itself: the region left of a positively
  // oriented ring has winding + 1 (kept whole), a negatively oriented one bounds nothing of positive winding (empty result).
  // Exact integer tests, O(n^2) for n around 4 .. 10; anything else takes the general arrangement below.
  bool simple = true;
  for (int i = 0; i < n && simple; ++i) {
    const Pt &a = ring[i], &b = ring[(i + 1) % n];
    const long long d1x = b.x - a.x, d1y = b.y - a.y;
    for (int j = i + 1; j < n && simple; ++j) {
      const Pt &c = ring[j], &d = ring[(j + 1) % n];
      const long long d2x = d.x - c.x, d2y = d.y - c.y;
      long long den = d1x * d2y - d1y * d2x;
      const bool adjacent = j == i + 1 || (i == 0 && j == n - 1);
      if (adjacent) {
        if (den == 0 || n == 3) simple = den != 0;   // collinear neighbours (a spike or a straight vertex): general path
        continue;
      }
      const long long wx = c.x - a.x, wy = c.y - a.y;
      if (den == 0) {
        if (wx * d1y - wy * d1x == 0) simple = false;   // on one line: they may overlap or touch
        continue;
      }
      long long tn = wx * d2y - wy * d2x, un = wx * d1y - wy * d1x;
      if (den < 0) {
        den = -den;
        tn = -tn;
        un = -un;
      }
      if (tn >= 0 && tn <= den && un >= 0 && un <= den) simple = false;
    }
  }
  if (simple) {
    if (shoelace2(ring) <= 0) return;
    pts = ring;
  } else {
  // a few dozen nodes at most: a flat list searched front to back (ids in order of first appearance, as a map would give them)
  // (the containers of this path live with the calling thread: a pool thread runs it for polygon after polygon)
  struct Scratch {
    std::vector<NodeKey> node_key;
    std::vector<long double> node_x, node_y;
    std::vector<std::vector<Split>> splits;
  };
  static thread_local Scratch sc;
  std::vector<NodeKey>& node_key = sc.node_key;
  std::vector<long double>&node_x = sc.node_x, &node_y = sc.node_y;
  node_key.clear();
  node_x.clear();
  node_y.clear();
  auto get_node = [&](i128 xn, i128 yn, i128 den) {
    const NodeKey k = make_key(xn, yn, den);
    for (size_t id = 0; id < node_key.size(); ++id)
      if (node_key[id].xn == k.xn && node_key[id].yn == k.yn && node_key[id].den == k.den) return (int)id;
    const int id = (int)node_x.size();
    node_key.push_back(k);
    node_x.push_back((long double)k.xn / (long double)k.den);
    node_y.push_back((long double)k.yn / (long double)k.den);
    return id;
  };
  std::vector<std::vector<Split>>& splits = sc.splits;
  if ((int)splits.size() < n) splits.resize(n);
  for (int i = 0; i < n; ++i) splits[i].clear();
  for (int i = 0; i < n; ++i) {
    const Pt &a = ring[i], &b = ring[(i + 1) % n];
    splits[i].push_back({0, 1, get_node(a.x, a.y, 1)});
    splits[i].push_back({1, 1, get_node(b.x, b.y, 1)});
  }
  for (int i = 0; i < n; ++i) {
    const Pt &a = ring[i], &b = ring[(i + 1) % n];
    const long long d1x = b.x - a.x, d1y = b.y - a.y;
    for (int j = i + 1; j < n; ++j) {
      const Pt &c = ring[j], &d = ring[(j + 1) % n];
      const long long d2x = d.x - c.x, d2y = d.y - c.y;
      long long den = d1x * d2y - d1y * d2x;
      if (den == 0) continue;  // parallel / collinear: no proper crossing
      const long long wx = c.x - a.x, wy = c.y - a.y;
      long long tn = wx * d2y - wy * d2x, un = wx * d1y - wy * d1x;
      if (den < 0) {
        den = -den;
        tn = -tn;
        un = -un;
      }
      if (tn < 0 || tn > den || un < 0 || un > den) continue;
      const int node = get_node((i128)a.x * den + (i128)tn * d1x, (i128)a.y * den + (i128)tn * d1y, den);
      if (tn > 0 && tn < den) splits[i].push_back({tn, den, node});
      if (un > 0 && un < den) splits[j].push_back({un, den, node});
    }
  }
  struct Edge {
    int a, b;
    bool used;
  };
  std::vector<Edge> edges;
  const long double eps = 1e-7L;
  for (int i = 0; i < n; ++i) {
    auto& sp = splits[i];
    std::sort(sp.begin(), sp.end(), [](const Split& p, const Split& q) { return p.tn * q.td < q.tn * p.td; });
    const Pt &a = ring[i], &b = ring[(i + 1) % n];
    const long double dx = b.x - a.x, dy = b.y - a.y;
    const long double len = std::sqrt((double)(dx * dx + dy * dy));
    const long double ux = dx / len, uy = dy / len;
    for (size_t k = 0; k + 1 < sp.size(); ++k) {
      const int n0 = sp[k].node, n1 = sp[k + 1].node;
      if (n0 == n1) continue;
      const long double mx = (node_x[n0] + node_x[n1]) / 2, my = (node_y[n0] + node_y[n1]) / 2;
      const int w_right = winding(ring, mx + eps * uy, my - eps * ux);  // right of the directed edge = (dy,-dx)
      const int w_left = winding(ring, mx - eps * uy, my + eps * ux);
      if ((w_left > 0) != (w_right > 0)) {
        if (w_left > 0) edges.push_back({n0, n1, false});
        else edges.push_back({n1, n0, false});
      }
    }
  }
  if (edges.empty()) return;
  std::vector<std::vector<int>> loops;
  for (size_t e0 = 0; e0 < edges.size(); ++e0) {
    if (edges[e0].used) continue;
    std::vector<int> loop;
    int cur = (int)e0;
    while (!edges[cur].used) {
      edges[cur].used = true;
      loop.push_back(edges[cur].a);
      const int at = edges[cur].b;
      int best = -1;
      double best_turn = 0;
      const double inx = (double)(node_x[edges[cur].b] - node_x[edges[cur].a]);
      const double iny = (double)(node_y[edges[cur].b] - node_y[edges[cur].a]);
      for (int c = 0; c < (int)edges.size(); ++c) {   // the edges leaving this node, in the order they were made
        if (edges[c].a != at || edges[c].used) continue;
        const double ox = (double)(node_x[edges[c].b] - node_x[edges[c].a]);
        const double oy = (double)(node_y[edges[c].b] - node_y[edges[c].a]);
        const double turn = std::atan2(inx * oy - iny * ox, inx * ox + iny * oy);
        if (best < 0 || turn < best_turn) {  // touching node: most clockwise turn keeps the outer loop whole
          best = c;
          best_turn = turn;
        }
      }
      if (best < 0) break;
      cur = best;
    }
    loops.push_back(std::move(loop));
  }
  // Clipper's first OutRec starts at the lowest local minimum (largest y)
  size_t pick = 0;
  long double best_y = -INFINITY, best_x = INFINITY;
  for (size_t l = 0; l < loops.size(); ++l) {
    long double my = -INFINITY, mx = INFINITY;
    for (int nd : loops[l]) {
      my = std::max(my, node_y[nd]);
      mx = std::min(mx, node_x[nd]);
    }
    if (my > best_y || (my == best_y && mx < best_x)) {
      best_y = my;
      best_x = mx;
      pick = l;
    }
  }
  for (int nd : loops[pick]) pts.push_back({(int)cround((double)node_x[nd]), (int)cround((double)node_y[nd])});
  }
  // FixupOutPolygon: drop duplicate and collinear vertices
  bool changed = true;
  while (changed && pts.size() >= 3) {
    changed = false;
    const size_t m = pts.size();
    for (size_t i = 0; i < m; ++i) {
      const Pt &p = pts[(i + m - 1) % m], &c = pts[i], &nx = pts[(i + 1) % m];
      if (c == nx || c == p || (long long)(c.y - p.y) * (nx.x - c.x) == (long long)(c.x - p.x) * (nx.y - c.y)) {
        pts.erase(pts.begin() + i);
        changed = true;
        break;
      }
    }
  }
  if (pts.size() < 3) return;
  if (shoelace2(pts) < 0) std::reverse(pts.begin(), pts.end());
  // BuildResult order: ends at the top-most vertex (ties: right-most)
  size_t top = 0;
  for (size_t i = 1; i < pts.size(); ++i)
    if (pts[i].y < pts[top].y || (pts[i].y == pts[top].y && pts[i].x > pts[top].x)) top = i;
  for (size_t i = 0; i < pts.size(); ++i) out.push_back(pts[(top + 1 + i) % pts.size()]);
}

bool expand_polygon(const std::vector<Pt>& pts, double factor, std::vector<Pt>& out) {
  const double d = offset_distance(pts, factor);
  std::vector<Pt> raw;
  raw_offset_ring(pts, d, raw);
  positive_union_outer(raw, out);
  return !out.empty();
}

// the device code's hypot (hypot_glibc.hpp) against this machine's libm on every integer pair up to `limit`: mismatches
long long hypot_port_mismatches(int limit) {
  long long bad = 0;
  for (int a = 0; a <= limit; ++a)
    for (int b = 0; b <= a; ++b) {
      if (std::hypot((double)a, (double)b) != hypot_glibc((double)a, (double)b)) ++bad;
      if (std::hypot((double)-b, (double)a) != hypot_glibc((double)-b, (double)a)) ++bad;
    }
  return bad;
}

// ---------------------------------------------------------------------------
// per-image driver, split around the GPU box-score step
// ---------------------------------------------------------------------------
static void candidates_of(const std::vector<std::vector<Pt>>& contours, std::vector<std::vector<Pt>>& cands);

void contour_candidates(const uint8_t* bitmap, int h, int w, std::vector<std::vector<Pt>>& cands) {
  std::vector<std::vector<Pt>> contours;
  find_contours(bitmap, h, w, contours);
  candidates_of(contours, cands);
}

void contour_candidates_bits(const uint32_t* bits, int h, int w, std::vector<std::vector<Pt>>& cands) {
  std::vector<std::vector<Pt>> contours;
  find_contours_bits(bits, h, w, contours);
  candidates_of(contours, cands);
}

void contour_candidates_packed(const uint32_t* pts, const int32_t* lens, int n_contours, std::vector<std::vector<Pt>>& cands) {
  std::vector<std::vector<Pt>> contours((size_t)std::max(n_contours, 0));
  size_t at = 0;
  for (int k = 0; k < n_contours; ++k) {
    contours[k].resize((size_t)lens[k]);
    for (int i = 0; i < lens[k]; ++i, ++at) contours[k][i] = {(int)(pts[at] & 0xffffu), (int)(pts[at] >> 16)};
  }
  candidates_of(contours, cands);
}

static void candidates_of(const std::vector<std::vector<Pt>>& contours, std::vector<std::vector<Pt>>& cands) {
  cands.clear();
  std::vector<Pt> pts;
  for (const auto& c : contours) {
    double eps = 0.01 * arc_length(c, true);  // metrics.rs:87-90
    if (eps == 0.0) eps = 0.01;
    approximate_polygon_dp(c, eps, true, pts);
    if (pts.size() > 1 && pts.front() == pts.back()) pts.pop_back();
    if (pts.size() < 4) continue;
    cands.push_back(pts);
  }
}

static inline uint32_t as_u32(double v) {  // Rust `as u32`: saturating, NaN -> 0
  if (!(v > 0.0)) return 0;
  if (v >= 4294967295.0) return 4294967295u;
  return (uint32_t)v;
}

bool finish_polygon(const std::vector<Pt>& cand, double score, double adj_x, double adj_y, const ocr_postproc_params_t& prm,
                    std::vector<uint32_t>& xy_out) {
  if (prm.box_thresh > score) return false;  // metrics.rs:100 (NaN passes, as in the reference)
  std::vector<Pt> expanded;
  if (!expand_polygon(cand, prm.unclip_ratio, expanded)) {
    if (prm.skip_degenerate) return false;
    throw DegeneratePolygon();
  }
  Pt box[4];
  if (min_area_bounding_box(expanded, box) < prm.min_size) return false;
  for (const Pt& p : expanded) {
    xy_out.push_back(as_u32(std::round((double)p.x / adj_x)));
    xy_out.push_back(as_u32(std::round((double)p.y / adj_y)));
  }
  return true;
}

}  // namespace geom
}  // namespace ocr
