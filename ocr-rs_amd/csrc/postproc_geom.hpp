// Host geometry of the detection post-processing (see postproc_geom.cpp).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../../include/ocr_amd.h"

namespace ocr {
namespace geom {

struct Pt {
  int x, y;
  bool operator==(const Pt& o) const { return x == o.x && y == o.y; }
};

// expand_polygon returned None: the reference `.unwrap()`s it (metrics.rs:103) and aborts.
struct DegeneratePolygon : std::runtime_error {
  DegeneratePolygon() : std::runtime_error("expand_polygon produced no polygon for a zero-area candidate; the reference unwraps None here and aborts (metrics.rs:103)") {}
};

void find_contours(const uint8_t* bitmap01, int h, int w, std::vector<std::vector<Pt>>& out);
// the same on a packed image: bit (i & 31) of word i >> 5 for the row-major pixel index i = y * w + x
void find_contours_bits(const uint32_t* bits, int h, int w, std::vector<std::vector<Pt>>& out);
double arc_length(const std::vector<Pt>& p, bool closed);
void approximate_polygon_dp(const std::vector<Pt>& curve, double eps, bool closed, std::vector<Pt>& out);
double min_area_bounding_box(const std::vector<Pt>& pts, Pt res[4]);
double offset_distance(const std::vector<Pt>& poly, double factor);
void raw_offset_ring(const std::vector<Pt>& poly, double delta, std::vector<Pt>& out);
void positive_union_outer(const std::vector<Pt>& ring, std::vector<Pt>& out);
bool expand_polygon(const std::vector<Pt>& pts, double factor, std::vector<Pt>& out);
long long hypot_port_mismatches(int limit);   // hypot_glibc.hpp against std::hypot (test hook)

// eval metrics (eval_metrics.cpp): metrics.rs:229-394
struct MetricsItem {  // metrics.rs:22-30
  double precision, recall, hmean;
  int gt_care, det_care, det_matched;
};
double polygon_area(const std::vector<Pt>& p);
double intersection_area(const std::vector<Pt>& a, const std::vector<Pt>& b);
double union_area(const std::vector<Pt>& a, const std::vector<Pt>& b);
MetricsItem evaluate_image(const std::vector<std::vector<Pt>>& gt, const std::vector<bool>& ignore,
                           const std::vector<std::vector<Pt>>& pred);
void combine_results(const MetricsItem* r, int n, double* precision, double* recall, double* hmean);

// contours -> Douglas-Peucker polygons with >= 4 points (metrics.rs:78-98)
void contour_candidates(const uint8_t* bitmap01, int h, int w, std::vector<std::vector<Pt>>& cands);
void contour_candidates_bits(const uint32_t* bits, int h, int w, std::vector<std::vector<Pt>>& cands);
// the same from contours traced elsewhere (contours.hip): points packed as y << 16 | x, one length per contour
void contour_candidates_packed(const uint32_t* pts, const int32_t* lens, int n_contours, std::vector<std::vector<Pt>>& cands);
// score threshold, unclip, min-size filter, round(p/adj) as u32 (metrics.rs:100-123).
// Appends x,y pairs to xy_out and returns true when the polygon is kept.
bool finish_polygon(const std::vector<Pt>& cand, double score, double adj_x, double adj_y,
                    const ocr_postproc_params_t& prm, std::vector<uint32_t>& xy_out);

}  // namespace geom
}  // namespace ocr
