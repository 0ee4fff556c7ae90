// Shared by the translation units that implement the extern "C" surface (api.hip, comm entry points) and by the
// test-hook library (test_hooks.hip -> libocr_amd_test.so, which links against libocr_amd.so).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "engine.hpp"
#include "postproc_geom.hpp"

struct ocr_det {
  ocr::Detector impl;
  ocr_det(const void* b, size_t n, int d, const char* options = nullptr) : impl(b, n, d, options) {}
};
struct ocr_rec {
  ocr::Recognizer impl;
  ocr_rec(const void* b, size_t n, int d) : impl(b, n, d) {}
};

namespace ocr {

extern thread_local std::string g_last_error;  // what ocr_last_error() returns (api.hip)

// Nothing throws across the C boundary: every entry point runs inside guard().
template <typename F>
int guard(F&& f) {
  try {
    g_last_error.clear();
    f();
    return OCR_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const geom::DegeneratePolygon& e) {
    g_last_error = e.what();
    return OCR_ERR_DEGENERATE;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return OCR_ERR_INTERNAL;
  } catch (...) {
    g_last_error = "unknown failure";
    return OCR_ERR_INTERNAL;
  }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// the library-owned storage behind an ocr_polygons_t* (released by ocr_polygons_free)
struct PolygonsOwned {
  ocr_polygons_t view;
  std::vector<int32_t> img_offsets, poly_offsets;
  std::vector<uint32_t> xy;
  std::vector<double> scores;
  void finish() {
    view.n_images = (int32_t)img_offsets.size() - 1;
    view.n_polygons = (int32_t)scores.size();
    view.n_vertices = (int32_t)(xy.size() / 2);
    view.img_offsets = img_offsets.data();
    view.poly_offsets = poly_offsets.data();
    view.xy = xy.data();
    view.scores = scores.data();
  }
};

// comm.hip
class Comm;
std::vector<uint8_t> pack_shard(const ocr_polygons_t& p);
void assemble_shards(const uint8_t* const* shards, const size_t* bytes, int world, PolygonsOwned& out);
void comm_unique_id(uint8_t* id128);
int comm_rccl_version();
Comm* comm_create(const uint8_t* id, int world, int rank, int device);
void comm_destroy(Comm* c);
int comm_world(const Comm* c);
int comm_rank(const Comm* c);
void comm_all_gather_polygons(Comm* c, const ocr_polygons_t& local, PolygonsOwned& out);
void comm_all_gather_labels(Comm* c, const int32_t* labels, int n_local, std::vector<int32_t>& all, std::vector<int32_t>& counts);

}  // namespace ocr
