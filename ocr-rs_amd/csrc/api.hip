// extern "C" boundary (include/ocr_amd.h).  Nothing throws across it.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.hpp"
#include "postproc_geom.hpp"

struct ocr_det {
  ocr::Detector impl;
  ocr_det(const void* b, size_t n, int d) : impl(b, n, d) {}
};
struct ocr_rec {
  ocr::Recognizer impl;
  ocr_rec(const void* b, size_t n, int d) : impl(b, n, d) {}
};

namespace {
thread_local std::string g_err;

template <typename F>
int guard(F&& f) {
  try {
    g_err.clear();
    f();
    return OCR_OK;
  } catch (const ocr::Error& e) {
    g_err = e.what();
    return e.code;
  } catch (const ocr::geom::DegeneratePolygon& e) {
    g_err = e.what();
    return OCR_ERR_DEGENERATE;
  } catch (const std::exception& e) {
    g_err = e.what();
    return OCR_ERR_INTERNAL;
  } catch (...) {
    g_err = "unknown failure";
    return OCR_ERR_INTERNAL;
  }
}

struct PolygonsOwned {
  ocr_polygons_t view;
  std::vector<int32_t> img_offsets, poly_offsets;
  std::vector<uint32_t> xy;
  std::vector<double> scores;
};

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// get_boxes_and_box_scores (metrics.rs:37-56) over the whole batch.
void postprocess(ocr::Detector& det, const float* prob, int n, int h, int w, int mem_kind, const double* adj,
                 const ocr_postproc_params_t& prm, ocr_polygons_t** out) {
  using namespace ocr;
  if (!prob || !adj || !out) fail(OCR_ERR_INVALID, "det_postprocess: null argument");
  if (n <= 0 || h <= 0 || w <= 0) fail(OCR_ERR_INVALID, "det_postprocess: bad shape");
  OCR_HIP(hipSetDevice(det.device()));
  hipStream_t s = det.stream();
  const size_t px = (size_t)n * h * w;
  // scratch: [prob copy if host] [bitmap]
  const size_t off_bitmap = mem_kind == OCR_MEM_HOST ? align256(px * 4) : 0;
  char* scratch = static_cast<char*>(det.scratch(0, off_bitmap + align256(px)));
  const float* prob_dev = prob;
  if (mem_kind == OCR_MEM_HOST) {
    OCR_HIP(hipMemcpyAsync(scratch, prob, px * 4, hipMemcpyHostToDevice, s));
    prob_dev = reinterpret_cast<const float*>(scratch);
  }
  uint8_t* bitmap_dev = reinterpret_cast<uint8_t*>(scratch + off_bitmap);
  launch_binarize(prob_dev, bitmap_dev, (float)prm.thresh, px, s);  // metrics.rs:41,129
  std::vector<uint8_t> bitmap(px);
  OCR_HIP(hipMemcpyAsync(bitmap.data(), bitmap_dev, px, hipMemcpyDeviceToHost, s));
  OCR_HIP(hipStreamSynchronize(s));

  // contour tracing + Douglas-Peucker on host threads, one image at a time per thread
  std::vector<std::vector<std::vector<geom::Pt>>> cands(n);
  {
    std::atomic<int> next{0};
    std::string err;
    std::atomic<bool> failed{false};
    auto work = [&]() {
      try {
        for (int b = next++; b < n; b = next++) geom::contour_candidates(bitmap.data() + (size_t)b * h * w, h, w, cands[b]);
      } catch (const std::exception& e) {
        if (!failed.exchange(true)) err = e.what();
      }
    };
    const int nt = std::max(1, std::min<int>(n, (int)std::thread::hardware_concurrency()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed) fail(OCR_ERR_INTERNAL, "contour stage: %s", err.c_str());
  }

  // box scores on the GPU (metrics.rs:99 -> :150-184)
  std::vector<BoxScoreJob> jobs;
  std::vector<int32_t> pts;
  for (int b = 0; b < n; ++b)
    for (const auto& c : cands[b]) {
      if ((int)c.size() > kBoxScoreMaxPts) fail(OCR_ERR_INVALID, "polygon with %zu vertices exceeds %d", c.size(), kBoxScoreMaxPts);
      int mnx = INT32_MAX, mxx = 0, mny = INT32_MAX, mxy = 0;
      for (const auto& p : c) {
        mnx = std::min(mnx, p.x);
        mxx = std::max(mxx, p.x);
        mny = std::min(mny, p.y);
        mxy = std::max(mxy, p.y);
      }
      // the reference clamps x by size[-2] (=H) and y by size[-1] (=W): metrics.rs:151-166
      const int cw = h, ch = w;
      mnx = std::clamp(mnx, 0, cw - 1);
      mxx = std::clamp(mxx, 0, cw - 1);
      mny = std::clamp(mny, 0, ch - 1);
      mxy = std::clamp(mxy, 0, ch - 1);
      if (mxx >= w || mxy >= h) fail(OCR_ERR_INVALID, "non-square map: box (%d,%d) leaves the %dx%d map (the reference would fail in narrow())", mxx, mxy, w, h);
      BoxScoreJob j{b, (int)(pts.size() / 2), (int)c.size(), mnx, mny, mxx - mnx + 1, mxy - mny + 1};
      jobs.push_back(j);
      for (const auto& p : c) {
        pts.push_back(p.x);
        pts.push_back(p.y);
      }
    }
  const int nj = (int)jobs.size();
  std::vector<double> sums(nj), counts(nj);
  if (nj > 0) {
    const size_t o_jobs = 0;
    const size_t o_pts = o_jobs + align256(jobs.size() * sizeof(BoxScoreJob));
    const size_t o_sum = o_pts + align256(pts.size() * 4);
    const size_t o_cnt = o_sum + align256((size_t)nj * 8);
    const size_t total = o_cnt + align256((size_t)nj * 8);
    scratch = static_cast<char*>(det.scratch(1, total));  // slot 0 (map copy) stays valid
    OCR_HIP(hipMemcpyAsync(scratch + o_jobs, jobs.data(), jobs.size() * sizeof(BoxScoreJob), hipMemcpyHostToDevice, s));
    OCR_HIP(hipMemcpyAsync(scratch + o_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice, s));
    launch_box_scores(prob_dev, h, w, reinterpret_cast<const BoxScoreJob*>(scratch + o_jobs),
                      reinterpret_cast<const int32_t*>(scratch + o_pts), nj, reinterpret_cast<double*>(scratch + o_sum),
                      reinterpret_cast<double*>(scratch + o_cnt), s);
    OCR_HIP(hipMemcpyAsync(sums.data(), scratch + o_sum, (size_t)nj * 8, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipMemcpyAsync(counts.data(), scratch + o_cnt, (size_t)nj * 8, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
  }

  // unclip + filters + coordinate adjustment (metrics.rs:100-123), assemble the CSR block
  auto res = std::make_unique<PolygonsOwned>();
  res->img_offsets.push_back(0);
  res->poly_offsets.push_back(0);
  int j = 0;
  for (int b = 0; b < n; ++b) {
    for (const auto& c : cands[b]) {
      const double score = sums[j] / counts[j];
      ++j;
      if (geom::finish_polygon(c, score, adj[2 * b], adj[2 * b + 1], prm, res->xy)) {
        res->poly_offsets.push_back((int32_t)(res->xy.size() / 2));
        res->scores.push_back(score);
      }
    }
    res->img_offsets.push_back((int32_t)res->scores.size());
  }
  res->view.n_images = n;
  res->view.n_polygons = (int32_t)res->scores.size();
  res->view.n_vertices = (int32_t)(res->xy.size() / 2);
  res->view.img_offsets = res->img_offsets.data();
  res->view.poly_offsets = res->poly_offsets.data();
  res->view.xy = res->xy.data();
  res->view.scores = res->scores.data();
  *out = &res.release()->view;
}
}  // namespace

extern "C" {

const char* ocr_last_error(void) { return g_err.c_str(); }
const char* ocr_version(void) { return "ocr_amd 0.1 gfx950"; }
int ocr_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int ocr_det_create(const void* weights, size_t bytes, int device, ocr_det_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_det_create: out is null");
    *out = nullptr;
    *out = new ocr_det(weights, bytes, device);
  });
}
int ocr_det_create_from_varstore(const char* path, int device, ocr_det_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_det_create_from_varstore: out is null");
    *out = nullptr;
    const std::vector<uint8_t> blob = ocr::varstore_to_blob(path, 1);
    *out = new ocr_det(blob.data(), blob.size(), device);
  });
}
void ocr_det_destroy(ocr_det_t* det) { delete det; }

int ocr_varstore_to_blob(const char* path, int kind, void** blob, size_t* blob_bytes) {
  return guard([&] {
    if (!blob || !blob_bytes) ocr::fail(OCR_ERR_INVALID, "ocr_varstore_to_blob: null output");
    *blob = nullptr;
    *blob_bytes = 0;
    if (kind < 0 || kind > 2) ocr::fail(OCR_ERR_INVALID, "ocr_varstore_to_blob: kind %d", kind);
    const std::vector<uint8_t> b = ocr::varstore_to_blob(path, kind);
    void* p = std::malloc(b.size() ? b.size() : 1);
    if (!p) ocr::fail(OCR_ERR_INTERNAL, "out of memory");
    std::memcpy(p, b.data(), b.size());
    *blob = p;
    *blob_bytes = b.size();
  });
}
void ocr_blob_free(void* blob) { std::free(blob); }

int ocr_det_set_stream(ocr_det_t* det, void* s) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.set_stream(static_cast<hipStream_t>(s));
  });
}

int ocr_det_set_precision(ocr_det_t* det, int precision) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.set_precision(precision);
  });
}

int ocr_det_forward(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, int mem_kind) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    if (mem_kind == OCR_MEM_HOST) {
      if (!x || !prob) ocr::fail(OCR_ERR_INVALID, "det_forward: null tensor");
      if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) ocr::fail(OCR_ERR_INVALID, "det_forward: N=%d H=%d W=%d (H and W must be positive multiples of 32)", n, h, w);
      det->impl.forward_host(x, n, h, w, prob);
    } else if (mem_kind == OCR_MEM_DEVICE) {
      det->impl.forward(x, n, h, w, prob, nullptr, 0.f, nullptr);
      det->impl.synchronize();
    } else {
      ocr::fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    }
  });
}

int ocr_det_forward_async(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.forward(x, n, h, w, prob, bitmap, thresh, nullptr);
  });
}

int ocr_det_synchronize(ocr_det_t* det) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.synchronize();
  });
}

int ocr_det_forward_profile(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, int max_entries,
                            const char** names, float* ms, double* flops, double* bytes, int* n_entries) {
  return guard([&] {
    if (!det || !n_entries) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<ocr::ProfileEntry> prof;
    det->impl.forward(x, n, h, w, prob, nullptr, 0.f, &prof);
    const int k = std::min<int>(max_entries, (int)prof.size());
    for (int i = 0; i < k; ++i) {
      if (names) names[i] = prof[i].name;
      if (ms) ms[i] = prof[i].ms;
      if (flops) flops[i] = prof[i].flops;
      if (bytes) bytes[i] = prof[i].bytes;
    }
    *n_entries = k;
  });
}

int ocr_preprocess_image(ocr_det_t* det, const uint8_t* rgba, int w, int h, int target_w, int target_h, uint8_t* gray,
                         float* gray_f32, double* adj_xy, int mem_kind) {
  return guard([&] {
    using namespace ocr;
    if (!det || !rgba || (!gray && !gray_f32)) fail(OCR_ERR_INVALID, "preprocess_image: null argument");
    if (w < 1 || h < 1 || target_w < 1 || target_h < 1) fail(OCR_ERR_INVALID, "preprocess_image: bad dimensions");
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const size_t px_in = (size_t)w * h * 4, px_out = (size_t)target_w * target_h;
    const size_t need = preprocess_scratch_bytes(w, h, target_w, target_h);
    if (mem_kind == OCR_MEM_DEVICE) {
      // gray may be null when only the f32 frame is wanted: stage it in scratch
      char* sc = static_cast<char*>(det->impl.scratch(1, need + align256(px_out)));
      uint8_t* g = gray ? gray : reinterpret_cast<uint8_t*>(sc + align256(need));
      launch_preprocess(rgba, w, h, target_w, target_h, g, gray_f32, sc, need, adj_xy, s);
      OCR_HIP(hipStreamSynchronize(s));
    } else {
      const size_t o_in = align256(need), o_g = o_in + align256(px_in), o_f = o_g + align256(px_out);
      char* sc = static_cast<char*>(det->impl.scratch(1, o_f + align256(px_out * 4)));
      OCR_HIP(hipMemcpyAsync(sc + o_in, rgba, px_in, hipMemcpyHostToDevice, s));
      launch_preprocess(reinterpret_cast<const unsigned char*>(sc + o_in), w, h, target_w, target_h,
                        reinterpret_cast<unsigned char*>(sc + o_g), gray_f32 ? reinterpret_cast<float*>(sc + o_f) : nullptr, sc, need,
                        adj_xy, s);
      if (gray) OCR_HIP(hipMemcpyAsync(gray, sc + o_g, px_out, hipMemcpyDeviceToHost, s));
      if (gray_f32) OCR_HIP(hipMemcpyAsync(gray_f32, sc + o_f, px_out * 4, hipMemcpyDeviceToHost, s));
      OCR_HIP(hipStreamSynchronize(s));
    }
  });
}

int ocr_extract_crops(ocr_det_t* det, const float* frames, int n, int h, int w, int mem_kind, const ocr_polygons_t* polys,
                      const double* adj_xy, float* crops) {
  return guard([&] {
    using namespace ocr;
    if (!det || !frames || !polys || !adj_xy || (!crops && polys->n_polygons > 0)) fail(OCR_ERR_INVALID, "extract_crops: null argument");
    if (polys->n_images != n) fail(OCR_ERR_INVALID, "extract_crops: polygon block holds %d images, frames %d", polys->n_images, n);
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    const int np = polys->n_polygons;
    if (np == 0) return;
    std::vector<CropBox> boxes;
    boxes.reserve(np);
    for (int b = 0; b < n; ++b) {
      const double ax = adj_xy[2 * b], ay = adj_xy[2 * b + 1];
      for (int k = polys->img_offsets[b]; k < polys->img_offsets[b + 1]; ++k) {
        double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
        for (int v = polys->poly_offsets[k]; v < polys->poly_offsets[k + 1]; ++v) {
          const double x = polys->xy[2 * v] * ax, y = polys->xy[2 * v + 1] * ay;  // back to frame coordinates
          mnx = std::min(mnx, x); mxx = std::max(mxx, x);
          mny = std::min(mny, y); mxy = std::max(mxy, y);
        }
        const double x0 = std::min(std::max(mnx, 0.0), w - 1.0), x1 = std::min(std::max(mxx + 1.0, x0 + 1.0), (double)w);
        const double y0 = std::min(std::max(mny, 0.0), h - 1.0), y1 = std::min(std::max(mxy + 1.0, y0 + 1.0), (double)h);
        boxes.push_back({b, (float)x0, (float)y0, (float)x1, (float)y1});
      }
    }
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const size_t fr_bytes = (size_t)n * h * w * 4, bx_bytes = boxes.size() * sizeof(CropBox), cr_bytes = (size_t)np * 784 * 4;
    if (mem_kind == OCR_MEM_DEVICE) {
      char* sc = static_cast<char*>(det->impl.scratch(1, align256(bx_bytes)));
      OCR_HIP(hipMemcpyAsync(sc, boxes.data(), bx_bytes, hipMemcpyHostToDevice, s));
      launch_crops(frames, h, w, reinterpret_cast<const CropBox*>(sc), np, crops, s);
      OCR_HIP(hipStreamSynchronize(s));
    } else {
      const size_t o_fr = align256(bx_bytes), o_cr = o_fr + align256(fr_bytes);
      char* sc = static_cast<char*>(det->impl.scratch(0, o_cr + align256(cr_bytes)));
      OCR_HIP(hipMemcpyAsync(sc, boxes.data(), bx_bytes, hipMemcpyHostToDevice, s));
      OCR_HIP(hipMemcpyAsync(sc + o_fr, frames, fr_bytes, hipMemcpyHostToDevice, s));
      launch_crops(reinterpret_cast<const float*>(sc + o_fr), h, w, reinterpret_cast<const CropBox*>(sc), np,
                   reinterpret_cast<float*>(sc + o_cr), s);
      OCR_HIP(hipMemcpyAsync(crops, sc + o_cr, cr_bytes, hipMemcpyDeviceToHost, s));
      OCR_HIP(hipStreamSynchronize(s));
    }
  });
}

static std::vector<std::vector<ocr::geom::Pt>> csr_polys(const uint32_t* xy, const int32_t* offsets, int n) {
  std::vector<std::vector<ocr::geom::Pt>> out(n);
  for (int k = 0; k < n; ++k)
    for (int v = offsets[k]; v < offsets[k + 1]; ++v) out[k].push_back({(int)xy[2 * v], (int)xy[2 * v + 1]});
  return out;
}

int ocr_evaluate_image(const uint32_t* gt_xy, const int32_t* gt_offsets, int n_gt, const uint8_t* ignore_flags,
                       const uint32_t* pred_xy, const int32_t* pred_offsets, int n_pred, ocr_metrics_item_t* out) {
  return guard([&] {
    if (!out || n_gt < 0 || n_pred < 0 || (n_gt > 0 && (!gt_xy || !gt_offsets || !ignore_flags)) || (n_pred > 0 && (!pred_xy || !pred_offsets)))
      ocr::fail(OCR_ERR_INVALID, "evaluate_image: null argument");
    std::vector<bool> ign(n_gt);
    for (int i = 0; i < n_gt; ++i) ign[i] = ignore_flags[i] != 0;
    const ocr::geom::MetricsItem m = ocr::geom::evaluate_image(csr_polys(gt_xy, gt_offsets, n_gt), ign, csr_polys(pred_xy, pred_offsets, n_pred));
    out->precision = m.precision;
    out->recall = m.recall;
    out->hmean = m.hmean;
    out->gt_care = m.gt_care;
    out->det_care = m.det_care;
    out->det_matched = m.det_matched;
  });
}

int ocr_combine_results(const ocr_metrics_item_t* items, int n, double* precision, double* recall, double* hmean) {
  return guard([&] {
    if ((n > 0 && !items) || !precision || !recall || !hmean) ocr::fail(OCR_ERR_INVALID, "combine_results: null argument");
    std::vector<ocr::geom::MetricsItem> v(n);
    for (int i = 0; i < n; ++i) v[i] = {items[i].precision, items[i].recall, items[i].hmean, items[i].gt_care, items[i].det_care, items[i].det_matched};
    ocr::geom::combine_results(v.data(), n, precision, recall, hmean);
  });
}

void ocr_postproc_default_params(ocr_postproc_params_t* p) {
  if (!p) return;
  p->thresh = 0.6;        // metrics.rs:38
  p->box_thresh = 0.7;    // metrics.rs:64
  p->min_size = 5.0;      // metrics.rs:66
  p->unclip_ratio = 2.0;  // metrics.rs:103
  p->skip_degenerate = 0; // faithful: the reference aborts on such a candidate
  p->reserved = 0;
}

int ocr_det_postprocess(ocr_det_t* det, const float* prob, int n, int h, int w, int mem_kind, const double* adj,
                        const ocr_postproc_params_t* params, ocr_polygons_t** out) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "det_postprocess needs a detector handle (GPU + stream)");
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) ocr::fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    ocr_postproc_params_t prm;
    ocr_postproc_default_params(&prm);
    if (params) prm = *params;
    if (out) *out = nullptr;
    postprocess(det->impl, prob, n, h, w, mem_kind, adj, prm, out);
  });
}

void ocr_polygons_free(ocr_polygons_t* p) {
  if (!p) return;
  delete reinterpret_cast<PolygonsOwned*>(reinterpret_cast<char*>(p) - offsetof(PolygonsOwned, view));
}

int ocr_rec_create(const void* weights, size_t bytes, int device, ocr_rec_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_rec_create: out is null");
    *out = nullptr;
    *out = new ocr_rec(weights, bytes, device);
  });
}
int ocr_rec_create_from_varstore(const char* path, int device, ocr_rec_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_rec_create_from_varstore: out is null");
    *out = nullptr;
    const std::vector<uint8_t> blob = ocr::varstore_to_blob(path, 2);
    *out = new ocr_rec(blob.data(), blob.size(), device);
  });
}
void ocr_rec_destroy(ocr_rec_t* rec) { delete rec; }
int ocr_rec_set_stream(ocr_rec_t* rec, void* s) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.set_stream(static_cast<hipStream_t>(s));
  });
}
int ocr_rec_synchronize(ocr_rec_t* rec) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.synchronize();
  });
}
int ocr_rec_forward(ocr_rec_t* rec, const float* crops, int n, float* logits, int mem_kind) {
  return guard([&] {
    if (!rec || !crops || !logits) ocr::fail(OCR_ERR_INVALID, "null argument");
    if (n < 0) ocr::fail(OCR_ERR_INVALID, "negative crop count");
    if (mem_kind == OCR_MEM_HOST) rec->impl.forward_host(crops, n, logits, nullptr, nullptr);
    else {
      rec->impl.classify(crops, n, logits, nullptr, nullptr);
      rec->impl.synchronize();
    }
  });
}
int ocr_rec_classify_async(ocr_rec_t* rec, const float* crops, int n, float* logits, int32_t* labels, double* probs) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.classify(crops, n, logits, labels, probs);
  });
}
int ocr_rec_classify_profile(ocr_rec_t* rec, const float* crops, int n, int32_t* labels, double* probs, int max_entries,
                             const char** names, float* ms, double* flops, double* bytes, int* n_entries) {
  return guard([&] {
    if (!rec || !n_entries) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<ocr::ProfileEntry> prof;
    rec->impl.classify(crops, n, nullptr, labels, probs, &prof);
    const int k = std::min<int>(max_entries, (int)prof.size());
    for (int i = 0; i < k; ++i) {
      if (names) names[i] = prof[i].name;
      if (ms) ms[i] = prof[i].ms;
      if (flops) flops[i] = prof[i].flops;
      if (bytes) bytes[i] = prof[i].bytes;
    }
    *n_entries = k;
  });
}
int ocr_rec_classify(ocr_rec_t* rec, const float* crops, int n, int32_t* labels, double* probs, int mem_kind) {
  return guard([&] {
    if (!rec || !crops) ocr::fail(OCR_ERR_INVALID, "null argument");
    if (n < 0) ocr::fail(OCR_ERR_INVALID, "negative crop count");
    if (mem_kind == OCR_MEM_HOST) rec->impl.forward_host(crops, n, nullptr, labels, probs);
    else {
      rec->impl.classify(crops, n, nullptr, labels, probs);
      rec->impl.synchronize();
    }
  });
}
const char* ocr_rec_alphabet(void) { return "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"; }

/* ---- host-geometry test hooks (no GPU needed): used by the CPU test-suite to pin the
 * product's C++ geometry against the reference KATs.  Not part of the drop-in surface. */
int ocr_test_contour_candidates(const uint8_t* bitmap01, int h, int w, int32_t* xy_out, int32_t* counts_out,
                                int max_pts, int max_polys, int* n_polys) {
  return guard([&] {
    std::vector<std::vector<ocr::geom::Pt>> cands;
    ocr::geom::contour_candidates(bitmap01, h, w, cands);
    int np = 0, used = 0;
    for (const auto& c : cands) {
      if (np >= max_polys || used + (int)c.size() > max_pts) ocr::fail(OCR_ERR_INVALID, "test buffer too small");
      counts_out[np++] = (int)c.size();
      for (const auto& p : c) {
        xy_out[2 * used] = p.x;
        xy_out[2 * used + 1] = p.y;
        ++used;
      }
    }
    *n_polys = np;
  });
}
int ocr_test_expand_polygon(const int32_t* xy, int n, double factor, int32_t* xy_out, int max_out, int* n_out,
                            double* sside_out) {
  return guard([&] {
    std::vector<ocr::geom::Pt> in(n), out;
    for (int i = 0; i < n; ++i) in[i] = {xy[2 * i], xy[2 * i + 1]};
    if (!ocr::geom::expand_polygon(in, factor, out)) {
      *n_out = 0;
      return;
    }
    if ((int)out.size() > max_out) ocr::fail(OCR_ERR_INVALID, "test buffer too small");
    for (size_t i = 0; i < out.size(); ++i) {
      xy_out[2 * i] = out[i].x;
      xy_out[2 * i + 1] = out[i].y;
    }
    *n_out = (int)out.size();
    ocr::geom::Pt box[4];
    if (sside_out) *sside_out = ocr::geom::min_area_bounding_box(out, box);
  });
}
int ocr_test_det_stage(ocr_det_t* det, int id, float* out_host, size_t capacity, size_t* elems) {
  return guard([&] {
    if (!det || !elems) ocr::fail(OCR_ERR_INVALID, "null argument");
    const float* p = det->impl.stage(id, elems);
    if (out_host) {
      if (*elems > capacity) ocr::fail(OCR_ERR_INVALID, "stage %d needs %zu floats", id, *elems);
      det->impl.synchronize();
      OCR_HIP(hipMemcpy(out_host, p, *elems * sizeof(float), hipMemcpyDeviceToHost));
    }
  });
}
// raw GPU box scores of given polygons over a host map (sum and pixel count per polygon)
int ocr_test_box_scores(ocr_det_t* det, const float* prob_host, int h, int w, const int32_t* xy, const int32_t* counts,
                        int n_polys, double* sums_out, double* counts_out) {
  return guard([&] {
    using namespace ocr;
    if (!det) fail(OCR_ERR_INVALID, "null handle");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    std::vector<BoxScoreJob> jobs;
    int pos = 0;
    for (int k = 0; k < n_polys; ++k) {
      int mnx = INT32_MAX, mxx = 0, mny = INT32_MAX, mxy = 0;
      for (int i = 0; i < counts[k]; ++i) {
        mnx = std::min(mnx, xy[2 * (pos + i)]);
        mxx = std::max(mxx, xy[2 * (pos + i)]);
        mny = std::min(mny, xy[2 * (pos + i) + 1]);
        mxy = std::max(mxy, xy[2 * (pos + i) + 1]);
      }
      mnx = std::clamp(mnx, 0, h - 1);
      mxx = std::clamp(mxx, 0, h - 1);
      mny = std::clamp(mny, 0, w - 1);
      mxy = std::clamp(mxy, 0, w - 1);
      if (mxx >= w || mxy >= h || counts[k] > kBoxScoreMaxPts) fail(OCR_ERR_INVALID, "bad test polygon");
      jobs.push_back({0, pos, counts[k], mnx, mny, mxx - mnx + 1, mxy - mny + 1});
      pos += counts[k];
    }
    const size_t o_jobs = align256((size_t)h * w * 4), o_pts = o_jobs + align256(jobs.size() * sizeof(BoxScoreJob));
    const size_t o_sum = o_pts + align256((size_t)pos * 8), o_cnt = o_sum + align256((size_t)n_polys * 8);
    char* sc = static_cast<char*>(det->impl.scratch(0, o_cnt + align256((size_t)n_polys * 8)));
    OCR_HIP(hipMemcpyAsync(sc, prob_host, (size_t)h * w * 4, hipMemcpyHostToDevice, s));
    OCR_HIP(hipMemcpyAsync(sc + o_jobs, jobs.data(), jobs.size() * sizeof(BoxScoreJob), hipMemcpyHostToDevice, s));
    OCR_HIP(hipMemcpyAsync(sc + o_pts, xy, (size_t)pos * 8, hipMemcpyHostToDevice, s));
    launch_box_scores(reinterpret_cast<const float*>(sc), h, w, reinterpret_cast<const BoxScoreJob*>(sc + o_jobs),
                      reinterpret_cast<const int32_t*>(sc + o_pts), n_polys, reinterpret_cast<double*>(sc + o_sum),
                      reinterpret_cast<double*>(sc + o_cnt), s);
    OCR_HIP(hipMemcpyAsync(sums_out, sc + o_sum, (size_t)n_polys * 8, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipMemcpyAsync(counts_out, sc + o_cnt, (size_t)n_polys * 8, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
  });
}
// one conv_igemm launch on caller data (kernel-level parity hook).  All host arrays are f32; with in_bf16 /
// out_bf16 they are rounded to bf16 (nearest even) on the way in and widened on the way out, so the caller
// compares against a reference computed from the SAME rounded operands.  in: NHWC; cat4: the four pyramid
// levels p5 (h/8), p4 (h/4), p3 (h/2), p2 (h) back to back, 64 channels each.  wgt: [cout][ks*ks][cin].
int ocr_test_conv_run(ocr_det_t* det, int in_bf16, int out_bf16, const float* in, int n, int h, int w, int cin,
                      const float* wgt, int cout, int ks, int stride, const float* scale, const float* bias,
                      const float* residual, const float* up_residual, int relu, int cat4, float* out, float* out2) {
  return guard([&] {
    using namespace ocr;
    if (!det || !in || !wgt) fail(OCR_ERR_INVALID, "null argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const int pad = (ks - 1) / 2;
    const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    size_t lvl[4] = {0, 0, 0, 0};
    size_t in_e = (size_t)n * h * w * cin;
    if (cat4) {
      if (cin != 256 || (h % 8) || (w % 8)) fail(OCR_ERR_INVALID, "cat4 needs cin 256 and h, w multiples of 8");
      in_e = 0;
      for (int l = 0; l < 4; ++l) {
        lvl[l] = in_e;
        in_e += (size_t)n * (h >> (3 - l)) * (w >> (3 - l)) * 64;
      }
    }
    const size_t w_e = (size_t)cout * ks * ks * cin, out_e = (size_t)n * ho * wo * cout;
    const size_t up_e = (size_t)n * (ho / 2) * (wo / 2) * cout;
    auto bf16_bits = [](float f) {
      uint32_t u;
      std::memcpy(&u, &f, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
      return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    std::vector<void*> allocs;
    auto up = [&](const float* src, size_t elems, bool bf) -> void* {
      if (!src) return nullptr;
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, elems * (bf ? 2 : 4)));
      allocs.push_back(d);
      if (bf) {
        std::vector<uint16_t> t(elems);
        for (size_t i = 0; i < elems; ++i) t[i] = bf16_bits(src[i]);
        OCR_HIP(hipMemcpy(d, t.data(), elems * 2, hipMemcpyHostToDevice));
      } else {
        OCR_HIP(hipMemcpy(d, src, elems * 4, hipMemcpyHostToDevice));
      }
      return d;
    };
    auto down = [&](float* dst, const void* dev, size_t elems, bool bf) {
      if (!dst) return;
      if (bf) {
        std::vector<uint16_t> t(elems);
        OCR_HIP(hipMemcpy(t.data(), dev, elems * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < elems; ++i) {
          const uint32_t u = (uint32_t)t[i] << 16;
          std::memcpy(&dst[i], &u, 4);
        }
      } else {
        OCR_HIP(hipMemcpy(dst, dev, elems * 4, hipMemcpyDeviceToHost));
      }
    };
    const size_t ies = in_bf16 ? 2 : 4, oes = out_bf16 ? 2 : 4;
    char* d_in = static_cast<char*>(up(in, in_e, in_bf16));
    ConvDesc d{};
    d.in_bf16 = in_bf16 ? 1 : 0;
    d.out_bf16 = out_bf16 ? 1 : 0;
    d.src_mode = cat4 ? SRC_CAT4 : SRC_PLAIN;
    d.src[0] = d_in;
    if (cat4) {
      for (int l = 0; l < 4; ++l) d.src[l] = d_in + lvl[l] * ies;
      d.src_base = d_in;
    }
    d.src_bytes = in_e * ies;
    d.wgt = up(wgt, w_e, in_bf16);
    d.wgt_bytes = w_e * ies;
    d.N = n; d.Hin = h; d.Win = w; d.Cin = cin; d.Ho = ho; d.Wo = wo; d.Cout = cout;
    d.ks = ks; d.stride = stride; d.pad = pad;
    d.scale = static_cast<const float*>(up(scale, cout, false));
    d.bias = static_cast<const float*>(up(bias, cout, false));
    d.residual = up(residual, out_e, out_bf16);
    d.up_residual = up(up_residual, up_e, out_bf16);
    d.relu = relu; d.store_mode = STORE_NHWC; d.name = "test_conv";
    void* d_out = nullptr;
    void* d_out2 = nullptr;
    if (out) { OCR_HIP(hipMalloc(&d_out, out_e * oes)); allocs.push_back(d_out); }
    if (out2) { OCR_HIP(hipMalloc(&d_out2, out_e * oes)); allocs.push_back(d_out2); }
    d.out = d_out;
    d.out2 = d_out2;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    launch_conv_igemm(d, s);
    OCR_HIP(hipStreamSynchronize(s));
    down(out, d_out, out_e, out_bf16);
    down(out2, d_out2, out_e, out_bf16);
  });
}
// one 3x3 s1 p1 conv (+ scale / bias / residual / ReLU) through the Winograd F(2x2,3x3) path on caller data:
// weight transform, input transform, batched 16-problem GEMM, output transform.  x: NHWC, wgt: [cout][9][cin].
int ocr_test_winograd_conv(ocr_det_t* det, const float* x, int n, int h, int w, int cin, const float* wgt, int cout,
                           const float* scale, const float* bias, const float* residual, int relu, float* out) {
  return guard([&] {
    using namespace ocr;
    if (!det || !x || !wgt || !out) fail(OCR_ERR_INVALID, "null argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const size_t th = (h + 1) / 2, tw = (w + 1) / 2, T = (size_t)n * th * tw;
    const size_t in_e = (size_t)n * h * w * cin, out_e = (size_t)n * h * w * cout;
    const std::vector<float> u = winograd_weights(wgt, cout, cin);
    std::vector<void*> allocs;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    auto dev = [&](const float* src, size_t elems) -> float* {
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, elems * 4));
      allocs.push_back(d);
      if (src) OCR_HIP(hipMemcpy(d, src, elems * 4, hipMemcpyHostToDevice));
      return static_cast<float*>(d);
    };
    float* d_x = dev(x, in_e);
    float* d_u = dev(u.data(), u.size());
    float* d_v = dev(nullptr, 16 * T * cin);
    float* d_m = dev(nullptr, 16 * T * cout);
    float* d_y = dev(nullptr, out_e);
    const float* d_sc = scale ? dev(scale, cout) : nullptr;
    const float* d_bi = bias ? dev(bias, cout) : nullptr;
    const float* d_res = residual ? dev(residual, out_e) : nullptr;
    if ((cin == 64 || cin == 128 || cin == 256) && cout % 64 == 0 && getenv("OCR_TEST_WINOGRAD_UNFUSED") == nullptr) {  // the fused kernel
      std::vector<float> un = u;
      for (size_t i = (size_t)12 * cout * cin; i < un.size(); ++i) un[i] = -un[i];
      float* d_un = dev(un.data(), un.size());
      launch_winograd_fused(d_x, d_un, d_sc, d_bi, d_res, relu, d_y, n, h, w, cin, cout, s);
      OCR_HIP(hipStreamSynchronize(s));
      OCR_HIP(hipMemcpy(out, d_y, out_e * 4, hipMemcpyDeviceToHost));
      return;
    }
    launch_winograd_input(d_x, d_v, n, h, w, cin, s);
    ConvDesc d{};
    d.src[0] = d_v;
    d.src_mode = SRC_PLAIN;
    d.src_bytes = 16 * T * cin * 4;
    d.wgt = d_u;
    d.wgt_bytes = u.size() * 4;
    d.batch = 16;
    d.N = 1; d.Hin = d.Ho = 1; d.Win = d.Wo = (int)T; d.Cin = cin; d.Cout = cout;
    d.ks = 1; d.stride = 1; d.pad = 0; d.store_mode = STORE_NHWC; d.out = d_m; d.name = "test_winograd";
    launch_conv_igemm(d, s);
    launch_winograd_output(d_m, d_sc, d_bi, d_res, relu, d_y, n, h, w, cout, s);
    OCR_HIP(hipStreamSynchronize(s));
    OCR_HIP(hipMemcpy(out, d_y, out_e * 4, hipMemcpyDeviceToHost));
  });
}
int ocr_test_set_conv_tile(int t) {
  ocr::set_conv_tile_override(t);
  return OCR_OK;
}
// micro-benchmark of one conv_igemm launch shape on constant data (kernel tuning aid)
int ocr_test_conv_bench(ocr_det_t* det, int n, int h, int w, int cin, int cout, int ks, int stride, int src_mode,
                        int iters, float* ms_out) {
  return guard([&] {
    using namespace ocr;
    if (!det) fail(OCR_ERR_INVALID, "null handle");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const int pad = (ks - 1) / 2;
    const int ho = (h + 2 * pad - ks) / stride + 1, wo = (w + 2 * pad - ks) / stride + 1;
    const size_t in_e = (size_t)n * h * w * cin, w_e = (size_t)cout * ks * ks * cin, out_e = (size_t)n * ho * wo * cout;
    float *in = nullptr, *wt = nullptr, *out = nullptr;
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&in), in_e * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&wt), w_e * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&out), out_e * 4));
    OCR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(in), 0x3f8ccccd, in_e, s));   // 1.1f
    OCR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(wt), 0x3c23d70a, w_e, s));    // 0.01f
    if (src_mode & 32) {  // random operands: MFMA power (and with it the clock) depends on the data
      src_mode &= ~32;
      std::vector<float> h(std::max(in_e, w_e));
      uint32_t st = 12345u;
      auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
      for (size_t i = 0; i < in_e; ++i) h[i] = rnd();
      OCR_HIP(hipMemcpy(in, h.data(), in_e * 4, hipMemcpyHostToDevice));
      for (size_t i = 0; i < w_e; ++i) h[i] = 0.05f * rnd();
      OCR_HIP(hipMemcpy(wt, h.data(), w_e * 4, hipMemcpyHostToDevice));
    }
    ConvDesc d{};
    d.src[0] = in;
    d.src_mode = SRC_PLAIN;
    const int bf = src_mode == 16 ? 1 : 0;  // src_mode 16: bf16 operands and output (the buffers are just reinterpreted)
    d.in_bf16 = d.out_bf16 = bf;
    d.src_bytes = in_e * (bf ? 2 : 4);
    d.wgt_bytes = w_e * (bf ? 2 : 4); d.N = n; d.Hin = h; d.Win = w; d.Cin = cin; d.Ho = ho; d.Wo = wo; d.Cout = cout;
    d.ks = ks; d.stride = stride; d.pad = pad; d.wgt = wt; d.relu = 1; d.store_mode = STORE_NHWC; d.out = out;
    d.name = "bench";
    hipEvent_t e0, e1;
    OCR_HIP(hipEventCreate(&e0));
    OCR_HIP(hipEventCreate(&e1));
    launch_conv_igemm(d, s);
    OCR_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_conv_igemm(d, s);
    OCR_HIP(hipEventRecord(e1, s));
    OCR_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    OCR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(in);
    (void)hipFree(wt);
    (void)hipFree(out);
  });
}
int ocr_test_min_area_box(const int32_t* xy, int n, int32_t* box_xy, double* sside) {
  return guard([&] {
    std::vector<ocr::geom::Pt> in(n);
    for (int i = 0; i < n; ++i) in[i] = {xy[2 * i], xy[2 * i + 1]};
    ocr::geom::Pt box[4];
    *sside = ocr::geom::min_area_bounding_box(in, box);
    for (int i = 0; i < 4; ++i) {
      box_xy[2 * i] = box[i].x;
      box_xy[2 * i + 1] = box[i].y;
    }
  });
}

}  // extern "C"
