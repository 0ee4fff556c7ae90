// Header-only C++ mirror of the reference's inference call sites over the C ABI
// (include/ocr_amd.h).  The reference is a Rust binary crate; Rust is not available
// in this image, so the host side above the C ABI is C++ with the reference's names,
// argument meaning and error behaviour (anyhow::Result -> ocr_rs::Error).
//
//   text_detection::resnet18(..) -> FuncT, FuncT::forward_t      model.rs:154-156, mod.rs:52-54
//   text_detection::metrics::get_boxes_and_box_scores            metrics.rs:37-56
//   char_recognition::Net::{new_, forward_t}, utils::topk        model.rs:13-39, utils.rs:28-43
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ocr_amd.h"

namespace ocr_rs {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int code) {
  if (code != OCR_OK) throw Error(code, ocr_last_error());
}

// N x C x H x W f32 host tensor (what a tch::Tensor of Kind::Float holds)
struct Tensor {
  std::vector<float> data;
  int n = 0, c = 0, h = 0, w = 0;
  Tensor() = default;
  Tensor(int n_, int c_, int h_, int w_) : data((size_t)n_ * c_ * h_ * w_), n(n_), c(c_), h(h_), w(w_) {}
};

namespace text_detection {

constexpr uint32_t DEFAULT_WIDTH = 800, DEFAULT_HEIGHT = 800;  // mod.rs:20-21

class FuncT {  // what resnet18(&nn::Path) returns
 public:
  FuncT(const void* weights, size_t bytes, int device) { check(ocr_det_create(weights, bytes, device, &h_)); }
  ~FuncT() { ocr_det_destroy(h_); }
  FuncT(const FuncT&) = delete;
  FuncT& operator=(const FuncT&) = delete;
  // ModuleT::forward_t(&self, xs, train)
  Tensor forward_t(const Tensor& xs, bool train) const {
    if (train) throw Error(OCR_ERR_INVALID, "inference-only build");
    if (xs.c != 1) throw Error(OCR_ERR_INVALID, "expected N x 1 x H x W");
    Tensor out(xs.n, 1, xs.h, xs.w);
    check(ocr_det_forward(h_, xs.data.data(), xs.n, xs.h, xs.w, out.data.data(), OCR_MEM_HOST));
    return out;
  }
  ocr_det_t* handle() const { return h_; }

 private:
  ocr_det_t* h_ = nullptr;
};
inline FuncT resnet18(const void* weights, size_t bytes, int device = 0) { return FuncT(weights, bytes, device); }

namespace metrics {
using Polygon = std::vector<std::pair<uint32_t, uint32_t>>;
using MultiPolygon = std::vector<Polygon>;
struct PolygonScores {  // metrics.rs:32-35
  std::vector<MultiPolygon> polygons;
  std::vector<std::vector<double>> scores;
};
// get_boxes_and_box_scores(pred: &Tensor, adjust_values: &Tensor) -> Result<PolygonScores>
inline PolygonScores get_boxes_and_box_scores(const FuncT& net, const Tensor& pred, const std::vector<double>& adjust_values) {
  if ((int)adjust_values.size() != 2 * pred.n) throw Error(OCR_ERR_INVALID, "adjust_values must be N x 2");
  ocr_polygons_t* r = nullptr;
  check(ocr_det_postprocess(net.handle(), pred.data.data(), pred.n, pred.h, pred.w, OCR_MEM_HOST, adjust_values.data(), nullptr, &r));
  PolygonScores out;
  for (int b = 0; b < r->n_images; ++b) {
    MultiPolygon mp;
    std::vector<double> sc;
    for (int k = r->img_offsets[b]; k < r->img_offsets[b + 1]; ++k) {
      Polygon p;
      for (int v = r->poly_offsets[k]; v < r->poly_offsets[k + 1]; ++v) p.emplace_back(r->xy[2 * v], r->xy[2 * v + 1]);
      mp.push_back(std::move(p));
      sc.push_back(r->scores[k]);
    }
    out.polygons.push_back(std::move(mp));
    out.scores.push_back(std::move(sc));
  }
  ocr_polygons_free(r);
  return out;
}
}  // namespace metrics
}  // namespace text_detection

namespace utils {
inline const char* VALUES() { return ocr_rec_alphabet(); }  // utils.rs:7
}

namespace char_recognition {
class Net {
 public:
  Net(const void* weights, size_t bytes, int device = 0) { check(ocr_rec_create(weights, bytes, device, &h_)); }
  ~Net() { ocr_rec_destroy(h_); }
  Net(const Net&) = delete;
  Net& operator=(const Net&) = delete;
  // forward_t: xs.view([-1, 1, 28, 28]) ... fc2 -> N x 62 logits
  std::vector<float> forward_t(const std::vector<float>& xs, bool train) const {
    if (train) throw Error(OCR_ERR_INVALID, "inference-only build");
    if (xs.size() % 784 != 0) throw Error(OCR_ERR_INVALID, "shape is invalid for view([-1, 1, 28, 28])");
    const int n = (int)(xs.size() / 784);
    std::vector<float> logits((size_t)n * 62);
    check(ocr_rec_forward(h_, xs.data(), n, logits.data(), OCR_MEM_HOST));
    return logits;
  }
  // run_prediction's tail: softmax(-1, Double) + topk(.., 1)[0]  (mod.rs:53-56)
  std::vector<std::pair<char, double>> predict(const std::vector<float>& xs) const {
    const int n = (int)(xs.size() / 784);
    std::vector<int32_t> labels(n);
    std::vector<double> probs(n);
    check(ocr_rec_classify(h_, xs.data(), n, labels.data(), probs.data(), OCR_MEM_HOST));
    std::vector<std::pair<char, double>> out;
    for (int i = 0; i < n; ++i) out.emplace_back(utils::VALUES()[labels[i]], probs[i]);
    return out;
  }

 private:
  ocr_rec_t* h_ = nullptr;
};
}  // namespace char_recognition

}  // namespace ocr_rs
