// Parser of the OCRW v1 weight blob (layout documented in ocr-rs_amd/weights.py).
#pragma once
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

namespace ocr {

struct TensorView {
  const float* data = nullptr;
  int ndim = 0;
  int dims[4] = {1, 1, 1, 1};
  size_t count = 0;
};

class WeightBlob {
 public:
  WeightBlob(const void* blob, size_t bytes) {
    const uint8_t* b = static_cast<const uint8_t*>(blob);
    if (!b || bytes < 16 || std::memcmp(b, "OCRW", 4) != 0) fail(OCR_ERR_WEIGHTS, "weight blob: bad magic");
    uint32_t ver, n;
    std::memcpy(&ver, b + 4, 4);
    std::memcpy(&n, b + 8, 4);
    if (ver != 1) fail(OCR_ERR_WEIGHTS, "weight blob: unsupported version %u", ver);
    const size_t rec = 64 + 4 + 16 + 8 + 8;
    if (16 + (size_t)n * rec > bytes) fail(OCR_ERR_WEIGHTS, "weight blob: truncated directory");
    for (uint32_t i = 0; i < n; ++i) {
      const uint8_t* r = b + 16 + (size_t)i * rec;
      char name[65];
      std::memcpy(name, r, 64);
      name[64] = 0;
      uint32_t ndim, dims[4];
      uint64_t off, cnt;
      std::memcpy(&ndim, r + 64, 4);
      std::memcpy(dims, r + 68, 16);
      std::memcpy(&off, r + 84, 8);
      std::memcpy(&cnt, r + 92, 8);
      if (ndim > 4 || off % 4 || off + cnt * 4 > bytes) fail(OCR_ERR_WEIGHTS, "weight blob: tensor %s out of range", name);
      size_t prod = 1;
      TensorView t;
      t.ndim = (int)ndim;
      for (uint32_t k = 0; k < 4; ++k) {
        t.dims[k] = k < ndim ? (int)dims[k] : 1;
        prod *= (size_t)t.dims[k];
      }
      if (prod != cnt) fail(OCR_ERR_WEIGHTS, "weight blob: tensor %s count mismatch", name);
      t.count = cnt;
      t.data = reinterpret_cast<const float*>(b + off);
      map_[name] = t;
    }
  }

  const TensorView& get(const std::string& name, std::initializer_list<int> shape) const {
    auto it = map_.find(name);
    if (it == map_.end()) fail(OCR_ERR_WEIGHTS, "weight blob: tensor '%s' missing", name.c_str());
    const TensorView& t = it->second;
    if ((int)shape.size() != t.ndim) fail(OCR_ERR_WEIGHTS, "tensor '%s': rank %d, expected %d", name.c_str(), t.ndim, (int)shape.size());
    int k = 0;
    for (int d : shape) {
      if (t.dims[k] != d) fail(OCR_ERR_WEIGHTS, "tensor '%s': dim %d is %d, expected %d", name.c_str(), k, t.dims[k], d);
      ++k;
    }
    return t;
  }

 private:
  std::map<std::string, TensorView> map_;
};

}  // namespace ocr
