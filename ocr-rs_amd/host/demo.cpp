// `ocr-rs td` / `ocr-rs cr` style plumbing in C++ over the mirror header:
//   demo <det_weights.ocrw> <rec_weights.ocrw>
// runs one synthetic 64x64 frame through detect -> polygons and 4 crops through recognise.
#include <cstdio>
#include <fstream>
#include <iterator>

#include "ocr_rs.hpp"

static std::vector<char> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw ocr_rs::Error(OCR_ERR_WEIGHTS, std::string("Model file ") + path + " doesn't exist");  // mod.rs:36-39
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s det.ocrw rec.ocrw\n", argv[0]);
    return 2;
  }
  try {
    using namespace ocr_rs;
    auto dw = slurp(argv[1]);
    auto rw = slurp(argv[2]);
    auto net = text_detection::resnet18(dw.data(), dw.size(), 0);
    Tensor x(1, 1, 64, 64);
    for (size_t i = 0; i < x.data.size(); ++i) x.data[i] = (float)((i * 2654435761u >> 24) & 255);
    Tensor pred = net.forward_t(x, false);
    std::printf("pred[0] = %f\n", pred.data[0]);
    char_recognition::Net rec(rw.data(), rw.size(), 0);
    std::vector<float> crops(4 * 784, 0.25f);
    for (auto& p : rec.predict(crops)) std::printf("classified as %c with %3.2f%% of certainty\n", p.first, p.second * 100.0);
  } catch (const ocr_rs::Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
