// Crop extraction detect -> recognise: one 28x28 bilinear crop per kept polygon.
// BUILD-DEFINED (the reference has no such step: README.md:20-26 leaves "Character Segmentation"
// unchecked); the rule is written down in oracle/crop_oracle.py and mirrored here operation for
// operation (f32, separately rounded: this file is compiled with -ffp-contract=off).
#include "common.hpp"

namespace ocr {
namespace {

__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ frames, int H, int W, const CropBox* __restrict__ boxes,
                                                   float* __restrict__ crops) {
  const CropBox bx = boxes[blockIdx.x];
  const float* img = frames + (size_t)bx.frame * H * W;
  const float sxs = (bx.x1 - bx.x0) / 28.0f, sys = (bx.y1 - bx.y0) / 28.0f;
  for (int o = threadIdx.x; o < 784; o += 256) {
    const int i = o / 28, j = o - i * 28;
    float sy = (bx.y0 + ((float)i + 0.5f) * sys) - 0.5f;
    sy = fminf(fmaxf(sy, 0.f), (float)(H - 1));
    const int iy0 = (int)floorf(sy), iy1 = min(iy0 + 1, H - 1);
    const float fy = sy - (float)iy0;
    float sx = (bx.x0 + ((float)j + 0.5f) * sxs) - 0.5f;
    sx = fminf(fmaxf(sx, 0.f), (float)(W - 1));
    const int ix0 = (int)floorf(sx), ix1 = min(ix0 + 1, W - 1);
    const float fx = sx - (float)ix0;
    const float a = img[(size_t)iy0 * W + ix0], b = img[(size_t)iy0 * W + ix1];
    const float c = img[(size_t)iy1 * W + ix0], d = img[(size_t)iy1 * W + ix1];
    const float top = a + fx * (b - a), bot = c + fx * (d - c);
    crops[(size_t)blockIdx.x * 784 + o] = (top + fy * (bot - top)) / 255.0f;
  }
}

}  // namespace

void launch_crops(const float* frames_dev, int H, int W, const CropBox* boxes_dev, int n_boxes, float* crops_dev, hipStream_t s) {
  if (n_boxes <= 0) return;
  hipLaunchKernelGGL(crop_kernel, dim3(n_boxes), dim3(256), 0, s, frames_dev, H, W, boxes_dev, crops_dev);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
