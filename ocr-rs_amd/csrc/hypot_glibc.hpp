// glibc 2.35's hypot kernel for operands that need no scaling (sysdeps/ieee754/dbl-64/e_hypot.c, the branch without FMA - what the
// x86-64 build runs), restated so that device code can add up a perimeter exactly as geo's euclidean_length does on the host
// (it calls libm's hypot, which is NOT sqrt(dx^2 + dy^2) to the last bit: 0.6 % of integer pairs differ by an ulp).
// Only valid where every operation is rounded separately (-ffp-contract=off).  postproc_geom.cpp checks it against std::hypot.
#pragma once
#include <cmath>
#ifdef __HIPCC__
#define OCR_HOST_DEVICE __host__ __device__
#else
#define OCR_HOST_DEVICE
#endif
namespace ocr {
OCR_HOST_DEVICE inline double hypot_glibc(double x, double y) {
  double ax = std::fabs(x), ay = std::fabs(y);
  if (ax < ay) {
    const double t = ax;
    ax = ay;
    ay = t;
  }
  if (ay == 0.0) return ax;
  double h = std::sqrt(ax * ax + ay * ay);
  double t1, t2;
  if (h <= 2.0 * ay) {
    const double delta = h - ay;
    t1 = ax * (2.0 * delta - ax);
    t2 = (delta - 2.0 * (ax - ay)) * delta;
  } else {
    const double delta = h - ax;
    t1 = 2.0 * delta * (ax - 2.0 * ay);
    t2 = (4.0 * delta - ay) * ay + delta * delta;
  }
  h -= (t1 + t2) / (2.0 * h);
  return h;
}
}  // namespace ocr
