// Test and tuning hooks (libocr_amd_test.so): NOT part of the drop-in surface and not in libocr_amd.so.  They reach
// into the product library's internals (it exports its C++ symbols) so that kernels and host geometry can be pinned
// one piece at a time: the host-geometry hooks need no GPU and pin the C++ geometry against the reference KATs.
#include <algorithm>
#include <atomic>
#include <cstring>

#include "api_internal.hpp"

using ocr::guard;
using ocr::align256;
namespace ocr { void winograd43_set_debug(int d); }
#ifdef W43_STAMPS
namespace ocr { void winograd43_read_stamps(long long* out); void winograd43_x3_read_stamps(long long* out); }
#endif
#ifdef STEM_STAMPS
namespace ocr { void stem_read_stamps(long long* out); }
extern "C" int ocr_test_stem_stamps(long long* out) { ocr::stem_read_stamps(out); return 0; }
#endif
#ifdef WS_STAMPS
#endif

extern "C" {

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
// the device tracer (contours.hip) on one 0/1 bitmap: raw contours (not the Douglas-Peucker candidates); status = the kernel's
// (0 ok, 1 buffers too small, 2 iteration guard, 3 parallel form only: a start outside its list).  max_pts / max_polys double as the kernel's capacities.
int ocr_test_device_contours(const uint8_t* bitmap01, int h, int w, int32_t* xy_out, int32_t* counts_out, int max_pts, int max_polys,
                             int* n_polys, int* status, int sequential) {
  return guard([&] {
    if (!ocr::contour_trace_fits(h, w)) ocr::fail(OCR_ERR_INVALID, "device contours: %dx%d does not fit", h, w);
    const size_t npx = (size_t)h * w, wpi = ocr::binarize_pack_words(npx);
    std::vector<uint32_t> bits(wpi, 0u);
    for (size_t i = 0; i < npx; ++i)
      if (bitmap01[i]) bits[i >> 5] |= 1u << (i & 31);
    uint32_t *d_bits = nullptr, *d_pts = nullptr, *d_pk = nullptr;
    int *d_st = nullptr, *d_hdr = nullptr, *d_ln = nullptr;
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_bits), wpi * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_pts), (size_t)max_pts * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_pk), (size_t)max_pts * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_st), ((size_t)max_polys + 1) * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_ln), (size_t)max_polys * 4));
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d_hdr), 16));
    void* d_spec = nullptr;
    OCR_HIP(hipMalloc(&d_spec, ocr::contour_spec_bytes(1)));
    OCR_HIP(hipMemcpy(d_bits, bits.data(), wpi * 4, hipMemcpyHostToDevice));
    ocr::launch_contour_trace(d_bits, wpi, 1, h, w, d_pts, max_pts, d_st, max_polys, d_hdr, d_pk, d_ln, d_spec, sequential, nullptr);
    int hdr[4];
    OCR_HIP(hipMemcpy(hdr, d_hdr, 16, hipMemcpyDeviceToHost));
    *status = hdr[2];
    *n_polys = 0;
    if (hdr[2] == 0) {
      std::vector<uint32_t> pts((size_t)hdr[1]);
      std::vector<int32_t> lens((size_t)hdr[0]);
      if (hdr[1]) OCR_HIP(hipMemcpy(pts.data(), d_pk, pts.size() * 4, hipMemcpyDeviceToHost));
      if (hdr[0]) OCR_HIP(hipMemcpy(lens.data(), d_ln, lens.size() * 4, hipMemcpyDeviceToHost));
      for (int k = 0; k < hdr[0]; ++k) counts_out[k] = lens[k];
      for (int i = 0; i < hdr[1]; ++i) {
        xy_out[2 * i] = (int32_t)(pts[i] & 0xffffu);
        xy_out[2 * i + 1] = (int32_t)(pts[i] >> 16);
      }
      *n_polys = hdr[0];
    }
    for (void* q : {(void*)d_bits, (void*)d_pts, (void*)d_pk, (void*)d_st, (void*)d_ln, (void*)d_hdr, d_spec}) (void)hipFree(q);
  });
}
// candidates.hip alone (arc length, Douglas-Peucker, the >= 4 points filter, job list with the clamped boxes) on contours GIVEN by the
// caller - any map size, e.g. the oracle's contours of the reference's 800 x 800 fixtures - laid out as the device tracer would have
// left them for one image.  Out: the candidate polygons in list order and per candidate its job box (min_x, min_y, bw, bh).
int ocr_test_device_candidates(const int32_t* xy, const int32_t* lens, int nc, int h, int w, int32_t* xy_out, int32_t* counts_out, int32_t* box_out,
                               int max_pts, int max_polys, int* n_polys) {
  return guard([&] {
    if (nc < 0 || nc > 65535 || h <= 0 || w <= 0 || h > 65535 || w > 65535) ocr::fail(OCR_ERR_INVALID, "device candidates: bad shape");
    size_t total = 0;
    for (int c = 0; c < nc; ++c) total += (size_t)lens[c];
    const int cap = (int)std::max<size_t>(total, 64), maxc = std::max(nc, 1);
    std::vector<uint32_t> pts((size_t)cap, 0u);
    std::vector<int> starts((size_t)maxc + 1, 0);
    size_t at = 0;
    for (int c = 0; c < nc; ++c) {
      starts[c] = (int)at;
      for (int i = 0; i < lens[c]; ++i, ++at) pts[at] = ((uint32_t)xy[2 * at + 1] << 16) | ((uint32_t)xy[2 * at] & 0xffffu);
    }
    for (int c = nc; c <= maxc; ++c) starts[c] = (int)at;
    const int hdr[4] = {nc, (int)total, 0, 0};
    const size_t sb = ocr::candidates_scratch_bytes(1, cap, maxc);
    char* d = nullptr;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_pts = al(16), o_st = o_pts + al((size_t)cap * 4), o_sc = o_st + al(((size_t)maxc + 1) * 4), o_jobs = o_sc + al(sb),
                 o_xy = o_jobs + al((size_t)max_polys * sizeof(ocr::BoxScoreJob)), o_tot = o_xy + al((size_t)max_pts * 8), o_hd = o_tot + 256, end = o_hd + 256;
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d), end));
    OCR_HIP(hipMemset(d, 0, end));
    OCR_HIP(hipMemcpy(d, hdr, 16, hipMemcpyHostToDevice));
    OCR_HIP(hipMemcpy(d + o_pts, pts.data(), (size_t)cap * 4, hipMemcpyHostToDevice));
    OCR_HIP(hipMemcpy(d + o_st, starts.data(), ((size_t)maxc + 1) * 4, hipMemcpyHostToDevice));
    ocr::launch_candidates(reinterpret_cast<const int*>(d), reinterpret_cast<const uint32_t*>(d + o_pts), cap, reinterpret_cast<const int*>(d + o_st), maxc, 1, h, w,
                           d + o_sc, reinterpret_cast<ocr::BoxScoreJob*>(d + o_jobs), max_polys, reinterpret_cast<int32_t*>(d + o_xy), max_pts,
                           reinterpret_cast<int*>(d + o_tot), reinterpret_cast<int*>(d + o_hd), nullptr);
    int totals[4] = {0, 0, 0, 0};
    OCR_HIP(hipMemcpy(totals, d + o_hd, 12, hipMemcpyDeviceToHost));
    if (totals[2] != 0) {
      (void)hipFree(d);
      ocr::fail(OCR_ERR_INVALID, "test buffer too small");
    }
    std::vector<ocr::BoxScoreJob> jobs((size_t)totals[0]);
    std::vector<int32_t> oxy(2 * (size_t)totals[1]);
    if (totals[0]) OCR_HIP(hipMemcpy(jobs.data(), d + o_jobs, jobs.size() * sizeof(ocr::BoxScoreJob), hipMemcpyDeviceToHost));
    if (totals[1]) OCR_HIP(hipMemcpy(oxy.data(), d + o_xy, oxy.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    size_t used = 0;
    for (int j = 0; j < totals[0]; ++j) {
      const auto& jb = jobs[j];
      counts_out[j] = jb.n_pts;
      box_out[4 * j] = jb.min_x;
      box_out[4 * j + 1] = jb.min_y;
      box_out[4 * j + 2] = jb.bw;
      box_out[4 * j + 3] = jb.bh;
      for (int i = 0; i < jb.n_pts; ++i, ++used) {
        xy_out[2 * used] = oxy[2 * ((size_t)jb.pt_offset + i)];
        xy_out[2 * used + 1] = oxy[2 * ((size_t)jb.pt_offset + i) + 1];
      }
    }
    *n_polys = totals[0];
  });
}
// the host tracer's raw contours, for the same comparison
int ocr_test_host_contours(const uint8_t* bitmap01, int h, int w, int32_t* xy_out, int32_t* counts_out, int max_pts, int max_polys, int* n_polys) {
  return guard([&] {
    std::vector<std::vector<ocr::geom::Pt>> cs;
    ocr::geom::find_contours(bitmap01, h, w, cs);
    int np = 0, used = 0;
    for (const auto& c : cs) {
      if (np >= max_polys || used + (int)c.size() > max_pts) ocr::fail(OCR_ERR_INVALID, "test buffer too small");
      counts_out[np++] = (int)c.size();
      for (const auto& q : c) {
        xy_out[2 * used] = q.x;
        xy_out[2 * used + 1] = q.y;
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
                      const float* residual, const float* up_residual, int relu, int cat4, int variant, float* out, float* out2) {
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
    // variant: low byte = kernel form (0 conv_igemm, 1 conv3x3_bf16_c64, 2 split-bf16 128-wide tiles, 3 split-bf16 256 x 128 persistent form);
    // variant >> 8 = B > 1: a batched GEMM of B problems (1x1 s1; in [B][n h w][cin], wgt [B][cout][cin], out [B][n h w][cout])
    const int nbatch = variant >> 8 > 1 ? variant >> 8 : 1;
    variant &= 0xff;
    if (nbatch > 1 && (ks != 1 || stride != 1 || cat4 || residual || up_residual || out2)) fail(OCR_ERR_INVALID, "batched test GEMM: 1x1 s1, no residuals");
    in_e *= (size_t)nbatch;
    const size_t w_e = (size_t)nbatch * cout * ks * ks * cin, out_e = (size_t)nbatch * n * ho * wo * cout;
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
    d.batch = nbatch;
    void* d_out = nullptr;
    void* d_out2 = nullptr;
    if (out) { OCR_HIP(hipMalloc(&d_out, out_e * oes)); allocs.push_back(d_out); }
    if (out2) { OCR_HIP(hipMalloc(&d_out2, out_e * oes)); allocs.push_back(d_out2); }
    d.out = d_out;
    d.out2 = d_out2;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    if (variant == 1) {  // conv3x3_bf16_c64.hip instead of conv_igemm
      if (!in_bf16 || !out_bf16 || ks != 3 || stride != 1 || cin != 64 || cout != 64 || cat4 || out2 || !out)
        fail(OCR_ERR_INVALID, "variant 1 is the bf16 3x3 s1 64 -> 64 kernel");
      const std::vector<uint16_t> fr = conv3x3_bf16_c64_fragments(wgt);
      void* d_fr = nullptr;
      OCR_HIP(hipMalloc(&d_fr, fr.size() * 2));
      allocs.push_back(d_fr);
      OCR_HIP(hipMemcpy(d_fr, fr.data(), fr.size() * 2, hipMemcpyHostToDevice));
      launch_conv3x3_bf16_c64(d_in, d_fr, d.scale, d.bias, d.residual, relu, d_out, n, h, w, 256, s);
      OCR_HIP(hipStreamSynchronize(s));
      down(out, d_out, out_e, true);
      return;
    }
    if (variant == 2 || variant == 3) {  // the split-bf16 form of the f32 conv: weights as three bf16 planes; 3: the 256 x 128 persistent kernel
      if (variant == 3 && !conv_x3_wide_applicable([&] { ConvDesc t = d; t.x3 = 1; return t; }())) fail(OCR_ERR_INVALID, "variant 3: the wide split-bf16 form does not take this launch");
      d.wide = variant == 3 ? 1 : 0;
      if (in_bf16 || out_bf16 || cat4) fail(OCR_ERR_INVALID, "variant 2 (split bf16) takes f32 tensors");
      const std::vector<uint16_t> planes = split3_weights_tiled(wgt, w_e, ks * ks * cin);
      void* d_pl = nullptr;
      OCR_HIP(hipMalloc(&d_pl, planes.size() * 2));
      allocs.push_back(d_pl);
      OCR_HIP(hipMemcpy(d_pl, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
      d.x3 = 1;
      d.wgt = d_pl;
      d.wgt_bytes = planes.size() * 2;
    }
    launch_conv_igemm(d, s);
    OCR_HIP(hipStreamSynchronize(s));
    down(out, d_out, out_e, out_bf16);
    down(out2, d_out2, out_e, out_bf16);
  });
}
// one BasicBlock 64 -> 64 of the bf16 precision on caller data (f32 arrays, rounded to bf16 on the way in, widened on the way out):
// fused != 0: basic_block_bf16_c64.hip, one launch (num_cus sizes its persistent grid, 0 = 256); fused == 0: the same block as two
// conv3x3_bf16_c64 launches.  x: NHWC, w1 / w2: [64][9][64]; iters > 1 repeats the block and returns the average time in ms
int ocr_test_bf16_basic_block(ocr_det_t* det, const float* x, int n, int h, int w, const float* w1, const float* scale1, const float* bias1,
                              const float* w2, const float* scale2, const float* bias2, int fused, int num_cus, int iters, float* out, float* ms_out) {
  return guard([&] {
    using namespace ocr;
    if (!det || !x || !w1 || !w2 || !out) fail(OCR_ERR_INVALID, "null argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const size_t e = (size_t)n * h * w * 64;
    auto bf16_bits = [](float f) {
      uint32_t u;
      std::memcpy(&u, &f, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
      return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    std::vector<void*> allocs;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    auto dev = [&](const void* src, size_t bytes) -> void* {
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, bytes));
      allocs.push_back(d);
      if (src) OCR_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
      return d;
    };
    std::vector<uint16_t> xb(e);
    for (size_t i = 0; i < e; ++i) xb[i] = bf16_bits(x[i]);
    void* d_x = dev(xb.data(), e * 2);
    void* d_t = dev(nullptr, e * 2);
    void* d_y = dev(nullptr, e * 2);
    const std::vector<uint16_t> f1 = conv3x3_bf16_c64_fragments(w1), f2 = conv3x3_bf16_c64_fragments(w2);
    void* d_f1 = dev(f1.data(), f1.size() * 2);
    void* d_f2 = dev(f2.data(), f2.size() * 2);
    const float* d_s1 = scale1 ? static_cast<const float*>(dev(scale1, 256)) : nullptr;
    const float* d_b1 = bias1 ? static_cast<const float*>(dev(bias1, 256)) : nullptr;
    const float* d_s2 = scale2 ? static_cast<const float*>(dev(scale2, 256)) : nullptr;
    const float* d_b2 = bias2 ? static_cast<const float*>(dev(bias2, 256)) : nullptr;
    const int cus = num_cus > 0 ? num_cus : 256;
    auto run = [&] {
      if (fused) {
        launch_basic_block_bf16_c64(d_x, d_f1, d_s1, d_b1, d_f2, d_s2, d_b2, d_y, n, h, w, cus, s);
      } else {
        launch_conv3x3_bf16_c64(d_x, d_f1, d_s1, d_b1, nullptr, 1, d_t, n, h, w, cus, s);
        launch_conv3x3_bf16_c64(d_t, d_f2, d_s2, d_b2, d_x, 1, d_y, n, h, w, cus, s);
      }
    };
    run();
    OCR_HIP(hipStreamSynchronize(s));
    if (iters > 1) {
      hipEvent_t e0, e1;
      OCR_HIP(hipEventCreate(&e0));
      OCR_HIP(hipEventCreate(&e1));
      OCR_HIP(hipEventRecord(e0, s));
      for (int i = 0; i < iters; ++i) run();
      OCR_HIP(hipEventRecord(e1, s));
      OCR_HIP(hipEventSynchronize(e1));
      float ms = 0.f;
      OCR_HIP(hipEventElapsedTime(&ms, e0, e1));
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      if (ms_out) *ms_out = ms / iters;
    }
    std::vector<uint16_t> yb(e);
    OCR_HIP(hipMemcpy(yb.data(), d_y, e * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < e; ++i) {
      const uint32_t u = (uint32_t)yb[i] << 16;
      std::memcpy(&out[i], &u, 4);
    }
  });
}
// one 3x3 s1 p1 conv (+ scale / bias / residual / ReLU) through the Winograd F(2x2,3x3) path on caller data:
// weight transform, input transform, batched 16-problem GEMM, output transform.  x: NHWC, wgt: [cout][9][cin].
int ocr_test_winograd_conv(ocr_det_t* det, const float* x, int n, int h, int w, int cin, const float* wgt, int cout,
                           const float* scale, const float* bias, const float* residual, int relu, int unfused_and_cus, float* out) {
  return guard([&] {
    int unfused = unfused_and_cus;
    using namespace ocr;
    if (!det || !x || !wgt || !out) fail(OCR_ERR_INVALID, "null argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const int num_cus = unfused >> 8;   // optional, for the fused kernel's grid
    unfused &= 0xff;
    const size_t wm = unfused >= 3 ? 4 : 2, wa = (wm + 2) * (wm + 2);   // unfused: 1 = F(2x2,3x3), 3 = F(4x4,3x3), both unfused; 4 = the fused F(4x4,3x3) kernel
    const size_t th = (h + wm - 1) / wm, tw = (w + wm - 1) / wm, T = (size_t)n * th * tw;
    const size_t in_e = (size_t)n * h * w * cin, out_e = (size_t)n * h * w * cout;
    const std::vector<float> u = winograd_weights(wgt, cout, cin, (int)wm);
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
    float* d_v = dev(nullptr, wa * T * cin);
    float* d_m = dev(nullptr, wa * T * cout);
    float* d_y = dev(nullptr, out_e);
    const float* d_sc = scale ? dev(scale, cout) : nullptr;
    const float* d_bi = bias ? dev(bias, cout) : nullptr;
    const float* d_res = residual ? dev(residual, out_e) : nullptr;
    if (unfused == 5) {  // fused F(4x4,3x3) on the bf16 matrix cores (split-bf16); one workgroup per CU
      const std::vector<uint16_t> uf = winograd43_x3_fragments(u, cout, cin);
      void* d_uf = nullptr;
      OCR_HIP(hipMalloc(&d_uf, uf.size() * 2));
      allocs.push_back(d_uf);
      OCR_HIP(hipMemcpy(d_uf, uf.data(), uf.size() * 2, hipMemcpyHostToDevice));
      launch_winograd43_x3(d_x, d_uf, d_sc, d_bi, d_res, relu, d_y, n, h, w, cin, cout, num_cus > 0 ? num_cus : 256, s);
      OCR_HIP(hipStreamSynchronize(s));
      OCR_HIP(hipMemcpy(out, d_y, out_e * 4, hipMemcpyDeviceToHost));
      return;
    }
    if (unfused == 4) {  // fused F(4x4,3x3); num_cus sizes its persistent grid (two workgroups per CU)
      float* d_uf = dev(winograd43_fragments(u, cout, cin).data(), u.size());
      launch_winograd43_fused(d_x, d_uf, d_sc, d_bi, d_res, relu, d_y, n, h, w, cin, cout, num_cus > 0 ? num_cus : 256, s);
      OCR_HIP(hipStreamSynchronize(s));
      OCR_HIP(hipMemcpy(out, d_y, out_e * 4, hipMemcpyDeviceToHost));
      return;
    }
    launch_winograd_input(d_x, d_v, n, h, w, cin, (int)wm, s);
    ConvDesc d{};
    d.src[0] = d_v;
    d.src_mode = SRC_PLAIN;
    d.src_bytes = wa * T * cin * 4;
    d.wgt = d_u;
    d.wgt_bytes = u.size() * 4;
    d.batch = (int)wa;
    d.N = 1; d.Hin = d.Ho = 1; d.Win = d.Wo = (int)T; d.Cin = cin; d.Cout = cout;
    d.ks = 1; d.stride = 1; d.pad = 0; d.store_mode = STORE_NHWC; d.out = d_m; d.name = "test_winograd";
    launch_conv_igemm(d, s);
    launch_winograd_output(d_m, d_sc, d_bi, d_res, relu, d_y, n, h, w, cout, (int)wm, s);
    OCR_HIP(hipStreamSynchronize(s));
    OCR_HIP(hipMemcpy(out, d_y, out_e * 4, hipMemcpyDeviceToHost));
  });
}

// Two kernels of complementary families on two streams (tuning aid): A = the fused F(4x4,3x3) kernel (f32 MFMA)
// on n x 160 x 160 x 64 with its persistent grid sized for `w43_cus` CUs (two workgroups each; 128 -> one workgroup per CU), B = one
// split-bf16 conv_igemm launch (which_b: 0 = 3x3 s2 64 -> 128 from 160 x 160, 1 = 3x3 s2 128 -> 256 from 80 x 80, 2 = the 36 batched
// Winograd GEMMs of layer3, 3 = 3x3 s2 256 -> 512 from 40 x 40).  ms_out: [0] A alone (per launch), [1] B alone (per launch), [2] wall
// time of reps_a launches of A on one stream beside reps_b launches of B on another, [3] the same 2 queues on ONE stream.
int ocr_test_dual_stream_bench(ocr_det_t* det, int n, int which_b, int w43_cus, int reps_a, int reps_b, float* ms_out) {
  return guard([&] {
    using namespace ocr;
    if (!det || !ms_out) fail(OCR_ERR_INVALID, "null argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    std::vector<void*> allocs;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    uint32_t st = 4242u;
    auto rnd = [&] { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    auto dev_rand = [&](size_t elems, float amp) -> float* {
      std::vector<float> hbuf(elems);
      for (auto& v : hbuf) v = amp * rnd();
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, elems * 4));
      allocs.push_back(d);
      OCR_HIP(hipMemcpy(d, hbuf.data(), elems * 4, hipMemcpyHostToDevice));
      return static_cast<float*>(d);
    };
    auto dev_bytes = [&](const void* src, size_t bytes) -> void* {
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, bytes));
      allocs.push_back(d);
      if (src) OCR_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
      return d;
    };
    // A
    const int ha = 160, wa = 160, ca = 64;
    float* ax = dev_rand((size_t)n * ha * wa * ca, 1.0f);
    float* ay = static_cast<float*>(dev_bytes(nullptr, (size_t)n * ha * wa * ca * 4));
    std::vector<float> wg((size_t)ca * 9 * ca);
    for (auto& v : wg) v = 0.05f * rnd();
    const std::vector<float> ufr = winograd43_fragments(winograd_weights(wg.data(), ca, ca, 4), ca, ca);
    float* auf = static_cast<float*>(dev_bytes(ufr.data(), ufr.size() * 4));
    // B
    ConvDesc d{};
    d.x3 = 1;
    d.src_mode = SRC_PLAIN;
    d.store_mode = STORE_NHWC;
    d.relu = 1;
    d.name = "dual";
    size_t in_e, w_e, out_e;
    if (which_b == 2) {
      const size_t T = (size_t)n * 10 * 10;
      in_e = 36 * T * 256; w_e = (size_t)36 * 256 * 256; out_e = 36 * T * 256;
      d.batch = 36; d.N = 1; d.Hin = d.Ho = 1; d.Win = d.Wo = (int)T; d.Cin = 256; d.Cout = 256; d.ks = 1; d.stride = 1; d.pad = 0;
    } else {
      const int hb = which_b == 0 ? 160 : which_b == 1 ? 80 : 40, cb = which_b == 0 ? 64 : which_b == 1 ? 128 : 256;
      in_e = (size_t)n * hb * hb * cb; w_e = (size_t)2 * cb * 9 * cb; out_e = (size_t)n * (hb / 2) * (hb / 2) * 2 * cb;
      d.N = n; d.Hin = d.Win = hb; d.Ho = d.Wo = hb / 2; d.Cin = cb; d.Cout = 2 * cb; d.ks = 3; d.stride = 2; d.pad = 1;
    }
    d.src[0] = dev_rand(in_e, 1.0f);
    d.src_bytes = in_e * 4;
    std::vector<float> hw(w_e);
    for (auto& v : hw) v = 0.05f * rnd();
    const std::vector<uint16_t> planes = split3_weights_tiled(hw.data(), w_e, d.ks * d.ks * d.Cin);
    d.wgt = dev_bytes(planes.data(), planes.size() * 2);
    d.wgt_bytes = w_e * 6;
    d.out = dev_bytes(nullptr, out_e * 4);
    hipStream_t sa, sb;
    OCR_HIP(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    OCR_HIP(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2;
    OCR_HIP(hipEventCreate(&e0));
    OCR_HIP(hipEventCreate(&e1));
    OCR_HIP(hipEventCreate(&e2));
    auto run_a = [&](hipStream_t s) { launch_winograd43_fused(ax, auf, nullptr, nullptr, nullptr, 1, ay, n, ha, wa, ca, ca, w43_cus, s); };
    auto run_b = [&](hipStream_t s) { launch_conv_igemm(d, s); };
    float ms = 0.f;
    for (int i = 0; i < 3; ++i) { run_a(sa); run_b(sa); }
    OCR_HIP(hipEventRecord(e0, sa));
    for (int i = 0; i < reps_a; ++i) run_a(sa);
    OCR_HIP(hipEventRecord(e1, sa));
    for (int i = 0; i < reps_b; ++i) run_b(sa);
    OCR_HIP(hipEventRecord(e2, sa));
    OCR_HIP(hipStreamSynchronize(sa));
    OCR_HIP(hipEventElapsedTime(&ms, e0, e1));
    ms_out[0] = ms / reps_a;
    OCR_HIP(hipEventElapsedTime(&ms, e1, e2));
    ms_out[1] = ms / reps_b;
    OCR_HIP(hipEventElapsedTime(&ms, e0, e2));
    ms_out[3] = ms;
    // together: both streams start behind e0, the end is when both have drained
    OCR_HIP(hipEventRecord(e0, sa));
    OCR_HIP(hipStreamWaitEvent(sb, e0, 0));
    for (int i = 0; i < std::max(reps_a, reps_b); ++i) {   // interleaved submission
      if (i < reps_a) run_a(sa);
      if (i < reps_b) run_b(sb);
    }
    OCR_HIP(hipEventRecord(e1, sb));
    OCR_HIP(hipStreamWaitEvent(sa, e1, 0));
    OCR_HIP(hipEventRecord(e2, sa));
    OCR_HIP(hipStreamSynchronize(sa));
    OCR_HIP(hipEventElapsedTime(&ms, e0, e2));
    ms_out[2] = ms;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    (void)hipStreamDestroy(sa); (void)hipStreamDestroy(sb);
  });
}

// unclip.hip against postproc_geom.cpp::finish_polygon on caller polygons (one image, given scores): every polygon the device settles
// (UNCLIP_KEEP / UNCLIP_DROP) must be what the host does with it; counts how many it hands back (UNCLIP_HOST).  stats: keep, host, drop, mismatches.
int ocr_test_unclip_compare(ocr_det_t* det, const int32_t* xy, const int32_t* counts, int n_polys, const double* scores, double adj_x,
                            double adj_y, double box_thresh, double unclip_ratio, double min_size, int32_t* stats, int32_t* status_out,
                            int32_t* len_out, uint32_t* xy_out) {   // the three optional: per polygon status, length, and 3 x its input points of (x, y) room
  return guard([&] {
    using namespace ocr;
    if (!det || !xy || !counts || !scores || !stats || n_polys <= 0) fail(OCR_ERR_INVALID, "bad argument");
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    std::vector<BoxScoreJob> jobs(n_polys);
    std::vector<double> sums(scores, scores + n_polys), cnts(n_polys, 1.0);
    size_t npts = 0;
    for (int k = 0; k < n_polys; ++k) {
      jobs[k] = BoxScoreJob{0, (int)npts, counts[k], 0, 0, 1, 1};
      npts += (size_t)counts[k];
    }
    std::vector<void*> allocs;
    struct Free { std::vector<void*>& v; ~Free() { for (void* p : v) (void)hipFree(p); } } free_all{allocs};
    auto dev = [&](const void* src, size_t bytes) -> void* {
      void* d = nullptr;
      OCR_HIP(hipMalloc(&d, std::max<size_t>(bytes, 16)));
      allocs.push_back(d);
      if (src) OCR_HIP(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
      return d;
    };
    const double adj[2] = {adj_x, adj_y};
    auto* d_jobs = static_cast<BoxScoreJob*>(dev(jobs.data(), jobs.size() * sizeof(BoxScoreJob)));
    auto* d_pts = static_cast<int32_t*>(dev(xy, npts * 8));
    auto* d_sum = static_cast<double*>(dev(sums.data(), sums.size() * 8));
    auto* d_cnt = static_cast<double*>(dev(cnts.data(), cnts.size() * 8));
    auto* d_adj = static_cast<double*>(dev(adj, 16));
    void* d_work = dev(nullptr, unclip_work_bytes(npts, n_polys));
    auto* d_oxy = static_cast<uint32_t*>(dev(nullptr, 3 * npts * 8));
    auto* d_len = static_cast<int32_t*>(dev(nullptr, (size_t)n_polys * 4));
    auto* d_st = static_cast<int32_t*>(dev(nullptr, (size_t)n_polys * 4));
    const UnclipParams up{box_thresh, unclip_ratio, min_size};
    launch_unclip(d_jobs, d_pts, nullptr, n_polys, npts, d_sum, d_cnt, d_adj, up, d_work, d_oxy, d_len, d_st, s);
    OCR_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> st(n_polys), len(n_polys);
    std::vector<uint32_t> oxy(3 * npts * 2);
    OCR_HIP(hipMemcpy(st.data(), d_st, st.size() * 4, hipMemcpyDeviceToHost));
    OCR_HIP(hipMemcpy(len.data(), d_len, len.size() * 4, hipMemcpyDeviceToHost));
    OCR_HIP(hipMemcpy(oxy.data(), d_oxy, oxy.size() * 4, hipMemcpyDeviceToHost));
    ocr_postproc_params_t prm{};
    ocr_postproc_default_params(&prm);
    prm.box_thresh = box_thresh;
    prm.unclip_ratio = unclip_ratio;
    prm.min_size = min_size;
    prm.skip_degenerate = 1;
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (status_out) std::copy(st.begin(), st.end(), status_out);
    if (len_out) std::copy(len.begin(), len.end(), len_out);
    if (xy_out) std::copy(oxy.begin(), oxy.end(), xy_out);
    for (int k = 0; k < n_polys; ++k) {
      std::vector<geom::Pt> c((size_t)counts[k]);
      for (int i = 0; i < counts[k]; ++i) c[i] = {xy[2 * (jobs[k].pt_offset + i)], xy[2 * (jobs[k].pt_offset + i) + 1]};
      std::vector<uint32_t> hxy;
      const bool kept = geom::finish_polygon(c, scores[k], adj_x, adj_y, prm, hxy);
      if (st[k] == UNCLIP_KEEP) {
        ++stats[0];
        const uint32_t* o = oxy.data() + 6 * (size_t)jobs[k].pt_offset;
        if (!kept || hxy.size() != 2 * (size_t)len[k] || !std::equal(hxy.begin(), hxy.end(), o)) ++stats[3];
      } else if (st[k] == UNCLIP_HOST) {
        ++stats[1];
      } else {
        ++stats[2];
        if (kept || !(box_thresh > scores[k])) ++stats[3];
      }
    }
  });
}
// no GPU: the device code's restatement of glibc's hypot against std::hypot on all integer pairs up to `limit`
long long ocr_test_hypot_port_mismatches(int limit) { return ocr::geom::hypot_port_mismatches(limit); }
int ocr_test_set_conv_debug(int d) {
  ocr::set_conv_debug(d);
  return OCR_OK;
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
    const int x3 = (src_mode & 64) ? 1 : 0;  // split-bf16 form: the weight buffer is reinterpreted as three bf16 planes
    src_mode &= ~64;
    const int wide = (src_mode & 128) ? 1 : 0;  // ... on the 256 x 128 persistent form (conv_x3w.hip) where it takes the launch
    src_mode &= ~128;
    if (x3) {
      (void)hipFree(wt);
      OCR_HIP(hipMalloc(reinterpret_cast<void**>(&wt), w_e * 6));
      std::vector<float> hw(w_e);
      uint32_t st = 777u;
      for (size_t i = 0; i < w_e; ++i) { st = st * 1664525u + 1013904223u; hw[i] = 0.05f * (((st >> 8) & 0xffff) / 32768.0f - 1.0f); }
      const std::vector<uint16_t> planes = split3_weights_tiled(hw.data(), w_e, ks * ks * cin);
      OCR_HIP(hipMemcpy(wt, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
    }
    ConvDesc d{};
    d.x3 = x3;
    d.wide = wide;
    d.src[0] = in;
    d.src_mode = SRC_PLAIN;
    const int bf = src_mode == 16 ? 1 : 0;  // src_mode 16: bf16 operands and output (the buffers are just reinterpreted)
    d.in_bf16 = d.out_bf16 = bf;
    d.src_bytes = in_e * (bf ? 2 : 4);
    d.wgt_bytes = w_e * (x3 ? 6 : bf ? 2 : 4); d.N = n; d.Hin = h; d.Win = w; d.Cin = cin; d.Ho = ho; d.Wo = wo; d.Cout = cout;
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

// the result all-gather without RCCL: packs every shard as its rank would and assembles them as the receiver does
int ocr_test_comm_assemble(const ocr_polygons_t* const* shards, int world, ocr_polygons_t** all) {
  return guard([&] {
    if (!shards || !all || world < 1) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<std::vector<uint8_t>> packed;
    std::vector<const uint8_t*> ptr;
    std::vector<size_t> len;
    for (int r = 0; r < world; ++r) {
      packed.push_back(ocr::pack_shard(*shards[r]));
      ptr.push_back(packed.back().data());
      len.push_back(packed.back().size());
    }
    auto res = std::make_unique<ocr::PolygonsOwned>();
    ocr::assemble_shards(ptr.data(), len.data(), world, *res);
    *all = &res.release()->view;
  });
}

#ifdef W43_STAMPS
int ocr_test_w43_stamps(long long* out) { ocr::winograd43_read_stamps(out); return 0; }
int ocr_test_w43x_stamps(long long* out) { ocr::winograd43_x3_read_stamps(out); return 0; }
#endif
int ocr_test_w43_debug(int d) { ocr::winograd43_set_debug(d); return 0; }
#ifdef WS_STAMPS
#endif

}  // extern "C"
