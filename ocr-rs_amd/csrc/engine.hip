// Host-side schedule of the detection / recognition graphs.
//   /root/reference/src/text_detection/model.rs:65-152   (resnet18, eval mode)
//   /root/reference/src/char_recognition/model.rs:13-39
// Weights are re-laid out once at create time (OIHW -> OHWI, eval batch norm folded
// to per-channel scale/bias); activations live in NHWC f32 workspaces in HBM.
#include <sched.h>

#include "engine.hpp"

#include <thread>

#include "thread_pool.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace ocr {

void check_device(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    fail(OCR_ERR_NOGPU, "no HIP device visible: this library has no CPU fallback");
  if (device < 0 || device >= count) fail(OCR_ERR_INVALID, "device %d out of range (%d visible)", device, count);
  hipDeviceProp_t prop;
  OCR_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    fail(OCR_ERR_NOGPU, "device %d is %s; the kernels are built for gfx950 (MI355X) only", device, prop.gcnArchName);
  OCR_HIP(hipSetDevice(device));
}

DeviceArena::~DeviceArena() {
  if (base_) (void)hipFree(base_);
}
void DeviceArena::reserve(size_t bytes) {
  OCR_HIP(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
  cap_ = bytes;
  used_ = 0;
}
float* DeviceArena::upload(const std::vector<float>& host) {
  const size_t bytes = (host.size() * sizeof(float) + 255) / 256 * 256;
  if (used_ + bytes > cap_) fail(OCR_ERR_INTERNAL, "weight arena overflow");
  float* p = reinterpret_cast<float*>(base_ + used_);
  OCR_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  used_ += bytes;
  return p;
}

void* DeviceArena::upload_u16(const std::vector<uint16_t>& host) {
  const size_t bytes = (host.size() * 2 + 255) / 256 * 256;
  if (used_ + bytes > cap_) fail(OCR_ERR_INTERNAL, "weight arena overflow");
  void* p = base_ + used_;
  OCR_HIP(hipMemcpy(p, host.data(), host.size() * 2, hipMemcpyHostToDevice));
  used_ += bytes;
  return p;
}

// eval batch norm (tch batch_norm2d default eps 1e-5) -> y = x*scale + bias
static void fold_bn(const WeightBlob& wb, const std::string& p, int c, std::vector<float>& scale, std::vector<float>& bias) {
  const float* g = wb.get(p + ".weight", {c}).data;
  const float* b = wb.get(p + ".bias", {c}).data;
  const float* m = wb.get(p + ".running_mean", {c}).data;
  const float* v = wb.get(p + ".running_var", {c}).data;
  scale.resize(c);
  bias.resize(c);
  for (int i = 0; i < c; ++i) {
    const float invstd = 1.0f / std::sqrt(v[i] + 1e-5f);
    scale[i] = g[i] * invstd;
    bias[i] = b[i] - m[i] * scale[i];
  }
}

ConvW Detector::make_conv(const WeightBlob& wb, const std::string& wname, const std::string& bn, int cout, int cin, int ks) {
  const float* w = wb.get(wname, {cout, cin, ks, ks}).data;
  std::vector<float> t((size_t)cout * ks * ks * cin);
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c)
      for (int k = 0; k < ks * ks; ++k) t[((size_t)o * ks * ks + k) * cin + c] = w[((size_t)o * cin + c) * ks * ks + k];
  ConvW cw;
  cw.w = arena_.upload(t);
  cw.w_bytes = t.size() * sizeof(float);
  cw.host = std::move(t);
  cw.cin = cin;
  cw.cout = cout;
  cw.ks = ks;
  if (!bn.empty()) {
    std::vector<float> s, b;
    fold_bn(wb, bn, cout, s, b);
    cw.scale = arena_.upload(s);
    cw.bias = arena_.upload(b);
    cw.host_scale = s;
  }
  return cw;
}

// out (3x3, 256 -> 64, OHWI) after in (1x1, cin -> 256): T[o][tap][i] = sum_c out[o][tap][c] * in[c][i], in f64
static std::vector<double> compose_taps(const ConvW& out, const ConvW& in) {
  const int cin = in.cin, mid = out.cin;
  std::vector<double> t((size_t)out.cout * 9 * cin, 0.0);
  for (int o = 0; o < out.cout; ++o)
    for (int k = 0; k < 9; ++k) {
      double* row = &t[((size_t)o * 9 + k) * cin];
      const float* ow = &out.host[((size_t)o * 9 + k) * mid];
      for (int c = 0; c < mid; ++c) {
        const double a = ow[c];
        const float* iw = &in.host[(size_t)c * cin];
        for (int i = 0; i < cin; ++i) row[i] += a * (double)iw[i];
      }
    }
  return t;
}

ConvW Detector::finish_composed(std::vector<float>&& t, int cout, int cin, int ks) {
  ConvW cw;
  cw.w = arena_.upload(t);
  cw.w_bytes = t.size() * sizeof(float);
  cw.host = std::move(t);
  cw.cin = cin;
  cw.cout = cout;
  cw.ks = ks;
  return cw;
}

// U = G g G^T per (cout, cin) in f64, rounded once, laid out [(m+2)^2 = (m+2) i + j][Cout][Cin]: that many 1x1-conv weight
// matrices for conv_igemm's batched mode.  F(2x2,3x3): G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]];
// F(4x4,3x3): G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
std::vector<float> winograd_weights(const float* ohwi, int cout, int cin, int m) {
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  if (m != 2 && m != 4) fail(OCR_ERR_INTERNAL, "winograd_weights: m = %d", m);
  const int a = m + 2;
  const double(*G)[3] = m == 2 ? G2 : G4;
  const size_t kc = (size_t)cout * cin;
  std::vector<float> u((size_t)a * a * kc);
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < cin; ++c) {
      double g[3][3];
      for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = ohwi[((size_t)o * 9 + t) * cin + c];
      for (int i = 0; i < a; ++i)
        for (int j = 0; j < a; ++j) {
          double acc = 0.0;
          for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) acc += G[i][p] * g[p][q] * G[j][q];
          u[(size_t)(a * i + j) * kc + (size_t)o * cin + c] = (float)acc;
        }
    }
  return u;
}

void Detector::add_winograd_weights(ConvW& cw) {
  cw.wino_tile = cw.cin >= winograd43_min_cin_ ? 4 : 2;
  const std::vector<float> u = winograd_weights(cw.host.data(), cw.cout, cw.cin, cw.wino_tile);
  cw.wino = arena_.upload(u);
  cw.wino_bytes = u.size() * sizeof(float);
  if (split_bf16_ && cw.cin % 32 == 0) cw.wino_x3 = arena_.upload_u16(split3_weights_tiled(u.data(), u.size(), cw.cin));   // [component][Cout][Cin]
}

void Detector::add_winograd_fused_weights(ConvW& cw) {
  if ((cw.cin != 64 && cw.cin != 128 && cw.cin != 256) || cw.cout % 64 || cw.ks != 3) fail(OCR_ERR_INTERNAL, "fused Winograd: unsupported conv shape");
  const std::vector<float> u = winograd_weights(cw.host.data(), cw.cout, cw.cin, 4);
  cw.wino43_fused = arena_.upload(winograd43_fragments(u, cw.cout, cw.cin));
  if (split_bf16_ && winograd43_x3_) cw.wino43_x3 = arena_.upload_u16(winograd43_x3_fragments(u, cw.cout, cw.cin));
}

// hi / mid / lo bf16 planes of a conv's f32 weights (and of its Winograd form): what conv_igemm's split-bf16 kernels read
void Detector::add_split_weights(ConvW& cw, int wrow) {
  if (wrow == 0) wrow = cw.ks * cw.ks * cw.cin;   // weights per output channel (phase convs: 2 x 2 taps; bin_conv1 over the pyramid: 21 tap slots)
  if (!cw.host.empty() && !cw.w_x3 && cw.cin % 32 == 0) cw.w_x3 = arena_.upload_u16(split3_weights_tiled(cw.host.data(), cw.host.size(), wrow));
}

// A_k = out_k o in_k
ConvW Detector::compose_lateral(const ConvW& out, const ConvW& in) {
  const std::vector<double> t = compose_taps(out, in);
  return finish_composed(std::vector<float>(t.begin(), t.end()), out.cout, in.cin, 3);
}

// A 3x3 conv (taps [cout][9][cin], f64) of the nearest-x-up upsample of a tensor, seen from the low-res
// grid.  High-res row up i + a with tap dy reads high-res row up i + a + dy - 1, i.e. low-res row
//   a = 0        : dy 0 -> i-1 (tap 0);  dy 1, 2 -> i (tap 1)            window starts at i-1
//   0 < a < up-1 : dy 0, 1, 2 -> i (tap 0); tap 1 is empty                window starts at i
//   a = up-1     : dy 0, 1 -> i (tap 0);  dy 2 -> i+1 (tap 1)             window starts at i
// (same for columns), and zero padding of the high-res tensor is zero padding of the low-res one.
// Layout [phase = up a + b][cout][2x2][cin].
ConvW Detector::phase_conv(const std::vector<double>& t, int cout, int cin, int up) {
  std::vector<float> w((size_t)up * up * cout * 4 * cin);
  auto tap_of = [up](int a, int d) { return a == 0 ? (d == 0 ? 0 : 1) : (a == up - 1 && d == 2) ? 1 : 0; };
  // phases strictly inside a row / column of phases use one tap in that direction; the kernel walks the
  // active taps as the prefix of the order t = kh * nw + kw (nw = taps per row of this phase) and skips the rest
  auto taps_of = [up](int a) { return (a == 0 || a == up - 1) ? 2 : 1; };
  std::vector<double> acc((size_t)4 * cin);
  for (int a = 0; a < up; ++a)
    for (int b = 0; b < up; ++b)
      for (int o = 0; o < cout; ++o) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx) {
            const int tp = tap_of(a, dy) * taps_of(b) + tap_of(b, dx);
            const double* src = &t[((size_t)o * 9 + dy * 3 + dx) * cin];
            for (int i = 0; i < cin; ++i) acc[(size_t)tp * cin + i] += src[i];
          }
        float* dst = &w[(((size_t)(a * up + b) * cout + o) * 4) * cin];
        for (size_t i = 0; i < acc.size(); ++i) dst[i] = (float)acc[i];
      }
  ConvW cw = finish_composed(std::move(w), cout, cin, 2);
  cw.up = up;
  return cw;
}

// B_k = out_k o up2 o in_{k+1}
ConvW Detector::compose_upsampled(const ConvW& out, const ConvW& in_up) {
  return phase_conv(compose_taps(out, in_up), out.cout, in_up.cin, 2);
}

// "key=value;key=value" -> engine options (include/ocr_amd.h, ocr_det_create_with_options).  The environment is
// never consulted: which schedule runs is the caller's explicit choice.
void Detector::parse_options(const char* options) {
  if (!options) return;
  std::string s(options);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find_first_of(";,", pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    pos = end + 1;
    if (item.find_first_not_of(" \t") == std::string::npos) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) fail(OCR_ERR_INVALID, "detector option '%s': expected key=value", item.c_str());
    auto trim = [](std::string v) {
      const size_t a = v.find_first_not_of(" \t"), b = v.find_last_not_of(" \t");
      return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
    };
    const std::string key = trim(item.substr(0, eq)), val = trim(item.substr(eq + 1));
    auto num = [&]() {
      char* endp = nullptr;
      const long v = std::strtol(val.c_str(), &endp, 10);
      if (val.empty() || *endp) fail(OCR_ERR_INVALID, "detector option %s: '%s' is not an integer", key.c_str(), val.c_str());
      return (int)v;
    };
    if (key == "winograd_fused") winograd_fused_ = num() != 0;
    else if (key == "winograd43_x3") winograd43_x3_ = num() != 0;
    else if (key == "out4_fused") out4_fused_ = num() != 0;
    else if (key == "winograd") winograd_min_cin_ = num() > 0 ? num() : (1 << 30);
    else if (key == "winograd43") winograd43_min_cin_ = num() > 0 ? num() : (1 << 30);
    else if (key == "fpn_unfused") fpn_composed_ = num() == 0;
    else if (key == "bin_pyr") bin_pyr_on_ = num() != 0;
    else if (key == "pyr_p2_direct") pyr_p2_direct_ = num() != 0;
    else if (key == "pyr_grouped") pyr_grouped_ = num() != 0;
    else if (key == "phase_windows") phase_windows_ = num() != 0;
    else if (key == "x3_wide") x3_wide_ = num() != 0;
    else if (key == "bf16_block_fuse") bf16_block_fuse_ = num() != 0;
    else if (key == "tail_unfused") fused_tail_ = num() == 0;
    else if (key == "overlap") {
      overlap_ = num();
      if (overlap_ < 0 || overlap_ > 3) fail(OCR_ERR_INVALID, "detector option overlap: %d (0, 1, 2 or 3)", overlap_);
    }
    else if (key == "w43_cus") {
      w43_cus_ = num();
      if (w43_cus_ < 0 || w43_cus_ > 4096) fail(OCR_ERR_INVALID, "detector option w43_cus: %d", w43_cus_);
    }
    else if (key == "w43_side_cus") {
      w43_side_cus_ = num();
      if (w43_side_cus_ < 0 || w43_side_cus_ > 4096) fail(OCR_ERR_INVALID, "detector option w43_side_cus: %d", w43_side_cus_);
    }
    else if (key == "post_threads") {
      post_threads_ = num();
      if (post_threads_ < 0 || post_threads_ > 256) fail(OCR_ERR_INVALID, "detector option post_threads: %d (0 = automatic, at most 256)", post_threads_);
    }
    else if (key == "device_contours") {
      if (val == "auto") device_contours_ = -1;
      else {
        device_contours_ = num();
        if (device_contours_ < 0 || device_contours_ > 2) fail(OCR_ERR_INVALID, "detector option device_contours: %d (auto, 0, 1 or 2)", device_contours_);
      }
    }
    else if (key == "post_priority") post_priority_ = num() != 0;
    else if (key == "head_cus_yield") {
      head_cus_yield_ = num();
      if (head_cus_yield_ < 0 || head_cus_yield_ > 4) fail(OCR_ERR_INVALID, "detector option head_cus_yield: %d (0 .. 4)", head_cus_yield_);
    }
    else if (key == "transform_fuse") transform_fuse_ = num() != 0;
    else if (key == "device_unclip") {
      device_unclip_ = num();
      if (device_unclip_ < 0 || device_unclip_ > 2) fail(OCR_ERR_INVALID, "detector option device_unclip: %d (0, 1 or 2)", device_unclip_);
    }
    else if (key == "device_polygons") device_polygons_ = num() != 0;
    else if (key == "mfma") {
      if (val == "split_bf16") split_bf16_ = true;
      else if (val == "f32") split_bf16_ = false;
      else fail(OCR_ERR_INVALID, "detector option mfma: '%s' (split_bf16 or f32)", val.c_str());
    }
    else if (key == "precision") {
      if (val == "bf16") opt_bf16_ = true;
      else if (val == "f32") opt_bf16_ = false;
      else fail(OCR_ERR_INVALID, "detector option precision: '%s' (f32 or bf16)", val.c_str());
    } else fail(OCR_ERR_INVALID, "unknown detector option '%s'", key.c_str());
  }
}

Detector::Detector(const void* blob, size_t bytes, int device, const char* options) : device_(device) {
  check_device(device);
  parse_options(options);
  WeightBlob wb(blob, bytes);
  OCR_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  stream_ = own_stream_;
  {
    hipDeviceProp_t prop;
    OCR_HIP(hipGetDeviceProperties(&prop, device));
    num_cus_ = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (overlap_) {
      OCR_HIP(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
      OCR_HIP(hipEventCreateWithFlags(&ev_x1_, hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&ev_x2_, hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&ev_x3_, hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&ev_side_, hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
  }
  // 12.2 M parameters = 48.7 MB f32, the Winograd forms (F(4x4): 36 matrices per 3x3 conv of layer3/4, 113 MB; fused F(2x2): 16),
  // composed FPN weights, 24.4 MB of bf16 copies and fragments on demand, padding
  // ... and with mfma=split_bf16 the bf16 planes (6 bytes per weight) of the convs that run that way: 170 MB for layer4's
  // Winograd matrices, 57 MB for layer3's, 33 MB for bin_conv1's phase weights
  arena_.reserve((size_t)(split_bf16_ ? 800 : 384) << 20);

  {  // conv1 [64,1,7,7] -> [49][64]; bn1
    const float* w = wb.get("conv1.weight", {64, 1, 7, 7}).data;
    std::vector<float> t(49 * 64);
    for (int c = 0; c < 64; ++c)
      for (int k = 0; k < 49; ++k) t[k * 64 + c] = w[c * 49 + k];
    stem_w_ = arena_.upload(t);
    stem_w_host_.assign(w, w + 64 * 49);
    std::vector<float> s, b;
    fold_bn(wb, "bn1", 64, s, b);
    stem_scale_ = arena_.upload(s);
    stem_bias_ = arena_.upload(b);
    if (split_bf16_) stem_wx3_ = arena_.upload_u16(stem_x3_fragments(w));
  }
  int cin = 64;
  for (int l = 0; l < 4; ++l) {
    const int cout = 64 << l;
    for (int b = 0; b < 2; ++b) {
      const std::string p = "layer" + std::to_string(l + 1) + "." + std::to_string(b);
      layer_[l][b][0] = make_conv(wb, p + ".conv1.weight", p + ".bn1", cout, b == 0 ? cin : cout, 3);
      layer_[l][b][1] = make_conv(wb, p + ".conv2.weight", p + ".bn2", cout, cout, 3);
    }
    if (l > 0) {
      const std::string p = "layer" + std::to_string(l + 1) + ".0.downsample";
      down_[l] = make_conv(wb, p + ".0.weight", p + ".1", cout, cin, 1);
    }
    cin = cout;
  }
  for (int l = 0; l < 4; ++l) {
    in_[l] = make_conv(wb, "in" + std::to_string(l + 2) + ".weight", "", 256, 64 << l, 1);
    out_[l] = make_conv(wb, "out" + std::to_string(l + 2) + ".weight", "", 64, 256, 3);
  }
  if (winograd_fused_) {
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 2; ++c) add_winograd_fused_weights(layer_[0][b][c]);
    // layer2's stride-1 convs (128 -> 128 at H/8)
    add_winograd_fused_weights(layer_[1][0][1]);
    add_winograd_fused_weights(layer_[1][1][0]);
    add_winograd_fused_weights(layer_[1][1][1]);
    // layer3 (256 -> 256 at H/16) and layer4 (20 x 20 grids) stay on the unfused F(4x4,3x3) path below: their 36-component
    // tensors are small enough to live in L2 / Infinity Cache and the GEMMs run as large split-bf16 tiles
  }
  for (int l = 0; l < 4; ++l) {
    if ((64 << l) < winograd_min_cin_) continue;
    add_winograd_weights(layer_[l][0][1]);
    add_winograd_weights(layer_[l][1][0]);
    add_winograd_weights(layer_[l][1][1]);
  }
  if (fpn_composed_)
    for (int l = 0; l < 2; ++l) {
      fpn_a_[l] = compose_lateral(out_[l], in_[l]);
      fpn_b_[l] = compose_upsampled(out_[l], in_[l + 1]);
    }
  if (fpn_composed_ && winograd_fused_) {
    add_winograd_fused_weights(fpn_a_[0]);  // p2 lateral term, 64 -> 64 at H/4
    add_winograd_fused_weights(fpn_a_[1]);  // p3 lateral term, 128 -> 64 at H/8
    if (out4_fused_) add_winograd_fused_weights(out_[2]);    // out4, 256 -> 64 at H/16
    else add_winograd_weights(out_[2]);                      // ... or through the unfused F(4x4,3x3) path of layer3 / layer4
    if (!out4_fused_) add_winograd_weights(out_[3]);         // out5 (256 -> 64 at H/32) likewise (direct conv: 100 tiles of K = 2304)
  }
  bin1_ = make_conv(wb, "bin_conv1.weight", "bin_bn1", 64, 256, 3);
  if (fpn_composed_) {
    // slice s of the 256 input channels (p5, p4, p3, p2 = s 0..3), bin_bn1's scale folded in
    auto slice = [&](int sidx) {
      std::vector<double> t((size_t)64 * 9 * 64);
      for (int o = 0; o < 64; ++o)
        for (int k = 0; k < 9; ++k)
          for (int c = 0; c < 64; ++c)
            t[((size_t)o * 9 + k) * 64 + c] = (double)bin1_.host_scale[o] * (double)bin1_.host[((size_t)o * 9 + k) * 256 + sidx * 64 + c];
      return t;
    };
    for (int l = 0; l < 3; ++l) bin_up_[l] = phase_conv(slice(2 - l), 64, 64, 2 << l);
    const std::vector<double> t2 = slice(3);
    bin_p2_ = finish_composed(std::vector<float>(t2.begin(), t2.end()), 64, 64, 3);
    bin_up_[2].bias = bin1_.bias;  // the p5 term is accumulated last: it adds the bias and applies the ReLU
    if (bin_pyr_on_) {
      // [phase = 8 a + b][cout][slot][64]: slots 4 s + (th * nw + tw) for the upsampled sources s = 0 (p5, up 8),
      // 1 (p4, up 4), 2 (p3, up 2) with the tap merging of phase_conv(), slots 12 + 3 dy + dx for p2
      std::vector<float> w((size_t)64 * 64 * 21 * 64, 0.f);
      std::vector<double> acc((size_t)4 * 64);
      for (int sidx = 0; sidx < 4; ++sidx) {
        const std::vector<double> t = slice(sidx);  // [cout][9][64], bin_bn1 scale folded
        const int up = 8 >> sidx;
        auto tap_of = [up](int a, int d) { return a == 0 ? (d == 0 ? 0 : 1) : (a == up - 1 && d == 2) ? 1 : 0; };
        auto taps_of = [up](int a) { return (a == 0 || a == up - 1) ? 2 : 1; };
        for (int a = 0; a < 8; ++a)
          for (int b = 0; b < 8; ++b)
            for (int o = 0; o < 64; ++o) {
              float* dst = &w[(((size_t)(a * 8 + b) * 64 + o) * 21 + (sidx < 3 ? 4 * sidx : 12)) * 64];
              if (sidx == 3) {
                for (int k = 0; k < 9; ++k)
                  for (int c = 0; c < 64; ++c) dst[k * 64 + c] = (float)t[((size_t)o * 9 + k) * 64 + c];
                continue;
              }
              const int as = a & (up - 1), bs = b & (up - 1);
              std::fill(acc.begin(), acc.end(), 0.0);
              for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx) {
                  const int tp = tap_of(as, dy) * taps_of(bs) + tap_of(bs, dx);
                  const double* src = &t[((size_t)o * 9 + dy * 3 + dx) * 64];
                  for (int c = 0; c < 64; ++c) acc[(size_t)tp * 64 + c] += src[c];
                }
              for (size_t i = 0; i < acc.size(); ++i) dst[i] = (float)acc[i];
            }
      }
      bin_pyr_ = finish_composed(std::move(w), 64, 64, 3);
      bin_pyr_.up = 8;
      bin_pyr_.bias = bin1_.bias;
      if (winograd_fused_) {
        // p2's 3x3 term leaves the phase launch and runs as a fused Winograd conv on top of it (bias + ReLU there)
        add_winograd_fused_weights(bin_p2_);
        bin_p2_.bias = bin1_.bias;
      }
    }
  }
  {  // bin_conv_tr1 [Cin=64][Cout=64][2][2] + bias, then bin_bn2:
     // GEMM B rows = (a*2+b)*64 + co over K = ci; (acc + bias)*s + t = acc*s + (bias*s + t)
    const float* w = wb.get("bin_conv_tr1.weight", {64, 64, 2, 2}).data;
    const float* bias = wb.get("bin_conv_tr1.bias", {64}).data;
    std::vector<float> t(256 * 64), s, b, s4(256), b4(256);
    for (int ci = 0; ci < 64; ++ci)
      for (int co = 0; co < 64; ++co)
        for (int k = 0; k < 4; ++k) t[((size_t)k * 64 + co) * 64 + ci] = w[((size_t)ci * 64 + co) * 4 + k];
    fold_bn(wb, "bin_bn2", 64, s, b);
    for (int k = 0; k < 4; ++k)
      for (int co = 0; co < 64; ++co) {
        s4[k * 64 + co] = s[co];
        b4[k * 64 + co] = bias[co] * s[co] + b[co];
      }
    tr1_.w = arena_.upload(t);
    tr1_.w_bytes = t.size() * sizeof(float);
    tr1_.host = t;  // kept for the bf16 copy (set_precision)
    tr1_.scale = arena_.upload(s4);
    tr1_.bias = arena_.upload(b4);
    tr1_.cin = 64;
    tr1_.cout = 256;
    tr1_.ks = 1;
    if (split_bf16_) tr1_.w_x3 = arena_.upload_u16(split3_weights(tr1_.host.data(), tr1_.host.size()));
  }
  {  // bin_conv_tr2 [64][1][2][2] -> [4][64]
    const float* w = wb.get("bin_conv_tr2.weight", {64, 1, 2, 2}).data;
    std::vector<float> t(4 * 64);
    for (int ci = 0; ci < 64; ++ci)
      for (int k = 0; k < 4; ++k) t[k * 64 + ci] = w[ci * 4 + k];
    tr2_w_ = arena_.upload(t);
    std::vector<float> tt(64 * 4);
    for (int ci = 0; ci < 64; ++ci)
      for (int k = 0; k < 4; ++k) tt[ci * 4 + k] = w[ci * 4 + k];
    tr2_wt_ = arena_.upload(tt);
    tr2_bias_ = wb.get("bin_conv_tr2.bias", {1}).data[0];
  }
  for (int l = 0; l < 4; ++l) {
    for (int b = 0; b < 2; ++b)
      for (int c = 0; c < 2; ++c) all_convs_.push_back(&layer_[l][b][c]);
    if (l > 0) all_convs_.push_back(&down_[l]);
    all_convs_.push_back(&in_[l]);
    all_convs_.push_back(&out_[l]);
  }
  all_convs_.push_back(&bin1_);
  if (fpn_composed_)
    for (int l = 0; l < 2; ++l) {
      all_convs_.push_back(&fpn_a_[l]);
      all_convs_.push_back(&fpn_b_[l]);
    }
  if (fpn_composed_) {
    for (int l = 0; l < 3; ++l) all_convs_.push_back(&bin_up_[l]);
    all_convs_.push_back(&bin_p2_);
    if (bin_pyr_on_) all_convs_.push_back(&bin_pyr_);
  }
  if (split_bf16_) {
    for (int l = 1; l < 4; ++l) add_split_weights(layer_[l][0][0]);   // the stride-2 3x3 convs
    add_split_weights(in_[3]);    // in5 (1x1, 512 -> 256 at H/32): few tiles, long K - MFMA-bound as well (0.040 -> 0.030 ms)
    add_split_weights(in_[2]);    // in4 (1x1, 256 -> 256 at H/16, with the top-down sum as its second output)
    for (int l = 1; l < 4; ++l) add_split_weights(down_[l]);   // the 1x1 stride-2 downsample convs
    if (fpn_composed_) {
      for (int l = 0; l < 2; ++l) add_split_weights(fpn_b_[l], 4 * fpn_b_[l].cin);       // phase convs of p2 / p3: [phase][Cout][2 x 2][Cin]
      for (int l = 0; l < 3; ++l) add_split_weights(bin_up_[l], 4 * bin_up_[l].cin);
      if (bin_pyr_on_) add_split_weights(bin_pyr_, 21 * 64);
    }
  }
  if (opt_bf16_) set_precision(1);
}

// f32 -> bf16, round to nearest even (NaN stays NaN)
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

void Detector::set_precision(int precision) {
  if (precision != 0 && precision != 1) fail(OCR_ERR_INVALID, "precision %d (0 = f32, 1 = bf16)", precision);
  OCR_HIP(hipSetDevice(device_));
  if (precision == 1) {
    auto upload_bf16 = [&](const uint16_t* h, size_t count) {
      std::vector<float> packed((count + 1) / 2);
      std::memcpy(packed.data(), h, count * 2);
      return static_cast<void*>(arena_.upload(packed));
    };
    if (!stem_wb_) {  // conv1 as bf16 MFMA fragments, bin_conv_tr1 as a bf16 GEMM operand
      const std::vector<uint16_t> fr = stem_bf16_fragments(stem_w_host_.data());
      stem_wb_ = upload_bf16(fr.data(), fr.size());
      std::vector<uint16_t> t1(tr1_.host.size());
      for (size_t i = 0; i < t1.size(); ++i) t1[i] = f32_to_bf16(tr1_.host[i]);
      tr1_.w_bf16 = upload_bf16(t1.data(), t1.size());
    }
    for (ConvW* cw : all_convs_) {
      if (cw->ks == 3 && cw->cin == 64 && cw->cout == 64 && !cw->w_bf16_c64 && !cw->host.empty()) {
        const std::vector<uint16_t> fr = conv3x3_bf16_c64_fragments(cw->host.data());
        cw->w_bf16_c64 = upload_bf16(fr.data(), fr.size());
      }
      if (cw->w_bf16) continue;
      std::vector<float> packed((cw->host.size() + 1) / 2);
      uint16_t* h = reinterpret_cast<uint16_t*>(packed.data());
      for (size_t i = 0; i < cw->host.size(); ++i) h[i] = f32_to_bf16(cw->host[i]);
      cw->w_bf16 = arena_.upload(packed);
    }
  }
  bf16_ = precision == 1;
}

void Detector::free_workspace() {
  for (void* p : ws_allocs_) (void)hipFree(p);
  ws_allocs_.clear();
  ws_n_ = ws_h_ = ws_w_ = 0;
}

Detector::~Detector() {
  (void)hipSetDevice(device_);
  // every stream that may still hold work of this handle - the caller's (ocr_det_set_stream + an _async call) included - drains
  // before anything it reads or writes is freed
  if (stream_ && stream_ != own_stream_) (void)hipStreamSynchronize(stream_);
  if (own_stream_) (void)hipStreamSynchronize(own_stream_);
  if (post_stream_) (void)hipStreamSynchronize(post_stream_);
  if (trace_stream_) (void)hipStreamSynchronize(trace_stream_);
  if (copy_stream_) {
    (void)hipStreamSynchronize(copy_stream_);
    (void)hipStreamSynchronize(out_stream_);
  }
  if (side_stream_) {
    (void)hipStreamSynchronize(side_stream_);
    (void)hipStreamDestroy(side_stream_);
    (void)hipEventDestroy(ev_x1_);
    (void)hipEventDestroy(ev_x2_);
    (void)hipEventDestroy(ev_x3_);
    (void)hipEventDestroy(ev_side_);
    (void)hipEventDestroy(ev_fork_);
    (void)hipEventDestroy(ev_join_);
  }
  if (post_stream_) {
    (void)hipStreamSynchronize(post_stream_);
    (void)hipStreamDestroy(post_stream_);
  }
  if (trace_stream_) (void)hipStreamDestroy(trace_stream_);
  if (trace_done_) (void)hipEventDestroy(trace_done_);
  for (hipEvent_t ev : pipe_ev_)
    if (ev) (void)hipEventDestroy(ev);
  free_workspace();
  if (host_scratch_) (void)hipHostFree(host_scratch_);
  if (host_adj_) (void)hipHostFree(host_adj_);
  for (void* p : scratch_)
    if (p) (void)hipFree(p);
  for (Staging& st : stage_)
    for (int i = 0; i < 2; ++i) {
      if (st.in[i]) (void)hipFree(st.in[i]);
      if (st.out[i]) (void)hipFree(st.out[i]);
      if (st.ev_in[i]) (void)hipEventDestroy(st.ev_in[i]);
      if (st.ev_fwd[i]) (void)hipEventDestroy(st.ev_fwd[i]);
      if (st.ev_out[i]) (void)hipEventDestroy(st.ev_out[i]);
    }
  if (ev_before_fwd_) (void)hipEventDestroy(ev_before_fwd_);
  if (copy_stream_) {
    (void)hipStreamSynchronize(copy_stream_);
    (void)hipStreamDestroy(copy_stream_);
    (void)hipStreamSynchronize(out_stream_);
    (void)hipStreamDestroy(out_stream_);
  }
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

// CPU share of this process: its affinity mask (a rank pinned by taskset / numactl), capped by the cgroup quota where there
// is one (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us: a container on a big host)
static int host_cpu_share() {
  unsigned n = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
  if (n == 0) n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  auto read_ll = [](const char* path, long long* v) {
    FILE* f = std::fopen(path, "r");
    if (!f) return false;
    const bool ok = std::fscanf(f, "%lld", v) == 1;
    std::fclose(f);
    return ok;
  };
  long long q1 = 0, p1 = 0;
  if (read_ll("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q1) && read_ll("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &p1) && q1 > 0 && p1 > 0) {
    const long long q = q1 / p1;
    if (q >= 1 && (unsigned)q < n) n = (unsigned)q;
  }
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32] = {0};
    long long period = 0;
    if (std::fscanf(f, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
      const long long q = std::atoll(quota) / period;
      if (q >= 1 && (unsigned)q < n) n = (unsigned)q;
    }
    std::fclose(f);
  }
  return (int)n;
}

int Detector::post_threads() const {
  if (post_threads_ > 0) return post_threads_;
  if (auto_threads_ == 0) auto_threads_ = std::min(16, host_cpu_share());   // (three sysfs reads: once per handle, not per call)
  return auto_threads_;
}

// auto (the default): the whole polygon chain on the device where the host pool has at most four threads - by measurement (DESIGN.md
// section 4, tools/device_contours_sweep.py, frames per second on text / dense pages, chain against host tracer + device unclip): f32, one
// thread 6.2 k / 6.1 k against 6.1 k / 5.0 k, two 6.2 k / 6.1 k against 6.2 k / 6.0 k, four 6.2 k / 6.1 k against 6.3 k / 6.1 k; bf16 15.7 k /
// 15.0 k against 10.5 k / 5.3 k, 15.6 k / 14.9 k against 15.0 k / 8.2 k, 15.7 k / 15.1 k against 15.7 k / 13.0 k.  With a large pool the host
// tracer is a few per cent ahead on sparse pages (nothing of the chain then competes with the next forward for CUs)
int Detector::device_contours() const { return device_contours_ >= 0 ? device_contours_ : (post_threads() <= 4 ? 1 : 0); }

ThreadPool& Detector::pool() {
  if (!pool_) pool_ = std::make_unique<ThreadPool>(post_threads() - 1);  // + the calling thread
  return *pool_;
}

// the post-processing and trace streams run short latency-bound kernels (tracer, Douglas-Peucker, box scores, unclip) beside the next
// forward's long ones: at the highest priority the device offers their workgroups are placed as soon as a CU drains instead of queueing
// behind the forward's (option post_priority=0: default priority)
static hipStream_t make_side_stream(bool high_priority) {
  hipStream_t s = nullptr;
  int lo = 0, hi = 0;
  if (high_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo &&
      hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) == hipSuccess)
    return s;
  (void)hipGetLastError();   // (no priorities on this device / runtime: an ordinary stream)
  OCR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}

hipStream_t Detector::post_stream() {
  if (!post_stream_) post_stream_ = make_side_stream(post_priority_);
  return post_stream_;
}

hipStream_t Detector::trace_stream() {
  if (!trace_stream_) trace_stream_ = make_side_stream(post_priority_);
  return trace_stream_;
}
hipEvent_t Detector::trace_done_event() {
  if (!trace_done_) OCR_HIP(hipEventCreateWithFlags(&trace_done_, hipEventDisableTiming));
  return trace_done_;
}

hipEvent_t Detector::pipeline_event() {
  hipEvent_t& e = pipe_ev_[pipe_ev_next_];
  pipe_ev_next_ ^= 1;
  if (!e) OCR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return e;
}

void Detector::synchronize() {
  OCR_HIP(hipSetDevice(device_));
  OCR_HIP(hipStreamSynchronize(stream_));
}

void* Detector::host_scratch(size_t bytes) {
  if (bytes > host_scratch_bytes_) {
    if (host_scratch_) OCR_HIP(hipHostFree(host_scratch_));
    host_scratch_ = nullptr;
    host_scratch_bytes_ = 0;
    const size_t want = bytes + bytes / 2;
    OCR_HIP(hipHostMalloc(&host_scratch_, want, hipHostMallocDefault));
    host_scratch_bytes_ = want;
  }
  return host_scratch_;
}

void* Detector::host_adj(size_t bytes) {
  if (bytes > host_adj_bytes_) {
    if (host_adj_) OCR_HIP(hipHostFree(host_adj_));
    host_adj_ = nullptr;
    host_adj_bytes_ = 0;
    const size_t want = bytes < 4096 ? 4096 : 2 * bytes;
    OCR_HIP(hipHostMalloc(&host_adj_, want, hipHostMallocDefault));
    host_adj_bytes_ = want;
  }
  return host_adj_;
}

void* Detector::scratch(int slot, size_t bytes) {
  if (slot < 0 || slot > 4) fail(OCR_ERR_INTERNAL, "scratch slot %d", slot);
  if (bytes > scratch_bytes_[slot]) {
    // every stream that may read or write a slot drains before a growing slot is freed (slot 2: post_stream_; slot 3: written on
    // trace_stream_, read on post_stream_)
    OCR_HIP(hipStreamSynchronize(stream_));
    if (post_stream_) OCR_HIP(hipStreamSynchronize(post_stream_));
    if (trace_stream_) OCR_HIP(hipStreamSynchronize(trace_stream_));
    if (scratch_[slot]) OCR_HIP(hipFree(scratch_[slot]));
    scratch_[slot] = nullptr;
    scratch_bytes_[slot] = 0;
    const size_t want = bytes + bytes / 4;  // head room: fewer re-allocations as batches vary
    OCR_HIP(hipMalloc(&scratch_[slot], want));
    scratch_bytes_[slot] = want;
  }
  return scratch_[slot];
}

void Detector::ensure_workspace(int n, int h, int w) {
  // sized for the largest batch seen at this frame size: a smaller batch (the remainder chunk of a split batch,
  // a caller alternating batch sizes) reuses it - every launch takes its extents from the call's own n
  last_n_ = n;
  if (n <= ws_n_ && h == ws_h_ && w == ws_w_ && ws_bf16_ == bf16_) return;
  OCR_HIP(hipStreamSynchronize(stream_));
  free_workspace();
  const size_t es = bf16_ ? 2 : 4;  // bytes per activation element
  auto alloc = [&](size_t bytes) {
    void* p = nullptr;
    OCR_HIP(hipMalloc(&p, bytes));
    ws_allocs_.push_back(p);
    return static_cast<char*>(p);
  };
  const size_t N = (size_t)n;
  s_ = alloc(N * (h / 4) * (w / 4) * 64 * es);
  size_t pcat_elems = 0;
  for (int l = 0; l < 4; ++l) pcat_elems += N * (h >> (2 + l)) * (w >> (2 + l)) * 64;
  pcat_ = alloc(pcat_elems * es);  // p2..p5 in ONE allocation: bin_conv1 gathers all four through one buffer descriptor
  pcat_bytes_ = pcat_elems * es;
  size_t pofs = 0;
  for (int l = 0; l < 4; ++l) {
    const size_t px = N * (h >> (2 + l)) * (w >> (2 + l));
    const size_t c = (size_t)64 << l;
    t_[l] = alloc(px * c * es);
    a_[l] = alloc(px * c * es);
    x_[l] = alloc(px * c * es);
    d_[l] = l > 0 ? alloc(px * c * es) : nullptr;
    // raw lateral in3..in5 (in2 is only ever used inside its sum) and the sums up2(in_{k+1}) + in_k; with
    // the composed FPN only in5 and the level-4 sum are ever materialised
    const bool need_i = fpn_composed_ ? l == 3 : l > 0;
    const bool need_sum = fpn_composed_ ? l == 2 : l < 3;
    i_[l] = need_i ? alloc(px * 256 * es) : nullptr;
    if (l < 3) sum_[l] = need_sum ? alloc(px * 256 * es) : nullptr;
    p_[l] = pcat_ + pofs * es;
    pofs += px * 64;
  }
  b1_ = reinterpret_cast<float*>(alloc(N * (h / 4) * (w / 4) * 64 * 4));
  wino_v_ = wino_m_ = nullptr;
  {
    size_t need = 0;  // components * tiles * channels of the largest Winograd layer
    for (int l = 0; l < 4; ++l)
      if (layer_[l][1][0].wino) {
        const size_t m = layer_[l][1][0].wino_tile, a = m + 2;
        const size_t th = ((h >> (2 + l)) + m - 1) / m, tw = ((w >> (2 + l)) + m - 1) / m;
        need = std::max(need, a * a * N * th * tw * ((size_t)64 << l));
      }
    if (need) {
      wino_v_ = reinterpret_cast<float*>(alloc(need * 4));
      wino_m_ = reinterpret_cast<float*>(alloc(need * 4));
    }
  }
  tr1buf_ = fused_tail_ ? nullptr : reinterpret_cast<float*>(alloc(N * (h / 2) * (w / 2) * 64 * 4));
  ws_bf16_ = bf16_;
  ws_n_ = n;
  ws_h_ = h;
  ws_w_ = w;
}

namespace {
// hipEvent pairs around every launch of ONE forward_chunk; entries are appended to the shared `prof` vector,
// so a batch that runs in several chunks keeps every chunk's launches (base = first entry of this chunk).
struct Recorder {
  std::vector<ProfileEntry>* prof;
  hipStream_t s;
  std::vector<hipEvent_t> ev;
  size_t base;
  Recorder(std::vector<ProfileEntry>* p, hipStream_t st) : prof(p), s(st), base(p ? p->size() : 0) {}
  Recorder(const Recorder&) = delete;
  ~Recorder() {  // also on the exceptional path: no event outlives its chunk
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
  void mark() {
    hipEvent_t e;
    OCR_HIP(hipEventCreate(&e));
    ev.push_back(e);
    OCR_HIP(hipEventRecord(e, s));
  }
  void begin() {
    if (prof) mark();
  }
  void end(const char* name, double flops, double bytes) {
    if (!prof) return;
    mark();
    prof->push_back({name, 0.f, flops, bytes});
  }
  void finish() {
    if (!prof) return;
    OCR_HIP(hipStreamSynchronize(s));
    for (size_t i = base; i < prof->size(); ++i)
      OCR_HIP(hipEventElapsedTime(&(*prof)[i].ms, ev[2 * (i - base)], ev[2 * (i - base) + 1]));
  }
};
}  // namespace

void Detector::forward(const void* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh,
                       std::vector<ProfileEntry>* prof, int x_u8, hipEvent_t wait_for) {
  if (!x || !prob) fail(OCR_ERR_INVALID, "det_forward: null tensor");
  if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32)
    fail(OCR_ERR_INVALID, "det_forward: N=%d H=%d W=%d (H and W must be positive multiples of 32)", n, h, w);
  OCR_HIP(hipSetDevice(device_));
  // The conv kernel addresses each tensor with 32-bit byte offsets below 2^31: the largest
  // workspace tensor holds N*(H/4)*(W/4)*256 floats = N*H*W*64 bytes.  Larger batches run in chunks.
  const long long per_frame = (long long)h * w * 64;
  const int max_n = (int)std::min<long long>(n, ((1ll << 31) - 1) / per_frame);
  if (max_n < 1) fail(OCR_ERR_INVALID, "det_forward: a %dx%d frame exceeds the 2^31-byte tensor limit", h, w);
  for (int b = 0; b < n; b += max_n) {
    const int nb = std::min(max_n, n - b);
    const size_t off = (size_t)b * h * w;
    if (wait_for && b == 0) OCR_HIP(hipStreamWaitEvent(stream_, wait_for, 0));   // e.g. the copy that brings x in
    forward_chunk(static_cast<const char*>(x) + off * (x_u8 ? 1 : 4), nb, h, w, prob + off, bitmap ? bitmap + off : nullptr, thresh, prof, x_u8);
  }
}

void Detector::forward_chunk(const void* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh,
                             std::vector<ProfileEntry>* prof, int x_u8) {
  ensure_workspace(n, h, w);
  Recorder rec(prof, stream_);

  struct Extra {
    const void* residual = nullptr;
    const void* up_residual = nullptr;
    void* out2 = nullptr;
    bool cat4 = false;
    bool pyr4 = false;      // SRC_PYR4: the four pyramid levels as sources of one phase-conv launch
    int pyr_nsrc = 4;       // ... or only the three upsampled ones (3)
    int pyr_group = 0;      // split-bf16 form: 1 = the phase blocks that share their operand rows as 128-column tiles, 2 = the corner phases
    bool f32_out = false;   // bf16 precision: keep this conv's result (and residual) in f32
    int store = STORE_NHWC;
  };
  const bool bf = bf16_;
  const size_t es = bf ? 2 : 4;
  hipStream_t cs = stream_;  // stream of the launches below (the side stream while the FPN branch is enqueued)
  // Persistent grids (fused Winograd, the bf16 64 -> 64 conv) are sized for every CU and deal their blocks statically: a workgroup
  // that finds its CU taken starts late and still carries its full share - the launch takes twice as long.  While the polygon chain of
  // the PREVIOUS batch runs (pipelined calls: its tracer holds one whole CU per image for about a millisecond, right when this forward
  // starts), layer1's launches are sized for the CUs that are free: + 14 % on four launches instead of + 100 %
  // (head_cus_yield=1, the round-5 form, while the tracer held a whole CU per image.)  With the tracer's one-plane LDS image a forward workgroup
  // shares the CU with it and runs at a fraction of its speed; head_cus_yield >= 2 OVERSUBSCRIBES layer1's grids instead - that many workgroups
  // per resident slot, so the hardware hands the later ones to whichever CU drains first: a CU that is shared simply gets fewer of them.
  const bool chain_beside = pending_.valid && pending_.prechained;
  const int busy_cus = head_cus_yield_ == 1 && chain_beside ? std::min(pending_.n, num_cus_ / 4) : 0;
  int grid_cus = num_cus_ - busy_cus;
  if (head_cus_yield_ >= 2 && chain_beside) grid_cus = num_cus_ * head_cus_yield_;
  const bool overlap3 = overlap_ >= 3 && !prof && fpn_composed_ && bin_pyr_on_ && fused_tail_ &&
                        (bf16_ ? (fpn_a_[0].w_bf16_c64 && bin_p2_.w_bf16_c64 && pyr_p2_direct_)
                               : (fpn_a_[0].wino43_fused && fpn_a_[1].wino43_fused && bin_p2_.wino43_fused && !fpn_a_[0].wino43_x3 && split_bf16_));
  const bool overlap = overlap3 || (overlap_ == 2 && !prof && fpn_composed_ && !bf16_);   // the FPN branch on the side stream, joined before bin_conv1
  const bool overlap_small = (overlap_ == 1 || overlap_ == 2) && !prof;   // (3: the side stream is the FPN branch's alone)
  // run `side_work` on the second stream from this point of the main stream on; join() makes the main stream
  // wait for it
  auto fork = [&](auto&& side_work) {
    OCR_HIP(hipEventRecord(ev_fork_, stream_));
    OCR_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
    cs = side_stream_;
    side_work();
    cs = stream_;
    OCR_HIP(hipEventRecord(ev_join_, side_stream_));
  };
  auto join = [&] { OCR_HIP(hipStreamWaitEvent(stream_, ev_join_, 0)); };
  auto conv = [&](const char* name, const ConvW& cw, const void* src, int hin, int win, int stride, void* out,
                  bool relu, const Extra& ex = Extra()) {
    ConvDesc d{};
    // the probability head (bin_conv_tr1 as a GEMM) always runs in f32; bin_conv1 reads bf16 and writes f32
    const bool in_bf = bf && ex.store != STORE_SHUFFLE2;
    d.in_bf16 = in_bf ? 1 : 0;
    // bin_conv1 (the gathered CAT4 conv) feeds the fused head in its own element type: bf16 in the bf16 precision
    d.out_bf16 = (in_bf && !(ex.cat4 && !fused_tail_) && !ex.f32_out) ? 1 : 0;
    d.pyr_nsrc = ex.pyr_nsrc;
    d.pyr_group = ex.pyr_group;
    d.up = cw.up;
    const size_t ies = in_bf ? 2 : 4;
    d.src[0] = src;
    d.src_mode = ex.cat4 ? SRC_CAT4 : ex.pyr4 ? SRC_PYR4 : SRC_PLAIN;
    if (ex.cat4 || ex.pyr4) {
      d.src[0] = p_[3];
      d.src[1] = p_[2];
      d.src[2] = p_[1];
      d.src[3] = p_[0];
      d.src_base = pcat_;
      d.src_bytes = pcat_bytes_;
    } else {
      d.src_bytes = (size_t)n * hin * win * cw.cin * ies;
    }
    d.wgt_bytes = in_bf ? cw.w_bytes / 2 : cw.w_bytes;
    d.N = n;
    d.Hin = hin;
    d.Win = win;
    d.Cin = cw.cin;
    d.ks = cw.ks;
    d.stride = stride;
    d.pad = ex.pyr4 ? 0 : ex.store == STORE_PHASE ? 1 : (cw.ks - 1) / 2;
    d.Ho = ex.store == STORE_PHASE ? hin : (hin + 2 * d.pad - cw.ks) / stride + 1;
    d.Wo = ex.store == STORE_PHASE ? win : (win + 2 * d.pad - cw.ks) / stride + 1;
    d.Cout = cw.cout;
    d.wgt = in_bf ? cw.w_bf16 : static_cast<const void*>(cw.w);
    if (!bf && split_bf16_ && cw.w_x3 && !ex.cat4 && ex.store != STORE_SHUFFLE2 && !(ex.pyr4 && ex.pyr_nsrc != 3)) {
      d.x3 = 1;
      d.wide = x3_wide_ ? 1 : 0;
      d.wgt = cw.w_x3;
      d.wgt_bytes = cw.w_bytes / 4 * 6;
    }
    // up-2 phase convs (the FPN's upsampled terms): rows = 2 x 2 windows, the four phases as column groups of one operand tile
    if ((d.x3 || in_bf) && ex.store == STORE_PHASE && !ex.pyr4 && cw.up == 2 && cw.cout == 64 && phase_windows_) d.win = 1;
    d.scale = cw.scale;
    d.bias = cw.bias;
    d.residual = ex.residual;
    d.up_residual = ex.up_residual;
    d.out2 = ex.out2;
    d.relu = relu ? 1 : 0;
    d.store_mode = ex.store;
    d.out = out;
    d.name = name;
    rec.begin();
    launch_conv_igemm(d, cs);
    const double M = (double)n * d.Ho * d.Wo;
    const double reps = ex.store == STORE_PHASE ? (double)(cw.up * cw.up) : 1.0;  // phase convs per low-res pixel
    // taps executed per low-res pixel over all phases: (up + 2)^2 (edge phases 2, inner phases 1 per direction)
    // PYR4 per cell: (8+2)^2 + 4 (4+2)^2 + 16 (2+2)^2 tap-phases of the upsampled levels + 64 * 9 of p2
    // (phase blocks: the corner phases' 4 x 12 tap-phases are their own launch)
    const double K = ex.pyr4 ? 64.0 * ((ex.pyr_group == 1 ? 452 : ex.pyr_group == 2 ? 48 : 500) + (ex.pyr_nsrc == 4 ? 576 : 0))
                             : ex.store == STORE_PHASE ? (double)cw.cin * (cw.up + 2) * (cw.up + 2) : (double)cw.ks * cw.ks * cw.cin;
    double in_bytes = (double)n * hin * win * cw.cin * (double)ies;
    if (ex.cat4) in_bytes = (double)n * hin * win * 64 * (double)ies * (1.0 + 0.25 + 1.0 / 16 + 1.0 / 64);
    if (ex.pyr4) in_bytes = (double)n * hin * win * 64 * (double)ies * (1.0 + 4.0 + 16.0 + (ex.pyr_nsrc == 4 ? 64.0 : 0.0));
    const double oes = d.out_bf16 ? 2.0 : 4.0;
    const double phases = ex.pyr_group == 1 ? 60.0 : ex.pyr_group == 2 ? 4.0 : reps;   // output pixels written per low-res pixel
    double out_bytes = M * phases * cw.cout * oes * ((out ? 1.0 : 0.0) + (ex.out2 ? 1.0 : 0.0) + (ex.residual ? 1.0 : 0.0) +
                                            (ex.up_residual ? 0.25 : 0.0));
    rec.end(conv_igemm_kernel_name(d), 2.0 * M * cw.cout * K, in_bytes + out_bytes + K * cw.cout * (double)ies);
  };

  const int h4 = h / 4, w4 = w / 4;
  rec.begin();
  if (bf) launch_stem_bf16(x, x_u8, stem_wb_, stem_scale_, stem_bias_, s_, n, h, w, stream_);
  else if (stem_wx3_) launch_stem_x3(x, x_u8, stem_wx3_, stem_scale_, stem_bias_, reinterpret_cast<float*>(s_), n, h, w, stream_);
  else launch_stem(x, x_u8, stem_w_, stem_scale_, stem_bias_, s_, 0, n, h, w, stream_);
  rec.end(!bf && stem_wx3_ ? "stem_x3_conv7x7_bn_relu_maxpool" : "stem_conv7x7_bn_relu_maxpool", 2.0 * n * (h / 2) * (w / 2) * 64 * 49,
          (double)n * h * w * (x_u8 ? 1 : 4) + (double)n * h4 * w4 * 64 * (double)es);

  // the three launches of an unfused Winograd conv (layer3 / layer4, out4, out5), on the main stream: x -> V, V -> M (36 or 16 GEMMs in one
  // batched launch), M -> y with the epilogue
  auto wino_dims = [&](const ConvW& cw, int hh, int ww, size_t& wm, size_t& wa, size_t& T) {
    wm = cw.wino_tile;
    wa = (wm + 2) * (wm + 2);
    T = (size_t)n * ((hh + wm - 1) / wm) * ((ww + wm - 1) / wm);
  };
  auto wino_in = [&](const ConvW& cw, const void* src, int hh, int ww) {
    size_t wm, wa, T;
    wino_dims(cw, hh, ww, wm, wa, T);
    rec.begin();
    launch_winograd_input(static_cast<const float*>(src), wino_v_, n, hh, ww, cw.cin, (int)wm, stream_);
    rec.end(wm == 4 ? "winograd43_input_transform" : "winograd_input_transform", 0.0,
            (double)n * hh * ww * cw.cin * 4.0 + (double)wa * T * cw.cin * 4.0);
  };
  auto wino_gemm = [&](const char* name, const ConvW& cw, int hh, int ww) {
    size_t wm, wa, T;
    wino_dims(cw, hh, ww, wm, wa, T);
    ConvDesc d{};
    d.src[0] = wino_v_;
    d.src_mode = SRC_PLAIN;
    d.src_bytes = wa * T * cw.cin * 4;
    d.wgt = cw.wino;
    d.wgt_bytes = cw.wino_bytes;
    if (split_bf16_ && cw.wino_x3) {
      d.x3 = 1;
      d.wide = x3_wide_ ? 1 : 0;
      d.wgt = cw.wino_x3;
      d.wgt_bytes = cw.wino_bytes / 4 * 6;
    }
    d.batch = (int)wa;
    d.N = 1;
    d.Hin = d.Ho = 1;
    d.Win = d.Wo = (int)T;
    d.Cin = cw.cin;
    d.Cout = cw.cout;
    d.ks = 1;
    d.stride = 1;
    d.pad = 0;
    d.store_mode = STORE_NHWC;
    d.out = wino_m_;
    d.name = name;
    rec.begin();
    launch_conv_igemm(d, stream_);
    rec.end(conv_igemm_kernel_name(d), 2.0 * wa * T * cw.cin * cw.cout,
            (double)wa * 4.0 * ((double)T * cw.cin + (double)T * cw.cout + (double)cw.cin * cw.cout));
  };
  auto wino_out = [&](const ConvW& cw, int hh, int ww, void* out, const void* residual, bool relu) {
    size_t wm, wa, T;
    wino_dims(cw, hh, ww, wm, wa, T);
    rec.begin();
    launch_winograd_output(wino_m_, cw.scale, cw.bias, static_cast<const float*>(residual), relu ? 1 : 0, static_cast<float*>(out),
                           n, hh, ww, cw.cout, (int)wm, stream_);
    rec.end(wm == 4 ? "winograd43_output_transform" : "winograd_output_transform", 0.0,
            (double)wa * T * cw.cout * 4.0 + (double)n * hh * ww * cw.cout * 4.0 * (residual ? 2.0 : 1.0));
  };
  // ... and the output transform of one conv fused with the input transform of the next (winograd.hip: the image of a workgroup's 16
  // channels stays in LDS); y may be null when only the next conv reads the activation
  auto wino_out_in = [&](const ConvW& cw, int hh, int ww, void* y, const void* residual, bool relu) {
    size_t wm, wa, T;
    wino_dims(cw, hh, ww, wm, wa, T);
    rec.begin();
    launch_winograd43_out_in(wino_m_, cw.scale, cw.bias, static_cast<const float*>(residual), relu ? 1 : 0, static_cast<float*>(y), wino_v_, n, hh, ww,
                             cw.cout, stream_);
    rec.end("winograd43_output+input_transform", 0.0, 2.0 * (double)wa * T * cw.cout * 4.0 + (double)n * hh * ww * cw.cout * 4.0 * ((residual ? 1.0 : 0.0) + (y ? 1.0 : 0.0)));
  };
  // can this conv take the unfused F(4x4) path with fused neighbours?  (the conditions of conv3x3's last branch + the LDS image)
  auto wino43_unfused = [&](const ConvW& cw, int hh, int ww) {
    if (bf || !cw.wino || !wino_v_ || cw.wino_tile != 4 || cw.wino43_x3 || cw.wino43_fused || cw.cin != cw.cout) return false;
    size_t wm, wa, T;
    wino_dims(cw, hh, ww, wm, wa, T);
    return wa * T * (size_t)cw.cin * 4 < ((size_t)1 << 31) && winograd43_out_in_fits(hh, ww, cw.cout);
  };

  // 3x3 s1 conv + BN (+ residual) + ReLU of the deep layers as Winograd F(2x2,3x3): input transform, sixteen
  // [tiles x Cin] x [Cin x Cout] GEMMs in one batched launch, output transform with the epilogue (f32 only)
  auto conv3x3 = [&](const char* name, const ConvW& cw, const void* src, int hh, int ww, void* out, const void* residual,
                     bool relu = true) {
    if (!bf && cw.wino43_x3) {  // transforms fused into the GEMM kernel, the GEMMs on the bf16 matrix cores (split-bf16)
      rec.begin();
      launch_winograd43_x3(static_cast<const float*>(src), cw.wino43_x3, cw.scale, cw.bias, static_cast<const float*>(residual),
                           relu ? 1 : 0, static_cast<float*>(out), n, hh, ww, cw.cin, cw.cout, num_cus_, cs);
      const double px43 = (double)n * hh * ww;
      rec.end(cw.cin == 64 ? "winograd43_fused_x3<c64>" : cw.cin == 128 ? "winograd43_fused_x3<c128>" : "winograd43_fused_x3<c256>",
              2.0 * 36.0 * (px43 / 16.0) * cw.cin * cw.cout, px43 * 4.0 * (cw.cin + cw.cout * (residual ? 2.0 : 1.0)) + 36.0 * cw.cin * cw.cout * 6);
      return;
    }
    if (!bf && cw.wino43_fused) {  // transforms fused into the GEMM kernel
      {
        rec.begin();
        launch_winograd43_fused(static_cast<const float*>(src), cw.wino43_fused, cw.scale, cw.bias, static_cast<const float*>(residual),
                                relu ? 1 : 0, static_cast<float*>(out), n, hh, ww, cw.cin, cw.cout,
                                cs != stream_ && w43_side_cus_ > 0 ? w43_side_cus_ : w43_cus_ > 0 ? w43_cus_ : cs == stream_ ? grid_cus : num_cus_, cs);
        const double px43 = (double)n * hh * ww;
        rec.end(cw.cin == 64 ? "winograd43_fused<c64>" : cw.cin == 128 ? "winograd43_fused<c128>" : "winograd43_fused<c256>", 2.0 * 36.0 * (px43 / 16.0) * cw.cin * cw.cout,
                px43 * 4.0 * (cw.cin + cw.cout * (residual ? 2.0 : 1.0)) + 36.0 * cw.cin * cw.cout * 4);
        return;
      }
    }
    if (bf && cw.w_bf16_c64 && (long long)n * hh * ww * 128 < (1ll << 31)) {  // bf16 64 -> 64: patch staged once, weights in registers
      rec.begin();
      launch_conv3x3_bf16_c64(src, cw.w_bf16_c64, cw.scale, cw.bias, residual, relu ? 1 : 0, out, n, hh, ww, cs == stream_ ? grid_cus : num_cus_, cs);
      const double px = (double)n * hh * ww;
      rec.end("conv3x3_bf16_c64", 2.0 * px * 64 * 576, px * 2.0 * 64 * (residual ? 3.0 : 2.0) + 9.0 * 64 * 64 * 2);
      return;
    }
    const size_t wm = cw.wino_tile, wa = (wm + 2) * (wm + 2);   // F(wm x wm, 3x3): wa components
    const size_t th = (hh + wm - 1) / wm, tw = (ww + wm - 1) / wm, T = (size_t)n * th * tw;
    if (bf || !cw.wino || !wino_v_ || wa * T * std::max(cw.cin, cw.cout) * 4 >= ((size_t)1 << 31)) {
      Extra ex;
      ex.residual = residual;
      conv(name, cw, src, hh, ww, 1, out, relu, ex);
      return;
    }
    wino_in(cw, src, hh, ww);
    wino_gemm(name, cw, hh, ww);
    wino_out(cw, hh, ww, out, residual, relu);
  };

  // composed FPN level lv (0: p2, 1: p3) and its term of bin_conv1: p_k = A_k * x_k + B_k *' x_{k+1} - the
  // upsampled term first (phase store), the lateral term on top - then that quarter of the concat into b1.
  // bin_conv1's four terms accumulate in the order p2, p3, p4, p5; the last one adds the bias and the ReLU.
  auto fpn_level = [&](int lv) {
    Extra up;
    up.store = STORE_PHASE;
    conv("fpn.upsampled", fpn_b_[lv], x_[lv + 1], h >> (3 + lv), w >> (3 + lv), 1, p_[lv], false, up);
    if ((fpn_a_[lv].wino43_fused && !bf) || (bf && fpn_a_[lv].w_bf16_c64)) {
      conv3x3("fpn.lateral", fpn_a_[lv], x_[lv], h >> (2 + lv), w >> (2 + lv), p_[lv], p_[lv], false);
    } else {
      Extra lat;
      lat.residual = p_[lv];
      conv("fpn.lateral", fpn_a_[lv], x_[lv], h >> (2 + lv), w >> (2 + lv), 1, p_[lv], false, lat);
    }
    if (bf || bin_pyr_on_) return;  // bf16 keeps the single gathered bin_conv1; PYR4 takes all four terms at once
    if (lv == 0) {
      Extra first;
      first.f32_out = true;
      conv("bin_conv1.p2", bin_p2_, p_[0], h4, w4, 1, b1_, false, first);
    } else {
      Extra t;
      t.store = STORE_PHASE;
      t.f32_out = true;
      t.residual = b1_;
      conv("bin_conv1.upsampled", bin_up_[0], p_[1], h >> 3, w >> 3, 1, b1_, false, t);
    }
  };

  // ResNet-18 trunk, model.rs:113-120 (basic_block :40-55)
  const void* cur = s_;
  for (int l = 0; l < 4; ++l) {
    const int hin = l == 0 ? h4 : (h >> (1 + l)), win = l == 0 ? w4 : (w >> (1 + l));
    const int ho = h >> (2 + l), wo = w >> (2 + l);
    const int stride = l == 0 ? 1 : 2;
    const bool block_fused = l == 0 && bf && bf16_block_fuse_ && layer_[0][0][0].w_bf16_c64 && layer_[0][0][1].w_bf16_c64 && layer_[0][1][0].w_bf16_c64 &&
        layer_[0][1][1].w_bf16_c64 && basic_block_bf16_c64_applicable(n, ho, wo);
    if (block_fused) {
      // bf16 precision: each of layer1's BasicBlocks as ONE launch, the activation between its two convs stays in LDS (same bits as
      // the two conv3x3_bf16_c64 launches; model.rs:40-55)
      const void* bin = cur;
      void* bout[2] = {a_[0], x_[0]};
      for (int b = 0; b < 2; ++b) {
        const ConvW &c1 = layer_[0][b][0], &c2 = layer_[0][b][1];
        rec.begin();
        launch_basic_block_bf16_c64(bin, c1.w_bf16_c64, c1.scale, c1.bias, c2.w_bf16_c64, c2.scale, c2.bias, bout[b], n, ho, wo,
                                    cs == stream_ ? grid_cus : num_cus_, cs);
        const double px = (double)n * ho * wo;
        rec.end("basic_block_bf16_c64", 2.0 * 2.0 * px * 64 * 576, px * 2.0 * 64 * 2.0 + 2.0 * 9.0 * 64 * 64 * 2);
        bin = bout[b];
      }
    } else {
      Extra sc;
      sc.residual = cur;
      if (l > 0 && overlap_small) {
        fork([&] { conv("layer.downsample", down_[l], cur, hin, win, stride, d_[l], false); });
        conv("layer.conv1", layer_[l][0][0], cur, hin, win, stride, t_[l], true);
        join();
        sc.residual = d_[l];
      } else {
        if (l == 0) conv3x3("layer.conv1", layer_[l][0][0], cur, hin, win, t_[l], nullptr);  // stride 1 in layer1
        else conv("layer.conv1", layer_[l][0][0], cur, hin, win, stride, t_[l], true);
        if (l > 0) {
          conv("layer.downsample", down_[l], cur, hin, win, stride, d_[l], false);
          sc.residual = d_[l];
        }
      }
      if (transform_fuse_ && wino43_unfused(layer_[l][0][1], ho, wo) && wino43_unfused(layer_[l][1][0], ho, wo) && wino43_unfused(layer_[l][1][1], ho, wo)) {
        // three unfused F(4x4) convs in a row (layer3 / layer4): conv1's output inside block 1 is read by conv2 only - M -> y -> V in one
        // launch, the activation never reaches HBM (model.rs:40-55)
        // (the pair around the block boundary keeps its two launches: there y = a_[l] must be written anyway - the residual of block 1 -
        // and the fused launch, one workgroup per CU for its LDS image, is slower than the two streaming kernels: 0.107 vs 0.077 ms at H/16)
        wino_in(layer_[l][0][1], t_[l], ho, wo);
        wino_gemm("layer.conv2", layer_[l][0][1], ho, wo);
        wino_out(layer_[l][0][1], ho, wo, a_[l], sc.residual, true);
        wino_in(layer_[l][1][0], a_[l], ho, wo);
        wino_gemm("layer.conv1", layer_[l][1][0], ho, wo);
        wino_out_in(layer_[l][1][0], ho, wo, nullptr, nullptr, true);
        wino_gemm("layer.conv2", layer_[l][1][1], ho, wo);
        wino_out(layer_[l][1][1], ho, wo, x_[l], a_[l], true);
      } else {
        conv3x3("layer.conv2", layer_[l][0][1], t_[l], ho, wo, a_[l], sc.residual);
        conv3x3("layer.conv1", layer_[l][1][0], a_[l], ho, wo, t_[l], nullptr);
        conv3x3("layer.conv2", layer_[l][1][1], t_[l], ho, wo, x_[l], a_[l]);
      }
    }
    cur = x_[l];
    if (l == 0) grid_cus = num_cus_;   // (the tracer of the previous batch is done by now: 1.1 ms against stem + layer1 = 1.1 ms f32)
    if (overlap3 && l == 0) {
      // Side stream, from here to bin_conv1: the FPN's fused-Winograd launches and bin_conv1's p2 term - f32 matrix instructions,
      // latency-bound at two waves per SIMD - beside layer2 / layer3 / layer4 / the small FPN convs (split-bf16 GEMMs, HBM-bound transforms):
      // the two families leave each other issue slots and idle CUs.  p2's lateral term only needs layer1's output;
      // its upsampled term, bin_conv1's p2 term (into layer1's free temporary) and p3's lateral term follow layer2, p3's upsampled term
      // layer3.  Sums are re-associated (lateral + upsampled instead of upsampled + lateral: the same bits; pyramid + bias + p2 term instead
      // of p2 term + bias + pyramid: one rounding apart)
      OCR_HIP(hipEventRecord(ev_x1_, stream_));
      OCR_HIP(hipStreamWaitEvent(side_stream_, ev_x1_, 0));
      cs = side_stream_;
      conv3x3("fpn.lateral", fpn_a_[0], x_[0], h4, w4, p_[0], nullptr, false);
      cs = stream_;
    } else if (overlap3 && l == 1) {
      OCR_HIP(hipEventRecord(ev_x2_, stream_));
      OCR_HIP(hipStreamWaitEvent(side_stream_, ev_x2_, 0));
      cs = side_stream_;
      {
        Extra up;
        up.store = STORE_PHASE;
        up.residual = p_[0];
        conv("fpn.upsampled", fpn_b_[0], x_[1], h >> 3, w >> 3, 1, p_[0], false, up);
      }
      {
        ConvW p2 = bin_p2_;
        p2.bias = nullptr;
        conv3x3("bin_conv1.p2", p2, p_[0], h4, w4, t_[0], nullptr, false);
      }
      conv3x3("fpn.lateral", fpn_a_[1], x_[1], h >> 3, w >> 3, p_[1], nullptr, false);
      cs = stream_;
    } else if (overlap3 && l == 2) {
      OCR_HIP(hipEventRecord(ev_x3_, stream_));
      OCR_HIP(hipStreamWaitEvent(side_stream_, ev_x3_, 0));
      cs = side_stream_;
      Extra up;
      up.store = STORE_PHASE;
      up.residual = p_[1];
      conv("fpn.upsampled", fpn_b_[1], x_[2], h >> 4, w >> 4, 1, p_[1], false, up);
      cs = stream_;
    } else if (overlap && (l == 1 || l == 2)) {
      // x_{l+1}... is ready: p2 (after layer2) / p3 (after layer3) and their bin_conv1 terms only need the trunk
      // features computed so far, so they go to the side stream and run next to the deeper layers
      hipEvent_t ev = l == 1 ? ev_x2_ : ev_x3_;
      OCR_HIP(hipEventRecord(ev, stream_));
      OCR_HIP(hipStreamWaitEvent(side_stream_, ev, 0));
      cs = side_stream_;
      fpn_level(l - 1);
      cs = stream_;
    }
  }
  // FPN laterals in5..in2 (model.rs:115-123), coarse to fine; each also emits the top-down sum
  // up2(in_{k+1}) + in_k that the out_k conv consumes (model.rs:126-137)
  conv("in5", in_[3], x_[3], h >> 5, w >> 5, 1, i_[3], false);
  if (fpn_composed_) {
    if (overlap_small) {
      fork([&] { conv("out5", out_[3], i_[3], h >> 5, w >> 5, 1, p_[3], false); });   // (side stream: the direct conv, no shared Winograd workspace)
      {
        Extra td;
        td.up_residual = i_[3];
        td.out2 = sum_[2];
        conv("in+topdown", in_[2], x_[2], h >> 4, w >> 4, 1, nullptr, false, td);
      }
      conv3x3("out", out_[2], sum_[2], h >> 4, w >> 4, p_[2], nullptr, false);
      join();
    } else {
      {
        Extra td;
        td.up_residual = i_[3];
        td.out2 = sum_[2];
        conv("in+topdown", in_[2], x_[2], h >> 4, w >> 4, 1, nullptr, false, td);
      }
      conv3x3("out", out_[2], sum_[2], h >> 4, w >> 4, p_[2], nullptr, false);
      if (out_[3].wino) conv3x3("out5", out_[3], i_[3], h >> 5, w >> 5, p_[3], nullptr, false);
      else conv("out5", out_[3], i_[3], h >> 5, w >> 5, 1, p_[3], false);
    }
    if (overlap) {
      OCR_HIP(hipEventRecord(ev_side_, side_stream_));
      OCR_HIP(hipStreamWaitEvent(stream_, ev_side_, 0));
    } else {
      fpn_level(0);
      fpn_level(1);
    }
  } else {
    for (int l = 2; l >= 0; --l) {
      Extra td;
      td.up_residual = i_[l + 1];
      td.out2 = sum_[l];
      conv("in+topdown", in_[l], x_[l], h >> (2 + l), w >> (2 + l), 1, l > 0 ? i_[l] : nullptr, false, td);
    }
    // p_k = out_k(up2(in_{k+1}) + in_k), p5 = out5(in5), model.rs:126-138
    for (int l = 0; l < 3; ++l) conv("out", out_[l], sum_[l], h >> (2 + l), w >> (2 + l), 1, p_[l], false);
    conv("out5", out_[3], i_[3], h >> 5, w >> 5, 1, p_[3], false);
  }
  // fuse = cat[p5 x8, p4 x4, p3 x2, p2]; bin_conv1 + bin_bn1 + relu, model.rs:140-145
  if (fpn_composed_ && bin_pyr_on_ && (!bf || fused_tail_)) {
    // one launch: per output phase (y mod 8, x mod 8) the taps that phase needs from p5, p4, p3 and p2
    Extra py;
    py.pyr4 = true;
    py.store = STORE_PHASE;
    py.f32_out = !bf;  // bf16 precision: the fused head reads bf16
    // split-bf16 and bf16 forms over p5, p4, p3: the phases that share their operand rows with a neighbour as 128-column tiles of phase
    // blocks, the four corner phases as a second, small launch (conv_igemm.hip, PYRG)
    auto pyramid3 = [&](const ConvW& cw, bool relu) {
      if (((!bf && split_bf16_ && cw.w_x3) || bf) && pyr_grouped_) {
        // (the two write disjoint phases of b1.  The corner launch - 4 x 100 tiles at 32 x 640 x 640, less than one round of the chip - on
        // the side stream beside the block launch: the same step, 4.557 / 4.547 against 4.551 / 4.555 ms on one box; not kept)
        py.pyr_group = 1;
        conv("bin_conv1.pyramid", cw, p_[3], h >> 5, w >> 5, 1, b1_, relu, py);
        py.pyr_group = 2;
        conv("bin_conv1.pyramid.corners", cw, p_[3], h >> 5, w >> 5, 1, b1_, relu, py);
        py.pyr_group = 0;
      } else {
        conv("bin_conv1.pyramid", cw, p_[3], h >> 5, w >> 5, 1, b1_, relu, py);
      }
    };
    if (overlap3) {
      // p2's term was computed on the side stream (into t_[0]): the phase launch over p5, p4, p3 adds it, the bias and the ReLU
      py.pyr_nsrc = 3;
      py.residual = t_[0];
      pyramid3(bin_pyr_, true);
    } else if (bf && bin_p2_.w_bf16_c64 && pyr_p2_direct_) {
      // bf16: the three upsampled sources in the phase launch, p2's 3x3 term on top as the patch-staged 64 -> 64 conv (bias + ReLU
      // there).  p2's nine taps are more than half of the phase launch's gathers (one pixel row per cell, tap and phase - the part
      // of that launch that costs); the direct kernel stages every p2 pixel once.  The partial sum passes through bf16 once more.
      py.pyr_nsrc = 3;
      ConvW up3 = bin_pyr_;
      up3.bias = nullptr;
      pyramid3(up3, false);
      ConvW p2 = bin_p2_;
      p2.bias = bin1_.bias;
      conv3x3("bin_conv1.p2", p2, p_[0], h4, w4, b1_, b1_, true);
    } else if (bf) {
      // all four sources in the one phase launch (112.8 GF instead of the gathered conv's 241.6), bias + ReLU in its epilogue
      conv("bin_conv1.pyramid", bin_pyr_, p_[3], h >> 5, w >> 5, 1, b1_, true, py);
    } else if (bin_p2_.wino43_fused) {
      // the three upsampled sources in the phase launch, p2's 3x3 term on top as a fused Winograd conv (+ bias, ReLU)
      py.pyr_nsrc = 3;
      ConvW up3 = bin_pyr_;
      up3.bias = nullptr;
      pyramid3(up3, false);
      conv3x3("bin_conv1.p2", bin_p2_, p_[0], h4, w4, b1_, b1_, true);
    } else {
      conv("bin_conv1.pyramid", bin_pyr_, p_[3], h >> 5, w >> 5, 1, b1_, true, py);
    }
  } else if (fpn_composed_ && !bf) {
    // the p2 quarter as a plain 3x3 conv and the three upsampled quarters as phase convs on their own grids;
    // partial sums live in b1.  (f32 only: in bf16 the f32 partial sums cost more HBM time than the skipped MFMA work
    // saves; that precision takes all four sources in the one phase launch above, or the single gathered conv)
    for (int l = 1; l <= 2; ++l) {  // p4, then p5 with bias + ReLU (the p2 and p3 terms are in b1 already)
      Extra up;
      up.store = STORE_PHASE;
      up.f32_out = true;
      up.residual = b1_;
      conv("bin_conv1.upsampled", bin_up_[l], p_[l + 1], h >> (3 + l), w >> (3 + l), 1, b1_, l == 2, up);
    }
  } else {
    Extra c4;
    c4.cat4 = true;
    conv("bin_conv1", bin1_, p_[3], h4, w4, 1, b1_, true, c4);
  }
  if (fused_tail_) {
    // bin_conv_tr1 + bias + bin_bn2 + relu + bin_conv_tr2 + bias + sigmoid (+ binarize) in one kernel:
    // the 64-channel H/2 x W/2 intermediate never reaches HBM.  model.rs:146-150
    rec.begin();
    const bool tail_x3 = !bf && split_bf16_ && tr1_.w_x3;
    launch_tail_fused(b1_, bf ? tr1_.w_bf16 : tail_x3 ? tr1_.w_x3 : static_cast<const void*>(tr1_.w), bf ? 1 : tail_x3 ? 2 : 0, tr1_.scale, tr1_.bias,
                      tr2_wt_, tr2_bias_, prob, bitmap, thresh, n, h4, w4, stream_);
    rec.end(tail_x3 ? "tail_x3_convt1_bn_relu_convt2_sigmoid" : "tail_convt1_bn_relu_convt2_sigmoid", 2.0 * n * h4 * w4 * 64.0 * 256 + 2.0 * n * (h / 2) * (w / 2) * 64 * 4,
            (double)n * h4 * w4 * 64 * (double)es + (double)n * h * w * 4);
  } else {
    // bin_conv_tr1 + bias + bin_bn2 + relu, model.rs:146-148
    {
      Extra sh;
      sh.store = STORE_SHUFFLE2;
      conv("bin_conv_tr1", tr1_, b1_, h4, w4, 1, tr1buf_, true, sh);
    }
    // bin_conv_tr2 + bias + sigmoid (+ binarize), model.rs:149-150
    rec.begin();
    launch_convt2_sigmoid(tr1buf_, tr2_w_, tr2_bias_, prob, bitmap, thresh, n, h / 2, w / 2, stream_);
    rec.end("convt2x2_sigmoid", 2.0 * n * (h / 2) * (w / 2) * 64 * 4,
            (double)n * (h / 2) * (w / 2) * 64 * 4 + (double)n * h * w * 4);
  }
  rec.finish();
}

const float* Detector::stage(int id, size_t* elems) const {
  if (ws_n_ == 0 || last_n_ == 0) fail(OCR_ERR_INVALID, "no forward has run yet");
  if (ws_bf16_) fail(OCR_ERR_INVALID, "stage read-back is f32 only (bf16 precision is active)");
  auto f = [](const char* p) { return reinterpret_cast<const float*>(p); };
  const size_t N = (size_t)last_n_;
  auto px = [&](int shift) { return N * (ws_h_ >> shift) * (ws_w_ >> shift); };
  if (id == 0) { *elems = px(2) * 64; return f(s_); }
  if (id >= 1 && id <= 4) { *elems = px(1 + id) * ((size_t)64 << (id - 1)); return f(x_[id - 1]); }
  if (id >= 5 && id <= 8 && !(id == 5 ? sum_[0] : i_[id - 5]))
    fail(OCR_ERR_INVALID, "stage %d is not materialised by the composed FPN (option fpn_unfused=1 keeps it)", id);
  if (id == 5) { *elems = px(2) * 256; return f(sum_[0]); }  // in2 only exists inside its top-down sum
  if (id >= 6 && id <= 8) { *elems = px(id - 3) * 256; return f(i_[id - 5]); }
  if (id >= 9 && id <= 12) { *elems = px(id - 7) * 64; return f(p_[id - 9]); }
  if (id == 13) { *elems = px(2) * 64; return b1_; }
  if (id == 14 && tr1buf_) { *elems = px(1) * 64; return tr1buf_; }
  fail(OCR_ERR_INVALID, "unknown stage %d", id);
}

// ---- host-memory entry points.  The reference hands CPU tensors to forward_t (text_detection/mod.rs:46-54); here the
// frames cross PCIe into a double-buffered device staging area on a copy stream while the previous piece computes.
// Pinned host memory (ocr_host_alloc) makes those copies asynchronous and full speed; pageable memory works too (the HIP
// runtime stages it itself and the call blocks for the duration of each copy).
void Detector::ensure_staging(int set, size_t in_bytes, size_t prob_elems) {
  if (set < 0 || set > 1) fail(OCR_ERR_INTERNAL, "staging set %d", set);
  Staging& st = stage_[set];
  if (!copy_stream_) {
    OCR_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    OCR_HIP(hipStreamCreateWithFlags(&out_stream_, hipStreamNonBlocking));
  }
  if (!st.ev_in[0])
    for (int i = 0; i < 2; ++i) {
      OCR_HIP(hipEventCreateWithFlags(&st.ev_in[i], hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&st.ev_fwd[i], hipEventDisableTiming));
      OCR_HIP(hipEventCreateWithFlags(&st.ev_out[i], hipEventDisableTiming));
    }
  if (staging_would_grow(set, in_bytes, prob_elems)) {
    // nothing may still read or write the slots being freed: the pipelined entry point has finished its pending batch (whose
    // map lives in a slot of set 1) before it comes here; the streams are drained for the copies and forwards in flight
    if (set == STAGE_PIPELINED && pending_.valid && pending_.prob && (pending_.prob == st.out[0] || pending_.prob == st.out[1]))
      fail(OCR_ERR_INTERNAL, "staging of the pipelined path grown under its pending batch");
    OCR_HIP(hipStreamSynchronize(stream_));
    OCR_HIP(hipStreamSynchronize(copy_stream_));
    OCR_HIP(hipStreamSynchronize(out_stream_));
    if (post_stream_) OCR_HIP(hipStreamSynchronize(post_stream_));
    for (int i = 0; i < 2; ++i) {
      if (st.in[i]) OCR_HIP(hipFree(st.in[i]));
      if (st.out[i]) OCR_HIP(hipFree(st.out[i]));
      st.in[i] = nullptr;
      st.out[i] = nullptr;
    }
    st.in_bytes = std::max(in_bytes, st.in_bytes);
    st.elems = std::max(prob_elems, st.elems);
    for (int i = 0; i < 2; ++i) {
      OCR_HIP(hipMalloc(&st.in[i], st.in_bytes));
      OCR_HIP(hipMalloc(reinterpret_cast<void**>(&st.out[i]), st.elems * 4));
    }
    st.uses = 0;
  }
}

// frames of the host batch -> staging slot (asynchronous when x is pinned); returns the device pointer and the event
// that marks their arrival.  The slot's previous user (two pieces ago) must have finished its forward: ev_fwd.
const void* Detector::stage_input(int set, int slot, const void* x_host, size_t bytes, hipEvent_t* arrived) {
  Staging& st = stage_[set];
  if (st.uses >= 2) OCR_HIP(hipStreamWaitEvent(copy_stream_, st.ev_fwd[slot], 0));
  OCR_HIP(hipMemcpyAsync(st.in[slot], x_host, bytes, hipMemcpyHostToDevice, copy_stream_));
  OCR_HIP(hipEventRecord(st.ev_in[slot], copy_stream_));
  *arrived = st.ev_in[slot];
  return st.in[slot];
}

void Detector::mark_before_forward() {
  if (!ev_before_fwd_) OCR_HIP(hipEventCreateWithFlags(&ev_before_fwd_, hipEventDisableTiming));
  OCR_HIP(hipEventRecord(ev_before_fwd_, stream_));
}

void Detector::forward_host(const void* x, int x_u8, int n, int h, int w, float* prob) {
  OCR_HIP(hipSetDevice(device_));
  const size_t es = x_u8 ? 1 : 4, frame = (size_t)h * w;
  // pieces of at least 8 frames, at most four of them: the copy in of piece i + 1 and the copy out of piece i - 1 run
  // beside the forward of piece i
  const int piece = std::max(8, (n + 3) / 4);
  ensure_staging(STAGE_FORWARD, (size_t)piece * frame * es, (size_t)piece * frame);
  Staging& st = stage_[STAGE_FORWARD];
  int k = 0;
  for (int b = 0; b < n; b += piece, ++k) {
    const int nb = std::min(piece, n - b), slot = k & 1;
    hipEvent_t arrived;
    const void* xd = stage_input(STAGE_FORWARD, slot, static_cast<const char*>(x) + (size_t)b * frame * es, (size_t)nb * frame * es, &arrived);
    if (st.uses >= 2) OCR_HIP(hipStreamWaitEvent(stream_, st.ev_out[slot], 0));   // the slot's previous map has left
    forward(xd, nb, h, w, st.out[slot], nullptr, 0.f, nullptr, x_u8, arrived);
    OCR_HIP(hipEventRecord(st.ev_fwd[slot], stream_));
    OCR_HIP(hipStreamWaitEvent(out_stream_, st.ev_fwd[slot], 0));
    OCR_HIP(hipMemcpyAsync(prob + (size_t)b * frame, st.out[slot], (size_t)nb * frame * 4, hipMemcpyDeviceToHost, out_stream_));
    OCR_HIP(hipEventRecord(st.ev_out[slot], out_stream_));
    ++st.uses;
  }
  OCR_HIP(hipStreamSynchronize(out_stream_));
  OCR_HIP(hipStreamSynchronize(stream_));
}

// ---------------------------------------------------------------------------

Recognizer::Recognizer(const void* blob, size_t bytes, int device) : device_(device) {
  check_device(device);
  WeightBlob wb(blob, bytes);
  OCR_HIP(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
  stream_ = own_stream_;
  arena_.reserve((size_t)8 << 20);
  auto vec = [&](const char* name, std::initializer_list<int> shape) {
    const TensorView& t = wb.get(name, shape);
    return std::vector<float>(t.data, t.data + t.count);
  };
  // char_recognition/model.rs:13-24; the two convs re-laid out as MFMA operand fragments (rec_net.hip)
  w_.c1f = arena_.upload(rec_conv1_fragments(vec("conv1.weight", {32, 1, 5, 5}).data()));
  w_.c1b = arena_.upload(vec("conv1.bias", {32}));
  w_.c2f = arena_.upload(rec_conv2_fragments(vec("conv2.weight", {64, 32, 5, 5}).data()));
  w_.c2b = arena_.upload(vec("conv2.bias", {64}));
  w_.f1w = arena_.upload(vec("fc1.weight", {512, 1024}));
  w_.f1s = arena_.upload(rec_fc1_small_weights(vec("fc1.weight", {512, 1024}).data()));
  w_.f1b = arena_.upload(vec("fc1.bias", {512}));
  std::vector<float> b2 = vec("fc2.bias", {62});
  b2.resize(64, 0.f);  // two padding columns: the kernel works on 32-column MFMA tiles
  w_.f2f = arena_.upload(rec_fc2_fragments(vec("fc2.weight", {62, 512}).data()));
  w_.f2s = arena_.upload(rec_fc2_small_fragments(vec("fc2.weight", {62, 512}).data()));
  w_.c2x = arena_.upload_u16(rec_conv2_small_x3_fragments(vec("conv2.weight", {64, 32, 5, 5}).data()));
  w_.f2b = arena_.upload(b2);
}

Recognizer::~Recognizer() {
  (void)hipSetDevice(device_);
  if (stream_ && stream_ != own_stream_) (void)hipStreamSynchronize(stream_);   // the caller's stream may still run a classify_async
  if (own_stream_) (void)hipStreamSynchronize(own_stream_);
  if (stage_) (void)hipFree(stage_);
  if (feat_) (void)hipFree(feat_);
  if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

void Recognizer::synchronize() {
  OCR_HIP(hipSetDevice(device_));
  OCR_HIP(hipStreamSynchronize(stream_));
}

// small_batch=0|1: whether batches of up to kRecSmallBatch crops take the latency-optimised kernels (default) or the same
// kernels as larger batches - then a crop's logits are bit-identical whatever the size of the batch it arrives in.
void Recognizer::set_options(const char* options) {
  if (!options) return;
  std::string s(options);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find_first_of(";,", pos);
    if (end == std::string::npos) end = s.size();
    std::string item = s.substr(pos, end - pos);
    pos = end + 1;
    item.erase(std::remove_if(item.begin(), item.end(), [](char c) { return c == ' ' || c == '\t'; }), item.end());
    if (item.empty()) continue;
    if (item == "small_batch=0") small_batch_ = false;
    else if (item == "small_batch=1") small_batch_ = true;
    else fail(OCR_ERR_INVALID, "unknown recogniser option '%s' (small_batch=0|1)", item.c_str());
  }
}

void Recognizer::ensure_workspace(int n) {
  if (n <= ws_cap_) return;
  OCR_HIP(hipStreamSynchronize(stream_));
  if (feat_) OCR_HIP(hipFree(feat_));
  feat_ = hid_ = nullptr;
  ws_cap_ = 0;
  const int cap = std::min(kChunk, std::max((n + 15) / 16 * 16, 256));  // whole 16-crop tiles (small-batch feat layout)
  OCR_HIP(hipMalloc(reinterpret_cast<void**>(&feat_), (size_t)cap * (1024 + 512) * sizeof(float)));
  hid_ = feat_ + (size_t)cap * 1024;
  ws_cap_ = cap;
}

void Recognizer::classify(const float* crops, int n, float* logits, int32_t* labels, double* probs,
                          std::vector<ProfileEntry>* prof) {
  if (!crops || n < 0) fail(OCR_ERR_INVALID, "rec: null crops or negative count");
  OCR_HIP(hipSetDevice(device_));
  for (int b = 0; b < n; b += kChunk) {
    const int nb = std::min(kChunk, n - b);
    ensure_workspace(nb);
    Recorder rec(prof, stream_);
    const bool small = small_batch_ && rec_small_batch(nb);
    if (small) {
      // configs[2]-sized batches: three launches built for latency (split-bf16 conv2, K-split fc1, 16-crop fc2 tiles)
      rec.begin();
      launch_rec_small(w_, crops + (size_t)b * 784, nb, feat_, hid_, nullptr, nullptr, nullptr, stream_, 0);
      rec.end("rec_conv_small_x3", 2.0 * nb * (576.0 * 32 * 26 + 64.0 * 64 * 800), (double)nb * (784 + 1024) * 4 + 13 * 64 * 4 + 25 * 2048 * 6);
      rec.begin();
      launch_rec_small(w_, nullptr, nb, feat_, hid_, nullptr, nullptr, nullptr, stream_, 1);
      rec.end("rec_fc1_ksplit", 2.0 * nb * 1024 * 512, 4.0 * ((double)nb * (1024 + 512) + 1024.0 * 512));
      rec.begin();
      launch_rec_small(w_, nullptr, nb, feat_, hid_, logits ? logits + (size_t)b * 62 : nullptr, labels ? labels + b : nullptr,
                       probs ? probs + b : nullptr, stream_, 2);
      rec.end("rec_fc2_small_softmax_top1", 2.0 * nb * 512 * 64, (double)nb * (512 * 4 + 12) + 64 * 512 * 4);
      rec.finish();
      continue;
    }
    rec.begin();
    launch_rec_conv(w_, crops + (size_t)b * 784, nb, feat_, stream_);
    rec.end(rec_crops_per_block(nb) == 2 ? "rec_conv<2>" : "rec_conv<4>",
            2.0 * nb * (576.0 * 32 * 26 + 64.0 * 64 * 800), (double)nb * (784 + 1024) * 4 + 13 * 64 * 4 + 25 * 2048 * 4);
    // fc1 + bias + ReLU as a plain GEMM over the batch: M = crops, K = Cin, N = Cout
    auto fc = [&](const char* name, const float* in, const float* wgt, const float* bias, int cin, int cout, bool relu, float* out) {
      ConvDesc d{};
      d.src[0] = in;
      d.src_mode = SRC_PLAIN;
      d.src_bytes = (size_t)nb * cin * 4;
      d.wgt = wgt;
      d.wgt_bytes = (size_t)cout * cin * 4;
      d.N = 1;
      d.Hin = d.Ho = 1;
      d.Win = d.Wo = nb;
      d.Cin = cin;
      d.Cout = cout;
      d.ks = 1;
      d.stride = 1;
      d.pad = 0;
      d.bias = bias;
      d.relu = relu ? 1 : 0;
      d.store_mode = STORE_NHWC;
      d.out = out;
      d.name = name;
      rec.begin();
      launch_conv_igemm(d, stream_);
      rec.end(name, 2.0 * nb * cin * cout, 4.0 * ((double)nb * (cin + cout) + (double)cin * cout));
    };
    fc("rec_fc1", feat_, w_.f1w, w_.f1b, 1024, 512, true, hid_);
    rec.begin();
    launch_rec_fc2_softmax(w_, hid_, nb, logits ? logits + (size_t)b * 62 : nullptr, labels ? labels + b : nullptr,
                           probs ? probs + b : nullptr, stream_);
    rec.end("rec_fc2_softmax_top1", 2.0 * nb * 512 * 64, (double)nb * (512 * 4 + 12) + 64 * 512 * 4);
    rec.finish();
  }
}

void Recognizer::forward_host(const float* crops, int n, float* logits, int32_t* labels, double* probs) {
  if (n <= 0) return;
  OCR_HIP(hipSetDevice(device_));
  const size_t b_in = (size_t)n * 784 * 4, b_lg = (size_t)n * 62 * 4, b_lab = (size_t)n * 4, b_pr = (size_t)n * 8;
  const size_t o_lg = (b_in + 255) / 256 * 256, o_pr = o_lg + (b_lg + 255) / 256 * 256, o_lab = o_pr + (b_pr + 255) / 256 * 256;
  const size_t total = o_lab + b_lab;
  if (total > stage_bytes_) {
    OCR_HIP(hipStreamSynchronize(stream_));
    if (stage_) OCR_HIP(hipFree(stage_));
    stage_ = nullptr;
    stage_bytes_ = 0;
    OCR_HIP(hipMalloc(&stage_, total));
    stage_bytes_ = total;
  }
  char* base = static_cast<char*>(stage_);
  OCR_HIP(hipMemcpyAsync(base, crops, b_in, hipMemcpyHostToDevice, stream_));
  classify(reinterpret_cast<float*>(base), n, reinterpret_cast<float*>(base + o_lg),
           reinterpret_cast<int32_t*>(base + o_lab), reinterpret_cast<double*>(base + o_pr));
  if (logits) OCR_HIP(hipMemcpyAsync(logits, base + o_lg, b_lg, hipMemcpyDeviceToHost, stream_));
  if (labels) OCR_HIP(hipMemcpyAsync(labels, base + o_lab, b_lab, hipMemcpyDeviceToHost, stream_));
  if (probs) OCR_HIP(hipMemcpyAsync(probs, base + o_pr, b_pr, hipMemcpyDeviceToHost, stream_));
  OCR_HIP(hipStreamSynchronize(stream_));
}

}  // namespace ocr
