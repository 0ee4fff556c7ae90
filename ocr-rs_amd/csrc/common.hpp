// Shared declarations of the MI355X-native OCR hot path (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ocr_amd.h"

namespace ocr {

// Error carried to the C boundary; never crosses it as an exception.
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define OCR_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      ::ocr::fail(OCR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

// ---- kernel launch descriptors ------------------------------------------------

// SRC_PYR4 (with STORE_PHASE, up 8): bin_conv1 over cat[up8(p5), up4(p4), up2(p3), p2] computed per output
// phase (y mod 8, x mod 8): a tile row is one cell of the p5 grid, its K dimension walks the 1-4 taps each
// upsampled level contributes at that phase and the 9 taps of p2; weights [64 phases][Cout][21 taps][64].
enum SrcMode { SRC_PLAIN = 0, SRC_CAT4 = 2, SRC_PYR4 = 3 };
// STORE_PHASE: the conv is the low-res form of "3x3 conv of a nearest-x-up upsampled tensor" (up = 2, 4, 8):
// ks = 2, up*up weight sets [phase = up a + b][Cout][2x2][Cin], output pixel (up i + a, up j + b) of a
// [N][up Ho][up Wo][Cout] tensor; the 2x2 window starts at row i-1 for a = 0 and at row i otherwise.
enum StoreMode { STORE_NHWC = 0, STORE_SHUFFLE2 = 1, STORE_PHASE = 2 };

// One convolution as an implicit GEMM: M = N*Ho*Wo pixels, N = Cout, K = ks*ks*Cin.
// Activations are NHWC f32, weights [Cout][ks*ks][Cin] f32.
struct ConvDesc {
  int in_bf16, out_bf16;  // element types: operands (src, wgt) and results (out, out2, residual, up_residual); 0 = f32
  // f32 conv on the bf16 matrix cores: every f32 operand is the exact sum of three bf16 terms (hi + mid + lo, 3 x 8
  // significant bits); six of the nine partial products (all but mid.lo, lo.mid, lo.lo: <= 2^-23 of the product) are
  // accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Activations stay f32 in HBM and are split in registers; wgt holds
  // split3_weights_tiled() of the f32 layout (three bf16 planes, wgt_bytes = 6 per weight).  f32 PLAIN / PYR4 convs only.
  int x3;
  const void* src[4];     // PLAIN: src[0]; CAT4: p5,p4,p3,p2 (all inside ONE allocation starting at src_base)
  const void* src_base;   // CAT4: start of the allocation holding the four sources (else unused)
  size_t src_bytes;       // bytes addressable from src[0] (PLAIN) / src_base (CAT4); must be < 2^31
  size_t wgt_bytes;
  int src_mode;
  int N, Hin, Win, Cin;   // logical input grid of this conv
  int Ho, Wo, Cout;
  int ks, stride, pad;
  int up;                 // STORE_PHASE: upsampling factor
  // batched GEMM (1x1 s1 PLAIN NHWC only, no residual / out2): `batch` independent problems of the same
  // shape, problem b reading src + b * M*Cin, wgt + b * Cout*Cin and writing out + b * M*Cout (elements);
  // src_bytes / wgt_bytes cover all of them.  0 or 1 = a single problem.
  int batch;
  int pyr_nsrc;           // SRC_PYR4: 0 / 4 = all four sources, 3 = p5, p4, p3 only (p2's term is computed elsewhere)
  int pyr_group;          // SRC_PYR4 over p5, p4, p3 (x3 or bf16 operands): 0 = all 64 phases, one 64-column tile each; 1 = the 60 phases that share their operand rows with a
                          // neighbour, as 128-column tiles of 2 x 2 / 1 x 2 / 2 x 1 phase blocks; 2 = the four corner phases (0 | 7, 0 | 7)
  int win;                // STORE_PHASE with up 2, Cout 64 (x3 or bf16 operands): the GEMM's rows are the (Hin + 1) x (Win + 1) 2 x 2 windows of the low-res grid and the four
                          // phases that read a window are the four 64-column groups of one 128-wide pair of tiles (same products, same order)
  int wide;               // x3: take the 256 x 128 persistent form (conv_x3w.hip) where it exists for the launch - same bits
  const void* wgt;
  const float* scale;     // per output column, may be null (then scale 1 / bias 0); always f32
  const float* bias;
  const void* residual;   // NHWC, shape of the output, may be null
  // second output: out2 = value + nearest_upsample_x2(up_residual) (FPN top-down sum,
  // model.rs:126-137); up_residual is [N][Ho/2][Wo/2][Cout].  out may be null when out2 is set.
  const void* up_residual;
  void* out2;
  int relu;
  int store_mode;
  void* out;
  const char* name;
};

// ctc.hip (extension, no reference counterpart): CTC greedy decode, one wave per crop; logits [n][t][c] -> labels [n][t] (-1 padded), lengths [n]
void launch_ctc_greedy(const float* logits_dev, int n, int t, int c, int blank, int32_t* labels_dev, int32_t* lengths_dev, hipStream_t s);
void launch_conv_igemm(const ConvDesc& d, hipStream_t s);
// conv_x3w.hip: the split-bf16 convs with NHWC stores and Cout a multiple of 128 as 256 x 128 tiles on one persistent workgroup per CU
// (bit-identical to conv_igemm's 128-wide split-bf16 tiles); `cus` = CUs of the device
bool conv_x3_wide_applicable(const ConvDesc& d);
void launch_conv_x3_wide(const ConvDesc& d, int cus, hipStream_t s);
// hi / mid / lo bf16 planes of an f32 weight array whose rows are multiples of 16 long ([3][count] bf16; inside every
// aligned group of 16 the k order is the one the split-bf16 kernel's A fragments have: 0-3, 8-11, 4-7, 12-15)
std::vector<uint16_t> split3_weights(const float* w, size_t count);
// ... and the form conv_igemm's split-bf16 kernels read: the same planes cut into blocks of 16 rows x one K-step (32 k, 64 bytes per
// row), each block stored as the 1 KB LDS image one DMA instruction writes (row r, 16-byte slot s = chunk s ^ ((r >> 2) & 3)), blocks
// ordered [row / 16][K-step][plane]: `wrow` = weights per row (taps x Cin), rows a multiple of 16
std::vector<uint16_t> split3_weights_tiled(const float* w, size_t count, int wrow);
const char* conv_igemm_kernel_name(const ConvDesc& d);
void set_conv_tile_override(int t);  // tuning aid: 0 = automatic
void set_conv_debug(int d);          // ablation bits, effective in -DIGEMM_DEBUG builds only

// stem: conv7x7 s2 p3 (Cin=1) + BN + ReLU + maxpool3x3 s2 p1 -> NHWC 64 channels at H/4
// x: N x H x W frames, f32 or (x_u8 != 0) u8 raw luma - the reference's u8 image converted to f32 without scaling
void launch_stem(const void* x, int x_u8, const float* w49x64, const float* scale, const float* bias,
                 void* out, int out_bf16, int N, int H, int W, hipStream_t s);
// bf16 precision: the same stem with conv1 on v_mfma_f32_32x32x16_bf16 (weights as fragments from stem_bf16_fragments)
std::vector<uint16_t> stem_bf16_fragments(const float* w64x49);
void launch_stem_bf16(const void* x, int x_u8, const void* wfrag, const float* scale, const float* bias, void* out, int N, int H, int W,
                      hipStream_t s);
// f32 precision on the bf16 matrix cores (mfma=split_bf16): frame and weights as three bf16 terms each, six partial products,
// f32 accumulate and f32 output (weights from stem_x3_fragments)
std::vector<uint16_t> stem_x3_fragments(const float* w64x49);
void launch_stem_x3(const void* x, int x_u8, const void* wfrag3, const float* scale, const float* bias, float* out, int N, int H, int W,
                    hipStream_t s);
// bf16 precision: 3x3 s1 p1 conv 64 -> 64 with the input patch staged once in LDS and register-resident weights
// (conv3x3_bf16_c64.hip); wfrag from conv3x3_bf16_c64_fragments([Cout 64][9][Cin 64] f32)
std::vector<uint16_t> conv3x3_bf16_c64_fragments(const float* ohwi);
void launch_conv3x3_bf16_c64(const void* x, const void* wfrag, const float* scale, const float* bias, const void* residual, int relu,
                             void* y, int N, int H, int W, int num_cus, hipStream_t s);
// bf16 precision: one BasicBlock 64 -> 64 (conv3x3 + BN + ReLU, conv3x3 + BN, + x, ReLU) as one launch, the activation between the two
// convs held in LDS (basic_block_bf16_c64.hip); fragments of both convs from conv3x3_bf16_c64_fragments.  Bit-identical to two
// launch_conv3x3_bf16_c64 calls.  num_cus sizes the persistent grid (one workgroup per CU)
bool basic_block_bf16_c64_applicable(int N, int H, int W);
void launch_basic_block_bf16_c64(const void* x, const void* wfrag1, const float* scale1, const float* bias1, const void* wfrag2, const float* scale2,
                                 const float* bias2, void* y, int N, int H, int W, int num_cus, hipStream_t s);
// Winograd F(m x m, 3x3) transforms, m = 2 or 4, around a batched GEMM of (m+2)^2 problems (winograd.hip): 3x3 s1 p1 convs
// of the deep, small-grid layers.  x: [N][H][W][C] f32 -> v: [(m+2)^2][T][C], T = N * ceil(H/m) * ceil(W/m) tiles (zero
// padding and ragged sizes handled here); mm: [(m+2)^2][T][K] -> y: [N][H][W][K] with folded BN, residual and ReLU.
void launch_winograd_input(const float* x, float* v, int N, int H, int W, int C, int m, hipStream_t s);
// F(4x4): output transform of one conv (+ BN, residual, ReLU; y may be null: nothing else reads the activation) and input transform of the next in one launch
bool winograd43_out_in_fits(int H, int W, int K);
void launch_winograd43_out_in(const float* mm, const float* scale, const float* bias, const float* residual, int relu, float* y, float* v, int N, int H, int W,
                              int K, hipStream_t s);
void launch_winograd_output(const float* mm, const float* scale, const float* bias, const float* residual, int relu,
                            float* y, int N, int H, int W, int K, int m, hipStream_t s);
// The same conv with both transforms fused into the GEMM kernel, F(4x4,3x3) (winograd43_fused.hip): C = 64, 128 or 256, K a multiple of 64.  ufrag: winograd43_fragments(winograd_weights(.., 4)).
std::vector<float> winograd43_fragments(const std::vector<float>& u, int cout, int cin);
void launch_winograd43_fused(const float* x, const float* ufrag, const float* scale, const float* bias, const float* residual,
                             int relu, float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s);
// ... and with its GEMMs on the bf16 matrix cores, f32 operands as three bf16 terms each (winograd43_x3.hip; mfma=split_bf16):
// two pixel blocks per workgroup, one workgroup per CU.  ufrag: winograd43_x3_fragments(winograd_weights(.., 4)).
std::vector<uint16_t> winograd43_x3_fragments(const std::vector<float>& u, int cout, int cin);
void launch_winograd43_x3(const float* x, const void* ufrag, const float* scale, const float* bias, const float* residual,
                          int relu, float* y, int N, int H, int W, int C, int K, int num_cus, hipStream_t s);
// tail: convT2x2 s2 64->1 + bias + sigmoid (+ optional binarize)
void launch_convt2_sigmoid(const float* in, const float* w4x64, float bias, float* prob,
                           uint8_t* bitmap, float thresh, int N, int H2, int W2, hipStream_t s);
// fused head: convT1 + bias + BN + ReLU + convT2 + bias + sigmoid (+ binarize), tail_fused.hip
// y [M][64] and wt1 [4][64][64] are f32, or bf16 when bf16 != 0 (bf16 MFMA, f32 accumulate and f32 epilogue)
void launch_tail_fused(const void* y, const void* wt1, int bf16, const float* s4, const float* b4, const float* w2t, float bias2,
                       float* prob, uint8_t* bitmap, float thresh, int N, int h4, int w4, hipStream_t s);
void launch_binarize(const float* prob, uint8_t* bitmap, float thresh, size_t n, hipStream_t s);
// the same as one bit per pixel, every image packed on its own (bit i & 31 of word i >> 5 for its pixel i, padded to
// binarize_pack_words(px) 32-bit words): what the host contour tracer reads
size_t binarize_pack_words(size_t px_per_image);
void launch_binarize_pack(const float* prob, uint32_t* bits, float thresh, int n_images, size_t px_per_image, hipStream_t s);
// contours.hip: Suzuki-Abe border following of the packed bit images on the device, one wave per image (maps whose bit image and
// two label planes fit a CU's LDS: contour_trace_fits).  pts [n][cap] (y << 16 | x), starts [n][maxc + 1], hdr [n][4] =
// {contours, points, status, 0}; pts_packed / lens_packed: the status-0 images' points and contour lengths, densely in image order
bool contour_trace_fits(int h, int w);
// spec: contour_spec_bytes(n) bytes of scratch for the parallel form (plausible starts walked by a lane each, the raster scan only
// replays the label tests); sequential != 0 or spec == nullptr or h > 1024: the one-wave-per-image form
size_t contour_spec_bytes(int n);
void launch_contour_trace(const uint32_t* bits, size_t words_per_image, int n, int h, int w, uint32_t* pts, int cap, int* starts, int maxc, int* hdr,
                          uint32_t* pts_packed, int* lens_packed, void* spec, int sequential, hipStream_t s);

// preprocess_image (image_ops.rs:188-220): Triangle resize + luma + zero pad, preprocess.hip
void resize_dimensions(int width, int height, int nwidth, int nheight, int* ow, int* oh);
size_t preprocess_scratch_bytes(int w, int h, int W, int H);
void launch_preprocess(const unsigned char* rgba_dev, int w, int h, int W, int H, unsigned char* gray_dev, float* gray_f32_dev,
                       void* scratch, size_t scratch_bytes, double* adj_xy, hipStream_t s);

// crop extraction detect -> recognise (build-defined rule, oracle/crop_oracle.py), crops.hip
struct CropBox {
  int frame;
  float x0, y0, x1, y1;
};
void launch_crops(const float* frames_dev, int H, int W, const CropBox* boxes_dev, int n_boxes, float* crops_dev, hipStream_t s);

// recognition net (rec_net.hip): conv1 + pool + conv2 + pool on the matrix cores -> feat [n][1024]; fc1 runs as a
// conv_igemm 1x1 GEMM over the whole batch; fc2 + softmax(f64) + top-1 in one kernel
struct RecWeights {
  const float *c1f, *c1b;   // conv1 as MFMA fragments [13][64], bias [32]
  const float *c2f, *c2b;   // conv2 as MFMA fragments [25][2][4][64][4], bias [64]
  const float *f1w, *f1b;   // fc1 [512][1024], bias [512]
  const float* f1s;         // fc1 in the operand order of the small-batch kernel (rec_fc1_small_weights)
  const float *f2f, *f2b;   // fc2 as MFMA fragments [2][64][64][4] (rows 62, 63 zero), bias [64]
  const void* c2x;          // conv2 as three bf16 fragment sets for the split-bf16 small-batch kernel (rec_conv2_small_x3_fragments)
  const float* f2s;         // fc2 in the fragment order of rec_fc2_small_kernel (rec_fc2_small_fragments)
};
std::vector<float> rec_conv1_fragments(const float* w_32x25);
std::vector<float> rec_conv2_fragments(const float* w_64x32x25);
int rec_crops_per_block(int n);
std::vector<float> rec_fc1_small_weights(const float* w_512x1024);
std::vector<uint16_t> rec_conv2_small_x3_fragments(const float* w_64x32x25);
std::vector<float> rec_fc2_small_fragments(const float* w_62x512);
// the small-batch pass, stage 0 conv (split-bf16 conv2), 1 fc1 (K split over waves), 2 fc2 + softmax + top-1
void launch_rec_small(const RecWeights& w, const float* crops, int n, float* feat, float* hid, float* logits62, int32_t* labels,
                      double* probs, hipStream_t s, int stage);
bool rec_small_batch(int n);   // the latency-optimised kernels (16x16x4 MFMA chains) take batches up to 1024 crops
void launch_rec_conv(const RecWeights& w, const float* crops, int n, float* feat, hipStream_t s);
std::vector<float> rec_fc2_fragments(const float* w_62x512);
void launch_rec_fc2_softmax(const RecWeights& w, const float* hid, int n, float* logits62, int32_t* labels, double* probs, hipStream_t s);

// tch VarStore archives (varstore.cpp): `vs.load(file)` without libtorch
struct NamedTensor {
  std::string name;
  std::vector<int> dims;
  std::vector<float> data;  // f32, row-major
};
std::vector<NamedTensor> read_varstore(const char* path);
void rename_tch_rec(std::vector<NamedTensor>& tensors);
std::vector<uint8_t> pack_ocrw(const std::vector<NamedTensor>& tensors);
// kind: 0 = names as stored, 1 = detector (same), 2 = recogniser (tch's name__N leafs mapped by shape)
std::vector<uint8_t> varstore_to_blob(const char* path, int kind);

// box score: masked mean of prob over rasterised polygons (metrics.rs:150-184)
struct BoxScoreJob {
  int image;      // batch index
  int pt_offset;  // first point in the shared point array (x,y pairs, map coordinates)
  int n_pts;
  int min_x, min_y, bw, bh;  // mask canvas = clamped bounding box (metrics.rs:151-166)
};
constexpr int kBoxScoreMaxPts = 2048;
void launch_box_scores(const float* prob, int H, int W, const BoxScoreJob* jobs_dev,
                       const int32_t* pts_xy_dev, int n_jobs, double* sums_dev, double* counts_dev,
                       hipStream_t s);

// the same with the job count on the device (candidates.hip's totals[0]): `grid` workgroups walk the list
void launch_box_scores_counted(const float* prob, int H, int W, const BoxScoreJob* jobs_dev, const int32_t* pts_xy_dev, const int* n_jobs_dev, int grid,
                               double* sums_dev, double* counts_dev, hipStream_t s);
// candidates.hip: device contours (contours.hip's hdr / pts / starts) -> arc length, Douglas-Peucker, >= 4 points -> the batch's
// box-score jobs and points in image / contour order.  totals = {jobs, points, overflow}; images with a tracer status contribute nothing
size_t candidates_scratch_bytes(int n, int cap, int maxc);
void launch_candidates(const int* hdr, const uint32_t* pts, int cap, const int* starts, int maxc, int n, int h, int w, void* scratch, BoxScoreJob* jobs,
                       int max_jobs, int32_t* pts_xy, int max_pts, int* tot /* [n][2] = {candidates, points} or -1 -1: the host takes the image */,
                       int* totals, hipStream_t s);

// unclip behind the box score (unclip.hip): score threshold, miter offset, the union's simple-ring case, min-size test, adjustment -
// one lane per candidate.  status: UNCLIP_DROP (filtered out), UNCLIP_KEEP (out_len[j] adjusted points at out_xy + 6 * pt_offset),
// UNCLIP_HOST (the host finishes this one with postproc_geom.cpp: non-simple ring, squared-off corner, short side within 3 px of min_size, ...)
struct UnclipParams {
  double box_thresh, unclip_ratio, min_size;
};
enum { UNCLIP_DROP = 0, UNCLIP_KEEP = 1, UNCLIP_HOST = 2 };
constexpr int kUnclipMaxPts = 256;   // candidates with more vertices go to the host
size_t unclip_work_bytes(size_t total_pts, int n_jobs);
// n_jobs_dev: optional device-side job count (the launch covers n_jobs lanes, lanes beyond *n_jobs_dev exit)
void launch_unclip(const BoxScoreJob* jobs_dev, const int32_t* pts_xy_dev, const int* n_jobs_dev, int n_jobs, size_t total_pts, const double* sums_dev,
                   const double* counts_dev, const double* adj_dev, const UnclipParams& prm, void* work_dev, uint32_t* out_xy_dev,
                   int32_t* out_len_dev, int32_t* status_dev, hipStream_t s);

}  // namespace ocr
