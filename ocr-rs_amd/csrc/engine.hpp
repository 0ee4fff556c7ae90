// Detector / recogniser engines behind the C ABI (one GPU + one stream each).
#pragma once
#include <string>
#include <vector>

#include <memory>

#include "common.hpp"
#include "weights_blob.hpp"

namespace ocr {

class ThreadPool;

struct ProfileEntry {
  const char* name;
  float ms;
  double flops;
  double bytes;
};

class DeviceArena {  // bump allocator over one hipMalloc (weights)
 public:
  ~DeviceArena();
  void reserve(size_t bytes);
  float* upload(const std::vector<float>& host);
  void* upload_u16(const std::vector<uint16_t>& host);
 private:
  char* base_ = nullptr;
  size_t cap_ = 0, used_ = 0;
};

// U = G g G^T of a 3x3 conv given as [Cout][3x3][Cin], laid out [(m+2)^2][Cout][Cin] for F(m x m, 3x3), m = 2 or 4 (winograd.hip)
std::vector<float> winograd_weights(const float* ohwi, int cout, int cin, int m = 2);

struct ConvW {
  float* w = nullptr;      // [Cout][ks*ks][Cin]
  size_t w_bytes = 0;
  void* w_bf16 = nullptr;  // same layout in bf16 (filled by Detector::set_precision)
  void* w_x3 = nullptr;    // the same weights as three bf16 planes hi / mid / lo, tiled for the DMA (split3_weights_tiled): the f32 conv on the bf16 matrix cores
  void* wino_x3 = nullptr; // ... of the Winograd weights `wino`
  void* w_bf16_c64 = nullptr;  // 3x3 64 -> 64 convs: bf16 MFMA fragments for conv3x3_bf16_c64.hip
  std::vector<float> host; // the f32 layout, kept for the bf16 conversion
  std::vector<float> host_scale;  // folded batch norm scale (empty: none), kept for weight composition
  float* wino = nullptr;   // Winograd F(2x2,3x3) weights U = G g G^T as [16][Cout][Cin] (3x3 s1 convs of the deep layers)
  size_t wino_bytes = 0;
  int wino_tile = 2;       // ... or F(4x4,3x3), [36][Cout][Cin], where the option winograd43 asks for it
  float* wino43_fused = nullptr;  // F(4x4,3x3) weights as MFMA B fragments for winograd43_fused.hip (Cin 64 / 128)
  void* wino43_x3 = nullptr;      // ... as three bf16 planes of B fragments for winograd43_x3.hip (mfma=split_bf16)
  int up = 0;              // STORE_PHASE convs: upsampling factor (weights hold up*up phase sets)
  float* scale = nullptr;  // folded eval batch norm, may stay null
  float* bias = nullptr;
  int cin = 0, cout = 0, ks = 0;
};

class Detector {
 public:
  // options: "key=value;..." (include/ocr_amd.h, ocr_det_create_with_options) or null for the defaults
  Detector(const void* blob, size_t bytes, int device, const char* options = nullptr);
  ~Detector();
  void set_stream(hipStream_t s) { stream_ = s ? s : own_stream_; }
  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }
  void synchronize();
  // 0: f32 everywhere (default).  1: trunk / FPN activations and conv weights in bf16, f32 accumulate,
  // f32 folded batch norm, f32 probability head (BASELINE config 5).
  void set_precision(int precision);
  int precision() const { return bf16_ ? 1 : 0; }
  // device pointers; enqueues on stream().  prof != null -> per-launch events.
  // x: N x 1 x H x W frames, f32 or (x_u8 != 0) u8 raw luma; wait_for: an event the first launch waits for (the
  // copy that brings x in), may be null
  void forward(const void* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh,
               std::vector<ProfileEntry>* prof, int x_u8 = 0, hipEvent_t wait_for = nullptr);
  // host tensors in and out, blocking; copies and forward pipelined over pieces of the batch
  void forward_host(const void* x, int x_u8, int n, int h, int w, float* prob);
  // device staging of the host-memory entry points: two input slots and two map slots per SET, a copy-in and a copy-out stream.
  // Set 0 belongs to forward_host, set 1 to ocr_det_detect_pipelined_host, whose pending batch keeps its map in a slot across
  // calls: neither a blocking host forward in between nor the other set's growth can touch it.  Growing a set frees its
  // slots: the pipelined entry point finishes its pending batch first (staging_would_grow).
  enum { STAGE_FORWARD = 0, STAGE_PIPELINED = 1 };
  bool staging_would_grow(int set, size_t in_bytes, size_t prob_elems) const {
    return in_bytes > stage_[set].in_bytes || prob_elems > stage_[set].elems;
  }
  void ensure_staging(int set, size_t in_bytes, size_t prob_elems);
  const void* stage_input(int set, int slot, const void* x_host, size_t bytes, hipEvent_t* arrived);
  float* stage_prob(int set, int slot) { return stage_[set].out[slot]; }
  hipStream_t out_stream() { return out_stream_; }
  hipEvent_t forward_done_event(int set, int slot) { return stage_[set].ev_fwd[slot]; }
  int next_stage_slot(int set) { return (int)(stage_[set].uses & 1); }
  void stage_used(int set) { ++stage_[set].uses; }
  // ocr_extract_crops beside a pipelined forward: everything queued on the handle's stream BEFORE that forward (recorded by the
  // pipelined entry points just ahead of it) - whatever produced the frames of the batch that has come back
  void mark_before_forward();
  hipEvent_t before_forward_event() const { return ev_before_fwd_; }
  int post_threads() const;      // host threads of the post-processing stages (option post_threads, default min(16, CPU share))
  // contours on the device (contours.hip)?  option device_contours=1; off by default: a wave follows a border at about the speed of
  // ONE host core per batch (measured), so it pays only where no host core can be spared
  int device_contours() const;   // 0 off, 1 parallel form, 2 one wave per image; option auto (default) = 1 when the pool has at most four threads
  // unclip (score threshold, miter offset, simple-ring union, min-size test, adjustment) on the device behind the box score
  // (unclip.hip; option device_unclip=0 keeps all of it on the host pool)
  bool device_unclip() const { return device_unclip_ != 0; }
  bool device_unclip_always() const { return device_unclip_ == 2; }   // option device_unclip=2: also for a handful of polygons (tests)
  // with device contours: Douglas-Peucker and the box-score job list on the device too (candidates.hip; option device_polygons=0
  // brings the contours back and leaves them to the host pool)
  bool device_polygons() const { return device_polygons_; }
  // growable device scratch for post-processing; slot 0: map copy + bitmap, slot 1: jobs / results, slot 2: device contours, slot 3: the same for the batch a pipelined call left pending, slot 4: that batch's polygon chain.
  // Growing a slot invalidates only that slot's previous contents.
  void* scratch(int slot, size_t bytes);
  // growable PINNED host memory where the polygon chain's results land (one buffer per handle: a call collects before it returns)
  // where the polygon chain of this handle's post-processing calls ran, cumulative (ocr_det_post_stats): [0] images traced on the device,
  // [1] images traced on the host (device_contours off, a map the tracer does not take, an image it gave up), [2] candidates the device
  // unclip settled, [3] candidates the host finished, [4] images whose whole chain stayed on the device, [5] calls
  long long post_stats[6] = {0, 0, 0, 0, 0, 0};
  void* host_scratch(size_t bytes);
  // pinned block for the adjust values of the batch a pipelined call leaves pending (its chain is queued with a really asynchronous upload)
  void* host_adj(size_t bytes);
  // host threads of the post-processing stages (created on first use, one image per task)
  ThreadPool& pool();
  // ocr_det_detect_pipelined: the batch whose forward is in flight and whose post-processing is still owed
  struct Pending {
    bool valid = false;
    const float* prob = nullptr;
    float* prob_host = nullptr;   // host-memory variant: where the caller wants the map as well (may be null)
    int n = 0, h = 0, w = 0;
    std::vector<double> adj;
    ocr_postproc_params_t params{};
    hipEvent_t event = nullptr;
    // device_contours: the bit images and contours of this batch were requested on the post-processing stream when the batch was
    // queued (scratch slot 3, layout of api.hip::ContourBuffers): the call that brings the polygons back only reads them
    bool pretraced = false;
    bool prechained = false;   // ... and so were Douglas-Peucker, box scores and unclip behind them (scratch slot 4, api.hip::ChainBuffers)
  };
  bool has_pending() const { return pending_.valid; }
  Pending& pending() { return pending_; }
  Pending swap_pending(Pending& next) {
    Pending prev = std::move(pending_);
    pending_ = std::move(next);
    return prev;
  }
  hipStream_t post_stream();     // second stream: post-processing kernels and copies next to the following forward
  hipStream_t trace_stream();    // third stream: the device contours of the pending batch (device_contours), behind its forward
  hipEvent_t trace_done_event(); // ... and the event that marks them done
  hipEvent_t pipeline_event();   // alternating pair of events marking the end of a pipelined forward
  // test hook: NHWC intermediate of the last forward (0 stem, 1-4 layer1-4, 5-8 in2-5,
  // 9-12 p2-p5 (before upsampling), 13 bin_conv1, 14 bin_conv_tr1)
  const float* stage(int id, size_t* elems) const;

 private:
  void ensure_workspace(int n, int h, int w);
  void free_workspace();
  ConvW make_conv(const WeightBlob& wb, const std::string& wname, const std::string& bn, int cout, int cin, int ks);

  void parse_options(const char* options);
  int device_;
  int num_cus_ = 256;      // multiProcessorCount of the device: sizes the persistent grids
  bool opt_bf16_ = false;  // option precision=bf16
  hipStream_t own_stream_ = nullptr, stream_ = nullptr;
  // Optional second stream (measured: no gain - the launches are MFMA / power bound).
  // overlap=1: the small independent launches side by side (the 1x1 s2 downsample next to its block's
  // 3x3 s2 conv1, out5 next to in4 / out4); =2: also the FPN branch (p2, p3) next to layer3 / layer4.
  // forward_profile always runs one stream (clean per-launch timing).
  int overlap_ = 3;        // 3 (default): the FPN's fused-Winograd launches and bin_conv1's p2 term beside layer3 / layer4 (f32 precision, default
                           // engine; anything else falls back to the one-stream schedule)
  int w43_side_cus_ = 0;   // ... of the fused Winograd launches that go to the side stream (overlap >= 2): room for the main stream's workgroups beside them
  int w43_cus_ = 0;        // option w43_cus (tuning): size the fused Winograd kernels' persistent grids for this many CUs (0 = the device's)
  hipStream_t side_stream_ = nullptr;
  hipEvent_t ev_x1_ = nullptr, ev_x2_ = nullptr, ev_x3_ = nullptr, ev_side_ = nullptr, ev_fork_ = nullptr, ev_join_ = nullptr;
  DeviceArena arena_;
  float *stem_w_ = nullptr, *stem_scale_ = nullptr, *stem_bias_ = nullptr;
  std::vector<float> stem_w_host_;  // conv1 [64][49], kept for the bf16 fragments
  void* stem_wb_ = nullptr;         // conv1 as bf16 MFMA fragments (stem_bf16_fragments), filled by set_precision
  void* stem_wx3_ = nullptr;        // conv1 as three bf16 fragment sets (stem_x3_fragments): mfma=split_bf16
  ConvW layer_[4][2][2];  // [layer][block][conv1|conv2]
  ConvW down_[4];         // [layer] (layer 0 unused)
  ConvW in_[4];           // in2..in5
  // p2 / p3 with the lateral folded into the 3x3 conv (both linear, nothing in between):
  //   out_k(up2(in_{k+1}(x_{k+1})) + in_k(x_k)) = A_k * x_k + B_k *' x_{k+1}
  // A_k = out_k o in_k (3x3, C_k -> 64); B_k = out_k o up2 o in_{k+1} as four 2x2 phase convs on the
  // low-res grid (C_{k+1} -> 64).  [0] = p2, [1] = p3.
  ConvW fpn_a_[2], fpn_b_[2];
  // bin_conv1 over cat[up8(p5), up4(p4), up2(p3), p2] as four terms accumulated in its f32 output: phase
  // convs on the low-res grids of p5 / p4 / p3 ([0] = p3 x2, [1] = p4 x4, [2] = p5 x8) and a plain 3x3 conv
  // of p2 that adds the bias and applies the ReLU; bin_bn1's scale is folded into all four weight sets.
  ConvW bin_up_[3], bin_p2_;
  // ... or all four terms in ONE launch (SRC_PYR4): per output phase (y mod 8, x mod 8) a weight row of 21 tap
  // slots (4 + 4 + 4 for p5, p4, p3 and 9 for p2) x 64 channels; partial sums stay in the accumulators.
  // option bin_pyr=0 keeps the four-launch form.
  ConvW bin_pyr_;
  bool bin_pyr_on_ = true;
  bool x3_wide_ = false;        // split-bf16 convs with NHWC stores and Cout % 128 == 0 on the 256 x 128 persistent form (conv_x3w.hip); 0: conv_igemm's 128-wide tiles.  Same bits
  bool bf16_block_fuse_ = true;  // bf16 precision: layer1's BasicBlocks as one launch each (basic_block_bf16_c64.hip); 0: two conv3x3_bf16_c64 launches.  Same bits
  bool phase_windows_ = true;   // split-bf16 up-2 phase convs indexed by 2 x 2 windows: one operand tile for the four phases (0: one 64-column tile per phase)
  bool pyr_grouped_ = true;     // split-bf16 / bf16 bin_conv1 over p5..p3: phase blocks as 128-column tiles + the corner phases (0: one 64-column tile per phase)
  bool pyr_p2_direct_ = true;   // bf16 precision: p2's 3x3 term of bin_conv1 as the patch-staged 64 -> 64 conv instead of nine taps of the phase launch
  ConvW finish_composed(std::vector<float>&& t, int cout, int cin, int ks);
  ConvW compose_lateral(const ConvW& out, const ConvW& in);
  ConvW compose_upsampled(const ConvW& out, const ConvW& in_up);
  ConvW phase_conv(const std::vector<double>& taps, int cout, int cin, int up);
  bool fpn_composed_ = true;
  // Winograd for 3x3 s1 trunk convs with Cin >= this (f32 precision only); option winograd=0 disables, =<cin> overrides
  int winograd_min_cin_ = 256;
  int winograd43_min_cin_ = 256;      // unfused Winograd layers with at least this many channels use F(4x4,3x3)
  float *wino_v_ = nullptr, *wino_m_ = nullptr;  // [(m+2)^2][T][C] and [(m+2)^2][T][K] scratch of the layer in flight
  void add_winograd_weights(ConvW& cw);
  void add_winograd_fused_weights(ConvW& cw);
  bool winograd43_x3_ = false;   // option winograd43_x3=1: the fused F(4x4,3x3) convs on the bf16 matrix cores too (winograd43_x3.hip; measured
                                 // slower than the f32-MFMA kernel, DESIGN.md section 3 - kept selectable for A/B)
  bool out4_fused_ = false;      // option out4_fused=1: out4 (256 -> 64 at H/16) on the fused F(4x4,3x3) kernel (0.102 ms) instead of the
                                 // unfused path of layer3 / layer4 (transform + split-bf16 GEMM + transform: 0.084 ms)
  bool transform_fuse_ = false;  // option transform_fuse=1: inside block 1 of layer3 / layer4 the output transform of conv1 and the input transform of conv2 as one
                                 // launch (winograd.hip; bit-identical, measured: no gain - DESIGN.md section 9 - so off)
  bool winograd_fused_ = true;   // option winograd_fused=0: direct / unfused-Winograd convs instead of the fused F(4x4,3x3) kernel
  // option mfma=split_bf16 (default) | f32: the MFMA-bound f32 convs without a Winograd kernel of their own (stride-2 3x3,
  // composed FPN phase convs, bin_conv1 over the pyramid, the 36 Winograd GEMMs of layer3 / layer4) run on the bf16 matrix
  // cores from operands split into three bf16 terms, six partial products, f32 accumulate (conv_igemm.hip, X3): f32-level
  // accuracy (profiles/r03_bf16x3_accuracy.txt) at up to 2.67x the f32 MFMA rate.  f32 keeps every conv on v_mfma_f32_32x32x2_f32.
  bool split_bf16_ = true;
  void add_split_weights(ConvW& cw, int wrow = 0);
  ConvW out_[4];          // out2..out5
  ConvW bin1_, tr1_;
  float* tr2_w_ = nullptr;
  float* tr2_wt_ = nullptr;   // [64 co][4 u] for the fused head
  bool fused_tail_ = true;    // option tail_unfused=1 keeps the two-kernel head (A/B and debugging)
  float tr2_bias_ = 0.f;

  bool bf16_ = false;
  int ws_n_ = 0, ws_h_ = 0, ws_w_ = 0;  // workspace capacity (frames) and frame size
  int last_n_ = 0;                      // frames of the most recent forward_chunk (stage read-back)
  bool ws_bf16_ = false;
  std::vector<void*> ws_allocs_;
  // activations are f32 or bf16 depending on the precision (b1_, tr1buf_ are always f32)
  char *s_ = nullptr, *t_[4] = {}, *a_[4] = {}, *d_[4] = {}, *x_[4] = {};
  char *i_[4] = {}, *sum_[3] = {}, *p_[4] = {}, *pcat_ = nullptr;
  float *b1_ = nullptr, *tr1buf_ = nullptr;
  std::vector<ConvW*> all_convs_;
  size_t pcat_bytes_ = 0;
  void forward_chunk(const void* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh,
                     std::vector<ProfileEntry>* prof, int x_u8);
  std::unique_ptr<ThreadPool> pool_;
  Pending pending_;
  hipStream_t post_stream_ = nullptr;
  hipStream_t trace_stream_ = nullptr;
  hipEvent_t trace_done_ = nullptr;
  hipEvent_t pipe_ev_[2] = {nullptr, nullptr};
  int pipe_ev_next_ = 0;
  void* host_scratch_ = nullptr;
  size_t host_scratch_bytes_ = 0;
  void* host_adj_ = nullptr;
  size_t host_adj_bytes_ = 0;
  mutable int auto_threads_ = 0;   // min(16, CPU share), read once
  void* scratch_[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes_[5] = {0, 0, 0, 0, 0};
  struct Staging {
    void* in[2] = {nullptr, nullptr};
    float* out[2] = {nullptr, nullptr};
    size_t in_bytes = 0, elems = 0;
    unsigned long long uses = 0;   // pieces staged since the buffers were (re)allocated
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_fwd[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  };
  Staging stage_[2];
  hipStream_t copy_stream_ = nullptr, out_stream_ = nullptr;
  hipEvent_t ev_before_fwd_ = nullptr;
  int post_threads_ = 0;   // option post_threads: 0 = automatic
  int device_contours_ = -1;  // option device_contours (-1 = auto)
  int device_unclip_ = 1;     // option device_unclip: 0 host, 1 device where it pays (default), 2 device always
  bool device_polygons_ = true;   // option device_polygons
  int head_cus_yield_ = 2;        // option head_cus_yield: layer1's persistent grids leave the previous batch's tracer its CUs (pipelined calls)
  bool post_priority_ = true;     // option post_priority: post-processing / trace streams at the device's highest stream priority
};

class Recognizer {
 public:
  Recognizer(const void* blob, size_t bytes, int device);
  ~Recognizer();
  void set_stream(hipStream_t s) { stream_ = s ? s : own_stream_; }
  void synchronize();
  // "key=value;..." (include/ocr_amd.h, ocr_rec_set_options): small_batch=0 keeps every batch on the throughput kernels
  void set_options(const char* options);
  // device pointers; enqueues on the stream.  prof != null -> per-launch events (as Detector::forward)
  void classify(const float* crops_dev, int n, float* logits_dev, int32_t* labels_dev, double* probs_dev,
                std::vector<ProfileEntry>* prof = nullptr);
  void forward_host(const float* crops, int n, float* logits, int32_t* labels, double* probs);
  int device() const { return device_; }
  hipStream_t stream() const { return stream_; }
  static constexpr int kChunk = 65536;  // crops per pass: bounds the feat / hidden workspace (6144 bytes per crop)

 private:
  int device_;
  hipStream_t own_stream_ = nullptr, stream_ = nullptr;
  DeviceArena arena_;
  RecWeights w_{};
  void* stage_ = nullptr;
  size_t stage_bytes_ = 0;
  void ensure_workspace(int n);
  float *feat_ = nullptr, *hid_ = nullptr;  // [cap][1024], [cap][512]
  int ws_cap_ = 0;
  bool small_batch_ = true;  // batches of up to kRecSmallBatch crops take the latency-optimised kernels
};

void check_device(int device);

}  // namespace ocr
