/* ocr_amd.h - C ABI of the MI355X-native OCR inference hot path.
 *
 * This is the FFI surface a maintainer of lazareviczoran/ocr-rs binds instead of
 * tch/libtorch for the two inference stages (INTEGRATION.md shows the Rust side).
 * Plain pointers and sizes only; no torch types; nothing unwinds across the
 * boundary (the reference builds with panic = "abort", Cargo.toml:12-15): every
 * entry point returns an int status (0 = OK) and records a thread-local message
 * readable through ocr_last_error().
 *
 * Tensors are dense row-major f32, NCHW as tch hands them to ATen.  Handles are
 * bound to one GPU and one HIP stream, are not thread-safe, and calls are
 * synchronous unless the name ends in _async (mirrors the reference's blocking,
 * single-threaded calls).  DIFFERENT handles are independent: the
 * library keeps no shared mutable state, so threads that each own their handles
 * (a serving process, or one thread per GPU) may call concurrently and get the
 * bits they would get alone (tests/test_gpu_threads.py).
 */
#ifndef OCR_AMD_H
#define OCR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCR_OK 0
#define OCR_ERR_INVALID 1   /* bad argument / shape                      */
#define OCR_ERR_WEIGHTS 2   /* weight blob malformed or tensor missing   */
#define OCR_ERR_HIP 3       /* HIP runtime error (message has the call)  */
#define OCR_ERR_NOGPU 4     /* no usable gfx950 device: there is NO CPU fallback */
#define OCR_ERR_INTERNAL 5
#define OCR_ERR_DEGENERATE 6 /* expand_polygon gave None: the reference unwraps it and aborts (metrics.rs:103) */

/* memory kind of the data pointers handed to a call */
#define OCR_MEM_HOST 0
#define OCR_MEM_DEVICE 1

typedef struct ocr_det ocr_det_t;     /* detector: resnet18() graph + its weights  */
typedef struct ocr_rec ocr_rec_t;     /* recogniser: char_recognition::Net         */

/* Thread-local description of the last failure on this thread ("" if none). */
const char* ocr_last_error(void);
/* Library / ABI version, "ocr_amd <major>.<minor> gfx950". */
const char* ocr_version(void);
/* Number of visible HIP devices (never throws; 0 when there is no GPU). */
int ocr_device_count(void);

/* VarStore file -> OCRW blob, host only (no GPU needed): kind 0 / 1 = names as stored (detector), 2 = recogniser
 * (tch's name__N leafs mapped by shape).  *blob is malloc'ed by the library; release it with ocr_blob_free. */
int ocr_varstore_to_blob(const char* path, int kind, void** blob, size_t* blob_bytes);
void ocr_blob_free(void* blob);

/* ---------------------------------------------------------------------------
 * Detector.  Replaces, for inference:
 *   let net = resnet18(&vs.root()); vs.load(file)      text_detection/mod.rs:35-44
 *   net.forward_t(&x.view((n,1,h,w)), false)           text_detection/mod.rs:52-54, :196-197
 * graph definition: text_detection/model.rs:65-156.
 * `weights` is an OCRW v1 blob (ocr-rs_amd/weights.py documents the layout) whose
 * tensor names are the VarStore names of model.rs:68-105; it is copied.
 * ------------------------------------------------------------------------- */
int ocr_det_create(const void* weights, size_t weights_bytes, int device, ocr_det_t** out);
/* The same with explicit engine options, "key=value;key=value" (NULL or "" = the defaults).  The library never reads
 * the environment: which schedule runs is the caller's choice.  Every combination computes the same graph and is
 * held to the same parity bars (tests/test_gpu_parity.py::test_engine_modes_agree); they exist for A/B measurements.
 *   winograd_fused=0|1   (1)    fused Winograd F(4x4,3x3) kernel (winograd43_fused.hip) for the 3x3 s1 convs with 64 or 128 input
 *                               channels (trunk, FPN lateral terms); 0 = direct / unfused-Winograd convs
 *   out4_fused=0|1       (0)    1 = out4 (256 -> 64 at H/16) on the fused kernel too and out5 as a direct conv; 0 = both through the
 *                               unfused F(4x4,3x3) path of layer3 / layer4 (faster at these grid sizes)
 *   winograd43_x3=0|1    (0)    with mfma=split_bf16: 1 = the fused F(4x4,3x3) convs multiply on the bf16 matrix cores as well
 *                               (winograd43_x3.hip, same error bound); measured slower than the f32-MFMA kernel, kept for A/B
 *   winograd=<cin>|0     (256)  unfused Winograd for 3x3 s1 trunk convs with Cin >= cin that have no fused form; 0 = off
 *   winograd43=<cin>|0   (256)  of those, the layers with Cin >= cin use F(4x4,3x3) (36 products per 16 outputs) instead of
 *                               F(2x2,3x3) (16 per 4); 0 = F(2x2) everywhere
 *   fpn_unfused=0|1      (0)    1 = layer-by-layer FPN (laterals, top-down sums, out_k, gathered bin_conv1) as model.rs writes it
 *   bin_pyr=0|1          (1)    bin_conv1 over the upsampled concat as one phase-conv launch (0: four launches)
 *   pyr_grouped=0|1      (1)    bin_conv1's phase launch over p5, p4, p3 (split-bf16 and bf16 kernels): the output phases y mod 8 in {1,2}, {3,4},
 *                               {5,6} (same along x) read the same source rows - such a block of phases is one 128-column tile that fetches
 *                               and splits the operand once; the four corner phases are a second, small launch.  0 = one 64-column tile per
 *                               phase.  Bit-identical
 *   phase_windows=0|1    (1)    the FPN's up-2 phase convs (split-bf16 and bf16 kernels): phase 1 of cell i and phase 0 of cell i + 1 read the same two low-res
 *                               rows, so the GEMM's rows are the (H + 1) x (W + 1) 2 x 2 windows and the four phases that read a window are four
 *                               column groups of one operand tile (outputs outside the map are dropped).  0 = one 64-column tile per phase.  Bit-identical
 *   pyr_p2_direct=0|1    (1)    bf16 precision only: p2's 3x3 term of bin_conv1 as the patch-staged 64 -> 64 conv on top of the phase
 *                               launch over p5, p4, p3 (0: all four sources in the phase launch)
 *   tail_unfused=0|1     (0)    1 = probability head as two launches
 *   transform_fuse=0|1   (0)    layer3 / layer4, block 1: conv1's Winograd output transform and conv2's input transform in one launch (the
 *                               activation between them stays in LDS); bit-identical, and measured to buy nothing (docs/history.md)
 *   overlap=0|1|2|3      (3)    second stream: 1 small independent launches; 2 the FPN branch as it stands; 3 the FPN's fused-Winograd
 *                               launches (lateral terms of p2 / p3) and bin_conv1's p2 term - f32 matrix instructions - beside layer2 / layer3 / layer4 /
 *                               the small FPN convs - bf16 matrix instructions and HBM-bound transforms: 2 % of the step in both precisions (default kernels;
 *                               otherwise, and under ocr_det_forward_profile, one stream).  Sums re-associate by one rounding.  Any other value is OCR_ERR_INVALID
 *   x3_wide=0|1          (0)    the split-bf16 convs with plain NHWC stores and Cout % 128 == 0 (stride-2 3x3, 1x1, the Winograd GEMMs) as 256 x 128 tiles
 *                               on one persistent workgroup per CU (conv_x3w.hip): 30 % fewer operand bytes per MFMA, bit-identical, and measured 0-35 %
 *                               SLOWER than the 128-wide tiles at two workgroups per CU: kept for A/B
 *   bf16_block_fuse=0|1  (1)    bf16 precision: each BasicBlock of layer1 (conv3x3 + BN + ReLU, conv3x3 + BN, + x, ReLU: model.rs:40-55) as ONE launch, the
 *                               activation between its two convs held in LDS (basic_block_bf16_c64.hip): half the HBM traffic of the two launches, 1.25 x
 *                               their matrix work, 5-10 % less time; 0 = two conv3x3_bf16_c64 launches.  The same bits either way
 *   w43_cus=<n>          (0)    tuning: size of the fused Winograd kernel's persistent grid in CUs (two workgroups each); 0 = every CU of the device.
 *   w43_side_cus=<n>     (0)    the same for the fused Winograd launches that overlap=3 puts on the side stream; 0 = every CU.  Both 0..4096; any
 *                               grid size gives the same bits (tests/test_gpu_conv_kernel.py)
 *   post_threads=<n>     (0)    host threads of the post-processing stages (contours, unclip), the calling thread included;
 *                               0 = min(16, CPU share of the process: cgroup quota or online cores).  One process per GPU on a
 *                               shared host should pass its share (cores / ranks)
 *   device_unclip=0|1|2  (1)    behind the box scores, per candidate polygon on the GPU (unclip.hip; 1: where a call has more than 40 candidates per pool
 *                               thread - the kernel is lane-serial, 0.2 ms however few it gets -, 2: always): score threshold, miter offset, the union where
 *                               the ring is simple or only crosses itself at its concave vertices, min-size test, round(p / adj).  What it does
 *                               not settle (other self-intersections, squared-off corners, a short side within 3 px of min_size) the host
 *                               finishes inside the same call; results are bit for bit the host path's (0)
 *   device_contours=auto|0|1|2 (auto)  the contour tracing of ocr_det_postprocess / the pipelined calls on the GPU (contours.hip; one bit plane of the map
 *                               in LDS: up to 1024 x 1024, the reference's 800 x 800 included - larger maps, and images the kernel gives up on, take the host tracer inside the same call).
 *                               1: plausible border starts walked in parallel, the raster scan only replays the label tests, a row at a time as
 *                               word-wide bit arithmetic (0.35-0.5 ms per batch; the pipelined calls request it when they queue a batch); 2: one wave per image.  Identical
 *                               contours either way.  auto: 1 where the host pool (post_threads) has at most four threads, else 0: by
 *                               measurement
 *   head_cus_yield=0..4  (2)    pipelined calls, while the polygon chain of the previous batch runs beside this forward's first launches: layer1's persistent
 *                               grids (their blocks are dealt statically: a workgroup that shares its CU with the tracer's waves holds the launch up) are
 *                               1: sized for the CUs the tracer leaves (the form of the whole-CU tracer), 2..4: launched with that many workgroups per
 *                               resident slot, so that the hardware hands the later ones to whichever CU drains first.  0: nothing
 *   post_priority=0|1    (1)    the post-processing / trace streams at the device's highest stream priority: their short kernels are placed as
 *                               soon as a CU drains instead of queueing behind the next forward's workgroups
 *   device_polygons=0|1  (1)    with device contours on square maps: Douglas-Peucker, the >= 4 points filter and the box-score job list on
 *                               the GPU as well (candidates.hip): with device_unclip the whole chain from the probability map to the adjusted
 *                               polygons stays on the device and the host only collects.  0: contours back to the host pool
 *   mfma=split_bf16|f32  (split_bf16)  how the f32 precision multiplies in the MFMA-bound convs that have no Winograd kernel
 *                               of their own (stride-2 3x3, in5, FPN phase convs, bin_conv1 over the pyramid, the Winograd GEMMs of
 *                               layer3 / layer4, out4 and out5).  split_bf16: every f32 operand as the exact sum of three bf16 terms, six partial
 *                               products per pair on v_mfma_f32_32x32x16_bf16, f32 accumulation - the error of an f32 FMA chain
 *                               (dropped terms <= 2^-23 of a product; profiles/r03_bf16x3_accuracy.txt), the same parity bars, up to
 *                               2.67 x the f32 matrix rate.  f32: every conv on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain).
 *   precision=f32|bf16   (f32)  same as ocr_det_set_precision */
int ocr_det_create_with_options(const void* weights, size_t weights_bytes, int device, const char* options,
                                ocr_det_t** out);
/* Size limit of one launch.  The kernels address every tensor with 32-bit BYTE offsets below the out-of-range marker 2^31
 * (that marker is how zero padding and ragged tiles are expressed: such a lane reads zeros / its store is dropped), so every
 * workspace tensor must stay under 2^31 bytes.  The largest one holds N x (H/4) x (W/4) x 256 f32 = 64 N H W bytes: the engine
 * runs a batch in chunks of floor((2^31 - 1) / (64 H W)) frames - 81 at 640 x 640, 31 at 1024 x 1024, 7 at 2048 x 2048 - one
 * after the other on the handle's stream.  Results do not depend on the chunking (frames are independent in eval mode;
 * tests/test_gpu_fullsize.py).  Callers see it only as launch granularity. */
/* The one-call replacement of `vs.load(file)` (text_detection/mod.rs:41-44): reads the file tch's
 * VarStore::save wrote (utils.rs:55-63) - a libtorch zip archive of named tensors - without libtorch or Python,
 * checks names and shapes against the graph of model.rs:68-105 and builds the detector. */
int ocr_det_create_from_varstore(const char* path, int device, ocr_det_t** out);
void ocr_det_destroy(ocr_det_t* det);

/* Run all later work of this handle on an existing hipStream_t (e.g. the stream
 * of a torch.cuda.Stream).  NULL restores the handle's own stream.
 * Lifetime: the stream must outlive the handle, or be reset to NULL before it is destroyed - ocr_det_destroy
 * (and ocr_rec_destroy) wait for the work they queued on it before freeing the buffers that work touches. */
int ocr_det_set_stream(ocr_det_t* det, void* hip_stream);

/* Arithmetic of the detector (the reference runs f32 only; BASELINE config 5 names bf16 as the optional
 * reduced precision).  OCR_PRECISION_F32 (default): everything f32, the parity configuration.
 * OCR_PRECISION_BF16: every convolution of the graph - conv1, layer1..4, in2..5, out2..5, bin_conv1 and
 * bin_conv_tr1 - takes bf16 activations and weights on the bf16 matrix cores with f32 accumulation; folded
 * batch norm, residual adds, ReLU are f32 on the accumulators and the stored activations are bf16.  The last
 * transposed conv (64 -> 1), the sigmoid and all of the post-processing stay f32.  Inputs and outputs of every
 * entry point keep their f32 layout (raw 0..255 luma is exact in bf16). */
#define OCR_PRECISION_F32 0
#define OCR_PRECISION_BF16 1
int ocr_det_set_precision(ocr_det_t* det, int precision);

/* forward_t(xs, train=false): x is N x 1 x H x W f32 (raw 0..255 luma, no
 * normalisation - text_detection/mod.rs:46-54), prob is N x 1 x H x W f32 in
 * (0,1).  H and W must be multiples of 32.  Blocking. */
int ocr_det_forward(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, int mem_kind);

/* The same from the u8 image itself: the reference's frame IS u8 luma, turned into the f32 tensor without scaling
 * (image_ops.rs:350-364, text_detection/mod.rs:46-51); here that conversion happens inside the first kernel, so a host
 * batch crosses PCIe as 1 byte per pixel instead of 4.  Bit-identical to ocr_det_forward on (float)x.  Blocking. */
int ocr_det_forward_u8(ocr_det_t* det, const uint8_t* x, int n, int h, int w, float* prob, int mem_kind);

/* Pinned (page-locked) host memory for frames and maps: host-memory entry points copy from / to such buffers
 * asynchronously at PCIe speed; ordinary pageable memory also works, at the runtime's staging speed, and makes the
 * copy part of a call synchronous. */
int ocr_host_alloc(size_t bytes, void** out);
void ocr_host_free(void* p);

/* Same, device pointers only, returns after enqueueing on the handle's stream.
 * If bitmap != NULL it also receives binarize(prob, thresh) (metrics.rs:129-131)
 * as N x 1 x H x W u8, fused into the last kernel. */
int ocr_det_forward_async(ocr_det_t* det, const float* x_dev, int n, int h, int w, float* prob_dev,
                          uint8_t* bitmap_dev, float thresh);
int ocr_det_synchronize(ocr_det_t* det);

/* Per-kernel timing of one forward (hipEvents on the handle's stream around every
 * launch).  names[i] points to a static string; ms/flops/bytes are per launch:
 * algorithmic 2*MAC FLOPs and compulsory read+write bytes of that launch.
 * Returns the number of launches written (<= max_entries) through n_entries. */
int ocr_det_forward_profile(ocr_det_t* det, const float* x_dev, int n, int h, int w, float* prob_dev,
                            int max_entries, const char** names, float* ms, double* flops,
                            double* bytes, int* n_entries);

/* ---------------------------------------------------------------------------
 * Pre-processing (the step in front of the detector).  Replaces the arithmetic of
 *   preprocess_image(file, (W, H)) -> (GrayImage, adjust_x, adjust_y)   image_ops.rs:188-220
 * after decoding: rgba is h x w x 4 u8.  Aspect-preserving Triangle resize (image 0.23.11
 * sampling: vertical then horizontal pass, u8-truncating), to_luma, zero padding to
 * target_w x target_h.  gray (u8) and/or gray_f32 (the raw 0..255 values as f32, i.e. the
 * N=1 input frame of ocr_det_forward) receive target_h x target_w values; adj_xy[2] =
 * resized / original (x, y).  `det` supplies the GPU and stream.  Blocking.
 * ------------------------------------------------------------------------- */
int ocr_preprocess_image(ocr_det_t* det, const uint8_t* rgba, int w, int h, int target_w, int target_h,
                         uint8_t* gray, float* gray_f32, double* adj_xy, int mem_kind);

/* ---------------------------------------------------------------------------
 * Detection post-processing.  Replaces
 *   get_boxes_and_box_scores(pred, adjust_values) -> Result<PolygonScores>
 *                                                  text_detection/metrics.rs:37-56
 * (binarize :129, get_polygons_from_bitmap :58-127, box_score_fast :150-184,
 *  get_min_area_bounding_box :133-148, polygon.rs expand_polygon :51-56).
 * PolygonScores{ polygons: Vec<MultiPolygon<u32>>, scores: Vec<Vec<f64>> } is
 * returned as one CSR block owned by the library.
 * ------------------------------------------------------------------------- */
typedef struct ocr_postproc_params {
  double thresh;        /* 0.6  metrics.rs:38  */
  double box_thresh;    /* 0.7  metrics.rs:64  */
  double min_size;      /* 5.0  metrics.rs:66  */
  double unclip_ratio;  /* 2.0  metrics.rs:103 */
  /* A candidate whose offset polygon is empty (zero-area contour) makes the reference
   * abort: `expand_polygon(..).unwrap()`, metrics.rs:103 with panic = "abort".
   * 0 (default) = report it as OCR_ERR_DEGENERATE; 1 = drop that candidate and go on. */
  int32_t skip_degenerate;
  int32_t reserved;
} ocr_postproc_params_t;

typedef struct ocr_polygons {
  int32_t n_images;
  int32_t n_polygons;          /* total over the batch                       */
  int32_t n_vertices;          /* total over the batch                       */
  const int32_t* img_offsets;  /* [n_images+1]   polygon range of each image */
  const int32_t* poly_offsets; /* [n_polygons+1] vertex range of each polygon */
  const uint32_t* xy;          /* [2*n_vertices] x,y in ORIGINAL-image pixels (round(p/adj) as u32) */
  const double* scores;        /* [n_polygons]   box_score_fast of each kept polygon */
} ocr_polygons_t;

void ocr_postproc_default_params(ocr_postproc_params_t* p);

/* prob: N x 1 x H x W f32, adj_xy: N x 2 f64 (host memory, x then y scale =
 * resized/original, image_ops.rs:200-202).  params == NULL -> reference constants.
 * `det` supplies the GPU and stream (binarisation and box scores run as HIP
 * kernels; contour tracing and Clipper-style offsetting run on host threads).
 * Blocking.  *out must be released with ocr_polygons_free. */
int ocr_det_postprocess(ocr_det_t* det, const float* prob, int n, int h, int w, int mem_kind,
                        const double* adj_xy, const ocr_postproc_params_t* params,
                        ocr_polygons_t** out);
void ocr_polygons_free(ocr_polygons_t* p);
/* Where the polygon chain of this handle's post-processing calls ran so far (cumulative counters; diagnostic - the reference has no
 * counterpart, get_boxes_and_box_scores metrics.rs:37-127 runs on one core): out[0] images whose contours were traced on the GPU,
 * out[1] images traced on the host (device_contours off, a map the tracer does not take, an image it gave up), out[2] candidate polygons
 * the device unclip settled, out[3] candidates finished on the host, out[4] images whose whole chain - trace, Douglas-Peucker, box
 * score, unclip - stayed on the device, out[5] post-processing passes.  Results never depend on where a step ran. */
int ocr_det_post_stats(ocr_det_t* det, int64_t out[6]);

/* forward_t + get_boxes_and_box_scores over a STREAM of batches, software-pipelined inside the library: the call
 * enqueues the forward of THIS batch (device pointers; x_dev N x 1 x H x W f32 -> prob_dev, which must stay untouched
 * until the next call has returned) and, while the GPU runs it, post-processes the batch handed in by the PREVIOUS call
 * - its host geometry on the handle's thread pool, its kernels and copies on a second stream.  *prev_out receives that
 * previous batch's polygons (NULL on the first call).  Finish with x_dev = NULL: nothing is enqueued, the last
 * batch's polygons come back.  Results are exactly those of ocr_det_forward + ocr_det_postprocess per batch. */
int ocr_det_detect_pipelined(ocr_det_t* det, const float* x_dev, int n, int h, int w, float* prob_dev,
                             const double* adj_xy, const ocr_postproc_params_t* params, ocr_polygons_t** prev_out);

/* The same pipeline for frames in HOST memory - the form the reference's call sites have (CPU tensors in,
 * text_detection/mod.rs:46-67).  x_host: N x 1 x H x W frames, OCR_ELEM_F32 or OCR_ELEM_U8 (raw luma, 1/4 of the
 * bytes).  The call copies them into a double-buffered device staging area on a copy stream (asynchronous for
 * ocr_host_alloc'ed memory) - beside the forward of the previous batch -, enqueues this batch's forward behind that copy,
 * and post-processes the previous batch while the GPU works.  The probability map stays on the device unless
 * prob_host != NULL, which then holds this batch's map once the NEXT call (the one returning its polygons) has returned.
 * x_host may be reused as soon as the call returns only if it is pageable; a pinned buffer must stay untouched until
 * the next call returns.  Finish with x_host = NULL.  Batches may differ in size and element kind from call to call; a
 * batch that needs larger staging slots than the pending one is enqueued only after the pending batch has been finished
 * (that one call loses its overlap, results are unchanged).  The staging of this entry point is its own: blocking
 * ocr_det_forward / ocr_det_forward_u8 calls on host memory between two pipelined calls do not disturb the pending batch. */
#define OCR_ELEM_F32 0
#define OCR_ELEM_U8 1
int ocr_det_detect_pipelined_host(ocr_det_t* det, const void* x_host, int x_elem, int n, int h, int w, float* prob_host,
                                  const double* adj_xy, const ocr_postproc_params_t* params, ocr_polygons_t** prev_out);

/* Detect -> recognise link (BUILD-DEFINED: the reference never implemented its "Character
 * Segmentation" step, README.md:20-26, so there is no reference rule to match).  For every polygon
 * of `polys` (as returned by ocr_det_postprocess for the same batch) the axis-aligned bounding box,
 * mapped back to frame coordinates with adj_xy, is resampled bilinearly to 28 x 28 and divided by
 * 255 (load_image_as_tensor's scaling, image_ops.rs:80-83): crops is n_polygons x 784 f32, ready
 * for ocr_rec_forward / ocr_rec_classify.  frames: N x 1 x H x W f32 (the detector's input).
 * frames and crops share mem_kind; polys and adj_xy are host memory.  Rule: oracle/crop_oracle.py.
 * Stream order: the crop kernel runs behind everything queued on the detector's stream when the call is made - except,
 * while a pipelined batch is pending (ocr_det_detect_pipelined*), the forward of that pending batch: the crops of the batch
 * that has come back are cut beside it, behind whatever was queued before that forward was.  Device frames written by work
 * queued AFTER the last pipelined call need the caller's own synchronisation. */
int ocr_extract_crops(ocr_det_t* det, const float* frames, int n, int h, int w, int mem_kind,
                      const ocr_polygons_t* polys, const double* adj_xy, float* crops);

/* ---------------------------------------------------------------------------
 * Detection quality metrics (host code; consumers of the polygon lists).  Replaces
 *   evaluate_image(gt, ignore_flags, pred) -> Result<MetricsItem>      metrics.rs:255-380
 *   combine_results(results) -> Result<(precision, recall, hmean)>     metrics.rs:229-253
 * (validate_measure :191-219 = drop predictions with score < 0.6, then evaluate_image per image;
 *  gather_measure :221-227 = combine_results over the concatenated items: see the host mirrors.)
 * Polygons are CSR: offsets[n+1] into x,y pairs, u32 as in MultiPolygon<u32>.
 * ------------------------------------------------------------------------- */
typedef struct ocr_metrics_item {   /* MetricsItem, metrics.rs:22-30 */
  double precision, recall, hmean;
  int32_t gt_care, det_care, det_matched;
} ocr_metrics_item_t;

int ocr_evaluate_image(const uint32_t* gt_xy, const int32_t* gt_offsets, int n_gt, const uint8_t* ignore_flags,
                       const uint32_t* pred_xy, const int32_t* pred_offsets, int n_pred, ocr_metrics_item_t* out);
int ocr_combine_results(const ocr_metrics_item_t* items, int n, double* precision, double* recall, double* hmean);

/* ---------------------------------------------------------------------------
 * Recogniser.  Replaces
 *   let net = Net::new(&weights.root()); weights.load(file)   char_recognition/mod.rs:44-46
 *   net.forward_t(&image_tensor, false)                       char_recognition/mod.rs:53-54
 *   .softmax(-1, Kind::Double); topk(&res, 1)                 mod.rs:55-56, utils.rs:28-43
 * graph: char_recognition/model.rs:13-39.
 * ------------------------------------------------------------------------- */
int ocr_rec_create(const void* weights, size_t weights_bytes, int device, ocr_rec_t** out);
/* `weights.load(file)` of char_recognition/mod.rs:46.  Net::new creates its four layers on one nn::Path
 * (model.rs:13-24), so the file holds tch's de-duplicated names weight, bias, weight__2, bias__3, ...; they are
 * mapped onto conv1 / conv2 / fc1 / fc2 by shape (all eight shapes differ).  conv1.weight ... names work too. */
int ocr_rec_create_from_varstore(const char* path, int device, ocr_rec_t** out);
void ocr_rec_destroy(ocr_rec_t* rec);
/* Same contract as ocr_det_set_stream, including the lifetime rule. */
int ocr_rec_set_stream(ocr_rec_t* rec, void* hip_stream);
/* Options, "key=value;key=value" (NULL or "" changes nothing; an unknown key is OCR_ERR_INVALID):
 *   small_batch=0|1   1 (default): batches of up to 1 024 crops take the latency-optimised kernels (see ocr_rec_forward);
 *                     0: every batch takes the throughput kernels, every multiply on the f32 matrix instructions - a
 *                     crop's logits are then bit-identical whatever the size of the batch it arrives in. */
int ocr_rec_set_options(ocr_rec_t* rec, const char* options);
int ocr_rec_synchronize(ocr_rec_t* rec);

/* forward_t: crops N x 784 (28x28, values in [0,1]) -> logits N x 62.  Blocking.
 * Batch-size invariance: batches of up to 1 024 crops and larger ones take different kernels (latency- against throughput-
 * optimised; the small-batch conv2 multiplies on the bf16 matrix cores from three-way split f32 operands).  Within a family a crop's
 * logits are bit-identical whatever batch it arrives in; across the two they agree to rounding (|dlogit| < 1e-4, |dp| < 1e-5:
 * tests/test_gpu_fullsize.py), so a crop whose two best classes tie within that margin may change label when the batch size
 * crosses 1 024 (ocr_rec_set_options "small_batch=0" removes the distinction).  The reference itself gives no stronger
 * guarantee (ATen's kernels differ with batch size too). */
int ocr_rec_forward(ocr_rec_t* rec, const float* crops, int n, float* logits, int mem_kind);
/* forward_t + softmax(-1, f64) + top-1: label index into VALUES (utils.rs:7) and
 * its probability.  logits may be NULL.  Device pointers; enqueues and returns. */
int ocr_rec_classify_async(ocr_rec_t* rec, const float* crops_dev, int n, float* logits_dev,
                           int32_t* labels_dev, double* probs_dev);
/* Per-kernel timing of one classify pass over n device-resident crops (hipEvents on the handle's stream around
 * every launch; same conventions as ocr_det_forward_profile: flops = MFMA FLOPs the launch executes). */
int ocr_rec_classify_profile(ocr_rec_t* rec, const float* crops_dev, int n, int32_t* labels_dev, double* probs_dev,
                             int max_entries, const char** names, float* ms, double* flops, double* bytes,
                             int* n_entries);
/* Blocking convenience over either memory kind. */
int ocr_rec_classify(ocr_rec_t* rec, const float* crops, int n, int32_t* labels, double* probs,
                     int mem_kind);
/* The label alphabet, utils.rs:7 ("A-Za-z0-9", 62 symbols, NUL terminated). */
const char* ocr_rec_alphabet(void);
/* EXTENSION - no counterpart in the reference (its recogniser classifies single 28 x 28 glyphs, char_recognition/model.rs:27-39):
 * CTC greedy (best-path) decode of a sequence recogniser's output, the stage BASELINE.json's north_star names.  logits N x T x C f32
 * (or any monotone transform of them), blank in [0, C): per crop the first class attaining each column's maximum, consecutive repeats
 * collapsed, blanks dropped -> labels N x T int32 (row i: lengths[i] classes, then -1) and lengths N.  One wave per crop on the handle's
 * GPU and stream; blocking; mem_kind says where the three buffers live.  Exact integers: oracle/ctc_oracle.py. */
int ocr_ctc_greedy_decode(ocr_rec_t* rec, const float* logits, int n, int t, int c, int blank, int mem_kind, int32_t* labels,
                          int32_t* lengths);

/* ---------------------------------------------------------------------------
 * Multi-GPU exchange.  Frames and crops are independent (eval-mode batch norm), so a batch shards over the GPUs of
 * a node with no data-path collective: one process (or thread) per GPU, each with its own detector / recogniser
 * handle and its own contiguous slice of the batch.  The single exchange step is the all-gather of the results,
 * over RCCL (xGMI inside a node).  The reference has no counterpart (it is single-device); a Rust
 * host binds these four calls next to the others.
 *   rank 0: ocr_comm_unique_id(id) and hand the 128 bytes to the other ranks (file, environment, launcher ...);
 *   every rank: ocr_comm_create(id, world, rank, device, &comm)      - collective
 *               ocr_comm_all_gather_polygons(comm, mine, &all)       - collective; `all` holds the frames of rank 0,
 *                                                                      then rank 1, ...; free with ocr_polygons_free
 * RCCL is loaded on first use (librccl.so.1); without it these calls fail with a message, nothing else is affected.
 * ------------------------------------------------------------------------- */
typedef struct ocr_comm ocr_comm_t;
#define OCR_COMM_ID_BYTES 128
int ocr_comm_unique_id(uint8_t* id /* [OCR_COMM_ID_BYTES] */);
int ocr_comm_rccl_version(int* version);
int ocr_comm_create(const uint8_t* id, int world, int rank, int device, ocr_comm_t** out);
void ocr_comm_destroy(ocr_comm_t* comm);
int ocr_comm_all_gather_polygons(ocr_comm_t* comm, const ocr_polygons_t* local, ocr_polygons_t** all);
/* labels of the local crops (host memory) -> labels of every rank's crops in rank order; counts[world] receives
 * the number each rank contributed (may be NULL); fails if more than `capacity` labels arrive. */
int ocr_comm_all_gather_labels(ocr_comm_t* comm, const int32_t* labels, int n_local, int32_t* all, int capacity,
                               int32_t* counts, int* n_all);

#ifdef __cplusplus
}
#endif
#endif /* OCR_AMD_H */
