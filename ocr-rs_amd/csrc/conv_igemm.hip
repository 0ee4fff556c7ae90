// NHWC f32 convolution as an implicit GEMM on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 FMA chain, 157 TF/s chip peak).
//
// Implements every conv of /root/reference/src/text_detection/model.rs:107-151:
//   3x3 s1/s2 p1 (basic_block :40-55, out2..5 :80-98, bin_conv1 :100), 1x1 s1/s2
//   (downsample :30-38, in2..in5 :75-78) and, as a 1x1 GEMM with a pixel-shuffle
//   store, conv_transpose2d k=2 s=2 (bin_conv_tr1 :103).
// Fused into the A-operand gather: the channel concat of the four FPN outputs with their
//   x8/x4/x2 nearest upsamples (:140) - the 256-channel fuse tensor is never written.
// Fused into the epilogue: eval-mode batch norm as scale/bias, residual add, ReLU, and the
//   FPN top-down sum  up2(in_{k+1}) + in_k  (:126-137) as a second output of the lateral conv.
// Composed forms of the linear part of the graph (engine.hip builds the weights):
//   STORE_PHASE - a 3x3 conv of a nearest-upsampled tensor as up*up 2x2 phase convs on the low-res grid
//                 (p2 / p3 with their lateral 1x1 convs folded in);
//   SRC_PYR4    - bin_conv1 over the upsampled concat per output phase (y mod 8, x mod 8) with p5, p4, p3, p2
//                 as four sources of one K loop;
//   batch = 16  - the sixteen GEMMs of a Winograd F(2x2,3x3) conv (winograd.hip) as grid slices of one launch.
// Phases that read the same operand rows share one operand tile (split-bf16 and bf16 kernels; bit-identical to one tile per phase):
//   PYRG        - SRC_PYR4: output phases y mod 8 in {1,2}, {3,4}, {5,6} (same along x) see the same pixels of p5, p4, p3: a block of such
//                 phases is one 128-column tile, 64 columns per phase;
//   WING        - STORE_PHASE with up 2: phase 1 of cell i and phase 0 of cell i + 1 read the same two rows: the GEMM's rows are the 2 x 2
//                 windows of the low-res grid, the four phases of a window its four column groups.
//
// Data movement: both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds):
// no staging registers, no ds_write, no per-K-step address arithmetic (the K-step offset is
// an SGPR), and zero padding is free - a lane whose tap lies outside the image carries an
// out-of-range buffer offset and the hardware writes zeros into LDS (verified on gfx950 by
// tools/probes/lds_dma_probe.hip).  An LDS stage holds BM + BN rows of 32 floats (128 B,
// unpadded because a DMA instruction writes 1 KiB lane-linearly = 8 rows); the 16-byte
// chunk c of row r lives in slot c ^ ((r >> 1) & 7), applied on the global-address side and
// undone by the fragment read, which makes every 16-lane ds_read_b128 group conflict-free.
// Two stages, one barrier per K-step: the DMA of step k+1 flies during the MFMAs of step k.
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile BM x BN x 128 bytes of K with BM,BN in
// {64,128}; each wave owns (BM/2) x (BN/2) as 32x32 MFMA tiles.
//
// Element types.  TI (operands) is f32 or bf16, TO (output, residuals) f32 or bf16; accumulation,
// scale/bias and the epilogue arithmetic are always f32.  A 128-byte LDS row holds 32 f32 or 64 bf16
// of K, so the DMA / swizzle / fragment-read machinery is byte-identical for both; bf16 feeds
// v_mfma_f32_32x32x16_bf16 (one 16-byte fragment = one MFMA instead of four).  The default (and the
// benchmarked BASELINE config 1) is f32 throughout; bf16 is the opt-in precision of config 5.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace ocr {
static int g_conv_debug = 0;  // -DIGEMM_DEBUG builds: ablation bits of the X3 kernel
void set_conv_debug(int d) { g_conv_debug = d; }
namespace igemm {  // named (not anonymous): the kernel stubs are referenced from templates below

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// 4 consecutive elements of an f32 / bf16 tensor as f32x4 (16- / 8-byte access)
template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ f32x4 load4(const void* base, size_t idx) { return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + idx); }
  static __device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) { *reinterpret_cast<f32x4*>(static_cast<float*>(base) + idx) = v; }
  static __device__ __forceinline__ float load1(const void* base, size_t idx) { return static_cast<const float*>(base)[idx]; }
  static __device__ __forceinline__ void store1(void* base, size_t idx, float v) { static_cast<float*>(base)[idx] = v; }
  // four elements at byte offset voff of a buffer descriptor (out of range: zeros / dropped)
  template <typename R> static __device__ __forceinline__ f32x4 bload4(R rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  }
  template <typename R> static __device__ __forceinline__ void bstore4(R rsrc, unsigned voff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, 0, 0);
  }
};
template <> struct Elem<__bf16> {
  static __device__ __forceinline__ f32x4 load4(const void* base, size_t idx) {
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(base) + idx);
    f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return v;
  }
  static __device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) {
    bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};  // round to nearest even
    *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(base) + idx) = h;
  }
  static __device__ __forceinline__ float load1(const void* base, size_t idx) { return (float)static_cast<const __bf16*>(base)[idx]; }
  static __device__ __forceinline__ void store1(void* base, size_t idx, float v) { static_cast<__bf16*>(base)[idx] = (__bf16)v; }
  template <typename R> static __device__ __forceinline__ f32x4 bload4(R rsrc, unsigned voff) {
    const bf16x4 h = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0));
    f32x4 v = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return v;
  }
  template <typename R> static __device__ __forceinline__ void bstore4(R rsrc, unsigned voff, f32x4 v) {
    bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};  // round to nearest even
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), rsrc, voff, 0, 0);
  }
};

// One 16-byte-per-lane LDS-DMA load (buffer_load_dwordx4 ... lds): lane l's 16 bytes land at lds_addr + 16 l.
// Written as inline asm, not __builtin_amdgcn_raw_ptr_buffer_load_lds: for the builtin the compiler knows the
// load writes LDS and, unable to prove that the operand stage being filled is not the one being read, puts
// s_waitcnt vmcnt(0) in front of the next ds_read - every wave then sits out the full DMA latency right after
// issuing it instead of multiplying the resident stage.  Here the ordering is explicit instead: the K loop
// waits (vmcnt(0)) and barriers before a stage is read, and the "memory" clobber keeps LDS accesses from
// being moved across the issue.  m0 carries the LDS address; the compiler sets m0 itself before any use of
// its own, so it is not preserved.
template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, int soff) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("" : "+s"(soff));  // a folded constant outside the inline range is not a valid soffset: keep it in an SGPR
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

struct ConvArgs {
  const void* src;        // PLAIN: the input tensor; CAT4: allocation holding p5,p4,p3,p2   (TI)
  const void* wgt;        //                                                                  (TI)
  const float* scale;
  const float* bias;
  const void* residual;   //                                                                  (TO)
  const void* up_residual;
  void* out;
  void* out2;
  unsigned src_bytes, wgt_bytes;
  int src_off[4];         // CAT4: element offset of p5,p4,p3,p2 inside src
  int N, Hin, Win, Cin;
  int Ho, Wo, Cout;
  int M;        // N*Ho*Wo
  int pad;
  int relu;
  int nblk_n;   // Cout / BN
  int nblk;     // total blocks
  int nblk_m;   // tiles along M (PHASE: per phase; the grid holds up*up phases)
  int up_shift; // PHASE: log2 of the upsampling factor (1, 2, 3)
  int batch;    // batched GEMM: number of problems (grid slices along M); 1 otherwise
  int pyr_chunked;  // PYR4 tile order (see the kernel)
  int pyr_nsrc;     // PYR4: 4 = p5, p4, p3, p2; 3 = without p2
  int win;          // STORE_PHASE, up 2, 128 columns: rows are 2 x 2 WINDOWS of the low-res grid ((Hin + 1) x (Win + 1) per image), the four phases
                    // that read a window are its four column groups (see WING in the kernel); Ho / Wo hold the window grid
  int pyr_group;    // PYR4 (split-bf16 / bf16 kernels): 0 = one tile per phase; 1 = phase blocks as column groups (128-wide tiles); 2 = the four corner phases
  unsigned mg_howo, sh_howo, mg_wo, sh_wo;  // magic numbers: x / (Ho*Wo), x / Wo
  int debug;    // builds with -DIGEMM_DEBUG only (ocr_test_set_conv_debug): X3 ablations - 1 no A DMA, 2 no B DMA, 4 no split, 8 no MFMA
};
#if defined(IGEMM_ABL)   // compile-time ablation (make EXTRA=-DIGEMM_ABL=<bits>): no run-time cost, for timing what is left
#define IGEMM_DBG(p, bit) ((IGEMM_ABL & (bit)) != 0)
#elif defined(IGEMM_DEBUG)
#define IGEMM_DBG(p, bit) ((p).debug & (bit))
#else
#define IGEMM_DBG(p, bit) false
#endif

// x / d for 0 <= x < 2^31 with host-computed (magic, shift): 5 instructions instead of ~25
__device__ __forceinline__ int fast_div(int x, unsigned magic, unsigned shift) {
  if (shift == 0xFFFFFFFFu) return x;  // d == 1 (wave-uniform)
  const unsigned t = __umulhi((unsigned)x, magic);
  return (int)((t + (((unsigned)x - t) >> 1)) >> shift);
}

// Consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its
// own L2).  Remap so that every XCD walks a contiguous run of tiles: neighbouring
// tiles share input halos and weight panels (bijective for any block count).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// f(integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

[[maybe_unused]] constexpr int ROWB = 128;                 // bytes of K per LDS row (one cache line): 32 f32 or 64 bf16
[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;     // voffset beyond any tensor (< 2^31 bytes): reads as zero

template <typename TI, typename TO, int BM, int BN, int KS, int STRIDE, int SRC, int STORE, bool X3 = false>
__global__ __launch_bounds__(256) void conv_igemm(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the body uses device-only types (buffer resources, LDS address space):
                                     // the host pass only needs the launch stub
  // X3: waves stacked along M (each owns 32 rows x all BN columns: every A element is split exactly once)
  constexpr int WGM = X3 ? 4 : 2, WGN = X3 ? 1 : 2;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int AI = BM / 32;
  constexpr int BSUB = BN / 64;               // X3: 64-row groups of a B plane (one DMA instruction per wave each)
  constexpr int BI = X3 ? 3 * BSUB : BN / 32; // B DMA instructions per wave and stage
  constexpr int EB = sizeof(TI);              // bytes per activation element
  constexpr int EBW = X3 ? 2 : EB;            // bytes per weight element (X3: bf16 planes)
  constexpr int BK = ROWB / EB;               // K elements per LDS row / K-step
  constexpr bool BF16 = EB == 2;
  // TS (tap sharing; bf16 3x3 stride-1 convs): the A operand of the three taps of a kernel row is ONE run of BM + 2 consecutive
  // input pixels (linear order over N x H x W), staged once and read at a row shift of kw; pixels that the linear order drags in
  // across a row / image boundary are zeroed in the fragment registers.  A third of the activation gathers of the tap-by-tap
  // form - the part of the operand traffic that costs (weights: contiguous, L2-resident, free in the ablation).
  constexpr bool TS = BF16 && !X3 && KS == 3 && STRIDE == 1 && SRC == SRC_PLAIN && STORE != STORE_PHASE;
  constexpr int A_ROWS = TS ? BM + 8 : BM;   // TS: rows 0 .. BM + 1 hold the run (pixel m0 - 1 + row), up to BM + 7 are DMA granularity
  constexpr int A_BYTES = A_ROWS * ROWB;
  constexpr int B_BYTES = X3 ? 3 * BN * 64 : BN * ROWB;   // X3: three bf16 planes of BN rows x 32 k (64-byte rows)
  constexpr int STAGE = A_BYTES + B_BYTES;    // bytes per LDS stage
  static_assert(!X3 || (BM == 128 && EB == 4 && (BN == 64 || BN == 128) && MT == 1), "X3 tile shapes");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = X3 ? wave : wave >> 1, wn = X3 ? 0 : wave & 1;

  const int bid = xcd_remap(blockIdx.x, p.nblk);
  const int tile_n = bid % p.nblk_n;
  int tile_m = bid / p.nblk_n;
  // PHASE (3x3 conv of a nearest-x-up upsampled tensor, computed on the low-res grid): output pixel
  // (up i + pa, up j + pb) sees a 2x2 low-res neighbourhood - rows i-1, i for pa = 0, rows i, i+1 otherwise
  // (same for columns) - with that phase's pre-summed weights; the up*up phases are consecutive slices of
  // the grid.
  int ph = 0, pa = 0, pb = 0;
  // PYRG (SRC_PYR4, 128 columns): the phases y mod 8 in {1,2}, {3,4}, {5,6} read the SAME source rows of p5, p4 and p3 (their
  // 3x3 windows cover the same low-res pixels; only the pre-summed weights differ), likewise along x.  Such a block of 2 x 2 (or
  // 1 x 2, 2 x 1) phases is ONE GEMM with 64 columns per phase: the A operand is fetched and split once for 128 columns instead
  // of once per 64, and 21 + 4 tiles gather a cell block's sources instead of 64.  Row groups ga = 0..4 <-> y mod 8 = {0}, {1,2},
  // {3,4}, {5,6}, {7}; a virtual tile is (ga, gb, 128-column half); the four corner phases (64 columns) run as launch kind 2.
  constexpr bool PYRG = SRC == SRC_PYR4 && BN == 128;
  // (both forms exist in the split-bf16 kernel and in the bf16 kernel; f32 operands on the f32 MFMA keep one tile per phase.  WING is
  // switched at run time by p.win - described where its tiles are decoded, below)
  constexpr bool WING = (X3 || sizeof(TI) == 2) && STORE == STORE_PHASE && SRC == SRC_PLAIN && KS == 2 && BN == 128;
  [[maybe_unused]] int g_nb = 1, g_half = 0;
  if constexpr (PYRG) {
    constexpr int CHS = 2, CH = 1 << CHS;
    const int nm = p.nblk_m;
    // tiles of row group ga: [0, 3 nm), [3 nm, 11 nm), [11 nm, 19 nm), [19 nm, 27 nm), [27 nm, 30 nm)
    int ga = 0, idx = tile_m, cnt = 3;
    if (tile_m >= 27 * nm) {
      ga = 4;
      idx = tile_m - 27 * nm;
    } else if (tile_m >= 3 * nm) {
      const int u = (tile_m - 3 * nm) / (8 * nm);
      ga = 1 + u;
      idx = tile_m - 3 * nm - u * 8 * nm;
      cnt = 8;
    }
    int v;
    if ((nm & (CH - 1)) == 0 && p.pyr_chunked) {
      // order (row group | chunk of CH cell blocks | virtual tile | block in chunk): the tiles of a chunk gather the same source
      // lines back to back inside one XCD's run
      const int per = cnt * CH, chunk = idx / per, rem = idx - chunk * per;
      v = rem >> CHS;
      tile_m = chunk * CH + (rem & (CH - 1));
    } else {
      v = idx / nm;
      tile_m = idx - v * nm;
    }
    int gb;
    if (cnt == 3) {          // one phase row: the corner columns belong to launch kind 2
      gb = 1 + v;
    } else if (v == 0 || v == 7) {
      gb = v == 0 ? 0 : 4;
    } else {
      gb = 1 + ((v - 1) >> 1);
      g_half = (v - 1) & 1;
    }
    pa = ga == 0 ? 0 : 2 * ga - 1;
    pb = gb == 0 ? 0 : 2 * gb - 1;
    g_nb = (gb == 0 || gb == 4) ? 1 : 2;
    g_nb = __builtin_amdgcn_readfirstlane(g_nb);
    g_half = __builtin_amdgcn_readfirstlane(g_half);
    ph = pa * 8 + pb;
  } else if constexpr (SRC == SRC_PYR4) {
    constexpr int CHS = BM == 128 ? 2 : 3, CH = 1 << CHS;   // cell blocks per chunk: 512 cells of the p5 grid
    if (p.pyr_group == 2) {   // the corner phases (0 | 7, 0 | 7), plain order
      const int c4 = tile_m / p.nblk_m;
      tile_m -= c4 * p.nblk_m;
      pa = (c4 >> 1) * 7;
      pb = (c4 & 1) * 7;
      ph = pa * 8 + pb;
    } else if ((p.nblk_m & (CH - 1)) == 0 && p.pyr_chunked) {
      // order (phase row a | chunk of CH cell blocks | phase column b | block in chunk): an XCD's contiguous run of
      // tiles is one phase row; the 8 x CH tiles of a chunk share the source lines they gather (3 MB, L2-sized)
      // and each weight set is used by CH consecutive tiles.  Phase-major order fetched 3.5 GB for 280 MB of sources.
      const int j = tile_m & (CH - 1);
      pb = (tile_m >> CHS) & 7;
      const int rest = tile_m >> (CHS + 3), nchunk = p.nblk_m >> CHS;
      pa = rest / nchunk;
      tile_m = (rest - pa * nchunk) * CH + j;
      ph = pa * 8 + pb;
    } else {
      ph = tile_m / p.nblk_m;
      tile_m -= ph * p.nblk_m;
      pa = ph >> 3;
      pb = ph & 7;
    }
  } else if (WING && p.win) {
    // WING: window (wy, wx) = low-res rows {wy - 1, wy} x columns {wx - 1, wx} is what phase (1, 1) of cell (wy - 1, wx - 1), phase (1, 0) of
    // (wy - 1, wx), (0, 1) of (wy, wx - 1) and (0, 0) of (wy, wx) read: with windows as the rows of the GEMM the four phases are four
    // column groups of 64 (their own weight rows) over ONE gathered, once-split operand tile; it produces the outputs
    // (2 wy - 1 + dy, 2 wx - 1 + dx), those outside the map are dropped.  Tiles (row tile, 128-column half) are neighbours in an XCD's run.
    g_half = __builtin_amdgcn_readfirstlane(tile_m & 1);
    tile_m >>= 1;
    g_nb = 2;
  } else if constexpr (STORE == STORE_PHASE) {
    // order (chunk of CH row tiles | phase | tile in chunk): the up x up phases of a low-res tile read the same 3 x 3
    // neighbourhood - as consecutive tiles of one XCD's run they find it in that L2 instead of fetching it once per phase
    // (phase-major order: 322 MB fetched for 79 MB of input), and a phase's weight set serves CH tiles in a row
    constexpr int CH = 8;
    const int nph = 1 << (2 * p.up_shift);
    if (p.nblk_m % CH == 0) {
      const int j = tile_m % CH, rest = tile_m / CH;
      ph = rest % nph;
      tile_m = (rest / nph) * CH + j;
    } else {
      ph = tile_m / p.nblk_m;
      tile_m -= ph * p.nblk_m;
    }
    pa = ph >> p.up_shift;
    pb = ph & ((1 << p.up_shift) - 1);
  }
  if constexpr (SRC == SRC_PYR4) {  // block coordinates are wave-uniform: say so (SGPRs, scalar branches)
    tile_m = __builtin_amdgcn_readfirstlane(tile_m);
    pa = __builtin_amdgcn_readfirstlane(pa);
    pb = __builtin_amdgcn_readfirstlane(pb);
    ph = __builtin_amdgcn_readfirstlane(ph);
  }
  // batched GEMM: slice bz of the grid is problem bz (own A, B and output blocks)
  int bz = 0;
  if constexpr (STORE == STORE_NHWC && KS == 1 && STRIDE == 1 && SRC == SRC_PLAIN) {
    if (p.batch > 1) {
      bz = tile_m / p.nblk_m;
      tile_m -= bz * p.nblk_m;
    }
  }
  // PHASE: only the first and the last phase of a row / column of phases see two low-res rows / columns; the
  // phases in between see one (their second tap has zero weight and is skipped).  Active taps are the prefix
  // t < nt of the tap order t = kh * nw + kw, which is also the order of this phase's weights.
  const int last_ph = (1 << p.up_shift) - 1;
  int nw = (STORE == STORE_PHASE) ? ((pb == 0 || pb == last_ph) ? 2 : 1) : KS;
  int nt = (STORE == STORE_PHASE) ? (((pa == 0 || pa == last_ph) ? 2 : 1) * nw) : KS * KS;
  // PYR4: taps of source s (0 = p5 .. 2 = p3 upsampled by u = 8 >> s, 3 = p2) at this output phase
  auto pyr_taps = [&](int s, int phase) { const int u = 8 >> s, q = phase & (u - 1); return s == 3 ? 3 : (q == 0 || q == u - 1) ? 2 : 1; };
  if constexpr (SRC == SRC_PYR4) {
    nw = pyr_taps(0, pb);
    nt = pyr_taps(0, pa) * nw;
  }
  const int m0 = tile_m * BM;
  const int n0 = (PYRG || (WING && p.win)) ? g_half * BN : tile_n * BN;
  // PYRG: 64-column group q of the block -> its phase (pa + dpa, pb + dpb)
  [[maybe_unused]] auto group_dpa = [&](int q) { return g_nb == 2 ? q >> 1 : q; };
  [[maybe_unused]] auto group_dpb = [&](int q) { return g_nb == 2 ? q & 1 : 0; };

  // descriptor inputs through readfirstlane: provably wave-uniform, so the descriptors stay in SGPRs however the
  // allocator treats the other kernel arguments (the inline-asm DMA needs them there)
  auto uniform_ptr = [](const void* q) {
    const unsigned long long u = (unsigned long long)q;
    return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
  };
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.src), 0, __builtin_amdgcn_readfirstlane((int)p.src_bytes), 0x00020000);
  const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.wgt), 0, __builtin_amdgcn_readfirstlane((int)p.wgt_bytes), 0x00020000);

  // ---- DMA coordinates: this lane fills LDS slot q of rows r + 32 i with global chunk q ^ f(r)
  const int r = tid >> 3;
  const int q = tid & 7;
  const int gq = q ^ ((r >> 1) & 7);  // f(row) only depends on row bits 1..3; row = r + 32 i
  int ih0[AI], iw0[AI], abase[AI];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + r + 32 * i;
    if (m < p.M) {
      const int n = fast_div(m, p.mg_howo, p.sh_howo);
      const int rem = m - n * HoWo;
      const int oh = fast_div(rem, p.mg_wo, p.sh_wo);
      const int ow = rem - oh * p.Wo;
      ih0[i] = oh * STRIDE - p.pad + (pa > 0);
      iw0[i] = ow * STRIDE - p.pad + (pb > 0);
      if constexpr (SRC == SRC_PYR4) {  // the row is cell (oh, ow) of the p5 grid
        ih0[i] = oh;
        iw0[i] = ow;
      }
      if constexpr (SRC == SRC_CAT4 || SRC == SRC_PYR4) abase[i] = n;  // image index; the pixel offset depends on the source
      else abase[i] = ((n * p.Hin + ih0[i]) * p.Win + iw0[i]) * p.Cin * EB + gq * 16 + bz * p.M * p.Cin * EB;  // bytes
    } else {
      ih0[i] = -(1 << 20);  // every tap out of range -> zeros
      iw0[i] = 0;
      abase[i] = 0;
    }
  }
  unsigned bvoff[X3 ? BSUB : BI];
  constexpr int PYR_TAPS = 21;  // weight row of SRC_PYR4: 4 + 4 + 4 tap slots of p5, p4, p3 and 9 of p2
  const int wrow = SRC == SRC_PYR4 ? PYR_TAPS * 64 : KS * KS * p.Cin;
  if constexpr (X3) {
    // a DMA instruction fills 16 rows x 64 B of one plane: lane l -> row l >> 2, slot l & 3 holding global chunk
    // slot ^ f(row), f(row) = (row >> 2) & 3 (= (l >> 4) & 3: the wave's rows start at a multiple of 16)
    // The planes are TILED on the host (split3_weights_tiled): block (16 rows, K-step, plane) = 1 KB in the order of its LDS image
    // (row r, slot s holding chunk s ^ f(r)), blocks ordered [row / 16][K-step][plane].  A DMA instruction reads 1 KB of
    // consecutive bytes - eight full cache lines - where the row-major planes gave it sixteen half lines 2 wrow bytes apart (the
    // other half of each line was fetched again a tap later): the weight stream was a quarter of these kernels' time.
    const int nK = wrow >> 5;   // K-steps per weight row
#pragma unroll
    for (int i = 0; i < BSUB; ++i) {
      int brow = (ph + bz) * p.Cout + n0 + 64 * i;
      if constexpr (PYRG) {   // weight rows [phase][64]: column group 2 g_half + i is its own phase
        const int q = 2 * g_half + i;
        brow = ((pa + group_dpa(q)) * 8 + pb + group_dpb(q)) * 64;
      }
      if constexpr (WING) {
        if (p.win) {   // column group q = (dy, dx) is phase (1 - dy, 1 - dx): weight rows [phase = 2 a + b][64]
          const int q = 2 * g_half + i;
          brow = (3 - q) * 64;
        }
      }
      bvoff[i] = (unsigned)(((brow >> 4) + wave) * nK * 3072 + lane * 16);
    }
  } else {
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      int brow = (ph + bz) * p.Cout + n0 + 32 * i;
      if constexpr (PYRG) {   // weight rows [phase][64]: column group 2 g_half + (i >> 1) is its own phase
        const int q = 2 * g_half + (i >> 1);
        brow = ((pa + group_dpa(q)) * 8 + pb + group_dpb(q)) * 64 + 32 * (i & 1);
      }
      if constexpr (WING) {
        if (p.win) brow = (3 - (2 * g_half + (i >> 1))) * 64 + 32 * (i & 1);   // column group q = (dy, dx) is phase (1 - dy, 1 - dx)
      }
      bvoff[i] = (unsigned)((brow + r) * wrow * EB + gq * 16);
    }
  }

  // K order.  PLAIN: channel chunk outer, tap inner - the KS*KS taps of one 32-channel chunk
  // touch the same few cache lines of neighbouring pixels back to back (L2 hits), instead of
  // coming back to each pixel's lines once per tap with a reuse distance far beyond L2 (the
  // 256-channel out2 conv fetched 8.8x its input that way).  All tap offsets are precomputed
  // per row; only SGPR offsets change in the loop.
  // CAT4: segments (tap, source) of 2 K-steps; per-lane offsets recomputed per segment.
  constexpr int NTAP = KS * KS;
  const int csteps = p.Cin / BK;
  unsigned avoff[NTAP][AI];
  // offsets of all taps for source s (PLAIN: the one input tensor; CAT4: p5,p4,p3,p2 = s 0..3,
  // nearest-upsampled by 8,4,2,1, 64 channels each)
  auto tap_offset = [&](int s, int t, int i) -> unsigned {
    int kh = t / KS, kw = t - kh * KS;
    if constexpr (SRC == SRC_PYR4) {
      // source pixel of output (8 i + pa, 8 j + pb) at the source's resolution, window start and tap decode
      const int ush = 3 - s, u = 1 << ush;
      const int snw = pyr_taps(s, pb);
      kh = snw == 3 ? t / 3 : snw == 2 ? t >> 1 : t;
      kw = t - kh * snw;
      const int ys = (ih0[i] << s) + (pa >> ush) - ((s == 3 || (pa & (u - 1)) == 0) ? 1 : 0) + kh;
      const int xs = (iw0[i] << s) + (pb >> ush) - ((s == 3 || (pb & (u - 1)) == 0) ? 1 : 0) + kw;
      const int Hs = p.Hin << s, Ws = p.Win << s;
      const bool inside = (unsigned)ys < (unsigned)Hs && (unsigned)xs < (unsigned)Ws;
      const unsigned o = (unsigned)((p.src_off[s] + ((abase[i] * Hs + ys) * Ws + xs) * 64) * EB + gq * 16);
      return inside ? o : OOB;
    } else if constexpr (STORE == STORE_PHASE) {
      kh = nw == 2 ? t >> 1 : t;
      kw = nw == 2 ? t & 1 : 0;
    }
    const int ih = ih0[i] + kh, iw = iw0[i] + kw;
    const bool ok = (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
    unsigned off;
    if constexpr (SRC == SRC_CAT4) {
      const int sh = 3 - s;
      off = (unsigned)((p.src_off[s] + ((abase[i] * (p.Hin >> sh) + (ih >> sh)) * (p.Win >> sh) + (iw >> sh)) * 64) * EB + gq * 16);
    } else {
      off = (unsigned)(abase[i] + (kh * p.Win + kw) * p.Cin * EB);
    }
    return ok ? off : OOB;
  };
  auto prep_source = [&](int s) {
    if constexpr (SRC == SRC_PYR4) {
      nw = pyr_taps(s, pb);
      nt = pyr_taps(s, pa) * nw;
    }
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      // PYR4: an upsampled source has at most 2 x 2 taps at a phase - the offsets of the others are never used, and this
      // integer arithmetic (per source and tile: taps x rows x ~15 VALU) is time taken from the matrix pipe
      if (SRC == SRC_PYR4 && t >= nt) continue;  // wave-uniform
#pragma unroll
      for (int i = 0; i < AI; ++i) avoff[t][i] = tap_offset(s, t, i);
    }
  };
  if constexpr (!(X3 && SRC == SRC_PYR4) && !TS) prep_source(0);
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)lds + (unsigned)(8 * wave * ROWB);
  auto issue_plain = [&](int stage, const unsigned (&av)[AI], int tap, int c, int kbase) {
    const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE));
#pragma unroll
    for (int i = 0; i < AI; ++i)
      if (!IGEMM_DBG(p, 1)) dma16(a_rsrc, st + 32 * i * ROWB, av[i], c * ROWB);
    if constexpr (!X3) {   // (the split-bf16 loop has its own DMA schedule)
#pragma unroll
      for (int i = 0; i < BI; ++i) dma16(b_rsrc, st + (A_ROWS + 32 * i) * ROWB, bvoff[i], (tap * p.Cin + kbase) * EB + c * ROWB);
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // MFMA operand fetch: lane l supplies row (l & 31) and the 16-byte chunk c = 2g + (l>>5) of K-group
  // g, stored in slot c ^ f(row) (same map for A and B).
  //   f32 : the chunk is 4 k; the j-th of four v_mfma_f32_32x32x2_f32 takes k = 8g+j (lanes 0-31) and
  //         8g+4+j (lanes 32-63);
  //   bf16: the chunk is 8 k = exactly one operand of v_mfma_f32_32x32x16_bf16 (k = 16g + 8 (l>>5) + j).
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  int xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = ((2 * g + (lane >> 5)) ^ fsw) * 16;
  const int a_row = (wm * WM + frow) * ROWB;
  const int b_row = (A_ROWS + wn * WN + frow) * ROWB;

  // X3 B fragment: row (lane & 31) of column tile j, chunk 2 kk + (lane >> 5) in slot chunk ^ f(row)
  const int bx_row = A_BYTES + frow * 64;
  const int bx_f = (frow >> 2) & 3;
  auto compute = [&](int stage) {
    const unsigned char* st = lds + stage * STAGE;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(st + a_row + i * 32 * ROWB + xoff[g]);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(st + b_row + j * 32 * ROWB + xoff[g]);
      if constexpr (BF16) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
                                                                 acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
  };

  // One pass = all channel chunks of one source (PLAIN: the only one; CAT4: four passes of 2 chunks).
  constexpr bool MULTI = SRC == SRC_CAT4 || SRC == SRC_PYR4;
  const int NSRC = SRC == SRC_PYR4 ? p.pyr_nsrc : MULTI ? 4 : 1;  // PYR4 may leave p2 (its last source) to another kernel
  const int pass_chunks = MULTI ? 64 / BK : csteps;
  int par = 0;  // LDS stage holding the K-step about to be multiplied
  if constexpr (X3) {
    // ---- split-bf16 main loop.  The K-steps form one flat sequence (source, channel chunk, tap); the DMA side walks it two
    // steps ahead of the multiplier with its own iterator.  A K-step is two MFMA groups (kk = 0, 1: sixteen k each); the
    // operand fragments of a group are read from LDS and split (VALU) during the MFMAs of the group before it, and the
    // barrier sits BETWEEN the two groups of a step:
    //   phase 1: MFMAs (k, 0) | read + split (k, 1) from stage par              ... wait for DMA k+1, barrier ...
    //   phase 2: DMA k+2 -> stage par | MFMAs (k, 1) | read + split (k+1, 0) from stage par ^ 1
    // so every MFMA group has the next group's LDS latency and ~44 split instructions to cover and no group starts cold.
    struct Frag {
      bf16x8 ah, am, al;
      bf16x8 bh[NT], bm[NT], bl[NT];
    };
    struct Raw { f32x4 a0, a1; };
    // LDS reads of one fragment set: the f32 A chunks (split later, in pieces, between the MFMAs) and the bf16 B planes
    auto read_frag = [&](int stage, int kk, Raw& r, Frag& f) {
      const unsigned char* st = lds + stage * STAGE;
      if (IGEMM_DBG(p, 32)) {   // no LDS reads: operands from whatever the registers hold
        asm volatile("" : "+v"(r.a0), "+v"(r.a1));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(f.bh[j]), "+v"(f.bm[j]), "+v"(f.bl[j]));
        return;
      }
      // eight k of this lane: chunks g = 2 kk, 2 kk + 1 -> k = 16 kk + 4 h + e and 16 kk + 8 + 4 h + e (the weights' k order
      // inside a 16-group is permuted to match on the host, split3_weights)
      r.a0 = *reinterpret_cast<const f32x4*>(st + a_row + xoff[2 * kk]);
      r.a1 = *reinterpret_cast<const f32x4*>(st + a_row + xoff[2 * kk + 1]);
      const int boff = bx_row + (((2 * kk + (lane >> 5)) ^ bx_f) * 16);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f.bh[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64);
        f.bm[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64 + BN * 64);
        f.bl[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64 + 2 * BN * 64);
      }
    };
    // elements [E0, E1) of the A fragment: x = hi + mid + lo, round to nearest even at every level (the remainders are
    // exact in f32, the third term has at most 8 significant bits left)
    auto split_part = [&](const Raw& r, Frag& f, auto e0c, auto e1c) {
      constexpr int E0 = decltype(e0c)::value, E1 = decltype(e1c)::value;
      if (IGEMM_DBG(p, 4)) {
        f.ah = __builtin_bit_cast(bf16x8, r.a0);
        f.am = __builtin_bit_cast(bf16x8, r.a1);
        f.al = __builtin_bit_cast(bf16x8, r.a0 + r.a1);
        return;
      }
#pragma unroll
      for (int e = E0; e < E1; ++e) {
        const float x = e < 4 ? r.a0[e] : r.a1[e - 4];
        const __bf16 h = (__bf16)x;
        const float r1 = x - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        f.ah[e] = h;
        f.am[e] = m;
        f.al[e] = (__bf16)r2;
      }
    };
    // six of the nine partial products per column tile (mid.lo, lo.mid, lo.lo are below 2^-23 of the product): mfma_at below
    // total K-steps and the DMA iterator
    int total = 0;
    if constexpr (SRC == SRC_PYR4) {
      for (int s = 0; s < 3; ++s) total += pass_chunks * pyr_taps(s, pa) * pyr_taps(s, pb);   // the three upsampled sources
    } else {
      total = pass_chunks * nt;
    }
    int it_s = 0, it_c = 0, it_t = 0;
    // PYR4 (three upsampled sources, at most 2 x 2 taps each at a phase): the tap tables of all sources up front, slot
    // 4 s + t = the weight row's tap slot - no table is rebuilt inside the K loop and no source index is dynamic
    constexpr bool PYRX = SRC == SRC_PYR4;
    // per source and row the offset of the window's first pixel (taken modulo 2^32: it may lie before the tensor) and one validity
    // bit per tap slot; a step's offsets are base + a scalar tap distance, or the out-of-range marker - 16 registers where the
    // twelve full tap tables took 48 (the 128-column form has to fit two workgroups per CU)
    unsigned axb[PYRX ? 3 : 1][AI], axv[AI];
    int nts[3] = {0, 0, 0};
    if constexpr (PYRX) {
#pragma unroll
      for (int i = 0; i < AI; ++i) axv[i] = 0u;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ush = 3 - s, u = 1 << ush;
        const int snw = pyr_taps(s, pb), snh = pyr_taps(s, pa);
        nts[s] = snh * snw;
        const int Hs = p.Hin << s, Ws = p.Win << s;
        const int dy = (pa >> ush) - ((pa & (u - 1)) == 0 ? 1 : 0), dx = (pb >> ush) - ((pb & (u - 1)) == 0 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const int ys = (ih0[i] << s) + dy, xs = (iw0[i] << s) + dx;
          axb[s][i] = (unsigned)((p.src_off[s] + ((abase[i] * Hs + ys) * Ws + xs) * 64) * EB + gq * 16);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int kh = snw == 2 ? t >> 1 : t, kw = t - kh * snw;
            if (t < nts[s] && (unsigned)(ys + kh) < (unsigned)Hs && (unsigned)(xs + kw) < (unsigned)Ws) axv[i] |= 1u << (4 * s + t);
          }
        }
      }
    }
    // the step the iterator points at: its A offsets (one static row of the tap table, picked by a wave-uniform switch)
    // and scalar parts; then advance
    unsigned cur_av[AI];
    int cur_soff_a = 0, cur_soff_b = 0;
    auto select_next = [&] {
      if constexpr (PYRX) {
        const int flat = 4 * it_s + it_t;
        const int snw = pyr_taps(it_s, pb);
        const int kh = snw == 2 ? it_t >> 1 : it_t, kw = it_t - kh * snw;
        const unsigned dist = (unsigned)((kh * (p.Win << it_s) + kw) * 64 * EB);   // scalar
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const unsigned b = it_s == 0 ? axb[0][i] : it_s == 1 ? axb[1][i] : axb[2][i];   // wave-uniform selects
          cur_av[i] = (axv[i] >> flat) & 1u ? b + dist : OOB;
        }
        cur_soff_a = it_c * ROWB;
        cur_soff_b = (flat * pass_chunks + it_c) * 3072;   // K-step (tap slot, chunk) of the tiled planes: 3 x 1 KB
        const int ntc = it_s == 0 ? nts[0] : it_s == 1 ? nts[1] : nts[2];
        if (++it_t == ntc) {
          it_t = 0;
          if (++it_c == pass_chunks) {
            it_c = 0;
            ++it_s;
          }
        }
        return;
      }
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
        if (it_t == t) {
#pragma unroll
          for (int i = 0; i < AI; ++i) cur_av[i] = avoff[t][i];
        }
      cur_soff_a = it_c * ROWB;
      cur_soff_b = (it_t * csteps + it_c) * 3072;   // K-step (tap, chunk) of the tiled planes: 3 x 1 KB
      if (++it_t == nt) {
        it_t = 0;
        ++it_c;
      }
    };
    constexpr int NP = AI + 3 * BSUB;   // DMA instructions per wave and step: A row groups, then B planes
    static_assert(NP == 10 || NP == 7, "the hand-counted s_waitcnt vmcnt values assume 128 x 128 or 128 x 64 tiles");
    auto dma_piece = [&](int stage, auto piece_c) {
      constexpr int P = decltype(piece_c)::value;
      if constexpr (P < AI) {
        if (!IGEMM_DBG(p, 1)) dma16(a_rsrc, __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE + 32 * P * ROWB)), cur_av[P], cur_soff_a);
      } else {
        constexpr int pl = (P - AI) / BSUB, i = (P - AI) % BSUB;
        if (!IGEMM_DBG(p, 2))
          dma16(b_rsrc, __builtin_amdgcn_readfirstlane(lds_base - (unsigned)(8 * wave * ROWB) + (unsigned)(stage * STAGE + A_BYTES + 16 * wave * 64 + pl * BN * 64 + i * 64 * 64)),
                bvoff[i], cur_soff_b + pl * 1024);
      }
    };
    auto dma_all = [&](int stage) {
      dma_piece(stage, std::integral_constant<int, 0>{});
      dma_piece(stage, std::integral_constant<int, 1>{});
      dma_piece(stage, std::integral_constant<int, 2>{});
      dma_piece(stage, std::integral_constant<int, 3>{});
      dma_piece(stage, std::integral_constant<int, 4>{});
      dma_piece(stage, std::integral_constant<int, 5>{});
      dma_piece(stage, std::integral_constant<int, 6>{});
      if constexpr (NP > 7) {
        dma_piece(stage, std::integral_constant<int, 7>{});
        dma_piece(stage, std::integral_constant<int, 8>{});
        dma_piece(stage, std::integral_constant<int, 9>{});
      }
    };
    // product pr (0..5, small terms first) of column tile j, as MFMA number idx = NT * pr + j of a group: consecutive
    // MFMAs of a wave go to DIFFERENT accumulators (a dependent MFMA issued back to back waits for its predecessor's
    // result; per accumulator the order of the six products is unchanged, so the sums are bit-identical either way)
    auto mfma_at = [&](const Frag& f, auto idx_c) {
      constexpr int idx = decltype(idx_c)::value, j = idx % NT, pr = idx / NT;
      if (IGEMM_DBG(p, 8)) {
        asm volatile("" ::"v"(f.al), "v"(f.ah), "v"(f.am), "v"(f.bh[j]), "v"(f.bm[j]), "v"(f.bl[j]));
        return;
      }
      if constexpr (pr == 0) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.bh[j], acc[0][j], 0, 0, 0);
      if constexpr (pr == 1) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bl[j], acc[0][j], 0, 0, 0);
      if constexpr (pr == 2) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am, f.bm[j], acc[0][j], 0, 0, 0);
      if constexpr (pr == 3) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am, f.bh[j], acc[0][j], 0, 0, 0);
      if constexpr (pr == 4) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bm[j], acc[0][j], 0, 0, 0);
      if constexpr (pr == 5) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.bh[j], acc[0][j], 0, 0, 0);
    };
    // One MFMA group = 6 NT MFMAs in four regions of 6 NT / 4; the other work of the wave is dealt out BETWEEN the MFMAs
    // (an MFMA keeps the matrix pipe busy for 32 cycles - room for ~7 other instructions of the same wave):
    //   region 0: the LDS reads of the next group's fragments (issued at its start, landing under its MFMAs)
    //   regions 1..3: the split of the next A fragment, a few VALU behind every MFMA
    //   region ends (phase 2 only): this wave's DMA instructions of the step after next
    // (the empty asm statements pin the split to its region: without them the compiler sinks the whole split - pure
    // arithmetic - to its first use behind the barrier, where no MFMA of this wave covers it)
    auto phase = [&](const Frag& cur, int rd_stage, int rd_kk, Raw& nraw, Frag& nxt, int dma_stage) {   // dma_stage < 0: no DMA in this phase
      constexpr int Q = 6 * NT / 4;   // MFMAs per region
#if defined(IGEMM_DMA_EARLY)      // experiment: every piece behind the first region / behind the last one
      constexpr int D1 = NP, D2 = NP, D3 = NP;
#elif defined(IGEMM_DMA_LATE)
      constexpr int D1 = 0, D2 = 0, D3 = 0;
#else
      constexpr int D1 = NP == 10 ? 3 : 2, D2 = NP == 10 ? 6 : 4, D3 = NP == 10 ? 8 : 6;
#endif
      constexpr int E1 = NT == 4 ? 4 : 3;   // split elements per region: [0, E1), [E1, 6), [6, 8) (NT = 4: whole cvt_pk pairs)
      read_frag(rd_stage, rd_kk, nraw, nxt);
      static_for<0, Q>([&](auto i) { mfma_at(cur, i); });
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dma_stage >= 0) static_for<0, D1>([&](auto i) { dma_piece(dma_stage, i); });
      asm volatile("" : "+v"(nraw.a0), "+v"(nraw.a1));
      static_for<Q, 2 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, E1>{});
      asm volatile("" : "+v"(nxt.ah), "+v"(nxt.am), "+v"(nxt.al));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NT == 4 ? 4 : 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dma_stage >= 0) static_for<D1, D2>([&](auto i) { dma_piece(dma_stage, i); });
      asm volatile("" : "+v"(nraw.a0), "+v"(nraw.a1));
      static_for<2 * Q, 3 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, E1>{}, std::integral_constant<int, 6>{});
      asm volatile("" : "+v"(nxt.ah), "+v"(nxt.am), "+v"(nxt.al));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NT == 4 ? 2 : 6, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dma_stage >= 0) static_for<D2, D3>([&](auto i) { dma_piece(dma_stage, i); });
      asm volatile("" : "+v"(nraw.a1));
      static_for<3 * Q, 4 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
      asm volatile("" : "+v"(nxt.ah), "+v"(nxt.am), "+v"(nxt.al));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NT == 4 ? 2 : 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dma_stage >= 0) static_for<D3, NP>([&](auto i) { dma_piece(dma_stage, i); });
    };
#ifdef IGEMM_STAGGER   // experiment: the second workgroup of a CU (odd wave slot) starts IGEMM_STAGGER x 8 k cycles late
    if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1)
      for (int i = 0; i < IGEMM_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    // the first two steps are requested back to back; the loop starts as soon as the first has landed
    select_next();
    dma_all(0);
    if (total > 1) {
      select_next();
      dma_all(1);
      if constexpr (NP == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    Frag fx = {}, fy = {};
    Raw rw = {};
    read_frag(0, 0, rw, fx);
    split_part(rw, fx, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    if (IGEMM_DBG(p, 256)) total = 1;
    for (int k = 0; k < total; ++k) {
      phase(fx, par, 1, rw, fy, -1);
      // this wave's reads of stage par are complete and its share of DMA k+1 has landed; after the barrier so is everyone's
      if (IGEMM_DBG(p, 512)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (timing only: does not wait for the DMA)
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (!IGEMM_DBG(p, 64)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool more = k + 2 < total;
      if (more) select_next();
      phase(fy, par ^ 1, 0, rw, fx, more ? par : -1);   // (after the last step: reads of a stale stage, never used)
      par ^= 1;
    }
    __syncthreads();
  } else if constexpr (TS) {
    // ---- tap-sharing loop.  Steps (channel chunk c, kernel row kh, kernel column kw); B stage = step parity, A stage = run parity
    // (run = (c, kh), three steps).  The next run is requested in three pieces over the steps of the current one, the next
    // step's weights at every step; one barrier per step as in the plain loop.
    // run row j <-> input pixel m0 - 1 + j + (kh - 1) W of the linear order; everything that may be negative sits in the voffset
    // (the range check covers the voffset only: a negative or too large one reads as zeros), the chunk in the soffset
    // (unsigned arithmetic: pixel -1 of the first tile and the rows past the last pixel wrap modulo 2^32 to offsets at or beyond
    // 2^31, the kh term is added modulo 2^32 as well)
    unsigned tsv[AI + 1];
#pragma unroll
    for (int i = 0; i <= AI; ++i) {
      const int j = i < AI ? r + 32 * i : BM + (tid >> 3);   // the last instruction (rows BM ..) is wave 0's
      tsv[i] = ((unsigned)(m0 - 1 + j) * (unsigned)p.Cin) * (unsigned)EB + (unsigned)(gq * 16);
    }
    const unsigned kh_bytes = (unsigned)(p.Win * p.Cin * EB);
    // validity of tap (kh, kw) for this lane's fragment rows: bit 3 kh + kw
    unsigned vmask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WM + i * 32 + frow;
      vmask[i] = 0;
      if (m < p.M) {
        const int n = fast_div(m, p.mg_howo, p.sh_howo);
        const int rem = m - n * HoWo;
        const int oh = fast_div(rem, p.mg_wo, p.sh_wo);
        const int ow = rem - oh * p.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t)
          if ((unsigned)(oh + t / 3 - 1) < (unsigned)p.Hin && (unsigned)(ow + t % 3 - 1) < (unsigned)p.Win) vmask[i] |= 1u << t;
      }
    }
    const unsigned a_dst = lds_base;                                                                   // + stage * STAGE + 32 i ROWB
    const unsigned x_dst = (unsigned)(size_t)(lds_void*)lds + (unsigned)(BM * ROWB);                   // rows BM .. BM + 7
    auto issue_a_piece = [&](int stage, int kh, int c, int kw) {   // piece kw of the run's AI (+ 1 for wave 0) instructions
      const unsigned khb = (unsigned)(kh - 1) * kh_bytes;   // kh = 0: minus one image row, modulo 2^32
#pragma unroll
      for (int i = 0; i <= AI; ++i) {
        if ((i * 3) / (AI + 1) != kw) continue;   // uniform
        if (IGEMM_DBG(p, 1)) continue;
        if (i < AI) dma16(a_rsrc, __builtin_amdgcn_readfirstlane(a_dst + (unsigned)(stage * STAGE + 32 * i * ROWB)), tsv[i] + khb, c * ROWB);
        else if (wave == 0) dma16(a_rsrc, __builtin_amdgcn_readfirstlane(x_dst + (unsigned)(stage * STAGE)), tsv[i] + khb, c * ROWB);
      }
    };
    auto issue_b = [&](int stage, int tap, int c) {
      const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(stage * STAGE));
#pragma unroll
      for (int i = 0; i < BI; ++i) dma16(b_rsrc, st + (A_ROWS + 32 * i) * ROWB, bvoff[i], (tap * p.Cin) * EB + c * ROWB);
    };
    auto compute_ts = [&](int a_stage, int b_stage, int kw, int tap) {
      const unsigned char* sa = lds + a_stage * STAGE + a_row + kw * ROWB;
      const unsigned char* sb = lds + b_stage * STAGE;
      const int fk = ((frow + kw) >> 1) & 7;
      const unsigned bit = 1u << tap;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 af[MT], bf[NT];
        const int xa = ((2 * g + (lane >> 5)) ^ fk) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(sa + i * 32 * ROWB + xa);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sb + b_row + j * 32 * ROWB + xoff[g]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          if (!(vmask[i] & bit)) af[i] = f32x4{0.f, 0.f, 0.f, 0.f};   // a pixel of another row / image / beyond the border
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
      }
    };
    issue_a_piece(0, 0, 0, 0);
    issue_a_piece(0, 0, 0, 1);
    issue_a_piece(0, 0, 0, 2);
    issue_b(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int run = 0, bpar = 0;
    for (int c = 0; c < csteps; ++c) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh, ++run) {
        const bool more_runs = kh < 2 || c + 1 < csteps;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = 3 * kh + kw;
          // the next step's weights, a third of the next run
          if (kw < 2) issue_b(bpar ^ 1, tap + 1, c);
          else if (kh < 2) issue_b(bpar ^ 1, tap + 1, c);
          else if (c + 1 < csteps) issue_b(bpar ^ 1, 0, c + 1);
          if (more_runs) issue_a_piece((run + 1) & 1, kh < 2 ? kh + 1 : 0, kh < 2 ? c : c + 1, kw);
          compute_ts(run & 1, bpar, kw, tap);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          bpar ^= 1;
        }
      }
    }
  } else {
  issue_plain(0, avoff[0], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < NSRC; ++s) {
    // first K-step of the NEXT source needs that source's tap-0 offsets while this source's
    // table is still live: computed up front (AI registers)
    unsigned next0[AI];
    if constexpr (MULTI) {
#pragma unroll
      for (int i = 0; i < AI; ++i) next0[i] = tap_offset(min(s + 1, NSRC - 1), 0, i);
    }
    // B operand position of this source's taps: CAT4 - channel block s of every tap; PYR4 - tap slots 4 s ..
    const int tap0 = SRC == SRC_PYR4 ? 4 * s : 0, kb = SRC == SRC_CAT4 ? s * 64 : 0;
    const int tap0n = SRC == SRC_PYR4 ? 4 * (s + 1) : 0, kbn = SRC == SRC_CAT4 ? (s + 1) * 64 : 0;
    for (int c = 0; c < pass_chunks; ++c) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        if (STORE == STORE_PHASE && t >= nt) continue;  // wave-uniform
        const int nxt = par ^ 1;
        // DMA of the next K-step flies while this one is multiplied
        if (t + 1 < NTAP && t + 1 < nt) issue_plain(nxt, avoff[(t + 1) % NTAP], tap0 + t + 1, c, kb);
        else if (c + 1 < pass_chunks) issue_plain(nxt, avoff[0], tap0, c + 1, kb);
        else if (MULTI && s + 1 < NSRC) issue_plain(nxt, next0, tap0n, 0, kbn);
        compute(par);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed ...
        __syncthreads();                                   // ... and so has everyone's; the old stage is free
        par ^= 1;
      }
    }
    if constexpr (MULTI) {
      if (s + 1 < NSRC) prep_source(s + 1);
    }
  }

  }

  if (IGEMM_DBG(p, 128)) {   // no epilogue: keep the accumulators alive, store nothing
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // ---- epilogue A (plain NHWC store, the MFMA-bound convs): straight from the accumulators.
  // C/D map of a 32x32 MFMA tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5); one store
  // instruction writes 2 rows x 128 contiguous bytes.  Residual rows are all requested before use.
  if (STORE == STORE_NHWC && sizeof(TO) == 4 && !p.out2) {
    const int colq = lane & 31;
    const int rowq = (lane >> 5) * 4;
    {
      // buffer stores relative to this wave's corner of the output: per store only accumulator read, scale / bias, ReLU
      // and one address add (row index x row bytes, a scalar); the column tile is an immediate; rows beyond M fall outside
      // the descriptor's range and are dropped by the hardware.  (A 64-bit index, a compare and an exec mask per store -
      // the first form of this loop - cost a third of the run time of the K = 256 Winograd GEMMs.)
      const unsigned row_b = (unsigned)p.Cout * 4u;
      const int r0 = m0 + wm * WM;
      const int rows = min(max(p.M - r0, 0), WM);
      const size_t corner = ((size_t)bz * p.M + r0) * p.Cout + (n0 + wn * WN);
      const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(static_cast<float*>(p.out) + corner), 0,
                                                            __builtin_amdgcn_readfirstlane((int)(rows * row_b)), 0x00020000);
      const size_t rcorner = (size_t)r0 * p.Cout + (n0 + wn * WN);
      const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(static_cast<const float*>(p.residual ? p.residual : p.out) + (p.residual ? rcorner : corner)), 0,
                                                            __builtin_amdgcn_readfirstlane((int)(rows * row_b)), 0x00020000);
      const unsigned voff = (unsigned)rowq * row_b + (unsigned)colq * 4u;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * WN + j * 32 + colq;
        const float sc = p.scale ? p.scale[col] : 1.f;
        const float bi = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float res[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) res[e] = 0.f;
          if (p.residual) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              res[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, voff + (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * row_b + j * 128, 0, 0));
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float v = acc[i][j][e] * sc + bi + res[e];
            if (p.relu) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, voff + (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * row_b + j * 128, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);  // one 32x32 tile at a time: keeps the residual staging at 16 registers
        }
      }
    }
    return;
  }

  // ---- epilogue of the phase convs (split-bf16 and bf16 kernels): as epilogue B below (tile through LDS, row-major float4 traffic), but every residual
  // row of the thread is requested BEFORE the tile goes through LDS - all loads in flight under the transpose instead of rounds of
  // load, wait, store behind it (the 128 x 128 phase-block tile's epilogue was a third of its launch: two workgroups per CU, and
  // the one in its epilogue multiplies nothing).  Row offsets by magic division per thread: no row table, one barrier less.
  if constexpr (STORE == STORE_PHASE && (X3 || sizeof(TI) == 2)) {
    constexpr int CPR = BN / 4, RPP = 256 / CPR, PASSES = BM / RPP;
    constexpr unsigned ES = sizeof(TO);
    static_assert(BM * BN * 4 <= 2 * STAGE, "accumulator tile must fit in the operand stages");
    float* tile = reinterpret_cast<float*>(lds);
    const int c4 = (tid % CPR) * 4, rr0 = tid / CPR;
    const int col = n0 + c4;
    int ch = col;
    unsigned colb = (unsigned)col * ES;
    if constexpr (PYRG) {   // this thread's column group is the phase (pa + dpa, pb + dpb) of the same 64 channels: dpa output rows down, dpb pixels right
      const int q = col >> 6;
      ch = col & 63;
      colb = (unsigned)((group_dpa(q) * (p.Wo << 3) + group_dpb(q)) * 64 + ch) * ES;
    }
    const int ush = p.up_shift;
    const bool win = WING && p.win;
    const unsigned out_elems = win ? (unsigned)(p.N * p.Hin * p.Win * 4) * (unsigned)p.Cout : ((unsigned)p.M << (2 * ush)) * (unsigned)p.Cout;
    const unsigned out_bytes = __builtin_amdgcn_readfirstlane((int)(out_elems * ES));
    if (win) {
      ch = col & 63;
      colb = (unsigned)ch * ES;
    }
    const int wdy = (col >> 6) >> 1, wdx = (col >> 6) & 1;   // WING: this thread's column group
    const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.out), 0, out_bytes, 0x00020000);
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.residual ? p.residual : p.out), 0, p.residual ? out_bytes : 0u, 0x00020000);
    unsigned ro[PASSES];
    f32x4 res[PASSES];
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int m = m0 + rr0 + k * RPP;
      const int n = fast_div(m, p.mg_howo, p.sh_howo);
      const int rem = m - n * HoWo;
      const int oh = fast_div(rem, p.mg_wo, p.sh_wo);
      const int ow = rem - oh * p.Wo;
      // rows beyond M get an out-of-range offset: loads return zeros, stores are dropped by the hardware
      ro[k] = m < p.M ? (unsigned)(((((n * p.Ho + oh) << ush) + pa) * (p.Wo << ush) + (ow << ush) + pb) * p.Cout) * ES + colb : OOB;
      if (win) {   // window (oh, ow) of image n, output (2 oh - 1 + dy, 2 ow - 1 + dx) where it exists
        const int oy = 2 * oh - 1 + wdy, ox = 2 * ow - 1 + wdx;
        const bool ok = m < p.M && (unsigned)oy < (unsigned)(2 * p.Hin) && (unsigned)ox < (unsigned)(2 * p.Win);
        ro[k] = ok ? (unsigned)(((n * 2 * p.Hin + oy) * 2 * p.Win + ox) * p.Cout) * ES + colb : OOB;
      }
      res[k] = Elem<TO>::bload4(r_rsrc, ro[k]);   // (no residual: an empty descriptor, zeros)
    }
    __syncthreads();              // every wave is done reading the last operand stage
    {
      const int colq = lane & 31, rowq = (lane >> 5) * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            tile[(wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + rowq) * BN + wn * WN + j * 32 + colq] = acc[i][j][e];
    }
    __syncthreads();
    const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + ch) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 bi = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&tile[(rr0 + k * RPP) * BN + c4]) * sc + bi + res[k];
      if (p.relu) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
      }
      Elem<TO>::bstore4(o_rsrc, ro[k], v);
    }
    return;
  }

  // ---- epilogue B (HBM-bound launches: lateral + top-down sum, transposed conv; every bf16 output).  The accumulator
  // tile goes through LDS (the operand stages are free now) so that global traffic is row-major
  // float4: a wave touches 512 contiguous bytes of one pixel row per instruction.  Per-row index
  // arithmetic (the only integer divisions) is done once per block by the first BM threads.
  static_assert(BM * BN * 4 <= 2 * STAGE, "accumulator tile must fit in the operand stages");
  float* tile = reinterpret_cast<float*>(lds);
  constexpr int CPR = BN / 4;          // float4 chunks per tile row
  constexpr int RPP = 256 / CPR;       // rows per pass
  constexpr int PASSES = BM / RPP;
  const int c4 = (tid % CPR) * 4;      // this thread's 4 columns (fixed)
  const int rr0 = tid / CPR;           // its row in pass 0
  const int col = n0 + c4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // per tile row: offset of its up_residual pixel / shuffled output pixel.  Lives behind the accumulator tile
  // in the (now free) operand stages when there is room, so that the 64x64 variant needs exactly 32 KB of LDS
  // and five workgroups fit a CU (160 KB); only the 128x128 tile fills both stages and needs its own array.
  int* row_aux;
  if constexpr (BM * BN * 4 + BM * 4 <= 2 * STAGE) {
    row_aux = reinterpret_cast<int*>(lds + BM * BN * 4);
  } else {
    __shared__ int row_aux_own[BM];
    row_aux = row_aux_own;
  }
  __syncthreads();              // every wave is done reading the last operand stage
  {
    const int colq = lane & 31;
    const int rowq = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tile[(wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + rowq) * BN + wn * WN + j * 32 + colq] = acc[i][j][e];
  }
  if (tid < BM && (STORE != STORE_NHWC || p.out2)) {
    const int m = min(m0 + tid, p.M - 1);
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    const int oh = rem / p.Wo;
    const int ow = rem - oh * p.Wo;
    if constexpr (STORE == STORE_SHUFFLE2) row_aux[tid] = ((n * (2 * p.Ho) + 2 * oh) * (2 * p.Wo) + 2 * ow) * 64;
    else if constexpr (STORE == STORE_PHASE)
      row_aux[tid] = ((((n * p.Ho + oh) << p.up_shift) + pa) * (p.Wo << p.up_shift) + (ow << p.up_shift) + pb) * p.Cout;
    else row_aux[tid] = ((n * (p.Ho >> 1) + (oh >> 1)) * (p.Wo >> 1) + (ow >> 1)) * p.Cout;
  }
  __syncthreads();
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = zero4;
  const int ch = PYRG ? col & 63 : col;   // PYRG: the block's column groups are phases of the same 64 channels
  if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + ch);
  if (p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + ch);
  if constexpr (STORE == STORE_PHASE || STORE == STORE_NHWC) {
    if (!p.out2) {
      // plain and phase stores (f32 or bf16): buffer loads / stores at (row offset + column) - rows beyond M lie outside the
      // descriptor's range (NHWC) or get an out-of-range offset (PHASE) and are dropped by the hardware; no 64-bit index, no
      // compare-and-branch per row (the general form below spent a fifth of the short bf16 kernels on them)
      constexpr unsigned ES = sizeof(TO);
      const unsigned out_elems = STORE == STORE_PHASE ? ((unsigned)p.M << (2 * p.up_shift)) * (unsigned)p.Cout : (unsigned)p.M * (unsigned)p.Cout;
      const unsigned out_bytes = __builtin_amdgcn_readfirstlane((int)(out_elems * ES));
      const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.out), 0, out_bytes, 0x00020000);
      const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.residual ? p.residual : p.out), 0, p.residual ? out_bytes : 0u, 0x00020000);
      unsigned colb = (unsigned)col * ES;
      [[maybe_unused]] const unsigned rowb = (unsigned)p.Cout * ES;
      if constexpr (PYRG) {   // this thread's column group is the phase (pa + dpa, pb + dpb): dpa output rows down, dpb pixels right
        const int q = col >> 6;
        colb = (unsigned)((group_dpa(q) * (p.Wo << 3) + group_dpb(q)) * 64 + ch) * ES;
      }
      auto rows = [&](auto relu_c) {
        constexpr int G = PASSES < 4 ? PASSES : 4;
#pragma unroll
        for (int k0 = 0; k0 < PASSES; k0 += G) {
          unsigned ro[G];
          f32x4 res[G];
#pragma unroll
          for (int k = 0; k < G; ++k) {
            const int rr = rr0 + (k0 + k) * RPP;
            if constexpr (STORE == STORE_PHASE) ro[k] = m0 + rr < p.M ? (unsigned)row_aux[rr] * ES + colb : OOB;
            else ro[k] = (unsigned)(m0 + rr) * rowb + colb;
            res[k] = Elem<TO>::bload4(r_rsrc, ro[k]);   // (no residual: zeros)
          }
#pragma unroll
          for (int k = 0; k < G; ++k) {
            const int rr = rr0 + (k0 + k) * RPP;
            f32x4 v = *reinterpret_cast<const f32x4*>(&tile[rr * BN + c4]) * sc + bi + res[k];
            if constexpr (decltype(relu_c)::value) {
#pragma unroll
              for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
            }
            Elem<TO>::bstore4(o_rsrc, ro[k], v);
          }
        }
      };
      if (p.relu) rows(std::true_type{});
      else rows(std::false_type{});
      return;
    }
  }
  if constexpr (STORE != STORE_SHUFFLE2) {
    constexpr int G = PASSES < 4 ? PASSES : 4;  // rows in flight per thread: bounds the staging registers
#pragma unroll
    for (int k0 = 0; k0 < PASSES; k0 += G) {
      f32x4 up[G], res[G];
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int rr = rr0 + (k0 + k) * RPP;
        const int m = min(m0 + rr, p.M - 1);
        up[k] = p.out2 ? Elem<TO>::load4(p.up_residual, (size_t)row_aux[rr] + col) : zero4;
        const size_t ro = STORE == STORE_PHASE ? (size_t)row_aux[rr] + col : (size_t)m * p.Cout + col;
        res[k] = p.residual ? Elem<TO>::load4(p.residual, ro) : zero4;
      }
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const int rr = rr0 + (k0 + k) * RPP;
        const int m = m0 + rr;
        f32x4 v = *reinterpret_cast<const f32x4*>(&tile[rr * BN + c4]) * sc + bi + res[k];
        if (p.relu) {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        if (m < p.M) {
          const size_t o = STORE == STORE_PHASE ? (size_t)row_aux[rr] + col : (size_t)m * p.Cout + col;
          if (p.out) Elem<TO>::store4(p.out, o, v);
          if (p.out2) Elem<TO>::store4(p.out2, o, up[k] + v);  // upsample(x_in{k+1}) + x_in{k}, model.rs:126-137
        }
      }
    }
  } else {
    // conv_transpose2d k=2 s=2: column = (a*2+b)*64 + co -> out[n][2i+a][2j+b][co]
    const int t = col >> 6, co = col & 63;
    const int toff = ((t >> 1) * (2 * p.Wo) + (t & 1)) * 64 + co;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int rr = rr0 + k * RPP;
      f32x4 v = *reinterpret_cast<const f32x4*>(&tile[rr * BN + c4]) * sc + bi;
      if (p.relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
      }
      if (m0 + rr < p.M) Elem<TO>::store4(p.out, (size_t)row_aux[rr] + toff, v);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// libdivide-style unsigned division by an invariant d >= 1
static void make_magic(unsigned d, unsigned* magic, unsigned* shift) {
  unsigned L = 0;
  while ((1ull << L) < d) ++L;
  *magic = L == 0 ? 0u : (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
  *shift = L == 0 ? 0xFFFFFFFFu : L - 1;  // d == 1: identity
}

template <typename TI, typename TO, int BM, int BN, int KS, int STRIDE, int SRC, int STORE, bool X3 = false>
void launch_inst(const ConvDesc& d, hipStream_t s) {
  ConvArgs a{};
  const bool multi = d.src_mode == SRC_CAT4 || d.src_mode == SRC_PYR4;
  a.src = multi ? d.src_base : d.src[0];
  a.src_bytes = (unsigned)d.src_bytes;
  a.wgt_bytes = (unsigned)d.wgt_bytes;
  // element offsets of the CAT4 sources inside their shared allocation
  for (int i = 0; i < 4; ++i)
    a.src_off[i] = multi ? (int)((static_cast<const char*>(d.src[i]) - static_cast<const char*>(d.src_base)) / (long)sizeof(TI)) : 0;
  a.wgt = d.wgt;
  a.scale = d.scale;
  a.bias = d.bias;
  a.residual = d.residual;
  a.up_residual = d.up_residual;
  a.out = d.out;
  a.out2 = d.out2;
  a.N = d.N;
  a.Hin = d.Hin;
  a.Win = d.Win;
  a.Cin = d.Cin;
  a.Ho = d.Ho;
  a.Wo = d.Wo;
  a.Cout = d.Cout;
  a.M = d.N * d.Ho * d.Wo;
  a.pad = d.pad;
  a.relu = d.relu;
  a.nblk_n = d.Cout / BN;
  a.nblk_m = (a.M + BM - 1) / BM;
  a.up_shift = d.up == 8 ? 3 : d.up == 4 ? 2 : 1;
  a.batch = d.batch > 1 ? d.batch : 1;
  a.pyr_chunked = 1;
  a.pyr_nsrc = d.pyr_nsrc == 3 ? 3 : 4;
  a.pyr_group = d.pyr_group;
  a.debug = g_conv_debug;
  a.nblk = a.nblk_m * a.nblk_n * (STORE == STORE_PHASE ? d.up * d.up : a.batch);
  if (d.win) {   // rows = windows, two 128-column halves per row tile
    a.win = 1;
    a.Ho = d.Hin + 1;
    a.Wo = d.Win + 1;
    a.M = d.N * a.Ho * a.Wo;
    a.nblk_m = (a.M + BM - 1) / BM;
    a.nblk_n = 1;
    a.nblk = 2 * a.nblk_m;
  }
  if (SRC == SRC_PYR4 && d.pyr_group) {   // 30 virtual tiles of 128 columns per cell block (60 phases), or the four corner phases
    a.nblk_n = 1;
    a.nblk = a.nblk_m * (d.pyr_group == 1 ? 30 : 4);
  }
  make_magic((unsigned)(a.Ho * a.Wo), &a.mg_howo, &a.sh_howo);
  make_magic((unsigned)a.Wo, &a.mg_wo, &a.sh_wo);
  hipLaunchKernelGGL((conv_igemm<TI, TO, BM, BN, KS, STRIDE, SRC, STORE, X3>), dim3(a.nblk), dim3(256), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace igemm
using namespace igemm;

// Host-side shape checks: the kernel assumes exactly these, and an out-of-bounds
// access on the GPU can take the whole node down.
static void check(const ConvDesc& d) {
  const int eb = d.in_bf16 ? 2 : 4;
  const int bk = 128 / eb;
  // x3: f32 activations, weights as three bf16 planes (hi, mid, lo) of the f32 layout -> 6 bytes per weight
  const int ebw = d.x3 ? 6 : eb;
  if (d.pyr_group && (d.pyr_group < 0 || d.pyr_group > 2 || !(d.x3 || d.in_bf16) || d.src_mode != SRC_PYR4 || d.pyr_nsrc != 3))
    fail(OCR_ERR_INVALID, "%s: phase blocks (pyr_group %d) exist for the split-bf16 and bf16 PYR4 forms over p5, p4, p3", d.name, d.pyr_group);
  if (d.win && (!(d.x3 || d.in_bf16) || d.store_mode != STORE_PHASE || d.src_mode != SRC_PLAIN || d.up != 2 || d.ks != 2 || d.Cout != 64 ||
                (long long)d.N * (d.Hin + 1) * (d.Win + 1) >= (1ll << 31)))
    fail(OCR_ERR_INVALID, "%s: the window-indexed form exists for the split-bf16 / bf16 up-2 phase convs with 64 output channels", d.name);
  if (d.x3 && d.src_mode == SRC_PYR4 && d.pyr_nsrc != 3) fail(OCR_ERR_INVALID, "%s: the split-bf16 PYR4 form takes the three upsampled sources only", d.name);
  if (d.x3 && (d.in_bf16 || d.out_bf16 || d.src_mode == SRC_CAT4 || d.store_mode == STORE_SHUFFLE2))
    fail(OCR_ERR_INVALID, "%s: the split-bf16 form exists for f32 PLAIN / PYR4 convs with NHWC or PHASE stores", d.name);
  if (d.Cin % bk != 0) fail(OCR_ERR_INVALID, "%s: Cin %d not a multiple of %d", d.name, d.Cin, bk);
  if (d.Cout % 64 != 0) fail(OCR_ERR_INVALID, "%s: Cout %d not a multiple of 64", d.name, d.Cout);
  const bool phase2 = d.store_mode == STORE_PHASE;
  if (d.src_mode == SRC_PYR4) {
    // rows = cells of the p5 grid (Hin x Win), output [N][8 Hin][8 Win][64], four 64-channel sources in ONE allocation
    if (!phase2 || d.up != 8 || d.ks != 3 || d.stride != 1 || d.pad != 0 || d.Cin != 64 || d.Cout != 64 || d.Ho != d.Hin ||
        d.Wo != d.Win || d.out2 || !d.out || (d.out_bf16 && !d.in_bf16) || d.batch > 1)
      fail(OCR_ERR_INVALID, "%s: PYR4 needs the 64->64 bin_conv1 form on the p5 grid with an up-8 PHASE store", d.name);
    // (rows beyond M are dropped through the out-of-range marker 2^31 as their byte offset: the output must end below it)
    if ((long long)d.N * d.Ho * d.Wo * 64 * d.Cout >= (1ll << 29)) fail(OCR_ERR_INVALID, "%s: PYR4 output too large (2^31 bytes)", d.name);
    if ((long long)d.wgt_bytes != 64ll * d.Cout * 21 * 64 * ebw) fail(OCR_ERR_INVALID, "%s: PYR4 weight bytes", d.name);
    if ((long long)d.src_bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "%s: sources must be < 2^31 bytes; split the batch", d.name);
    for (int i = 0; i < 4; ++i) {
      const long long need = (long long)d.N * (d.Hin << i) * (d.Win << i) * 64 * eb;
      const long long off = static_cast<const char*>(d.src[i]) - static_cast<const char*>(d.src_base);
      if (!d.src[i] || off < 0 || off % eb || off + need > (long long)d.src_bytes)
        fail(OCR_ERR_INVALID, "%s: PYR4 source %d lies outside the shared allocation", d.name, i);
    }
    if (!d.wgt) fail(OCR_ERR_INVALID, "%s: null operand", d.name);
    return;
  }
  if (phase2) {
    // up*up 2x2 phase convs on the low-res grid: out is [N][up Ho][up Wo][Cout], wgt [up*up][Cout][2x2][Cin]
    if (d.up != 2 && d.up != 4 && d.up != 8) fail(OCR_ERR_INVALID, "%s: PHASE store with up = %d", d.name, d.up);
    if (d.ks != 2 || d.stride != 1 || d.pad != 1 || d.Ho != d.Hin || d.Wo != d.Win || d.out2 || d.src_mode != SRC_PLAIN || !d.out)
      fail(OCR_ERR_INVALID, "%s: PHASE store needs a 2x2 s1 pad-1 conv on the low-res grid", d.name);
    // the f32 epilogue addresses it through one buffer descriptor and drops rows beyond M through the out-of-range marker 2^31
    if ((long long)d.N * d.Ho * d.Wo * d.up * d.up * d.Cout >= (1ll << 29))
      fail(OCR_ERR_INVALID, "%s: PHASE output too large (2^31 bytes)", d.name);
  } else {
    if (d.ks != 1 && d.ks != 3) fail(OCR_ERR_INVALID, "%s: kernel size %d", d.name, d.ks);
    if (d.stride != 1 && d.stride != 2) fail(OCR_ERR_INVALID, "%s: stride %d", d.name, d.stride);
    if (d.pad != (d.ks - 1) / 2) fail(OCR_ERR_INVALID, "%s: pad %d", d.name, d.pad);
    if (d.Ho != (d.Hin + 2 * d.pad - d.ks) / d.stride + 1 || d.Wo != (d.Win + 2 * d.pad - d.ks) / d.stride + 1)
      fail(OCR_ERR_INVALID, "%s: output grid %dx%d does not follow from input %dx%d", d.name, d.Ho, d.Wo, d.Hin, d.Win);
  }
  if ((long long)d.N * d.Ho * d.Wo >= (1ll << 31)) fail(OCR_ERR_INVALID, "%s: M overflows int", d.name);
  const int nb = d.batch > 1 ? d.batch : 1;
  if (nb > 1 && (d.ks != 1 || d.stride != 1 || d.src_mode != SRC_PLAIN || d.store_mode != STORE_NHWC || d.residual || d.out2 ||
                 d.in_bf16 || d.out_bf16 || !d.out))
    fail(OCR_ERR_INVALID, "%s: batched GEMM needs a plain f32 1x1 s1 conv without residual", d.name);
  if ((long long)nb * d.N * d.Ho * d.Wo * d.Cout >= (1ll << 40)) fail(OCR_ERR_INVALID, "%s: batched output too large", d.name);
  // the kernel addresses its operands with 32-bit BYTE offsets below the out-of-range marker 2^31
  const long long in_bytes = d.src_mode == SRC_CAT4 ? (long long)d.src_bytes : (long long)nb * d.N * d.Hin * d.Win * d.Cin * eb;
  if (in_bytes >= (1ll << 31) || (long long)d.src_bytes >= (1ll << 31) || (long long)d.src_bytes < in_bytes)
    fail(OCR_ERR_INVALID, "%s: input of %lld bytes (addressable %zu) must be < 2^31 bytes; split the batch", d.name, in_bytes, d.src_bytes);
  if ((long long)d.wgt_bytes != (long long)(phase2 ? d.up * d.up : nb) * d.Cout * d.ks * d.ks * d.Cin * ebw) fail(OCR_ERR_INVALID, "%s: weight bytes", d.name);
  if ((long long)d.wgt_bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "%s: weights too large", d.name);
  if (d.src_mode != SRC_PLAIN && d.src_mode != SRC_CAT4) fail(OCR_ERR_INVALID, "%s: source mode %d", d.name, d.src_mode);
  if (d.src_mode == SRC_CAT4) {
    if (d.Cin != 256 || ((d.Hin | d.Win) & 7) || d.ks != 3 || d.stride != 1 || d.Cout != 64)
      fail(OCR_ERR_INVALID, "%s: CAT4 needs a 3x3 s1 256->64 conv on a grid divisible by 8", d.name);
    for (int i = 0; i < 4; ++i) {
      const long long need = (long long)d.N * (d.Hin >> (3 - i)) * (d.Win >> (3 - i)) * 64 * eb;
      const long long off = static_cast<const char*>(d.src[i]) - static_cast<const char*>(d.src_base);
      if (!d.src[i] || off < 0 || off % eb || off + need > (long long)d.src_bytes)
        fail(OCR_ERR_INVALID, "%s: CAT4 source %d lies outside the shared allocation", d.name, i);
    }
  }
  if (d.store_mode == STORE_SHUFFLE2 && (d.Cout != 256 || d.ks != 1 || d.residual || d.out2 || d.in_bf16 || d.out_bf16))
    fail(OCR_ERR_INVALID, "%s: SHUFFLE2 store needs a plain f32 1x1 conv with Cout 4*64", d.name);
  if (d.out2 && (!d.up_residual || ((d.Ho | d.Wo) & 1))) fail(OCR_ERR_INVALID, "%s: out2 needs up_residual and an even grid", d.name);
  if (!d.src[0] || !d.wgt || (!d.out && !d.out2)) fail(OCR_ERR_INVALID, "%s: null operand", d.name);
  if (!d.in_bf16 && d.out_bf16) fail(OCR_ERR_INVALID, "%s: f32 operands with bf16 output is not instantiated", d.name);
  if (d.in_bf16 && !d.out_bf16 && d.src_mode != SRC_CAT4 && !(phase2 || (d.ks == 3 && d.stride == 1)))
    fail(OCR_ERR_INVALID, "%s: bf16 -> f32 exists for the CAT4, 3x3 s1 and PHASE convs only", d.name);
}

// Tile choice (measured per layer shape with tools/bench_conv_tiles.py, profiles/r01_tile_scan.txt).
// f32: the MFMA time of a K-step is long (64 cycles per 32x32x2), so the small tile loses nothing in the
// loop and wins on residency (5 workgroups per CU hide each other's prologue / epilogue / barrier waits)
// and on the last partial round of tiles: 64x64 for every 2x2 / 3x3 conv.  The 1x1 convs are HBM-bound and
// keep the largest tile that still leaves >= 8 tiles per CU.
// bf16: a K-step is 4x shorter, operand traffic per flop decides: 128x128 for the deep layers (Cin >= 256),
// 128x64 otherwise.
enum Tile { T128x128, T128x64, T64x64 };
static int g_tile_override = 0;  // tuning aid (ocr_test_set_conv_tile): 1 = 128x128, 2 = 128x64, 3 = 64x64
void set_conv_tile_override(int t) { g_tile_override = t; }
static Tile pick_tile(const ConvDesc& d) {
  if (d.x3) {  // the split-bf16 kernels exist as 128 x 128 and 128 x 64 only
    const long long M = (long long)d.N * d.Ho * d.Wo;
    const int reps = d.store_mode == STORE_PHASE ? d.up * d.up : (d.batch > 1 ? d.batch : 1);
    (void)M;
    (void)reps;
    // 128 x 128 wherever Cout allows it, however few tiles that leaves (measured on every launch shape of the detector,
    // tools/profile_layers.py with the tile override: operand DMA per MFMA is what the wide tile saves)
    if (d.src_mode == SRC_PYR4) return d.pyr_group == 1 ? T128x128 : T128x64;
    if (d.win) return T128x128;
    if (g_tile_override == 2 || d.Cout % 128) return T128x64;
    return T128x128;
  }
  if (d.win) return T128x128;
  if (g_tile_override == 1 && d.Cout % 128 == 0) return T128x128;
  if (g_tile_override == 2) return T128x64;
  if (g_tile_override == 3) return T64x64;
  const long long M = (long long)d.N * d.Ho * d.Wo;
  const int reps = d.store_mode == STORE_PHASE ? d.up * d.up : (d.batch > 1 ? d.batch : 1);
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (d.Cout / bn) * reps; };
  if (d.src_mode == SRC_PYR4) return d.pyr_group == 1 ? T128x128 : T64x64;  // bf16 too: 0.254 vs 0.265 ms with 128x64
  // bf16 3x3: the wide tile from 256 input channels on, and for every stride-2 conv (0.070 -> 0.064, 0.052 -> 0.046 ms at layer2 / layer3)
  if (d.in_bf16 && d.ks > 1) return (d.Cout % 128 == 0 && (d.Cin >= 256 || d.stride == 2)) ? T128x128 : T128x64;
  if ((d.ks > 1 || d.batch > 1) && d.src_mode == SRC_PLAIN) return T64x64;
  if (d.Cout % 128 == 0 && blocks(128, 128) >= 2048) return T128x128;
  if (blocks(128, 64) >= 2048) return T128x64;
  return T64x64;
}

static const char* tile_name(Tile t) { return t == T128x128 ? "128x128" : t == T128x64 ? "128x64" : "64x64"; }

const char* conv_igemm_kernel_name(const ConvDesc& d) {
  static thread_local char buf[96];
  const bool wide = d.x3 && d.wide && conv_x3_wide_applicable(d);
  snprintf(buf, sizeof buf, "conv_igemm_%s<%s,k%d,s%d,%s%s>", d.x3 ? "x3" : d.in_bf16 ? "bf16" : "f32", wide ? "256x128" : tile_name(pick_tile(d)), d.ks, d.stride,
           d.src_mode == SRC_CAT4 ? "CAT4" : d.src_mode == SRC_PYR4 ? "PYR4" : "PLAIN", d.store_mode == STORE_SHUFFLE2 ? ",SHUFFLE2" : d.store_mode == STORE_PHASE ? (d.up == 2 ? ",PHASE2" : d.up == 4 ? ",PHASE4" : ",PHASE8")
           : d.batch > 1 ? ",BATCHED" : "");
  // names must outlive the call: intern them
  static thread_local std::vector<std::string>* pool = new std::vector<std::string>();
  for (const auto& s : *pool)
    if (s == buf) return s.c_str();
  pool->reserve(64);
  pool->push_back(buf);
  return pool->back().c_str();
}

template <typename TI, typename TO, int KS, int STRIDE, int SRC, int STORE>
static void launch_tiles(const ConvDesc& d, hipStream_t s) {
  switch (pick_tile(d)) {
    case T128x128: launch_inst<TI, TO, 128, 128, KS, STRIDE, SRC, STORE>(d, s); break;
    case T128x64: launch_inst<TI, TO, 128, 64, KS, STRIDE, SRC, STORE>(d, s); break;
    case T64x64: launch_inst<TI, TO, 64, 64, KS, STRIDE, SRC, STORE>(d, s); break;
  }
}

// f32 -> bf16, round to nearest even (weights are finite)
static inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

std::vector<uint16_t> split3_weights(const float* w, size_t count) {
  if (count % 16) fail(OCR_ERR_INTERNAL, "split3_weights: %zu weights are not whole groups of 16", count);
  std::vector<uint16_t> out(3 * count);
  static const int perm[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};  // position -> k inside the group
  for (size_t g = 0; g < count; g += 16)
    for (int pos = 0; pos < 16; ++pos) {
      const float x = w[g + perm[pos]];
      const uint16_t h = bf16_rne(x);
      const float r1 = x - bf16_f32(h);      // exact
      const uint16_t m = bf16_rne(r1);
      const float r2 = r1 - bf16_f32(m);     // exact, at most 8 significant bits left
      out[g + pos] = h;
      out[count + g + pos] = m;
      out[2 * count + g + pos] = bf16_rne(r2);
    }
  return out;
}

std::vector<uint16_t> split3_weights_tiled(const float* w, size_t count, int wrow) {
  if (wrow <= 0 || wrow % 32 || count % ((size_t)wrow * 16)) fail(OCR_ERR_INTERNAL, "split3_weights_tiled: %zu weights in rows of %d are not whole blocks of 16 rows x 32", count, wrow);
  const std::vector<uint16_t> planes = split3_weights(w, count);
  std::vector<uint16_t> out(3 * count);
  const size_t rows = count / wrow, nK = wrow / 32;
  for (size_t r16 = 0; r16 < rows / 16; ++r16)
    for (size_t ks = 0; ks < nK; ++ks)
      for (int pl = 0; pl < 3; ++pl) {
        uint16_t* blk = &out[((r16 * nK + ks) * 3 + pl) * 512];
        for (int r = 0; r < 16; ++r)
          for (int sl = 0; sl < 4; ++sl) {
            const uint16_t* src = &planes[(size_t)pl * count + (r16 * 16 + r) * wrow + ks * 32 + (size_t)((sl ^ ((r >> 2) & 3)) * 8)];
            for (int e = 0; e < 8; ++e) blk[r * 32 + sl * 8 + e] = src[e];
          }
      }
  return out;
}

template <int KS, int STRIDE, int SRC, int STORE>
static void launch_x3(const ConvDesc& d, hipStream_t s) {
  if (pick_tile(d) == T128x128) launch_inst<float, float, 128, 128, KS, STRIDE, SRC, STORE, true>(d, s);
  else launch_inst<float, float, 128, 64, KS, STRIDE, SRC, STORE, true>(d, s);
}

// CUs of the current device (one query per process and device index)
static int device_cus() {
  static int cached[64] = {0};
  int dev = 0;
  OCR_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cached[dev]) {
    int n = 0;
    OCR_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cached[dev] = n > 0 ? n : 256;
  }
  return cached[dev];
}

void launch_conv_igemm(const ConvDesc& d, hipStream_t s) {
  check(d);
  if (d.x3 && d.wide && conv_x3_wide_applicable(d)) return launch_conv_x3_wide(d, device_cus(), s);
  if (d.x3) {
    if (d.src_mode == SRC_PYR4 && d.pyr_group == 1) return launch_inst<float, float, 128, 128, 3, 1, SRC_PYR4, STORE_PHASE, true>(d, s);
    if (d.src_mode == SRC_PYR4) return launch_inst<float, float, 128, 64, 3, 1, SRC_PYR4, STORE_PHASE, true>(d, s);
    if (d.store_mode == STORE_PHASE) return launch_x3<2, 1, SRC_PLAIN, STORE_PHASE>(d, s);
    if (d.ks == 3 && d.stride == 1) return launch_x3<3, 1, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 3 && d.stride == 2) return launch_x3<3, 2, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 1 && d.stride == 1) return launch_x3<1, 1, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 1 && d.stride == 2) return launch_x3<1, 2, SRC_PLAIN, STORE_NHWC>(d, s);
    fail(OCR_ERR_INVALID, "%s: no split-bf16 conv_igemm variant for ks=%d stride=%d", d.name, d.ks, d.stride);
  }
  if (d.in_bf16) {
    if (d.src_mode == SRC_PYR4 && d.pyr_group == 1 && !d.out_bf16) return launch_inst<__bf16, float, 128, 128, 3, 1, SRC_PYR4, STORE_PHASE>(d, s);
    if (d.src_mode == SRC_PYR4 && d.pyr_group == 1) return launch_inst<__bf16, __bf16, 128, 128, 3, 1, SRC_PYR4, STORE_PHASE>(d, s);
    if (d.src_mode == SRC_PYR4 && !d.out_bf16) return launch_inst<__bf16, float, 64, 64, 3, 1, SRC_PYR4, STORE_PHASE>(d, s);
    if (d.src_mode == SRC_PYR4) return launch_inst<__bf16, __bf16, 64, 64, 3, 1, SRC_PYR4, STORE_PHASE>(d, s);
    if (d.store_mode == STORE_PHASE && !d.out_bf16) return launch_tiles<__bf16, float, 2, 1, SRC_PLAIN, STORE_PHASE>(d, s);
    if (d.store_mode == STORE_PHASE) return launch_tiles<__bf16, __bf16, 2, 1, SRC_PLAIN, STORE_PHASE>(d, s);
    if (d.ks == 3 && d.stride == 1 && !d.out_bf16 && d.src_mode == SRC_PLAIN) return launch_tiles<__bf16, float, 3, 1, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.src_mode == SRC_CAT4 && !d.out_bf16) return launch_tiles<__bf16, float, 3, 1, SRC_CAT4, STORE_NHWC>(d, s);
    if (d.src_mode == SRC_CAT4) return launch_tiles<__bf16, __bf16, 3, 1, SRC_CAT4, STORE_NHWC>(d, s);
    if (d.ks == 3 && d.stride == 1) return launch_tiles<__bf16, __bf16, 3, 1, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 3 && d.stride == 2) return launch_tiles<__bf16, __bf16, 3, 2, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 1 && d.stride == 1) return launch_tiles<__bf16, __bf16, 1, 1, SRC_PLAIN, STORE_NHWC>(d, s);
    if (d.ks == 1 && d.stride == 2) return launch_tiles<__bf16, __bf16, 1, 2, SRC_PLAIN, STORE_NHWC>(d, s);
    fail(OCR_ERR_INVALID, "%s: no bf16 conv_igemm variant for ks=%d stride=%d", d.name, d.ks, d.stride);
  }
  if (d.store_mode == STORE_SHUFFLE2) return launch_tiles<float, float, 1, 1, SRC_PLAIN, STORE_SHUFFLE2>(d, s);
  if (d.src_mode == SRC_PYR4) return launch_inst<float, float, 64, 64, 3, 1, SRC_PYR4, STORE_PHASE>(d, s);
  if (d.store_mode == STORE_PHASE) return launch_tiles<float, float, 2, 1, SRC_PLAIN, STORE_PHASE>(d, s);
  if (d.src_mode == SRC_CAT4) return launch_tiles<float, float, 3, 1, SRC_CAT4, STORE_NHWC>(d, s);
  if (d.ks == 3 && d.stride == 1) return launch_tiles<float, float, 3, 1, SRC_PLAIN, STORE_NHWC>(d, s);
  if (d.ks == 3 && d.stride == 2) return launch_tiles<float, float, 3, 2, SRC_PLAIN, STORE_NHWC>(d, s);
  if (d.ks == 1 && d.stride == 1) return launch_tiles<float, float, 1, 1, SRC_PLAIN, STORE_NHWC>(d, s);
  if (d.ks == 1 && d.stride == 2) return launch_tiles<float, float, 1, 2, SRC_PLAIN, STORE_NHWC>(d, s);
  fail(OCR_ERR_INVALID, "%s: no conv_igemm variant for ks=%d stride=%d", d.name, d.ks, d.stride);
}

}  // namespace ocr
