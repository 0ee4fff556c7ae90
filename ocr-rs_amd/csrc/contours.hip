// Contour tracing on the device (/root/reference/src/text_detection/metrics.rs:78, imageproc::contours::find_contours):
// the Suzuki-Abe border following of postproc_geom.cpp::find_contours_bits, statement for statement, one wave per image.
//
// The algorithm is sequential inside an image - whether a pixel starts a border depends on the labels every earlier border left
// behind - and that order is part of the contract (start points and the order of the contours decide what Douglas-Peucker keeps),
// so it is NOT re-derived in a parallel form: an image's packed bit map and its two label bit planes (has a border passed / did
// one leave the pixel negative) live in the wave's LDS (3 x H W / 8 bytes: 150 KB at 640 x 640), the 64 lanes find the run
// boundaries of a row together (the only pixels that can start a border), and the border following itself runs wave-uniform on
// scalar values - every step ONE round trip to LDS (the three rows of the 3 x 3 neighbourhood, read together), the search for the
// next border pixel as bit arithmetic on the neighbour byte, two fire-and-forget label bit ORs and one 4-byte store of the point.
// One wave walks at 0.44 us per border pixel (an instruction every four to five cycles, ~60 dependent scalar instructions and one LDS
// round trip per step) against 13-20 ns on a host core: 5-6 ms for a batch of dense maps.  The parallel form further down is what
// the engine uses (option device_contours=1; =2 is this form); both are off by default - DESIGN.md section 4 has the measurements.
//
// Outputs per image: points (y << 16 | x) in tracing order, the start offset of every contour (+ a sentinel), a header
// {contours, points, status}.  status != 0 (more contours / points than the buffers hold, or the iteration guard) sends that image
// to the host tracer; a second kernel packs the good images' points and contour lengths densely for one copy home.
#include "common.hpp"

namespace ocr {
namespace {

constexpr int kLdsWords = 39808;            // 155.5 KB of the CU's 160 (the parallel form adds 4 KB of scan scratch)
constexpr int kContourPool = 1 << 17;       // points of the speculative walks per image (parallel form)
constexpr unsigned DXP = 0u | 0u << 2 | 1u << 4 | 2u << 6 | 2u << 8 | 2u << 10 | 1u << 12 | 0u << 14;   // kDx[d] + 1, two bits per direction
constexpr unsigned DYP = 1u | 0u << 2 | 0u << 4 | 0u << 6 | 1u << 8 | 2u << 10 | 2u << 12 | 2u << 14;   // kDy[d] + 1
__device__ __forceinline__ int ddx(int d) { return (int)((DXP >> (2 * d)) & 3u) - 1; }   // W NW N NE E SE S SW (clockwise, y down)
__device__ __forceinline__ int ddy(int d) { return (int)((DYP >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ __launch_bounds__(64) void contour_trace_kernel(const uint32_t* __restrict__ bits_all, int wpi, int h, int w, uint32_t* __restrict__ pts_all, int cap,
                                                          int* __restrict__ starts_all, int maxc, int* __restrict__ hdr_all) {
  __shared__ uint32_t lds[kLdsWords];
  const int img = blockIdx.x, lane = threadIdx.x;
  const unsigned npx = (unsigned)h * (unsigned)w, nw = (npx + 31) / 32;
  uint32_t* bits = lds;                       // nw + 1 words (get3 reads the word behind as well)
  volatile uint32_t* seen = lds + nw + 1;     // a border has passed this pixel
  volatile uint32_t* neg = lds + 2 * nw + 1;  // ... and left it negative (it was a right edge)
  const uint32_t* g = bits_all + (size_t)img * wpi;
  for (unsigned i = lane; i < nw; i += 64) {
    bits[i] = g[i];
    seen[i] = 0;
    neg[i] = 0;
  }
  if (lane == 0) bits[nw] = 0;
  __syncthreads();

  uint32_t* pts = pts_all + (size_t)img * cap;
  int* starts = starts_all + (size_t)img * (maxc + 1);
  int ncont = 0, npts = 0, status = 0;
  auto fg = [&](unsigned i) -> unsigned { return (bits[i >> 5] >> (i & 31)) & 1u; };
  auto nz = [&](int x, int y) -> bool { return x >= 0 && x < w && y >= 0 && y < h && fg((unsigned)y * w + x); };
  // bits s, s + 1, s + 2: both words unconditionally (no branch between the three rows' reads: one wait for all of them)
  auto get3 = [&](unsigned s) -> unsigned {
    const unsigned q = s >> 5, sh = s & 31;
    const unsigned long long v = (unsigned long long)bits[q] | ((unsigned long long)bits[q + 1] << 32);
    return (unsigned)(v >> sh) & 7u;
  };
  auto emit = [&](int x, int y) {
    if (npts >= cap) {
      status = 1;
      return;
    }
    if (lane == 0) pts[npts] = ((uint32_t)y << 16) | (uint32_t)x;
    ++npts;
  };
  auto mark = [&](int x, int y, bool negative) {
    const unsigned i = (unsigned)y * w + x;
    if (lane == 0) {   // ds_or_b32 without return: nothing waits for it; LDS executes a wave's operations in order
      atomicOr(const_cast<uint32_t*>(seen) + (i >> 5), 1u << (i & 31));
      if (negative) atomicOr(const_cast<uint32_t*>(neg) + (i >> 5), 1u << (i & 31));
    }
  };
  const int guard = (int)(8u * npx + 64u);   // a border passes a pixel at most a few times: an exit every wave reaches
  auto trace = [&](int x, int y, int start) {   // start: direction of the adjacent zero pixel (0 = W, 4 = E)
    if (ncont >= maxc) {
      status = 1;
      return;
    }
    if (lane == 0) starts[ncont] = npts;
    ++ncont;
    int p1x = 0, p1y = 0, d1 = 0;
    bool found = false;
    for (int k = 0; k < 8 && !found; ++k) {   // clockwise from the adjacent zero pixel
      const int d = (start + k) & 7;
      if (nz(x + ddx(d), y + ddy(d))) {
        p1x = x + ddx(d);
        p1y = y + ddy(d);
        d1 = d;
        found = true;
      }
    }
    found = uni(found);
    if (!found) {
      emit(x, y);
      mark(x, y, true);
      return;
    }
    p1x = uni(p1x);
    p1y = uni(p1y);
    // the walk: position and direction are wave-uniform (scalar registers); per step one round trip to LDS for the neighbourhood,
    // and ONE lane-0 region with the point store and the label ORs
    int p3x = x, p3y = y, base = uni(d1);
    for (int it = 0;; ++it) {
      if (it > guard || npts >= cap) {
        status = npts >= cap ? 1 : 2;
        return;
      }
      const unsigned i3 = (unsigned)p3y * w + p3x;
      int dn = 0;
      bool right_edge = false;
      if (p3x > 0 && p3y > 0 && p3x + 1 < w && p3y + 1 < h) {
        const unsigned top = get3(i3 - w - 1), mid = get3(i3 - 1), bot = get3(i3 + w - 1);
        const unsigned m = (mid & 1u) | (top & 1u) << 1 | (top & 2u) << 1 | (top & 4u) << 1 | (mid & 4u) << 2 | (bot & 4u) << 3 | (bot & 2u) << 5 | (bot & 1u) << 7;
        // the step table of postproc_geom.cpp (StepTable) as arithmetic: bit j of r = neighbour (base + j) & 7; the search goes
        // counter-clockwise from base - 1, i.e. j = 7, 6, ... 0 (j = 0 is the pixel we came from: set) - the first hit is r's
        // highest set bit; the E neighbour (j_E) counts as examined-and-zero when it lies above that bit
        const unsigned mu = (unsigned)uni((int)m);
        const unsigned r = (((mu | mu << 8) >> base) & 0xffu) | 1u;
        const int j = 31 - __builtin_clz(r);
        dn = (base + j) & 7;
        right_edge = ((4 - base) & 7) > j;
      } else {
        for (int k = 1; k <= 8; ++k) {   // counter-clockwise, starting just after the previous pixel's direction
          const int d = (base - k) & 7;
          if (nz(p3x + ddx(d), p3y + ddy(d))) {
            dn = d;
            break;
          }
          if (d == 4) right_edge = true;
        }
        dn = uni(dn);
        right_edge = uni(right_edge);
      }
      if (lane == 0) {
        pts[npts] = ((uint32_t)p3y << 16) | (uint32_t)p3x;
        atomicOr(const_cast<uint32_t*>(seen) + (i3 >> 5), 1u << (i3 & 31));
        if (p3x + 1 == w || right_edge) atomicOr(const_cast<uint32_t*>(neg) + (i3 >> 5), 1u << (i3 & 31));
      }
      ++npts;
      const int p4x = p3x + ddx(dn), p4y = p3y + ddy(dn);
      if (p4x == x && p4y == y && p3x == p1x && p3y == p1y) break;
      p3x = p4x;
      p3y = p4y;
      base = (dn + 4) & 7;
    }
  };

  // raster scan over the run boundaries: 64 lanes take the words of a row (w <= 2048, a multiple of 32)
  const int wr = w >> 5;
  for (int y = 0; y < h && !status; ++y) {
    uint32_t cand = 0;
    if (lane < wr) {
      const uint32_t* row = bits + (size_t)y * wr;
      const uint32_t cur = row[lane];
      const uint32_t lbit = lane > 0 ? row[lane - 1] >> 31 : 0u;
      const uint32_t rbit = lane + 1 < wr ? row[lane + 1] & 1u : 0u;
      cand = (cur & ~((cur << 1) | lbit)) | (cur & ~((cur >> 1) | (rbit << 31)));   // first and last pixel of every run
    }
    unsigned long long words = __ballot(cand != 0);
    while (words && !status) {
      const int wi = uni(__builtin_ctzll(words));
      words &= words - 1;
      uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cand, wi);
      while (c && !status) {
        const int x = wi * 32 + uni(__builtin_ctz(c));
        c &= c - 1;
        const unsigned i = (unsigned)y * w + x;
        const unsigned sb = uni((int)((seen[i >> 5] >> (i & 31)) & 1u)), nb = uni((int)((neg[i >> 5] >> (i & 31)) & 1u));
        if (!sb && x > 0 && !uni((int)fg(i - 1))) trace(x, y, 0);                     // value == 1, W neighbour 0: outer border
        else if (!nb && x + 1 < w && !uni((int)fg(i + 1))) trace(x, y, 4);            // value > 0, E neighbour 0: hole border
      }
    }
  }
  if (lane == 0) {
    if (ncont <= maxc) starts[ncont] = npts;
    int* hdr = hdr_all + 4 * img;
    hdr[0] = ncont;
    hdr[1] = npts;
    hdr[2] = status;
    hdr[3] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same contours, most of the walking done in parallel.  A border is a pure function of the bit image and its start (pixel,
// type); what is sequential is only WHICH candidates start one - the label tests of the raster scan.  So:
//   A  every thread of a 1024-thread workgroup takes a row and lists its PLAUSIBLE starts in raster order - the first pixel of a run
//      with no foreground above it (the only place a component's first pixel can be: an outer border) and the last pixel of a run
//      whose gap to the next run has no background above it (the only place the pixel left of a hole's first pixel can be: a hole
//      border) - then each listed start is walked by its own lane, once for the length and once, behind a prefix sum, for the points
//      (bit 31 of a point: this visit leaves the pixel negative);
//   B  wave 0 replays the raster scan of the sequential kernel over ALL run boundaries with the label planes; a candidate that
//      passes its label test takes its border from the list - 64 lanes OR the labels in and copy the points out - instead of walking.
// A candidate that passes its test and is NOT in the list (Suzuki-Abe says there is none; the implementation's x > 0 rule for outer
// starts makes some for components whose first pixel sits in column 0) ends the image with status 3: it goes to the host tracer like an
// overflow.  Exact by construction: the same decisions on the same labels, the same walks.
constexpr int kMaxStarts = 8192;     // plausible starts per image
__device__ __forceinline__ bool span_any(const uint32_t* row, int xa, int xb) {   // a set bit in [xa, xb]
  for (int q = xa >> 5; q <= (xb >> 5); ++q) {
    uint32_t m = ~0u;
    if (q == (xa >> 5)) m &= ~0u << (xa & 31);
    if (q == (xb >> 5)) m &= ~0u >> (31 - (xb & 31));
    if (row[q] & m) return true;
  }
  return false;
}
__device__ __forceinline__ bool span_all(const uint32_t* row, int xa, int xb) {   // every bit of [xa, xb] set
  for (int q = xa >> 5; q <= (xb >> 5); ++q) {
    uint32_t m = ~0u;
    if (q == (xa >> 5)) m &= ~0u << (xa & 31);
    if (q == (xb >> 5)) m &= ~0u >> (31 - (xb & 31));
    if ((row[q] & m) != m) return false;
  }
  return true;
}
// first set (want = 1) / clear (want = 0) bit of the row at or after x, or w
__device__ __forceinline__ int next_bit(const uint32_t* row, int w, int x, int want) {
  while (x < w) {
    uint32_t v = row[x >> 5];
    if (!want) v = ~v;
    v &= ~0u << (x & 31);
    if (v) return min((x & ~31) + __builtin_ctz(v), w);
    x = (x & ~31) + 32;
  }
  return w;
}
// the plausible starts of row y in raster order: key = 2 * pixel index + type (0 outer, 1 hole); out == nullptr: count only
__device__ int row_starts(const uint32_t* bits, int h, int w, int y, int* out) {
  const uint32_t* row = bits + (size_t)y * (w >> 5);
  const uint32_t* up = y > 0 ? row - (w >> 5) : nullptr;
  int n = 0, prev_x1 = -1, x = 0;
  while (x < w) {
    const int x0 = next_bit(row, w, x, 1);
    if (x0 >= w) break;
    const int x1 = next_bit(row, w, x0, 0) - 1;
    if (prev_x1 >= 0 && up && span_all(up, prev_x1 + 1, x0 - 1)) {   // the gap is the top row of a hole
      if (out) out[n] = 2 * (y * w + prev_x1) + 1;
      ++n;
    }
    if (x0 > 0 && (!up || !span_any(up, max(x0 - 1, 0), min(x1 + 1, w - 1)))) {   // nothing of the component above this run
      if (out) out[n] = 2 * (y * w + x0);
      ++n;
    }
    prev_x1 = x1;
    x = x1 + 1;
  }
  return n;
}
// Pool entry of a walked point: x in bits 0-11, y in bits 12-27, bit 28 = the pixel is the FIRST of its row's run (W neighbour background or
// x == 0), bit 29 = the LAST of its run (E neighbour background or x + 1 == w), bit 31 = this visit leaves the pixel negative.  The two run
// flags are what lets the scan keep ONE label bit per pixel (see the kernel): they come for free from the neighbourhood the walk reads anyway.
constexpr uint32_t kPtRunStart = 1u << 28, kPtRunEnd = 1u << 29, kPtNeg = 1u << 31;
// one border walked by one lane: its length (-1: longer than limit); with out != nullptr the points too
__device__ int walk_border(const uint32_t* bits, int h, int w, int x, int y, int start, uint32_t* out, int limit) {
  auto fg = [&](unsigned i) -> unsigned { return (bits[i >> 5] >> (i & 31)) & 1u; };
  auto nz = [&](int xx, int yy) -> bool { return xx >= 0 && xx < w && yy >= 0 && yy < h && fg((unsigned)yy * w + xx); };
  auto get3 = [&](unsigned s) -> unsigned {
    const unsigned q = s >> 5, sh = s & 31;
    const unsigned long long v = (unsigned long long)bits[q] | ((unsigned long long)bits[q + 1] << 32);
    return (unsigned)(v >> sh) & 7u;
  };
  int p1x = 0, p1y = 0, d1 = 0;
  bool found = false;
  for (int k = 0; k < 8 && !found; ++k) {
    const int d = (start + k) & 7;
    if (nz(x + ddx(d), y + ddy(d))) {
      p1x = x + ddx(d);
      p1y = y + ddy(d);
      d1 = d;
      found = true;
    }
  }
  if (!found) {   // an isolated pixel: a run of its own
    if (out) out[0] = ((uint32_t)y << 12) | (uint32_t)x | kPtNeg | kPtRunStart | kPtRunEnd;
    return 1;
  }
  int p3x = x, p3y = y, base = d1, n = 0;
  for (;;) {
    if (n >= limit) return -1;
    const unsigned i3 = (unsigned)p3y * w + p3x;
    int dn = 0;
    bool right_edge = false;
    uint32_t run = 0;
    if (p3x > 0 && p3y > 0 && p3x + 1 < w && p3y + 1 < h) {
      const unsigned top = get3(i3 - w - 1), mid = get3(i3 - 1), bot = get3(i3 + w - 1);
      const unsigned m = (mid & 1u) | (top & 1u) << 1 | (top & 2u) << 1 | (top & 4u) << 1 | (mid & 4u) << 2 | (bot & 4u) << 3 | (bot & 2u) << 5 | (bot & 1u) << 7;
      const unsigned r = (((m | m << 8) >> base) & 0xffu) | 1u;
      const int j = 31 - __builtin_clz(r);
      dn = (base + j) & 7;
      right_edge = ((4 - base) & 7) > j;
      run = ((mid & 1u) ? 0u : kPtRunStart) | ((mid & 4u) ? 0u : kPtRunEnd);
    } else {
      for (int k = 1; k <= 8; ++k) {
        const int d = (base - k) & 7;
        if (nz(p3x + ddx(d), p3y + ddy(d))) {
          dn = d;
          break;
        }
        if (d == 4) right_edge = true;
      }
      run = (nz(p3x - 1, p3y) ? 0u : kPtRunStart) | (nz(p3x + 1, p3y) ? 0u : kPtRunEnd);
    }
    if (out) out[n] = ((uint32_t)p3y << 12) | (uint32_t)p3x | run | ((p3x + 1 == w || right_edge) ? kPtNeg : 0u);
    ++n;
    const int p4x = p3x + ddx(dn), p4y = p3y + ddy(dn);
    if (p4x == x && p4y == y && p3x == p1x && p3y == p1y) break;
    p3x = p4x;
    p3y = p4y;
    base = (dn + 4) & 7;
  }
  return n;
}

// LDS of the parallel form: ONE plane of a bit per pixel (+ a word), used twice - the bit image while the plausible starts are walked
// (phase A: random access, LDS latency), then, zeroed, the label plane of the raster scan (phase B), whose foreground rows stream in from
// global memory through a small band buffer behind the plane.  One label bit per pixel is enough because the scan only ever asks
// "has a border passed" at the FIRST pixel of a run and "was it left negative" at the LAST pixel of a run:
//     pixel is first of its run:            its bit = a border has passed
//     last of its run, not first:           its bit = left negative
//     first AND last (a one-pixel run):     the bit of the background pixel to its right = left negative
// (that pixel is in the row - the scan never asks about the last column - and nobody else's bit).  A 640 x 640 map needs 55 KB instead of the
// three planes' 150 KB, the reference's own 800 x 800 frames (/root/reference/src/text_detection/mod.rs:20-21) fit with 85 KB, and the CU keeps
// room for the next forward's workgroups beside the tracer.
constexpr int kBandRows = 64;
constexpr int kParWords = 35840;   // largest LDS image of the parallel form (140 KB + scan scratch + the start cache): 1024 x 1024 fits
constexpr int kKeyCache = 1024;   // plausible starts whose (key, length, offset) the scan reads from LDS instead of global memory
template <int LW>
__global__ __launch_bounds__(1024) void contour_parallel_kernel(const uint32_t* __restrict__ bits_all, int wpi, int h, int w, uint32_t* __restrict__ pts_all, int cap,
                                                              int* __restrict__ starts_all, int maxc, int* __restrict__ hdr_all, int* __restrict__ spec_all,
                                                              uint32_t* __restrict__ pool_all, int pool_cap) {
  __shared__ uint32_t lds[LW];
  __shared__ int sc[1024];
  // the scan of phase B looks a start up in the list by a serial walk (one wave, dependent loads): from global memory that walk - a
  // microsecond per entry, a few hundred entries per page - was most of the kernel; the first kKeyCache entries come from LDS
  __shared__ int ckey[kKeyCache], clen[kKeyCache], coff[kKeyCache];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned npx = (unsigned)h * (unsigned)w, nw = (npx + 31) / 32;
  uint32_t* bits = lds;                        // phase A: nw + 1 words (get3 reads the word behind as well)
  volatile uint32_t* lab = lds;                // phase B: the label plane, same words
  volatile uint32_t* band = lds + nw + 1;      // phase B: kBandRows rows of the bit image
  const uint32_t* g = bits_all + (size_t)img * wpi;
  for (unsigned i = tid; i < nw; i += 1024) bits[i] = g[i];
  if (tid == 0) bits[nw] = 0;
  __syncthreads();
  int* keys = spec_all + (size_t)img * 3 * kMaxStarts;
  int* rlen = keys + kMaxStarts;
  int* roff = rlen + kMaxStarts;
  uint32_t* pool = pool_all + (size_t)img * pool_cap;
  int status = 0;
  // block-wide exclusive prefix sum of one int per thread (Hillis-Steele in sc); returns the total through `total`
  auto block_scan = [&](int v, int& total) -> int {
    sc[tid] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int t = tid >= d ? sc[tid - d] : 0;
      __syncthreads();
      sc[tid] += t;
      __syncthreads();
    }
    const int incl = sc[tid];
    total = sc[1023];
    __syncthreads();
    return incl - v;
  };
  // ---- A1: plausible starts, row by row (h <= 1024: one row per thread)
  const int mine = tid < h ? row_starts(bits, h, w, tid, nullptr) : 0;
  int K = 0;
  const int kbase = block_scan(mine, K);
  if (K > kMaxStarts) status = 1;
  if (!status && tid < h && mine) row_starts(bits, h, w, tid, keys + kbase);
  __syncthreads();
  // ---- A2: lengths
  const int guard = (int)min(8u * npx + 64u, (unsigned)pool_cap);
  int bad = 0;
  if (!status)
    for (int k = tid; k < K; k += 1024) {
      const int key = keys[k], px = key >> 1;
      const int len = walk_border(bits, h, w, px % w, px / w, (key & 1) ? 4 : 0, nullptr, guard);
      rlen[k] = len;
      bad |= len < 0;
    }
  bad = __syncthreads_or(bad);
  if (bad) status = 1;
  // ---- A3: offsets (eight consecutive entries per thread)
  int T = 0;
  {
    int local[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * tid + j;
      local[j] = (!status && k < K) ? rlen[k] : 0;
      sum += local[j];
    }
    int off = block_scan(sum, T);
    if (!status && T <= pool_cap) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * tid + j;
        if (k < K) roff[k] = off;
        off += local[j];
      }
    }
  }
  if (T > pool_cap) status = 1;
  __syncthreads();
  // ---- A4: points
  if (!status)
    for (int k = tid; k < K; k += 1024) {
      const int key = keys[k], px = key >> 1;
      walk_border(bits, h, w, px % w, px / w, (key & 1) ? 4 : 0, pool + roff[k], guard);
    }
  __syncthreads();
  // the plane becomes the label plane
  for (unsigned i = tid; i <= nw; i += 1024) lds[i] = 0;
  if (!status && tid < K && tid < kKeyCache) {
    ckey[tid] = keys[tid];
    clen[tid] = rlen[tid];
    coff[tid] = roff[tid];
  }
  __syncthreads();
  if (wave != 0) return;
  auto key_at = [&](int i) -> int { return i < kKeyCache ? ckey[i] : keys[i]; };
  // ---- B: the raster scan with the label tests; borders come from the list
  uint32_t* pts = pts_all + (size_t)img * cap;
  int* starts = starts_all + (size_t)img * (maxc + 1);
  int ncont = 0, npts = 0, ka = 0;
  const int wr = w >> 5;
  for (int y = 0; y < h && !status; ++y) {
    const int yb = y & ~(kBandRows - 1);
    if (y == yb) {   // the next band of foreground rows: coalesced, every load in flight at once
      const int words = min(kBandRows, h - yb) * wr;
      const uint32_t* src = g + (size_t)yb * wr;
      for (int i = lane; i < words; i += 64) band[i] = src[i];
      __builtin_amdgcn_wave_barrier();   // (LDS serves a wave's operations in order: the reads below see these writes)
    }
    // A pixel can start a border only at the first pixel of a run (outer border: label still clear, W neighbour zero) or at its last
    // (hole border: not negative, E neighbour zero).  Both tests are word-wide bit arithmetic on the row's foreground and label
    // words, one word per lane: the scan only stops at pixels that DO start a border (a few dozen per image) instead of stepping through
    // every run boundary (a few thousand, five dependent LDS reads each: that walk was most of the kernel).  A border traced in this row
    // changes the labels of pixels further right, so the masks are rebuilt after every trace, for x beyond it.
    uint32_t rsf = 0, rs = 0, re = 0;
    const size_t rw = (size_t)y * wr + lane;
    if (lane < wr) {
      const volatile uint32_t* row = band + (size_t)(y - yb) * wr;
      const uint32_t cur = row[lane];
      const uint32_t lbit = lane > 0 ? row[lane - 1] >> 31 : 0u;
      const uint32_t rbit = lane + 1 < wr ? row[lane + 1] & 1u : 0u;
      rsf = cur & ~((cur << 1) | lbit);           // first pixels of runs
      re = cur & ~((cur >> 1) | (rbit << 31));    // last pixels of runs
      rs = rsf;
      if (lane == 0) rs &= ~1u;                   // x > 0
      if (lane == wr - 1) re &= ~(1u << 31);      // x + 1 < w
    }
    if (!__ballot((rs | re) != 0)) continue;
    int xmin = 0;
    while (!status) {
      uint32_t t0 = 0, t1 = 0;
      if (lane < wr) {
        const uint32_t lw = lab[rw];
        const uint32_t ln = lane + 1 < wr ? lab[rw + 1] & 1u : 0u;
        t0 = rs & ~lw;                            // value == 1 and the W neighbour is zero: outer border start
        // left negative: the pixel's own bit, or - for a one-pixel run, whose own bit says "passed" - the bit to its right
        const uint32_t ng = (lw & ~rsf) | (((lw >> 1) | (ln << 31)) & rsf);
        t1 = re & ~ng & ~t0;                      // else value > 0 and the E neighbour is zero: hole border start
        const int lo = xmin - 32 * lane;          // only pixels at or beyond xmin
        const uint32_t keep = lo <= 0 ? ~0u : lo >= 32 ? 0u : ~0u << lo;
        t0 &= keep;
        t1 &= keep;
      }
      const unsigned long long words = __ballot((t0 | t1) != 0);
      if (!words) break;
      const int wi = uni(__builtin_ctzll(words));
      const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, wi), c1 = (uint32_t)__builtin_amdgcn_readlane((int)t1, wi);
      const int bit = uni(__builtin_ctz(c0 | c1));
      const int x = wi * 32 + bit;
      const int type = (c0 >> bit) & 1u ? 0 : 1;
      xmin = x + 1;
      const unsigned i = (unsigned)y * w + x;
      const int key = 2 * (int)i + type;
      while (ka < K && uni(key_at(ka)) < key) ++ka;
      if (ka >= K || uni(key_at(ka)) != key) {   // a start the list does not hold: this image goes to the host tracer
        status = 3;
        break;
      }
      const int len = uni(ka < kKeyCache ? clen[ka] : rlen[ka]), off = uni(ka < kKeyCache ? coff[ka] : roff[ka]);
      if (ncont >= maxc || npts + len > cap) {
        status = 1;
        break;
      }
      if (lane == 0) starts[ncont] = npts;
      ++ncont;
      for (int q = lane; q < len; q += 64) {
        const uint32_t p = pool[off + q];
        const unsigned px = p & 0xfffu, py = (p >> 12) & 0xffffu;
        const unsigned pi = py * (unsigned)w + px;
        if (p & kPtRunStart) atomicOr(const_cast<uint32_t*>(lab) + (pi >> 5), 1u << (pi & 31));
        if ((p & kPtNeg) && (p & kPtRunEnd) && px + 1 < (unsigned)w) {
          const unsigned j = (p & kPtRunStart) ? pi + 1 : pi;
          atomicOr(const_cast<uint32_t*>(lab) + (j >> 5), 1u << (j & 31));
        }
        pts[npts + q] = (py << 16) | px;
      }
      npts += len;
      __builtin_amdgcn_wave_barrier();   // the label ORs of every lane are issued before the masks are read again (LDS serves a wave in order)
    }
  }
  if (lane == 0) {
    if (ncont <= maxc) starts[ncont] = npts;
    int* hdr = hdr_all + 4 * img;
    hdr[0] = ncont;
    hdr[1] = npts;
    hdr[2] = status;
    hdr[3] = K;
  }
}

// the good images' points and contour lengths, densely in image order (offsets recomputed from the headers by every block)
__global__ __launch_bounds__(256) void contour_compact_kernel(const int* __restrict__ hdr_all, const uint32_t* __restrict__ pts_all, int cap,
                                                            const int* __restrict__ starts_all, int maxc, uint32_t* __restrict__ pts_out, int* __restrict__ lens_out) {
  const int img = blockIdx.x;
  const int* hdr = hdr_all + 4 * img;
  if (hdr[2] != 0) return;
  size_t po = 0, co = 0;
  for (int j = 0; j < img; ++j)
    if (hdr_all[4 * j + 2] == 0) {
      co += hdr_all[4 * j];
      po += hdr_all[4 * j + 1];
    }
  const uint32_t* src = pts_all + (size_t)img * cap;
  const int* st = starts_all + (size_t)img * (maxc + 1);
  for (int i = threadIdx.x; i < hdr[1]; i += 256) pts_out[po + i] = src[i];
  for (int k = threadIdx.x; k < hdr[0]; k += 256) lens_out[co + k] = st[k + 1] - st[k];
}

}  // namespace

// LDS words of the parallel form for an h x w map: the plane, its extra word, the band buffer
static size_t parallel_words(int h, int w) { return ((size_t)h * w + 31) / 32 + 1 + (size_t)kBandRows * (w >> 5); }
static bool shape_ok(int h, int w) { return h > 0 && w > 0 && !(w & 31) && w <= 2048 && h <= 32767; }
// the one-wave form keeps the bit image and two label planes in LDS
static bool sequential_fits(int h, int w) { return shape_ok(h, w) && 3 * (((size_t)h * w + 31) / 32) + 1 <= (size_t)kLdsWords; }
static bool parallel_fits(int h, int w) { return shape_ok(h, w) && h <= 1024 && parallel_words(h, w) <= (size_t)kParWords; }

bool contour_trace_fits(int h, int w) { return parallel_fits(h, w) || sequential_fits(h, w); }

size_t contour_spec_bytes(int n) { return (size_t)n * (3 * (size_t)kMaxStarts * 4 + (size_t)kContourPool * 4); }

void launch_contour_trace(const uint32_t* bits, size_t words_per_image, int n, int h, int w, uint32_t* pts, int cap, int* starts, int maxc, int* hdr,
                          uint32_t* pts_packed, int* lens_packed, void* spec, int sequential, hipStream_t s) {
  if (n <= 0) return;
  if (!contour_trace_fits(h, w)) fail(OCR_ERR_INTERNAL, "contour_trace: a %dx%d map does not fit a CU's LDS", h, w);
  if (cap <= 0 || maxc <= 0 || words_per_image > 0x7fffffffu) fail(OCR_ERR_INTERNAL, "contour_trace: bad capacities");
  if (words_per_image < ((size_t)h * w + 31) / 32) fail(OCR_ERR_INTERNAL, "contour_trace: %zu words per image for a %dx%d map", words_per_image, h, w);
  // the parallel form unless the one-wave form was asked for (and the map fits its three planes), or only the one-wave form takes the shape
  const bool par = spec && parallel_fits(h, w) && !(sequential && sequential_fits(h, w));
  if (!par) {
    if (!sequential_fits(h, w)) fail(OCR_ERR_INTERNAL, "contour_trace: a %dx%d map needs the parallel form's scratch", h, w);
    hipLaunchKernelGGL(contour_trace_kernel, dim3((unsigned)n), dim3(64), 0, s, bits, (int)words_per_image, h, w, pts, cap, starts, maxc, hdr);
  } else {
    int* spec_i = static_cast<int*>(spec);
    uint32_t* pool = reinterpret_cast<uint32_t*>(spec_i + (size_t)n * 3 * kMaxStarts);
    // the smallest LDS footprint that holds the map: 56 KB up to 640 x 640, 88 KB up to 800 x 800, else the CU's
    const size_t need = parallel_words(h, w);
    if (need <= 14336)
      hipLaunchKernelGGL(contour_parallel_kernel<14336>, dim3((unsigned)n), dim3(1024), 0, s, bits, (int)words_per_image, h, w, pts, cap, starts, maxc, hdr, spec_i, pool, kContourPool);
    else if (need <= 22528)
      hipLaunchKernelGGL(contour_parallel_kernel<22528>, dim3((unsigned)n), dim3(1024), 0, s, bits, (int)words_per_image, h, w, pts, cap, starts, maxc, hdr, spec_i, pool, kContourPool);
    else
      hipLaunchKernelGGL(contour_parallel_kernel<kParWords>, dim3((unsigned)n), dim3(1024), 0, s, bits, (int)words_per_image, h, w, pts, cap, starts, maxc, hdr, spec_i, pool, kContourPool);
  }
  OCR_HIP(hipGetLastError());
  hipLaunchKernelGGL(contour_compact_kernel, dim3((unsigned)n), dim3(256), 0, s, hdr, pts, cap, starts, maxc, pts_packed, lens_packed);
  OCR_HIP(hipGetLastError());
}

}  // namespace ocr
