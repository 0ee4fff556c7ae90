// Split-bf16 implicit-GEMM convolution, wide form: 256 x 128 tiles, eight waves, one persistent workgroup per CU.
//
// Same arithmetic as conv_igemm<.., X3> (conv_igemm.hip) - every f32 operand the exact sum of three bf16 terms, six of the nine
// partial products per pair on v_mfma_f32_32x32x16_bf16, f32 accumulate, K order (channel chunk | tap), products small terms first -
// so the results are BIT-IDENTICAL to that kernel's; what changes is how many operand bytes an MFMA costs.  The 128 x 128 form (four
// waves stacked along M, two workgroups per CU) moves 16 KB of activations + 24 KB of pre-split weights into LDS per K-step for 48 MFMAs
// per wave and has every wave read all 24 KB of weight planes back: 26 B per clock and CU of L2 -> LDS stream at the full MFMA rate
// against the 31-35 a CU gathers, LDS 77 % booked - its launches sit at MFMA-busy 0.50-0.57.  Here
//   * a workgroup owns 256 rows x 128 columns: one 24 KB weight stage serves twice the rows      -> 28 KB of DMA per 128 x 128 of output;
//   * the eight waves form a 4 x 2 grid, 64 x 64 per wave: a wave reads 8 KB of A and 12 KB of B  -> 108 KB of LDS traffic per 128 x 128
//     (was 152); the price is that the two waves of a row split the same activations (VALU under the MFMAs);
//   * the workgroup is persistent and its K-steps form ONE flat sequence across its tiles: the operand DMA runs two steps ahead of the
//     multiplier and crosses tile boundaries, so a tile's epilogue (straight from the accumulators, no LDS) and the next tile's first
//     loads overlap - with one workgroup per CU nothing else would hide them;
//   * work is dealt in STRIPS of 64 rows (one row of waves): workgroup r of a column tile takes strips [r S / R, (r + 1) S / R) as tiles of
//     up to four strips; waves whose strip is not part of a tile skip its MFMAs.  The 128 x 128 form's 1 600 / 800 / 400 tiles on 512
//     slots (3.13 / 1.56 / 0.78 rounds) become 12.5 / 6.25 / 3.1 strips per CU.
//   * the workgroups of one XCD that share a rank walk the same rows for the different column tiles at the same time (A from that L2).
// Replaces for the f32 precision: the stride-2 3x3 convs, the 1x1 laterals / downsamples with Cout a multiple of 128 and the batched
// Winograd GEMMs of layer3 / layer4  (/root/reference/src/text_detection/model.rs:30-55,75-78,84-98).
#include <type_traits>

#include "common.hpp"

namespace ocr {
namespace x3w {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

[[maybe_unused]] constexpr unsigned OOB = 0x80000000u;   // voffset beyond any tensor (< 2^31 bytes): the DMA writes zeros
constexpr int ROWB = 128;               // bytes of K per LDS row of A: 32 f32
constexpr int BM = 256, BN = 128;
constexpr int A_BYTES = BM * ROWB;      // 32 KB
constexpr int B_BYTES = 3 * BN * 64;    // 24 KB: three bf16 planes of 128 rows x 32 k
[[maybe_unused]] constexpr int STAGE = A_BYTES + B_BYTES;
[[maybe_unused]] constexpr int NP = 7;                   // DMA instructions per wave and K-step: 4 of A, 3 of B

struct Args {
  const float* src;
  const void* wgt;
  const float* scale;
  const float* bias;
  const float* residual;
  float* out;
  unsigned src_bytes, wgt_bytes;
  int N, Hin, Win, Cin, Ho, Wo, Cout, M, pad, relu;
  int nblk_n;   // Cout / 128
  int batch;    // batched GEMM: problems; 1 otherwise
  int sp;       // strips of 64 rows per problem: ceil(M / 64)
  int ranks;    // workgroups per column tile (grid / nblk_n)
  unsigned mg_howo, sh_howo, mg_wo, sh_wo, mg_sp, sh_sp;
};

__device__ __forceinline__ int fast_div(int x, unsigned magic, unsigned shift) {
  if (shift == 0xFFFFFFFFu) return x;  // d == 1 (wave-uniform)
  const unsigned t = __umulhi((unsigned)x, magic);
  return (int)((t + (((unsigned)x - t) >> 1)) >> shift);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// one 16-byte-per-lane LDS-DMA load; see conv_igemm.hip::dma16 for why this is inline asm
template <typename R>
__device__ __forceinline__ void dma16(R rsrc, unsigned lds_addr, unsigned voff, int soff) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("" : "+s"(soff));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// compile-time ablations (make EXTRA=-DX3W_ABL=<bits>, tools/build_abl_x3w.sh): what is left of a launch's time without one ingredient -
// 1 no A DMA, 2 no B DMA, 4 no split, 8 no MFMA, 32 no LDS fragment reads, 64 no mid-step barrier, 128 no epilogue, 512 no wait for the DMA.
// The results of an ablated kernel are garbage; only its time means something.
#if defined(X3W_ABL)
#define X3W_DBG(bit) ((X3W_ABL & (bit)) != 0)
#else
#define X3W_DBG(bit) false
#endif

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}

template <int KS, int STRIDE>
__global__ __launch_bounds__(512) void conv_x3_wide(Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NTAP = KS * KS;
  constexpr int MT = 2, NT = 2;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // strip of the tile / 64-column half (waves 2 s, 2 s + 1 sit on two different SIMDs)

  // ---- which column tile, which strips.  Workgroup ids are dealt round-robin over the XCDs: id & 7 is the XCD, id >> 3 the slot in it.
  // Slots c, c + nblk_n, ... of an XCD serve column tile c; the workgroups of an XCD with the same slot / nblk_n share their rank's rows.
  const int id = blockIdx.x, per_xcd = gridDim.x >> 3;
  const int slot = id >> 3;
  const int ctile = slot % p.nblk_n;
  const int rank = (id & 7) * (per_xcd / p.nblk_n) + slot / p.nblk_n;
  const int spc = p.batch * p.sp;   // strips of one column tile over all problems
  const int s_begin = (int)((long long)rank * spc / p.ranks), s_end = (int)((long long)(rank + 1) * spc / p.ranks);
  if (s_begin >= s_end) return;     // (the whole workgroup: nothing has synchronised yet)
  const int n0 = ctile * BN;

  auto uniform_ptr = [](const void* q) {
    const unsigned long long u = (unsigned long long)q;
    return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
  };
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.src), 0, uni((int)p.src_bytes), 0x00020000);
  const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.wgt), 0, uni((int)p.wgt_bytes), 0x00020000);

  // a tile = strips [cur, cur + cnt) of one problem: problem, first row, strips
  struct Tile { int bz, m0, cnt; };
  auto tile_at = [&](int cur) -> Tile {
    const int bz = fast_div(cur, p.mg_sp, p.sh_sp);
    const int s0 = cur - bz * p.sp;
    return Tile{bz, s0 * 64, min(min(4, s_end - cur), p.sp - s0)};
  };

  // ---- DMA side.  Lane -> (row r of a 64-row group, 16-byte slot q holding global chunk q ^ f(r)); rows r + 64 i, i = 0..3.
  // Per row the byte offset of its window's first tap (modulo 2^32: it may lie before the tensor) and one validity bit per tap; a
  // step's offset is base + a scalar tap distance, or the out-of-range marker (zero padding, rows beyond M, strips not in the tile).
  const int r = tid >> 3, q = tid & 7;
  const int gq = q ^ ((r >> 1) & 7);
  const int HoWo = p.Ho * p.Wo;
  const int csteps = p.Cin >> 5;
  const int total = csteps * NTAP;      // K-steps per tile
  const int nK = (NTAP * p.Cin) >> 5;   // K-steps per weight row
  unsigned abase[4], amask[4], bvoff = 0;
  int d_cur = s_begin, d_cnt = 0;       // the tile the DMA iterator is in
  bool d_done = false;
  auto prep_tile = [&](int cur) {
    const Tile t = tile_at(cur);
    d_cnt = t.cnt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = t.m0 + r + 64 * i;
      abase[i] = 0;
      amask[i] = 0;
      if (i < t.cnt && m < p.M) {
        const int n = fast_div(m, p.mg_howo, p.sh_howo);
        const int rem = m - n * HoWo;
        const int oh = fast_div(rem, p.mg_wo, p.sh_wo);
        const int ow = rem - oh * p.Wo;
        const int ih0 = oh * STRIDE - p.pad, iw0 = ow * STRIDE - p.pad;
        abase[i] = (unsigned)(((n * p.Hin + ih0) * p.Win + iw0) * p.Cin * 4 + gq * 16 + t.bz * p.M * p.Cin * 4);
#pragma unroll
        for (int tp = 0; tp < NTAP; ++tp)
          if ((unsigned)(ih0 + tp / KS) < (unsigned)p.Hin && (unsigned)(iw0 + tp % KS) < (unsigned)p.Win) amask[i] |= 1u << tp;
      }
    }
    // weights: blocks [row / 16][K-step][plane] of 1 KB (split3_weights_tiled); this wave fills rows 16 wave .. + 15 of every plane
    bvoff = (unsigned)((((t.bz * p.Cout + n0) >> 4) + wave) * nK * 3072 + lane * 16);
  };
  int it_c = 0, it_t = 0;
  unsigned cur_av[4], cur_bv = 0;   // (the step's own copies: advancing into the next tile rewrites the tables)
  int cur_soff_a = 0, cur_soff_b = 0;
  auto select_next = [&] {   // the step the iterator points at; then advance (into the next tile when this one is through)
    const int kh = it_t / KS, kw = it_t - kh * KS;
    const unsigned dist = (unsigned)((kh * p.Win + kw) * p.Cin * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) cur_av[i] = (amask[i] >> it_t) & 1u ? abase[i] + dist : OOB;
    cur_bv = bvoff;
    cur_soff_a = it_c * ROWB;
    cur_soff_b = (it_t * csteps + it_c) * 3072;
    if (++it_t == NTAP) {
      it_t = 0;
      if (++it_c == csteps) {
        it_c = 0;
        d_cur += d_cnt;
        if (d_cur < s_end) prep_tile(d_cur);
        else d_done = true;
      }
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)lds;
  auto dma_piece = [&](int stage, auto piece_c) {
    constexpr int P = decltype(piece_c)::value;
    if constexpr (P < 4) {
      if (!X3W_DBG(1)) dma16(a_rsrc, uni((int)(lds0 + (unsigned)(stage * STAGE + (8 * wave + 64 * P) * ROWB))), cur_av[P], cur_soff_a);
    } else {
      constexpr int pl = P - 4;
      if (!X3W_DBG(2)) dma16(b_rsrc, uni((int)(lds0 + (unsigned)(stage * STAGE + A_BYTES + pl * BN * 64 + 16 * wave * 64))), cur_bv, cur_soff_b + pl * 1024);
    }
  };
  auto dma_all = [&](int stage) { static_for<0, NP>([&](auto i) { dma_piece(stage, i); }); };

  // ---- multiplier side
  f32x16 acc[MT][NT];
  auto zero_acc = [&] {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  };
  zero_acc();
  // operand fetch: lane l supplies row (l & 31) of a 32-row tile and the 16-byte chunk 2 g + (l >> 5) of K-group g (slot chunk ^ f(row))
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;
  int xoff[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xoff[g] = ((2 * g + (lane >> 5)) ^ fsw) * 16;
  const int a_row = (wm * 64 + frow) * ROWB;
  const int bx_row = A_BYTES + (wn * 64 + frow) * 64;
  const int bx_f = (frow >> 2) & 3;
  struct Frag {
    bf16x8 ah[MT], am[MT], al[MT];
    bf16x8 bh[NT], bm[NT], bl[NT];
  };
  struct Raw { f32x4 a0[MT], a1[MT]; };
  auto read_frag = [&](int stage, int kk, Raw& rw, Frag& f) {
    const unsigned char* st = lds + stage * STAGE;
    if (X3W_DBG(32)) {   // no LDS reads: operands from whatever the registers hold
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(rw.a0[i]), "+v"(rw.a1[i]));
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(f.bh[j]), "+v"(f.bm[j]), "+v"(f.bl[j]));
      return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      rw.a0[i] = *reinterpret_cast<const f32x4*>(st + a_row + i * 32 * ROWB + xoff[2 * kk]);
      rw.a1[i] = *reinterpret_cast<const f32x4*>(st + a_row + i * 32 * ROWB + xoff[2 * kk + 1]);
    }
    const int boff = bx_row + (((2 * kk + (lane >> 5)) ^ bx_f) * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      f.bh[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64);
      f.bm[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64 + BN * 64);
      f.bl[j] = *reinterpret_cast<const bf16x8*>(st + boff + j * 32 * 64 + 2 * BN * 64);
    }
  };
  // elements [E0, E1) of the 16 this lane holds (row tile e >> 3): x = hi + mid + lo, round to nearest even at every level
  auto split_part = [&](const Raw& rw, Frag& f, auto e0c, auto e1c) {
    constexpr int E0 = decltype(e0c)::value, E1 = decltype(e1c)::value;
    if (X3W_DBG(4)) {
      if constexpr (E0 == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          f.ah[i] = __builtin_bit_cast(bf16x8, rw.a0[i]);
          f.am[i] = __builtin_bit_cast(bf16x8, rw.a1[i]);
          f.al[i] = __builtin_bit_cast(bf16x8, rw.a0[i] + rw.a1[i]);
        }
      }
      return;
    }
#pragma unroll
    for (int e = E0; e < E1; ++e) {
      const int i = e >> 3, k = e & 7;
      const float x = k < 4 ? rw.a0[i][k] : rw.a1[i][k - 4];
      const __bf16 h = (__bf16)x;
      const float r1 = x - (float)h;
      const __bf16 m = (__bf16)r1;
      const float r2 = r1 - (float)m;
      f.ah[i][k] = h;
      f.am[i][k] = m;
      f.al[i][k] = (__bf16)r2;
    }
  };
  // MFMA number idx of a group: product idx / 4 (small terms first) of accumulator idx % 4 - consecutive MFMAs go to different
  // accumulators; per accumulator the order of the six products is conv_igemm's
  auto mfma_at = [&](const Frag& f, auto idx_c) {
    constexpr int idx = decltype(idx_c)::value, t = idx % (MT * NT), i = t / NT, j = t % NT, pr = idx / (MT * NT);
    if (X3W_DBG(8)) {
      asm volatile("" ::"v"(f.al[i]), "v"(f.ah[i]), "v"(f.am[i]), "v"(f.bh[j]), "v"(f.bm[j]), "v"(f.bl[j]));
      return;
    }
    if constexpr (pr == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
    if constexpr (pr == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
    if constexpr (pr == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am[i], f.bm[j], acc[i][j], 0, 0, 0);
    if constexpr (pr == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.am[i], f.bh[j], acc[i][j], 0, 0, 0);
    if constexpr (pr == 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bm[j], acc[i][j], 0, 0, 0);
    if constexpr (pr == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
  };
  // One MFMA group = 24 MFMAs in four regions of six; between them: region 0 the LDS reads of the next group's fragments, regions 1-3
  // the split of the next A fragments (sixteen elements), and at the region ends this wave's DMA instructions of the step after next.
  // `active`: this wave's strip belongs to the tile being multiplied or to the one being prefetched (wave-uniform); an idle wave only
  // issues its DMA share.
  auto phase = [&](bool active, const Frag& cur, int rd_stage, int rd_kk, Raw& nraw, Frag& nxt, int dma_stage) {
    constexpr int Q = 6;
    if (active) {
      read_frag(rd_stage, rd_kk, nraw, nxt);
      static_for<0, Q>([&](auto i) { mfma_at(cur, i); });
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
    }
#if defined(X3W_DMA_EARLY)   // experiments: every piece behind the first region / behind the last one
    constexpr int P1 = NP, P2 = NP, P3 = NP;
#elif defined(X3W_DMA_LATE)
    constexpr int P1 = 0, P2 = 0, P3 = 0;
#else
    constexpr int P1 = 2, P2 = 4, P3 = 6;
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (dma_stage >= 0) static_for<0, P1>([&](auto i) { dma_piece(dma_stage, i); });
    if (active) {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(nraw.a0[i]), "+v"(nraw.a1[i]));
      static_for<Q, 2 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(nxt.ah[i]), "+v"(nxt.am[i]), "+v"(nxt.al[i]));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (dma_stage >= 0) static_for<P1, P2>([&](auto i) { dma_piece(dma_stage, i); });
    if (active) {
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(nraw.a0[i]), "+v"(nraw.a1[i]));
      static_for<2 * Q, 3 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, 6>{}, std::integral_constant<int, 12>{});
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(nxt.ah[i]), "+v"(nxt.am[i]), "+v"(nxt.al[i]));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (dma_stage >= 0) static_for<P2, P3>([&](auto i) { dma_piece(dma_stage, i); });
    if (active) {
      asm volatile("" : "+v"(nraw.a0[1]), "+v"(nraw.a1[1]));
      static_for<3 * Q, 4 * Q>([&](auto i) { mfma_at(cur, i); });
      split_part(nraw, nxt, std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
#pragma unroll
      for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(nxt.ah[i]), "+v"(nxt.am[i]), "+v"(nxt.al[i]));
      for (int i = 0; i < Q; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (dma_stage >= 0) static_for<P3, NP>([&](auto i) { dma_piece(dma_stage, i); });
  };

  // ---- epilogue of one tile, straight from the accumulators (C/D map of a 32x32 tile: col = lane & 31, row = (e & 3) + 8 (e >> 2) +
  // 4 (lane >> 5)); rows beyond M fall outside the descriptor's range and are dropped by the hardware
  auto epilogue = [&](const Tile& t) {
    if (X3W_DBG(128)) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if (wm < t.cnt) {
      const int colq = lane & 31, rowq = (lane >> 5) * 4;
      const unsigned row_b = (unsigned)p.Cout * 4u;
      const int r0 = t.m0 + wm * 64;
      const int rows = min(max(p.M - r0, 0), 64);
      const size_t corner = ((size_t)t.bz * p.M + r0) * p.Cout + (n0 + wn * 64);
      const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(p.out + corner), 0, uni((int)(rows * row_b)), 0x00020000);
      const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((p.residual ? p.residual : p.out) + corner), 0, uni((int)(rows * row_b)), 0x00020000);
      const unsigned voff = (unsigned)rowq * row_b + (unsigned)colq * 4u;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * 64 + j * 32 + colq;
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
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    zero_acc();
  };

  // ---- the flat sequence of K-steps over this workgroup's tiles
  prep_tile(d_cur);
  select_next();
  dma_all(0);
  if (!d_done) {
    select_next();
    dma_all(1);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  int m_cur = s_begin;
  Tile mt = tile_at(m_cur);
  Frag fx = {}, fy = {};
  Raw rw = {};
  if (wm < mt.cnt) {
    read_frag(0, 0, rw, fx);
    split_part(rw, fx, std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
  }
  int par = 0;
#if defined(X3W_PRIO)   // experiment: static priority for the second-dispatched half (the arbitration loser of every SIMD pair)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  for (;;) {
    const bool act = wm < mt.cnt;
    const int nxt_cur = m_cur + mt.cnt;
    const bool last_tile = nxt_cur >= s_end;
    Tile nt = mt;
    if (!last_tile) nt = tile_at(nxt_cur);
    const bool act_next = !last_tile && wm < nt.cnt;
    for (int k = 0; k < total; ++k) {
      phase(act, fx, par, 1, rw, fy, -1);
      // this wave's reads of stage par are complete and its share of the next step's DMA has landed; after the barrier so is everyone's
      if (X3W_DBG(512)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (!X3W_DBG(64)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool more = !d_done;   // the flat sequence has a step two ahead of this one
      if (more) select_next();
      // (k, 1) multiplied; (k + 1, 0) - of the next tile behind this tile's last step - read and split from the other stage
      phase(k + 1 < total ? act : (act || act_next), fy, par ^ 1, 0, rw, fx, more ? par : -1);
      par ^= 1;
    }
    epilogue(mt);
    if (last_tile) break;
    m_cur = nxt_cur;
    mt = nt;
  }
#endif
}

static void make_magic(unsigned d, unsigned* magic, unsigned* shift) {
  unsigned L = 0;
  while ((1ull << L) < d) ++L;
  *magic = L == 0 ? 0u : (unsigned)(((1ull << 32) * ((1ull << L) - d)) / d + 1);
  *shift = L == 0 ? 0xFFFFFFFFu : L - 1;
}

template <int KS, int STRIDE>
static void launch(const ConvDesc& d, int cus, hipStream_t s) {
  Args a{};
  a.src = static_cast<const float*>(d.src[0]);
  a.wgt = d.wgt;
  a.scale = d.scale;
  a.bias = d.bias;
  a.residual = static_cast<const float*>(d.residual);
  a.out = static_cast<float*>(d.out);
  a.src_bytes = (unsigned)d.src_bytes;
  a.wgt_bytes = (unsigned)d.wgt_bytes;
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
  a.batch = d.batch > 1 ? d.batch : 1;
  a.sp = (a.M + 63) / 64;
  // one workgroup per CU (112 KB of LDS): 8 XCDs x a multiple of nblk_n slots
  int per_xcd = cus / 8;
  per_xcd -= per_xcd % a.nblk_n;
  if (per_xcd <= 0) fail(OCR_ERR_INTERNAL, "%s: %d CUs cannot hold %d column tiles per XCD", d.name, cus, a.nblk_n);
  const int grid = 8 * per_xcd;
  a.ranks = grid / a.nblk_n;
  make_magic((unsigned)(a.Ho * a.Wo), &a.mg_howo, &a.sh_howo);
  make_magic((unsigned)a.Wo, &a.mg_wo, &a.sh_wo);
  make_magic((unsigned)a.sp, &a.mg_sp, &a.sh_sp);
  hipLaunchKernelGGL((conv_x3_wide<KS, STRIDE>), dim3(grid), dim3(512), 0, s, a);
  OCR_HIP(hipGetLastError());
}

}  // namespace x3w

// the launches the wide form exists for (everything else stays with conv_igemm's 128-wide tiles)
bool conv_x3_wide_applicable(const ConvDesc& d) {
  if (!d.x3 || d.in_bf16 || d.out_bf16 || d.src_mode != SRC_PLAIN || d.store_mode != STORE_NHWC || d.out2 || d.up_residual || !d.out) return false;
  if (d.Cout % 128 != 0 || d.Cout / 128 > 4 || d.Cin % 32 != 0) return false;
  if (!((d.ks == 3 && d.stride == 2) || (d.ks == 1 && (d.stride == 1 || d.stride == 2)))) return false;
  const int nb = d.batch > 1 ? d.batch : 1;
  if (nb > 1 && (d.residual || d.ks != 1 || d.stride != 1)) return false;
  const long long M = (long long)d.N * d.Ho * d.Wo;
  // strips x problems must leave every CU a few strips, and all index arithmetic in int
  if (M < 64 || (long long)nb * ((M + 63) / 64) * (d.Cout / 128) < 256 || (long long)nb * M * d.Cout >= (1ll << 31) / 4 * 4 || (long long)nb * M * d.Cin * 4 >= (1ll << 31)) return false;
  return true;
}

void launch_conv_x3_wide(const ConvDesc& d, int cus, hipStream_t s) {
  if (!conv_x3_wide_applicable(d)) fail(OCR_ERR_INTERNAL, "%s: not a launch of the wide split-bf16 form", d.name);
  const int nb = d.batch > 1 ? d.batch : 1;
  // the same operand checks as conv_igemm.hip::check for these forms (an out-of-bounds access on the GPU can take the node down)
  if (d.pad != (d.ks - 1) / 2 || d.Ho != (d.Hin + 2 * d.pad - d.ks) / d.stride + 1 || d.Wo != (d.Win + 2 * d.pad - d.ks) / d.stride + 1)
    fail(OCR_ERR_INVALID, "%s: output grid %dx%d does not follow from input %dx%d", d.name, d.Ho, d.Wo, d.Hin, d.Win);
  const long long in_bytes = (long long)nb * d.N * d.Hin * d.Win * d.Cin * 4;
  if (in_bytes >= (1ll << 31) || (long long)d.src_bytes >= (1ll << 31) || (long long)d.src_bytes < in_bytes)
    fail(OCR_ERR_INVALID, "%s: input of %lld bytes (addressable %zu) must be < 2^31 bytes", d.name, in_bytes, d.src_bytes);
  if ((long long)d.wgt_bytes != (long long)nb * d.Cout * d.ks * d.ks * d.Cin * 6 || (long long)d.wgt_bytes >= (1ll << 31)) fail(OCR_ERR_INVALID, "%s: weight bytes", d.name);
  if (!d.src[0] || !d.wgt) fail(OCR_ERR_INVALID, "%s: null operand", d.name);
  if (d.ks == 3) return x3w::launch<3, 2>(d, cus, s);
  if (d.stride == 2) return x3w::launch<1, 2>(d, cus, s);
  return x3w::launch<1, 1>(d, cus, s);
}

}  // namespace ocr
