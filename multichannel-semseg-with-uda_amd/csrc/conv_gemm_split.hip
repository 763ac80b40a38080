// Implicit-GEMM convolution (forward and data-gradient) with fp32 operands carried through the 16-bit matrix pipe as a
// sum of exact piece products -- the split policies of split.h: SplitBf16x6 (three bf16 pieces, six cross terms) and
// SplitF16x3 (two fp16 pieces and a per-tensor power-of-two scale, three cross terms; the default).
//
// Structure is that of conv_gemm.hip (buffer-load gathers, channel-chunk-outer / tap-inner K order, XCD-aware
// tiles, double-buffered LDS, fused BatchNorm statistics); what differs is the operand path:
//   * weights are pre-split on the host side of the ABI into the exact LDS image
//     [k-step][piece NP][k-half 2][Mp][8 x 16 bit], so a K-step's slab is 2*NP contiguous runs copied 16 B per lane;
//   * one thread gathers 8 consecutive channels of one pixel (8 coalesced dword loads, lanes = pixels), splits
//     them and writes NP 16-B fragments [piece][k-half][pixel][8] -- exactly what one lane of the MFMA consumes, so
//     fragment reads are ds_read_b128 over 512 contiguous bytes per half-wave; with a pre-split companion written by the
//     operand's producer (bn.hip) the fragments come straight from memory;
//   * SplitF16x3 multiplies the accumulators by scale(operand) * scale(weights) -- an exact power of two -- before the
//     epilogue (bias, BatchNorm statistics, affine).
#include <cstdlib>
#include <type_traits>

#include "conv_split_params.h"
#include "options.h"

namespace {


template <class P, int WM, int WN, int WAVES_M, int WAVES_N, bool DGRAD, bool PRESPLIT>
__global__ __launch_bounds__(256, 2) void conv_gemm_split_kernel(ConvSplitParams p) {
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  constexpr int NT = 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int B_ITEMS = BN * 2 / NT;             // (pixel, k-half) items per thread: 1 (BN=128) or 2 (BN=256)
  constexpr int NP = P::NP;                        // pieces per operand
  constexpr int NQ = 2 * NP;                       // (piece, k-half) planes per K-step
  typedef typename P::frag frag;
  constexpr int A_CHUNKS = NQ * BM;                // 16-byte chunks of the weight slab per K-step
  constexpr int A_ITERS = (A_CHUNKS + NT - 1) / NT;
  constexpr bool A_EXACT = (A_CHUNKS % NT) == 0;
  constexpr bool A_DMA = A_EXACT;                  // weight slab by LDS-DMA when it tiles the workgroup exactly
  // ... of which only the pieces the policy multiplies are moved (piece-major slab) -- when those still tile the workgroup
  // exactly (every wave must issue the same number of DMAs: the K loop's waits are counted)
  constexpr int A_ITERS_DMA = ((2 * P::NPU * BM) % NT == 0) ? (2 * P::NPU * BM) / NT : A_ITERS;
  constexpr int A_BYTES = NQ * BM * 16, B_BYTES = NQ * BN * 16;

  // B_DMA: the pre-split operand also goes global -> LDS by DMA (see dma_b); that K loop keeps NSTAGE LDS stages: with three,
  // the DMAs of step s+2 are in flight while step s multiplies -- a K-step of the three-term arithmetic is only ~400-800
  // cycles, less than a round trip beyond the XCD's L2
  // (the 128 x 256 tile: two k-halves per thread.  The 64 x 256 tile of the 64-channel layers stays on the register-staged two-stage
  // loop: on the all-DMA loop it is SLOWER -- forward 0.086 -> 0.094 ms, data gradient 0.105 -> 0.123 in the step; round 5)
  constexpr bool B_DMA = PRESPLIT && A_DMA && (B_ITEMS == 1 || (BM == 128 && BN == 256));
  constexpr int NSTAGE = (B_DMA && 3 * (A_BYTES + B_BYTES) <= 80 * 1024) ? 3 : 2;  // two workgroups per CU must still fit 160 KiB

  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;                       // [NSTAGE][piece][half][BM][16 B]
  unsigned char* Bs = smem + NSTAGE * A_BYTES;    // [NSTAGE][piece][half][BN][16 B]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  const int m_tiles = p.Mp / BM;
  const bool SUB = DGRAD && p.sub != 0;
  const int n_tiles = SUB ? p.cls_tile0[4] : p.tile_n1;  // (the tile window of this launch: tile_n0 .. tile_n1 - 1, never with SUB)
  const int tile_n0 = SUB ? 0 : p.tile_n0;
  const int per_xcd = (n_tiles - tile_n0 + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int tile_m = slot % m_tiles;
  const int tile_n = tile_n0 + xcd * per_xcd + slot / m_tiles;
  if (tile_n >= n_tiles) return;

  // ---- this thread's gather pixel (same pixel for both of its k-halves when BN = 256)
  const int bj = t % BN;
  const int bh0 = __builtin_amdgcn_readfirstlane(t / BN);  // k-half of item 0 (wave-uniform)
  const int HWd = p.Hd * p.Wd;
  const int HWs = p.Hs * p.Ws;
  // parity class of this tile (uniform): ry, rx, class grid Hc x Wc, first tile of the class
  int cls = 0;
  // (sub == 2: the four classes have the same tile count and are interleaved -- tiles 4k .. 4k+3 are the four parities of ONE region,
  // neighbours on one XCD: they gather the same dY window through that L2 and their stride-2 stores meet there as whole lines)
  const bool ILV = SUB && p.sub == 2;
  // (sub == 3: ROW classes -- a tile holds consecutive pixels of rows of one y-parity, BOTH x-parities: a wave's stores are as dense as
  // a stride-1 layer's instead of 4 bytes in every 8.  The K loop walks every tap of the row class; a lane whose x-parity a tap does not
  // reach gathers zeros for it (an out-of-range offset: no memory request) -- twice the matrix instructions of the four-class form for
  // the same sums in the same order, in kernels that use a sixteenth of the pipe)
  const bool ROW = SUB && p.sub == 3;
  if (SUB) cls = ILV ? (tile_n & 3) : (tile_n >= p.cls_tile0[1]) + (tile_n >= p.cls_tile0[2]) + (tile_n >= p.cls_tile0[3]);
  const int ry = ROW ? cls : cls >> 1, rx = cls & 1;  // (ROW: the x-parity is the lane's own)
  const int Hc = SUB ? (p.Hd - ry + 1) >> 1 : p.Hd;
  const int Wc = SUB && !ROW ? (p.Wd - rx + 1) >> 1 : p.Wd;
  const int HWc = Hc * Wc;
  const int Pc = SUB ? p.N * HWc : p.P;
  const int ltile = SUB ? (ILV ? tile_n >> 2 : tile_n - p.cls_tile0[cls]) : tile_n;
  const int pix = ltile * BN + bj;
  const bool pv = pix < Pc;
  int pn = 0, py = 0, px = 0, yc = 0, xc = 0;
  if (pv) {
    pn = pix / HWc;
    const int rem = pix - pn * HWc;
    yc = rem / Wc;
    xc = rem - yc * Wc;
    py = SUB ? 2 * yc + ry : yc;
    px = SUB && !ROW ? 2 * xc + rx : xc;
  }
  // taps of this class (all taps outside the class ordering)
  unsigned cls_taps = 0;
  if (SUB) {
    for (int q = 0; q < p.KH * p.KW; ++q) {
      const int qy = q / p.KW, qx = q - qy * p.KW;
      if ((((ry + p.pad - qy * p.dil) | (ROW ? 0 : rx + p.pad - qx * p.dil)) & 1) == 0) cls_taps |= 1u << q;
    }
  }
  constexpr unsigned OOB = 0x80000000u;
  // PRESPLIT: the gathered operand was split into 16-bit pieces by its producer (bn_apply_cb / bn_bwd_apply_cb); a pixel's
  // 8-channel group is one 16-B fragment, so offsets count 16-B units and a K-step needs NP x 16-B loads and no VALU.
  constexpr unsigned UNIT = PRESPLIT ? 16u : 4u;
  const __amdgpu_buffer_rsrc_t src_rs = PRESPLIT ? __builtin_amdgcn_make_buffer_rsrc((void*)p.src_cb, 0, p.cb_bytes, 0x00020000)
                                                 : __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  // one descriptor per piece of the companion (each below 2 GiB; the pieces of a batch slice are not adjacent)
  __amdgpu_buffer_rsrc_t cb_rs[NP];
#pragma unroll
  for (int pc = 0; pc < NP; ++pc)
    cb_rs[pc] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src_cb + pc * p.cb_piece_stride), 0, PRESPLIT ? p.cb_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.wp_bytes, 0x00020000);
  const int C8 = p.Cs >> 3;
  const float inv_src_scale = 1.f / operand_scale<P>(p.src_bound);  // in-loop split only
  const unsigned pix_base = PRESPLIT ? (unsigned)pn * (unsigned)C8 * (unsigned)HWs : (unsigned)pn * (unsigned)p.Cs * (unsigned)HWs;
  const bool ragged = p.Kp != p.Cs;
  const int taps = p.KH * p.KW;

  int l_tap = 0, l_c0 = 0, l_ky = 0, l_kx = 0, l_kstep = 0;
  unsigned l_voff = OOB;
  auto tap_geom = [&]() {
    bool ok;
    int off;
    if (!DGRAD) {
      const int sy = py * p.stride + l_ky * p.dil - p.pad;
      const int sx = px * p.stride + l_kx * p.dil - p.pad;
      ok = pv && sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws;
      off = sy * p.Ws + sx;
    } else {
      const int ty = py + p.pad - l_ky * p.dil;
      const int tx = px + p.pad - l_kx * p.dil;
      ok = pv && ty >= 0 && tx >= 0;
      int sy = ty, sx = tx;
      if (p.stride != 1) {
        sy = ty / p.stride;
        sx = tx / p.stride;
        ok = ok && (sy * p.stride == ty) && (sx * p.stride == tx);
      }
      ok = ok && sy < p.Hs && sx < p.Ws;
      off = sy * p.Ws + sx;
    }
    l_voff = ok ? (pix_base + (unsigned)off) * UNIT : OOB;
  };
  // The pixel of a thread never changes.  For unit-stride geometry the gather offset of tap (ky,kx) is the thread's
  // own base plus a tap offset that is the SAME for every thread (scalar ALU), and whether the tap falls into the zero
  // padding is one bit of a per-thread mask computed once -- the K loop spends three VALU ops per step on addressing
  // instead of ~25 that would compete with the MFMA issue slots.  Strided dgrad keeps the general form.
  const bool fast_taps = taps <= 32 && (!DGRAD || p.stride == 1 || SUB);
  unsigned valid_mask = 0;
  unsigned vbase = 0;
  if (fast_taps) {
    for (int q = 0; q < taps; ++q) {
      l_ky = q / p.KW;
      l_kx = q - l_ky * p.KW;
      tap_geom();
      valid_mask |= (l_voff != OOB ? 1u : 0u) << q;
    }
    l_ky = 0;
    l_kx = 0;
    vbase = SUB ? pix_base + (unsigned)(yc * p.Ws + (ROW ? 0 : xc))
                : (DGRAD ? pix_base + (unsigned)(py * p.Ws + px) : pix_base + (unsigned)(py * p.stride * p.Ws + px * p.stride));
  }
  auto fast_voff = [&]() {
    // wave-uniform; in a parity class the source pixel is (yc + (ry + pad - ky*dil)/2, ...), the numerators being even
    const int rel = SUB ? ((ry + p.pad - l_ky * p.dil) >> 1) * p.Ws + (ROW ? 0 : (rx + p.pad - l_kx * p.dil) >> 1)
                        : (DGRAD ? (p.pad - l_ky * p.dil) * p.Ws + (p.pad - l_kx * p.dil)
                                 : (l_ky * p.dil - p.pad) * p.Ws + (l_kx * p.dil - p.pad));
    // (ROW: the source column is the lane's own -- (px + pad - kx dil) / 2, even wherever the mask bit is set)
    const unsigned col = ROW ? (unsigned)((px + p.pad - l_kx * p.dil) >> 1) : 0u;
    l_voff = ((valid_mask >> l_tap) & 1u) ? (vbase + (unsigned)rel + col) * UNIT : OOB;
  };
  if (SUB && cls_taps != 0) {  // start at the class's first tap
    l_tap = __builtin_ctz(cls_taps);
    l_ky = l_tap / p.KW;
    l_kx = l_tap - l_ky * p.KW;
    l_kstep = l_tap;
  }
  if (fast_taps)
    fast_voff();
  else
    tap_geom();
  auto advance = [&]() {
    if (SUB) {  // next tap of the class, or its first tap in the next channel chunk
      const unsigned rest = cls_taps & ~((2u << l_tap) - 1u);
      if (rest) {
        l_tap = __builtin_ctz(rest);
      } else {
        l_tap = __builtin_ctz(cls_taps);
        l_c0 += 16;
      }
      l_ky = l_tap / p.KW;
      l_kx = l_tap - l_ky * p.KW;
      l_kstep = (l_c0 >> 4) * taps + l_tap;
    } else {
    ++l_kstep;
    ++l_tap;
    ++l_kx;
    if (l_kx == p.KW) {
      l_kx = 0;
      ++l_ky;
    }
    if (l_tap == taps) {
      l_tap = 0;
      l_kx = 0;
      l_ky = 0;
      l_c0 += 16;
    }
    }
    if (fast_taps)
      fast_voff();
    else
      tap_geom();
  };

  unsigned a_voff[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int id = t + i * NT;
    const int plane = id / BM;
    const int m = id - plane * BM;
    a_voff[i] = (A_EXACT || id < A_CHUNKS) ? ((unsigned)plane * (unsigned)p.Mp + (unsigned)m) * 16u : OOB;
  }

  // two register sets: gathers run TWO K-steps ahead of the MFMAs (a K-step of six K=16 MFMAs per tile is only ~800
  // cycles, less than a MALL/HBM round trip, so one step of prefetch distance leaves the first tap of every channel
  // chunk exposed)
  float breg[2][B_ITEMS][PRESPLIT ? 1 : 8];
  f32x4 bsplit[2][B_ITEMS][PRESPLIT ? NP : 1];
  f32x4 areg[2][A_ITERS];
  auto load_regs = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int h = (B_ITEMS == 1) ? bh0 : it;
      if constexpr (PRESPLIT) {
        const int grp = (l_c0 >> 3) + h;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          const int soff = grp < C8 ? grp * HWs * 16 : 0x7FFFFFFF;
          const auto q = __builtin_amdgcn_raw_buffer_load_b128(cb_rs[pc], l_voff, soff, 0);
          bsplit[SET][it][pc][0] = __uint_as_float(q[0]);
          bsplit[SET][it][pc][1] = __uint_as_float(q[1]);
          bsplit[SET][it][pc][2] = __uint_as_float(q[2]);
          bsplit[SET][it][pc][3] = __uint_as_float(q[3]);
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = l_c0 + 8 * h + e;
        const int soff = (!ragged || c < p.Cs) ? c * HWs * 4 : 0x7FFFFFFF;
        if constexpr (!PRESPLIT) breg[SET][it][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src_rs, l_voff, soff, 0));
      }
    }
    const int a_soff = (l_kstep * NQ * p.Mp + tile_m * BM) * 16;
    if (A_DMA) return;  // the weight slab goes global -> LDS directly (dma_weights)
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const auto q = __builtin_amdgcn_raw_buffer_load_b128(wp_rs, a_voff[i], a_soff, 0);
      areg[SET][i][0] = __uint_as_float(q[0]);
      areg[SET][i][1] = __uint_as_float(q[1]);
      areg[SET][i][2] = __uint_as_float(q[2]);
      areg[SET][i][3] = __uint_as_float(q[3]);
    }
  };
  // Pre-split weights are stored in exactly the LDS image order, so a K-step's slab is moved by LDS-DMA
  // (buffer_load_dwordx4 ... lds: wave-uniform LDS base + lane * 16 B): no VGPRs, no ds_write, no VALU.
  auto dma_weights = [&](int buf) {
    const int a_soff = (l_kstep * NQ * p.Mp + tile_m * BM) * 16;
    unsigned char* adst = As + buf * A_BYTES + wave * 64 * 16;
#if defined(__HIP_DEVICE_COMPILE__)  // the LDS address space does not exist in the host pass of this translation unit
#pragma unroll
    for (int i = 0; i < A_ITERS_DMA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wp_rs, (__attribute__((address_space(3))) void*)(adst + i * NT * 16), 16, a_voff[i], a_soff, 0, 0);
#else
    (void)a_soff;
    (void)adst;
#endif
  };
  auto store_lds = [&](int buf, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    unsigned char* bdst = Bs + buf * B_BYTES;
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int h = (B_ITEMS == 1) ? bh0 : it;
      if constexpr (PRESPLIT) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<f32x4*>(bdst + ((pc * 2 + h) * BN + bj) * 16) = bsplit[SET][it][pc];
        continue;
      }
      if constexpr (!PRESPLIT) {
        frag pieces[NP];
        split_frag<P>(breg[SET][it], inv_src_scale, pieces);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<frag*>(bdst + ((pc * 2 + h) * BN + bj) * 16) = pieces[pc];
      }
    }
    if (!A_DMA) {
      unsigned char* adst = As + buf * A_BYTES;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        const int id = t + i * NT;
        if (A_EXACT || id < A_CHUNKS) *reinterpret_cast<f32x4*>(adst + id * 16) = areg[SET][i];
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (SUB ? __builtin_popcount(cls_taps) : taps) * (p.Kp / 16);  // 0 for a class no tap reaches: dx = 0 there
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // loader state runs two steps ahead: regs set k&1 holds the gathers of step k
  int l_dma_kstep = l_kstep;  // weight DMA runs one step ahead
  int dma_tap = l_tap, dma_c0 = 0;
  // B_DMA: the pre-split operand also goes global -> LDS by DMA.  Its LDS image [piece][half][pixel][16 B] is lane-linear
  // per wave (64 consecutive pixels of one k-half), the gather address is per lane, and a padding pixel's out-of-range
  // offset makes the DMA deposit zeros -- so the K loop issues six DMAs per thread and touches no vector register for
  // staging: no ds_write, no register prefetch sets.
  auto dma_b = [&](int buf) {
    const int wave_px = __builtin_amdgcn_readfirstlane(bj - lane);
#pragma unroll
    for (int it = 0; it < B_ITEMS; ++it) {
      const int h = (B_ITEMS == 1) ? bh0 : it;  // BN = 256: the thread's pixel in both k-halves
      const int grp = (l_c0 >> 3) + h;
      unsigned char* bdst = Bs + buf * B_BYTES + (h * BN + wave_px) * 16;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int pc = 0; pc < P::NPU; ++pc) {
        const int soff = grp < C8 ? grp * HWs * 16 : 0x7FFFFFFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cb_rs[pc], (__attribute__((address_space(3))) void*)(bdst + pc * 2 * BN * 16), 16, l_voff, soff, 0, 0);
      }
#else
      (void)grp;
      (void)bdst;
#endif
    }
  };
  constexpr int DMA_PER_STEP = A_ITERS_DMA + P::NPU * B_ITEMS;  // LDS-DMA instructions one thread issues per K-step (B_DMA loop)
  if constexpr (B_DMA) {
    if (nsteps > 0) {
      dma_weights(0);
      dma_b(0);
    }
    if (NSTAGE == 3 && nsteps > 1) {  // step 1 -> stage 1, left in flight behind step 0
      advance();
      dma_weights(1);
      dma_b(1);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_STEP) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    if (nsteps > 0) {
      if (A_DMA) dma_weights(0);
      load_regs(S0{});
      store_lds(0, S0{});
      if (nsteps > 1) {
        advance();
        load_regs(S1{});
      }
    }
    __syncthreads();
  }

  frag fa[NP][WM], fb[NP][WN];
  auto read_frags = [&](int cur) {
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * (32 * WM) + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * (32 * WN) + l31) * 16;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[pc][i] = *reinterpret_cast<const frag*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[pc][j] = *reinterpret_cast<const frag*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
  };
  auto mfma_frags = [&]() {
#if defined(MCD_SETPRIO)
    __builtin_amdgcn_s_setprio(1);
#endif
    // the policy's cross terms, smallest first -- TERM-MAJOR: consecutive matrix instructions go to different accumulator tiles, so
    // none has to wait for the one just issued (tile-major order chained three dependent MFMAs per tile; same sums, same order per
    // tile, bit-identical results)
#if defined(MCD_MFMA_TILE_MAJOR)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int tm = 0; tm < P::NTERMS; ++tm) acc[i][j] = P::mfma(fa[P::TA[tm]][i], fb[P::TB[tm]][j], acc[i][j]);
#else
#pragma unroll
    for (int tm = 0; tm < P::NTERMS; ++tm)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa[P::TA[tm]][i], fb[P::TB[tm]][j], acc[i][j]);
#endif
#if defined(MCD_SETPRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  auto mfma_step = [&](int cur) {
    read_frags(cur);
    mfma_frags();
  };
  // one K-step s (cur = s & 1): gathers of step s+2 go to the register set that step s just vacated, the weight slab of
  // step s+1 is DMA'd into the LDS buffer released by the previous barrier, MFMAs of step s, then step s+1's gathers
  // (loaded one step earlier) are split and written to LDS.
  auto k_step = [&](int s, auto cur_c, auto steady_c) {
    constexpr int CUR = decltype(cur_c)::value;
    // STEADY: the caller guarantees s + 2 < nsteps, so nothing below is conditional.  That matters beyond the saved
    // branches: the compiler's wait-count bookkeeping merges its state over every path through the loop body, and with
    // the tail conditions inside it it concludes that older gathers may still be pending and drains vmcnt to 0 before
    // issuing new ones.
    constexpr bool STEADY = decltype(steady_c)::value;
    using SAME = std::integral_constant<int, CUR>;
    using OTHER = std::integral_constant<int, CUR ^ 1>;
    if (A_DMA && (STEADY || s + 1 < nsteps)) {
      if (SUB) {
        const unsigned rest = cls_taps & ~((2u << dma_tap) - 1u);
        if (rest) {
          dma_tap = __builtin_ctz(rest);
        } else {
          dma_tap = __builtin_ctz(cls_taps);
          dma_c0 += 16;
        }
        l_dma_kstep = (dma_c0 >> 4) * taps + dma_tap;
      } else {
        ++l_dma_kstep;
      }
      const int a_soff = (l_dma_kstep * NQ * p.Mp + tile_m * BM) * 16;
      unsigned char* adst = As + (CUR ^ 1) * A_BYTES + wave * 64 * 16;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int i = 0; i < A_ITERS_DMA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wp_rs, (__attribute__((address_space(3))) void*)(adst + i * NT * 16), 16, a_voff[i], a_soff, 0, 0);
#else
      (void)a_soff;
      (void)adst;
#endif
    }
    // the counted wait at the end of the step relies on program order: weight DMA first, gathers after it
    asm volatile("" ::: "memory");
    if (STEADY || s + 2 < nsteps) {
      advance();
      load_regs(SAME{});
    }
    mfma_step(CUR);
    if (STEADY || s + 1 < nsteps) store_lds(CUR ^ 1, OTHER{});
    if constexpr (A_DMA) {
      // __syncthreads() would drain every vector-memory operation (vmcnt(0)) because an LDS-DMA is pending, i.e. also
      // the gathers issued for step s+2 -- the prefetch distance would collapse to one MFMA phase.  What the barrier
      // has to guarantee is only: this wave's LDS writes have landed (lgkmcnt(0)) and the weight slab DMA'd for step s+1
      // is complete -- it is OLDER than the gathers of step s+2, so a counted wait leaves exactly those in flight.
      constexpr int AHEAD = (PRESPLIT ? NP : 8) * B_ITEMS;
      if (STEADY || s + 2 < nsteps)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(AHEAD) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      __syncthreads();
    }
  };
  using YES = std::integral_constant<bool, true>;
  using NO = std::integral_constant<bool, false>;
  if constexpr (B_DMA && NSTAGE == 3) {
    // stage of step s = s mod 3.  Entering iteration s: step s is complete in LDS (all waves), step s+1 is in flight.
    int cur = 0, nxt2 = 2;
    for (int s = 0; s < nsteps; ++s) {
      const bool more2 = s + 2 < nsteps;
      // fragment reads go out first -- right behind the barrier -- and the loader's scalar bookkeeping and DMA issue
      // run while they are in flight; the matrix instructions follow
#if defined(MCD_ABLATE) && (MCD_ABLATE & 4)
      if (s == 0)
#endif
      read_frags(cur);
      __builtin_amdgcn_sched_barrier(0);
      if (more2) {  // both operands of step s+2 -> the stage the barrier below step s-1 released
        advance();
#if !defined(MCD_ABLATE) || !(MCD_ABLATE & 1)  // MCD_ABLATE: timing-only variant builds (tools/build_variant.py), never shipped
        dma_weights(nxt2);
#endif
#if defined(MCD_ABLATE) && (MCD_ABLATE & 64)  // timing only: the pixel operand moved for ONE tap of a channel chunk (what a staged window would move)
        if (l_tap == 0) dma_b(nxt2);
#elif !defined(MCD_ABLATE) || !(MCD_ABLATE & 2)
        dma_b(nxt2);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#if !defined(MCD_ABLATE) || !(MCD_ABLATE & 8)
      mfma_frags();
#endif
      // step s+1 must have landed before the next iteration; the DMAs of step s+2 (younger) stay in flight
#if defined(MCD_ABLATE) && (MCD_ABLATE & 16)  // timing only: never wait for the DMAs (wrong results) -- what does their latency cost?
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(MCD_ABLATE) && (MCD_ABLATE & 32)  // timing only: no workgroup barrier (wrong results) -- what does the barrier skew cost?
      if (more2)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA_PER_STEP) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#elif defined(MCD_ABLATE) && (MCD_ABLATE & 128)  // timing only: a workgroup barrier and a DMA wait every SECOND K-step (wrong results)
      if (s & 1) {
        if (more2)
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_STEP) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#else
      if (more2)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_STEP) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
      cur = cur == 2 ? 0 : cur + 1;
      nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
  } else if constexpr (B_DMA) {
    for (int s = 0; s < nsteps; ++s) {
      const int cur = s & 1;
      if (s + 1 < nsteps) {  // both operands of step s+1 -> the buffer the barrier below step s-1 released
        advance();
        dma_weights(cur ^ 1);
        dma_b(cur ^ 1);
      }
      mfma_step(cur);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    int s = 0;
    for (; s + 3 < nsteps; s += 2) {  // steady state: both steps of the pair still have two successors
      k_step(s, S0{}, YES{});
      k_step(s + 1, S1{}, YES{});
    }
    for (; s < nsteps; s += 2) {
      k_step(s, S0{}, NO{});
      if (s + 1 < nsteps) k_step(s + 1, S1{}, NO{});
    }
  }

  // ---- epilogue (identical to conv_gemm.hip): acc[i][j][r] = D[row][col], row = (r&3) + 8*(r>>2) + 4*lh, col = l31
  if constexpr (P::SCALED) {
    const float osc = mcd_scale_of_bound(*p.src_bound) * mcd_scale_of_bound(*p.w_bound);  // exact power of two
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= osc;
  }
  const int m_wave = tile_m * BM + wm * (32 * WM);
  const int p_wave = tile_n * BN + wn * (32 * WN);
  if (!DGRAD && p.bias != nullptr) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j][r] += bv;
      }
  }
  if (!DGRAD && p.ep_scale != nullptr) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float sc = (m < p.M) ? p.ep_scale[m] : 0.f;
        const float sh = (m < p.M) ? p.ep_shift[m] : 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j][r] = fmaf(acc[i][j][r], sc, sh);
      }
  }
  bool colv[WN];
  size_t dbase[WN];
  size_t ubase[P::HALF_OUT ? WN : 1];  // (16-bit blocked output: the pixel's first unit)
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int pp = ltile * BN + wn * (32 * WN) + j * 32 + l31;  // pixel index inside the (class) ordering
    colv[j] = pp < Pc;
    int n = 0, rem = 0;
    if (colv[j]) {
      n = pp / HWc;
      rem = pp - n * HWc;
      if (SUB) {
        const int cy = rem / Wc;
        rem = ROW ? (2 * cy + ry) * p.Wd + (rem - cy * Wc) : (2 * cy + ry) * p.Wd + 2 * (rem - cy * Wc) + rx;
      }
    }
    dbase[j] = (size_t)n * p.M * HWd + rem;
    if constexpr (P::HALF_OUT) ubase[j] = (size_t)n * (p.M >> 3) * HWd + rem;
  }
  // ---- 16-bit channel-blocked output (the one-term arithmetic only; see conv_gemm_split_pp.hip): registers r = 4 q .. 4 q + 3 of a
  // tile are channels 8 q + 4 lh .. + 3 of one pixel, half of a 16-byte unit
  bool half_out = false;
  if constexpr (P::HALF_OUT) half_out = p.dst16 != nullptr;
  if constexpr (P::HALF_OUT) {
    if (half_out) {
      float inv_zs = 1.f;
      if (!DGRAD) {
        const float zb = (float)(taps * p.Cs) * (*p.src_bound) * (*p.w_bound);  // |z| <= taps Cs max|x| max|w|
        inv_zs = 1.f / mcd_scale_of_bound(zb);
        if (blockIdx.x == 0 && t == 0 && p.dst_bound != nullptr) *p.dst_bound = zb;
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m0 = m_wave + i * 32 + 8 * q;
          if (m0 < p.M) {  // (M is a multiple of 8: host)
            const size_t gofs = (size_t)(m0 >> 3) * HWd;
#pragma unroll
            for (int j = 0; j < WN; ++j)
              if (colv[j]) {
                const size_t byte = (ubase[j] + gofs) * 16 + lh * 8;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * q + k];
                if (DGRAD) {
                  if (p.ep_res16 != nullptr) {
                    const mcd_bf16x4 add = *reinterpret_cast<const mcd_bf16x4*>((const char*)p.ep_res16 + byte);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += (float)add[k];
                  }
                  mcd_bf16x4 o;
#pragma unroll
                  for (int k = 0; k < 4; ++k) o[k] = (__bf16)v[k];
                  *reinterpret_cast<mcd_bf16x4*>((char*)p.dst16 + byte) = o;
                } else {
                  mcd_f16x4 o;
#pragma unroll
                  for (int k = 0; k < 4; ++k) o[k] = (_Float16)(v[k] * inv_zs);
                  *reinterpret_cast<mcd_f16x4*>((char*)p.dst16 + byte) = o;
                }
              }
          }
        }
    }
  }
  if (!half_out) {
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (colv[j]) {
            float v = acc[i][j][r];
            if (!DGRAD && p.ep_scale != nullptr) {
              if (p.ep_res != nullptr) v += p.ep_res[dbase[j] + (size_t)m * HWd];
              if (p.ep_relu) v = fmaxf(v, 0.f);
            }
            // data gradient: the other gradient of the same tensor (a residual block's shortcut), added as autograd would add it
            if (DGRAD && p.ep_res != nullptr) v += p.ep_res[dbase[j] + (size_t)m * HWd];
            p.dst[dbase[j] + (size_t)m * HWd] = v;
          }
      }
    }
  }
  if (!DGRAD && p.stats != nullptr) {
    int cntw = p.P - p_wave;
    cntw = cntw < 0 ? 0 : (cntw > 32 * WN ? 32 * WN : cntw);
    const float inv = cntw > 0 ? 1.f / (float)cntw : 0.f;
    const size_t srow = ((size_t)(tile_n * WAVES_N + wn) * 3) * p.Mp;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      float my_mean = 0.f, my_m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) v += colv[j] ? acc[i][j][r] : 0.f;
        const float mean = wave_half_sum(v) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const float d = acc[i][j][r] - mean;
          q += colv[j] ? d * d : 0.f;
        }
        q = wave_half_sum(q);
        if (l31 == r) {
          my_mean = mean;
          my_m2 = q;
        }
      }
      if (l31 < 16) {
        const int m = m_wave + i * 32 + (l31 & 3) + 8 * (l31 >> 2) + 4 * lh;
        p.stats[srow + m] = (float)cntw;
        p.stats[srow + p.Mp + m] = my_mean;
        p.stats[srow + 2 * (size_t)p.Mp + m] = my_m2;
      }
    }
  }
}

// ---- weight packing: w[Cout][Cin][T] (fp32) -> [chunk*T + tap][piece][half][Mp][8] 16-bit pieces, chunk = 16 K-channels
// MODE 0: m = cout, k = cin (forward); MODE 1: m = cin, k = cout (dgrad)
// blockIdx.y selects the image, so both are produced by one launch (weights are re-packed after every optimizer step)
template <class P>
__global__ void pack_weights_split_kernel(const float* __restrict__ w, typename P::elem* __restrict__ out_fprop,
                                          typename P::elem* __restrict__ out_dgrad, const float* __restrict__ w_bound, int Cout, int Cin,
                                          int T) {
  constexpr int NP = P::NP;
  const int mode = blockIdx.y;
  typename P::elem* __restrict__ out = mode == 0 ? out_fprop : out_dgrad;
  if (out == nullptr) return;
  const float inv_scale = 1.f / operand_scale<P>(w_bound);
  const int M = mode == 0 ? Cout : Cin;
  const int K = mode == 0 ? Cin : Cout;
  const int Mp = M <= 32 ? 32 : (M <= 64 ? 64 : ((M + 127) / 128) * 128);  // mcd_mp
  const int Kp = ((K + 15) / 16) * 16;
  const int64_t total = (int64_t)(Kp / 16) * T * 2 * Mp * 8;  // one thread per (kstep, half, m, e): writes all pieces
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    int64_t r = i >> 3;
    const int m = (int)(r % Mp);
    r /= Mp;
    const int h = (int)(r & 1);
    const int64_t kstep = r >> 1;
    const int tap = (int)(kstep % T);
    const int chunk = (int)(kstep / T);
    const int k = chunk * 16 + 8 * h + e;
    float v = 0.f;
    if (m < M && k < K) v = mode == 0 ? w[((int64_t)m * Cin + k) * T + tap] : w[((int64_t)k * Cin + m) * T + tap];
    typename P::elem q[NP];
    P::split(v, inv_scale, q);
    const int64_t base = kstep * (2 * NP) * (int64_t)Mp * 8;
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) out[base + ((pc * 2 + h) * (int64_t)Mp + m) * 8 + e] = q[pc];
  }
}

// bound[0] = max |x|: integer atomic max on the bit pattern of |x| -- order-preserving for non-negative floats, exact and
// independent of the order of arrival; a NaN pattern compares above inf, so non-finite data yields a non-finite bound
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out) {
  unsigned m = 0u;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {  // 16-byte loads, four in flight
    const uint4* __restrict__ x4 = reinterpret_cast<const uint4*>(x);
    const int64_t n4 = n >> 2;
    auto amax4 = [](uint4 q) {
      const unsigned a = q.x & 0x7FFFFFFFu, b = q.y & 0x7FFFFFFFu, c = q.z & 0x7FFFFFFFu, d = q.w & 0x7FFFFFFFu;
      const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
      return ab > cd ? ab : cd;
    };
    int64_t j = i;
    for (; j + 3 * step < n4; j += 4 * step) {
      const uint4 q0 = x4[j], q1 = x4[j + step], q2 = x4[j + 2 * step], q3 = x4[j + 3 * step];
      const unsigned a = amax4(q0), b = amax4(q1), c = amax4(q2), d = amax4(q3);
      const unsigned ab = a > b ? a : b, cd = c > d ? c : d, t = ab > cd ? ab : cd;
      m = t > m ? t : m;
    }
    for (; j < n4; j += step) {
      const unsigned t = amax4(x4[j]);
      m = t > m ? t : m;
    }
    i += n4 << 2;  // the tail of up to three elements
  }
  for (; i < n; i += step) {
    const unsigned b = __float_as_uint(x[i]) & 0x7FFFFFFFu;
    m = b > m ? b : m;
  }
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(out, m);
}

// ---- all convolutions of a model in two launches (one optimizer step re-packs every image): entry e of the device tables is
// ptrs[e] = {w, out_fprop, out_dgrad, bound}, dims[e] = {Cout, Cin, T, 0}; blockIdx.y = entry (absmax) or 2 * entry + mode (pack)
__global__ __launch_bounds__(256) void absmax_multi_kernel(const int64_t* __restrict__ ptrs, const int32_t* __restrict__ dims) {
  const int e = blockIdx.y;
  const float* __restrict__ x = reinterpret_cast<const float*>(ptrs[4 * e + 0]);
  unsigned* __restrict__ out = reinterpret_cast<unsigned*>(ptrs[4 * e + 3]);
  const int64_t n = (int64_t)dims[4 * e + 0] * dims[4 * e + 1] * dims[4 * e + 2];
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  unsigned m = 0u;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i + 3 * step < n; i += 4 * step) {  // four loads in flight
    const unsigned b0 = __float_as_uint(x[i]) & 0x7FFFFFFFu, b1 = __float_as_uint(x[i + step]) & 0x7FFFFFFFu;
    const unsigned b2 = __float_as_uint(x[i + 2 * step]) & 0x7FFFFFFFu, b3 = __float_as_uint(x[i + 3 * step]) & 0x7FFFFFFFu;
    const unsigned c0 = b0 > b1 ? b0 : b1, c1 = b2 > b3 ? b2 : b3;
    const unsigned c = c0 > c1 ? c0 : c1;
    m = c > m ? c : m;
  }
  for (; i < n; i += step) {
    const unsigned b = __float_as_uint(x[i]) & 0x7FFFFFFFu;
    m = b > m ? b : m;
  }
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(out, m);
}

// A workgroup moves tiles of (MT output rows) x (16 contraction channels) x (all taps) through LDS: the fp32 kernel is read in
// contiguous runs (16 T floats per row for the forward image, MT T floats per channel for the dgrad image -- a thread-per-unit
// gather strides 18 KB between lanes and pays a DRAM row miss per 64 bytes), the image is written in runs of MT 16-byte units.
template <class P>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const int64_t* __restrict__ ptrs, const int32_t* __restrict__ dims) {
  constexpr int NP = P::NP;
  __shared__ float tile[32 * 16 * 9 + 32];
  const int e = blockIdx.y >> 1, mode = blockIdx.y & 1;
  const float* __restrict__ w = reinterpret_cast<const float*>(ptrs[4 * e + 0]);
  typename P::elem* __restrict__ out = reinterpret_cast<typename P::elem*>(ptrs[4 * e + 1 + mode]);
  if (out == nullptr) return;
  const float inv_scale = 1.f / operand_scale<P>(reinterpret_cast<const float*>(ptrs[4 * e + 3]));
  const int Cout = dims[4 * e + 0], Cin = dims[4 * e + 1], T = dims[4 * e + 2];
  const int M = mode == 0 ? Cout : Cin;
  const int K = mode == 0 ? Cin : Cout;
  const int Mp = M <= 32 ? 32 : (M <= 64 ? 64 : ((M + 127) / 128) * 128);  // mcd_mp
  const int Kp = ((K + 15) / 16) * 16;
  const int MT = T <= 9 ? 32 : (T <= 18 ? 16 : (T <= 36 ? 8 : 4));  // rows per tile: MT * 16 * T floats of LDS (T <= 72)
  const int row_len = 16 * T + 1;                                   // one tile row (16 channels x T taps), padded
  const int m_tiles = Mp / MT, chunks = Kp / 16;
  const int t = threadIdx.x;
  for (int tl = blockIdx.x; tl < m_tiles * chunks; tl += gridDim.x) {
    const int chunk = tl / m_tiles, m0 = (tl - chunk * m_tiles) * MT, k0 = chunk * 16;
    __syncthreads();  // the previous tile has been consumed
    if (mode == 0) {  // rows of the source = output rows: 16 T contiguous floats each
      for (int idx = t; idx < MT * 16 * T; idx += 256) {
        const int mm = idx / (16 * T), col = idx - mm * (16 * T);
        const int kk = col / T;
        tile[mm * row_len + col] = (m0 + mm < M && k0 + kk < K) ? w[((int64_t)(m0 + mm) * Cin + k0) * T + col] : 0.f;
      }
    } else {  // rows of the source = contraction channels: MT T contiguous floats each
      for (int idx = t; idx < 16 * MT * T; idx += 256) {
        const int kk = idx / (MT * T), col = idx - kk * (MT * T);
        const int mm = col / T, tap = col - mm * T;
        tile[mm * row_len + kk * T + tap] = (m0 + mm < M && k0 + kk < K) ? w[((int64_t)(k0 + kk) * Cin + m0) * T + col] : 0.f;
      }
    }
    __syncthreads();
    for (int u = t; u < T * 2 * MT; u += 256) {  // unit = (tap, half, row): MT consecutive units are contiguous in the image
      const int mm = u % MT, r = u / MT;
      const int h = r & 1, tap = r >> 1;
      float v[8];
#pragma unroll
      for (int el = 0; el < 8; ++el) v[el] = tile[mm * row_len + (8 * h + el) * T + tap];
      typename P::frag q[NP];
      split_frag<P>(v, inv_scale, q);
      const int64_t base = ((int64_t)chunk * T + tap) * (2 * NP) * (int64_t)Mp * 8;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<typename P::frag*>(out + base + ((pc * 2 + h) * (int64_t)Mp + m0 + mm) * 8) = q[pc];
    }
  }
}

int split_check(const mcdseg_conv_desc* d, int math, const char* who, bool operands = true) {
  MCD_REQUIRE(d != nullptr, "%s: null descriptor", who);
  MCD_REQUIRE(mcd_math_known(math), "%s: math must be MCDSEG_MATH_BF16X6, MCDSEG_MATH_F16X3 or MCDSEG_MATH_F16X1 (got %d)", who, math);
  MCD_REQUIRE(d->N > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "%s: non-positive dims", who);
  MCD_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "%s: bad kernel geometry", who);
  const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  MCD_REQUIRE(ho == d->Ho && wo == d->Wo, "%s: Ho/Wo (%d,%d) do not match geometry (%d,%d)", who, d->Ho, d->Wo, ho, wo);
  MCD_REQUIRE(!operands || ((int64_t)d->N * d->Cin * d->H * d->W * 4 < (1ll << 31) && (int64_t)d->N * d->Cout * d->Ho * d->Wo * 4 < (1ll << 31)),
              "%s: activation tensor must stay below 2 GiB (32-bit buffer offsets); split the batch", who);
  return 0;
}

int64_t split_image_bytes(int math, int M, int K, int T) {
  return (int64_t)(round_up(K, 16) / 16) * T * 2 * mcd_math_pieces(math) * mcd_mp(M) * 16;
}

// development knob: number of tile slots from which the 256 x 128 tile is preferred (see launch<>)
int big_tile_min_slots() { return (int)mcd_opt(MCD_OPT_BIGTILE_MIN_SLOTS); }  // (a test runs one problem on several tiles)

template <class P, int WM, int WN, int WAVES_M, int WAVES_N, bool DGRAD>
void launch_cfg(const ConvSplitParams& p, int64_t pix0, hipStream_t st) {
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  ConvSplitParams q = p;
  q.sub = 0;
  int n_tiles = ceil_div(p.P, BN);
  q.tile_n0 = (int)(pix0 / BN);  // (pix0: a multiple of 256, the pixels the ping-pong kernel has taken)
  q.tile_n1 = n_tiles;
  n_tiles -= q.tile_n0;
  if (DGRAD && p.stride == 2 && p.KH * p.KW <= 32) {  // parity classes: tiles never straddle a class
    q.sub = 1;
    int t0 = 0;
    for (int c = 0; c < 4; ++c) {
      q.cls_tile0[c] = t0;
      const int hc = (p.Hd - (c >> 1) + 1) / 2, wc = (p.Wd - (c & 1) + 1) / 2;
      t0 += ceil_div(p.N * hc * wc, BN);
    }
    q.cls_tile0[4] = t0;
    n_tiles = t0;
    const int c0 = q.cls_tile0[1];
    const bool interleave = mcd_opt(MCD_OPT_DGRAD_INTERLEAVE) != 0;
    if (interleave && q.cls_tile0[2] == 2 * c0 && q.cls_tile0[3] == 3 * c0 && t0 == 4 * c0) q.sub = 2;
    // row classes (both x-parities in one tile: dense stores, twice the K-steps): measured alone at BASELINE config 2's shapes
    // (tools/probes/dgrad_s2_ab.py, profiles/r06_dgrad_s2_forms.txt) they win where there is next to no K loop -- the 1 x 1 projections
    // without an addend, 0.074 -> 0.050 ms at 32 -> 64 -- and lose on the 3 x 3 layers (0.253 -> 0.294 ms at 16 -> 32: those launches
    // are bound by their short K loops' latency, not by the store pattern).  Option value 2 forces them everywhere (tests).
    const bool row_auto = interleave && p.KH * p.KW == 1 && p.ep_res == nullptr && p.ep_res16 == nullptr;
    if (mcd_opt(MCD_OPT_DGRAD_INTERLEAVE) == 2 || row_auto) {
      q.sub = 3;
      t0 = 0;
      for (int c = 0; c < 2; ++c) {
        q.cls_tile0[c] = t0;
        t0 += ceil_div(p.N * ((p.Hd - c + 1) / 2) * p.Wd, BN);
      }
      q.cls_tile0[2] = q.cls_tile0[3] = q.cls_tile0[4] = t0;
      n_tiles = t0;
    }
  }
  dim3 grid(8 * ceil_div(n_tiles, 8) * (p.Mp / BM));
  if (p.src_cb != nullptr)
    hipLaunchKernelGGL((conv_gemm_split_kernel<P, WM, WN, WAVES_M, WAVES_N, DGRAD, true>), grid, dim3(256), 0, st, q);
  else
    hipLaunchKernelGGL((conv_gemm_split_kernel<P, WM, WN, WAVES_M, WAVES_N, DGRAD, false>), grid, dim3(256), 0, st, q);
}

// development knob: number of 128 x 256 tiles from which that tile is preferred over 128 x 128 (see tile_config)
int wide_tile_min_slots() { return (int)mcd_opt(MCD_OPT_WIDETILE_MIN_SLOTS); }

// Workgroup tile of the implicit GEMM with M output rows (padded Mp) and P pixels, as WM WN WAVES_M WAVES_N packed into decimal
// digits: 4222 = 256 x 128 (each wave 128 x 64), 4214 = 128 x 256 (each wave 128 x 64, four waves along the pixels), 2222 =
// 128 x 128, 2214 = 64 x 256, 1214 = 32 x 256.  The two large tiles need the pre-split operand (all-DMA K loop).
//   256 x 128: the gathered operand is fetched once per TWO row tiles and a K-step issues fewer fragment reads per MFMA -- worth
//     +5 % on the 512-channel layers, but only while the grid still holds two full rounds of the 512 workgroup slots this tile
//     leaves (256-channel layers and small batches lose);
//   128 x 256: the same wave tile for the 128- and 256-channel layers -- the WEIGHT slab is fetched once per two pixel tiles.
//     Results (outputs and BatchNorm partial rows, one per 64 pixels) are bit for bit those of the 128 x 128 tile.
int tile_config(int M, int64_t P, bool presplit) {
  const int bm = mcd_bm(M), mp = mcd_mp(M);
  if (bm == 128 && presplit && (mp % 256) == 0 && ceil_div64(P, 128) * (mp / 256) >= big_tile_min_slots()) return 4222;
  if (bm == 128 && presplit && ceil_div64(P, 256) * (mp / 128) >= wide_tile_min_slots()) return 4214;
  if (bm == 128) return 2222;
  return bm == 64 ? 2214 : 1214;
}

template <class P, bool DGRAD>
void launch(const ConvSplitParams& p, int64_t pix0, hipStream_t st) {
  switch (tile_config(p.M, p.P, p.src_cb != nullptr)) {  // (of the WHOLE problem, whatever the ping-pong kernel has taken: one row numbering)
    case 4222: launch_cfg<P, 4, 2, 2, 2, DGRAD>(p, pix0, st); break;
    case 4214: launch_cfg<P, 4, 2, 1, 4, DGRAD>(p, pix0, st); break;
    case 2222: launch_cfg<P, 2, 2, 2, 2, DGRAD>(p, pix0, st); break;
    case 2214: launch_cfg<P, 2, 2, 1, 4, DGRAD>(p, pix0, st); break;
    default: launch_cfg<P, 1, 2, 1, 4, DGRAD>(p, pix0, st); break;
  }
}

}  // namespace

// the 8-wave ping-pong tile (conv_gemm_split_pp.hip) takes whole rounds of 256 x 256 tiles, the kernels of this file the rest
int64_t mcdseg_internal_conv_pp_pixels(const ConvSplitParams& p, int math, bool dgrad);
int mcdseg_internal_conv_pp_launch(const ConvSplitParams& p, int math, bool dgrad, int64_t pixels, hipStream_t st);
int mcdseg_internal_conv_pp_rest(const ConvSplitParams& p, int math, bool dgrad);
int mcdseg_internal_conv_pp_wide(const ConvSplitParams& p, int math, bool dgrad);
int mcdseg_internal_conv_pp_deep(const ConvSplitParams& p, int math);
int mcdseg_internal_conv_pp_rest_launch(const ConvSplitParams& p, int math, bool dgrad, int64_t pix0, hipStream_t st);

namespace {

// part 0: the whole convolution; 1: only the pixels of the ping-pong kernel; 2: only the rest (mcdseg_conv_split_parts)
template <bool DGRAD>
int launch_math(int math, const ConvSplitParams& p, int part, hipStream_t st) {
  const int64_t pp = mcdseg_internal_conv_pp_pixels(p, math, DGRAD);
  if (pp > 0 && part != 2)
    if (int rc = mcdseg_internal_conv_pp_launch(p, math, DGRAD, pp, st)) return rc;
  if (pp >= p.P || part == 1) return 0;
  if (mcdseg_internal_conv_pp_rest(p, math, DGRAD)) return mcdseg_internal_conv_pp_rest_launch(p, math, DGRAD, pp, st);
  if (math == MCDSEG_MATH_F16X3)
    launch<SplitF16x3, DGRAD>(p, pp, st);
  else if (math == MCDSEG_MATH_F16X1)
    launch<SplitF16x1, DGRAD>(p, pp, st);  // the same operands, one term
  else
    launch<SplitBf16x6, DGRAD>(p, pp, st);
  return 0;
}

}  // namespace

// direct convolution for the network stem (conv_stem_x6.hip; bf16x6 arithmetic whatever `math` says -- its operand is
// the network input, which no producer pre-splits, and the layer is 0.5 % of the step)
bool mcdseg_internal_stem_ok(const mcdseg_conv_desc* d);
int64_t mcdseg_internal_stem_stat_rows(const mcdseg_conv_desc* d);
int64_t mcdseg_internal_stem_image_bytes();
int mcdseg_internal_stem_pack(const mcdseg_conv_desc* d, const float* w, void* out, hipStream_t st);
int mcdseg_internal_stem_fprop(const mcdseg_conv_desc* d, const float* x, const void* wp, const float* bias, float* y, float* stats,
                               const float* ep_scale, const float* ep_shift, const float* ep_res, int ep_relu, hipStream_t st);

// LDS-window kernels of the thin 3x3 layers (conv_thin_window.hip): forward and stride-1 dgrad, f16x3 with a pre-split operand
int mcdseg_internal_thin_window_ok(const mcdseg_conv_desc* d, int dgrad);
int64_t mcdseg_internal_thin_window_stat_rows(const mcdseg_conv_desc* d);
int mcdseg_internal_thin_window_launch(const mcdseg_conv_desc* d, int dgrad, const void* src_cb, const float* src_bound, const void* wp,
                                       int64_t wp_bytes, const float* w_bound, float* dst, float* stats, hipStream_t st);

// the stem's forward image: [standard split image (window kernel)][the direct kernel's own image]
static int64_t stem_image_offset(const mcdseg_conv_desc* d, int math) {
  return (split_image_bytes(math, d->Cout, d->Cin, d->KH * d->KW) + 255) / 256 * 256;
}

extern "C" int32_t mcdseg_conv_split_direct_ok(const mcdseg_conv_desc* d) { return d != nullptr && mcdseg_internal_stem_ok(d) ? 1 : 0; }

static int64_t stat_rows_of(const mcdseg_conv_desc* d, bool presplit) {
  const int cfg = tile_config(d->Cout, (int64_t)d->N * d->Ho * d->Wo, presplit);
  const int64_t waves_n = cfg % 10, bn = 32 * ((cfg / 100) % 10) * waves_n;  // pixel tile and column waves of launch<>
  return ceil_div64((int64_t)d->N * d->Ho * d->Wo, bn) * waves_n;
}

extern "C" int64_t mcdseg_conv_split_stat_rows(const mcdseg_conv_desc* d) {
  if (d == nullptr) return -22;
  if (mcdseg_internal_stem_ok(d)) return mcdseg_internal_stem_stat_rows(d);
  return stat_rows_of(d, false);
}

static int fill_fprop_params(const mcdseg_conv_desc* d, int math, const void* x_cb, ConvSplitParams& p);

extern "C" int64_t mcdseg_conv_split_stat_rows_for(const mcdseg_conv_desc* d, int32_t math, int32_t presplit) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  if (d == nullptr) return -22;
  if (math == MCDSEG_MATH_F16X3 && presplit && mcdseg_internal_thin_window_ok(d, 0)) return mcdseg_internal_thin_window_stat_rows(d);
  if (mcdseg_internal_stem_ok(d)) return mcdseg_internal_stem_stat_rows(d);
  if (presplit) {  // the 256 x 320 ping-pong tile: one row per wave column of 160 pixels
    ConvSplitParams p;
    if (fill_fprop_params(d, math, d, p) == 0) {
      const int kind = mcdseg_internal_conv_pp_wide(p, math, false);
      if (kind == 3) return ceil_div64((int64_t)d->N * d->Ho * d->Wo, 160);  // (256 x 160: one wave column)
      if (kind) return 2 * ceil_div64((int64_t)d->N * d->Ho * d->Wo, 320);
    }
  }
  return stat_rows_of(d, presplit != 0);
}

extern "C" int32_t mcdseg_conv_split_tile_config(int32_t M, int64_t P, int32_t presplit) { return tile_config(M, P, presplit != 0); }

// 1 when mcdseg_conv_split_fprop / _dgrad run this geometry on the LDS-window kernel (for profilers and the benchmark's accounting)
extern "C" int32_t mcdseg_conv_split_window_ok(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  return d != nullptr && math == MCDSEG_MATH_F16X3 && presplit && mcdseg_internal_thin_window_ok(d, dgrad) ? 1 : 0;
}

extern "C" int mcdseg_absmax(const float* x, int64_t n, float* bound, void* stream) {
  MCD_REQUIRE(x && bound && n > 0, "absmax: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(bound, 0, sizeof(float), st);
  int64_t blocks = ceil_div64(n, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, (unsigned*)bound);
  MCD_LAUNCH_CHECK("absmax");
  return 0;
}

extern "C" int mcdseg_conv_split_packed_bytes(const mcdseg_conv_desc* d, int32_t math, int64_t* fprop_bytes, int64_t* dgrad_bytes) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(d != nullptr, "conv_split_packed_bytes: null descriptor");
  MCD_REQUIRE(math == MCDSEG_MATH_BF16X6 || math == MCDSEG_MATH_F16X3, "conv_split_packed_bytes: unknown math %d", math);
  const int T = d->KH * d->KW;
  if (fprop_bytes) {
    *fprop_bytes = split_image_bytes(math, d->Cout, d->Cin, T);
    if (mcdseg_internal_stem_ok(d)) *fprop_bytes = stem_image_offset(d, math) + mcdseg_internal_stem_image_bytes();
  }
  if (dgrad_bytes) *dgrad_bytes = split_image_bytes(math, d->Cin, d->Cout, T);
  return 0;
}

extern "C" int mcdseg_conv_split_pack_weights(const mcdseg_conv_desc* d, int32_t math, const float* w, void* wp_fprop, void* wp_dgrad,
                                              float* w_bound, void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  if (int rc = split_check(d, math, "conv_split_pack_weights", false)) return rc;  // weights only: any batch size
  MCD_REQUIRE(w != nullptr && (wp_fprop != nullptr || wp_dgrad != nullptr), "conv_split_pack_weights: null pointer");
  MCD_REQUIRE(math != MCDSEG_MATH_F16X3 || w_bound != nullptr, "conv_split_pack_weights: the f16x3 images need the weight bound scalar");
  const int T = d->KH * d->KW;
  hipStream_t st = (hipStream_t)stream;
  if (wp_fprop != nullptr && mcdseg_internal_stem_ok(d))  // the stem: the direct kernel's own image behind the standard one
    if (int rc = mcdseg_internal_stem_pack(d, w, (char*)wp_fprop + stem_image_offset(d, math), st)) return rc;
  if (math == MCDSEG_MATH_F16X3)
    if (int rc = mcdseg_absmax(w, (int64_t)d->Cout * d->Cin * T, w_bound, stream)) return rc;
  int64_t most = 0;
  for (int mode = 0; mode < 2; ++mode) {
    const int M = mode == 0 ? d->Cout : d->Cin, K = mode == 0 ? d->Cin : d->Cout;
    const int64_t total = (int64_t)(round_up(K, 16) / 16) * T * 2 * mcd_mp(M) * 8;
    if ((mode == 0 ? wp_fprop : wp_dgrad) && total > most) most = total;
  }
  int64_t blocks = ceil_div64(most, 256);
  if (blocks > 2048) blocks = 2048;
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(pack_weights_split_kernel<SplitF16x3>, dim3((unsigned)blocks, 2), dim3(256), 0, st, w, (_Float16*)wp_fprop,
                       (_Float16*)wp_dgrad, (const float*)w_bound, d->Cout, d->Cin, T);
  else
    hipLaunchKernelGGL(pack_weights_split_kernel<SplitBf16x6>, dim3((unsigned)blocks, 2), dim3(256), 0, st, w, (__bf16*)wp_fprop,
                       (__bf16*)wp_dgrad, (const float*)nullptr, d->Cout, d->Cin, T);
  MCD_LAUNCH_CHECK("conv_split_pack_weights");
  return 0;
}

extern "C" int mcdseg_conv_split_pack_weights_multi(const int64_t* ptrs, const int32_t* dims, int32_t n, int32_t math, float* bounds,
                                                    void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(ptrs && dims && n > 0 && n <= 32767, "conv_split_pack_weights_multi: bad table");
  MCD_REQUIRE(math == MCDSEG_MATH_BF16X6 || math == MCDSEG_MATH_F16X3, "conv_split_pack_weights_multi: unknown math %d", math);
  MCD_REQUIRE(math != MCDSEG_MATH_F16X3 || bounds != nullptr, "conv_split_pack_weights_multi: f16x3 needs the bounds array");
  // (kernels of up to 72 taps: the pack kernel stages tiles of 4 rows x 16 channels x T taps in 18 KB of LDS; larger ones
  // take mcdseg_conv_split_pack_weights)
  hipStream_t st = (hipStream_t)stream;
  // workgroups per (convolution, image): a 512 -> 512 3 x 3 kernel is 512 tiles of 32 rows x 16 channels x 9 taps, and a workgroup walks
  // its tiles one after the other (stage, barrier, convert, store)
  const unsigned pack_blocks = (unsigned)(mcd_opt(MCD_OPT_PACK_BLOCKS) > 0 ? mcd_opt(MCD_OPT_PACK_BLOCKS) : 192);  // (96 -> 192: 0.194 -> 0.177 ms per optimizer step, round 5)
  if (math == MCDSEG_MATH_F16X3) {
    (void)hipMemsetAsync(bounds, 0, sizeof(float) * (size_t)n, st);
    hipLaunchKernelGGL(absmax_multi_kernel, dim3(48, (unsigned)n), dim3(256), 0, st, ptrs, dims);
    MCD_LAUNCH_CHECK("absmax_multi");
    hipLaunchKernelGGL(pack_weights_multi_kernel<SplitF16x3>, dim3(pack_blocks, 2u * (unsigned)n), dim3(256), 0, st, ptrs, dims);
  } else {
    hipLaunchKernelGGL(pack_weights_multi_kernel<SplitBf16x6>, dim3(pack_blocks, 2u * (unsigned)n), dim3(256), 0, st, ptrs, dims);
  }
  MCD_LAUNCH_CHECK("conv_split_pack_weights_multi");
  return 0;
}

// one piece of the companion of this call's N images, and the distance between the pieces (the companion's own batch d->Ncb)
// (F16X1 multiplies -- and therefore reads -- the leading piece only: its piece stride is 0, so that a companion stored as ONE piece,
// the 2-byte activation storage of round 6, can never be read past its end by a kernel that addresses "piece 1")
static int split_cb_bytes(const mcdseg_conv_desc* d, int math, int C, int HW, const void* cb, int* piece_bytes, long long* piece_stride) {
  *piece_bytes = 0;
  *piece_stride = 0;
  if (cb == nullptr) return 0;
  MCD_REQUIRE((C % 8) == 0, "conv_split: a pre-split operand needs a channel count divisible by 8 (got %d)", C);
  MCD_REQUIRE(d->Ncb == 0 || d->Ncb >= d->N, "conv_split: Ncb (%d) must be 0 or at least N (%d)", d->Ncb, d->N);
  const int64_t b = (int64_t)d->N * C * HW * 2;
  MCD_REQUIRE(b < (1ll << 31), "conv_split: one piece of the pre-split operand exceeds 2 GiB; split the batch");
  *piece_bytes = (int)b;
  *piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * C * HW * 2;
  return 0;
}

static int fill_fprop_params(const mcdseg_conv_desc* d, int math, const void* x_cb, ConvSplitParams& p);

static int split_fprop_impl(const mcdseg_conv_desc* d, int math, const float* x, const void* x_cb, const float* x_bound, const void* wp,
                            const float* w_bound, const float* bias, float* y, float* stats, const float* ep_scale,
                            const float* ep_shift, const float* ep_res, int ep_relu, void* stream, int part = 0, void* z16 = nullptr,
                            float* z_bound = nullptr) {
  if (int rc = split_check(d, math, "conv_split_fprop", z16 == nullptr)) return rc;
  MCD_REQUIRE((x || x_cb) && wp && (y || z16), "conv_split_fprop: null pointer");
  if (z16 != nullptr) {
    MCD_REQUIRE(mcdseg_conv_split_half_ok(d, math, 0) && x_cb != nullptr && z_bound != nullptr && bias == nullptr && ep_scale == nullptr,
                "conv_split_fprop_half: the 16-bit output needs f16x1, a pre-split input, channel counts divisible by 8, a bound scalar "
                "to write and no bias / affine epilogue (mcdseg_conv_split_half_ok tells)");
    MCD_REQUIRE((int64_t)d->N * d->Cout * d->Ho * d->Wo * 2 < (1ll << 32), "conv_split_fprop_half: output above 4 GiB; split the batch");
  }
  const bool stem = mcdseg_internal_stem_ok(d);
  // the stem with a (zero-padded, 8-channel) companion of the network input runs on the window kernel; its fp32 form (and every
  // bias / affine epilogue) on the direct bf16x6 kernel
  const int smath = mcd_storage_math(math);
  const bool stem_window = stem && smath == MCDSEG_MATH_F16X3 && x_cb != nullptr && bias == nullptr && ep_scale == nullptr &&
                           mcdseg_internal_thin_window_ok(d, 0);
  if (part == 1 && (stem || (smath == MCDSEG_MATH_F16X3 && x_cb != nullptr && mcdseg_internal_thin_window_ok(d, 0)))) return 0;  // (no ping-pong part)
  if (stem && !stem_window) {
    MCD_REQUIRE(x != nullptr, "conv_split_fprop: the stem kernel reads the fp32 input");
    return mcdseg_internal_stem_fprop(d, x, (const char*)wp + stem_image_offset(d, math), bias, y, stats, ep_scale, ep_shift, ep_res,
                                      ep_relu, (hipStream_t)stream);
  }
  MCD_REQUIRE(smath != MCDSEG_MATH_F16X3 || (x_bound && w_bound), "conv_split_fprop: f16x3 needs the operand and weight bound scalars");
  if (smath == MCDSEG_MATH_F16X3 && x_cb != nullptr && mcdseg_internal_thin_window_ok(d, 0)) {
    // the window kernel has no bias / affine epilogue: those callers (conv+bias heads, folded-BN inference) have no companion
    MCD_REQUIRE(bias == nullptr && ep_scale == nullptr, "conv_split_fprop: the thin-layer window kernel takes no bias / affine epilogue");
    return mcdseg_internal_thin_window_launch(d, 0, x_cb, x_bound, wp, split_image_bytes(math, d->Cout, d->Cin, d->KH * d->KW), w_bound, y,
                                              stats, (hipStream_t)stream);
  }
  ConvSplitParams p;
  if (int rc = fill_fprop_params(d, math, x_cb, p)) return rc;
  p.src_bound = x_bound; p.w_bound = w_bound;
  p.src = x; p.wp = wp; p.bias = bias; p.dst = y; p.stats = stats;
  p.ep_scale = ep_scale; p.ep_shift = ep_shift; p.ep_res = ep_res; p.ep_relu = ep_relu;
  p.dst16 = z16; p.dst_bound = z_bound;
  if (int rc = launch_math<false>(math, p, part, (hipStream_t)stream)) return rc;
  MCD_LAUNCH_CHECK("conv_split_fprop");
  return 0;
}

// geometry of the forward implicit GEMM (everything but the data pointers)
static int fill_fprop_params(const mcdseg_conv_desc* d, int math, const void* x_cb, ConvSplitParams& p) {
  if (int rc = split_cb_bytes(d, math, d->Cin, d->H * d->W, x_cb, &p.cb_bytes, &p.cb_piece_stride)) return rc;
  p.src_cb = x_cb;
  p.N = d->N;
  p.Cs = d->Cin; p.Hs = d->H; p.Ws = d->W;
  p.M = d->Cout; p.Hd = d->Ho; p.Wd = d->Wo;
  p.Mp = mcd_mp(d->Cout);
  p.Kp = round_up(d->Cin, 16);
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.P = d->N * d->Ho * d->Wo;
  p.src_bytes = (int)((int64_t)d->N * d->Cin * d->H * d->W * 4);
  const int64_t wb = split_image_bytes(math, d->Cout, d->Cin, d->KH * d->KW);
  MCD_REQUIRE(wb < (1ll << 31), "conv_split_fprop: packed weights exceed 2 GiB");
  p.wp_bytes = (int)wb;
  p.sub = 0; p.tile_n0 = 0; p.tile_n1 = 0;
  p.ep_res_lds = 0; p.ep_res_bytes = 0;
  p.dst16 = nullptr; p.dst_bound = nullptr; p.ep_res16 = nullptr;
  return 0;
}

extern "C" int mcdseg_conv_split_fprop(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                                       const void* wp_fprop, const float* w_bound, const float* bias, float* y, float* stat_partials,
                                       void* stream) {
  return split_fprop_impl(d, math, x, x_cb, x_bound, wp_fprop, w_bound, bias, y, stat_partials, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mcdseg_conv_split_fprop_affine(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb,
                                              const float* x_bound, const void* wp_fprop, const float* w_bound, const float* scale,
                                              const float* shift, const float* residual, int32_t relu, float* y, void* stream) {
  MCD_REQUIRE(scale && shift, "conv_split_fprop_affine: null scale/shift");
  return split_fprop_impl(d, math, x, x_cb, x_bound, wp_fprop, w_bound, nullptr, y, nullptr, scale, shift, residual, relu, stream);
}

static int fill_dgrad_params(const mcdseg_conv_desc* d, int math, const void* dy_cb, ConvSplitParams& p) {
  if (int rc = split_cb_bytes(d, math, d->Cout, d->Ho * d->Wo, dy_cb, &p.cb_bytes, &p.cb_piece_stride)) return rc;
  p.src_cb = dy_cb;
  p.N = d->N;
  p.Cs = d->Cout; p.Hs = d->Ho; p.Ws = d->Wo;
  p.M = d->Cin; p.Hd = d->H; p.Wd = d->W;
  p.Mp = mcd_mp(d->Cin);
  p.Kp = round_up(d->Cout, 16);
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.P = d->N * d->H * d->W;
  p.src_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 4);
  const int64_t wb = split_image_bytes(math, d->Cin, d->Cout, d->KH * d->KW);
  MCD_REQUIRE(wb < (1ll << 31), "conv_split_dgrad: packed weights exceed 2 GiB");
  p.wp_bytes = (int)wb;
  p.sub = 0; p.tile_n0 = 0; p.tile_n1 = 0;
  p.ep_res_lds = 0; p.ep_res_bytes = 0;
  p.dst16 = nullptr; p.dst_bound = nullptr; p.ep_res16 = nullptr;
  return 0;
}

static int split_dgrad_impl(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                            const void* wp_dgrad, const float* w_bound, float* dx, void* stream, int part, const float* addend = nullptr,
                            void* dx16 = nullptr, const void* addend16 = nullptr) {
  if (int rc = split_check(d, math, "conv_split_dgrad", dx16 == nullptr)) return rc;
  MCD_REQUIRE((dy || dy_cb) && wp_dgrad && (dx || dx16), "conv_split_dgrad: null pointer");
  if (dx16 != nullptr) {
    MCD_REQUIRE(mcdseg_conv_split_half_ok(d, math, 1) && dy_cb != nullptr && addend == nullptr,
                "conv_split_dgrad_half: the 16-bit output needs f16x1, a pre-split gradient and channel counts divisible by 8 "
                "(mcdseg_conv_split_half_ok tells)");
    MCD_REQUIRE((int64_t)d->N * d->Cin * d->H * d->W * 2 < (1ll << 32), "conv_split_dgrad_half: output above 4 GiB; split the batch");
  }
  const int smath = mcd_storage_math(math);
  MCD_REQUIRE(smath != MCDSEG_MATH_F16X3 || (dy_bound && w_bound), "conv_split_dgrad: f16x3 needs the operand and weight bound scalars");
  MCD_REQUIRE(addend == nullptr || !(smath == MCDSEG_MATH_F16X3 && dy_cb != nullptr && mcdseg_internal_thin_window_ok(d, 1)),
              "conv_split_dgrad_add: the thin-layer window kernel takes no addend (mcdseg_conv_split_window_ok tells)");
  if (smath == MCDSEG_MATH_F16X3 && dy_cb != nullptr && mcdseg_internal_thin_window_ok(d, 1))
    return part == 1 ? 0
                     : mcdseg_internal_thin_window_launch(d, 1, dy_cb, dy_bound, wp_dgrad, split_image_bytes(math, d->Cin, d->Cout, d->KH * d->KW),
                                                          w_bound, dx, nullptr, (hipStream_t)stream);
  ConvSplitParams p;
  if (int rc = fill_dgrad_params(d, math, dy_cb, p)) return rc;
  p.src_bound = dy_bound; p.w_bound = w_bound;
  p.src = dy; p.wp = wp_dgrad; p.bias = nullptr; p.dst = dx; p.stats = nullptr;
  p.ep_scale = nullptr; p.ep_shift = nullptr; p.ep_res = addend; p.ep_relu = 0;
  p.dst16 = dx16; p.ep_res16 = addend16;
  if (addend != nullptr) {
    const bool stage = mcd_opt(MCD_OPT_DGRAD_ADD_LDS) != 0;  // development knob
    const int64_t bytes = (int64_t)d->N * d->Cin * d->H * d->W * 4;
    p.ep_res_lds = stage && ((d->H * d->W) & 3) == 0 && (reinterpret_cast<uintptr_t>(addend) & 15) == 0 && bytes < (1ll << 32) ? 1 : 0;
    p.ep_res_bytes = (unsigned)(bytes < (1ll << 32) ? bytes : 0);
  }
  if (int rc = launch_math<true>(math, p, part, (hipStream_t)stream)) return rc;
  MCD_LAUNCH_CHECK("conv_split_dgrad");
  return 0;
}

extern "C" int mcdseg_conv_split_dgrad(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                                       const void* wp_dgrad, const float* w_bound, float* dx, void* stream) {
  return split_dgrad_impl(d, math, dy, dy_cb, dy_bound, wp_dgrad, w_bound, dx, stream, 0);
}

// dx = data gradient + addend (an fp32 tensor of dx's shape; it may not alias dx): what autograd computes with a separate element-wise
// add when the convolution's input has a second consumer -- the shortcut of a residual block (models/drn.py:43-59, 79-100).  The sum
// is formed in the epilogue in the order of that add, so the result is bit for bit dgrad-then-add.  `part` as in _dgrad_part.
extern "C" int mcdseg_conv_split_dgrad_add(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                                           const void* wp_dgrad, const float* w_bound, const float* addend, float* dx, int32_t part,
                                           void* stream) {
  MCD_REQUIRE(part >= 0 && part <= 2, "conv_split_dgrad_add: part must be 0 (whole), 1 (ping-pong tiles) or 2 (the rest)");
  MCD_REQUIRE(addend != nullptr && addend != dx, "conv_split_dgrad_add: the addend is a tensor of its own");
  return split_dgrad_impl(d, math, dy, dy_cb, dy_bound, wp_dgrad, w_bound, dx, stream, part, addend);
}

// ---- the two launches of one convolution, separately (profilers and bench.py's per-kernel HIP events; results are those of the whole call)
extern "C" int64_t mcdseg_conv_split_parts(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad) {
  if (d == nullptr || !mcd_math_known(math) || !presplit) return 0;
  const int smath = mcd_storage_math(math);
  if (smath == MCDSEG_MATH_F16X3 && mcdseg_internal_thin_window_ok(d, dgrad ? 1 : 0)) return 0;
  if (!dgrad && mcdseg_internal_stem_ok(d)) return 0;
  ConvSplitParams p;
  if (dgrad ? fill_dgrad_params(d, math, d, p) : fill_fprop_params(d, math, d, p)) return 0;
  return mcdseg_internal_conv_pp_pixels(p, math, dgrad != 0);
}

// 1 when the launch that `part` 2 stands for (every pixel mcdseg_conv_split_parts does not give to the 256 x 256 tile) runs on the
// 256 x 128 ping-pong tile, 0 when it runs on the 4-wave tiles (kernel names for profilers / bench.py)
extern "C" int32_t mcdseg_conv_split_rest_pingpong(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad) {
  if (d == nullptr || !mcd_math_known(math) || !presplit) return 0;
  const int smath = mcd_storage_math(math);
  if (smath == MCDSEG_MATH_F16X3 && mcdseg_internal_thin_window_ok(d, dgrad ? 1 : 0)) return 0;
  if (!dgrad && mcdseg_internal_stem_ok(d)) return 0;
  ConvSplitParams p;
  if (dgrad ? fill_dgrad_params(d, math, d, p) : fill_fprop_params(d, math, d, p)) return 0;
  return mcdseg_internal_conv_pp_rest(p, math, dgrad != 0);
}

// 1 when the whole convolution runs on the 256 x 320 ping-pong tile (mcdseg_conv_split_parts then returns every pixel)
extern "C" int32_t mcdseg_conv_split_wide_pingpong(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad) {
  if (d == nullptr || !mcd_math_known(math) || !presplit) return 0;
  const int smath = mcd_storage_math(math);
  if (smath == MCDSEG_MATH_F16X3 && mcdseg_internal_thin_window_ok(d, dgrad ? 1 : 0)) return 0;
  if (!dgrad && mcdseg_internal_stem_ok(d)) return 0;
  ConvSplitParams p;
  if (dgrad ? fill_dgrad_params(d, math, d, p) : fill_fprop_params(d, math, d, p)) return 0;
  return mcdseg_internal_conv_pp_wide(p, math, dgrad != 0);  // 1: 256 x 320; 2: 128 x 320 (output rows a multiple of 128 only); 3: 256 x 160
}

extern "C" int mcdseg_conv_split_fprop_part(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                                            const void* wp_fprop, const float* w_bound, const float* bias, float* y, float* stat_partials,
                                            int32_t part, void* stream) {
  MCD_REQUIRE(part >= 0 && part <= 2, "conv_split_fprop_part: part must be 0 (whole), 1 (ping-pong tiles) or 2 (the rest)");
  return split_fprop_impl(d, math, x, x_cb, x_bound, wp_fprop, w_bound, bias, y, stat_partials, nullptr, nullptr, nullptr, 0, stream, part);
}

extern "C" int mcdseg_conv_split_dgrad_part(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                                            const void* wp_dgrad, const float* w_bound, float* dx, int32_t part, void* stream) {
  MCD_REQUIRE(part >= 0 && part <= 2, "conv_split_dgrad_part: part must be 0 (whole), 1 (ping-pong tiles) or 2 (the rest)");
  return split_dgrad_impl(d, math, dy, dy_cb, dy_bound, wp_dgrad, w_bound, dx, stream, part);
}

// ---- 2-byte activation storage (round 6; BASELINE config 5 "bf16"): the same convolutions writing channel-blocked 16-bit tensors
// [N][C/8][H*W][8] -- the unit layout of the companions -- instead of fp32 NCHW.  F16X1 only.
//   forward        z as fp16 of z / scale(z_bound); the kernel writes z_bound = KH KW Cin * x_bound * w_bound (|z| cannot exceed it), the
//                  BatchNorm partial rows come from the fp32 accumulators as always;
//   data gradient  dx as bf16 (+ an addend in the same layout: the other gradient of a residual block's input).
// 1 when these entry points take the geometry (`dgrad`: the data gradient): one-term arithmetic, both channel counts multiples of 8,
// at least 16 contraction channels, not the thin layers' window kernels, not the stem.
extern "C" int32_t mcdseg_conv_split_half_ok(const mcdseg_conv_desc* d, int32_t math, int32_t dgrad) {
  if (d == nullptr || math != MCDSEG_MATH_F16X1) return 0;
  if ((d->Cin & 7) != 0 || (d->Cout & 7) != 0 || (dgrad ? d->Cout : d->Cin) < 16) return 0;
  if (mcdseg_internal_thin_window_ok(d, dgrad ? 1 : 0) || mcdseg_internal_stem_ok(d)) return 0;
  return 1;
}

extern "C" int mcdseg_conv_split_fprop_half(const mcdseg_conv_desc* d, int32_t math, const void* x_cb, const float* x_bound,
                                            const void* wp_fprop, const float* w_bound, void* z16, float* z_bound, float* stat_partials,
                                            int32_t part, void* stream) {
  MCD_REQUIRE(part >= 0 && part <= 2, "conv_split_fprop_half: part must be 0 (whole), 1 (ping-pong tiles) or 2 (the rest)");
  MCD_REQUIRE(z16 != nullptr, "conv_split_fprop_half: null output");
  return split_fprop_impl(d, math, nullptr, x_cb, x_bound, wp_fprop, w_bound, nullptr, nullptr, stat_partials, nullptr, nullptr, nullptr, 0,
                          stream, part, z16, z_bound);
}

extern "C" int mcdseg_conv_split_dgrad_half(const mcdseg_conv_desc* d, int32_t math, const void* dy_cb, const float* dy_bound,
                                            const void* wp_dgrad, const float* w_bound, const void* addend16, void* dx16, int32_t part,
                                            void* stream) {
  MCD_REQUIRE(part >= 0 && part <= 2, "conv_split_dgrad_half: part must be 0 (whole), 1 (ping-pong tiles) or 2 (the rest)");
  MCD_REQUIRE(dx16 != nullptr && addend16 != dx16, "conv_split_dgrad_half: null output (or the addend aliases it)");
  return split_dgrad_impl(d, math, nullptr, dy_cb, dy_bound, wp_dgrad, w_bound, nullptr, stream, part, nullptr, dx16, addend16);
}

// 1 when the ping-pong launches of this convolution (forward, or `dgrad`) run with two K-steps per barrier interval (the policy name
// rocprofv3 prints is then SplitF16x1D; the 256 x 128 tile never does)
extern "C" int32_t mcdseg_conv_split_pp_deep(const mcdseg_conv_desc* d, int32_t math, int32_t dgrad) {
  if (d == nullptr || !mcd_math_known(math)) return 0;
  ConvSplitParams p;
  if (dgrad ? fill_dgrad_params(d, math, d, p) : fill_fprop_params(d, math, d, p)) return 0;
  return mcdseg_internal_conv_pp_deep(p, math);
}
