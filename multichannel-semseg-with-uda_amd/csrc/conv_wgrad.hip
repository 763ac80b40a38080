// Weight-gradient convolution on the f32-input MFMA of gfx950.
//
//   dW[co][ci][tap] = sum_{n,oy,ox} dY[n][co][oy][ox] * X[n][ci][oy*s + ky*d - pad][ox*s + kx*d - pad]
//
// GEMM view per tap: D[co][ci] = A[co][pix] * B[pix][ci], contraction over pixels.  Both operands are
// pixel-contiguous in NCHW, so each wave instruction of the staging loop reads two 128-B pixel runs;
// the LDS image is [row][32 pixels + 1 pad] which makes the transposed MFMA fragment reads
// (32 consecutive rows, one pixel) conflict-free.  The pixel axis is split over workgroups
// (image x chunk); every workgroup writes its partial D tile to a slab and a second kernel sums the
// slabs in a fixed order (fp64 accumulate) -- deterministic, no float atomics.
#include <cstdlib>
#include "split.h"

namespace {

constexpr int BKP_MAX = 32;    // pixels per K-step: 16 for the 128x128 tile (4 workgroups/CU), else 32

struct WgradParams {
  const float* x;
  const float* dy;
  float* slab;
  int N, Cin, H, W, Cout, Ho, Wo;
  int KH, KW, stride, pad, dil;
  int co_p, ci_p;
  int chunk, chunks_per_img, splits;
  int x_bytes, dy_bytes;
  int groups;
};

struct WgradPlan {
  int cfg;  // 0: 128x128, 1: 64x64, 2: 32x32 (one wave)
  int bm, bn;
  int co_p, ci_p;
  int chunk, chunks_per_img, splits;
  int cinp, groups;  // cfg 3: padded input channels per tap (8/16) and number of tap groups
  int64_t slab_floats;
};

WgradPlan make_plan(const mcdseg_conv_desc* d) {
  WgradPlan pl;
  const int T = d->KH * d->KW;
  const int lo = d->Cout < d->Cin ? d->Cout : d->Cin;
  pl.cinp = 0;
  pl.groups = T;
  if (d->Cin <= 16 && T > 1) {
    // thin inputs (stem, layer1, layer2): several taps share one 32-column MFMA tile, columns = (tap_local, ci)
    pl.cfg = 3; pl.bm = 32; pl.bn = 32;
    pl.cinp = d->Cin <= 8 ? 8 : 16;
    pl.groups = ceil_div(T, 32 / pl.cinp);
  } else if (lo > 64) {
    pl.cfg = 0; pl.bm = 128; pl.bn = 128;
  } else if (lo > 32 || (lo > 16 && (d->Cin & 7) == 0 && (d->Cout & 7) == 0)) {
    // (24- and 32-channel sides too when the channel-blocked layout applies: half of a 64 x 64 tile is padding, but the plan that reads
    // both companions -- conv_wgrad_split_tr64_kernel -- moves the 32 -> 64 stride-2 layers at their bytes where the one-wave f32
    // tiles below ran at 35 TFLOP/s: round 5)
    pl.cfg = 1; pl.bm = 64; pl.bn = 64;
  } else {
    pl.cfg = 2; pl.bm = 32; pl.bn = 32;
  }
  pl.co_p = round_up(d->Cout, pl.bm);
  pl.ci_p = pl.cfg == 3 ? 32 : round_up(d->Cin, pl.bn);
  const int64_t tiles = (int64_t)(pl.co_p / pl.bm) * (pl.ci_p / pl.bn) * pl.groups;
  const int64_t tuned_wgs = mcd_opt(MCD_OPT_WGRAD_WGS);  // development knob: target workgroup count of the 128/64-tile plans
  const int64_t want_wgs = (pl.cfg >= 2) ? 4096 : tuned_wgs;
  const int64_t want_splits = ceil_div64(want_wgs, tiles);
  const int hw = d->Ho * d->Wo;
  int cpi = (int)ceil_div64(want_splits, d->N);
  if (cpi < 1) cpi = 1;
  // whole pixel splits are dealt round-robin to the 8 XCDs (L2 locality): a split count that is not a multiple of 8 leaves
  // some XCDs with one split more than others (N = 10: 62 % efficiency).  Up to 4x more chunks per image buys that back.
  if (pl.cfg <= 1 && (int64_t)d->N * cpi > 8 && ((int64_t)d->N * cpi) % 8 != 0) {
    for (int m = 2; m <= 4; ++m)
      if (((int64_t)d->N * cpi * m) % 8 == 0) {
        cpi *= m;
        break;
      }
  }
  int chunk = round_up(ceil_div(hw, cpi), BKP_MAX);
  if (chunk < 8 * BKP_MAX) chunk = 8 * BKP_MAX;  // at least 8 K-steps of work per workgroup
  if (chunk > round_up(hw, BKP_MAX)) chunk = round_up(hw, BKP_MAX);
  pl.chunk = chunk;
  pl.chunks_per_img = ceil_div(hw, chunk);
  pl.splits = d->N * pl.chunks_per_img;
  pl.slab_floats = (int64_t)pl.splits * pl.groups * pl.co_p * pl.ci_p;
  return pl;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int BKP>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N == 4 ? 3 : 1)) void conv_wgrad_kernel(WgradParams p) {
  constexpr int LDP = BKP + 1;  // padded LDS row
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  constexpr int RP = NT / BKP;  // rows staged per pass
  constexpr int A_ITERS = BM / RP;
  constexpr int B_ITERS = BN / RP;

  __shared__ float smem[2 * (BM + BN) * LDP];
  float* As = smem;                  // [2][BM][LDP]
  float* Bs = smem + 2 * BM * LDP;   // [2][BN][LDP]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware order (ids b, b+8 share an L2): an XCD owns whole pixel splits; inside a split the tap index
  // runs fastest, then the ci tile, then the co tile -- workgroups that stream the same dY rows / the same
  // (shifted) X rows are resident together and advance through the pixels in step, so the streams are fetched
  // once per XCD and re-used from L2.
  const int ci_tiles = p.ci_p / BN;
  const int co_tiles = p.co_p / BM;
  const int T_ = p.KH * p.KW;
  const int per_split = co_tiles * ci_tiles * T_;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  int rem = slot % per_split;
  const int tap = rem % T_;
  rem /= T_;
  const int tile_ci = rem % ci_tiles;
  const int tile_co = rem / ci_tiles;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int ky = tap / p.KW;
  const int kx = tap - ky * p.KW;
  const int HoWo = p.Ho * p.Wo;
  const int HW = p.H * p.W;
  const int r_begin = chunk_id * p.chunk;
  int r_end = r_begin + p.chunk;
  if (r_end > HoWo) r_end = HoWo;

  const int sk = t % BKP;
  const int srg = t / BKP;
  // Buffer loads (descriptor + per-row offset in SGPRs, one 32-bit per-lane offset per K-step): a lane past the
  // end of its pixel chunk, or whose tap falls into the zero padding, uses an offset beyond num_records and reads 0.
  // Rows of the zero-padded tile tail (co >= Cout / ci >= Cin) are NOT masked: they alias finite data of the next
  // image (or read 0 past the tensor end) and only feed slab rows/columns the reduction never reads.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int a_soff0 = (n * p.Cout + tile_co * BM) * HoWo * 4;
  const int b_soff0 = (n * p.Cin + tile_ci * BN) * HW * 4;
  const int a_sstep = RP * HoWo * 4;
  const int b_sstep = RP * HW * 4;
  const unsigned a_row = (unsigned)srg * (unsigned)HoWo;
  const unsigned b_row = (unsigned)srg * (unsigned)HW;

  float areg[A_ITERS], breg[B_ITERS];
  auto load_regs = [&](int r0) {
    const int r = r0 + sk;
    const bool rv = r < r_end;
    unsigned a_voff = OOB, b_voff = OOB;
    if (rv) {
      a_voff = (a_row + (unsigned)r) * 4u;
      const int oy = r / p.Wo;
      const int ox = r - oy * p.Wo;
      const int iy = oy * p.stride + ky * p.dil - p.pad;
      const int ix = ox * p.stride + kx * p.dil - p.pad;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) b_voff = (b_row + (unsigned)(iy * p.W + ix)) * 4u;
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i)
      areg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dy_rs, a_voff, a_soff0 + i * a_sstep, 0));
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i)
      breg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, b_voff, b_soff0 + i * b_sstep, 0));
  };
  auto store_lds = [&](int buf) {
    float* a = As + buf * BM * LDP;
    float* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) a[(srg + i * RP) * LDP + sk] = areg[i];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) b[(srg + i * RP) * LDP + sk] = breg[i];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (r_end - r_begin + BKP - 1) / BKP;
  if (nsteps > 0) {
    load_regs(r_begin);
    store_lds(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_regs(r_begin + (s + 1) * BKP);
    const float* a_base = As + cur * BM * LDP + (wm * 32 * WM + l31) * LDP + lh;
    const float* b_base = Bs + cur * BN * LDP + (wn * 32 * WN + l31) * LDP + lh;
    // fragments of pixel-pair ks+1 are read from LDS while the MFMAs of pixel-pair ks run
    float a[2][WM], b[2][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[0][i] = a_base[i * 32 * LDP];
#pragma unroll
    for (int j = 0; j < WN; ++j) b[0][j] = b_base[j * 32 * LDP];
#pragma unroll
    for (int ks = 0; ks < BKP / 2; ++ks) {
      const int cb = ks & 1;
      if (ks + 1 < BKP / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[cb ^ 1][i] = a_base[i * 32 * LDP + 2 * (ks + 1)];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[cb ^ 1][j] = b_base[j * 32 * LDP + 2 * (ks + 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][i], b[cb][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  const int T = p.KH * p.KW;
  float* out = p.slab + ((size_t)split * T + tap) * p.co_p * p.ci_p;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_co * BM + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = tile_ci * BN + wn * 32 * WN + j * 32 + l31;
        out[(size_t)row * p.ci_p + col] = acc[i][j][r];
      }
    }
}

// Thin-input variant (Cin <= 16: 7x7 stem, layer1, layer2).  One wave per workgroup, a 32 (co) x 32 (columns) tile
// whose columns are (tap_local, ci) pairs: 32/CINP taps share one MFMA tile instead of one tap per tile padded from
// Cin to 32 columns -- 4x fewer MFMAs and dY re-reads for the 6-channel stem, 2x for 16 channels.
// ROWS16 (Cout <= 16, the stem and layer1): 16-row tile on v_mfma_f32_16x16x4_f32 instead of padding the output channels
// to the 32 rows of v_mfma_f32_32x32x2_f32 -- the same FLOP rate per instruction, half the instructions.
template <int CINP, bool ROWS16>
__global__ __launch_bounds__(64) void conv_wgrad_thin_kernel(WgradParams p) {
  constexpr int BKP = 32, LDP = BKP + 1, BM = ROWS16 ? 16 : 32, BN = 32;
  constexpr int A_REGS = BM / 2;
  constexpr int TPT = 32 / CINP;  // taps per tile
  __shared__ float smem[2 * (BM + BN) * LDP];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDP;
  const int t = threadIdx.x;
  const int l31 = t & 31, lh = t >> 5;

  const int co_tiles = ROWS16 ? 1 : p.co_p / BM;  // the host plans 32-row tiles; 16 rows cover Cout <= 16 in one
  const int per_split = co_tiles * p.groups;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  const int rem = slot % per_split;
  const int group = rem % p.groups;
  const int tile_co = rem / p.groups;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int HoWo = p.Ho * p.Wo;
  const int HW = p.H * p.W;
  const int r_begin = chunk_id * p.chunk;
  int r_end = r_begin + p.chunk;
  if (r_end > HoWo) r_end = HoWo;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int sk = l31, srg = lh;  // pixel within the step, row parity
  const int a_soff0 = (n * p.Cout + tile_co * BM) * HoWo * 4;
  const int b_soff0 = n * p.Cin * HW * 4;
  const unsigned a_row = (unsigned)srg * (unsigned)HoWo;
  const unsigned b_row = (unsigned)srg * (unsigned)HW;
  // tap geometry of this group's TPT taps (taps past the kernel window only feed columns nobody reads)
  int tdy[TPT], tdx[TPT];
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int tap = group * TPT + q;
    const int ky = tap / p.KW;
    const int kx = tap - ky * p.KW;
    tdy[q] = ky * p.dil - p.pad;
    tdx[q] = kx * p.dil - p.pad;
  }

  float areg[A_REGS], breg[16];
  auto load_regs = [&](int r0) {
    const int r = r0 + sk;
    const bool rv = r < r_end;
    unsigned a_voff = OOB;
    unsigned b_voff[TPT];
#pragma unroll
    for (int q = 0; q < TPT; ++q) b_voff[q] = OOB;
    if (rv) {
      a_voff = (a_row + (unsigned)r) * 4u;
      const int oy = r / p.Wo;
      const int ox = r - oy * p.Wo;
#pragma unroll
      for (int q = 0; q < TPT; ++q) {
        const int iy = oy * p.stride + tdy[q];
        const int ix = ox * p.stride + tdx[q];
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) b_voff[q] = (b_row + (unsigned)(iy * p.W + ix)) * 4u;
      }
    }
#pragma unroll
    for (int i = 0; i < A_REGS; ++i)  // rows srg + 2i of dY
      areg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dy_rs, a_voff, a_soff0 + 2 * i * HoWo * 4, 0));
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // column srg + 2i = (tap_local, ci): tap_local = 2i / CINP, ci = 2i % CINP + srg
      const int q = (2 * i) / CINP;
      const int ci_even = (2 * i) % CINP;
      breg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, b_voff[q], b_soff0 + ci_even * HW * 4, 0));
    }
  };
  auto store_lds = [&](int buf) {
    float* a = As + buf * BM * LDP;
    float* b = Bs + buf * BN * LDP;
#pragma unroll
    for (int i = 0; i < A_REGS; ++i) a[(srg + 2 * i) * LDP + sk] = areg[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[(srg + 2 * i) * LDP + sk] = breg[i];
  };

  f32x16 acc;  // ROWS16: two 16x16 blocks (columns 0-15 and 16-31) in acc[0..3] and acc[4..7]
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = {0.f, 0.f, 0.f, 0.f};
  const int l15 = t & 15, lq = t >> 4;
  const int nsteps = (r_end - r_begin + BKP - 1) / BKP;
  if (nsteps > 0) {
    load_regs(r_begin);
    store_lds(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_regs(r_begin + (s + 1) * BKP);
    if constexpr (ROWS16) {
      // 16x16x4: lane (row / column l & 15, k = l >> 4) holds one element of A and of B
      const float* a_base = As + cur * BM * LDP + l15 * LDP + lq;
      const float* b_base = Bs + cur * BN * LDP + l15 * LDP + lq;
#pragma unroll
      for (int kk = 0; kk < BKP; kk += 4) {
        const float av = a_base[kk];
        acc_lo = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_base[kk], acc_lo, 0, 0, 0);
        acc_hi = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_base[16 * LDP + kk], acc_hi, 0, 0, 0);
      }
    } else {
      const float* a_base = As + cur * BM * LDP + l31 * LDP + lh;
      const float* b_base = Bs + cur * BN * LDP + l31 * LDP + lh;
#pragma unroll
      for (int kk = 0; kk < BKP; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_base[kk], b_base[kk], acc, 0, 0, 0);
    }
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }
  float* out = p.slab + ((size_t)split * p.groups + group) * p.co_p * 32;
  if constexpr (ROWS16) {  // D[row 4*(l>>4) + r][column l & 15 (+16)]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      out[(size_t)row * 32 + l15] = acc_lo[r];
      out[(size_t)row * 32 + 16 + l15] = acc_hi[r];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tile_co * BM + (r & 3) + 8 * (r >> 2) + 4 * lh;
    out[(size_t)row * 32 + l31] = acc[r];
  }
}

// one wave per output element: lanes stride over the splits (hundreds of them for the full-resolution layers), then a
// wave reduction in fp64 -- fixed order, so still deterministic
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout,
                                                                int Cin, int T, int co_p, int groups, int cinp, int splits) {
  const int64_t total = (int64_t)T * Cout * Cin;
  const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= total) return;
  const int lane = threadIdx.x & 63;
  const int ci = (int)(i % Cin);
  const int64_t r = i / Cin;
  const int co = (int)(r % Cout);
  const int tap = (int)(r / Cout);
  const int tpt = 32 / cinp;
  const int group = tap / tpt;
  const int col = (tap - group * tpt) * cinp + ci;
  const size_t stride = (size_t)groups * co_p * 32;
  const float* src = slab + ((size_t)group * co_p + co) * 32 + col;
  double s = 0.0;
  for (int k = lane; k < splits; k += 64) s += (double)src[(size_t)k * stride];
  s = wave_sum_d(s);
  if (lane == 0) dw[((size_t)co * Cin + ci) * T + tap] = (float)s;
}

// x_bound / dy_bound (NULL unless the slabs come from the SplitF16x3 kernels): the slab sums are in scaled units and are
// multiplied by scale(x) * scale(dy), an exact power of two
// A block owns PB consecutive (co, ci) pairs x T taps -- one contiguous run of dw [Cout][Cin][T].  Its threads stride over the
// (tap, pair) items, pair fastest, so the slab reads are coalesced along ci; each item is summed over the splits in a fixed
// order (fp64) and parked in LDS, from where the run is written out contiguously (writing from registers would scatter 4-byte
// stores 4 T bytes apart).  PB = 64 (256 for 1x1 kernels), halved down to 16 while the layer has fewer than 512 blocks.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                                           int T, int co_p, int ci_p, int splits, int PB,
                                                           const float* __restrict__ x_bound, const float* __restrict__ dy_bound) {
  extern __shared__ float stage[];  // [PB][T]
  const int64_t pairs = (int64_t)Cout * Cin;
  const int64_t p0 = blockIdx.x * (int64_t)PB;
  const int64_t left = pairs - p0;
  const int npair = (int)(left < PB ? left : PB);
  const double sc = x_bound != nullptr ? (double)mcd_scale_of_bound(*x_bound) * (double)mcd_scale_of_bound(*dy_bound) : 1.0;
  const size_t stride = (size_t)T * co_p * ci_p;
  for (int i = threadIdx.x; i < PB * T; i += 256) {
    const int tap = i / PB;
    const int pl = i - tap * PB;
    if (pl >= npair) continue;
    const int64_t pr = p0 + pl;
    const int ci = (int)(pr % Cin);
    const int co = (int)(pr / Cin);
    const float* src = slab + ((size_t)tap * co_p + co) * ci_p + ci;
    double s = 0.0;
    int k = 0;
    // (a thread owns one or two items and walks ~100 slabs: the pass is bound by the latency of its dependent load batches, not by
    // bytes -- 39 us for 67 MB with four loads in flight.  Sixteen in flight, ADDED IN THE SAME ORDER: the same bits, round 5)
    for (; k + 16 <= splits; k += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; k + 4 <= splits; k += 4) {  // four loads in flight, added in slab order
      const float v0 = src[(size_t)k * stride], v1 = src[(size_t)(k + 1) * stride];
      const float v2 = src[(size_t)(k + 2) * stride], v3 = src[(size_t)(k + 3) * stride];
      s += (double)v0;
      s += (double)v1;
      s += (double)v2;
      s += (double)v3;
    }
    for (; k < splits; ++k) s += (double)src[(size_t)k * stride];
    stage[pl * T + tap] = (float)(s * sc);
  }
  __syncthreads();
  const int n = npair * T;
  float* out = dw + p0 * T;
  for (int i = threadIdx.x; i < n; i += 256) out[i] = stage[i];
}

}  // namespace

int mcdseg_internal_wgrad_split_launch(const mcdseg_conv_desc* d, int math, const float* x, const float* x_bound, const float* dy,
                                       const float* dy_bound, float* slab, int co_p, int ci_p, int chunk, int chunks_per_img, int splits,
                                       hipStream_t st);
int mcdseg_internal_wgrad_split_cb_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb, float* slab,
                                          int co_p, int ci_p, int chunks_per_img, int splits, hipStream_t st);

int mcdseg_internal_wgrad_cb_variant(const mcdseg_conv_desc* d, int math, int co_p, int ci_p, int splits);
int mcdseg_internal_wgrad_split_tr64_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb, float* slab, int co_p,
                                            int ci_p, int chunks_per_img, int splits, hipStream_t st);

// thin layers (Cin = 16, Cout 16 / 32, 3x3) from both pre-split companions: conv_wgrad_thin_tr.hip (MCDSEG_WGRAD_THIN_TR=0 turns it off)
int mcdseg_internal_wgrad_thin_tr_ok(const mcdseg_conv_desc* d);
size_t mcdseg_internal_wgrad_thin_tr_ws(const mcdseg_conv_desc* d);
int mcdseg_internal_wgrad_thin_tr_launch(const mcdseg_conv_desc* d, const void* x_cb, const float* x_bound, const void* dy_cb,
                                         const float* dy_bound, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
// the 256-channel-and-wider layers from both companions: eight-wave ping-pong kernel over a stream-K decomposition (conv_wgrad_split_pp.hip)
int mcdseg_internal_wgrad_pp_plan(const mcdseg_conv_desc* d, int math, int* L, size_t* slab_floats, int* slabs = nullptr);
int mcdseg_internal_wgrad_pp_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const float* x_bound, const void* dy_cb,
                                    const float* dy_bound, float* dw, float* slab, hipStream_t st);
// ... and the 128-channel layers: the same kernel structure on tiles of 128 (co) x one kernel row of three taps x 128 (ci)
int mcdseg_internal_wgrad_pp3_plan(const mcdseg_conv_desc* d, int math, int* L, size_t* slab_floats, int* slabs = nullptr);
int mcdseg_internal_wgrad_pp3_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const float* x_bound, const void* dy_cb,
                                     const float* dy_bound, float* dw, float* slab, hipStream_t st);
static bool thin_tr_applies(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb) {
  const bool on = mcd_opt(MCD_OPT_WGRAD_THIN_TR) != 0;
  return on && mcd_storage_math(math) == MCDSEG_MATH_F16X3 && x_cb && dy_cb && mcdseg_internal_wgrad_thin_tr_ok(d);
}

// the 64 x 64 plan (32 < min(Cin, Cout) <= 64) from both pre-split companions: f16x3 only (MCDSEG_WGRAD_TR64=0 turns it off)
static bool tr64_applies(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb, int cfg) {
  const bool on = mcd_opt(MCD_OPT_WGRAD_TR64) != 0;
  return on && cfg == 1 && mcd_storage_math(math) == MCDSEG_MATH_F16X3 && x_cb && dy_cb && (d->Cin & 7) == 0 && (d->Cout & 7) == 0;
}

static int wgrad_impl(const mcdseg_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                      int math, const void* x_cb, const float* x_bound, const void* dy_cb, const float* dy_bound, void* stream);

// Which kernel mcdseg_conv_wgrad / mcdseg_conv_split_wgrad launch for this geometry (math = 0 for mcdseg_conv_wgrad):
// 0..3 the f32 plans (128x128, 64x64, 32x32 tiles, tap-packed thin), 10 split arithmetic from fp32 operands, 11 / 12 / 13 / 14 from
// both pre-split companions: register-transposing, transposed-read 128x128, transposed-read 256x128, transposed-read 64-channel
// tap pairs, 15 the thin-layer window kernel, 16 transposed-read 128-channel tiles with two taps per workgroup, 17 the eight-wave
// ping-pong kernel (256 x 256 tiles), 18 its row-of-taps form for the 128-channel layers.  For profilers and the benchmark's per-kernel accounting; never needed to call the operators.
extern "C" int32_t mcdseg_conv_wgrad_variant(const mcdseg_conv_desc* d, int32_t math, int32_t presplit) {
  if (d == nullptr) return -22;
  const WgradPlan pl = make_plan(d);
  if (presplit && thin_tr_applies(d, math, d, d)) return 15;
  if (presplit && math && pl.cfg == 0 && mcdseg_internal_wgrad_pp_plan(d, math, nullptr, nullptr) > 0) return 17;
  if (presplit && math && pl.cfg == 0 && mcdseg_internal_wgrad_pp3_plan(d, math, nullptr, nullptr) > 0) return 18;
  if (pl.cfg == 1 && presplit && tr64_applies(d, math, d, d, 1)) return 14;
  if (pl.cfg != 0 || math == 0) return pl.cfg;
  if (!(presplit && (d->Cin & 7) == 0 && (d->Cout & 7) == 0)) return 10;
  const int v = mcdseg_internal_wgrad_cb_variant(d, math, pl.co_p, pl.ci_p, pl.splits);
  return v == 3 ? 16 : 11 + v;  // 16: transposed-read 128-row tiles, two taps per workgroup
}

extern "C" int mcdseg_conv_wgrad(const mcdseg_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return wgrad_impl(d, x, dy, dw, workspace, workspace_bytes, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

// split arithmetic on the 128x128-tile plan (thin layers fall through to the f32 kernels); with BOTH pre-split
// companions the 128x128 plan reads them instead of the fp32 tensors
extern "C" int mcdseg_conv_split_wgrad(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                                       const float* dy, const void* dy_cb, const float* dy_bound, float* dw, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(mcd_math_known(math), "conv_split_wgrad: unknown math %d", math);
  return wgrad_impl(d, x, dy, dw, workspace, workspace_bytes, math, x_cb, x_bound, dy_cb, dy_bound, stream);
}

// One launch addresses an operand with 32-bit buffer offsets.  The kernels that read the fp32 tensors express padding and ragged
// channel tails as offsets up to a 128-channel tile past the tensor, which the buffer range check must still reject: (N*C + 128)
// planes below 2 GiB.  The plans that read both pre-split companions (variants 11..16) mark such accesses with an explicit
// out-of-range offset and check their own limits (one piece of a companion below 2 GiB): no slack.
static bool wgrad_reads_companions(const mcdseg_conv_desc* d, int math, bool have_cb) {
  if (!math || !have_cb) return false;
  if (thin_tr_applies(d, math, d, d)) return true;
  const int cfg = make_plan(d).cfg;
  return tr64_applies(d, math, d, d, cfg) || ((d->Cin & 7) == 0 && (d->Cout & 7) == 0 && cfg == 0);
}
static bool wgrad_operands_fit(const mcdseg_conv_desc* d, int math, bool have_cb) {
  const int64_t slack = wgrad_reads_companions(d, math, have_cb) ? 0 : 128;
  return ((int64_t)d->N * d->Cin + slack) * d->H * d->W * 4 < (1ll << 31) && ((int64_t)d->N * d->Cout + slack) * d->Ho * d->Wo * 4 < (1ll << 31);
}

// 1 when ONE launch of mcdseg_conv_wgrad (math = 0) / mcdseg_conv_split_wgrad can address this descriptor's operands (presplit:
// both companions are passed); the host cuts larger batches along N (mcdseg/ops.py::_batch_pieces).  Host-side arithmetic only.
extern "C" int32_t mcdseg_conv_wgrad_fits(const mcdseg_conv_desc* d, int32_t math, int32_t presplit) {
  if (d == nullptr || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0) return 0;
  return wgrad_operands_fit(d, math, presplit != 0) ? 1 : 0;
}

extern "C" size_t mcdseg_conv_wgrad_workspace_bytes(const mcdseg_conv_desc* d) {
  if (d == nullptr) return 0;
  size_t a = (size_t)make_plan(d).slab_floats * sizeof(float), c = 0;
  const size_t b = mcdseg_internal_wgrad_thin_tr_ws(d);
  if (mcdseg_internal_wgrad_pp_plan(d, MCDSEG_MATH_F16X3, nullptr, &c) > 0 && c * sizeof(float) > a) a = c * sizeof(float);
  c = 0;
  if (make_plan(d).cfg == 0 && mcdseg_internal_wgrad_pp3_plan(d, MCDSEG_MATH_F16X3, nullptr, &c) > 0 && c * sizeof(float) > a) a = c * sizeof(float);
  return a > b ? a : b;
}

static int wgrad_impl(const mcdseg_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                      int math, const void* x_cb, const float* x_bound, const void* dy_cb, const float* dy_bound, void* stream) {
  MCD_REQUIRE(d && dw && workspace, "conv_wgrad: null pointer");
  if (thin_tr_applies(d, math, x_cb, dy_cb) && x_bound && dy_bound)
    return mcdseg_internal_wgrad_thin_tr_launch(d, x_cb, x_bound, dy_cb, dy_bound, dw, workspace, workspace_bytes, (hipStream_t)stream);
  if (math && x_cb && dy_cb && x_bound && dy_bound && make_plan(d).cfg == 0) {
    size_t sf = 0;
    if (mcdseg_internal_wgrad_pp_plan(d, math, nullptr, &sf) > 0) {
      MCD_REQUIRE(workspace_bytes >= sf * sizeof(float), "conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, sf * sizeof(float));
      return mcdseg_internal_wgrad_pp_launch(d, math, x_cb, x_bound, dy_cb, dy_bound, dw, (float*)workspace, (hipStream_t)stream);
    }
    sf = 0;
    if (mcdseg_internal_wgrad_pp3_plan(d, math, nullptr, &sf) > 0) {
      MCD_REQUIRE(workspace_bytes >= sf * sizeof(float), "conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, sf * sizeof(float));
      return mcdseg_internal_wgrad_pp3_launch(d, math, x_cb, x_bound, dy_cb, dy_bound, dw, (float*)workspace, (hipStream_t)stream);
    }
  }
  const bool tr64 = tr64_applies(d, math, x_cb, dy_cb, make_plan(d).cfg);
  const bool cb_path = tr64 || (math && x_cb && dy_cb && (d->Cin & 7) == 0 && (d->Cout & 7) == 0 && make_plan(d).cfg == 0);
  const bool split_plan = tr64 || (math && make_plan(d).cfg == 0);
  MCD_REQUIRE(!(split_plan && mcd_storage_math(math) == MCDSEG_MATH_F16X3) || (x_bound && dy_bound), "conv_split_wgrad: f16x3 needs both bound scalars");
  MCD_REQUIRE(cb_path || (x && dy), "conv_wgrad: x and dy may be NULL only when the pre-split 128x128 plan applies");
  MCD_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->Ho > 0 && d->Wo > 0, "conv_wgrad: bad dims");
  MCD_REQUIRE(wgrad_operands_fit(d, math, x_cb && dy_cb),
              "conv_wgrad: activation tensor must stay below 2 GiB (32-bit buffer offsets, plus a 128-channel tile of slack for the kernels "
              "that read the fp32 operands); split the batch (mcdseg_conv_wgrad_fits)");
  const WgradPlan pl = make_plan(d);
  MCD_REQUIRE(workspace_bytes >= (size_t)pl.slab_floats * sizeof(float), "conv_wgrad: workspace too small (%zu < %zu)",
              workspace_bytes, (size_t)pl.slab_floats * sizeof(float));
  const int T = d->KH * d->KW;
  WgradParams p;
  p.x = x; p.dy = dy; p.slab = (float*)workspace;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_p = pl.co_p; p.ci_p = pl.ci_p; p.chunk = pl.chunk; p.chunks_per_img = pl.chunks_per_img; p.splits = pl.splits;
  p.x_bytes = (int)((int64_t)d->N * d->Cin * d->H * d->W * 4);
  p.dy_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 4);
  p.groups = pl.groups;
  const int64_t per_split = (int64_t)(pl.co_p / pl.bm) * (pl.ci_p / pl.bn) * pl.groups;
  const int64_t nwg = 8 * ceil_div64(pl.splits, 8) * per_split;
  MCD_REQUIRE(nwg < (1ll << 31), "conv_wgrad: grid too large");
  dim3 grid((unsigned)nwg);
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)T * d->Cout * d->Cin;
  if (pl.cfg == 3) {
    const bool rows16 = d->Cout <= 16 && pl.co_p == 32;
    if (pl.cinp == 8 && rows16)
      hipLaunchKernelGGL((conv_wgrad_thin_kernel<8, true>), grid, dim3(64), 0, st, p);
    else if (pl.cinp == 8)
      hipLaunchKernelGGL((conv_wgrad_thin_kernel<8, false>), grid, dim3(64), 0, st, p);
    else if (rows16)
      hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, true>), grid, dim3(64), 0, st, p);
    else
      hipLaunchKernelGGL((conv_wgrad_thin_kernel<16, false>), grid, dim3(64), 0, st, p);
    MCD_LAUNCH_CHECK("conv_wgrad_thin");
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3((unsigned)ceil_div64(total, 4)), dim3(256), 0, st, (const float*)workspace,
                       dw, d->Cout, d->Cin, T, pl.co_p, pl.groups, pl.cinp, pl.splits);
    MCD_LAUNCH_CHECK("conv_wgrad_thin_reduce");
    return 0;
  }
  const bool scaled = split_plan && mcd_storage_math(math) == MCDSEG_MATH_F16X3;
  if (tr64) {
    if (int rc = mcdseg_internal_wgrad_split_tr64_launch(d, math, x_cb, dy_cb, (float*)workspace, pl.co_p, pl.ci_p, pl.chunks_per_img,
                                                         pl.splits, st))
      return rc;
  } else if (cb_path) {
    if (int rc = mcdseg_internal_wgrad_split_cb_launch(d, math, x_cb, dy_cb, (float*)workspace, pl.co_p, pl.ci_p, pl.chunks_per_img,
                                                       pl.splits, st))
      return rc;
  } else if (pl.cfg == 0 && math) {
    if (int rc = mcdseg_internal_wgrad_split_launch(d, math, x, x_bound, dy, dy_bound, (float*)workspace, pl.co_p, pl.ci_p, pl.chunk,
                                                    pl.chunks_per_img, pl.splits, st))
      return rc;
  } else if (pl.cfg == 0)
    hipLaunchKernelGGL((conv_wgrad_kernel<2, 2, 2, 2, 16>), grid, dim3(256), 0, st, p);
  else if (pl.cfg == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2, 2, 32>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 1, 1, 32>), grid, dim3(64), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad");
  const int64_t pairs_all = (int64_t)d->Cout * d->Cin;
  int PB = T == 1 ? 256 : 64;
  while (PB > 16 && pairs_all / PB < 512) PB >>= 1;  // small layers: more, smaller blocks
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div64((int64_t)d->Cout * d->Cin, PB)), dim3(256),
                     (size_t)PB * T * sizeof(float), st, (const float*)workspace, dw,
                     d->Cout, d->Cin, T, pl.co_p, pl.ci_p, pl.splits, PB, scaled ? x_bound : (const float*)nullptr,
                     scaled ? dy_bound : (const float*)nullptr);
  MCD_LAUNCH_CHECK("conv_wgrad_reduce");
  return 0;
}
