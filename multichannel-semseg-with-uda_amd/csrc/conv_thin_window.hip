// Forward and data gradient of the thin full-resolution stride-1 3x3 layers (16 contraction channels, 16 or 32 output channels:
// layer1 16 -> 16 of the DRN trunk) in the split arithmetic, from the pre-split companion.
//
// The implicit-GEMM kernel (conv_gemm_split_kernel<..,1,2,1,4,..>) fetches the pixel operand once per tap from L2 -- nine
// tap-shifted copies of a tensor that is the whole HBM traffic of the layer -- and runs at 2.4-3.4x the layer's HBM floor.  Here
// a workgroup stages the input WINDOW of a TR x 32 output tile once (LDS-DMA, the companion's own 16-byte units of 8 channels
// x 1 pixel, zeros outside the image: the staging of conv_wgrad_thin_tr.hip) and every tap is a unit offset into it.
// v_mfma_f32_16x16x32_f16 with M = 16 output channels, N = 16 consecutive pixels of a row, K = 2 taps x 16 channels: the B
// fragment of lane (pixel n, k-group g) is ONE unit -- channels 8 (g & 1) .. +7 of tap 2 ks + (g >> 1) at pixel n -- i.e. one
// aligned ds_read_b128, no transposition; the A fragments (weights, 5 K-steps) come straight from the packed image of
// mcdseg_conv_split_pack_weights and stay in registers for the workgroup's life.  Three cross terms, accumulators scaled by
// scale(x) * scale(w); the forward form also emits the BatchNorm partial statistics (count, mean, M2 per channel: one row per
// workgroup and wave, kept as running values over the workgroup's tiles) that mcdseg_bn_stats_finalize merges.  The output
// goes through a per-wave LDS image so that a store instruction covers whole 128-byte rows.
#include "split.h"

namespace {

constexpr int TW_TC = 32;
constexpr int TW_MAXK = 8;
// K-steps: K = 32 = two taps x 16 channels (CIN8 = 2; 5 steps for a 3x3 kernel, the tenth tap is zero weights) or four taps x one
// 8-channel unit (CIN8 = 1: the 7x7 stem on its zero-padded companion; 13 steps)
constexpr unsigned TW_OOB = 0x80000000u;

struct ThinWinParams {
  const void* src_cb;  // [piece 2][N][CIN8][Hs*Ws][8 x fp16]
  const void* wp;      // packed weight image [tap][piece][half][Mp 32][8]
  const float* src_bound;
  const float* w_bound;
  float* dst;          // [N][M][Hd][Wd]
  float* stats;        // forward only, may be NULL: [(workgroup * 4 + wave)][3][32]
  int N, Hs, Ws, Hd, Wd, M, KH, KW, stride, pad, dil;
  int src_bytes, src_piece_bytes, wp_bytes;
  int tiles_x, tiles_y, ntiles;
  int WR, WC, uxp;
};

template <int CIN8, int MT, int TW_KS, bool DGRAD>
__global__ __launch_bounds__(256) void conv_thin_window_kernel(ThinWinParams p) {
  constexpr int TR = 8;  // a wave owns two rows of the tile (the output stage below relies on it)
  extern __shared__ __attribute__((aligned(16))) unsigned char tw_smem[];
  constexpr int GR = (TR + 3) / 4;  // rows of the tile a wave owns
  constexpr int NG = 2 * GR;        // 16-pixel groups a wave owns
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = p.KH * p.KW;
  const int wrc = p.WR * p.WC;
  const int ux = CIN8 * wrc;
  const int sx = p.uxp >> 6;
  const int piece_lds = p.uxp * 16;
  const int n16 = lane & 15, g = lane >> 4;

  // ---- tile-invariant part of the window staging (see conv_wgrad_thin_tr.hip)
  int rel[TW_MAXK], rc[TW_MAXK];
#pragma unroll
  for (int k = 0; k < TW_MAXK; ++k) {
    const int s = wave + 4 * k;
    rel[k] = 0;
    rc[k] = -1;
    if (s < sx) {
      const int u = 64 * s + lane;
      if (u < ux) {
        const int grp = u / wrc;
        const int rem = u - grp * wrc;
        const int wr = rem / p.WC;
        const int wc = rem - wr * p.WC;
        rel[k] = (grp * p.Hs + wr) * p.Ws + wc;
        rc[k] = (wr << 16) | wc;
      }
    }
  }
  const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src_cb, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.wp_bytes, 0x00020000);

  // ---- weights: A fragment of lane (row m = n16, k-group g) for K-step ks = 8 channels (half g & 1) of tap 2 ks + (g >> 1), or
  // (CIN8 = 1) the one 8-channel unit of tap 4 ks + g
  const int half = CIN8 == 2 ? (g & 1) : 0;
  f16x8 fa[MT][TW_KS][2];
  int toff[TW_KS];  // this lane's window offset (bytes) of its tap in each K-step
#pragma unroll
  for (int ks = 0; ks < TW_KS; ++ks) {
    const int tap = CIN8 == 2 ? 2 * ks + (g >> 1) : 4 * ks + g;
    const bool tv = tap < T;
    const int tc = tv ? tap : 0;
    const int ky = tc / p.KW, kx = tc - ky * p.KW;
    const int oy = DGRAD ? (p.KH - 1 - ky) * p.dil : ky * p.dil;
    const int ox = DGRAD ? (p.KW - 1 - kx) * p.dil : kx * p.dil;
    toff[ks] = (oy * p.WC + ox) * 16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const unsigned voff = tv ? (unsigned)((((tap * 2 + pc) * 2 + half) * 32 + 16 * i + n16) * 16) : TW_OOB;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(wp_rs, voff, 0, 0);
        fa[i][ks][pc] = __builtin_bit_cast(f16x8, q);
      }
  }
  const float osc = mcd_scale_of_bound(*p.src_bound) * mcd_scale_of_bound(*p.w_bound);
  const int b_lane = (half * wrc + n16 * p.stride) * 16;

  // BatchNorm partial statistics: every lane keeps a running (count, mean, M2) of the pixels it has produced for each of its
  // channels (Welford's update, shifted form as everywhere in this library); merged over the 16 pixel lanes of a channel group and
  // written ONCE per wave at the end -- one row per (workgroup, wave) instead of one per (tile, wave)
  float st_n = 0.f, st_mean[MT][4], st_m2[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) st_mean[i][r] = st_m2[i][r] = 0.f;

  const int per_img = p.tiles_x * p.tiles_y;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int band_end = (xcd + 1) * per_xcd < p.ntiles ? (xcd + 1) * per_xcd : p.ntiles;
  for (int tile = xcd * per_xcd + wg_in_xcd; tile < band_end; tile += wgs_per_xcd) {
    const int n = tile / per_img;
    const int tr_ = tile - n * per_img;
    const int ty = tr_ / p.tiles_x, tx = tr_ - ty * p.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TW_TC;
    const int wy0 = DGRAD ? oy0 + p.pad - (p.KH - 1) * p.dil : oy0 * p.stride - p.pad;
    const int wx0 = DGRAD ? ox0 + p.pad - (p.KW - 1) * p.dil : ox0 * p.stride - p.pad;
    const int xbase = n * CIN8 * p.Hs * p.Ws + wy0 * p.Ws + wx0;
#pragma unroll
    for (int k = 0; k < TW_MAXK; ++k) {
      const int s = wave + 4 * k;
      if (s >= sx) break;  // wave-uniform
      const int r = rc[k] >> 16, c = rc[k] & 0xFFFF;
      const bool ok = rc[k] >= 0 && (unsigned)(wy0 + r) < (unsigned)p.Hs && (unsigned)(wx0 + c) < (unsigned)p.Ws;
      const unsigned voff = ok ? (unsigned)(xbase + rel[k]) * 16u : TW_OOB;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rs, (__attribute__((address_space(3))) void*)(tw_smem + pc * piece_lds + s * 1024),
                                                 16, voff, pc * p.src_piece_bytes, 0, 0);
#else
      (void)voff;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[NG][MT];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int rt = wave + 4 * (gi >> 1);
      const int c0 = 16 * (gi & 1);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[gi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rt < TR) {  // wave-uniform
        const int base = (rt * p.stride * p.WC + c0 * p.stride) * 16 + b_lane;
#pragma unroll
        for (int ks = 0; ks < TW_KS; ++ks) {
          const f16x8 b0 = *reinterpret_cast<const f16x8*>(tw_smem + base + toff[ks]);
          const f16x8 b1 = *reinterpret_cast<const f16x8*>(tw_smem + piece_lds + base + toff[ks]);
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            acc[gi][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][ks][1], b0, acc[gi][i], 0, 0, 0);
            acc[gi][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][ks][0], b1, acc[gi][i], 0, 0, 0);
            acc[gi][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][ks][0], b0, acc[gi][i], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue: D[row = channel 4 g + r][column = pixel n16]; scale, store, BatchNorm partial statistics
    bool pv[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int rt = wave + 4 * (gi >> 1);
      const int oy = oy0 + rt, ox = ox0 + 16 * (gi & 1) + n16;
      const bool rowv = rt < TR && oy < p.Hd;
      pv[gi] = rowv && ox < p.Wd;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[gi][i][r] *= osc;
    }
    // ---- output through LDS: the 16x16 accumulator layout would store 64-byte runs (16 pixels of one channel per instruction
    // and lane group), which cost as much as the rest of the kernel; transposed through a per-wave [channel][32 * GR pixels]
    // image a store instruction covers GR whole 128-byte rows instead
    {
      constexpr int OP = 32 * GR + 1;  // padded pixel stride (floats)
      float* ot = reinterpret_cast<float*>(tw_smem + 2 * piece_lds) + wave * (16 * MT * OP);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) ot[(16 * i + 4 * g + r) * OP + 32 * (gi >> 1) + 16 * (gi & 1) + n16] = acc[gi][i][r];
      // the image is private to the wave: its own LDS writes only have to land
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int rsel = lane >> 5, pcol = lane & 31;  // lanes 0-31: the wave's first row, 32-63: its second
      const int rt = wave + 4 * rsel;
      const int oy = oy0 + rt, ox = ox0 + pcol;
      const bool ok = oy < p.Hd && ox < p.Wd;
#pragma unroll
      for (int co = 0; co < 16 * MT; ++co) {
        const float v = ot[co * OP + 32 * rsel + pcol];
        if (ok && co < p.M) p.dst[(((size_t)n * p.M + co) * p.Hd + oy) * p.Wd + ox] = v;
      }
    }
    if (!DGRAD && p.stats != nullptr) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi)
        if (pv[gi]) {
          st_n += 1.f;
          const float inv = 1.f / st_n;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float dlt = acc[gi][i][r] - st_mean[i][r];
              st_mean[i][r] = fmaf(dlt, inv, st_mean[i][r]);
              st_m2[i][r] = fmaf(dlt, acc[gi][i][r] - st_mean[i][r], st_m2[i][r]);
            }
        }
    }
    __syncthreads();  // the stage is free for the next tile's DMAs
  }
  if (!DGRAD && p.stats != nullptr) {
    // Chan's merge over the 16 lanes that hold the same channels (xor 1, 2, 4, 8 stay inside a 16-lane group)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float nb = __shfl_xor(st_n, o);
      const float nt = st_n + nb;
      const float wgt = nt > 0.f ? nb / nt : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mb = __shfl_xor(st_mean[i][r], o), qb = __shfl_xor(st_m2[i][r], o);
          const float dlt = mb - st_mean[i][r];
          st_m2[i][r] = st_m2[i][r] + qb + dlt * dlt * st_n * wgt;
          st_mean[i][r] = fmaf(dlt, wgt, st_mean[i][r]);
        }
      st_n = nt;
    }
    float* row = p.stats + (size_t)(blockIdx.x * 4 + wave) * 3 * 32;
    if (n16 == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * i + 4 * g + r;
          row[co] = st_n;
          row[32 + co] = st_mean[i][r];
          row[64 + co] = st_m2[i][r];
        }
    }
    if (MT == 1 && lane < 16) {  // channels 16..31 of the 32-wide row do not exist: empty entries
      row[16 + lane] = 0.f;
      row[32 + 16 + lane] = 0.f;
      row[64 + 16 + lane] = 0.f;
    }
  }
}

struct ThinWinPlan {
  bool ok;
  int cin8, mt, tr, WR, WC, uxp, lds, tiles_x, tiles_y, ntiles, blocks;
};

ThinWinPlan thin_win_plan(const mcdseg_conv_desc* d, bool dgrad) {
  ThinWinPlan pl{};
  pl.ok = false;
  if (d->Ncb != 0 && d->Ncb != d->N) return pl;  // a batch slice of a larger companion: its pieces are not adjacent (implicit GEMM handles it)
  const bool on = mcd_opt(MCD_OPT_THIN_WINDOW) != 0;  // development knob: 0 = the implicit-GEMM kernels for these layers too
  if (!on || d->stride != 1) return pl;  // (a stride-2 form, 16 -> 32, measured no faster than the implicit GEMM: 0.134 vs 0.131 ms)
  const int T = d->KH * d->KW;
  const int ks = dgrad ? d->Cout : d->Cin, m = dgrad ? d->Cin : d->Cout;
  if (ks == 16 && T == 9 && (m == 16 || (m == 32 && !dgrad))) {
    pl.cin8 = 2;  // instantiated: <2,1,5,fwd>, <2,2,5,fwd>, <2,1,5,dgrad>
  } else if (!dgrad && ks <= 8 && T == 49 && m == 16) {
    pl.cin8 = 1;  // <1,1,13,fwd>: the stem on the zero-padded companion of the network input
  } else {
    return pl;
  }
  pl.mt = m / 16;
  pl.tr = 8;
  pl.WR = (pl.tr - 1) * d->stride + (d->KH - 1) * d->dil + 1;
  pl.WC = (TW_TC - 1) * d->stride + (d->KW - 1) * d->dil + 1;
  pl.uxp = round_up(pl.cin8 * pl.WR * pl.WC, 64);
  pl.lds = 2 * pl.uxp * 16 + 4 * 16 * pl.mt * (32 * ((pl.tr + 3) / 4) + 1) * 4;  // window (two pieces) + the waves' output images
  if (pl.lds > 64 * 1024 || pl.uxp / 64 > 4 * TW_MAXK || pl.WC >= 65536) return pl;
  const int hd = dgrad ? d->H : d->Ho, wd = dgrad ? d->W : d->Wo;
  pl.tiles_x = ceil_div(wd, TW_TC);
  pl.tiles_y = ceil_div(hd, pl.tr);
  const int64_t nt = (int64_t)d->N * pl.tiles_x * pl.tiles_y;
  if (nt * 4 >= (1ll << 31)) return pl;
  pl.ntiles = (int)nt;
  const int resident = 256 * ((pl.mt == 2 || pl.cin8 == 1) ? 2 : 4);  // by registers (196 / ~200 vs 124 VGPRs) and LDS
  pl.blocks = pl.ntiles < resident ? round_up(pl.ntiles, 8) : resident;
  const int hs = dgrad ? d->Ho : d->H, ws = dgrad ? d->Wo : d->W;
  if (2ll * d->N * (8 * pl.cin8) * hs * ws * 2 + 4096 >= (1ll << 31) || (int64_t)d->N * m * hd * wd * 4 >= (1ll << 31)) return pl;
  pl.ok = true;
  return pl;
}

}  // namespace

int mcdseg_internal_thin_window_ok(const mcdseg_conv_desc* d, int dgrad) { return thin_win_plan(d, dgrad != 0).ok ? 1 : 0; }

int64_t mcdseg_internal_thin_window_stat_rows(const mcdseg_conv_desc* d) {
  const ThinWinPlan pl = thin_win_plan(d, false);
  return pl.ok ? (int64_t)pl.blocks * 4 : 0;
}

int mcdseg_internal_thin_window_launch(const mcdseg_conv_desc* d, int dgrad, const void* src_cb, const float* src_bound, const void* wp,
                                       int64_t wp_bytes, const float* w_bound, float* dst, float* stats, hipStream_t st) {
  const ThinWinPlan pl = thin_win_plan(d, dgrad != 0);
  MCD_REQUIRE(pl.ok && src_cb && src_bound && wp && w_bound && dst, "conv_thin_window: bad arguments");
  ThinWinParams p;
  p.src_cb = src_cb; p.wp = wp; p.src_bound = src_bound; p.w_bound = w_bound; p.dst = dst; p.stats = dgrad ? nullptr : stats;
  p.N = d->N;
  p.Hs = dgrad ? d->Ho : d->H; p.Ws = dgrad ? d->Wo : d->W;
  p.Hd = dgrad ? d->H : d->Ho; p.Wd = dgrad ? d->W : d->Wo;
  p.M = dgrad ? d->Cin : d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.src_piece_bytes = (int)((int64_t)d->N * (8 * pl.cin8) * p.Hs * p.Ws * 2);
  p.src_bytes = 2 * p.src_piece_bytes;
  p.wp_bytes = (int)wp_bytes;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.WR = pl.WR; p.WC = pl.WC; p.uxp = pl.uxp;
  const dim3 grid(pl.blocks), block(256);
  if (dgrad)
    hipLaunchKernelGGL((conv_thin_window_kernel<2, 1, 5, true>), grid, block, pl.lds, st, p);
  else if (pl.cin8 == 1)
    hipLaunchKernelGGL((conv_thin_window_kernel<1, 1, 13, false>), grid, block, pl.lds, st, p);
  else if (pl.mt == 1)
    hipLaunchKernelGGL((conv_thin_window_kernel<2, 1, 5, false>), grid, block, pl.lds, st, p);
  else
    hipLaunchKernelGGL((conv_thin_window_kernel<2, 2, 5, false>), grid, block, pl.lds, st, p);
  MCD_LAUNCH_CHECK("conv_thin_window");
  return 0;
}
