// Weight gradient of the thin, full-resolution layers (Cin <= 16, Cout <= 32: the 7x7 stem, layer1, layer2) in the split
// arithmetic, from the two pre-split companions (the stem's 6-channel input: the zero-padded one of mcdseg_split_cb_padded).
//
// These layers have too few channels for the 128/64-wide tiles of conv_wgrad_split.hip: what is short is not the matrix
// rate but the operand stream -- the f32 kernels (conv_wgrad_thin_kernel) gather every input pixel once per tap group from
// global memory.  Here a workgroup stages, once per tile of TR x 32 output pixels, the input WINDOW of the tile (all taps)
// and the tile of dZ into LDS by LDS-DMA, exactly as the companions sit in memory: 16-byte units of 8 channels x 1 pixel.
// A tap is then nothing but a unit offset into the window -- always 16-byte aligned, whatever the shift -- and
// ds_read_b64_tr_b16 turns "8 channels of one pixel" into the MFMA fragment "8 pixels of one channel" for free (see
// conv_wgrad_split_tr_kernel).  The matrix instruction is v_mfma_f32_16x16x32_f16: M = 16 output channels, K = the 32
// pixels of one row segment, N = 16 columns = 16 input channels of one tap (Cin = 16) or 8 channel slots of two taps
// (Cin <= 8: lanes 4q+2, 4q+3 of a transposing read simply point at the other tap's unit).  Every lane always reads
// in-bounds LDS (the window is sized for the whole tile) and EXEC stays all ones, as the transposing read requires.
//
// Each wave owns the output rows w, w+4, .. of the tile and ALL column tiles and accumulates over the workgroup's tiles; at
// the end the four waves' accumulators are summed through LDS (fixed order) into one partial per workgroup, and a second
// kernel sums the partials in a fixed order in fp64, applies scale(x) * scale(dz) and scatters into dw[Cout][Cin][T].
#include "split.h"

namespace {

constexpr int TT_TC = 32;    // output columns of a tile = K of one matrix instruction
constexpr int TT_MAXK = 8;   // LDS-DMA slots a wave may own per piece
constexpr unsigned TT_OOB = 0x80000000u;

struct ThinTrParams {
  const void* x_cb;   // [piece 2][N][CIN8][H*W][8 x fp16]
  const void* dy_cb;  // [piece 2][N][Co8][Ho*Wo][8 x fp16]
  float* slab;        // [workgroup][MT * NTL * 4][64]
  int N, H, W, Ho, Wo, KH, KW, stride, pad, dil;
  int x_cb_bytes, dy_cb_bytes;
  int x_piece_bytes, dy_piece_bytes;
  int tiles_x, tiles_y, ntiles;
  int WR, WC;       // window rows / columns
  int uxp;          // window units per piece, padded to whole DMA slots
};

typedef short tt_s16x4 __attribute__((ext_vector_type(4)));
typedef short tt_s16x8 __attribute__((ext_vector_type(8)));

template <int CIN8, int MT, int NTL, int TR>
__global__ __launch_bounds__(256) void conv_wgrad_thin_tr_kernel(ThinTrParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tt_smem[];
  constexpr int CO8 = 2 * MT;
  constexpr int UD = CO8 * TR * TT_TC;  // dZ units per piece (a multiple of 64)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int T = p.KH * p.KW;
  const int wrc = p.WR * p.WC;
  const int ux = CIN8 * wrc;
  const int sx = p.uxp >> 6;           // window slots
  const int st = sx + (UD >> 6);       // slots per piece
  const int piece_lds = (p.uxp + UD) * 16;

  // ---- tile-invariant part of this lane's DMA addresses: slot s = wave + 4 k covers LDS units 64 s .. 64 s + 63
  int rel[TT_MAXK], rc[TT_MAXK];  // unit offset relative to the tile's base; (row << 16 | column), -1 = never valid
#pragma unroll
  for (int k = 0; k < TT_MAXK; ++k) {
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
        rel[k] = (grp * p.H + wr) * p.W + wc;
        rc[k] = (wr << 16) | wc;
      }
    } else if (s < st) {
      const int u = 64 * (s - sx) + lane;
      const int grp = u / (TR * TT_TC);
      const int rt = (u / TT_TC) % TR;
      const int c = u % TT_TC;
      rel[k] = (grp * p.Ho + rt) * p.Wo + c;
      rc[k] = (rt << 16) | c;
    }
  }
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x_cb, 0, p.x_cb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy_cb, 0, p.dy_cb_bytes, 0x00020000);

  // ---- transposing-read addresses (bytes inside a piece image): 16-lane group g = K-group, lane 4q+pp of it supplies pixel q,
  // channels 4 pp .. 4 pp + 3 -- pp >> 1 selects the second 8-channel unit (next channel group, or the second tap of a pair)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int hi_unit = pp >> 1;
  const int a_lane = ((hi_unit * TR) * TT_TC + 8 * g + q) * 16 + 8 * (pp & 1);  // + (2 mt TR + rt) * 32 * 16, + 4 h * 16
  const int b_lane = (((CIN8 == 2 ? hi_unit * wrc : 0) + (8 * g + q) * p.stride) * 16) + 8 * (pp & 1);
  // tap offsets (units), wave-uniform: column tile j is tap j (CIN8 = 2) or taps 2 j, 2 j + 1 (CIN8 = 1)
  int toff[NTL][2];
#pragma unroll
  for (int j = 0; j < NTL; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int tap = CIN8 == 2 ? j : 2 * j + e;
      if (tap > T - 1) tap = T - 1;  // past the last tap: any in-bounds address (the column is dropped by the reduce)
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      toff[j][e] = (ky * p.dil * p.WC + kx * p.dil) * 16;
    }

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tr_read = [&](int addr) -> tt_s16x4 {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tt_s16x4*)(tt_smem + addr));
#else
    (void)addr;
    return tt_s16x4{};
#endif
  };
  auto frag_at = [&](int addr, int step) -> f16x8 {  // two 4-pixel blocks `step` bytes apart -> 8 consecutive k
    const tt_s16x4 lo = tr_read(addr);
    const tt_s16x4 hi = tr_read(addr + step);
    const tt_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
  };

  const int per_img = p.tiles_x * p.tiles_y;
  // workgroup b runs on XCD b % 8: every XCD gets one contiguous band of tiles, walked in order by its gridDim.x / 8
  // workgroups, so the halo rows two neighbouring tiles share are fetched into ONE L2 (gridDim.x is a multiple of 8)
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int band_end = (xcd + 1) * per_xcd < p.ntiles ? (xcd + 1) * per_xcd : p.ntiles;
  for (int tile = xcd * per_xcd + wg_in_xcd; tile < band_end; tile += wgs_per_xcd) {
    const int n = tile / per_img;
    const int tr_ = tile - n * per_img;
    const int ty = tr_ / p.tiles_x, tx = tr_ - ty * p.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TT_TC;
    const int wy0 = oy0 * p.stride - p.pad, wx0 = ox0 * p.stride - p.pad;
    const int xbase = n * CIN8 * p.H * p.W + wy0 * p.W + wx0;
    const int dbase = n * CO8 * p.Ho * p.Wo + oy0 * p.Wo + ox0;
    // ---- stage the tile: window of x and tile of dZ, both pieces, zeros wherever the image ends
#pragma unroll
    for (int k = 0; k < TT_MAXK; ++k) {
      const int s = wave + 4 * k;
      if (s >= st) break;  // wave-uniform
      const bool isx = s < sx;
      const int r = rc[k] >> 16, c = rc[k] & 0xFFFF;
      const bool ok = rc[k] >= 0 && (isx ? ((unsigned)(wy0 + r) < (unsigned)p.H && (unsigned)(wx0 + c) < (unsigned)p.W)
                                         : (oy0 + r < p.Ho && ox0 + c < p.Wo));
      const unsigned voff = ok ? (unsigned)((isx ? xbase : dbase) + rel[k]) * 16u : TT_OOB;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        unsigned char* dst = tt_smem + pc * piece_lds + s * 1024;
        if (isx)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (__attribute__((address_space(3))) void*)dst, 16, voff, pc * p.x_piece_bytes, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(dy_rs, (__attribute__((address_space(3))) void*)dst, 16, voff, pc * p.dy_piece_bytes, 0, 0);
      }
#else
      (void)voff;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- multiply: this wave's rows of the tile, one K = 32 block per row
#pragma unroll
    for (int rr = 0; rr < (TR + 3) / 4; ++rr) {
      const int rt = wave + 4 * rr;
      if (rt >= TR) break;
      f16x8 fa[MT][2];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          fa[i][pc] = frag_at(pc * piece_lds + p.uxp * 16 + ((2 * i * TR + rt) * TT_TC) * 16 + a_lane, 4 * 16);
      const int brow = rt * p.stride * p.WC * 16 + b_lane;
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int to = CIN8 == 2 ? toff[j][0] : (hi_unit ? toff[j][1] : toff[j][0]);
        const f16x8 b0 = frag_at(brow + to, 4 * p.stride * 16);
        const f16x8 b1 = frag_at(piece_lds + brow + to, 4 * p.stride * 16);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][1], b0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], b1, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][0], b0, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // everyone is done with the stage before the next tile's DMAs overwrite it
  }

  // ---- one partial per workgroup: waves 2, 3 hand their accumulators to waves 0, 1 through LDS, then wave 1 to wave 0 (a fixed
  // order; the stage buffers are free by now -- the launcher sizes LDS for two accumulator images)
  constexpr int NACC = MT * NTL * 4;
  float* red = reinterpret_cast<float*>(tt_smem);
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int senders_from = round == 0 ? 2 : 1;  // round 0: waves 2,3 -> 0,1; round 1: wave 1 -> 0
    if (wave >= senders_from && wave < 2 * senders_from) {
      float* dst = red + (size_t)(wave - senders_from) * NACC * 64 + lane;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[((i * NTL + j) * 4 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (wave < senders_from) {
      const float* src = red + (size_t)wave * NACC * 64 + lane;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * NTL + j) * 4 + r) * 64];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  float* out = p.slab + ((size_t)blockIdx.x * NACC) * 64 + lane;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(size_t)((i * NTL + j) * 4 + r) * 64] = acc[i][j][r];
}

// raw accumulator index o = ((mt * NTL + j) * 4 + r) * 64 + lane  ->  dw[co][ci][tap]; 16 outputs x 16 partial groups per block
__global__ __launch_bounds__(256) void wgrad_thin_tr_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nraw,
                                                                   int partials, int NTL, int cin8, int Cout, int Cin, int T,
                                                                   const float* __restrict__ x_bound, const float* __restrict__ dy_bound) {
  __shared__ double sh[16][17];
  const int ol = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;
  double s = 0.0;
  if (o < nraw) {
    // eight loads in flight, added in the same order (one at a time the thread's partials / 16 loads were a chain of dependent round trips)
    int k = part;
    for (; k + 7 * 16 < partials; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slab[(size_t)(k + 16 * j) * nraw + o];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)v[j];
    }
    for (; k < partials; k += 16) s += (double)slab[(size_t)k * nraw + o];
  }
  sh[part][ol] = s;
  __syncthreads();
  if (part != 0 || o >= nraw) return;
  double t = 0.0;
  for (int k = 0; k < 16; ++k) t += sh[k][ol];
  t *= (double)mcd_scale_of_bound(*x_bound) * (double)mcd_scale_of_bound(*dy_bound);
  const int ln = o & 63, r = (o >> 6) & 3, tj = o >> 8;
  const int j = tj % NTL, mt = tj / NTL;
  const int co = 16 * mt + 4 * (ln >> 4) + r;
  const int nn = ln & 15;
  const int ci = cin8 == 2 ? nn : (nn & 7);
  const int tap = cin8 == 2 ? j : 2 * j + (nn >> 3);
  if (co < Cout && ci < Cin && tap < T) dw[((size_t)co * Cin + ci) * T + tap] = (float)t;
}

struct ThinTrPlan {
  bool ok;
  int cin8, mt, ntl, tr, WR, WC, uxp, ud, lds, tiles_x, tiles_y, ntiles, blocks, nraw;
};

ThinTrPlan thin_tr_plan(const mcdseg_conv_desc* d) {
  ThinTrPlan pl{};
  const int T = d->KH * d->KW;
  pl.ok = false;
  if (d->Ncb != 0 && d->Ncb != d->N) return pl;  // a batch slice of a larger companion: its pieces are not adjacent
  // instantiated: <2,1,9,8> (16 -> 16, 3x3), <2,2,9,4> (16 -> 32, 3x3 stride 2), <1,1,25,8> (<= 8 -> 16, 7x7: the stem, whose
  // input companion is the padded one of mcdseg_split_cb_padded)
  if (d->Cin == 16 && T == 9 && d->Cout == 16 && d->stride == 1) {
    pl.cin8 = 2; pl.mt = 1; pl.ntl = 9; pl.tr = 8;
  } else if (d->Cin == 16 && T == 9 && d->Cout == 32 && d->stride == 2) {
    pl.cin8 = 2; pl.mt = 2; pl.ntl = 9; pl.tr = 4;
  } else if (d->Cin <= 8 && T == 49 && d->Cout == 16 && d->stride == 1) {
    pl.cin8 = 1; pl.mt = 1; pl.ntl = 25; pl.tr = 8;
  } else {
    return pl;
  }
  pl.WR = (pl.tr - 1) * d->stride + (d->KH - 1) * d->dil + 1;
  pl.WC = (TT_TC - 1) * d->stride + (d->KW - 1) * d->dil + 1;
  pl.uxp = round_up(pl.cin8 * pl.WR * pl.WC, 64);
  pl.ud = 2 * pl.mt * pl.tr * TT_TC;
  pl.lds = 2 * (pl.uxp + pl.ud) * 16;
  const int red_bytes = 2 * pl.mt * pl.ntl * 4 * 64 * 4;  // two accumulator images for the cross-wave sum at the end
  if (pl.lds < red_bytes) pl.lds = red_bytes;
  const int slots = (pl.uxp + pl.ud) / 64;
  if (pl.lds > 64 * 1024 || slots > 4 * TT_MAXK || pl.WR >= 32768 || pl.WC >= 65536) return pl;
  pl.tiles_x = ceil_div(d->Wo, TT_TC);
  pl.tiles_y = ceil_div(d->Ho, pl.tr);
  const int64_t nt = (int64_t)d->N * pl.tiles_x * pl.tiles_y;
  if (nt >= (1ll << 31)) return pl;
  pl.ntiles = (int)nt;
  // persistent workgroups: exactly as many as are resident at once (256 CUs x 2 / 4 / 3 by registers and LDS -- a third
  // workgroup per CU of the 216-register stem kernel would only start when another has finished); a multiple of 8: one band
  // of tiles per XCD
  const int resident = 256 * (pl.cin8 == 1 ? 2 : (pl.mt == 1 ? 4 : 3));
  pl.blocks = pl.ntiles < resident ? round_up(pl.ntiles, 8) : resident;
  pl.nraw = pl.mt * pl.ntl * 256;
  const int64_t xb = 2ll * d->N * (8 * pl.cin8) * d->H * d->W * 2, yb = 2ll * d->N * d->Cout * d->Ho * d->Wo * 2;
  if (xb + 4096 >= (1ll << 31) || yb + 4096 >= (1ll << 31)) return pl;
  pl.ok = true;
  return pl;
}

}  // namespace

int mcdseg_internal_wgrad_thin_tr_ok(const mcdseg_conv_desc* d) { return thin_tr_plan(d).ok ? 1 : 0; }

// the instantiation conv_wgrad_thin_tr_kernel<cin8, mt, ntl, tr> this geometry runs on, as cin8 * 1000000 + mt * 10000 + ntl * 100 + tr
// (0: the kernel does not take it) -- kernel names for profilers
extern "C" int32_t mcdseg_conv_wgrad_thin_tr_config(const mcdseg_conv_desc* d) {
  if (d == nullptr) return 0;
  const ThinTrPlan pl = thin_tr_plan(d);
  return pl.ok ? pl.cin8 * 1000000 + pl.mt * 10000 + pl.ntl * 100 + pl.tr : 0;
}

size_t mcdseg_internal_wgrad_thin_tr_ws(const mcdseg_conv_desc* d) {
  const ThinTrPlan pl = thin_tr_plan(d);
  return pl.ok ? (size_t)pl.blocks * pl.nraw * sizeof(float) : 0;
}

int mcdseg_internal_wgrad_thin_tr_launch(const mcdseg_conv_desc* d, const void* x_cb, const float* x_bound, const void* dy_cb,
                                         const float* dy_bound, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
  const ThinTrPlan pl = thin_tr_plan(d);
  MCD_REQUIRE(pl.ok && x_cb && dy_cb && x_bound && dy_bound && dw && ws, "conv_wgrad_thin_tr: bad arguments");
  MCD_REQUIRE(ws_bytes >= mcdseg_internal_wgrad_thin_tr_ws(d), "conv_wgrad_thin_tr: workspace too small");
  ThinTrParams p;
  p.x_cb = x_cb; p.dy_cb = dy_cb; p.slab = (float*)ws;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.x_piece_bytes = (int)((int64_t)d->N * (8 * pl.cin8) * d->H * d->W * 2);
  p.dy_piece_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 2);
  p.x_cb_bytes = 2 * p.x_piece_bytes;
  p.dy_cb_bytes = 2 * p.dy_piece_bytes;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.WR = pl.WR; p.WC = pl.WC; p.uxp = pl.uxp;
  if (pl.cin8 == 1)
    hipLaunchKernelGGL((conv_wgrad_thin_tr_kernel<1, 1, 25, 8>), dim3(pl.blocks), dim3(256), pl.lds, st, p);
  else if (pl.mt == 1)
    hipLaunchKernelGGL((conv_wgrad_thin_tr_kernel<2, 1, 9, 8>), dim3(pl.blocks), dim3(256), pl.lds, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_thin_tr_kernel<2, 2, 9, 4>), dim3(pl.blocks), dim3(256), pl.lds, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_thin_tr");
  hipLaunchKernelGGL(wgrad_thin_tr_reduce_kernel, dim3(ceil_div(pl.nraw, 16)), dim3(256), 0, st, (const float*)ws, dw, pl.nraw,
                     pl.blocks, pl.ntl, pl.cin8, d->Cout, d->Cin, d->KH * d->KW, x_bound, dy_bound);
  MCD_LAUNCH_CHECK("conv_wgrad_thin_tr_reduce");
  return 0;
}
