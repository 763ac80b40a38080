// Implicit-GEMM convolution (forward and data-gradient) on the f32-input MFMA of gfx950.
//
//   D[m][p] = sum_{tap, c} Wp[tap][c][m] * S[c][src(p, tap)]
//
// forward : m = output channel, p = output pixel (n,oy,ox), S = x, src = (oy*s + ky*d - pad, ox*s + kx*d - pad)
// dgrad   : m = input channel,  p = input pixel (n,y,x),    S = dy, src = ((y + pad - ky*d)/s, (x + pad - kx*d)/s)
//
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 FMA chain (no TF32 on gfx950), so results stay
// within fp32 rounding of the reference's oneDNN/cuDNN path.  One workgroup = 4 waves = a BM x BN
// tile of D; K runs over (tap, channel-chunk of BK).  Per K-step the weight slab Wp[tap][c0..c0+BK)[BM]
// arrives by coalesced float4 loads, the activation slab is gathered pixel-contiguous (each wave
// instruction reads 64 consecutive pixels of one channel = 256 B of NCHW), both go through a
// double-buffered LDS image that the MFMA fragments read conflict-free (32 consecutive floats per
// half-wave).  Global loads of step s+1 are issued before the MFMAs of step s.
//
// Forward epilogue (fused BatchNorm statistics): every wave reduces its 32*WN pixel columns per channel
// with wave shuffles into (count, mean, M2) -- a shifted/Welford form, not raw sum-of-squares -- and
// writes one partial row; mcdseg_bn_stats_finalize merges the rows in fp64.
#include "common.h"

namespace {

struct ConvGemmParams {
  const float* src;
  const float* wp;
  const float* bias;
  float* dst;
  float* stats;
  const float* ep_scale;  // inference epilogue: y = act(scale[m]*acc + shift[m] + residual)
  const float* ep_shift;
  const float* ep_res;
  int ep_relu;
  int N;
  int Cs, Hs, Ws;  // source tensor (gathered)
  int M, Hd, Wd;   // destination tensor
  int Mp, Kp;
  int KH, KW, stride, pad, dil;
  int P;           // N*Hd*Wd
  int src_bytes, wp_bytes;  // buffer-descriptor ranges (< 2 GiB)
};

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool DGRAD>
__global__ __launch_bounds__(256, 3) void conv_gemm_kernel(ConvGemmParams p) {
  constexpr int BM = 32 * WM * WAVES_M;
  constexpr int BN = 32 * WN * WAVES_N;
  constexpr int NT = 256;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int RPP = NT / BN;                  // activation rows gathered per pass
  constexpr int B_ITERS = BK / RPP;
  constexpr int A_VECS = BK * BM / 4;           // float4 per K-step
  constexpr int A_ITERS = (A_VECS + NT - 1) / NT;
  constexpr bool A_EXACT = (A_VECS % NT) == 0;  // no tail guard needed
  static_assert(BK % RPP == 0, "BK vs BN");

  __shared__ float smem[2 * BK * (BM + BN)];
  float* As = smem;                 // [2][BK][BM]
  float* Bs = smem + 2 * BK * BM;   // [2][BK][BN]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so
  // id -> (xcd = b % 8, slot = b / 8).  Inside one XCD consecutive slots walk the M tiles of one pixel tile
  // first (they gather the same activations: one HBM/MALL fetch, then L2 hits), and every XCD owns a contiguous
  // range of pixel tiles (neighbouring tiles share their halo rows).  Placement only affects speed.
  const int m_tiles = p.Mp / BM;
  const int n_tiles = (p.P + BN - 1) / BN;
  const int per_xcd = (n_tiles + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int tile_m = slot % m_tiles;
  const int tile_n = xcd * per_xcd + slot / m_tiles;
  if (tile_n >= n_tiles) return;

  // ---- this thread's gather pixel
  const int bj = t % BN;
  const int brg = t / BN;
  const int HWd = p.Hd * p.Wd;
  const int HWs = p.Hs * p.Ws;
  const int pix = tile_n * BN + bj;
  const bool pv = pix < p.P;
  int pn = 0, py = 0, px = 0;
  if (pv) {
    pn = pix / HWd;
    const int rem = pix - pn * HWd;
    py = rem / p.Wd;
    px = rem - py * p.Wd;
  }
  // Gathers go through buffer loads: the 128-bit descriptor and the per-load channel offset live in SGPRs, the
  // per-lane part is ONE 32-bit byte offset recomputed per tap, and a lane whose tap falls into the zero padding
  // (or past the last pixel) gets an offset beyond num_records -- the hardware range check returns 0 for it.
  // No exec-mask juggling, no 64-bit address arithmetic in the K loop.
  constexpr unsigned OOB = 0x80000000u;  // tensors are < 2 GiB (checked on the host), so this is out of range
  const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, p.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.wp_bytes, 0x00020000);
  const unsigned pix_base = (unsigned)pn * (unsigned)p.Cs * (unsigned)HWs;
  const int brg_u = __builtin_amdgcn_readfirstlane(brg);  // wave-uniform (BN >= 64)
  const bool ragged = p.Kp != p.Cs;

  // ---- K-step state of the loader (one step ahead of the MFMAs)
  int l_tap = 0, l_c0 = 0, l_ky = 0, l_kx = 0;
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
    l_voff = ok ? (pix_base + (unsigned)off) * 4u : OOB;
  };
  tap_geom();

  // weight slab: per-thread constant offsets inside the [BK][BM] slab, per-step uniform base in soffset
  unsigned a_voff[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int idx = t + i * NT;
    const int k = idx / (BM / 4);
    const int m4 = idx - k * (BM / 4);
    a_voff[i] = (A_EXACT || idx < A_VECS) ? ((unsigned)k * (unsigned)p.Mp + (unsigned)m4 * 4u) * 4u : OOB;
  }

  float breg[B_ITERS];
  f32x4 areg[A_ITERS];
  auto load_regs = [&]() {
    const int c_base = l_c0 + brg_u;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int c = c_base + i * RPP;
      // channels of the zero-padded K tail: a scalar offset past every record (voffset + soffset never wraps)
      const int soff = (!ragged || c < p.Cs) ? c * HWs * 4 : 0x7FFFFFFF;
      breg[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src_rs, l_voff, soff, 0));
    }
    const int a_soff = ((l_tap * p.Kp + l_c0) * p.Mp + tile_m * BM) * 4;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const auto q = __builtin_amdgcn_raw_buffer_load_b128(wp_rs, a_voff[i], a_soff, 0);
      areg[i][0] = __uint_as_float(q[0]);
      areg[i][1] = __uint_as_float(q[1]);
      areg[i][2] = __uint_as_float(q[2]);
      areg[i][3] = __uint_as_float(q[3]);
    }
  };
  auto store_lds = [&](int buf) {
    float* bdst = Bs + buf * BK * BN;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) bdst[(brg + i * RPP) * BN + bj] = breg[i];
    float* adst = As + buf * BK * BM;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int idx = t + i * NT;
      if (A_EXACT || idx < A_VECS) *reinterpret_cast<f32x4*>(adst + idx * 4) = areg[i];
    }
  };
  // K order: channel chunk OUTER, tap INNER -- the KH*KW shifted re-reads of one 16-channel activation slab
  // are back to back, so all but the first come from L1/L2 instead of MALL/HBM.
  const int taps = p.KH * p.KW;
  auto advance = [&]() {
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
      l_c0 += BK;
    }
    tap_geom();
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = taps * (p.Kp / BK);
  load_regs();
  store_lds(0);
  __syncthreads();

  const int l31 = lane & 31;
  const int lh = lane >> 5;
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) {
      advance();
      load_regs();
    }
    const float* a_base = As + cur * BK * BM + wm * (32 * WM) + l31;
    const float* b_base = Bs + cur * BK * BN + wn * (32 * WN) + l31;
    // fragments of k-pair ks+1 are read from LDS while the MFMAs of k-pair ks run
    float a[2][WM], b[2][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[0][i] = a_base[lh * BM + i * 32];
#pragma unroll
    for (int j = 0; j < WN; ++j) b[0][j] = b_base[lh * BN + j * 32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const int cb = ks & 1;
      if (ks + 1 < BK / 2) {
        const int kr = 2 * (ks + 1) + lh;
#pragma unroll
        for (int i = 0; i < WM; ++i) a[cb ^ 1][i] = a_base[kr * BM + i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[cb ^ 1][j] = b_base[kr * BN + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of the NEXT k-pair ahead of these MFMAs
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

  // ---- epilogue.  acc[i][j][r] = D[row][col], row = (r&3) + 8*(r>>2) + 4*lh, col = l31 of the 32x32 tile
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
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int pp = p_wave + j * 32 + l31;
    colv[j] = pp < p.P;
    int n = 0, rem = 0;
    if (colv[j]) {
      n = pp / HWd;
      rem = pp - n * HWd;
    }
    dbase[j] = (size_t)n * p.M * HWd + rem;
  }
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
            p.dst[dbase[j] + (size_t)m * HWd] = v;
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

// ---- weight packing: w[Cout][Cin][T] -> fprop image [T][Kp_f][Mp_f] (m = cout, k = cin) and
//                                         dgrad image [T][Kp_d][Mp_d] (m = cin,  k = cout)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Cout,
                                    int Cin, int T, int Mp_f, int Kp_f, int Mp_d, int Kp_d) {
  const int64_t nf = (int64_t)T * Kp_f * Mp_f;
  const int64_t nd = (int64_t)T * Kp_d * Mp_d;
  const int64_t total = (wf ? nf : 0) + (wd ? nd : 0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (wf && i < nf) {
      const int m = (int)(i % Mp_f);
      const int64_t r = i / Mp_f;
      const int k = (int)(r % Kp_f);
      const int tap = (int)(r / Kp_f);
      wf[i] = (m < Cout && k < Cin) ? w[((int64_t)m * Cin + k) * T + tap] : 0.f;
    } else {
      const int64_t ii = i - (wf ? nf : 0);
      const int m = (int)(ii % Mp_d);
      const int64_t r = ii / Mp_d;
      const int k = (int)(r % Kp_d);
      const int tap = (int)(r / Kp_d);
      wd[ii] = (m < Cin && k < Cout) ? w[((int64_t)k * Cin + m) * T + tap] : 0.f;
    }
  }
}

int check_desc(const mcdseg_conv_desc* d, const char* who) {
  MCD_REQUIRE(d != nullptr, "%s: null descriptor", who);
  MCD_REQUIRE(d->N > 0 && d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0, "%s: non-positive dims", who);
  MCD_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "%s: bad kernel geometry", who);
  const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  MCD_REQUIRE(ho == d->Ho && wo == d->Wo, "%s: Ho/Wo (%d,%d) do not match geometry (%d,%d)", who, d->Ho, d->Wo, ho, wo);
  MCD_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31),
              "%s: pixel count exceeds int32", who);
  MCD_REQUIRE((int64_t)d->N * d->Cin * d->H * d->W * 4 < (1ll << 31) && (int64_t)d->N * d->Cout * d->Ho * d->Wo * 4 < (1ll << 31),
              "%s: activation tensor must stay below 2 GiB (32-bit buffer offsets); split the batch", who);
  return 0;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, bool DGRAD>
void launch_cfg(const ConvGemmParams& p, int bk, hipStream_t st) {
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  const int n_tiles = ceil_div(p.P, BN);
  dim3 grid(8 * ceil_div(n_tiles, 8) * (p.Mp / BM));
  if (bk == 8)
    hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, WAVES_M, WAVES_N, 8, DGRAD>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, WAVES_M, WAVES_N, 16, DGRAD>), grid, dim3(256), 0, st, p);
}

template <bool DGRAD>
void launch(const ConvGemmParams& p, hipStream_t st) {
  const int bm = mcd_bm(p.M);
  const int bk = mcd_bk(DGRAD ? p.Cs : p.Cs);
  if (bm == 128)
    launch_cfg<2, 2, 2, 2, DGRAD>(p, bk, st);
  else if (bm == 64)
    launch_cfg<2, 2, 1, 4, DGRAD>(p, bk, st);
  else
    launch_cfg<1, 2, 1, 4, DGRAD>(p, bk, st);
}

// number of pixel columns per workgroup for a given M
int bn_for(int M) { return mcd_bm(M) == 128 ? 128 : 256; }
int waves_n_for(int M) { return mcd_bm(M) == 128 ? 2 : 4; }

}  // namespace

extern "C" int mcdseg_conv_packed_dims(const mcdseg_conv_desc* d, int32_t* Mp_f, int32_t* Kp_f, int32_t* Mp_d, int32_t* Kp_d) {
  MCD_REQUIRE(d != nullptr, "conv_packed_dims: null descriptor");
  if (Mp_f) *Mp_f = mcd_mp(d->Cout);
  if (Kp_f) *Kp_f = mcd_kp(d->Cin);
  if (Mp_d) *Mp_d = mcd_mp(d->Cin);
  if (Kp_d) *Kp_d = mcd_kp(d->Cout);
  return 0;
}

extern "C" int mcdseg_conv_pack_weights(const mcdseg_conv_desc* d, const float* w, float* wp_fprop, float* wp_dgrad, void* stream) {
  if (int rc = check_desc(d, "conv_pack_weights")) return rc;
  MCD_REQUIRE(w != nullptr && (wp_fprop != nullptr || wp_dgrad != nullptr), "conv_pack_weights: null pointer");
  const int T = d->KH * d->KW;
  const int64_t total = (wp_fprop ? (int64_t)T * mcd_kp(d->Cin) * mcd_mp(d->Cout) : 0) +
                        (wp_dgrad ? (int64_t)T * mcd_kp(d->Cout) * mcd_mp(d->Cin) : 0);
  const int blocks = (int)(ceil_div64(total, 256) > 4096 ? 4096 : ceil_div64(total, 256));
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wp_fprop, wp_dgrad, d->Cout,
                     d->Cin, T, mcd_mp(d->Cout), mcd_kp(d->Cin), mcd_mp(d->Cin), mcd_kp(d->Cout));
  MCD_LAUNCH_CHECK("conv_pack_weights");
  return 0;
}

extern "C" int64_t mcdseg_conv_stat_rows(const mcdseg_conv_desc* d) {
  if (d == nullptr) return -22;
  const int64_t P = (int64_t)d->N * d->Ho * d->Wo;
  return ceil_div64(P, bn_for(d->Cout)) * waves_n_for(d->Cout);
}

static int conv_fprop_impl(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* bias, float* y,
                           float* stat_partials, const float* ep_scale, const float* ep_shift, const float* ep_res, int ep_relu,
                           void* stream);

extern "C" int mcdseg_conv_fprop(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* bias, float* y,
                                 float* stat_partials, void* stream) {
  return conv_fprop_impl(d, x, wp_fprop, bias, y, stat_partials, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int mcdseg_conv_fprop_affine(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* scale,
                                        const float* shift, const float* residual, int32_t relu, float* y, void* stream) {
  MCD_REQUIRE(scale && shift, "conv_fprop_affine: null scale/shift");
  return conv_fprop_impl(d, x, wp_fprop, nullptr, y, nullptr, scale, shift, residual, relu, stream);
}

static int conv_fprop_impl(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* bias, float* y,
                           float* stat_partials, const float* ep_scale, const float* ep_shift, const float* ep_res, int ep_relu,
                           void* stream) {
  if (int rc = check_desc(d, "conv_fprop")) return rc;
  MCD_REQUIRE(x && wp_fprop && y, "conv_fprop: null pointer");
  ConvGemmParams p;
  p.ep_scale = ep_scale; p.ep_shift = ep_shift; p.ep_res = ep_res; p.ep_relu = ep_relu;
  p.src = x;
  p.wp = wp_fprop;
  p.bias = bias;
  p.dst = y;
  p.stats = stat_partials;
  p.N = d->N;
  p.Cs = d->Cin; p.Hs = d->H; p.Ws = d->W;
  p.M = d->Cout; p.Hd = d->Ho; p.Wd = d->Wo;
  p.Mp = mcd_mp(d->Cout);
  p.Kp = mcd_kp(d->Cin);
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.P = d->N * d->Ho * d->Wo;
  p.src_bytes = (int)((int64_t)d->N * d->Cin * d->H * d->W * 4);
  p.wp_bytes = (int)((int64_t)d->KH * d->KW * p.Kp * p.Mp * 4);
  launch<false>(p, (hipStream_t)stream);
  MCD_LAUNCH_CHECK("conv_fprop");
  return 0;
}

extern "C" int mcdseg_conv_dgrad(const mcdseg_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* stream) {
  if (int rc = check_desc(d, "conv_dgrad")) return rc;
  MCD_REQUIRE(dy && wp_dgrad && dx, "conv_dgrad: null pointer");
  ConvGemmParams p;
  p.ep_scale = nullptr; p.ep_shift = nullptr; p.ep_res = nullptr; p.ep_relu = 0;
  p.src = dy;
  p.wp = wp_dgrad;
  p.bias = nullptr;
  p.dst = dx;
  p.stats = nullptr;
  p.N = d->N;
  p.Cs = d->Cout; p.Hs = d->Ho; p.Ws = d->Wo;
  p.M = d->Cin; p.Hd = d->H; p.Wd = d->W;
  p.Mp = mcd_mp(d->Cin);
  p.Kp = mcd_kp(d->Cout);
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.P = d->N * d->H * d->W;
  p.src_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 4);
  p.wp_bytes = (int)((int64_t)d->KH * d->KW * p.Kp * p.Mp * 4);
  launch<true>(p, (hipStream_t)stream);
  MCD_LAUNCH_CHECK("conv_dgrad");
  return 0;
}
