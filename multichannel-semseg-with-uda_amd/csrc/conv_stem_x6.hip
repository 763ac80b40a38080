// Direct convolution for the network stem (models/drn.py:126-131: 7x7, stride 1, pad 3, 6 -> 16 channels at full
// resolution) in the bf16x6 split arithmetic of conv_gemm_x6.hip.
//
// The implicit-GEMM kernels gather every input pixel once per tap (49x) and pad the 16 output channels to a 32-row MFMA
// tile and the 6 input channels to a 16-deep K-step; for this layer that is 5-15x the work the arithmetic needs.  Here
//   * a workgroup stages its input tile WITH halo (8+6 rows x 64+6 columns) into LDS once, already split into the three
//     bf16 pieces, one 16-B entry per pixel = 8 channel slots (Cin <= 8): every input value is split once per tile,
//     not once per tap;
//   * the MFMA is v_mfma_f32_16x16x32_bf16: 16 output channels x 16 pixels, K = 32 = four taps x eight channel slots.
//     A B-operand fragment (8 consecutive k of one pixel) is exactly one LDS entry of the tap-shifted pixel, so the
//     "im2col" is nothing but the ds_read_b128 address;
//   * the weights (13 K-steps x 3 pieces) live in registers for the whole tile: lane (row, k-group) keeps its 39
//     fragments, loaded once from the packed image [k-step][piece][k-group][row][8 bf16];
//   * a wave owns 2 rows x 64 columns = 8 pixel blocks, so one set of weight fragments feeds 8 x 6 MFMAs per K-step.
// Epilogue as in the GEMM kernels: bias, BN batch-statistics partial rows (count, mean, M2 per wave), or the eval-mode
// affine + residual + ReLU.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ST_TW = 64, ST_TH = 8;  // output tile
constexpr int ST_KS = 7;              // kernel size this file is built for

struct StemParams {
  const float* x;
  const void* wp;
  const float* bias;
  float* y;
  float* stats;
  const float* ep_scale;
  const float* ep_shift;
  const float* ep_res;
  int ep_relu;
  int N, Cin, H, W, Cout, Mp;
  int tiles_x, tiles_y;
};

__global__ __launch_bounds__(256, 2) void conv_stem_x6_kernel(StemParams p) {
  constexpr int K = ST_KS, PAD = K / 2;
  constexpr int HW_ = ST_TW + K - 1, HH_ = ST_TH + K - 1;  // halo tile 70 x 14
  constexpr int TPIX = HW_ * HH_;
  constexpr int KSTEPS = (K * K + 3) / 4;                  // 13
  __shared__ __attribute__((aligned(16))) unsigned char tile[3 * TPIX * 16];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int j = lane & 15, kg = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x;
  b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int n = b / p.tiles_y;
  const int x0 = tx * ST_TW, y0 = ty * ST_TH;
  const size_t HW = (size_t)p.H * p.W;

  // ---- weights -> registers (contiguous 1 KB per (k-step, piece))
  bf16x8 afr[KSTEPS][3];
  {
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(p.wp);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) afr[ks][pc] = wp[(ks * 3 + pc) * 64 + lane];
  }
  // ---- input tile with halo -> LDS, split once
  const float* xin = p.x + (size_t)n * p.Cin * HW;
  for (int idx = t; idx < TPIX; idx += 256) {
    const int hy = idx / HW_;
    const int hx = idx - hy * HW_;
    const int iy = y0 - PAD + hy, ix = x0 - PAD + hx;
    const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (ok && c < p.Cin) ? xin[(size_t)c * HW + (size_t)iy * p.W + ix] : 0.f;
    bf16x8 p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 a = (__bf16)v[e];
      const float r1 = v[e] - (float)a;
      const __bf16 bb = (__bf16)r1;
      p1[e] = a;
      p2[e] = bb;
      p3[e] = (__bf16)(r1 - (float)bb);
    }
    *reinterpret_cast<bf16x8*>(tile + (0 * TPIX + idx) * 16) = p1;
    *reinterpret_cast<bf16x8*>(tile + (1 * TPIX + idx) * 16) = p2;
    *reinterpret_cast<bf16x8*>(tile + (2 * TPIX + idx) * 16) = p3;
  }
  __syncthreads();

  f32x4 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane_base = ((2 * wave) * HW_ + j) * 16;
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    int ts = 4 * ks + kg;
    if (ts >= K * K) ts = 0;  // zero weights there; any finite operand will do
    const int ky = ts / K;
    const int kx = ts - ky * K;
    const unsigned char* src = tile + lane_base + (ky * HW_ + kx) * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int boff = ((q >> 2) * HW_ + (q & 3) * 16) * 16;
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(src + 0 * TPIX * 16 + boff);
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(src + 1 * TPIX * 16 + boff);
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(src + 2 * TPIX * 16 + boff);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][2], b1, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][0], b3, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][1], b2, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][1], b1, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][0], b2, acc[q], 0, 0, 0);
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks][0], b1, acc[q], 0, 0, 0);
    }
  }

  // ---- epilogue: acc[q][r] = D[channel 4*kg + r][pixel (row 2*wave + (q>>2), column 16*(q&3) + j)]
  bool pv[8];
  size_t pbase[8];
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int oy = y0 + 2 * wave + (q >> 2), ox = x0 + (q & 3) * 16 + j;
    pv[q] = oy < p.H && ox < p.W;
    pbase[q] = (size_t)n * p.Cout * HW + (size_t)oy * p.W + ox;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * kg + r;
    const float bv = (p.bias != nullptr && m < p.Cout) ? p.bias[m] : 0.f;
    const float sc = (p.ep_scale != nullptr && m < p.Cout) ? p.ep_scale[m] : 1.f;
    const float sh = (p.ep_scale != nullptr && m < p.Cout) ? p.ep_shift[m] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[q][r] + bv;
      if (p.ep_scale != nullptr) {
        v = fmaf(v, sc, sh);
        if (p.ep_res != nullptr && pv[q] && m < p.Cout) v += p.ep_res[pbase[q] + (size_t)m * HW];
        if (p.ep_relu) v = fmaxf(v, 0.f);
      }
      acc[q][r] = v;
      if (pv[q] && m < p.Cout) p.y[pbase[q] + (size_t)m * HW] = v;
    }
  }
  if (p.stats != nullptr) {
    // valid pixels of this wave (the same for every channel): rows 2*wave, 2*wave+1 of the tile, 64 columns
    const int rows_in = min(max(p.H - (y0 + 2 * wave), 0), 2);
    const int cols_in = min(max(p.W - x0, 0), ST_TW);
    cnt = rows_in * cols_in;
    const float inv = cnt > 0 ? 1.f / (float)cnt : 0.f;
    const size_t srow = ((size_t)blockIdx.x * 4 + wave) * 3 * p.Mp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += pv[q] ? acc[q][r] : 0.f;
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      const float mean = s * inv;
      float m2 = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float d = acc[q][r] - mean;
        m2 += pv[q] ? d * d : 0.f;
      }
      m2 += __shfl_xor(m2, 1);
      m2 += __shfl_xor(m2, 2);
      m2 += __shfl_xor(m2, 4);
      m2 += __shfl_xor(m2, 8);
      if (j == r) {
        const int m = 4 * kg + r;
        p.stats[srow + m] = (float)cnt;
        p.stats[srow + p.Mp + m] = mean;
        p.stats[srow + 2 * (size_t)p.Mp + m] = m2;
      }
    }
  }
}

// w[Cout][Cin][K*K] fp32 -> [k-step][piece 3][k-group 4][row 16][8 bf16]; k slot (ks, kg, e) = tap 4*ks + kg, channel e
__global__ void pack_stem_x6_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin) {
  constexpr int K = ST_KS, KSTEPS = (K * K + 3) / 4;
  const int total = KSTEPS * 4 * 16 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 7;
    const int row = (i >> 3) & 15;
    const int kg = (i >> 7) & 3;
    const int ks = i >> 9;
    const int tap = 4 * ks + kg;
    float v = 0.f;
    if (row < Cout && e < Cin && tap < K * K) v = w[((size_t)row * Cin + e) * (K * K) + tap];
    const __bf16 a = (__bf16)v;
    const float r1 = v - (float)a;
    const __bf16 bb = (__bf16)r1;
    const size_t base = (size_t)ks * 3 * 64 * 8;
    out[base + ((0 * 4 + kg) * 16 + row) * 8 + e] = a;
    out[base + ((1 * 4 + kg) * 16 + row) * 8 + e] = bb;
    out[base + ((2 * 4 + kg) * 16 + row) * 8 + e] = (__bf16)(r1 - (float)bb);
  }
}

}  // namespace

// ---- internal interface used by conv_gemm_x6.hip
bool mcdseg_internal_stem_ok(const mcdseg_conv_desc* d) {
  return d->KH == ST_KS && d->KW == ST_KS && d->stride == 1 && d->dil == 1 && d->pad == ST_KS / 2 && d->Cin <= 8 && d->Cout <= 16 &&
         d->Ho == d->H && d->Wo == d->W;
}

int64_t mcdseg_internal_stem_stat_rows(const mcdseg_conv_desc* d) {
  return (int64_t)d->N * ceil_div(d->H, ST_TH) * ceil_div(d->W, ST_TW) * 4;
}

int64_t mcdseg_internal_stem_image_bytes() { return (int64_t)((ST_KS * ST_KS + 3) / 4) * 3 * 64 * 16; }

int mcdseg_internal_stem_pack(const mcdseg_conv_desc* d, const float* w, void* out, hipStream_t st) {
  hipLaunchKernelGGL(pack_stem_x6_kernel, dim3(26), dim3(256), 0, st, w, (__bf16*)out, d->Cout, d->Cin);
  MCD_LAUNCH_CHECK("conv_x6_pack_weights(stem)");
  return 0;
}

int mcdseg_internal_stem_fprop(const mcdseg_conv_desc* d, const float* x, const void* wp, const float* bias, float* y, float* stats,
                               const float* ep_scale, const float* ep_shift, const float* ep_res, int ep_relu, hipStream_t st) {
  StemParams p;
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.stats = stats;
  p.ep_scale = ep_scale; p.ep_shift = ep_shift; p.ep_res = ep_res; p.ep_relu = ep_relu;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Mp = mcd_mp(d->Cout);
  p.tiles_x = ceil_div(d->W, ST_TW);
  p.tiles_y = ceil_div(d->H, ST_TH);
  const int64_t nwg = (int64_t)d->N * p.tiles_x * p.tiles_y;
  MCD_REQUIRE(nwg < (1ll << 31), "conv_x6_fprop(stem): grid too large");
  hipLaunchKernelGGL(conv_stem_x6_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
  MCD_LAUNCH_CHECK("conv_x6_fprop(stem)");
  return 0;
}
