// Weight-gradient convolution on the bf16 matrix pipe with 3-way split fp32 operands ("bf16x6", see
// conv_gemm_x6.hip for the arithmetic).  128 x 128 (co x ci) tiles only -- the layers that carry the FLOPs; thin
// layers stay on the f32 kernels of conv_wgrad.hip.
//
//   D[co][ci] (one tap) = sum_pix dY[co][pix] * X[ci][pix + shift(tap)]
//
// The contraction runs over pixels, 16 per K-step (one K=16 MFMA block).  A thread stages one pixel PAIR of four
// dY rows and four X rows: lanes run along the pixels of a row, so global reads stay pixel-contiguous; each pair is
// split into bf16 pieces and lands as one 32-bit word in the fragment image [piece][k-half][row][8 bf16], which the
// MFMA lanes read back as ds_read_b128 over 512 contiguous bytes per half-wave.  Slabs + fixed-order fp64 reduce
// as in conv_wgrad.hip (same plan, same workspace).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct WgradX6Params {
  const float* x;
  const float* dy;
  float* slab;
  int N, Cin, H, W, Cout, Ho, Wo;
  int KH, KW, stride, pad, dil;
  int co_p, ci_p;
  int chunk, chunks_per_img, splits;
  int x_bytes, dy_bytes;
};

__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& w1, unsigned& w2, unsigned& w3) {
  const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
  const float r0 = v0 - (float)a0, r1 = v1 - (float)a1;
  const __bf16 b0 = (__bf16)r0, b1 = (__bf16)r1;
  const __bf16 c0 = (__bf16)(r0 - (float)b0), c1 = (__bf16)(r1 - (float)b1);
  w1 = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
  w2 = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
  w3 = (unsigned)__builtin_bit_cast(unsigned short, c0) | ((unsigned)__builtin_bit_cast(unsigned short, c1) << 16);
}

__global__ __launch_bounds__(256, 2) void conv_wgrad_x6_kernel(WgradX6Params p) {
  constexpr int BM = 128, BN = 128, BKP = 16, NT = 256;
  constexpr int WM = 2, WN = 2, WAVES_N = 2;
  constexpr int A_BYTES = 6 * BM * 16, B_BYTES = 6 * BN * 16;  // [piece 3][k-half 2][row][16 B]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * A_BYTES;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

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

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int q = t & 7;      // pixel pair within the 16-pixel step
  const int row0 = t >> 3;  // rows row0 + 32 i, i < 4
  const int a_soff0 = (n * p.Cout + tile_co * BM) * HoWo * 4;
  const int b_soff0 = (n * p.Cin + tile_ci * BN) * HW * 4;
  const unsigned a_row = (unsigned)row0 * (unsigned)HoWo;
  const unsigned b_row = (unsigned)row0 * (unsigned)HW;
  const int lds_word = ((q >> 2) * BM + row0) * 16 + (q & 3) * 4;  // + piece*2*BM*16 + 32*i*16

  float areg[4][2], breg[4][2];
  auto load_regs = [&](int r0) {
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = r0 + 2 * q + e;
      a_voff[e] = OOB;
      b_voff[e] = OOB;
      if (r < r_end) {
        a_voff[e] = (a_row + (unsigned)r) * 4u;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        const int iy = oy * p.stride + ky * p.dil - p.pad;
        const int ix = ox * p.stride + kx * p.dil - p.pad;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) b_voff[e] = (b_row + (unsigned)(iy * p.W + ix)) * 4u;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        areg[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dy_rs, a_voff[e], a_soff0 + 32 * i * HoWo * 4, 0));
        breg[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, b_voff[e], b_soff0 + 32 * i * HW * 4, 0));
      }
  };
  auto store_lds = [&](int buf) {
    unsigned char* a = As + buf * A_BYTES + lds_word;
    unsigned char* b = Bs + buf * B_BYTES + lds_word;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned w1, w2, w3;
      split_pair(areg[i][0], areg[i][1], w1, w2, w3);
      *reinterpret_cast<unsigned*>(a + (0 * 2 * BM + 32 * i) * 16) = w1;
      *reinterpret_cast<unsigned*>(a + (1 * 2 * BM + 32 * i) * 16) = w2;
      *reinterpret_cast<unsigned*>(a + (2 * 2 * BM + 32 * i) * 16) = w3;
      split_pair(breg[i][0], breg[i][1], w1, w2, w3);
      *reinterpret_cast<unsigned*>(b + (0 * 2 * BN + 32 * i) * 16) = w1;
      *reinterpret_cast<unsigned*>(b + (1 * 2 * BN + 32 * i) * 16) = w2;
      *reinterpret_cast<unsigned*>(b + (2 * 2 * BN + 32 * i) * 16) = w3;
    }
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
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * 64 + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * 64 + l31) * 16;
    bf16x8 a[3][WM], b[3][WN];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < WM; ++i) a[pc][i] = *reinterpret_cast<const bf16x8*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[pc][j] = *reinterpret_cast<const bf16x8*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  float* out = p.slab + ((size_t)split * T_ + tap) * p.co_p * p.ci_p;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_co * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = tile_ci * BN + wn * 64 + j * 32 + l31;
        out[(size_t)row * p.ci_p + col] = acc[i][j][r];
      }
    }
}

}  // namespace

// launched by mcdseg_conv_wgrad (conv_wgrad.hip) when the 128x128 plan applies and bf16x6 math is requested
int mcdseg_internal_wgrad_x6_launch(const mcdseg_conv_desc* d, const float* x, const float* dy, float* slab, int co_p, int ci_p,
                                    int chunk, int chunks_per_img, int splits, hipStream_t st) {
  WgradX6Params p;
  p.x = x; p.dy = dy; p.slab = slab;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_p = co_p; p.ci_p = ci_p; p.chunk = chunk; p.chunks_per_img = chunks_per_img; p.splits = splits;
  p.x_bytes = (int)((int64_t)d->N * d->Cin * d->H * d->W * 4);
  p.dy_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 4);
  const int64_t per_split = (int64_t)(co_p / 128) * (ci_p / 128) * d->KH * d->KW;
  const int64_t nwg = 8 * ceil_div64(splits, 8) * per_split;
  if (nwg >= (1ll << 31)) {
    mcdseg_set_error("conv_wgrad_x6: grid too large");
    return -22;
  }
  hipLaunchKernelGGL(conv_wgrad_x6_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_x6");
  return 0;
}
