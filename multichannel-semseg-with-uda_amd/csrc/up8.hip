// x8 learned up-sampler: depthwise ConvTranspose2d(C, C, 16, stride 8, pad 4, groups=C, bias=False)
// (models/dilated_fcn.py:357-366).  out[n,c,oy,ox] = sum_{iy,ix} in[n,c,iy,ix] * w[c,oy+4-8iy,ox+4-8ix];
// the kernel index must lie in [0,16) so every output pixel has at most 2x2 contributing inputs.
// All three kernels are HBM-bound on the full-resolution tensor (read or written exactly once by
// design; the 64x smaller input / 1 KB of weights per channel live in LDS and cache).
#include "common.h"

namespace {

// forward: one thread = 4 consecutive output pixels of one row (one float4 store).  For ox0 % 4 == 0 the
// four pixels share their input columns: hi = (ox0+4)>>3 and hi-1, kernel columns kx0..kx0+3 (+8).
template <bool DUAL>
__global__ __launch_bounds__(256) void up8_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ x2, const float* __restrict__ w2,
                                                      float* __restrict__ y, int C, int Hi, int Wi) {
  __shared__ float wsh[2][256];
  const int plane = blockIdx.x;  // n*C + c
  const int c = plane % C;
  wsh[0][threadIdx.x] = w[c * 256 + threadIdx.x];
  if (DUAL) wsh[1][threadIdx.x] = w2[c * 256 + threadIdx.x];
  __syncthreads();
  const int Wo = Wi * 8, Ho = Hi * 8;
  const int q_per_row = Wo >> 2;
  const int total = Ho * q_per_row;
  const float* xin = x + (size_t)plane * Hi * Wi;
  const float* xin2 = DUAL ? x2 + (size_t)plane * Hi * Wi : nullptr;
  float4* yout = reinterpret_cast<float4*>(y + (size_t)plane * Ho * Wo);
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
    const int oy = idx / q_per_row;
    const int q = idx - oy * q_per_row;
    const int ox0 = q << 2;
    const int iy_hi = (oy + 4) >> 3, ky0 = (oy + 4) & 7;
    const int ix_hi = (ox0 + 4) >> 3, kx0 = (ox0 + 4) & 7;  // kx0 is 0 or 4
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int iy = iy_hi - a;
      if (iy < 0 || iy >= Hi) continue;
      const int ky = ky0 + 8 * a;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ix = ix_hi - b;
        if (ix < 0 || ix >= Wi) continue;
        const int kx = kx0 + 8 * b;
        const float v = xin[iy * Wi + ix];
        const float4 wv = *reinterpret_cast<const float4*>(&wsh[0][ky * 16 + kx]);
        o.x = fmaf(v, wv.x, o.x); o.y = fmaf(v, wv.y, o.y); o.z = fmaf(v, wv.z, o.z); o.w = fmaf(v, wv.w, o.w);
        if (DUAL) {
          const float v2 = xin2[iy * Wi + ix];
          const float4 wv2 = *reinterpret_cast<const float4*>(&wsh[1][ky * 16 + kx]);
          o.x = fmaf(v2, wv2.x, o.x); o.y = fmaf(v2, wv2.y, o.y); o.z = fmaf(v2, wv2.z, o.z); o.w = fmaf(v2, wv2.w, o.w);
        }
      }
    }
    yout[idx] = o;
  }
}

// backward w.r.t. the input: dx[n,c,iy,ix] = sum_{ky,kx} dy[n,c,8iy-4+ky,8ix-4+kx] * w[c,ky,kx].
// One thread per input pixel; its 16x16 window of dy is read as 16 rows x 4 float4 (rows of neighbouring
// threads overlap by half and are served by L1/L2).
__global__ __launch_bounds__(256) void up8_bwd_input_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int C, int Hi, int Wi) {
  __shared__ float wsh[256];
  const int plane = blockIdx.x;
  const int c = plane % C;
  wsh[threadIdx.x] = w[c * 256 + threadIdx.x];
  __syncthreads();
  const int Wo = Wi * 8, Ho = Hi * 8;
  const float* g = dy + (size_t)plane * Ho * Wo;
  const int total = Hi * Wi;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
    const int iy = idx / Wi;
    const int ix = idx - iy * Wi;
    const int oy0 = 8 * iy - 4, ox0 = 8 * ix - 4;
    float acc = 0.f;
    for (int ky = 0; ky < 16; ++ky) {
      const int oy = oy0 + ky;
      if (oy < 0 || oy >= Ho) continue;
      const float* row = g + (size_t)oy * Wo;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const int ox = ox0 + 4 * v4;  // multiple of 4; the float4 is fully inside or fully outside the row
        if (ox < 0 || ox >= Wo) continue;
        const float4 gv = *reinterpret_cast<const float4*>(row + ox);
        const float4 wv = *reinterpret_cast<const float4*>(&wsh[ky * 16 + 4 * v4]);
        acc = fmaf(gv.x, wv.x, acc); acc = fmaf(gv.y, wv.y, acc); acc = fmaf(gv.z, wv.z, acc); acc = fmaf(gv.w, wv.w, acc);
      }
    }
    dx[(size_t)plane * total + idx] = acc;
  }
}

// backward w.r.t. the weights: dw[c,ky,kx] = sum_{n,iy,ix} x[n,c,iy,ix] * dy[n,c,8iy-4+ky,8ix-4+kx].
// thread = (ky,kx); a workgroup walks the input pixels of one (n,c) plane band and writes one partial.
__global__ __launch_bounds__(256) void up8_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ part, int N, int C, int Hi, int Wi,
                                                             int bands) {
  const int c = blockIdx.x;
  const int nb = blockIdx.y;  // n*bands + band
  const int n = nb / bands;
  const int band = nb - n * bands;
  const int rows_per_band = (Hi + bands - 1) / bands;
  const int iy_begin = band * rows_per_band;
  int iy_end = iy_begin + rows_per_band;
  if (iy_end > Hi) iy_end = Hi;
  const int ky = threadIdx.x >> 4, kx = threadIdx.x & 15;
  const int Wo = Wi * 8, Ho = Hi * 8;
  const size_t plane = (size_t)n * C + c;
  const float* g = dy + plane * Ho * Wo;
  const float* xin = x + plane * Hi * Wi;
  // four independent accumulators keep four row loads in flight; only the first and last input column can fall
  // outside the output row, so the bounds test is a select on the loaded value rather than a branch
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int iy = iy_begin; iy < iy_end; ++iy) {
    const int oy = 8 * iy - 4 + ky;
    if (oy < 0 || oy >= Ho) continue;
    const float* row = g + (size_t)oy * Wo;
    const float* xrow = xin + iy * Wi;
    int ix = 0;
    for (; ix + 4 <= Wi; ix += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = 8 * (ix + j) - 4 + kx;
        const bool ok = ox >= 0 && ox < Wo;
        const float gv = row[ok ? ox : 0];
        acc[j] = fmaf(xrow[ix + j], ok ? gv : 0.f, acc[j]);
      }
    }
    for (; ix < Wi; ++ix) {
      const int ox = 8 * ix - 4 + kx;
      if (ox < 0 || ox >= Wo) continue;
      acc[0] = fmaf(xrow[ix], row[ox], acc[0]);
    }
  }
  part[((size_t)nb * C + c) * 256 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

__global__ void up8_bwd_weight_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int C, int slabs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 256) return;
  // 16 loads in flight, added in slab order (one at a time this was a chain of N x bands dependent HBM round trips: 35 us for 96 slabs)
  double s = 0.0;
  const size_t SL = (size_t)C * 256;
  const float* src = part + i;
  int k = 0;
  for (; k + 16 <= slabs; k += 16, src += 16 * SL) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[(size_t)j * SL];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += (double)v[j];
  }
  for (; k < slabs; ++k, src += SL) s += (double)*src;
  dw[i] = (float)s;
}

// ---- both backward passes from ONE staged read of the logit gradient -------------------------------------------------------
// The two kernels above read every dy value from L1/L2 two (weights) to four (input) times through 16- or 4-byte accesses of
// overlapping windows and reach 0.2-0.3 of the HBM rate.  Here a workgroup owns a band of input rows of one (n, c) plane and
// walks it row by row over a ring of three LDS slots, each holding 8 rows of dy (chunk k = rows 8k-4 .. 8k+3; input row iy
// reads chunks iy and iy+1), filled by LDS-DMA two rows ahead; rows outside the plane are zero-filled by the buffer
// resource's range check.  LDS row stride = whole 1 KB DMA units + 64 B, i.e. = 16 banks mod 64: the weight-gradient access
// (lane = tap (ky, kx): 4 rows x 16 consecutive floats per wave) and the input-gradient access (consecutive lanes read
// consecutive float4) are both conflict-free.  dx: item = (ix, quarter v4 of the 16 kernel columns), 16 float4 reads x the
// thread's 16 weight float4 in registers, quad sum.  dw: thread = tap, x[iy][ix] is wave-uniform (scalar loads).
constexpr unsigned UB_OOB = 0x80000000u;

struct Up8BandParams {
  int N, C, Hi, Wi, bands, rows_per_band;
  int row_insts, row_stride;  // 1 KB DMA units per dy row; LDS row stride in bytes
  int x_off, xrow_bytes;      // the weight gradient's ring of input rows behind the three dy slots: offset and slot size in bytes
};

template <bool DX, bool DW>
__global__ __launch_bounds__(256) void up8_bwd_band_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           const float* __restrict__ x, float* __restrict__ dx,
                                                           float* __restrict__ part, Up8BandParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ub_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int plane = blockIdx.x, band = blockIdx.y;
  const int c = plane % p.C, n = plane / p.C;
  const int Wi = p.Wi, Hi = p.Hi, Wo = 8 * Wi, Ho = 8 * Hi;
  const int iy0 = band * p.rows_per_band;
  const int iy1 = iy0 + p.rows_per_band < Hi ? iy0 + p.rows_per_band : Hi;
  const int slot_bytes = 8 * p.row_stride;
  const float* g = dy + (size_t)plane * Ho * Wo;
  const mcd_i32x4 rs = mcd_raw_rsrc(g, Ho * Wo * 4);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ub_smem;
  const int chunk_insts = 8 * p.row_insts;
  const int row_bytes = 4 * Wo;
  // (the weight gradient's input values: wave-uniform, and until round 5 scalar loads inside the row loop -- five dependent round trips to
  // the scalar cache per row of 80, 1.5 us of the row's 2; now a row ahead in LDS, read as broadcasts)
  const mcd_i32x4 xrs = mcd_raw_rsrc(DW ? x + (size_t)plane * Hi * Wi : dy, DW ? Hi * Wi * 4 : 0);

  // (hidden from the compiler: through the builtin, the first LDS read of every row waits vmcnt(0) for the chunk issued a moment
  // before it -- no look-ahead at all; see mcd_hidden_dma)
  auto issue = [&](int k) {  // chunk k -> slot k % 3
    const unsigned slot = lds0 + (unsigned)((k % 3) * slot_bytes);
    for (int q = wave; q < chunk_insts; q += 4) {  // wave-uniform
      const int r = q / p.row_insts, j = q - r * p.row_insts;
      const int oy = 8 * k - 4 + r;
      const int cb = j * 1024 + lane * 16;
      const unsigned voff = ((unsigned)oy < (unsigned)Ho && cb < row_bytes) ? (unsigned)(oy * row_bytes + cb) : UB_OOB;
      mcd_hidden_dma<16>(rs, __builtin_amdgcn_readfirstlane(slot + (unsigned)(r * p.row_stride + j * 1024)), voff);
    }
    if (DW) {  // input row k - 1 travels with chunk k: row iy is read with chunks iy, iy + 1 (one dword per lane; instruction j by wave j & 3)
      const int rx = k - 1;
      for (int j = wave; j * 64 < Wi; j += 4) {
        const int col = j * 64 + lane;
        const unsigned voff = ((unsigned)rx < (unsigned)Hi && col < Wi) ? (unsigned)((rx * Wi + col) * 4) : UB_OOB;
        mcd_hidden_dma<4>(xrs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(p.x_off + (k % 3) * p.xrow_bytes + j * 256)), voff);
      }
    }
  };

  // dx: this thread's quarter of the kernel columns
  const int v4 = tid & 3;
  float4 wq[16];
  if (DX) {
#pragma unroll
    for (int ky = 0; ky < 16; ++ky) wq[ky] = *reinterpret_cast<const float4*>(w + c * 256 + ky * 16 + 4 * v4);
    // arrived HERE: left to the compiler, the wait for these loads sits in front of their first use -- inside the row loop, where
    // a vmcnt(0) also drains the chunk just issued, every row
#pragma unroll
    for (int ky = 0; ky < 16; ++ky) asm volatile("" : "+v"(wq[ky].x), "+v"(wq[ky].y), "+v"(wq[ky].z), "+v"(wq[ky].w));
  }
  // dw: this thread's tap
  const int ky_t = tid >> 4, kx_t = tid & 15;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};

  issue(iy0);
  issue(iy0 + 1);
  for (int iy = iy0; iy < iy1; ++iy) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // chunks iy, iy+1 are in LDS; every wave is done with row iy-1, whose older slot is refilled now
    if (iy + 1 < iy1) issue(iy + 2);
    const unsigned char* sa = ub_smem + (iy % 3) * slot_bytes;
    const unsigned char* sb = ub_smem + ((iy + 1) % 3) * slot_bytes;
    if (DX) {
      for (int item = tid; item < 4 * Wi; item += 256) {
        const int ix = item >> 2;
        const int colb = (8 * ix - 4 + 4 * v4) * 4;
        const bool ok = !((ix == 0 && v4 == 0) || (ix == Wi - 1 && v4 == 3));
        float a = 0.f;
        if (ok) {
          // all 16 reads in flight before the first multiply-add (left alone, the compiler re-uses one register quad and waits
          // out the LDS latency 16 times per item)
          f32x4 gv[16];
#pragma unroll
          for (int ky = 0; ky < 16; ++ky) gv[ky] = *reinterpret_cast<const f32x4*>((ky < 8 ? sa : sb) + (ky & 7) * p.row_stride + colb);
          asm volatile("" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]), "+v"(gv[8]),
                            "+v"(gv[9]), "+v"(gv[10]), "+v"(gv[11]), "+v"(gv[12]), "+v"(gv[13]), "+v"(gv[14]), "+v"(gv[15]));
#pragma unroll
          for (int ky = 0; ky < 16; ++ky) {
            a = fmaf(gv[ky].x, wq[ky].x, a); a = fmaf(gv[ky].y, wq[ky].y, a); a = fmaf(gv[ky].z, wq[ky].z, a); a = fmaf(gv[ky].w, wq[ky].w, a);
          }
        }
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 2);
        if (v4 == 0) dx[((size_t)plane * Hi + iy) * Wi + ix] = a;
      }
    }
    if (DW) {
      const unsigned char* row = (ky_t < 8 ? sa : sb) + (ky_t & 7) * p.row_stride + (kx_t - 4) * 4;
      const float* xr = reinterpret_cast<const float*>(ub_smem + p.x_off + ((iy + 1) % 3) * p.xrow_bytes);
      {  // first and last input column: part of the window lies outside the row
        const float g0 = (kx_t >= 4 && (Wi > 1 || kx_t < 12)) ? *reinterpret_cast<const float*>(row) : 0.f;
        acc[0] = fmaf(xr[0], g0, acc[0]);
        if (Wi > 1) {
          const float g1 = kx_t < 12 ? *reinterpret_cast<const float*>(row + 32 * (Wi - 1)) : 0.f;
          acc[1] = fmaf(xr[Wi - 1], g1, acc[1]);
        }
      }
      int ix = 1;
      // 16 columns at a time: their 32 LDS reads in flight together (the empty asm is what holds the compiler to that: left alone it
      // waits for each group of four on its own); same accumulators in the same order
      for (; ix + 16 <= Wi - 1; ix += 16) {
        f32x4 xq[4], gq[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          xq[j >> 2][j & 3] = xr[ix + j];
          gq[j >> 2][j & 3] = *reinterpret_cast<const float*>(row + 32 * (ix + j));
        }
        asm volatile("" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]));
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = fmaf(xq[j >> 2][j & 3], gq[j >> 2][j & 3], acc[j & 3]);
      }
      for (; ix + 4 <= Wi - 1; ix += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(xr[ix + j], *reinterpret_cast<const float*>(row + 32 * (ix + j)), acc[j]);
      }
      for (; ix < Wi - 1; ++ix) acc[0] = fmaf(xr[ix], *reinterpret_cast<const float*>(row + 32 * ix), acc[0]);
    }
  }
  if (DW) part[((size_t)(n * p.bands + band) * p.C + c) * 256 + tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

struct Up8BandPlan {
  bool ok;
  int rows_per_band, bands, row_insts, row_stride, lds;
};

Up8BandPlan up8_band_plan(int N, int C, int Hi, int Wi) {
  Up8BandPlan pl{};
  const int knob = (int)mcd_opt(MCD_OPT_UP8_BAND_ROWS);  // development knob: 0 = the two separate kernels, n = rows per band
  if (knob == 0) return pl;
  pl.row_insts = ceil_div(32 * Wi, 1024);
  pl.row_stride = pl.row_insts * 1024 + 64;
  pl.lds = 3 * 8 * pl.row_stride + 3 * round_up(Wi, 64) * 4;  // + the weight gradient's three input rows
  if (pl.lds > 160 * 1024 - 1024 || (int64_t)Hi * Wi * 256 >= (1ll << 31)) return pl;
  pl.rows_per_band = knob > 0 ? knob : (Hi >= 20 ? 10 : Hi);
  if (pl.rows_per_band > Hi) pl.rows_per_band = Hi;
  pl.bands = ceil_div(Hi, pl.rows_per_band);
  if (pl.bands > 65535 || (int64_t)N * C >= (1ll << 31)) return pl;
  pl.ok = true;
  return pl;
}

int bands_for(int N, int C, int Hi) {
  int bands = ceil_div(2048, N * C);
  if (bands < 1) bands = 1;
  if (bands > Hi) bands = Hi;
  return bands;
}

}  // namespace

extern "C" int mcdseg_up8_fwd(const float* x, const float* w, const float* x2, const float* w2, float* y, int32_t N, int32_t C,
                              int32_t Hi, int32_t Wi, void* stream) {
  MCD_REQUIRE(x && w && y, "up8_fwd: null pointer");
  MCD_REQUIRE((x2 == nullptr) == (w2 == nullptr), "up8_fwd: x2 and w2 come together");
  MCD_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0, "up8_fwd: bad dims");
  const int total = Hi * 8 * Wi * 2;
  int chunks = ceil_div(total, 256 * 4);
  if (chunks > 256) chunks = 256;
  dim3 grid(N * C, chunks);
  if (x2)
    hipLaunchKernelGGL(up8_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, x2, w2, y, C, Hi, Wi);
  else
    hipLaunchKernelGGL(up8_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, x2, w2, y, C, Hi, Wi);
  MCD_LAUNCH_CHECK("up8_fwd");
  return 0;
}

extern "C" int mcdseg_up8_bwd_input(const float* dy, const float* w, float* dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                    void* stream) {
  MCD_REQUIRE(dy && w && dx, "up8_bwd_input: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0, "up8_bwd_input: bad dims");
  int chunks = ceil_div(Hi * Wi, 256);
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(up8_bwd_input_kernel, dim3(N * C, chunks), dim3(256), 0, (hipStream_t)stream, dy, w, dx, C, Hi, Wi);
  MCD_LAUNCH_CHECK("up8_bwd_input");
  return 0;
}

extern "C" size_t mcdseg_up8_bwd_weight_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi) {
  if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return 0;
  return (size_t)N * bands_for(N, C, Hi) * C * 256 * sizeof(float);
}

extern "C" int mcdseg_up8_bwd_weight(const float* dy, const float* x, float* dw, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(dy && x && dw && workspace, "up8_bwd_weight: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0, "up8_bwd_weight: bad dims");
  const int bands = bands_for(N, C, Hi);
  MCD_REQUIRE(workspace_bytes >= (size_t)N * bands * C * 256 * sizeof(float), "up8_bwd_weight: workspace too small");
  MCD_REQUIRE(N * bands <= 65535, "up8_bwd_weight: grid too large");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(up8_bwd_weight_kernel, dim3(C, N * bands), dim3(256), 0, st, dy, x, (float*)workspace, N, C, Hi, Wi, bands);
  MCD_LAUNCH_CHECK("up8_bwd_weight");
  hipLaunchKernelGGL(up8_bwd_weight_reduce_kernel, dim3(ceil_div(C * 256, 256)), dim3(256), 0, st, (const float*)workspace, dw, C,
                     N * bands);
  MCD_LAUNCH_CHECK("up8_bwd_weight_reduce");
  return 0;
}

extern "C" size_t mcdseg_up8_bwd_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi) {
  if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return 0;
  const Up8BandPlan pl = up8_band_plan(N, C, Hi, Wi);
  const size_t two = mcdseg_up8_bwd_weight_workspace_bytes(N, C, Hi, Wi);
  const size_t one = pl.ok ? (size_t)N * pl.bands * C * 256 * sizeof(float) : 0;
  return one > two ? one : two;
}

extern "C" int mcdseg_up8_bwd(const float* dy, const float* w, const float* x, float* dx, float* dw, int32_t N, int32_t C, int32_t Hi,
                              int32_t Wi, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(dy && (dx || dw), "up8_bwd: null pointer");
  MCD_REQUIRE((dx == nullptr || w != nullptr) && (dw == nullptr || (x != nullptr && workspace != nullptr)),
              "up8_bwd: the input gradient needs w, the weight gradient needs x and a workspace");
  MCD_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0, "up8_bwd: bad dims");
  const Up8BandPlan pl = up8_band_plan(N, C, Hi, Wi);
  if (!pl.ok) {  // wider than the LDS ring allows: the two separate kernels
    if (dx)
      if (int rc = mcdseg_up8_bwd_input(dy, w, dx, N, C, Hi, Wi, stream)) return rc;
    if (dw)
      if (int rc = mcdseg_up8_bwd_weight(dy, x, dw, N, C, Hi, Wi, workspace, workspace_bytes, stream)) return rc;
    return 0;
  }
  MCD_REQUIRE(dw == nullptr || workspace_bytes >= (size_t)N * pl.bands * C * 256 * sizeof(float), "up8_bwd: workspace too small");
  Up8BandParams p;
  p.N = N; p.C = C; p.Hi = Hi; p.Wi = Wi; p.bands = pl.bands; p.rows_per_band = pl.rows_per_band;
  p.row_insts = pl.row_insts; p.row_stride = pl.row_stride;
  p.x_off = 3 * 8 * pl.row_stride; p.xrow_bytes = round_up(Wi, 64) * 4;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(N * C, pl.bands), block(256);
  static const bool attr = [] {
    (void)hipFuncSetAttribute((const void*)up8_bwd_band_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)up8_bwd_band_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)up8_bwd_band_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  if (dx && dw)
    hipLaunchKernelGGL((up8_bwd_band_kernel<true, true>), grid, block, pl.lds, st, dy, w, x, dx, (float*)workspace, p);
  else if (dx)
    hipLaunchKernelGGL((up8_bwd_band_kernel<true, false>), grid, block, pl.lds, st, dy, w, x, dx, (float*)workspace, p);
  else
    hipLaunchKernelGGL((up8_bwd_band_kernel<false, true>), grid, block, pl.lds, st, dy, w, x, dx, (float*)workspace, p);
  MCD_LAUNCH_CHECK("up8_bwd");
  if (dw) {
    hipLaunchKernelGGL(up8_bwd_weight_reduce_kernel, dim3(ceil_div(C * 256, 256)), dim3(256), 0, st, (const float*)workspace, dw, C,
                       N * pl.bands);
    MCD_LAUNCH_CHECK("up8_bwd_weight_reduce");
  }
  return 0;
}
