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
  double s = 0.0;
  for (int k = 0; k < slabs; ++k) s += (double)part[(size_t)k * C * 256 + i];
  dw[i] = (float)s;
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
