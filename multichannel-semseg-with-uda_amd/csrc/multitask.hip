// Kernels only the multitask (segmentation + HHA regression) decoder needs -- BASELINE config 4:
//   bilinear x8 up-sampling, align_corners = False  (nn.Upsample(scale_factor=8, mode='bilinear'),
//                                                    models/dilated_fcn.py:676, 684-695)
//   mean-squared error with its gradient             (F.mse_loss, models/dilated_fcn.py:712-714)
// Both are HBM-bound streaming kernels on the full-resolution tensor.
#include "common.h"

namespace {

// source index / weights of torch's upsample_bilinear2d (align_corners=False): src = (dst + 0.5)/8 - 0.5, clamped at 0
__device__ __forceinline__ void src_index(int dst, int in_size, int& i0, int& i1, float& l0, float& l1) {
  float s = 0.125f * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void bilinear8_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi) {
  const int plane = blockIdx.x;
  const int Wo = Wi * 8, Ho = Hi * 8;
  const int q_per_row = Wo >> 2;
  const int total = Ho * q_per_row;
  const float* xin = x + (size_t)plane * Hi * Wi;
  float4* yout = reinterpret_cast<float4*>(y + (size_t)plane * Ho * Wo);
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
    const int oy = idx / q_per_row;
    const int ox0 = (idx - oy * q_per_row) << 2;
    int y0, y1;
    float ly0, ly1;
    src_index(oy, Hi, y0, y1, ly0, ly1);
    const float* r0 = xin + y0 * Wi;
    const float* r1 = xin + y1 * Wi;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1;
      float lx0, lx1;
      src_index(ox0 + j, Wi, x0, x1, lx0, lx1);
      o[j] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
    }
    yout[idx] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// gather form of the backward: input pixel (iy,ix) collects from output rows/cols [8i-4, 8i+12)
__device__ __forceinline__ float tap_weight(int dst, int in_size, int i) {
  int i0, i1;
  float l0, l1;
  src_index(dst, in_size, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

__global__ __launch_bounds__(256) void bilinear8_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi) {
  const int plane = blockIdx.x;
  const int Wo = Wi * 8, Ho = Hi * 8;
  const float* g = dy + (size_t)plane * Ho * Wo;
  const int total = Hi * Wi;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
    const int iy = idx / Wi;
    const int ix = idx - iy * Wi;
    const int oy0 = 8 * iy - 4, ox0 = 8 * ix - 4;
    float wx[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ox = ox0 + k;
      wx[k] = (ox >= 0 && ox < Wo) ? tap_weight(ox, Wi, ix) : 0.f;
    }
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) {
      const int oy = oy0 + k;
      if (oy < 0 || oy >= Ho) continue;
      const float wy = tap_weight(oy, Hi, iy);
      const float* row = g + (size_t)oy * Wo;
      float r = 0.f;
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        const int ox = ox0 + 4 * v4;  // multiple of 4: the float4 is fully inside or fully outside the row
        if (ox < 0 || ox >= Wo) continue;
        const float4 gv = *reinterpret_cast<const float4*>(row + ox);
        r = fmaf(gv.x, wx[4 * v4 + 0], r);
        r = fmaf(gv.y, wx[4 * v4 + 1], r);
        r = fmaf(gv.z, wx[4 * v4 + 2], r);
        r = fmaf(gv.w, wx[4 * v4 + 3], r);
      }
      acc = fmaf(wy, r, acc);
    }
    dx[(size_t)plane * total + idx] = acc;
  }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ grad,
                                                  float* __restrict__ part, int64_t n4, int64_t n, float gscale) {
  const float4* p4 = reinterpret_cast<const float4*>(p);
  const float4* t4 = reinterpret_cast<const float4*>(t);
  float4* g4 = reinterpret_cast<float4*>(grad);
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = p4[i], b = t4[i];
    const float4 d = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    if (grad) g4[i] = make_float4(d.x * gscale, d.y * gscale, d.z * gscale, d.w * gscale);
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = p[i] - t[i];
    s += d * d;
    if (grad) grad[i] = d * gscale;
  }
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void mse_finalize_kernel(const float* __restrict__ part, int nblk, double inv_n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[i];
  __shared__ double sh[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * inv_n);
}

int mse_blocks(int64_t n) {
  const int64_t b = ceil_div64(n, 256 * 16);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int mcdseg_bilinear8_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t Hi, int32_t Wi, void* stream) {
  MCD_REQUIRE(x && y && N > 0 && C > 0 && Hi > 0 && Wi > 0, "bilinear8_fwd: bad arguments");
  int chunks = ceil_div(Hi * 8 * Wi * 2, 256 * 4);
  if (chunks > 256) chunks = 256;
  hipLaunchKernelGGL(bilinear8_fwd_kernel, dim3(N * C, chunks), dim3(256), 0, (hipStream_t)stream, x, y, Hi, Wi);
  MCD_LAUNCH_CHECK("bilinear8_fwd");
  return 0;
}

extern "C" int mcdseg_bilinear8_bwd(const float* dy, float* dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi, void* stream) {
  MCD_REQUIRE(dy && dx && N > 0 && C > 0 && Hi > 0 && Wi > 0, "bilinear8_bwd: bad arguments");
  int chunks = ceil_div(Hi * Wi, 256);
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(bilinear8_bwd_kernel, dim3(N * C, chunks), dim3(256), 0, (hipStream_t)stream, dy, dx, Hi, Wi);
  MCD_LAUNCH_CHECK("bilinear8_bwd");
  return 0;
}

extern "C" size_t mcdseg_mse_workspace_bytes(int64_t n) { return n > 0 ? (size_t)mse_blocks(n) * sizeof(float) : 0; }

extern "C" int mcdseg_mse(const float* pred, const float* target, float* grad, float* loss, int64_t n, void* workspace,
                          size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(pred && target && loss && workspace && n > 0, "mse: bad arguments");
  const int nb = mse_blocks(n);
  MCD_REQUIRE(workspace_bytes >= (size_t)nb * sizeof(float), "mse: workspace too small");
  const bool al = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, pred, target, grad, (float*)workspace, n4, n, (float)(2.0 / (double)n));
  MCD_LAUNCH_CHECK("mse");
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, 1.0 / (double)n, loss);
  MCD_LAUNCH_CHECK("mse_finalize");
  return 0;
}
