// Either side of the train/test step: the device part of the input pipeline and of the evaluation.
//
//  * normalize_u8   -- ToTensor() + Normalize(mean, std) of the reference's image transform (transform.py:302-315):
//                      uint8 HWC (PIL / numpy layout, as it comes out of the loader's pinned buffer) -> fp32 NCHW,
//                      ((u/255) - mean[c]) / std[c] with IEEE fp32 division, the arithmetic torchvision performs.
//                      RGB and HHA are two source images written into channel ranges of one 6-channel batch.
//  * resize_*_u8    -- Scale(img_shape, Image.BILINEAR) / Scale(img_shape, Image.NEAREST) in front of them (transform.py:303, 320):
//                      torchvision's Scale is PIL.Image.resize, whose 8-bit arithmetic is integer once the filter
//                      coefficients exist -- Pillow's ImagingResample (triangle filter stretched by the shrink factor,
//                      coefficients normalised in double and rounded to 22-bit fixed point, horizontal then vertical pass with a
//                      uint8 intermediate) and ImagingScaleAffine for NEAREST (source index = int of a double advanced by
//                      repeated addition).  The tables are built on the device in the same double arithmetic (the library is
//                      compiled with -ffp-contract=off), the passes are integer: results equal Pillow's bit for bit.
//  * relabel_u8     -- ToLabel() + ReLabel(255 -> n_class-1) of the label transform (transform.py:21-48, 319-325).
//  * confusion_hist -- fast_hist of eval.py:21-23: bincount(n*gt + pred) over pixels whose gt lies in [0, n).
// All three are single-pass HBM streams; the histogram accumulates in LDS (32-bit integer atomics, exact) and flushes
// with 64-bit integer atomics, so the result does not depend on the order of execution.
#include <cmath>
#include "common.h"

namespace {

template <int CS>
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                           const float* __restrict__ mean, const float* __restrict__ stdv,
                                                           int HW, int C, int c_off, int cs_rt) {
  const int cs = CS > 0 ? CS : cs_rt;
  const int n = blockIdx.y;
  const uint8_t* s = src + (size_t)n * HW * cs;
  float* d = dst + ((size_t)n * C + c_off) * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const uint8_t* px = s + (size_t)p * cs;
#pragma unroll
    for (int c = 0; c < (CS > 0 ? CS : 8); ++c) {
      if (c >= cs) break;
      const float v = (float)px[c] / 255.0f;
      d[(size_t)c * HW + p] = (v - mean[c]) / stdv[c];
    }
  }
}

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

// precompute_coeffs + normalize_coeffs_8bpc (Resample.c) for the triangle filter over a whole axis: one thread per output index
__global__ void resize_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const double center = ((double)xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  int* k = kk + (size_t)xx * ksize;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double v = ((double)(x + xmin) - center + 0.5) * ss;
    if (v < 0.0) v = -v;
    ww += v < 1.0 ? 1.0 - v : 0.0;
  }
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double v = ((double)(x + xmin) - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      w = v < 1.0 ? 1.0 - v : 0.0;
      if (ww != 0.0) w /= ww;
    }
    k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << RS_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << RS_PRECISION_BITS));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

// one pass of ImagingResample{Horizontal,Vertical}_8bpc: src [N][L0][L1][C] resampled along axis 1 (VERT) or 2 (horizontal)
template <bool VERT>
__global__ __launch_bounds__(256) void resize_pass_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                             int N, int H, int W, int C, int OH, int OW) {
  // output dims: VERT: [N][OH][W][C] from [N][H][W][C];  horizontal: [N][H][OW][C] from [N][H][W][C]
  const int64_t total = VERT ? (int64_t)N * OH * W * C : (int64_t)N * H * OW * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    int x, y, n;
    if (VERT) {
      x = (int)(r % W); r /= W;
      y = (int)(r % OH); n = (int)(r / OH);
    } else {
      x = (int)(r % OW); r /= OW;
      y = (int)(r % H); n = (int)(r / H);
    }
    const int o = VERT ? y : x;
    const int lo = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int ss = 1 << (RS_PRECISION_BITS - 1);
    for (int t = 0; t < cnt; ++t) {
      const size_t sidx = VERT ? (((size_t)n * H + (lo + t)) * W + x) * C + c : (((size_t)n * H + y) * W + (lo + t)) * C + c;
      ss += (int)src[sidx] * k[t];
    }
    int v = ss >> RS_PRECISION_BITS;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[i] = (uint8_t)v;
  }
}

// ImagingScaleAffine (Geometry.c) source indices for NEAREST over a whole axis: xo = a0/2, then xo += a0 per step (sequential,
// as Pillow accumulates it), index = xo < 0 ? -1 : (int)xo
__global__ void nearest_index_kernel(int in_size, int out_size, int* __restrict__ idx) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double a0 = (double)in_size / (double)out_size;
  double xo = a0 * 0.5;
  for (int x = 0; x < out_size; ++x) {
    int xin = xo < 0.0 ? -1 : (int)xo;
    xin = xin < 0 ? 0 : (xin > in_size - 1 ? in_size - 1 : xin);
    idx[x] = xin;
    xo += a0;
  }
}

__global__ __launch_bounds__(256) void resize_nearest_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                const int* __restrict__ ix, const int* __restrict__ iy, int N, int H,
                                                                int W, int OH, int OW) {
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW);
    const int64_t r = i / OW;
    const int y = (int)(r % OH), n = (int)(r / OH);
    dst[i] = src[((size_t)n * H + iy[y]) * W + ix[x]];
  }
}

static int resize_ksize(int in_size, int out_size) {
  double fs = (double)in_size / (double)out_size;
  if (fs < 1.0) fs = 1.0;
  return (int)ceil(1.0 * fs) * 2 + 1;
}

__global__ __launch_bounds__(256) void relabel_u8_kernel(const uint8_t* __restrict__ src, int64_t* __restrict__ dst, int64_t count,
                                                         int olabel, int nlabel) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = src[i];
    dst[i] = v == olabel ? nlabel : v;
  }
}

__global__ __launch_bounds__(256) void confusion_hist_kernel(const int64_t* __restrict__ gt, const int64_t* __restrict__ pred,
                                                             int64_t count, int n, unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int bins[];  // n*n (n <= 64), else straight to global memory
  const int nn = n * n;
  const bool local = n <= 64;
  if (local) {
    for (int i = threadIdx.x; i < nn; i += blockDim.x) bins[i] = 0;
    __syncthreads();
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = gt[i], b = pred[i];
    if (a < 0 || a >= n || b < 0 || b >= n) continue;
    const int k = (int)a * n + (int)b;
    if (local)
      atomicAdd(&bins[k], 1u);
    else
      atomicAdd(&hist[k], 1ull);
  }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < nn; i += blockDim.x)
      if (bins[i]) atomicAdd(&hist[i], (unsigned long long)bins[i]);
  }
}

}  // namespace

extern "C" int mcdseg_normalize_u8(const uint8_t* src, float* dst, const float* mean, const float* stdv, int32_t N, int32_t H,
                                   int32_t W, int32_t Cs, int32_t C, int32_t c_off, void* stream) {
  MCD_REQUIRE(src && dst && mean && stdv, "normalize_u8: null pointer");
  MCD_REQUIRE(N > 0 && H > 0 && W > 0 && Cs > 0 && Cs <= 8 && c_off >= 0 && c_off + Cs <= C, "normalize_u8: bad dims");
  MCD_REQUIRE(N <= 65535 && (int64_t)H * W < (1ll << 31), "normalize_u8: image too large");
  const int HW = H * W;
  int blocks = ceil_div(HW, 256);
  if (blocks > 1024) blocks = 1024;
  dim3 grid(blocks, N);
  hipStream_t st = (hipStream_t)stream;
  if (Cs == 3)
    hipLaunchKernelGGL(normalize_u8_kernel<3>, grid, dim3(256), 0, st, src, dst, mean, stdv, HW, C, c_off, Cs);
  else if (Cs == 1)
    hipLaunchKernelGGL(normalize_u8_kernel<1>, grid, dim3(256), 0, st, src, dst, mean, stdv, HW, C, c_off, Cs);
  else
    hipLaunchKernelGGL(normalize_u8_kernel<0>, grid, dim3(256), 0, st, src, dst, mean, stdv, HW, C, c_off, Cs);
  MCD_LAUNCH_CHECK("normalize_u8");
  return 0;
}

extern "C" size_t mcdseg_resize_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return 0;
  const size_t tables = ((size_t)OW * (2 + resize_ksize(W, OW)) + (size_t)OH * (2 + resize_ksize(H, OH))) * sizeof(int);
  return tables + 256 + (size_t)N * H * OW * C;  // + the uint8 image between the two passes
}

extern "C" int mcdseg_resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH,
                                         int32_t OW, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(src && dst && workspace, "resize_bilinear_u8: null pointer");
  MCD_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "resize_bilinear_u8: bad dims");
  MCD_REQUIRE(workspace_bytes >= mcdseg_resize_workspace_bytes(N, H, W, C, OH, OW), "resize_bilinear_u8: workspace too small");
  MCD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 3) == 0, "resize_bilinear_u8: workspace must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int ksx = resize_ksize(W, OW), ksy = resize_ksize(H, OH);
  int* bx = (int*)workspace;
  int* kx = bx + 2 * OW;
  int* by = kx + (size_t)OW * ksx;
  int* ky = by + 2 * OH;
  uint8_t* tmp = (uint8_t*)(ky + (size_t)OH * ksy);
  const bool horiz = OW != W, vert = OH != H;
  if (!horiz && !vert) {
    (void)hipMemcpyAsync(dst, src, (size_t)N * H * W * C, hipMemcpyDeviceToDevice, st);
    return 0;
  }
  auto blocks = [](int64_t n) { int64_t b = ceil_div64(n, 256); return (unsigned)(b > 8192 ? 8192 : b); };
  const uint8_t* cur = src;
  if (horiz) {
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(ceil_div(OW, 128)), dim3(128), 0, st, W, OW, ksx, bx, kx);
    uint8_t* out = vert ? tmp : dst;
    hipLaunchKernelGGL(resize_pass_u8_kernel<false>, dim3(blocks((int64_t)N * H * OW * C)), dim3(256), 0, st, cur, out, bx, kx, ksx, N, H, W, C,
                       H, OW);
    cur = out;
  }
  if (vert) {
    const int Wc = horiz ? OW : W;
    hipLaunchKernelGGL(resize_coeffs_kernel, dim3(ceil_div(OH, 128)), dim3(128), 0, st, H, OH, ksy, by, ky);
    hipLaunchKernelGGL(resize_pass_u8_kernel<true>, dim3(blocks((int64_t)N * OH * Wc * C)), dim3(256), 0, st, cur, dst, by, ky, ksy, N, H, Wc,
                       C, OH, Wc);
  }
  MCD_LAUNCH_CHECK("resize_bilinear_u8");
  return 0;
}

extern "C" int mcdseg_resize_nearest_u8(const uint8_t* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(src && dst && workspace, "resize_nearest_u8: null pointer");
  MCD_REQUIRE(N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "resize_nearest_u8: bad dims");
  MCD_REQUIRE(workspace_bytes >= (size_t)(OW + OH) * sizeof(int), "resize_nearest_u8: workspace too small");
  MCD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 3) == 0, "resize_nearest_u8: workspace must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (OW == W && OH == H) {
    (void)hipMemcpyAsync(dst, src, (size_t)N * H * W, hipMemcpyDeviceToDevice, st);
    return 0;
  }
  int* ix = (int*)workspace;
  int* iy = ix + OW;
  hipLaunchKernelGGL(nearest_index_kernel, dim3(1), dim3(64), 0, st, W, OW, ix);
  hipLaunchKernelGGL(nearest_index_kernel, dim3(1), dim3(64), 0, st, H, OH, iy);
  int64_t b = ceil_div64((int64_t)N * OH * OW, 256);
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3((unsigned)b), dim3(256), 0, st, src, dst, ix, iy, N, H, W, OH, OW);
  MCD_LAUNCH_CHECK("resize_nearest_u8");
  return 0;
}

extern "C" int mcdseg_relabel_u8(const uint8_t* src, int64_t* dst, int64_t count, int32_t olabel, int32_t nlabel, void* stream) {
  MCD_REQUIRE(src && dst, "relabel_u8: null pointer");
  MCD_REQUIRE(count > 0, "relabel_u8: bad count");
  int64_t blocks = ceil_div64(count, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(relabel_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, count, olabel, nlabel);
  MCD_LAUNCH_CHECK("relabel_u8");
  return 0;
}

extern "C" int mcdseg_confusion_hist(const int64_t* gt, const int64_t* pred, int64_t count, int32_t n, int64_t* hist, void* stream) {
  MCD_REQUIRE(gt && pred && hist, "confusion_hist: null pointer");
  MCD_REQUIRE(count > 0 && n > 0 && n <= 4096, "confusion_hist: bad dims");
  int64_t blocks = ceil_div64(count, 256 * 16);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const size_t lds = n <= 64 ? (size_t)n * n * sizeof(unsigned int) : 0;
  hipLaunchKernelGGL(confusion_hist_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, gt, pred, count, n,
                     (unsigned long long*)hist);
  MCD_LAUNCH_CHECK("confusion_hist");
  return 0;
}
