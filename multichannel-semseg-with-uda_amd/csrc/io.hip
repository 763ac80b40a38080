// Either side of the train/test step: the device part of the input pipeline and of the evaluation.
//
//  * normalize_u8   -- ToTensor() + Normalize(mean, std) of the reference's image transform (transform.py:302-315):
//                      uint8 HWC (PIL / numpy layout, as it comes out of the loader's pinned buffer) -> fp32 NCHW,
//                      ((u/255) - mean[c]) / std[c] with IEEE fp32 division, the arithmetic torchvision performs.
//                      RGB and HHA are two source images written into channel ranges of one 6-channel batch.
//  * relabel_u8     -- ToLabel() + ReLabel(255 -> n_class-1) of the label transform (transform.py:21-48, 319-325).
//  * confusion_hist -- fast_hist of eval.py:21-23: bincount(n*gt + pred) over pixels whose gt lies in [0, n).
// All three are single-pass HBM streams; the histogram accumulates in LDS (32-bit integer atomics, exact) and flushes
// with 64-bit integer atomics, so the result does not depend on the order of execution.
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
