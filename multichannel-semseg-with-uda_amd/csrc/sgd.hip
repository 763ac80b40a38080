// SGD with momentum and weight decay on flat fp32 buffers (torch.optim.SGD as configured by
// models/model_util.py:289-292: dampening 0, no Nesterov).  One streaming pass: reads p, g, v and
// writes p, v (20 B per parameter).  grad_scale folds the 1/world_size of the data-parallel
// all-reduce into the same pass.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                           int64_t n4, int64_t n, float lr, float mu, float wd, float gs) {
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p4[i];
    const float4 gg = g4[i];
    float4 vv = v4[i];
    vv.x = fmaf(mu, vv.x, fmaf(wd, pp.x, gg.x * gs));
    vv.y = fmaf(mu, vv.y, fmaf(wd, pp.y, gg.y * gs));
    vv.z = fmaf(mu, vv.z, fmaf(wd, pp.z, gg.z * gs));
    vv.w = fmaf(mu, vv.w, fmaf(wd, pp.w, gg.w * gs));
    pp.x = fmaf(-lr, vv.x, pp.x);
    pp.y = fmaf(-lr, vv.y, pp.y);
    pp.z = fmaf(-lr, vv.z, pp.z);
    pp.w = fmaf(-lr, vv.w, pp.w);
    v4[i] = vv;
    p4[i] = pp;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float vv = fmaf(mu, v[i], fmaf(wd, p[i], g[i] * gs));
    v[i] = vv;
    p[i] = fmaf(-lr, vv, p[i]);
  }
}

}  // namespace

extern "C" int mcdseg_sgd_momentum_flat(float* p, const float* g, float* v, int64_t n, float lr, float momentum, float weight_decay,
                                        float grad_scale, void* stream) {
  MCD_REQUIRE(p && g && v && n >= 0, "sgd_momentum_flat: bad arguments");
  if (n == 0) return 0;
  const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  int64_t blocks = ceil_div64(n4 > 0 ? n4 : n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, v, n4, n, lr, momentum,
                     weight_decay, grad_scale);
  MCD_LAUNCH_CHECK("sgd_momentum_flat");
  return 0;
}
