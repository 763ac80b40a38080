// Late-fusion operators of the MFNet classifiers other than the plain sum (models/fusion.py:6-50) and the loss that
// goes with the gated ones (ProbCrossEntropyLoss2d, loss.py:16-30).  All HBM-bound single passes.
//
//  gate_mix      out = x1*s + x2*(1-s),  s = sigmoid(g)                         (GateFusion.forward, fusion.py:19-22)
//                dx1 = dy*s ; dx2 = dy*(1-s) ; dg = dy*(x1-x2)*s*(1-s)
//  softmax_ch    y = softmax over the channel axis of NCHW                      (F.softmax, fusion.py:13-15)
//                dx = y*(dy - sum_c dy*y)
//  prob_nll      loss = sum_i w[y_i]*(-log p[y_i]) / sum_i w[y_i]                (NLLLoss2d(log(p)), loss.py:30)
//                dp[n,c,i] = (c == y_i) ? -w[y_i] / (p[y_i] * sum w) : 0
#include "common.h"

namespace {


__global__ __launch_bounds__(256) void gate_mix_fwd_kernel(const float4* __restrict__ x1, const float4* __restrict__ x2,
                                                           const float4* __restrict__ g, float4* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = x1[i], b = x2[i], gg = g[i];
    float4 o;
    float s;
    s = 1.0f / (1.0f + expf(-gg.x)); o.x = a.x * s + b.x * (1.0f - s);
    s = 1.0f / (1.0f + expf(-gg.y)); o.y = a.y * s + b.y * (1.0f - s);
    s = 1.0f / (1.0f + expf(-gg.z)); o.z = a.z * s + b.z * (1.0f - s);
    s = 1.0f / (1.0f + expf(-gg.w)); o.w = a.w * s + b.w * (1.0f - s);
    out[i] = o;
  }
}

__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ x1,
                                                           const float4* __restrict__ x2, const float4* __restrict__ g,
                                                           float4* __restrict__ dx1, float4* __restrict__ dx2,
                                                           float4* __restrict__ dg, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 d = dy[i], a = x1[i], b = x2[i], gg = g[i];
    float4 o1, o2, og;
    float s;
#define MCD_GATE_LANE(f)                         \
    s = 1.0f / (1.0f + expf(-gg.f));             \
    o1.f = d.f * s;                              \
    o2.f = d.f * (1.0f - s);                     \
    og.f = d.f * (a.f - b.f) * (s * (1.0f - s));
    MCD_GATE_LANE(x) MCD_GATE_LANE(y) MCD_GATE_LANE(z) MCD_GATE_LANE(w)
#undef MCD_GATE_LANE
    dx1[i] = o1;
    dx2[i] = o2;
    dg[i] = og;
  }
}

// one lane = one pixel, channels walked with stride HW (every per-channel access of a wave is a contiguous 256-B run)
template <int NCMAX>
__global__ __launch_bounds__(256) void softmax_ch_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const float* xi = x + (size_t)n * C * HW + p;
  float* yi = y + (size_t)n * C * HW + p;
  float v[NCMAX];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
    if (c < C) {
      v[c] = xi[(size_t)c * HW];
      m = fmaxf(m, v[c]);
    }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
    if (c < C) {
      v[c] = expf(v[c] - m);
      s += v[c];
    }
  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
    if (c < C) yi[(size_t)c * HW] = v[c] * inv;
}

template <int NCMAX>
__global__ __launch_bounds__(256) void softmax_ch_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             float* __restrict__ dx, int C, int HW) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const size_t base = (size_t)n * C * HW + p;
  float yy[NCMAX], dd[NCMAX];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
    if (c < C) {
      yy[c] = y[base + (size_t)c * HW];
      dd[c] = dy[base + (size_t)c * HW];
      dot += yy[c] * dd[c];
    }
#pragma unroll
  for (int c = 0; c < NCMAX; ++c)
    if (c < C) dx[base + (size_t)c * HW] = yy[c] * (dd[c] - dot);
}

// pass 1: per-block partial (sum w*(-log p), sum w) in fp64
__global__ __launch_bounds__(256) void prob_nll_partial_kernel(const float* __restrict__ p, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ weight, int64_t ignore_index, int C,
                                                               int HW, int64_t npix, double* __restrict__ part) {
  double ls = 0.0, ws = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t y = labels[i];
    if (y == ignore_index || y < 0 || y >= C) continue;
    const int64_t n = i / HW;
    const float w = weight ? weight[y] : 1.0f;
    const float pv = p[((size_t)n * C + y) * HW + (i - n * HW)];
    ls += (double)(w * -logf(pv));
    ws += (double)w;
  }
  ls = wave_sum_d(ls);
  ws = wave_sum_d(ws);
  __shared__ double sl[4], sw[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sl[wave] = ls;
    sw[wave] = ws;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
    part[2 * blockIdx.x + 1] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
  }
}

__global__ void prob_nll_finalize_kernel(const double* __restrict__ part, int blocks, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ls = 0.0, ws = 0.0;
  for (int b = 0; b < blocks; ++b) {
    ls += part[2 * b];
    ws += part[2 * b + 1];
  }
  loss[0] = (float)(ls / ws);  // size_average=True
  loss[1] = (float)ls;         // size_average=False
  loss[2] = (float)ws;
}

// pass 2: dense gradient (zeros off the label channel), scaled by 1/sum w read from the device
__global__ __launch_bounds__(256) void prob_nll_grad_kernel(const float* __restrict__ p, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ weight, int64_t ignore_index,
                                                            const float* __restrict__ loss, int size_average, int C, int HW,
                                                            float* __restrict__ grad) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  const int64_t y = labels[(size_t)n * HW + i];
  const bool live = !(y == ignore_index || y < 0 || y >= C);
  const float scale = size_average ? 1.0f / loss[2] : 1.0f;
  const size_t base = (size_t)n * C * HW + i;
  for (int c = 0; c < C; ++c) {
    float gv = 0.f;
    if (live && c == (int)y) gv = -(weight ? weight[y] : 1.0f) * scale / p[base + (size_t)c * HW];
    grad[base + (size_t)c * HW] = gv;
  }
}

int stream_blocks(int64_t n) {
  int64_t b = ceil_div64(n, 256 * 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mcdseg_gate_mix_fwd(const float* x1, const float* x2, const float* g, float* out, int64_t n, void* stream) {
  MCD_REQUIRE(x1 && x2 && g && out, "gate_mix_fwd: null pointer");
  MCD_REQUIRE(n > 0 && (n & 3) == 0, "gate_mix_fwd: element count must be a positive multiple of 4 (got %lld)", (long long)n);
  hipLaunchKernelGGL(gate_mix_fwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x1,
                     (const float4*)x2, (const float4*)g, (float4*)out, n / 4);
  MCD_LAUNCH_CHECK("gate_mix_fwd");
  return 0;
}

extern "C" int mcdseg_gate_mix_bwd(const float* dy, const float* x1, const float* x2, const float* g, float* dx1, float* dx2,
                                   float* dg, int64_t n, void* stream) {
  MCD_REQUIRE(dy && x1 && x2 && g && dx1 && dx2 && dg, "gate_mix_bwd: null pointer");
  MCD_REQUIRE(n > 0 && (n & 3) == 0, "gate_mix_bwd: element count must be a positive multiple of 4 (got %lld)", (long long)n);
  hipLaunchKernelGGL(gate_mix_bwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy,
                     (const float4*)x1, (const float4*)x2, (const float4*)g, (float4*)dx1, (float4*)dx2, (float4*)dg, n / 4);
  MCD_LAUNCH_CHECK("gate_mix_bwd");
  return 0;
}

extern "C" int mcdseg_softmax_ch_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t HW, void* stream) {
  MCD_REQUIRE(x && y, "softmax_ch_fwd: null pointer");
  MCD_REQUIRE(N > 0 && N <= 65535 && C > 0 && C <= 64 && HW > 0, "softmax_ch_fwd: bad dims (C <= 64)");
  dim3 grid(ceil_div(HW, 256), N);
  if (C <= 24)
    hipLaunchKernelGGL(softmax_ch_fwd_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW);
  else if (C <= 48)
    hipLaunchKernelGGL(softmax_ch_fwd_kernel<48>, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW);
  else
    hipLaunchKernelGGL(softmax_ch_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, y, C, HW);
  MCD_LAUNCH_CHECK("softmax_ch_fwd");
  return 0;
}

extern "C" int mcdseg_softmax_ch_bwd(const float* dy, const float* y, float* dx, int32_t N, int32_t C, int32_t HW, void* stream) {
  MCD_REQUIRE(dy && y && dx, "softmax_ch_bwd: null pointer");
  MCD_REQUIRE(N > 0 && N <= 65535 && C > 0 && C <= 64 && HW > 0, "softmax_ch_bwd: bad dims (C <= 64)");
  dim3 grid(ceil_div(HW, 256), N);
  if (C <= 24)
    hipLaunchKernelGGL(softmax_ch_bwd_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, dx, C, HW);
  else if (C <= 48)
    hipLaunchKernelGGL(softmax_ch_bwd_kernel<48>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, dx, C, HW);
  else
    hipLaunchKernelGGL(softmax_ch_bwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, dx, C, HW);
  MCD_LAUNCH_CHECK("softmax_ch_bwd");
  return 0;
}

extern "C" size_t mcdseg_prob_nll_workspace_bytes(int32_t N, int32_t HW) {
  if (N <= 0 || HW <= 0) return 0;
  return (size_t)stream_blocks((int64_t)N * HW) * 2 * sizeof(double);
}

extern "C" int mcdseg_prob_nll(const float* p, const int64_t* labels, const float* weight, int64_t ignore_index,
                               int32_t size_average, float* grad, float* loss, int32_t N, int32_t C, int32_t HW, void* workspace,
                               size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(p && labels && loss && workspace, "prob_nll: null pointer");
  MCD_REQUIRE(N > 0 && N <= 65535 && C > 0 && HW > 0, "prob_nll: bad dims");
  const int64_t npix = (int64_t)N * HW;
  const int blocks = stream_blocks(npix);
  MCD_REQUIRE(workspace_bytes >= (size_t)blocks * 2 * sizeof(double), "prob_nll: workspace too small");
  MCD_REQUIRE(((uintptr_t)workspace & 7) == 0, "prob_nll: workspace must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(prob_nll_partial_kernel, dim3(blocks), dim3(256), 0, st, p, labels, weight, ignore_index, C, HW, npix,
                     (double*)workspace);
  MCD_LAUNCH_CHECK("prob_nll_partial");
  hipLaunchKernelGGL(prob_nll_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)workspace, blocks, loss);
  MCD_LAUNCH_CHECK("prob_nll_finalize");
  if (grad) {
    hipLaunchKernelGGL(prob_nll_grad_kernel, dim3(ceil_div(HW, 256), N), dim3(256), 0, st, p, labels, weight, ignore_index,
                       (const float*)loss, size_average, C, HW, grad);
    MCD_LAUNCH_CHECK("prob_nll_grad");
  }
  return 0;
}
