// BatchNorm2d pieces around the conv kernels (all HBM-bound streaming / reduction kernels).
//
//   stats_finalize : merge the conv epilogue's (count, mean, M2) rows in fp64 -> mean, rstd, running stats
//   apply          : y = relu(gamma*(z-mean)*rstd + beta + residual)             float4 streaming
//   bwd_reduce     : dgamma = sum dy_m*xhat, dbeta = sum dy_m  (dy_m = dy masked by y>0)
//   bwd_apply      : dz = gamma*rstd*(dy_m - dbeta/n - xhat*dgamma/n), dres = dy_m  float4 streaming
//
// Reference semantics: nn.BatchNorm2d(eps=1e-5, momentum=0.1) in train mode normalises with the biased
// batch variance and moves running_var with the unbiased one (models/drn.py:34,38,129,179,202);
// ReLU is in place after the residual add (models/drn.py:48,55-57).
#include "options.h"
#include "split.h"

namespace {

// ------------------------------------------------------------------------------------------------
// finalize, two stages.  Stage 1: grid (C/32, S); a block of 32 channels x 8 row-groups folds its slice of the
// partial rows into fp64 (sum n, sum n*mean, sum M2 + n*mean^2).  Stage 2: one thread per channel merges the
// S slices: mean = S1/N, var = (Q - S1^2/N)/N -- in fp64 the subtraction costs ~1e-16 * mean^2/var, far below
// fp32 resolution, while the fp32 data itself was centred per wave (shifted form) in the conv epilogue.
int stats_slices(int64_t rows, int C) {
  // enough blocks to fill the chip: ceil(C/32) channel blocks x S row slices ~ 1024, at least 8 rows per slice
  const int cb = ceil_div(C, 32);
  int64_t s = ceil_div64(1024, cb);
  if (s > rows / 8) s = rows / 8;
  return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ part, int64_t rows, int C, int Mp,
                                                               double* __restrict__ out, float* __restrict__ y_bound) {
  __shared__ double sh[3][8][33];
  if (y_bound != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *y_bound = 0.f;  // finalize takes an atomic max
  const int cx = threadIdx.x & 31;
  const int g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const int S = gridDim.y;
  const int64_t per = (rows + S - 1) / S;
  const int64_t r0 = blockIdx.y * per;
  int64_t r1 = r0 + per;
  if (r1 > rows) r1 = rows;
  double n = 0.0, s1 = 0.0, q = 0.0;
  if (c < C) {
    for (int64_t r = r0 + g; r < r1; r += 8) {
      const float* row = part + (size_t)r * 3 * Mp;
      const double cnt = (double)row[c];
      const double mu = (double)row[Mp + c];
      n += cnt;
      s1 += cnt * mu;
      q += (double)row[2 * (size_t)Mp + c] + cnt * mu * mu;
    }
  }
  sh[0][g][cx] = n;
  sh[1][g][cx] = s1;
  sh[2][g][cx] = q;
  __syncthreads();
  if (g < 3 && c < C) {
    double t = 0.0;
    for (int k = 0; k < 8; ++k) t += sh[g][k][cx];
    out[((size_t)blockIdx.y * 3 + g) * C + c] = t;
  }
}

// Stage 2: one wave per channel (4 channels per block): lanes stride over the S slices, fp64 wave reduction.  Also forms the
// bound of the tensor bn_apply is about to write (see mcdseg.h): an integer atomic max on the bit pattern (non-negative floats
// order like their bits; the scalar was zeroed by stage 1).
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const double* __restrict__ sl, int S, int C,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, int64_t* nbt,
                                                                float momentum, float eps, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ res_bound, float* __restrict__ y_bound, int updates) {
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += updates;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;  // wave-uniform
  double n = 0.0, s1 = 0.0, q = 0.0;
  for (int k = lane; k < S; k += 64) {
    n += sl[((size_t)k * 3 + 0) * C + c];
    s1 += sl[((size_t)k * 3 + 1) * C + c];
    q += sl[((size_t)k * 3 + 2) * C + c];
  }
  n = wave_sum_d(n);
  s1 = wave_sum_d(s1);
  q = wave_sum_d(q);
  if (lane != 0) return;
  const double mean = n > 0.0 ? s1 / n : 0.0;
  double m2 = q - s1 * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = n > 0.0 ? m2 / n : 0.0;
  mean_out[c] = (float)mean;
  rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    float rm = running_mean[c], rv = running_var[c];
    for (int u = 0; u < updates; ++u) {  // one update per forward pass this launch stands for (rounded to fp32 each time, as separate launches would)
      rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mean);
      rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
    }
    running_mean[c] = rm;
    running_var[c] = rv;
  }
  if (y_bound != nullptr) {
    float bound = fabsf(gamma[c]) * (float)sqrt(n > 1.0 ? n - 1.0 : 1.0) * 1.0001f + fabsf(beta[c]);
    bound += res_bound ? *res_bound : 0.f;
    if (!(bound == bound)) bound = __uint_as_float(0x7FC00000u);  // NaN parameters -> non-finite bound
    atomicMax(reinterpret_cast<unsigned*>(y_bound), __float_as_uint(bound) & 0x7FFFFFFFu);
  }
}

// Both stages in ONE launch, for the layers whose partial rows are few (the 1/8-resolution maps: 480 rows at BASELINE config 2, three
// quarters of a network's BatchNorm layers): one wave per channel does what stage 1's blocks and stage 2's wave do for that channel, in
// the SAME order -- lane L owns the slices L, L + 64, ...; per slice the eight row-group sums are formed row by row and added in group
// order, exactly as stage 1's threads and its shared-memory fold do; then stage 2's wave reduction -- so mean, rstd and the running
// statistics are bit for bit those of the two launches.  The output bound depends on gamma, beta and the pixel count only: the first
// wave of the grid forms its maximum over all channels directly (a max is order-independent), no zeroing launch and no atomics.
__global__ __launch_bounds__(256) void bn_stats_one_kernel(const float* __restrict__ part, int64_t rows, int S, int C, int Mp,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           int64_t* nbt, float momentum, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res_bound,
                                                           float* __restrict__ y_bound, int updates) {
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += updates;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;  // wave-uniform
  const int64_t per = (rows + S - 1) / S;
  double n = 0.0, s1 = 0.0, q = 0.0;
  for (int k = lane; k < S; k += 64) {
    const int64_t r0 = k * per;
    int64_t r1 = r0 + per;
    if (r1 > rows) r1 = rows;
    // eight rows at a time -- one per row group, 24 independent loads in flight -- each added to its own group's sums, so that a
    // group still sees its rows r0 + g, r0 + g + 8, ... in order
    double gn[8], gs[8], gq[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) gn[g] = gs[g] = gq[g] = 0.0;
    for (int64_t rb = r0; rb < r1; rb += 8) {
      float cnt[8], mu[8], m2[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float* row = part + (size_t)(rb + g < r1 ? rb + g : r0) * 3 * Mp;
        cnt[g] = row[c];
        mu[g] = row[Mp + c];
        m2[g] = row[2 * (size_t)Mp + c];
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (rb + g < r1) {
          const double dc = (double)cnt[g], dm = (double)mu[g];
          gn[g] += dc;
          gs[g] += dc * dm;
          gq[g] += (double)m2[g] + dc * dm * dm;
        }
      }
    }
    double tn = 0.0, ts = 0.0, tq = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      tn += gn[g];
      ts += gs[g];
      tq += gq[g];
    }
    n += tn;
    s1 += ts;
    q += tq;
  }
  n = wave_sum_d(n);
  s1 = wave_sum_d(s1);
  q = wave_sum_d(q);
  n = __shfl(n, 0);  // (every lane of the first wave needs the count for the bound below)
  if (lane == 0) {
    const double mean = n > 0.0 ? s1 / n : 0.0;
    double m2 = q - s1 * mean;
    if (m2 < 0.0) m2 = 0.0;
    const double var = n > 0.0 ? m2 / n : 0.0;
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
      float rm = running_mean[c], rv = running_var[c];
      for (int u = 0; u < updates; ++u) {
        rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mean);
        rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
      }
      running_mean[c] = rm;
      running_var[c] = rv;
    }
  }
  if (y_bound != nullptr && c == 0) {  // the first wave of the grid: max over all channels of what stage 2 feeds its atomic max
    const float sq = (float)sqrt(n > 1.0 ? n - 1.0 : 1.0);
    const float rb = res_bound ? *res_bound : 0.f;
    unsigned m = 0u;
    for (int cc = lane; cc < C; cc += 64) {
      float bound = fabsf(gamma[cc]) * sq * 1.0001f + fabsf(beta[cc]);
      bound += rb;
      if (!(bound == bound)) bound = __uint_as_float(0x7FC00000u);
      const unsigned bits = __float_as_uint(bound) & 0x7FFFFFFFu;
      m = bits > m ? bits : m;
    }
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)m, o);
      m = other > m ? other : m;
    }
    if (lane == 0) *y_bound = __uint_as_float(m);
  }
}

// rows up to which the one-launch form is used (option BN_STATS_ONE: tests compare the two forms; 0 = never)
int64_t stats_one_rows() { return mcd_opt(MCD_OPT_BN_STATS_ONE); }

__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    mean[c] = rm[c];
    rstd[c] = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
  }
}

// eval-mode BatchNorm as a per-channel affine map (folded into the conv epilogue at inference):
//   scale = gamma / sqrt(running_var + eps), shift = beta + (conv_bias - running_mean) * scale
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                      const float* __restrict__ rv, const float* __restrict__ conv_bias, int C, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float a = gamma[c] * (float)(1.0 / sqrt((double)rv[c] + (double)eps));
    scale[c] = a;
    shift[c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * a;
  }
}

// ------------------------------------------------------------------------------------------------
// apply: grid.x = plane (n*C + c), grid.y = chunk of the plane; float4 when HW % 4 == 0.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ res,
                                                       float* __restrict__ y, int C, int HW, int relu) {
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float a = gamma[c] * rstd[c];
  const float b = beta[c] - mean[c] * a;
  const size_t base = (size_t)plane * HW;
  if (VEC) {
    const int n4 = HW >> 2;
    const float4* z4 = reinterpret_cast<const float4*>(z + base);
    const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
    float4* y4 = reinterpret_cast<float4*>(y + base);
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n4; i += gridDim.y * blockDim.x) {
      float4 v = z4[i];
      v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
      if (r4) {
        const float4 q = r4[i];
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      y4[i] = v;
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x) {
      float v = fmaf(z[base + i], a, b);
      if (res) v += res[base + i];
      if (relu) v = fmaxf(v, 0.f);
      y[base + i] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward reduce: grid = (C, S).  Split s covers work items (n, chunk) with item = s, s+S, ...
struct BwdPlan {
  int cpp;    // chunks per plane
  int chunk;  // elements per chunk (multiple of 4 when HW % 4 == 0)
  int S;      // splits per channel
};

BwdPlan bwd_plan(int N, int C, int HW) {
  BwdPlan pl;
  int64_t want = ceil_div64(2048, C);  // splits per channel to fill the chip
  int cpp = (int)ceil_div64(want, N);
  const int max_cpp = HW / 2048 > 0 ? HW / 2048 : 1;
  if (cpp > max_cpp) cpp = max_cpp;
  if (cpp < 1) cpp = 1;
  int chunk = ceil_div(HW, cpp);
  chunk = round_up(chunk, 256);  // (whole 256-pixel blocks: a wave of the float4 loop then walks exactly one block of the ReLU bit-plane)
  pl.cpp = ceil_div(HW, chunk);
  pl.chunk = chunk;
  int64_t items = (int64_t)N * pl.cpp;
  pl.S = (int)(items < want ? items : want);
  if (pl.S < 1) pl.S = 1;
  return pl;
}

// ReLU bit-plane of a group WITH residual (round 5).  Such a group's backward pass needs y > 0 and cannot recompute it from z (the
// residual enters), so until round 5 both backward kernels read the fp32 y for it: 4 of their 12 / 20 bytes per element.  The forward
// apply kernel now also writes the mask -- per (image, channel) and block of 256 pixels four 64-bit words, bit l of word j = (y > 0) of
// pixel 256 blk + 4 l + j: the ballots of a wave of the four-pixel kernels, whose lane l owns pixels 4 l .. 4 l + 3 of the block -- and
// the backward kernels read 1 bit where they read 32.  The same mask, bit for bit: y > 0 evaluated once, on the value that was stored.
__device__ __forceinline__ unsigned long long mcd_readlane64(unsigned long long v, int src_lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), src_lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src_lane);
  return ((unsigned long long)hi << 32) | lo;
}

// MASK (compile-time: with the four forms behind run-time branches in one loop the compiler carries the register hazards of one
// form's loads into the others and waits for every load where it is issued): 0 no ReLU, 1 y > 0 from the fp32 y, 2 from z
// (mgamma / mbeta: a group without residual), 3 from the bit-plane (rmask).
template <bool VEC, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ part, int N,
                                                            int C, int HW, int relu, int cpp, int chunk,
                                                            float* __restrict__ dz_bound, const float* __restrict__ mgamma,
                                                            const float* __restrict__ mbeta,
                                                            const unsigned long long* __restrict__ rmask, int nblk) {
  const int c = blockIdx.x;
  const int S = gridDim.y;
  if (dz_bound != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dz_bound = 0.f;  // finalize: atomic max
  const float mu = z ? mean[c] : 0.f;
  const float rs = z ? rstd[c] : 0.f;
  // ReLU mask without reading y (mbeta != NULL; a group WITHOUT residual): y > 0 <=> fma(z, a, b) > 0 with the forward kernels'
  // own a = gamma rstd, b = beta - mean a (bn_apply_kernel / bn_apply_cb_kernel: the same expressions, bit for bit)
  constexpr bool zm = MASK == 2;
  const float ma = zm ? mgamma[c] * rs : 0.f;
  const float mb = zm ? mbeta[c] - mu * ma : 0.f;
  float s_dy = 0.f, s_dyx = 0.f, m_g = 0.f;
  const int items = N * cpp;
  for (int item = blockIdx.y; item < items; item += S) {
    const int n = item / cpp;
    const int ch = item - n * cpp;
    const int e0 = ch * chunk;
    int e1 = e0 + chunk;
    if (e1 > HW) e1 = HW;
    const size_t base = ((size_t)n * C + c) * HW;
    if (VEC) {
      const float4* dy4 = reinterpret_cast<const float4*>(dy + base);
      const float4* y4 = reinterpret_cast<const float4*>(y ? y + base : nullptr);
      const float4* z4 = reinterpret_cast<const float4*>(z ? z + base : nullptr);
      const int lane = threadIdx.x & 63;
      // (stepped per WAVE: ib is the wave's first float4 -- a multiple of 64, chunks being whole 256-pixel blocks -- so that the four
      // words of the wave's block of the ReLU bit-plane are loaded once, by lanes 0-3; per thread the same elements in the same order
      // as a plain strided loop)
      // Four steps of the strided loop at a time, every load of the four issued before the first sum: a thread has 2 x 4 float4 in
      // flight instead of 2 (a 256-channel layer of the benchmark gives a thread ten steps in all).  Worth 2 % of the kernel in the
      // benchmark step -- at 8 waves per SIMD the other waves already covered most of the latency.  Same elements in the same order per
      // thread: the same sums, bit for bit.
      const int nq = e1 >> 2;
      for (int ib = (e0 >> 2) + (threadIdx.x & ~63); ib < nq; ib += 4 * 256) {
        float4 gq[4], vq[4], oq[4];
        unsigned long long mq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // (no divergent branch around a load -- the compiler waits at the join of each: clamped addresses instead; steps past the
          // end repeat the last element and are skipped below)
          const int ibu = ib + u * 256 < nq ? ib + u * 256 : ((nq - 1) & ~63);  // wave-uniform
          const int i = ibu + lane < nq ? ibu + lane : nq - 1;
          vq[u] = oq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          mq[u] = 0ull;
          if (MASK == 3) mq[u] = rmask[(((size_t)n * C + c) * nblk + (ibu >> 6)) * 4 + (lane & 3)];
          gq[u] = dy4[i];
          if (z) vq[u] = z4[i];
          if (MASK == 1) oq[u] = y4[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ibu = ib + u * 256;
          if (ibu >= nq) break;  // wave-uniform
          unsigned long long w0 = 0ull, w1 = 0ull, w2 = 0ull, w3 = 0ull;
          if (MASK == 3) {  // (wave-uniform control flow up to here: every lane takes part in the broadcasts)
            w0 = mcd_readlane64(mq[u], 0); w1 = mcd_readlane64(mq[u], 1); w2 = mcd_readlane64(mq[u], 2); w3 = mcd_readlane64(mq[u], 3);
          }
          if (ibu + lane >= nq) continue;
          float4 g = gq[u];
          const float4 v = vq[u];
          if (MASK == 3) {
            g.x = ((w0 >> lane) & 1ull) ? g.x : 0.f; g.y = ((w1 >> lane) & 1ull) ? g.y : 0.f;
            g.z = ((w2 >> lane) & 1ull) ? g.z : 0.f; g.w = ((w3 >> lane) & 1ull) ? g.w : 0.f;
          } else if (MASK == 2) {
            g.x = fmaf(v.x, ma, mb) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, ma, mb) > 0.f ? g.y : 0.f;
            g.z = fmaf(v.z, ma, mb) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, ma, mb) > 0.f ? g.w : 0.f;
          } else if (MASK == 1) {
            const float4 o = oq[u];
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
          }
          s_dy += (g.x + g.y) + (g.z + g.w);
          m_g = fmaxf(m_g, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
          if (z) {
            s_dyx += (g.x * ((v.x - mu) * rs) + g.y * ((v.y - mu) * rs)) + (g.z * ((v.z - mu) * rs) + g.w * ((v.w - mu) * rs));
          }
        }
      }
    } else {
      for (int i = e0 + threadIdx.x; i < e1; i += 256) {
        float g = dy[base + i];
        if (MASK == 2) {
          if (!(fmaf(z[base + i], ma, mb) > 0.f)) g = 0.f;
        } else if (MASK == 1 && !(y[base + i] > 0.f)) {
          g = 0.f;
        }
        s_dy += g;
        m_g = fmaxf(m_g, fabsf(g));
        if (z) s_dyx += g * ((z[base + i] - mu) * rs);
      }
    }
  }
  __shared__ float sh[3][4];
  s_dy = wave_sum(s_dy);
  s_dyx = wave_sum(s_dyx);
  for (int o = 1; o < 64; o <<= 1) m_g = fmaxf(m_g, __shfl_xor(m_g, o));  // fmaxf drops NaN: a NaN gradient shows in the sums
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s_dy;
    sh[1][threadIdx.x >> 6] = s_dyx;
    sh[2][threadIdx.x >> 6] = m_g;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((size_t)blockIdx.y * 3 + 0) * C + c] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    part[((size_t)blockIdx.y * 3 + 1) * C + c] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    part[((size_t)blockIdx.y * 3 + 2) * C + c] = fmaxf(fmaxf(sh[2][0], sh[2][1]), fmaxf(sh[2][2], sh[2][3]));
  }
}

// the instantiation for (vectorised?, mask form)
template <typename... A>
void launch_bwd_reduce(bool vec, int mask, dim3 grid, hipStream_t st, A... a) {
#define MCD_RED(V, M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<V, M>), grid, dim3(256), 0, st, a...)
  if (vec) {
    if (mask == 0) MCD_RED(true, 0); else if (mask == 1) MCD_RED(true, 1); else if (mask == 2) MCD_RED(true, 2); else MCD_RED(true, 3);
  } else {
    if (mask == 0) MCD_RED(false, 0); else if (mask == 1) MCD_RED(false, 1); else MCD_RED(false, 2);
  }
#undef MCD_RED
}

// dgamma / dbeta from the S partial rows (fp64) and, when asked, the bound of the dz tensor bn_bwd_apply is about to write
// (mcdseg.h): |dz| <= |gamma rstd| (max|g| + |dbeta|/n + sqrt(n-1) |dgamma|/n) since |xhat| <= sqrt(n-1).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int S, int C, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, const float* __restrict__ gamma,
                                                              const float* __restrict__ rstd, float n_total, int train,
                                                              float* __restrict__ dz_bound) {
  // one wave per channel: lanes stride over the S partial rows (hundreds for the full-resolution layers), fixed-order fp64 sums
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;  // wave-uniform
  double a = 0.0, b = 0.0;
  float mg = 0.f;
  for (int s = lane; s < S; s += 64) {
    a += (double)part[((size_t)s * 3 + 0) * C + c];
    b += (double)part[((size_t)s * 3 + 1) * C + c];
    mg = fmaxf(mg, part[((size_t)s * 3 + 2) * C + c]);
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  for (int o = 1; o < 64; o <<= 1) mg = fmaxf(mg, __shfl_xor(mg, o));
  if (lane != 0) return;
  if (dbeta) dbeta[c] = (float)a;
  if (dgamma) dgamma[c] = (float)b;
  if (dz_bound != nullptr) {
    const float ar = fabsf(gamma[c] * rstd[c]);
    float bound = train ? ar * (mg + fabsf((float)a) / n_total + sqrtf(n_total > 1.f ? n_total - 1.f : 1.f) * fabsf((float)b) / n_total) * 1.0001f
                        : ar * mg;
    if (!(bound == bound)) bound = __uint_as_float(0x7FC00000u);  // NaN statistics -> non-finite bound
    atomicMax(reinterpret_cast<unsigned*>(dz_bound), __float_as_uint(bound) & 0x7FFFFFFFu);
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float* __restrict__ dz, float* __restrict__ dres, int N, int C,
                                                           int HW, int relu, int train) {
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float mu = mean[c], rs = rstd[c];
  const float a = gamma[c] * rs;
  const float inv_n = 1.f / ((float)N * (float)HW);
  const float k1 = train ? dbeta[c] * inv_n : 0.f;
  const float k2 = train ? dgamma[c] * inv_n : 0.f;
  const size_t base = (size_t)plane * HW;
  if (VEC) {
    const int n4 = HW >> 2;
    const float4* dy4 = reinterpret_cast<const float4*>(dy + base);
    const float4* y4 = reinterpret_cast<const float4*>(y ? y + base : nullptr);
    const float4* z4 = reinterpret_cast<const float4*>(z + base);
    float4* dz4 = reinterpret_cast<float4*>(dz + base);
    float4* dr4 = dres ? reinterpret_cast<float4*>(dres + base) : nullptr;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n4; i += gridDim.y * blockDim.x) {
      float4 g = dy4[i];
      if (relu) {
        const float4 o = y4[i];
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      if (dr4) dr4[i] = g;
      const float4 v = z4[i];
      float4 o;
      o.x = a * (g.x - k1 - ((v.x - mu) * rs) * k2);
      o.y = a * (g.y - k1 - ((v.y - mu) * rs) * k2);
      o.z = a * (g.z - k1 - ((v.z - mu) * rs) * k2);
      o.w = a * (g.w - k1 - ((v.w - mu) * rs) * k2);
      dz4[i] = o;
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < HW; i += gridDim.y * blockDim.x) {
      float g = dy[base + i];
      if (relu && !(y[base + i] > 0.f)) g = 0.f;
      if (dres) dres[base + i] = g;
      dz[base + i] = a * (g - k1 - ((z[base + i] - mu) * rs) * k2);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// "cb" variants: besides the fp32 tensor, emit its split (policy P of split.h) in the channel-blocked layout
//   cb[piece NP][N][C/8][HW][8 x 16 bit]
// which is what the split convolutions consume as their gathered operand (one 16-B fragment per pixel and 8-channel
// group, consecutive pixels contiguous).  Splitting here -- once, in an HBM-bound kernel whose VALU is idle -- instead
// of inside the convolution's K loop (where every activation is re-split for each of the 9 taps and 4 M-tiles and the
// conversion competes with MFMA issue slots) is worth ~1.3x on the convolution.  One thread = one pixel x 8 channels:
// loads and fp32 stores are pixel-contiguous per channel, the 16-B split stores are contiguous across lanes.
template <class P>
__device__ __forceinline__ void split_store(const float (&v)[8], float inv_scale, typename P::elem* __restrict__ cb, size_t piece_stride,
                                            size_t idx16) {
  typename P::frag pieces[P::NP];
  split_frag<P>(v, inv_scale, pieces);
#pragma unroll
  for (int pc = 0; pc < P::NP; ++pc) *reinterpret_cast<typename P::frag*>(cb + pc * piece_stride + idx16 * 8) = pieces[pc];
}

// the inverse: 8 channels of one pixel back from the companion, value = scale * (sum of the pieces) -- exact for the bf16
// split, the 22 leading bits for the fp16 one
// (piece_stride == 0: the caller reads the leading piece only -- MCDSEG_MATH_F16X1, whose companions may be stored as ONE piece)
template <class P>
__device__ __forceinline__ void join_load(const typename P::elem* __restrict__ cb, float scale, size_t piece_stride, size_t idx16,
                                          float (&v)[8]) {
  typename P::frag pieces[P::NP];
#pragma unroll
  for (int pc = 0; pc < P::NP; ++pc) pieces[pc] = *reinterpret_cast<const typename P::frag*>(cb + pc * piece_stride + idx16 * 8);
  const bool lead_only = piece_stride == 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float t = 0.f;
#pragma unroll
    for (int pc = P::NP - 1; pc >= 1; --pc) t += lead_only ? 0.f : (float)pieces[pc][e];  // smallest piece first: exact
    t += (float)pieces[0][e];
    v[e] = t * scale;
  }
}

// ReLU mask of 8 channels of one pixel from the leading piece of the activation's companion (y > 0 <=> its leading piece > 0,
// except for 0 < y < 2^-25 of the scale, which the piece rounds to zero)
template <class P>
__device__ __forceinline__ unsigned mask_load(const typename P::elem* __restrict__ cb, size_t idx16) {
  const typename P::frag lead = *reinterpret_cast<const typename P::frag*>(cb + idx16 * 8);
  unsigned m = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) m |= ((float)lead[e] > 0.f ? 1u : 0u) << e;
  return m;
}

// pixels per thread of the apply kernels (256 apart).  Measured on the benchmark step: 1 -> 0.069 / 0.081 ms per launch
// (forward / backward apply), 2 -> 0.073 / 0.088, 4 -> 0.082 / 0.097: these passes want many short workgroups, not long ones
#ifndef BN_PIX_ITERS
#define BN_PIX_ITERS 1
#endif

template <class P>
__global__ __launch_bounds__(256) void bn_apply_cb_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ res,
                                                          const typename P::elem* __restrict__ res_cb,
                                                          const float* __restrict__ res_bound, float* __restrict__ y,
                                                          typename P::elem* __restrict__ cb, const float* __restrict__ y_bound, int N,
                                                          int C, int HW, int relu, int res_lead_only) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;  // n * C8 + g
  const int g = ng % C8;
  const int n = ng / C8;
  const float inv_scale = 1.f / operand_scale<P>(y_bound);
  float ca[8], cbeta[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * g + e;
    ca[e] = gamma[c] * rstd[c];
    cbeta[e] = beta[c] - mean[c] * ca[e];
  }
#pragma unroll
  for (int it = 0; it < BN_PIX_ITERS; ++it) {
    const int pix = (blockIdx.x * BN_PIX_ITERS + it) * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
    float r[8];
    if (res_cb != nullptr) join_load<P>(res_cb, operand_scale<P>(res_bound), res_lead_only ? (size_t)0 : (size_t)N * C * HW, (size_t)ng * HW + pix, r);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf(z[base + (size_t)e * HW], ca[e], cbeta[e]);
      if (res) t += res[base + (size_t)e * HW];
      if (res_cb != nullptr) t += r[e];
      if (relu) t = fmaxf(t, 0.f);
      v[e] = t;
      if (y != nullptr) y[base + (size_t)e * HW] = t;  // compact activation storage: the companion is the activation
    }
    split_store<P>(v, inv_scale, cb, (size_t)N * C * HW, (size_t)ng * HW + pix);
  }
}

#ifndef BN_V4_NT
#define BN_V4_NT 256  // threads per workgroup of the four-pixel kernels
#endif
// The same pass with 16-byte accesses to the fp32 planes: one thread = FOUR adjacent pixels x 8 channels (HW % 4 == 0).  The
// thread's four companion units would be 16-byte stores 64 bytes apart; they go through a wave-private LDS image instead, so
// that every store instruction of the wave covers 1 KB of consecutive units.
template <class P>
__device__ __forceinline__ void split_store_x4(const float (&v)[4][8], float inv_scale, typename P::elem* __restrict__ cb,
                                               size_t piece_stride, size_t unit0_wave, int units_wave, typename P::frag* lds_wave) {
  const int lane = threadIdx.x & 63;
  typename P::frag pieces[4][P::NP];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_frag<P>(v[j], inv_scale, pieces[j]);
#pragma unroll
  for (int pc = 0; pc < P::NP; ++pc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) lds_wave[4 * lane + j] = pieces[j][pc];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = 64 * j + lane;
      const typename P::frag q = lds_wave[u];
      if (u < units_wave) *reinterpret_cast<typename P::frag*>(cb + pc * piece_stride + (unit0_wave + u) * 8) = q;
    }
  }
}

// RES (compile-time: behind a run-time branch the residual's loads are issued one channel at a time, each waited for where it is
// issued -- eight HBM round trips in a row per thread; with all 16 loads issued first the forward apply kernels of a benchmark step take
// 6 % less): 0 no residual, 1 fp32 residual, 2 the residual as its companion.
template <class P, int RES>
__global__ __launch_bounds__(BN_V4_NT) void bn_apply_cb_v4_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res,
                                                             const typename P::elem* __restrict__ res_cb,
                                                             const float* __restrict__ res_bound, float* __restrict__ y,
                                                             typename P::elem* __restrict__ cb, const float* __restrict__ y_bound, int N,
                                                             int C, int HW, int relu, int rev,
                                                             unsigned long long* __restrict__ rmask, int res_lead_only) {
  __shared__ typename P::frag lds[BN_V4_NT / 64][256];
  const int C8 = C >> 3;
  // rev: walk the tensor from its END -- the convolution that has just written z did so front to back, so the tail is what the
  // Infinity Cache still holds
  const int ng = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;  // n * C8 + g
  const int bx = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int g = ng % C8;
  const int n = ng / C8;
  const float inv_scale = 1.f / operand_scale<P>(y_bound);
  float ca[8], cbeta[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * g + e;
    ca[e] = gamma[c] * rstd[c];
    cbeta[e] = beta[c] - mean[c] * ca[e];
  }
  const int wave = threadIdx.x >> 6;
  const int pix_wave = (bx * BN_V4_NT + 64 * wave) * 4;  // first pixel of this wave's 256
  const int pix = pix_wave + 4 * (threadIdx.x & 63);
  float v[4][8];
  if (pix < HW) {
    const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
    float r[4][8];
    if (RES == 2) {  // compact activation storage: the residual exists only as its companion
      const float rscale = operand_scale<P>(res_bound);
#pragma unroll
      for (int j = 0; j < 4; ++j) join_load<P>(res_cb, rscale, res_lead_only ? (size_t)0 : (size_t)N * C * HW, (size_t)ng * HW + pix + j, r[j]);
    }
    float4 zq[8], rq[8];  // every load of the thread in flight before the first use
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      zq[e] = *reinterpret_cast<const float4*>(z + base + (size_t)e * HW);
      if (RES == 1) rq[e] = *reinterpret_cast<const float4*>(res + base + (size_t)e * HW);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float4 zv = zq[e];
      float4 t;
      t.x = fmaf(zv.x, ca[e], cbeta[e]); t.y = fmaf(zv.y, ca[e], cbeta[e]);
      t.z = fmaf(zv.z, ca[e], cbeta[e]); t.w = fmaf(zv.w, ca[e], cbeta[e]);
      if (RES == 1) {
        const float4 rv = rq[e];
        t.x += rv.x; t.y += rv.y; t.z += rv.z; t.w += rv.w;
      }
      if (RES == 2) { t.x += r[0][e]; t.y += r[1][e]; t.z += r[2][e]; t.w += r[3][e]; }
      if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      v[0][e] = t.x; v[1][e] = t.y; v[2][e] = t.z; v[3][e] = t.w;
      if (y != nullptr) *reinterpret_cast<float4*>(y + base + (size_t)e * HW) = t;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
  }
  if (pix_wave < HW) {  // wave-uniform
    if (rmask != nullptr) {  // the ReLU bit-plane (see mcd_readlane64): lane 4 e + j keeps the ballot of (channel e, pixel slot j)
      const int lane = threadIdx.x & 63;
      unsigned long long mine = 0ull;
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned long long b = __ballot(v[j][e] > 0.f);
          if (lane == 4 * e + j) mine = b;
        }
      const int nblk = (HW + 255) >> 8;
      if (lane < 32) rmask[(((size_t)n * C + 8 * g + (lane >> 2)) * nblk + (pix_wave >> 8)) * 4 + (lane & 3)] = mine;
    }
    split_store_x4<P>(v, inv_scale, cb, (size_t)N * C * HW, (size_t)ng * HW + pix_wave, HW - pix_wave < 256 ? HW - pix_wave : 256, lds[wave]);
  }
}

template <class P>
__global__ __launch_bounds__(256) void bn_bwd_apply_cb_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                              float* __restrict__ dz, float* __restrict__ dres,
                                                              typename P::elem* __restrict__ cb, const float* __restrict__ dz_bound,
                                                              const typename P::elem* __restrict__ y_cb, int N, int C, int HW, int relu,
                                                              int train, const float* __restrict__ mbeta) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  const float inv_scale = 1.f / operand_scale<P>(dz_bound);
  const float inv_n = 1.f / ((float)N * (float)HW);
  const bool zm = relu && mbeta != nullptr;  // mask from z (see bn_bwd_reduce_kernel): y is not read
  float cmu[8], crs[8], ca[8], k1[8], k2[8], cmb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * g + e;
    cmu[e] = mean[c];
    crs[e] = rstd[c];
    ca[e] = gamma[c] * crs[e];
    cmb[e] = zm ? mbeta[c] - cmu[e] * ca[e] : 0.f;
    k1[e] = train ? dbeta[c] * inv_n : 0.f;
    k2[e] = train ? dgamma[c] * inv_n : 0.f;
  }
#pragma unroll
  for (int it = 0; it < BN_PIX_ITERS; ++it) {
    const int pix = (blockIdx.x * BN_PIX_ITERS + it) * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
    const unsigned ymask = (relu && !zm && y == nullptr) ? mask_load<P>(y_cb, (size_t)ng * HW + pix) : 0xFFu;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gv = dy[base + (size_t)e * HW];
      const float zv = z[base + (size_t)e * HW];
      if (zm) {
        if (!(fmaf(zv, ca[e], cmb[e]) > 0.f)) gv = 0.f;
      } else if (relu && (y != nullptr ? !(y[base + (size_t)e * HW] > 0.f) : !((ymask >> e) & 1u))) {
        gv = 0.f;
      }
      if (dres) dres[base + (size_t)e * HW] = gv;
      const float t = ca[e] * (gv - k1[e] - ((zv - cmu[e]) * crs[e]) * k2[e]);
      v[e] = t;
      if (dz) dz[base + (size_t)e * HW] = t;  // optional: the split dgrad and wgrad read only the companion
    }
    split_store<P>(v, inv_scale, cb, (size_t)N * C * HW, (size_t)ng * HW + pix);
  }
}

// four adjacent pixels per thread (see bn_apply_cb_v4_kernel).  MASK (compile-time, for the reason given at bn_bwd_reduce_kernel): where
// y > 0 comes from -- 0 no ReLU, 1 the fp32 y, 2 z (mbeta: a group without residual), 3 the bit-plane (rmask), 4 y's companion.
template <class P, int MASK>
__global__ __launch_bounds__(BN_V4_NT) void bn_bwd_apply_cb_v4_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                 const float* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                 float* __restrict__ dz, float* __restrict__ dres,
                                                                 typename P::elem* __restrict__ cb, const float* __restrict__ dz_bound,
                                                                 const typename P::elem* __restrict__ y_cb, int N, int C, int HW,
                                                                 int relu, int train, const float* __restrict__ mbeta, int rev,
                                                                 const unsigned long long* __restrict__ rmask) {
  __shared__ typename P::frag lds[BN_V4_NT / 64][256];
  const int C8 = C >> 3;
  // rev: walk the tensor from its END -- the reduce pass that ran just before read dy and z front to back, so their tails are what the
  // Infinity Cache still holds (MI355X_MICROARCH.md: a line survives while everything touched since fits in ~256 MB)
  const int ng = rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int bx = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int g = ng % C8;
  const int n = ng / C8;
  const float inv_scale = 1.f / operand_scale<P>(dz_bound);
  const float inv_n = 1.f / ((float)N * (float)HW);
  constexpr bool zm = MASK == 2;
  float cmu[8], crs[8], ca[8], k1[8], k2[8], cmb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * g + e;
    cmu[e] = mean[c];
    crs[e] = rstd[c];
    ca[e] = gamma[c] * crs[e];
    cmb[e] = zm ? mbeta[c] - cmu[e] * ca[e] : 0.f;
    k1[e] = train ? dbeta[c] * inv_n : 0.f;
    k2[e] = train ? dgamma[c] * inv_n : 0.f;
  }
  const int wave = threadIdx.x >> 6;
  const int pix_wave = (bx * BN_V4_NT + 64 * wave) * 4;
  const int pix = pix_wave + 4 * (threadIdx.x & 63);
  // the ReLU bit-plane of a group with residual: the wave's 32 words (8 channels x 4 pixel slots), word 4 e + j in lane 4 e + j
  constexpr bool bm = MASK == 3;
  unsigned long long mw = 0ull;
  if (bm && pix_wave < HW && (threadIdx.x & 63) < 32)
    mw = rmask[(((size_t)n * C + 8 * g + ((threadIdx.x & 63) >> 2)) * ((HW + 255) >> 8) + (pix_wave >> 8)) * 4 + (threadIdx.x & 3)];
  float v[4][8];
  if (pix < HW) {
    const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
    unsigned ymask[4] = {0xFFu, 0xFFu, 0xFFu, 0xFFu};
    if (MASK == 4) {  // compact activation storage: the mask from the leading piece of y's companion
#pragma unroll
      for (int j = 0; j < 4; ++j) ymask[j] = mask_load<P>(y_cb, (size_t)ng * HW + pix + j);
    }
    float4 gq[8], zq[8], yq[8];  // every load of the thread in flight before the first use
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gq[e] = *reinterpret_cast<const float4*>(dy + base + (size_t)e * HW);
      zq[e] = *reinterpret_cast<const float4*>(z + base + (size_t)e * HW);
      if (MASK == 1) yq[e] = *reinterpret_cast<const float4*>(y + base + (size_t)e * HW);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float4 g4 = gq[e];
      const float4 z4 = zq[e];
      float gv[4] = {g4.x, g4.y, g4.z, g4.w};
      const float zv[4] = {z4.x, z4.y, z4.z, z4.w};
      if (bm) {  // (v_readlane reads its source lane whatever the execution mask: lanes 0-31 loaded their words above, unconditionally)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((mcd_readlane64(mw, 4 * e + j) >> (threadIdx.x & 63)) & 1ull)) gv[j] = 0.f;
      } else if (zm) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(fmaf(zv[j], ca[e], cmb[e]) > 0.f)) gv[j] = 0.f;
      } else if (MASK == 1) {
        const float4 y4 = yq[e];
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(yv[j] > 0.f)) gv[j] = 0.f;
      } else if (MASK == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((ymask[j] >> e) & 1u)) gv[j] = 0.f;
      }
      if (dres) *reinterpret_cast<float4*>(dres + base + (size_t)e * HW) = make_float4(gv[0], gv[1], gv[2], gv[3]);
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = ca[e] * (gv[j] - k1[e] - ((zv[j] - cmu[e]) * crs[e]) * k2[e]);
        v[j][e] = t[j];
      }
      if (dz) *reinterpret_cast<float4*>(dz + base + (size_t)e * HW) = make_float4(t[0], t[1], t[2], t[3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
  }
  if (pix_wave < HW)  // wave-uniform
    split_store_x4<P>(v, inv_scale, cb, (size_t)N * C * HW, (size_t)ng * HW + pix_wave, HW - pix_wave < 256 ? HW - pix_wave : 256, lds[wave]);
}

// backward reduce with the ReLU mask read from the activation's companion (compact activation storage: no fp32 y exists).
// One thread = one pixel x 8 channels, as the apply kernels; grid (slices, N * C/8): a block strides over the pixels of its
// (image, channel group) and writes one partial row per channel: part[(n * slices + slice) * 3 + {0,1,2}][C]
template <class P>
__global__ __launch_bounds__(256) void bn_bwd_reduce_cb_kernel(const float* __restrict__ dy, const typename P::elem* __restrict__ y_cb,
                                                               const float* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ part, int N, int C,
                                                               int HW, int relu, float* __restrict__ dz_bound) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  if (dz_bound != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dz_bound = 0.f;  // finalize: atomic max
  float mu[8], rs[8], s_dy[8], s_dyx[8], m_g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[8 * g + e];
    rs[e] = rstd[8 * g + e];
    s_dy[e] = s_dyx[e] = m_g[e] = 0.f;
  }
  for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < HW; pix += gridDim.x * blockDim.x) {
    const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
    const unsigned ymask = relu ? mask_load<P>(y_cb, (size_t)ng * HW + pix) : 0xFFu;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gv = dy[base + (size_t)e * HW];
      if (!((ymask >> e) & 1u)) gv = 0.f;
      s_dy[e] += gv;
      m_g[e] = fmaxf(m_g[e], fabsf(gv));
      s_dyx[e] += gv * ((z[base + (size_t)e * HW] - mu[e]) * rs[e]);
    }
  }
  __shared__ float sh[3][8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = wave_sum(s_dy[e]), b = wave_sum(s_dyx[e]);
    float m = m_g[e];
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
      sh[0][e][threadIdx.x >> 6] = a;
      sh[1][e][threadIdx.x >> 6] = b;
      sh[2][e][threadIdx.x >> 6] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 24) {
    const int k = threadIdx.x >> 3, e = threadIdx.x & 7;
    const float v = k == 2 ? fmaxf(fmaxf(sh[2][e][0], sh[2][e][1]), fmaxf(sh[2][e][2], sh[2][e][3]))
                           : (sh[k][e][0] + sh[k][e][1]) + (sh[k][e][2] + sh[k][e][3]);
    part[((size_t)(n * gridDim.x + blockIdx.x) * 3 + k) * C + 8 * g + e] = v;
  }
}

int bwd_cb_slices(int N, int C, int HW) {
  // ~2048 blocks, at least 256 pixels per block
  int s = (int)ceil_div64(2048, (int64_t)N * (C / 8));
  const int most = ceil_div(HW, 256);
  if (s > most) s = most;
  return s < 1 ? 1 : s;
}

// companion -> fp32 NCHW (an activation kept only as its companion, for a consumer outside the split kernels)
template <class P>
__global__ __launch_bounds__(256) void unsplit_cb_kernel(const typename P::elem* __restrict__ cb, const float* __restrict__ bound,
                                                         float* __restrict__ x, int N, int C, int HW, int lead_only) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= HW) return;
  float v[8];
  join_load<P>(cb, operand_scale<P>(bound), lead_only ? (size_t)0 : (size_t)N * C * HW, (size_t)ng * HW + pix, v);
  const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[base + (size_t)e * HW] = v[e];
}

// the split alone, for operands no fused BN group produced (network inputs, gradients arriving from outside the encoder)
template <class P>
__global__ __launch_bounds__(256) void split_cb_kernel(const float* __restrict__ x, typename P::elem* __restrict__ cb,
                                                       const float* __restrict__ x_bound, int N, int C, int HW) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= HW) return;
  const float inv_scale = 1.f / operand_scale<P>(x_bound);
  const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = x[base + (size_t)e * HW];
  split_store<P>(v, inv_scale, cb, (size_t)N * C * HW, (size_t)ng * HW + pix);
}

// the same for a channel count that is not a multiple of 8 (the 6-channel network input): the companion has ceil(C/8) channel
// groups, the missing channels are zeros
template <class P>
__global__ __launch_bounds__(256) void split_cb_padded_kernel(const float* __restrict__ x, typename P::elem* __restrict__ cb,
                                                              const float* __restrict__ x_bound, int N, int C, int HW) {
  const int C8 = (C + 7) >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= HW) return;
  const float inv_scale = 1.f / operand_scale<P>(x_bound);
  const size_t base = ((size_t)n * C + 8 * g) * HW + pix;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (8 * g + e < C) ? x[base + (size_t)e * HW] : 0.f;
  split_store<P>(v, inv_scale, cb, (size_t)N * C8 * 8 * HW, (size_t)ng * HW + pix);
}

int plane_chunks(int HW, bool vec) {
  const int work = vec ? HW / 4 : HW;
  int chunks = ceil_div(work, 256 * 4);  // ~4 elements (float4s) per thread
  if (chunks < 1) chunks = 1;
  if (chunks > 1024) chunks = 1024;
  return chunks;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the four-pixel apply kernels walk their tensors back to front (see the kernels); MCDSEG_BN_REVERSE=0: front to back (development knob)
int bn_reverse_walk() { return (int)mcd_opt(MCD_OPT_BN_REVERSE); }

}  // namespace

extern "C" size_t mcdseg_bn_stats_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)stats_slices(rows, C) * 3 * C * sizeof(double);
}

extern "C" int mcdseg_bn_stats_finalize(const float* stat_partials, int64_t rows, int32_t C, int32_t Mp, float* mean, float* rstd,
                                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                        float eps, const float* gamma, const float* beta, const float* res_bound, float* y_bound,
                                        int32_t running_updates, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(stat_partials && mean && rstd && workspace, "bn_stats_finalize: null pointer");
  MCD_REQUIRE(running_updates >= 1 && running_updates <= 64, "bn_stats_finalize: running_updates must be 1..64 (got %d)", running_updates);
  MCD_REQUIRE(rows > 0 && C > 0 && Mp >= C, "bn_stats_finalize: bad dims rows=%lld C=%d Mp=%d", (long long)rows, C, Mp);
  MCD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_finalize: running stats must come in pairs");
  MCD_REQUIRE(y_bound == nullptr || (gamma && beta), "bn_stats_finalize: the output bound needs gamma and beta");
  MCD_REQUIRE(workspace_bytes >= mcdseg_bn_stats_workspace_bytes(rows, C), "bn_stats_finalize: workspace too small");
  MCD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "bn_stats_finalize: workspace must be 8-byte aligned");
  const int S = stats_slices(rows, C);
  hipStream_t st = (hipStream_t)stream;
  if (rows <= stats_one_rows()) {
    hipLaunchKernelGGL(bn_stats_one_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, stat_partials, rows, S, C, Mp, mean, rstd, running_mean,
                       running_var, num_batches_tracked, momentum, eps, gamma, beta, res_bound, y_bound, running_updates);
    MCD_LAUNCH_CHECK("bn_stats_one");
    return 0;
  }
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(ceil_div(C, 32), S), dim3(256), 0, st, stat_partials, rows, C, Mp,
                     (double*)workspace, y_bound);
  MCD_LAUNCH_CHECK("bn_stats_partial");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const double*)workspace, S, C, mean,
                     rstd, running_mean, running_var, num_batches_tracked, momentum, eps, gamma, beta, res_bound, y_bound, running_updates);
  MCD_LAUNCH_CHECK("bn_stats_finalize");
  return 0;
}

extern "C" int mcdseg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps, float* mean,
                                    float* rstd, void* stream) {
  MCD_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean, running_var,
                     C, eps, mean, rstd);
  MCD_LAUNCH_CHECK("bn_eval_stats");
  return 0;
}

extern "C" int mcdseg_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                     const float* conv_bias, int32_t C, float eps, float* scale, float* shift, void* stream) {
  MCD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_eval_affine: bad arguments");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, conv_bias, C, eps, scale, shift);
  MCD_LAUNCH_CHECK("bn_eval_affine");
  return 0;
}

extern "C" int mcdseg_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const float* residual, float* y, int32_t N, int32_t C, int32_t HW, int32_t relu, void* stream) {
  MCD_REQUIRE(z && mean && rstd && gamma && beta && y, "bn_apply: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_apply: bad dims");
  const bool vec = (HW % 4 == 0) && aligned16(z) && aligned16(y) && (!residual || aligned16(residual));
  dim3 grid(N * C, plane_chunks(HW, vec));
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, y, C,
                       HW, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, y, C,
                       HW, relu);
  MCD_LAUNCH_CHECK("bn_apply");
  return 0;
}

static int cb_check(const char* who, int32_t math, const float* bound, int32_t N, int32_t C, int32_t HW) {
  MCD_REQUIRE(math == MCDSEG_MATH_BF16X6 || math == MCDSEG_MATH_F16X3, "%s: unknown math %d", who, math);
  MCD_REQUIRE(math != MCDSEG_MATH_F16X3 || bound != nullptr, "%s: the f16x3 split needs the bound scalar of the tensor", who);
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0 && (C % 8) == 0, "%s: C must be a positive multiple of 8", who);
  MCD_REQUIRE((int64_t)N * (C / 8) <= 65535, "%s: N*C/8 exceeds the grid limit", who);
  return 0;
}

extern "C" int mcdseg_split_cb(const float* x, void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW,
                               void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(x && x_cb, "split_cb: null pointer");
  if (int rc = cb_check("split_cb", math, x_bound, N, C, HW)) return rc;
  const dim3 grid(ceil_div(HW, 256), N * (C / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(split_cb_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, x, (_Float16*)x_cb, x_bound, N, C, HW);
  else
    hipLaunchKernelGGL(split_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, x, (__bf16*)x_cb, x_bound, N, C, HW);
  MCD_LAUNCH_CHECK("split_cb");
  return 0;
}

extern "C" int mcdseg_split_cb_padded(const float* x, void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW,
                                      void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(x && x_cb, "split_cb_padded: null pointer");
  MCD_REQUIRE(math == MCDSEG_MATH_BF16X6 || math == MCDSEG_MATH_F16X3, "split_cb_padded: unknown math %d", math);
  MCD_REQUIRE(math != MCDSEG_MATH_F16X3 || x_bound != nullptr, "split_cb_padded: the f16x3 split needs the bound scalar of the tensor");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0 && (int64_t)N * ((C + 7) / 8) <= 65535, "split_cb_padded: bad dims");
  const dim3 grid(ceil_div(HW, 256), N * ((C + 7) / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(split_cb_padded_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, x, (_Float16*)x_cb, x_bound, N, C, HW);
  else
    hipLaunchKernelGGL(split_cb_padded_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, x, (__bf16*)x_cb, x_bound, N, C, HW);
  MCD_LAUNCH_CHECK("split_cb_padded");
  return 0;
}

// the four-pixels-per-thread forms of the two companion-writing apply kernels (MCDSEG_BN_V4=0: the one-pixel forms)
static bool bn_v4_on() { return mcd_opt(MCD_OPT_BN_V4) != 0; }

template <class P, typename... A>
static void launch_apply_v4(const float* res, const void* res_cb, dim3 grid, hipStream_t st, A... a) {
  if (res != nullptr)
    hipLaunchKernelGGL((bn_apply_cb_v4_kernel<P, 1>), grid, dim3(BN_V4_NT), 0, st, a...);
  else if (res_cb != nullptr)
    hipLaunchKernelGGL((bn_apply_cb_v4_kernel<P, 2>), grid, dim3(BN_V4_NT), 0, st, a...);
  else
    hipLaunchKernelGGL((bn_apply_cb_v4_kernel<P, 0>), grid, dim3(BN_V4_NT), 0, st, a...);
}

template <class P, typename... A>
static void launch_bwd_apply_v4(int mask, dim3 grid, hipStream_t st, A... a) {
#define MCD_BA(M) hipLaunchKernelGGL((bn_bwd_apply_cb_v4_kernel<P, M>), grid, dim3(BN_V4_NT), 0, st, a...)
  if (mask == 0) MCD_BA(0); else if (mask == 1) MCD_BA(1); else if (mask == 2) MCD_BA(2); else if (mask == 3) MCD_BA(3); else MCD_BA(4);
#undef MCD_BA
}

extern "C" int mcdseg_bn_apply_cb(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* residual, const void* res_cb, const float* res_bound, float* y, void* y_cb,
                                  const float* y_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t relu, void* stream) {
  const int lead = math == MCDSEG_MATH_F16X1;  // (F16X1 reads the leading piece of a companion only)
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(z && mean && rstd && gamma && beta && y_cb, "bn_apply_cb: null pointer");
  MCD_REQUIRE(!(residual && res_cb), "bn_apply_cb: the residual comes either as fp32 or as its companion, not both");
  MCD_REQUIRE(res_cb == nullptr || math != MCDSEG_MATH_F16X3 || res_bound != nullptr, "bn_apply_cb: the residual companion needs its bound");
  if (int rc = cb_check("bn_apply_cb", math, y_bound, N, C, HW)) return rc;
  if (bn_v4_on() && (HW & 3) == 0 && (((uintptr_t)z | (uintptr_t)y | (uintptr_t)residual) & 15) == 0) {
    const dim3 grid4(ceil_div(HW, 4 * BN_V4_NT), N * (C / 8));
    const int rev = bn_reverse_walk();
    if (math == MCDSEG_MATH_F16X3)
      launch_apply_v4<SplitF16x3>(residual, res_cb, grid4, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, (const _Float16*)res_cb,
                                  res_bound, y, (_Float16*)y_cb, y_bound, N, C, HW, relu, rev, (unsigned long long*)nullptr, lead);
    else
      launch_apply_v4<SplitBf16x6>(residual, res_cb, grid4, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, (const __bf16*)res_cb,
                                   res_bound, y, (__bf16*)y_cb, y_bound, N, C, HW, relu, rev, (unsigned long long*)nullptr, lead);
    MCD_LAUNCH_CHECK("bn_apply_cb");
    return 0;
  }
  const dim3 grid(ceil_div(HW, 256 * BN_PIX_ITERS), N * (C / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(bn_apply_cb_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual,
                       (const _Float16*)res_cb, res_bound, y, (_Float16*)y_cb, y_bound, N, C, HW, relu, lead);
  else
    hipLaunchKernelGGL(bn_apply_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual,
                       (const __bf16*)res_cb, res_bound, y, (__bf16*)y_cb, y_bound, N, C, HW, relu, lead);
  MCD_LAUNCH_CHECK("bn_apply_cb");
  return 0;
}

extern "C" int mcdseg_unsplit_cb(const void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW, float* x,
                                 void* stream) {
  const int lead = math == MCDSEG_MATH_F16X1;  // (the leading piece only: an F16X1 companion may be stored as one piece)
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(x_cb && x, "unsplit_cb: null pointer");
  if (int rc = cb_check("unsplit_cb", math, x_bound, N, C, HW)) return rc;
  const dim3 grid(ceil_div(HW, 256), N * (C / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(unsplit_cb_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_cb, x_bound, x, N, C, HW, lead);
  else
    hipLaunchKernelGGL(unsplit_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x_cb, x_bound, x, N, C, HW, lead);
  MCD_LAUNCH_CHECK("unsplit_cb");
  return 0;
}

static bool bwd_apply_v4(const float* dy, const float* y, const void* y_cb, const float* z, const float* mean, const float* rstd,
                         const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres, void* dz_cb,
                         const float* dz_bound, int math, int N, int C, int HW, int relu, int train, const float* mbeta, hipStream_t st,
                         const unsigned long long* rmask = nullptr) {
  if (!bn_v4_on() || (HW & 3) != 0) return false;
  if ((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)dres) & 15) != 0) return false;
  const dim3 grid(ceil_div(HW, 4 * BN_V4_NT), N * (C / 8));
  const int rev = bn_reverse_walk();
  const int mask = !relu ? 0 : (mbeta != nullptr ? 2 : (rmask != nullptr ? 3 : (y != nullptr ? 1 : 4)));  // (the kernel's own order of preference)
  if (math == MCDSEG_MATH_F16X3)
    launch_bwd_apply_v4<SplitF16x3>(mask, grid, st, dy, y, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, (_Float16*)dz_cb, dz_bound,
                                    (const _Float16*)y_cb, N, C, HW, relu, train, mbeta, rev, rmask);
  else
    launch_bwd_apply_v4<SplitBf16x6>(mask, grid, st, dy, y, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, (__bf16*)dz_cb, dz_bound,
                                     (const __bf16*)y_cb, N, C, HW, relu, train, mbeta, rev, rmask);
  return true;
}

// ---- the ReLU bit-plane of a group with residual (round 5; see mcd_readlane64 above): three entry points beside the fp32-y forms
extern "C" size_t mcdseg_bn_relu_mask_bytes(int32_t N, int32_t C, int32_t HW) {
  // (the bit-plane is written and read by the four-pixel kernels only: HW a multiple of 4, channel groups of 8)
  if (N <= 0 || C <= 0 || HW <= 0 || (C & 7) != 0 || (HW & 3) != 0 || !bn_v4_on()) return 0;
  return (size_t)N * C * ((HW + 255) >> 8) * 32;
}

extern "C" int mcdseg_bn_apply_cb_mask(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       const float* residual, float* y, void* y_cb, const float* y_bound, void* relu_mask, int32_t math,
                                       int32_t N, int32_t C, int32_t HW, void* stream) {
  math = mcd_storage_math(math);
  MCD_REQUIRE(z && mean && rstd && gamma && beta && y_cb && relu_mask, "bn_apply_cb_mask: null pointer");
  if (int rc = cb_check("bn_apply_cb_mask", math, y_bound, N, C, HW)) return rc;
  MCD_REQUIRE(mcdseg_bn_relu_mask_bytes(N, C, HW) > 0 && (((uintptr_t)z | (uintptr_t)y | (uintptr_t)residual) & 15) == 0 &&
                  ((uintptr_t)relu_mask & 7) == 0,
              "bn_apply_cb_mask: needs HW %% 4 == 0, 16-byte aligned tensors and an 8-byte aligned mask (mcdseg_bn_relu_mask_bytes)");
  const dim3 grid4(ceil_div(HW, 4 * BN_V4_NT), N * (C / 8));
  const int rev = bn_reverse_walk();
  if (math == MCDSEG_MATH_F16X3)
    launch_apply_v4<SplitF16x3>(residual, nullptr, grid4, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, (const _Float16*)nullptr,
                                (const float*)nullptr, y, (_Float16*)y_cb, y_bound, N, C, HW, 1, rev, (unsigned long long*)relu_mask, 0);
  else
    launch_apply_v4<SplitBf16x6>(residual, nullptr, grid4, (hipStream_t)stream, z, mean, rstd, gamma, beta, residual, (const __bf16*)nullptr,
                                 (const float*)nullptr, y, (__bf16*)y_cb, y_bound, N, C, HW, 1, rev, (unsigned long long*)relu_mask, 0);
  MCD_LAUNCH_CHECK("bn_apply_cb_mask");
  return 0;
}

extern "C" int mcdseg_bn_bwd_apply_cb_mask(const float* dy, const void* relu_mask, const float* z, const float* mean, const float* rstd,
                                           const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres, void* dz_cb,
                                           const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t train,
                                           void* stream) {
  math = mcd_storage_math(math);
  MCD_REQUIRE(dy && relu_mask && z && mean && rstd && gamma && dz_cb, "bn_bwd_apply_cb_mask: null pointer");
  MCD_REQUIRE(!train || (dgamma && dbeta), "bn_bwd_apply_cb_mask: train mode needs dgamma/dbeta");
  if (int rc = cb_check("bn_bwd_apply_cb_mask", math, dz_bound, N, C, HW)) return rc;
  MCD_REQUIRE(mcdseg_bn_relu_mask_bytes(N, C, HW) > 0, "bn_bwd_apply_cb_mask: no bit-plane exists for this geometry");
  MCD_REQUIRE(bwd_apply_v4(dy, nullptr, nullptr, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, dz_cb, dz_bound, math, N, C, HW, 1, train, nullptr,
                           (hipStream_t)stream, (const unsigned long long*)relu_mask),
              "bn_bwd_apply_cb_mask: tensors must be 16-byte aligned");
  MCD_LAUNCH_CHECK("bn_bwd_apply_cb_mask");
  return 0;
}

extern "C" int mcdseg_bn_bwd_apply_cb(const float* dy, const float* y, const void* y_cb, const float* z, const float* mean,
                                      const float* rstd, const float* gamma, const float* dgamma, const float* dbeta, float* dz,
                                      float* dres, void* dz_cb, const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW,
                                      int32_t relu, int32_t train, void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(dy && z && mean && rstd && gamma && dz_cb, "bn_bwd_apply_cb: null pointer");
  MCD_REQUIRE(!relu || y || y_cb, "bn_bwd_apply_cb: relu mask needs y or its companion");
  MCD_REQUIRE(!train || (dgamma && dbeta), "bn_bwd_apply_cb: train mode needs dgamma/dbeta");
  if (int rc = cb_check("bn_bwd_apply_cb", math, dz_bound, N, C, HW)) return rc;
  if (bwd_apply_v4(dy, y, y_cb, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, dz_cb, dz_bound, math, N, C, HW, relu, train, nullptr,
                   (hipStream_t)stream)) {
    MCD_LAUNCH_CHECK("bn_bwd_apply_cb");
    return 0;
  }
  const dim3 grid(ceil_div(HW, 256 * BN_PIX_ITERS), N * (C / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(bn_bwd_apply_cb_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, z, mean, rstd, gamma, dgamma,
                       dbeta, dz, dres, (_Float16*)dz_cb, dz_bound, (const _Float16*)y_cb, N, C, HW, relu, train, (const float*)nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, z, mean, rstd, gamma, dgamma,
                       dbeta, dz, dres, (__bf16*)dz_cb, dz_bound, (const __bf16*)y_cb, N, C, HW, relu, train, (const float*)nullptr);
  MCD_LAUNCH_CHECK("bn_bwd_apply_cb");
  return 0;
}

// the same for a ReLU group WITHOUT residual, the mask recomputed from z (y > 0 <=> fma(z, gamma rstd, beta - mean gamma rstd) > 0, the
// forward kernels' own expression): y is not read -- 12 instead of 16 bytes per element
extern "C" int mcdseg_bn_bwd_apply_cb_zmask(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                            const float* beta, const float* dgamma, const float* dbeta, float* dz, void* dz_cb,
                                            const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t train,
                                            void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(dy && z && mean && rstd && gamma && beta && dz_cb, "bn_bwd_apply_cb_zmask: null pointer");
  MCD_REQUIRE(!train || (dgamma && dbeta), "bn_bwd_apply_cb_zmask: train mode needs dgamma/dbeta");
  if (int rc = cb_check("bn_bwd_apply_cb_zmask", math, dz_bound, N, C, HW)) return rc;
  if (bwd_apply_v4(dy, nullptr, nullptr, z, mean, rstd, gamma, dgamma, dbeta, dz, nullptr, dz_cb, dz_bound, math, N, C, HW, 1, train, beta,
                   (hipStream_t)stream)) {
    MCD_LAUNCH_CHECK("bn_bwd_apply_cb_zmask");
    return 0;
  }
  const dim3 grid(ceil_div(HW, 256 * BN_PIX_ITERS), N * (C / 8));
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(bn_bwd_apply_cb_kernel<SplitF16x3>, grid, dim3(256), 0, (hipStream_t)stream, dy, (const float*)nullptr, z, mean, rstd,
                       gamma, dgamma, dbeta, dz, (float*)nullptr, (_Float16*)dz_cb, dz_bound, (const _Float16*)nullptr, N, C, HW, 1, train,
                       beta);
  else
    hipLaunchKernelGGL(bn_bwd_apply_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, (hipStream_t)stream, dy, (const float*)nullptr, z, mean, rstd,
                       gamma, dgamma, dbeta, dz, (float*)nullptr, (__bf16*)dz_cb, dz_bound, (const __bf16*)nullptr, N, C, HW, 1, train, beta);
  MCD_LAUNCH_CHECK("bn_bwd_apply_cb_zmask");
  return 0;
}

extern "C" size_t mcdseg_bn_bwd_workspace_bytes(int32_t N, int32_t C, int32_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  const BwdPlan pl = bwd_plan(N, C, HW);
  size_t rows = (size_t)pl.S;
  if ((C % 8) == 0) {  // the companion-mask form of the reduce has its own slicing
    const size_t r2 = (size_t)N * bwd_cb_slices(N, C, HW);
    if (r2 > rows) rows = r2;
  }
  return rows * 3 * C * sizeof(float);
}

extern "C" int mcdseg_bn_bwd_reduce(const float* dy, const float* y, const void* y_cb, int32_t math, const float* z, const float* mean,
                                    const float* rstd, float* dgamma, float* dbeta, const float* gamma, float* dz_bound, int32_t train,
                                    int32_t N, int32_t C, int32_t HW, int32_t relu, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  math = mcd_storage_math(math);  // F16X1 shares F16X3's storage
  MCD_REQUIRE(dy && workspace && (dgamma || dbeta), "bn_bwd_reduce: null pointer");
  MCD_REQUIRE(!relu || y || y_cb, "bn_bwd_reduce: relu mask needs y or its companion");
  MCD_REQUIRE(!z || (mean && rstd), "bn_bwd_reduce: z needs mean/rstd");
  MCD_REQUIRE(dz_bound == nullptr || (z && gamma), "bn_bwd_reduce: the dz bound needs z, rstd and gamma");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_bwd_reduce: bad dims");
  MCD_REQUIRE(workspace_bytes >= mcdseg_bn_bwd_workspace_bytes(N, C, HW), "bn_bwd_reduce: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int S;
  if (relu && y == nullptr) {  // mask from the companion: pixel x 8-channel threads
    MCD_REQUIRE(z != nullptr, "bn_bwd_reduce: the companion-mask form is the BatchNorm form (needs z)");
    if (int rc = cb_check("bn_bwd_reduce", math, math == MCDSEG_MATH_F16X3 ? rstd : nullptr, N, C, HW)) return rc;  // layout rules only
    const int sl = bwd_cb_slices(N, C, HW);
    S = N * sl;
    const dim3 grid(sl, N * (C / 8));
    if (math == MCDSEG_MATH_F16X3)
      hipLaunchKernelGGL(bn_bwd_reduce_cb_kernel<SplitF16x3>, grid, dim3(256), 0, st, dy, (const _Float16*)y_cb, z, mean, rstd,
                         (float*)workspace, N, C, HW, relu, dz_bound);
    else
      hipLaunchKernelGGL(bn_bwd_reduce_cb_kernel<SplitBf16x6>, grid, dim3(256), 0, st, dy, (const __bf16*)y_cb, z, mean, rstd,
                         (float*)workspace, N, C, HW, relu, dz_bound);
  } else {
    const BwdPlan pl = bwd_plan(N, C, HW);
    MCD_REQUIRE(pl.S <= 65535, "bn_bwd_reduce: too many splits");
    S = pl.S;
    const bool vec = (HW % 4 == 0) && aligned16(dy) && (!y || aligned16(y)) && (!z || aligned16(z));
    dim3 grid(C, pl.S);
    launch_bwd_reduce(vec, relu ? 1 : 0, grid, st, dy, y, z, mean, rstd, (float*)workspace, N, C, HW, relu, pl.cpp, pl.chunk, dz_bound,
                      (const float*)nullptr, (const float*)nullptr, (const unsigned long long*)nullptr, 0);
  }
  MCD_LAUNCH_CHECK("bn_bwd_reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, S, C,
                     z ? dgamma : nullptr, dbeta, gamma, rstd, (float)N * (float)HW, train, dz_bound);
  MCD_LAUNCH_CHECK("bn_bwd_finalize");
  return 0;
}

extern "C" int mcdseg_bn_bwd_reduce_zmask(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, float* dgamma, float* dbeta, float* dz_bound, int32_t train, int32_t N,
                                          int32_t C, int32_t HW, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(dy && z && mean && rstd && gamma && beta && dgamma && dbeta && workspace, "bn_bwd_reduce_zmask: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_bwd_reduce_zmask: bad dims");
  MCD_REQUIRE(workspace_bytes >= mcdseg_bn_bwd_workspace_bytes(N, C, HW), "bn_bwd_reduce_zmask: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BwdPlan pl = bwd_plan(N, C, HW);
  MCD_REQUIRE(pl.S <= 65535, "bn_bwd_reduce_zmask: too many splits");
  const bool vec = (HW % 4 == 0) && aligned16(dy) && aligned16(z);
  dim3 grid(C, pl.S);
  launch_bwd_reduce(vec, 2, grid, st, dy, (const float*)nullptr, z, mean, rstd, (float*)workspace, N, C, HW, 1, pl.cpp, pl.chunk, dz_bound,
                    gamma, beta, (const unsigned long long*)nullptr, 0);
  MCD_LAUNCH_CHECK("bn_bwd_reduce_zmask");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, pl.S, C, dgamma, dbeta, gamma,
                     rstd, (float)N * (float)HW, train, dz_bound);
  MCD_LAUNCH_CHECK("bn_bwd_finalize");
  return 0;
}

// backward reduce of a ReLU group WITH residual from its bit-plane: 8 instead of 12 bytes per element
extern "C" int mcdseg_bn_bwd_reduce_mask(const float* dy, const void* relu_mask, const float* z, const float* mean, const float* rstd,
                                         const float* gamma, float* dgamma, float* dbeta, float* dz_bound, int32_t train, int32_t N, int32_t C,
                                         int32_t HW, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(dy && relu_mask && z && mean && rstd && dgamma && dbeta && workspace, "bn_bwd_reduce_mask: null pointer");
  MCD_REQUIRE(dz_bound == nullptr || gamma, "bn_bwd_reduce_mask: the dz bound needs gamma");
  MCD_REQUIRE(mcdseg_bn_relu_mask_bytes(N, C, HW) > 0 && aligned16(dy) && aligned16(z), "bn_bwd_reduce_mask: no bit-plane exists for this geometry");
  MCD_REQUIRE(workspace_bytes >= mcdseg_bn_bwd_workspace_bytes(N, C, HW), "bn_bwd_reduce_mask: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const BwdPlan pl = bwd_plan(N, C, HW);
  MCD_REQUIRE(pl.S <= 65535, "bn_bwd_reduce_mask: too many splits");
  launch_bwd_reduce(true, 3, dim3(C, pl.S), st, dy, (const float*)nullptr, z, mean, rstd, (float*)workspace, N, C, HW, 1, pl.cpp, pl.chunk,
                    dz_bound, (const float*)nullptr, (const float*)nullptr, (const unsigned long long*)relu_mask, (HW + 255) >> 8);
  MCD_LAUNCH_CHECK("bn_bwd_reduce_mask");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, pl.S, C, dgamma, dbeta, gamma,
                     rstd, (float)N * (float)HW, train, dz_bound);
  MCD_LAUNCH_CHECK("bn_bwd_finalize");
  return 0;
}

extern "C" int mcdseg_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                                   const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres, int32_t N,
                                   int32_t C, int32_t HW, int32_t relu, int32_t train, void* stream) {
  MCD_REQUIRE(dy && z && mean && rstd && gamma && dz, "bn_bwd_apply: null pointer");
  MCD_REQUIRE(!relu || y, "bn_bwd_apply: relu mask needs y");
  MCD_REQUIRE(!train || (dgamma && dbeta), "bn_bwd_apply: train mode needs dgamma/dbeta");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_bwd_apply: bad dims");
  const bool vec = (HW % 4 == 0) && aligned16(dy) && aligned16(z) && aligned16(dz) && (!y || aligned16(y)) &&
                   (!dres || aligned16(dres));
  dim3 grid(N * C, plane_chunks(HW, vec));
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(256), 0, st, dy, y, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, N,
                       C, HW, relu, train);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(256), 0, st, dy, y, z, mean, rstd, gamma, dgamma, dbeta, dz, dres, N,
                       C, HW, relu, train);
  MCD_LAUNCH_CHECK("bn_bwd_apply");
  return 0;
}

// ================================================================================================
// 2-byte activation storage (round 6; BASELINE config 5 "bf16", reference network models/drn.py:62-100, 344-348).  Inside a trunk run in
// the one-term arithmetic (MCDSEG_MATH_F16X1) every tensor a BatchNorm pass touches is ONE 16-bit value per element, all in the
// companions' unit layout [N][C/8][HW][8]:
//   z     the convolution's output, fp16 of z / scale(z_bound)      written by the convolution's epilogue (conv_gemm_split*.hip), after
//                                                                   it took the BatchNorm partial sums from its fp32 accumulators
//   y     the activation, fp16 of y / scale(y_bound): ONE piece     it IS the next convolution's operand (SplitF16x1 stages one piece)
//   dy    the incoming gradient, bf16                               written by the next convolution's data gradient
//   dz    fp16 of dz / scale(dz_bound): one piece                   the operand of this convolution's data and weight gradient
//   dres  the residual's gradient, bf16 (= dy under the ReLU mask: exact)
// so a thread of these kernels owns whole 16-byte units -- 8 channels of one pixel -- of every tensor: no plane-to-unit transposition,
// every load and store instruction of a wave covers 1 KB of consecutive bytes, and the passes move 4-6 (forward) / 4-6 (reduce) /
// 6-10 (backward apply) bytes per element where the fp32 forms move 8-12 / 8-10 / 12-18.  All arithmetic is fp32: mean, rstd, the
// running statistics and the bounds come from the same finalize kernels as ever.
namespace {

// (the kernels take their 16-byte units as uint4: rocprofv3's demangler gives up on _Float16 / __bf16 vector types in a signature)
// the forward map of one channel, shared by the forward kernel and the backward kernels' ReLU mask (y > 0 recomputed from z: the same
// expression on the same 16-bit z, so the same mask, bit for bit): y = fma(zh, a zs, b), a = gamma rstd, b = beta - mean a, zh the stored
// half, zs the (power-of-two) scale of z -- a zs is exact
struct HalfAffine {
  float a[8], b[8];
  __device__ __forceinline__ void load(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, int c0, float zs) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ca = gamma[c0 + e] * rstd[c0 + e];
      b[e] = beta[c0 + e] - mean[c0 + e] * ca;
      a[e] = ca * zs;
    }
  }
};

#ifndef BN_H_UNITS
#define BN_H_UNITS 2  // units (pixels) per thread of the two streaming kernels, 256 apart
#endif

// RES: 0 none, 2 the residual as its (leading) companion piece
template <int RES>
__global__ __launch_bounds__(256) void bn_apply_h_kernel(const uint4* __restrict__ z, const float* __restrict__ z_bound,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const uint4* __restrict__ res_cb, const float* __restrict__ res_bound,
                                                         uint4* __restrict__ y_cb, const float* __restrict__ y_bound, int C8, int HW, int relu) {
  const int ng = blockIdx.y;  // n * C8 + g
  const int g = ng % C8;
  HalfAffine f;
  f.load(gamma, beta, mean, rstd, 8 * g, mcd_scale_of_bound(*z_bound));
  const float inv_ys = 1.f / mcd_scale_of_bound(*y_bound);
  const float rs = RES == 2 ? mcd_scale_of_bound(*res_bound) : 0.f;
  const size_t base = (size_t)ng * HW;
  f16x8 zq[BN_H_UNITS], rq[BN_H_UNITS];
  int pix[BN_H_UNITS];
#pragma unroll
  for (int u = 0; u < BN_H_UNITS; ++u) {  // every load of the thread in flight before the first use (clamped: no branch around a load)
    pix[u] = (blockIdx.x * BN_H_UNITS + u) * 256 + threadIdx.x;
    const int pc = pix[u] < HW ? pix[u] : HW - 1;
    zq[u] = __builtin_bit_cast(f16x8, z[base + pc]);
    if (RES == 2) rq[u] = __builtin_bit_cast(f16x8, res_cb[base + pc]);
  }
#pragma unroll
  for (int u = 0; u < BN_H_UNITS; ++u) {
    if (pix[u] >= HW) continue;
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf((float)zq[u][e], f.a[e], f.b[e]);
      if (RES == 2) t += (float)rq[u][e] * rs;
      if (relu) t = fmaxf(t, 0.f);
      o[e] = (_Float16)(t * inv_ys);
    }
    y_cb[base + pix[u]] = __builtin_bit_cast(uint4, o);
  }
}

// ReLU mask of a unit.  MASK: 0 none, 2 recomputed from z (a group without residual), 4 from the activation's piece (y > 0 <=> its
// fp16 image > 0, except 0 < y < 2^-25 of the scale, which the piece rounds to zero -- and which the next convolution multiplies as zero)
template <int MASK>
__device__ __forceinline__ unsigned half_mask(const f16x8& zq, const f16x8& yq, const HalfAffine& f) {
  if (MASK == 0) return 0xFFu;
  unsigned m = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool on = MASK == 2 ? fmaf((float)zq[e], f.a[e], f.b[e]) > 0.f : (float)yq[e] > 0.f;
    m |= (on ? 1u : 0u) << e;
  }
  return m;
}

// backward reduce: grid (slices, N * C8); a block strides over the pixels of its (image, channel group) slice, four units in flight per
// thread, and writes one partial row per channel: part[((n * slices + slice) * 3 + {sum dy, sum dy xhat, max |dy|}) * C + c] -- the
// rows bn_bwd_finalize_kernel merges in fp64
template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_h_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y_cb,
                                                              const uint4* __restrict__ z, const float* __restrict__ z_bound,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ part, int C, int HW, int per_slice,
                                                              float* __restrict__ dz_bound) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const int n = ng / C8;
  if (dz_bound != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *dz_bound = 0.f;  // finalize: atomic max
  const float zs = mcd_scale_of_bound(*z_bound);
  HalfAffine f;
  f.load(gamma, beta, mean, rstd, 8 * g, zs);
  float mu[8], rs[8], s_dy[8], s_dyx[8], m_g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = mean[8 * g + e];
    rs[e] = rstd[8 * g + e];
    s_dy[e] = s_dyx[e] = m_g[e] = 0.f;
  }
  const size_t base = (size_t)ng * HW;
  const int p0 = blockIdx.x * per_slice;
  int p1 = p0 + per_slice;
  if (p1 > HW) p1 = HW;
  for (int pb = p0 + (int)threadIdx.x; pb < p1; pb += 4 * 256) {
    bf16x8 gq[4];
    f16x8 zq[4], yq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pc = pb + u * 256 < p1 ? pb + u * 256 : p1 - 1;
      gq[u] = __builtin_bit_cast(bf16x8, dy[base + pc]);
      zq[u] = __builtin_bit_cast(f16x8, z[base + pc]);
      if (MASK == 4) yq[u] = __builtin_bit_cast(f16x8, y_cb[base + pc]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pb + u * 256 >= p1) break;
      const unsigned m = half_mask<MASK>(zq[u], yq[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gv = ((m >> e) & 1u) ? (float)gq[u][e] : 0.f;
        s_dy[e] += gv;
        m_g[e] = fmaxf(m_g[e], fabsf(gv));
        s_dyx[e] += gv * (((float)zq[u][e] * zs - mu[e]) * rs[e]);
      }
    }
  }
  __shared__ float sh[3][8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = wave_sum(s_dy[e]), b = wave_sum(s_dyx[e]);
    float m = m_g[e];
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) {
      sh[0][e][threadIdx.x >> 6] = a;
      sh[1][e][threadIdx.x >> 6] = b;
      sh[2][e][threadIdx.x >> 6] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 24) {
    const int k = threadIdx.x >> 3, e = threadIdx.x & 7;
    const float v = k == 2 ? fmaxf(fmaxf(sh[2][e][0], sh[2][e][1]), fmaxf(sh[2][e][2], sh[2][e][3]))
                           : (sh[k][e][0] + sh[k][e][1]) + (sh[k][e][2] + sh[k][e][3]);
    part[((size_t)(n * gridDim.x + blockIdx.x) * 3 + k) * C + 8 * g + e] = v;
  }
}

// slices per (image, channel group): ~4096 blocks in all, at least 1024 pixels per block
int bwd_h_slices(int N, int C, int HW) {
  int s = (int)ceil_div64(4096, (int64_t)N * (C / 8));
  const int most = ceil_div(HW, 1024);
  if (s > most) s = most;
  return s < 1 ? 1 : s;
}

template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_h_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y_cb,
                                                             const uint4* __restrict__ z, const float* __restrict__ z_bound,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                             uint4* __restrict__ dz_cb, const float* __restrict__ dz_bound,
                                                             uint4* __restrict__ dres, int N, int C8, int HW, int train) {
  const int ng = blockIdx.y;
  const int g = ng % C8;
  const float zs = mcd_scale_of_bound(*z_bound);
  HalfAffine f;
  f.load(gamma, beta, mean, rstd, 8 * g, zs);
  const float inv_s = 1.f / mcd_scale_of_bound(*dz_bound);
  const float inv_n = 1.f / ((float)N * (float)HW);
  float cmu[8], crs[8], ca[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * g + e;
    cmu[e] = mean[c];
    crs[e] = rstd[c];
    ca[e] = gamma[c] * crs[e];
    k1[e] = train ? dbeta[c] * inv_n : 0.f;
    k2[e] = train ? dgamma[c] * inv_n : 0.f;
  }
  const size_t base = (size_t)ng * HW;
  bf16x8 gq[BN_H_UNITS];
  f16x8 zq[BN_H_UNITS], yq[BN_H_UNITS];
  int pix[BN_H_UNITS];
#pragma unroll
  for (int u = 0; u < BN_H_UNITS; ++u) {
    pix[u] = (blockIdx.x * BN_H_UNITS + u) * 256 + threadIdx.x;
    const int pc = pix[u] < HW ? pix[u] : HW - 1;
    gq[u] = __builtin_bit_cast(bf16x8, dy[base + pc]);
    zq[u] = __builtin_bit_cast(f16x8, z[base + pc]);
    if (MASK == 4) yq[u] = __builtin_bit_cast(f16x8, y_cb[base + pc]);
  }
#pragma unroll
  for (int u = 0; u < BN_H_UNITS; ++u) {
    if (pix[u] >= HW) continue;
    const unsigned m = half_mask<MASK>(zq[u], yq[u], f);
    f16x8 o;
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool on = ((m >> e) & 1u) != 0;
      const float gv = on ? (float)gq[u][e] : 0.f;
      r[e] = on ? gq[u][e] : (__bf16)0.f;
      const float t = ca[e] * (gv - k1[e] - (((float)zq[u][e] * zs - cmu[e]) * crs[e]) * k2[e]);
      o[e] = (_Float16)(t * inv_s);
    }
    dz_cb[base + pix[u]] = __builtin_bit_cast(uint4, o);
    if (dres != nullptr) dres[base + pix[u]] = __builtin_bit_cast(uint4, r);
  }
}

// fp32 NCHW -> bf16 units (a gradient a kernel without the 16-bit epilogue produced, for a consumer inside the 2-byte chain) and back
__global__ __launch_bounds__(256) void pack_bf16_units_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C, int HW) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8, n = ng / C8;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const size_t src = ((size_t)n * C + 8 * g) * HW + pix;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)x[src + (size_t)e * HW];
  out[(size_t)ng * HW + pix] = __builtin_bit_cast(uint4, o);
}

__global__ __launch_bounds__(256) void unpack_bf16_units_kernel(const uint4* __restrict__ in, float* __restrict__ x, int C, int HW) {
  const int C8 = C >> 3;
  const int ng = blockIdx.y;
  const int g = ng % C8, n = ng / C8;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const bf16x8 v = __builtin_bit_cast(bf16x8, in[(size_t)ng * HW + pix]);
  const size_t dst = ((size_t)n * C + 8 * g) * HW + pix;
#pragma unroll
  for (int e = 0; e < 8; ++e) x[dst + (size_t)e * HW] = (float)v[e];
}

int half_check(const char* who, int32_t N, int32_t C, int32_t HW) {
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0 && (C % 8) == 0, "%s: C must be a positive multiple of 8", who);
  MCD_REQUIRE((int64_t)N * (C / 8) <= 65535, "%s: N*C/8 exceeds the grid limit", who);
  return 0;
}

}  // namespace

extern "C" int mcdseg_bn_apply_half(const void* z16, const float* z_bound, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, const void* res_cb, const float* res_bound, void* y_cb, const float* y_bound, int32_t N,
                                    int32_t C, int32_t HW, int32_t relu, void* stream) {
  MCD_REQUIRE(z16 && z_bound && mean && rstd && gamma && beta && y_cb && y_bound, "bn_apply_half: null pointer");
  MCD_REQUIRE(res_cb == nullptr || res_bound != nullptr, "bn_apply_half: the residual companion needs its bound");
  if (int rc = half_check("bn_apply_half", N, C, HW)) return rc;
  const dim3 grid(ceil_div(HW, 256 * BN_H_UNITS), N * (C / 8));
  if (res_cb != nullptr)
    hipLaunchKernelGGL(bn_apply_h_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)z16, z_bound, mean, rstd, gamma, beta,
                       (const uint4*)res_cb, res_bound, (uint4*)y_cb, y_bound, C / 8, HW, relu);
  else
    hipLaunchKernelGGL(bn_apply_h_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const uint4*)z16, z_bound, mean, rstd, gamma, beta,
                       (const uint4*)nullptr, (const float*)nullptr, (uint4*)y_cb, y_bound, C / 8, HW, relu);
  MCD_LAUNCH_CHECK("bn_apply_half");
  return 0;
}

extern "C" size_t mcdseg_bn_bwd_half_workspace_bytes(int32_t N, int32_t C, int32_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0 || (C & 7) != 0) return 0;
  return (size_t)N * bwd_h_slices(N, C, HW) * 3 * C * sizeof(float);
}

// mask_kind: 0 no ReLU, 2 the mask recomputed from z (a ReLU group without residual: needs gamma / beta), 4 from the activation's piece y_cb
extern "C" int mcdseg_bn_bwd_reduce_half(const void* dy16, const void* y_cb, const void* z16, const float* z_bound, const float* mean,
                                         const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                                         float* dz_bound, int32_t mask_kind, int32_t train, int32_t N, int32_t C, int32_t HW, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(dy16 && z16 && z_bound && mean && rstd && gamma && beta && dgamma && dbeta && workspace, "bn_bwd_reduce_half: null pointer");
  MCD_REQUIRE(mask_kind == 0 || mask_kind == 2 || (mask_kind == 4 && y_cb != nullptr), "bn_bwd_reduce_half: mask_kind must be 0, 2 or 4 (with y_cb)");
  if (int rc = half_check("bn_bwd_reduce_half", N, C, HW)) return rc;
  MCD_REQUIRE(workspace_bytes >= mcdseg_bn_bwd_half_workspace_bytes(N, C, HW), "bn_bwd_reduce_half: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int sl = bwd_h_slices(N, C, HW);
  const int per = round_up(ceil_div(HW, sl), 256);
  const dim3 grid(ceil_div(HW, per), N * (C / 8));
  const int S = N * (int)grid.x;
#define MCD_RH(M)                                                                                                                       \
  hipLaunchKernelGGL(bn_bwd_reduce_h_kernel<M>, grid, dim3(256), 0, st, (const uint4*)dy16, (const uint4*)y_cb, (const uint4*)z16, z_bound, \
                     mean, rstd, gamma, beta, (float*)workspace, C, HW, per, dz_bound)
  if (mask_kind == 0) MCD_RH(0); else if (mask_kind == 2) MCD_RH(2); else MCD_RH(4);
#undef MCD_RH
  MCD_LAUNCH_CHECK("bn_bwd_reduce_half");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 4)), dim3(256), 0, st, (const float*)workspace, S, C, dgamma, dbeta, gamma, rstd,
                     (float)N * (float)HW, train, dz_bound);
  MCD_LAUNCH_CHECK("bn_bwd_finalize");
  return 0;
}

extern "C" int mcdseg_bn_bwd_apply_half(const void* dy16, const void* y_cb, const void* z16, const float* z_bound, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta, const float* dgamma, const float* dbeta,
                                        void* dz_cb, const float* dz_bound, void* dres16, int32_t mask_kind, int32_t train, int32_t N, int32_t C,
                                        int32_t HW, void* stream) {
  MCD_REQUIRE(dy16 && z16 && z_bound && mean && rstd && gamma && beta && dz_cb && dz_bound, "bn_bwd_apply_half: null pointer");
  MCD_REQUIRE(!train || (dgamma && dbeta), "bn_bwd_apply_half: train mode needs dgamma/dbeta");
  MCD_REQUIRE(mask_kind == 0 || mask_kind == 2 || (mask_kind == 4 && y_cb != nullptr), "bn_bwd_apply_half: mask_kind must be 0, 2 or 4 (with y_cb)");
  if (int rc = half_check("bn_bwd_apply_half", N, C, HW)) return rc;
  const dim3 grid(ceil_div(HW, 256 * BN_H_UNITS), N * (C / 8));
  hipStream_t st = (hipStream_t)stream;
#define MCD_AH(M)                                                                                                                      \
  hipLaunchKernelGGL(bn_bwd_apply_h_kernel<M>, grid, dim3(256), 0, st, (const uint4*)dy16, (const uint4*)y_cb, (const uint4*)z16, z_bound, \
                     mean, rstd, gamma, beta, dgamma, dbeta, (uint4*)dz_cb, dz_bound, (uint4*)dres16, N, C / 8, HW, train)
  if (mask_kind == 0) MCD_AH(0); else if (mask_kind == 2) MCD_AH(2); else MCD_AH(4);
#undef MCD_AH
  MCD_LAUNCH_CHECK("bn_bwd_apply_half");
  return 0;
}

extern "C" int mcdseg_pack_bf16_units(const float* x, void* out16, int32_t N, int32_t C, int32_t HW, void* stream) {
  MCD_REQUIRE(x && out16, "pack_bf16_units: null pointer");
  if (int rc = half_check("pack_bf16_units", N, C, HW)) return rc;
  hipLaunchKernelGGL(pack_bf16_units_kernel, dim3(ceil_div(HW, 256), N * (C / 8)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)out16, C, HW);
  MCD_LAUNCH_CHECK("pack_bf16_units");
  return 0;
}

extern "C" int mcdseg_unpack_bf16_units(const void* in16, float* x, int32_t N, int32_t C, int32_t HW, void* stream) {
  MCD_REQUIRE(x && in16, "unpack_bf16_units: null pointer");
  if (int rc = half_check("unpack_bf16_units", N, C, HW)) return rc;
  hipLaunchKernelGGL(unpack_bf16_units_kernel, dim3(ceil_div(HW, 256), N * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in16, x, C, HW);
  MCD_LAUNCH_CHECK("unpack_bf16_units");
  return 0;
}
