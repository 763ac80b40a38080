// Fused per-pixel softmax -> weighted cross-entropy (one or two heads) -> L1 discrepancy, with the
// gradients w.r.t. both logit tensors written in the same pass.
//
// Reference ops this replaces (each a separate full-tensor ATen pass over [N,C,H,W]):
//   CrossEntropyLoss2d = log_softmax(dim 1) + weighted-mean NLL   (loss.py:7-13)
//   Diff2d             = mean |softmax(o1) - softmax(o2)|          (loss.py:93-100)
// Maths, per pixel i, p = softmax(z), W = sum_i w[y_i], M = N*C*H*W:
//   dCE/dz_c   = w[y_i] (p_c - [c == y_i]) / W
//   dDiff/dz1_c =  p1_c (s_c - sum_k s_k p1_k),  dDiff/dz2_c = -p2_c (s_c - sum_k s_k p2_k),  s = sign(p1-p2)/M
//
// One lane owns one pixel and keeps all C logits of both heads in registers; the class stride of NCHW is
// H*W so every per-class load/store of a wave is one contiguous 256-B run.  Algorithmic traffic:
// (2C reads + 2C writes) * 4 B + 8 B label per pixel; everything else stays in registers.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int LOSS_BLOCK = 256;

__global__ __launch_bounds__(256) void label_wsum_kernel(const int64_t* __restrict__ labels, const float* __restrict__ cw,
                                                         int64_t ignore_index, int C, int64_t P, float* __restrict__ part) {
  float s = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t y = labels[i];
    if (y != ignore_index && y >= 0 && y < C) s += cw ? cw[y] : 1.f;
  }
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void wsum_finalize_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  __shared__ double sh[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// The per-pixel maths shared by the two front ends below: a[] / b[] hold the pixel's logits (-inf past C) and leave as its
// probabilities; the gradients go to g1 / g2 at base + c * HW.
// Precondition: a[c] = b[c] = -inf for c >= C, so the exponentials need no per-class select (exp(-inf) = 0).  The class loops
// are cut into groups by branches on an opaque, always-true scalar: basic-block boundaries are the only fence the instruction
// scheduler respects here, and without them it interleaves the exponential chains of all classes (about seven registers
// each) and spills hundreds of registers.
#define MCD_OPAQUE_TRUE(name) \
  int name = 1;               \
  asm volatile("" : "+s"(name))

// exp of a NON-POSITIVE argument (a logit minus the pixel's maximum): v_exp_f32(x log2 e), two instructions where expf spends fifteen on
// a range reduction and a scaling that such an argument never needs (round 6: the fused loss kernel evaluates 82 of these per pixel and
// was bound by them after round 5's wait fix).  Relative error: the instruction's 1 ulp plus |x| 2^-24 from rounding x log2 e -- below
// 2e-7 for every class that carries probability (x > -3); exp(-inf) = 0 for the padding classes; results below 2^-126 flush to zero.
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

template <int NCMAX, bool TWO>
__device__ __forceinline__ void pixel_losses(float (&a)[NCMAX], float (&b)[NCMAX], int y, float wy, float ce_coef, float diff_coef,
                                             const float* __restrict__ losses_w, float* __restrict__ g1, float* __restrict__ g2,
                                             size_t base, size_t HW, int C, float inv_m, float& ce1, float& ce2, float& dsum) {
  float m1 = a[0], m2 = TWO ? b[0] : 0.f;
#pragma unroll
  for (int c = 1; c < NCMAX; ++c) {
    m1 = fmaxf(m1, a[c]);
    if (TWO) m2 = fmaxf(m2, b[c]);
  }
  float s1 = 0.f, s2 = 0.f, zy1 = 0.f, zy2 = 0.f;
#pragma unroll
  for (int c0 = 0; c0 < NCMAX; c0 += 4) {
    MCD_OPAQUE_TRUE(go);
    if (go) {
#pragma unroll
      for (int c = c0; c < c0 + 4; ++c) {
        if (c >= NCMAX) continue;
        if (c == y) {
          zy1 = a[c];
          if (TWO) zy2 = b[c];
        }
        a[c] = exp_nonpos(a[c] - m1);
        s1 += a[c];
        if (TWO) {
          b[c] = exp_nonpos(b[c] - m2);
          s2 += b[c];
        }
      }
    }
  }
  if (y >= 0) {
    ce1 = wy * ((m1 + logf(s1)) - zy1);
    if (TWO) ce2 = wy * ((m2 + logf(s2)) - zy2);
  }
  const float r1 = 1.f / s1, r2 = TWO ? 1.f / s2 : 0.f;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int c0 = 0; c0 < NCMAX; c0 += 8) {
    MCD_OPAQUE_TRUE(go);
    if (go) {
#pragma unroll
      for (int c = c0; c < c0 + 8; ++c) {
        if (c >= NCMAX) continue;
        a[c] *= r1;
        if (TWO) {
          b[c] *= r2;
          const float d = a[c] - b[c];
          const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
          dsum += fabsf(d);
          t1 = fmaf(sg, a[c], t1);
          t2 = fmaf(sg, b[c], t2);
        }
      }
    }
  }
  if (g1 != nullptr || g2 != nullptr) {
    const float kce = (ce_coef != 0.f && y >= 0) ? ce_coef * wy / losses_w[3] : 0.f;
    const float kd = diff_coef * inv_m;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      if (c < C) {
        const float oh = (c == y) ? 1.f : 0.f;
        float sg = 0.f;
        if (TWO) {
          const float d = a[c] - b[c];
          sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        }
        if (g1 != nullptr) g1[base + (size_t)c * HW] = kce * (a[c] - oh) + (TWO ? kd * a[c] * (sg - t1) : 0.f);
        if (TWO && g2 != nullptr) g2[base + (size_t)c * HW] = kce * (b[c] - oh) - kd * b[c] * (sg - t2);
      }
    }
  }
}

template <int NCMAX, bool TWO>
__global__ __launch_bounds__(256) void softmax_ce_l1_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ cw,
                                                            int64_t ignore_index, float ce_coef, float diff_coef,
                                                            const float* __restrict__ losses_w, float* __restrict__ g1,
                                                            float* __restrict__ g2, float* __restrict__ part, int C, int HW,
                                                            int64_t P, float inv_m) {
  const int64_t pix = blockIdx.x * (int64_t)LOSS_BLOCK + threadIdx.x;
  const bool valid = pix < P;
  float ce1 = 0.f, ce2 = 0.f, dsum = 0.f;
  if (valid) {
    const int64_t n = pix / HW;
    const int hw = (int)(pix - n * HW);
    const size_t base = (size_t)n * C * HW + hw;
    float a[NCMAX], b[NCMAX];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      a[c] = (c < C) ? z1[base + (size_t)c * HW] : -INFINITY;
      b[c] = (TWO && c < C) ? z2[base + (size_t)c * HW] : -INFINITY;
    }
    int y = -1;
    float wy = 0.f;
    if (labels != nullptr) {
      const int64_t yl = labels[pix];
      if (yl != ignore_index && yl >= 0 && yl < C) {
        y = (int)yl;
        wy = cw ? cw[y] : 1.f;
      }
    }
    pixel_losses<NCMAX, TWO>(a, b, y, wy, ce_coef, diff_coef, losses_w, g1, g2, base, (size_t)HW, C, inv_m, ce1, ce2, dsum);
  }
  __shared__ float sh[3][4];
  ce1 = wave_sum(ce1);
  ce2 = wave_sum(ce2);
  dsum = wave_sum(dsum);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = ce1;
    sh[1][threadIdx.x >> 6] = ce2;
    sh[2][threadIdx.x >> 6] = dsum;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    part[(size_t)blockIdx.x * 3 + q] = (sh[q][0] + sh[q][1]) + (sh[q][2] + sh[q][3]);
  }
}

// The same losses with the x8 learned up-sampler (up8.hip) computed on the fly: the classifiers of the MCD configuration are
// nothing but that up-sampler (models/dilated_fcn.py:357-366 behind DRNSegPixelClassifier), so their full-resolution logits
// need never exist in memory -- each is four multiply-adds on the 64x smaller score map.
//
// Persistent workgroups of 8 waves, one per CU.  The 16x16 kernels of every class and head stay in LDS for the workgroup's
// life (re-ordered so that the four taps a pixel needs are one 16-byte read); the work items are patches of 8 rows x 64
// columns whose rows share their two input rows -- output rows 8i-4 .. 8i+3 -- so an item stages only 2 x 10 scores per class
// and head, fetched into registers while the previous item is being computed.  Wave k of the workgroup owns the patch row with
// kernel rows (k, k+8).  The multiply-adds run in the order of up8_fwd_kernel (absent inputs staged as zeros), so logits,
// the losses' summands and the gradients equal the two-pass result bit for bit; only the summation order of the loss
// values differs (per lane over its items, then the block, then fp64 over blocks).
constexpr int UP_COLS = 64, UP_JP = 9, UP_NT = 512;  // UP_JP: input-column pairs (ix-1, ix) a 64-pixel row segment touches

template <int NCMAX, bool TWO>
__global__ __launch_bounds__(UP_NT) void up8_softmax_ce_l1_kernel(const float* __restrict__ s1, const float* __restrict__ w1,
                                                                  const float* __restrict__ s2, const float* __restrict__ w2,
                                                                  const int64_t* __restrict__ labels, const float* __restrict__ cw,
                                                                  int64_t ignore_index, float ce_coef, float diff_coef,
                                                                  const float* __restrict__ losses_w, float* __restrict__ g1,
                                                                  float* __restrict__ g2, float* __restrict__ part, int N, int C,
                                                                  int Hi, int Wi, float inv_m) {
  extern __shared__ __attribute__((aligned(16))) float up_sm[];
  constexpr int HEADS = TWO ? 2 : 1;
  constexpr int SREG = (HEADS * NCMAX * UP_JP * 4 + UP_NT - 1) / UP_NT;  // staged scores per thread and item
  // Class stride NCMAX, not C, and the four taps / four scores of a pixel as one 16-byte unit: every LDS read below is then
  // the lane's base address plus an immediate offset.  (With C in the stride, or with the scores as plain rows read by
  // ds_read2_b32 -- whose offset field reaches 1 KB -- the 2 x NCMAX addresses become registers of their own and the
  // kernel spills hundreds of them.)
  float* wl = up_sm;                         // [head][NCMAX][ky0 8][kx0 8][a 2][b 2]
  float* sin = up_sm + HEADS * NCMAX * 256;  // [head][NCMAX][pair UP_JP][a 2][b 2]: score (row iyg - a, column ixb + pair + 1 - b)
  const int Wo = 8 * Wi, Ho = 8 * Hi;
  const int nseg = (Wo + UP_COLS - 1) / UP_COLS;
  const int items = N * (Hi + 1) * nseg;
  const int ky0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < HEADS * NCMAX * 256; i += UP_NT) {
    const int b = i & 1, a = (i >> 1) & 1, kx = (i >> 2) & 7, ky = (i >> 5) & 7, hc = i >> 8;
    const int c = hc % NCMAX;
    // classes past C: taps (1, 0, 0, 0) against scores (-inf, 0, 0, 0) below give the logit -inf with no test in the pixel loop
    wl[i] = c < C ? (hc >= NCMAX ? w2 : w1)[c * 256 + (ky + 8 * a) * 16 + kx + 8 * b] : ((a | b) == 0 ? 1.f : 0.f);
  }
  constexpr int nstage = HEADS * NCMAX * UP_JP * 4;
  float sreg[SREG];
  auto fetch = [&](int item) {  // scores of one item -> registers (zeros outside the map)
    const int seg = item % nseg, r = item / nseg;
    const int iyg = r % (Hi + 1), n = r / (Hi + 1);
    const int ixb = seg * (UP_COLS / 8) - 1;
#pragma unroll
    for (int k = 0; k < SREG; ++k) {
      const int i = threadIdx.x + k * UP_NT;
      float v = 0.f;
      if (i < nstage) {
        const int b = i & 1, a = (i >> 1) & 1, q = i >> 2;
        const int j = q % UP_JP, hc = q / UP_JP;
        const int c = hc % NCMAX;
        const int iy = iyg - a, ix = ixb + j + 1 - b;
        if (c >= C)
          v = (a | b) == 0 ? -INFINITY : 0.f;
        else if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi)
          v = (hc >= NCMAX ? s2 : s1)[(((size_t)n * C + c) * Hi + iy) * Wi + ix];
      }
      sreg[k] = v;
    }
  };
  float ce1 = 0.f, ce2 = 0.f, dsum = 0.f;
  int item = blockIdx.x;
  if (item < items) fetch(item);
  for (; item < items; item += gridDim.x) {
    __syncthreads();  // the previous item's readers are done (and, first time round, the kernels are staged)
#pragma unroll
    for (int k = 0; k < SREG; ++k) {
      const int i = threadIdx.x + k * UP_NT;
      if (i < nstage) sin[i] = sreg[k];
    }
    __syncthreads();
    if (item + (int)gridDim.x < items) fetch(item + gridDim.x);
    const int seg = item % nseg, r = item / nseg;
    const int iyg = r % (Hi + 1), n = r / (Hi + 1);
    const int oy = 8 * iyg - 4 + ky0;
    const int ox = seg * UP_COLS + lane;
    if (oy >= 0 && oy < Ho && ox < Wo) {
      // the class count re-read as an opaque scalar: otherwise the NCMAX "c < C" store guards are hoisted out of the item loop
      // and their results spill
      int Cv = C;
      asm volatile("" : "+s"(Cv));
      const int kx0 = (ox + 4) & 7;
      const int jp = ((ox + 4) >> 3) - seg * (UP_COLS / 8);  // pair whose b = 0 member is this pixel's right-hand input column
      // the lane's LDS offsets, opaque too: the kernel taps do not depend on the item ((ox + 4) & 7 is the lane's), and left
      // alone the compiler hoists all 2 x NCMAX 16-byte reads out of the item loop
      int woff = (ky0 * 8 + kx0) * 4, soff = jp * 4;
      asm volatile("" : "+v"(woff), "+v"(soff));
      float a[NCMAX], b[NCMAX];
#pragma unroll
      for (int c0 = 0; c0 < NCMAX; c0 += 4) {
        MCD_OPAQUE_TRUE(go);  // groups of four classes in basic blocks of their own (see pixel_losses)
        if (go) {
#pragma unroll
          for (int c = c0; c < c0 + 4; ++c) {
            if (c >= NCMAX) continue;
            b[c] = -INFINITY;
#pragma unroll
            for (int h = 0; h < HEADS; ++h) {
              const float4 wv = *reinterpret_cast<const float4*>(wl + woff + (h * NCMAX + c) * 256);
              const float4 sv = *reinterpret_cast<const float4*>(sin + soff + (h * NCMAX + c) * (UP_JP * 4));
              float o = 0.f;
              o = fmaf(sv.x, wv.x, o);
              o = fmaf(sv.y, wv.y, o);
              o = fmaf(sv.z, wv.z, o);
              o = fmaf(sv.w, wv.w, o);
              if (h == 0)
                a[c] = o;
              else
                b[c] = o;
            }
          }
        }
      }
      const size_t HW = (size_t)Ho * Wo;
      const size_t hw = (size_t)oy * Wo + ox;
      int y = -1;
      float wy = 0.f;
      if (labels != nullptr) {
        const int64_t yl = labels[(size_t)n * HW + hw];
        if (yl != ignore_index && yl >= 0 && yl < C) {
          y = (int)yl;
          wy = cw ? cw[y] : 1.f;
        }
      }
      float e1 = 0.f, e2 = 0.f, ds = 0.f;
      pixel_losses<NCMAX, TWO>(a, b, y, wy, ce_coef, diff_coef, losses_w, g1, g2, (size_t)n * C * HW + hw, HW, Cv, inv_m, e1, e2, ds);
      ce1 += e1;
      ce2 += e2;
      dsum += ds;
    }
  }
  __shared__ float sh[3][UP_NT / 64];
  ce1 = wave_sum(ce1);
  ce2 = wave_sum(ce2);
  dsum = wave_sum(dsum);
  if (lane == 0) {
    sh[0][ky0] = ce1;
    sh[1][ky0] = ce2;
    sh[2][ky0] = dsum;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    float t = 0.f;
    for (int k = 0; k < UP_NT / 64; ++k) t += sh[q][k];
    part[(size_t)blockIdx.x * 3 + q] = t;
  }
}

// ---- the same kernel with nothing between an item's stores and the next item's inputs ----------------------------------------
// The kernel above fetches the next item's scores into registers (and reads labels and class weights from memory inside the item):
// loads the compiler tracks.  The counter that tracks them (vmcnt) also counts the item's 2 C gradient stores, in issue order, so
// the wait in front of the first use of any of those loads -- `s_waitcnt vmcnt(0)`, the stores sit behind data-dependent branches the
// compiler cannot count -- drains every store of the previous item: the workgroup, the only one on its CU, alternates between
// computing and writing (at the benchmark's shape 0.75 ms per launch with the gradients, 0.50 ms without them, for 0.34 ms of HBM writes at
// the rate a copy reaches; this kernel: 0.58 / 0.38 ms -- tools/probes/up8_loss_probe.py).
// Here every input of the loop arrives by LDS-DMA, which the compiler does not track, into two buffers: the next item's scores
// (one dword per lane, the staging layout as it is), its labels (16 B per lane: one row segment of 64 labels per wave) -- issued
// BEFORE the current item's stores, so that a counted `s_waitcnt vmcnt(63)` at the top of the next item proves them complete
// while up to 63 of the 2 C stores of each wave are still in flight (a wave that issued fewer than 63 stores waits for 0); class
// weights and the up-sampling kernels stay in LDS for the workgroup's life.  One barrier per item instead of two.  Classes past C
// (EXACT = false) get their -inf by a select instead of through staged constants (the DMA fills those slots with zeros).
// Arithmetic, item order and summation order are the register kernel's: gradients and loss values are bitwise the same.
constexpr unsigned UP_OOB = 0x80000000u;
template <int NCMAX, bool TWO>
struct UpDmaLayout {
  static constexpr int HEADS = TWO ? 2 : 1;
  static constexpr int SLOTS = NCMAX * UP_JP * 4;            // staged scores per head and item
  static constexpr int SPK = (SLOTS + UP_NT - 1) / UP_NT;    // DMA instructions per head, item and wave
  static constexpr int SP = SPK * UP_NT;                     // floats per head and buffer (the tail takes the DMA's zero fill)
  static constexpr int W_FLOATS = HEADS * NCMAX * 256;
  static constexpr int S_FLOATS = 2 * HEADS * SP;
  static constexpr int LAB_BYTES = 2 * (UP_NT / 64) * 1024;  // per buffer and wave: 64 labels (512 B) + 512 B of zero fill
  static constexpr int CW_FLOATS = (NCMAX + 63) / 64 * 64;
  static constexpr size_t BYTES = (size_t)(W_FLOATS + S_FLOATS + CW_FLOATS) * 4 + LAB_BYTES;
};

template <int NCMAX, bool TWO, bool EXACT>
__global__ __launch_bounds__(UP_NT) void up8_softmax_ce_l1_dma_kernel(const float* __restrict__ s1, const float* __restrict__ w1,
                                                                      const float* __restrict__ s2, const float* __restrict__ w2,
                                                                      const int64_t* __restrict__ labels, const float* __restrict__ cw,
                                                                      int64_t ignore_index, float ce_coef, float diff_coef,
                                                                      const float* __restrict__ losses_w, float* __restrict__ g1,
                                                                      float* __restrict__ g2, float* __restrict__ part, int N, int C,
                                                                      int Hi, int Wi, float inv_m) {
  extern __shared__ __attribute__((aligned(16))) float up_sm[];
  using L = UpDmaLayout<NCMAX, TWO>;
  constexpr int HEADS = L::HEADS, SPK = L::SPK, SP = L::SP;
  float* wl = up_sm;                                                   // [head][NCMAX][ky0 8][kx0 8][a 2][b 2]
  float* sin0 = up_sm + L::W_FLOATS;                                   // [buffer 2][head][SP]: [NCMAX][pair UP_JP][a 2][b 2] + tail
  float* cwl = sin0 + L::S_FLOATS;                                     // [NCMAX] class weights (1 without)
  unsigned char* lab0 = reinterpret_cast<unsigned char*>(cwl + L::CW_FLOATS);  // [buffer 2][wave 8][1 KB]
  const int Wo = 8 * Wi, Ho = 8 * Hi;
  const int nseg = (Wo + UP_COLS - 1) / UP_COLS;
  const int items = N * (Hi + 1) * nseg;
  const int lane = threadIdx.x & 63;
  const int ky0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < HEADS * NCMAX * 256; i += UP_NT) {
    const int b = i & 1, a = (i >> 1) & 1, kx = (i >> 2) & 7, ky = (i >> 5) & 7, hc = i >> 8;
    const int c = hc % NCMAX;
    wl[i] = c < C ? (hc >= NCMAX ? w2 : w1)[c * 256 + (ky + 8 * a) * 16 + kx + 8 * b] : 0.f;
  }
  for (int i = threadIdx.x; i < NCMAX; i += UP_NT) cwl[i] = (cw != nullptr && i < C) ? cw[i] : 1.f;
  const mcd_i32x4 rs1 = mcd_raw_rsrc(s1, N * C * Hi * Wi * 4);
  const mcd_i32x4 rs2 = mcd_raw_rsrc(TWO ? s2 : s1, N * C * Hi * Wi * 4);
  const mcd_i32x4 rsl = mcd_raw_rsrc(labels != nullptr ? (const void*)labels : (const void*)s1, labels != nullptr ? N * Ho * Wo * 8 : 0);
  const unsigned lds_s = (unsigned)(size_t)(__attribute__((address_space(3))) float*)sin0;
  const unsigned lds_l = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lab0;
  auto issue = [&](int item, int buf) {  // the inputs of one item -> LDS buffer `buf` (zeros outside the map)
    const int seg = item % nseg, r = item / nseg;
    const int iyg = r % (Hi + 1), n = r / (Hi + 1);
    const int ixb = seg * (UP_COLS / 8) - 1;
#pragma unroll
    for (int k = 0; k < SPK; ++k) {
      const int i = threadIdx.x + k * UP_NT;
      const int b = i & 1, a = (i >> 1) & 1, q = i >> 2;
      const int j = q % UP_JP, c = q / UP_JP;
      const int iy = iyg - a, ix = ixb + j + 1 - b;
      const bool ok = c < C && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      const unsigned voff = ok ? (unsigned)((((n * C + c) * Hi + iy) * Wi + ix) * 4) : UP_OOB;
      mcd_hidden_dma<4>(rs1, __builtin_amdgcn_readfirstlane(lds_s + 4u * ((buf * HEADS) * SP + k * UP_NT + ky0 * 64)), voff);
      if (TWO) mcd_hidden_dma<4>(rs2, __builtin_amdgcn_readfirstlane(lds_s + 4u * ((buf * HEADS + 1) * SP + k * UP_NT + ky0 * 64)), voff);
    }
    if (labels != nullptr) {
      const int oy = 8 * iyg - 4 + ky0;
      const unsigned voff = (lane < 32 && (unsigned)oy < (unsigned)Ho) ? (unsigned)(((n * Ho + oy) * Wo + seg * UP_COLS + 2 * lane) * 8) : UP_OOB;
      mcd_hidden_dma<16>(rsl, __builtin_amdgcn_readfirstlane(lds_l + 1024u * (buf * (UP_NT / 64) + ky0)), voff);
    }
  };
  const int nst = (g1 != nullptr ? C : 0) + ((TWO && g2 != nullptr) ? C : 0);  // stores per wave and item
  float ce1 = 0.f, ce2 = 0.f, dsum = 0.f;
  int item = blockIdx.x, buf = 0;
  int behind = 0;  // wave-uniform: the stores this wave issued after its last DMA
  __syncthreads();  // (the kernels and class weights are staged before anybody's DMA could be mistaken for them -- and for the first barrier below)
  if (item < items) issue(item, 0);
  for (; item < items; item += gridDim.x, buf ^= 1) {
    // this item's inputs have landed (issued before `behind` stores: in-order counter), every wave is done with the other buffer
    if (__builtin_amdgcn_readfirstlane(behind) >= 63)
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (outside the branch: one barrier whatever the compiler makes of it)
    if (item + (int)gridDim.x < items) issue(item + gridDim.x, buf ^ 1);
    behind = 0;
    const int seg = item % nseg, r = item / nseg;
    const int iyg = r % (Hi + 1), n = r / (Hi + 1);
    const int oy = 8 * iyg - 4 + ky0;
    const int ox = seg * UP_COLS + lane;
    if (oy >= 0 && oy < Ho) {  // wave-uniform
      behind = nst;
      if (ox < Wo) {
        int Cv = C;
        asm volatile("" : "+s"(Cv));
        const int kx0 = (ox + 4) & 7;
        const int jp = ((ox + 4) >> 3) - seg * (UP_COLS / 8);
        int woff = (ky0 * 8 + kx0) * 4, soff = jp * 4 + buf * HEADS * SP;
        asm volatile("" : "+v"(woff), "+v"(soff));
        float a[NCMAX], b[NCMAX];
#pragma unroll
        for (int c0 = 0; c0 < NCMAX; c0 += 4) {
          MCD_OPAQUE_TRUE(go);
          if (go) {
#pragma unroll
            for (int c = c0; c < c0 + 4; ++c) {
              if (c >= NCMAX) continue;
              b[c] = -INFINITY;
#pragma unroll
              for (int h = 0; h < HEADS; ++h) {
                const float4 wv = *reinterpret_cast<const float4*>(wl + woff + (h * NCMAX + c) * 256);
                const float4 sv = *reinterpret_cast<const float4*>(sin0 + soff + h * SP + c * (UP_JP * 4));
                float o = 0.f;
                o = fmaf(sv.x, wv.x, o);
                o = fmaf(sv.y, wv.y, o);
                o = fmaf(sv.z, wv.z, o);
                o = fmaf(sv.w, wv.w, o);
                if (!EXACT && c >= Cv) o = -INFINITY;
                if (h == 0)
                  a[c] = o;
                else
                  b[c] = o;
              }
            }
          }
        }
        const size_t HW = (size_t)Ho * Wo;
        const size_t hw = (size_t)oy * Wo + ox;
        int y = -1;
        float wy = 0.f;
        if (labels != nullptr) {
          const int64_t yl = *reinterpret_cast<const int64_t*>(lab0 + (buf * (UP_NT / 64) + ky0) * 1024 + lane * 8);
          if (yl != ignore_index && yl >= 0 && yl < C) {
            y = (int)yl;
            wy = cwl[y];
          }
        }
        float e1 = 0.f, e2 = 0.f, ds = 0.f;
        pixel_losses<NCMAX, TWO>(a, b, y, wy, ce_coef, diff_coef, losses_w, g1, g2, (size_t)n * C * HW + hw, HW, Cv, inv_m, e1, e2, ds);
        ce1 += e1;
        ce2 += e2;
        dsum += ds;
      }
    }
  }
  __shared__ float sh[3][UP_NT / 64];
  ce1 = wave_sum(ce1);
  ce2 = wave_sum(ce2);
  dsum = wave_sum(dsum);
  if (lane == 0) {
    sh[0][ky0] = ce1;
    sh[1][ky0] = ce2;
    sh[2][ky0] = dsum;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int q = threadIdx.x;
    float t = 0.f;
    for (int k = 0; k < UP_NT / 64; ++k) t += sh[q][k];
    part[(size_t)blockIdx.x * 3 + q] = t;
  }
}

// has_ce: a cross-entropy term was requested.  With an all-background / all-ignored batch the normaliser W = sum w[y] is 0
// and the reference's weighted mean is 0/0: nn.NLLLoss2d returns NaN and NaN gradients (loss.py:7-13).  The kernel does the
// same -- value here, gradients through kce = ce_coef * w[y] / W in the main kernel -- so the failure is as visible as in
// the reference instead of a healthy-looking 0 in the log while NaN gradients reach the optimizer.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part, int64_t nblk, float* __restrict__ losses,
                                                            double inv_m, int has_ce) {
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t i = threadIdx.x; i < nblk; i += 256) {
    s[0] += (double)part[i * 3 + 0];
    s[1] += (double)part[i * 3 + 1];
    s[2] += (double)part[i * 3 + 2];
  }
  __shared__ double sh[3][4];
  for (int q = 0; q < 3; ++q) {
    const double v = wave_sum_d(s[q]);
    if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double W = (double)losses[3];
    const double c1 = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    const double c2 = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    const double d = (sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]);
    losses[0] = has_ce ? (float)(c1 / W) : 0.f;
    losses[1] = has_ce ? (float)(c2 / W) : 0.f;
    losses[2] = (float)(d * inv_m);
  }
}

// Inference tail (adapt_tester.py:101-124, util.py:44-48): o = z1 or (z1 + z2)/2; label = argmax over the first C_used
// classes (the background channel is excluded unless it was trained); entropy term = sum_c p_c log(p_c + 1e-6) over ALL classes.
template <int NCMAX>
__global__ __launch_bounds__(256) void predict_kernel(const float* __restrict__ z1, const float* __restrict__ z2,
                                                      uint8_t* __restrict__ labels, float* __restrict__ part, int C, int C_used,
                                                      int HW, int64_t P) {
  const int64_t pix = blockIdx.x * (int64_t)LOSS_BLOCK + threadIdx.x;
  float ent = 0.f;
  if (pix < P) {
    const int64_t n = pix / HW;
    const int hw = (int)(pix - n * HW);
    const size_t base = (size_t)n * C * HW + hw;
    float a[NCMAX];
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      float v = -INFINITY;
      if (c < C) {
        v = z1[base + (size_t)c * HW];
        if (z2 != nullptr) v = (v + z2[base + (size_t)c * HW]) / 2.f;
      }
      a[c] = v;
    }
    float m = a[0], mu = a[0];
    int best = 0;
#pragma unroll
    for (int c = 1; c < NCMAX; ++c) {
      m = fmaxf(m, a[c]);
      if (c < C_used && a[c] > mu) {
        mu = a[c];
        best = c;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c) {
      a[c] = (c < C) ? expf(a[c] - m) : 0.f;
      s += a[c];
    }
    const float r = 1.f / s;
#pragma unroll
    for (int c = 0; c < NCMAX; ++c)
      if (c < C) {
        const float pc = a[c] * r;
        ent += pc * logf(pc + 1e-6f);
      }
    labels[pix] = (uint8_t)best;
  }
  __shared__ float sh[4];
  ent = wave_sum(ent);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ent;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void predict_finalize_kernel(const float* __restrict__ part, int64_t nblk, double neg_inv_m,
                                                               float* __restrict__ out) {
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < nblk; i += 256) s += (double)part[i];
  __shared__ double sh[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * neg_inv_m);
}

__global__ void scale_by_device_scalar_kernel(float* __restrict__ buf, const float* __restrict__ scale, int64_t n4, int64_t n) {
  const float s = *scale;
  float4* b4 = reinterpret_cast<float4*>(buf);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = b4[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    b4[i] = v;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) buf[i] *= s;
}

int wsum_blocks(int64_t P) {
  const int64_t b = ceil_div64(P, 256 * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

template <int NCMAX>
void launch_loss(bool two, dim3 grid, hipStream_t st, const float* z1, const float* z2, const int64_t* labels, const float* cw,
                 int64_t ignore_index, float ce_coef, float diff_coef, const float* losses, float* g1, float* g2, float* part,
                 int C, int HW, int64_t P, float inv_m) {
  if (two)
    hipLaunchKernelGGL((softmax_ce_l1_kernel<NCMAX, true>), grid, dim3(LOSS_BLOCK), 0, st, z1, z2, labels, cw, ignore_index, ce_coef,
                       diff_coef, losses, g1, g2, part, C, HW, P, inv_m);
  else
    hipLaunchKernelGGL((softmax_ce_l1_kernel<NCMAX, false>), grid, dim3(LOSS_BLOCK), 0, st, z1, z2, labels, cw, ignore_index,
                       ce_coef, diff_coef, losses, g1, g2, part, C, HW, P, inv_m);
}

template <int NCMAX>
int launch_up_loss(bool two, int blocks, size_t lds, hipStream_t st, const float* s1, const float* w1, const float* s2, const float* w2,
                   const int64_t* labels, const float* cw, int64_t ignore_index, float ce_coef, float diff_coef, const float* losses,
                   float* g1, float* g2, float* part, int N, int C, int Hi, int Wi, float inv_m) {
  auto go = [&](auto kern) {
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        mcdseg_set_error("up8_softmax_ce_l1: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return -5;
      }
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(UP_NT), lds, st, s1, w1, s2, w2, labels, cw, ignore_index, ce_coef, diff_coef, losses,
                       g1, g2, part, N, C, Hi, Wi, inv_m);
    return 0;
  };
  return two ? go(up8_softmax_ce_l1_kernel<NCMAX, true>) : go(up8_softmax_ce_l1_kernel<NCMAX, false>);
}

template <int NCMAX, bool EXACT>
int launch_up_loss_dma(bool two, int blocks, hipStream_t st, const float* s1, const float* w1, const float* s2, const float* w2,
                       const int64_t* labels, const float* cw, int64_t ignore_index, float ce_coef, float diff_coef, const float* losses,
                       float* g1, float* g2, float* part, int N, int C, int Hi, int Wi, float inv_m) {
  auto go = [&](auto kern, size_t lds) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      mcdseg_set_error("up8_softmax_ce_l1: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return -5;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(UP_NT), lds, st, s1, w1, s2, w2, labels, cw, ignore_index, ce_coef, diff_coef, losses,
                       g1, g2, part, N, C, Hi, Wi, inv_m);
    return 0;
  };
  return two ? go(up8_softmax_ce_l1_dma_kernel<NCMAX, true, EXACT>, UpDmaLayout<NCMAX, true>::BYTES)
             : go(up8_softmax_ce_l1_dma_kernel<NCMAX, false, EXACT>, UpDmaLayout<NCMAX, false>::BYTES);
}

// the instantiation of the DMA kernel for C classes: the benchmark's 41 has its own (no padding classes: 15 % less arithmetic)
int up_loss_dma_ncmax(int C) { return C <= 16 ? 16 : (C <= 24 ? 24 : (C == 41 ? 41 : 48)); }

bool up_loss_dma_ok(int64_t N, int64_t C, int64_t Hi, int64_t Wi, bool labels) {
  const bool on = mcd_opt(MCD_OPT_UP8_LOSS_DMA) != 0;  // 0: the register-staged kernel (A/B, and the parity test's other side)
  // buffer resources address 32 bits (the DMA's offsets are formed in int)
  return on && N * C * Hi * Wi * 4 < (1ll << 31) && (!labels || N * Hi * Wi * 64 * 8 < (1ll << 31));
}

}  // namespace

extern "C" size_t mcdseg_loss_workspace_bytes(int32_t N, int32_t HW) {
  if (N <= 0 || HW <= 0) return 0;
  const int64_t P = (int64_t)N * HW;
  return (size_t)(ceil_div64(P, LOSS_BLOCK) * 3 + wsum_blocks(P)) * sizeof(float);
}

extern "C" size_t mcdseg_label_weight_sum_workspace_bytes(int64_t P) { return P > 0 ? (size_t)wsum_blocks(P) * sizeof(float) : 0; }

extern "C" int mcdseg_label_weight_sum(const int64_t* labels, const float* class_weight, int64_t ignore_index, int32_t C, int64_t P,
                                       float* out, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(labels && out && workspace && C > 0 && P > 0, "label_weight_sum: bad arguments");
  const int wb = wsum_blocks(P);
  MCD_REQUIRE(workspace_bytes >= (size_t)wb * sizeof(float), "label_weight_sum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(label_wsum_kernel, dim3(wb), dim3(256), 0, st, labels, class_weight, ignore_index, C, P, (float*)workspace);
  MCD_LAUNCH_CHECK("label_wsum");
  hipLaunchKernelGGL(wsum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, wb, out);
  MCD_LAUNCH_CHECK("wsum_finalize");
  return 0;
}

extern "C" int mcdseg_softmax_ce_l1(const float* z1, const float* z2, const int64_t* labels, const float* class_weight,
                                    int64_t ignore_index, float ce_coef, float diff_coef, const float* wsum_in, float* g1, float* g2,
                                    float* losses, int32_t N, int32_t C, int32_t HW, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  MCD_REQUIRE(z1 && losses && workspace, "softmax_ce_l1: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && HW > 0, "softmax_ce_l1: bad dims");
  MCD_REQUIRE(C <= 48, "softmax_ce_l1: at most 48 classes are kept in registers (got %d)", C);
  MCD_REQUIRE(z2 != nullptr || (g2 == nullptr && diff_coef == 0.f), "softmax_ce_l1: discrepancy needs z2");
  MCD_REQUIRE(labels != nullptr || ce_coef == 0.f, "softmax_ce_l1: cross-entropy needs labels");
  MCD_REQUIRE(workspace_bytes >= mcdseg_loss_workspace_bytes(N, HW), "softmax_ce_l1: workspace too small");
  const int64_t P = (int64_t)N * HW;
  const int64_t nblk = ceil_div64(P, LOSS_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* wpart = part + nblk * 3;
  const int wb = wsum_blocks(P);
  if (wsum_in != nullptr) {
    // normaliser supplied by the caller (data parallel: the all-reduced sum of w[y] over every rank's shard)
    (void)hipMemcpyAsync(losses + 3, wsum_in, sizeof(float), hipMemcpyDeviceToDevice, st);
  } else if (labels != nullptr) {
    hipLaunchKernelGGL(label_wsum_kernel, dim3(wb), dim3(256), 0, st, labels, class_weight, ignore_index, C, P, wpart);
    MCD_LAUNCH_CHECK("label_wsum");
    hipLaunchKernelGGL(wsum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)wpart, wb, losses + 3);
    MCD_LAUNCH_CHECK("wsum_finalize");
  } else {
    (void)hipMemsetAsync(losses + 3, 0, sizeof(float), st);
  }
  const double inv_m = 1.0 / ((double)P * (double)C);
  dim3 grid((unsigned)nblk);
  const bool two = z2 != nullptr;
  if (C <= 16)
    launch_loss<16>(two, grid, st, z1, z2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2, part, C, HW, P,
                    (float)inv_m);
  else if (C <= 24)
    launch_loss<24>(two, grid, st, z1, z2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2, part, C, HW, P,
                    (float)inv_m);
  else
    launch_loss<48>(two, grid, st, z1, z2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2, part, C, HW, P,
                    (float)inv_m);
  MCD_LAUNCH_CHECK("softmax_ce_l1");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nblk, losses, inv_m,
                     labels != nullptr ? 1 : 0);
  MCD_LAUNCH_CHECK("loss_finalize");
  return 0;
}

// persistent workgroups: one per CU (the kernels of all classes fill most of a CU's LDS), never more than there are items
static int up_loss_blocks(int32_t N, int32_t Hi, int32_t Wi) {
  const int64_t items = (int64_t)N * (Hi + 1) * ceil_div(8 * Wi, UP_COLS);
  return (int)(items < 256 ? items : 256);
}

extern "C" size_t mcdseg_up8_loss_workspace_bytes(int32_t N, int32_t Hi, int32_t Wi) {
  if (N <= 0 || Hi <= 0 || Wi <= 0) return 0;
  return (size_t)(up_loss_blocks(N, Hi, Wi) * 3 + wsum_blocks((int64_t)N * Hi * Wi * 64)) * sizeof(float);
}

extern "C" int mcdseg_up8_softmax_ce_l1(const float* s1, const float* w1, const float* s2, const float* w2, const int64_t* labels,
                                        const float* class_weight, int64_t ignore_index, float ce_coef, float diff_coef,
                                        const float* wsum_in, float* g1, float* g2, float* losses, int32_t N, int32_t C, int32_t Hi,
                                        int32_t Wi, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(s1 && w1 && losses && workspace, "up8_softmax_ce_l1: null pointer");
  MCD_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0, "up8_softmax_ce_l1: bad dims");
  MCD_REQUIRE(C <= 48, "up8_softmax_ce_l1: at most 48 classes are kept in registers (got %d)", C);
  MCD_REQUIRE((s2 == nullptr) == (w2 == nullptr), "up8_softmax_ce_l1: s2 and w2 come together");
  MCD_REQUIRE(s2 != nullptr || (g2 == nullptr && diff_coef == 0.f), "up8_softmax_ce_l1: discrepancy needs the second head");
  MCD_REQUIRE(labels != nullptr || ce_coef == 0.f, "up8_softmax_ce_l1: cross-entropy needs labels");
  MCD_REQUIRE((int64_t)N * (Hi + 1) * ceil_div(8 * Wi, UP_COLS) < (1ll << 31), "up8_softmax_ce_l1: too many patches");
  MCD_REQUIRE(workspace_bytes >= mcdseg_up8_loss_workspace_bytes(N, Hi, Wi), "up8_softmax_ce_l1: workspace too small");
  const int64_t P = (int64_t)N * Hi * Wi * 64;
  const int nblk = up_loss_blocks(N, Hi, Wi);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* wpart = part + (size_t)nblk * 3;
  const int wb = wsum_blocks(P);
  if (wsum_in != nullptr) {
    (void)hipMemcpyAsync(losses + 3, wsum_in, sizeof(float), hipMemcpyDeviceToDevice, st);
  } else if (labels != nullptr) {
    hipLaunchKernelGGL(label_wsum_kernel, dim3(wb), dim3(256), 0, st, labels, class_weight, ignore_index, C, P, wpart);
    MCD_LAUNCH_CHECK("label_wsum");
    hipLaunchKernelGGL(wsum_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)wpart, wb, losses + 3);
    MCD_LAUNCH_CHECK("wsum_finalize");
  } else {
    (void)hipMemsetAsync(losses + 3, 0, sizeof(float), st);
  }
  const double inv_m = 1.0 / ((double)P * (double)C);
  const bool two = s2 != nullptr;
  if (up_loss_dma_ok(N, C, Hi, Wi, labels != nullptr)) {
    int rc;
#define MCD_UP_DMA(NC)                                                                                                              \
  rc = (C == NC) ? launch_up_loss_dma<NC, true>(two, nblk, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef, diff_coef, \
                                                losses, g1, g2, part, N, C, Hi, Wi, (float)inv_m)                                    \
                 : launch_up_loss_dma<NC, false>(two, nblk, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef,         \
                                                 diff_coef, losses, g1, g2, part, N, C, Hi, Wi, (float)inv_m)
    switch (up_loss_dma_ncmax(C)) {
      case 16: MCD_UP_DMA(16); break;
      case 24: MCD_UP_DMA(24); break;
      case 41: rc = launch_up_loss_dma<41, true>(two, nblk, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef, diff_coef,
                                                 losses, g1, g2, part, N, C, Hi, Wi, (float)inv_m); break;
      default: MCD_UP_DMA(48); break;
    }
#undef MCD_UP_DMA
    if (rc != 0) return rc;
    MCD_LAUNCH_CHECK("up8_softmax_ce_l1 (dma)");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, (int64_t)nblk, losses, inv_m,
                       labels != nullptr ? 1 : 0);
    MCD_LAUNCH_CHECK("loss_finalize");
    return 0;
  }
  const int ncmax = C <= 16 ? 16 : (C <= 24 ? 24 : 48);  // the instantiation chosen below
  const size_t lds = (size_t)(two ? 2 : 1) * ncmax * (256 + 4 * UP_JP) * sizeof(float);
  int rc;
  if (C <= 16)
    rc = launch_up_loss<16>(two, nblk, lds, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2,
                            part, N, C, Hi, Wi, (float)inv_m);
  else if (C <= 24)
    rc = launch_up_loss<24>(two, nblk, lds, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2,
                            part, N, C, Hi, Wi, (float)inv_m);
  else
    rc = launch_up_loss<48>(two, nblk, lds, st, s1, w1, s2, w2, labels, class_weight, ignore_index, ce_coef, diff_coef, losses, g1, g2,
                            part, N, C, Hi, Wi, (float)inv_m);
  if (rc != 0) return rc;
  MCD_LAUNCH_CHECK("up8_softmax_ce_l1");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, (int64_t)nblk, losses, inv_m,
                     labels != nullptr ? 1 : 0);
  MCD_LAUNCH_CHECK("loss_finalize");
  return 0;
}

extern "C" size_t mcdseg_predict_workspace_bytes(int32_t N, int32_t HW) {
  return (N > 0 && HW > 0) ? (size_t)ceil_div64((int64_t)N * HW, LOSS_BLOCK) * sizeof(float) : 0;
}

extern "C" int mcdseg_predict_labels(const float* z1, const float* z2, uint8_t* labels, float* entropy, int32_t N, int32_t C,
                                     int32_t C_used, int32_t HW, void* workspace, size_t workspace_bytes, void* stream) {
  MCD_REQUIRE(z1 && labels && entropy && workspace, "predict_labels: null pointer");
  MCD_REQUIRE(N > 0 && HW > 0 && C > 0 && C <= 48 && C_used > 0 && C_used <= C && C_used <= 256, "predict_labels: bad dims");
  MCD_REQUIRE(workspace_bytes >= mcdseg_predict_workspace_bytes(N, HW), "predict_labels: workspace too small");
  const int64_t P = (int64_t)N * HW;
  const int64_t nblk = ceil_div64(P, LOSS_BLOCK);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 16)
    hipLaunchKernelGGL(predict_kernel<16>, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, st, z1, z2, labels, (float*)workspace, C, C_used, HW, P);
  else if (C <= 24)
    hipLaunchKernelGGL(predict_kernel<24>, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, st, z1, z2, labels, (float*)workspace, C, C_used, HW, P);
  else
    hipLaunchKernelGGL(predict_kernel<48>, dim3((unsigned)nblk), dim3(LOSS_BLOCK), 0, st, z1, z2, labels, (float*)workspace, C, C_used, HW, P);
  MCD_LAUNCH_CHECK("predict_labels");
  hipLaunchKernelGGL(predict_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nblk, -1.0 / ((double)P * (double)C), entropy);
  MCD_LAUNCH_CHECK("predict_finalize");
  return 0;
}

extern "C" int mcdseg_scale_by_device_scalar(float* buf, const float* scale, int64_t n, void* stream) {
  MCD_REQUIRE(buf && scale && n >= 0, "scale_by_device_scalar: bad arguments");
  if (n == 0) return 0;
  MCD_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "scale_by_device_scalar: buffer must be 16-byte aligned");
  const int64_t n4 = n / 4;
  int64_t blocks = ceil_div64(n4 > 0 ? n4 : n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, buf, scale, n4, n);
  MCD_LAUNCH_CHECK("scale_by_device_scalar");
  return 0;
}

// Which kernel mcdseg_up8_softmax_ce_l1 launches for a problem (profilers, bench.py's per-kernel tables): the class count of the LDS-DMA
// kernel's instantiation (16, 24, 41, 48), or MINUS that of the register-staged kernel (16, 24, 48) when the option UP8_LOSS_DMA is 0
// or a tensor outgrows a 32-bit buffer resource.
extern "C" int32_t mcdseg_up8_loss_variant(int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t labelled) {
  if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0) return 0;
  if (up_loss_dma_ok(N, C, Hi, Wi, labelled != 0)) return up_loss_dma_ncmax(C);
  return -(C <= 16 ? 16 : (C <= 24 ? 24 : 48));
}
