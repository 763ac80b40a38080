// Operand-split policies of the matrix-pipe convolutions (conv_gemm_split.hip, conv_wgrad_split.hip, bn.hip).
//
// gfx950 has no fp32-grade matrix instruction faster than the vector rate (no xf32/TF32), so an fp32 product a*b is
// carried through the 16-bit MFMA pipe as a sum of exact piece products:
//
//   SplitBf16x6   a = a1 + a2 + a3 with bf16 pieces (8+8+8 significant bits, fp32's exponent range: no scaling needed),
//                 six cross terms a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1 on v_mfma_f32_32x32x16_bf16;
//   SplitF16x3    a = s * (h1 + h2) with fp16 pieces (11+11 significant bits) and a power-of-two scale s per TENSOR,
//                 three cross terms h1h1' + h1h2' + h2h1' on v_mfma_f32_32x32x16_f16 -- half the matrix instructions
//                 and two thirds of the operand bytes of the bf16 form for the same measured accuracy against the
//                 reference's fp64 gradients (tools/split_numerics.py; tests/test_model_gpu.py).
//
// fp16 has 5 exponent bits, so SplitF16x3 needs the scale: s = 2^(e-15) where 2^e >= B and B is an UPPER BOUND of |x| over
// the tensor, known before the tensor is written (BatchNorm outputs: Samuelson's inequality |xhat| <= sqrt(n-1), see bn.hip;
// weights and foreign tensors: a measured absolute maximum).  Then |x/s| <= 2^15 < 65504: no finite value overflows.  An
// element keeps 22 significant bits while |x/s| >= 2^-3 and an ABSOLUTE error of 2^-25 s below that (fp16 subnormals are
// kept by v_cvt_f16_f32 and by the MFMA: tools/probes/f16_denorm.hip) -- for a dot product it is the absolute error relative
// to the TYPICAL magnitude that counts, and typical values sit within 2^13 of the bound for every tensor of the network.
// Non-finite values stay non-finite: inf becomes NaN in the remainder (inf - inf), as in the bf16 split.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// scale of a tensor whose absolute values are bounded by `bound`: 2^(e-15) with 2^e >= bound; 1 for an empty / all-zero /
// non-finite bound (non-finite data then overflows into inf/NaN, which is the contract)
__host__ __device__ __forceinline__ float mcd_scale_of_bound(float bound) {
  if (!(bound > 0.f) || !(bound <= 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(bound, &e);  // bound = m * 2^e, 0.5 <= m < 1
  e -= 15;
  if (e < -100) e = -100;
  if (e > 100) e = 100;
  return ldexpf(1.f, e);
}

struct SplitBf16x6 {
  static constexpr bool HALF_OUT = false;  // (16-bit channel-blocked epilogues: the one-term arithmetic only)
  static constexpr int KDEEP = 1;          // (K-steps of 16 channels per barrier interval of the ping-pong kernels)
  static constexpr int NP = 3;
  static constexpr int NPU = 3;  // pieces a kernel stages into LDS (all of them)
  static constexpr int NTERMS = 6;
  static constexpr bool SCALED = false;
  typedef __bf16 elem;
  typedef bf16x8 frag;
  // cross terms (piece of A, piece of B), smallest first
  static constexpr int TA[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int TB[6] = {0, 2, 1, 0, 1, 0};
  __device__ __forceinline__ static f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  __device__ __forceinline__ static void split(float v, float /*inv_scale*/, elem (&p)[NP]) {
    const __bf16 a = (__bf16)v;
    const float r1 = v - (float)a;  // exact
    const __bf16 b = (__bf16)r1;
    p[0] = a;
    p[1] = b;
    p[2] = (__bf16)(r1 - (float)b);
  }
};

struct SplitF16x3 {
  static constexpr bool HALF_OUT = false;
  static constexpr int KDEEP = 1;
  static constexpr int NP = 2;
  static constexpr int NPU = 2;
  static constexpr int NTERMS = 3;
  static constexpr bool SCALED = true;
  typedef _Float16 elem;
  typedef f16x8 frag;
  static constexpr int TA[3] = {1, 0, 0};
  static constexpr int TB[3] = {0, 1, 0};
  __device__ __forceinline__ static f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  __device__ __forceinline__ static void split(float v, float inv_scale, elem (&p)[NP]) {
    const float q = v * inv_scale;  // exact: the scale is a power of two
    const _Float16 h = (_Float16)q;
    p[0] = h;
    p[1] = (_Float16)(q - (float)h);
  }
};

// SplitF16x1: the REDUCED-PRECISION arithmetic (BASELINE config 5 "bf16"; bench.py --dtype f16): the operands are stored exactly
// as SplitF16x3 stores them (two scaled fp16 pieces -- same companions, weight images, bounds, producers), but a product keeps
// only the leading term h1 h1': one MFMA instead of three, operands rounded to fp16's 11 significant bits (bf16 would keep 8).
struct SplitF16x1 : SplitF16x3 {
  static constexpr bool HALF_OUT = true;  // its convolutions can write 16-bit channel-blocked tensors (round 6: 2-byte activation storage)
  static constexpr int NPU = 1;  // pieces a kernel has to STAGE (the second piece is stored but never multiplied)
  static constexpr int NTERMS = 1;
  static constexpr int TA[1] = {0};
  static constexpr int TB[1] = {0};
};

// SplitF16x1D: the same arithmetic for the 8-wave ping-pong convolution with TWO K-steps of 16 channels per barrier interval.  A K-step
// of the one-term arithmetic is 8 matrix instructions per wave (256 cycles) against the three-term form's 24, while its barriers, counted
// waits and DMA issue cost what they cost: the kernels balanced for three terms reach 0.30 of the matrix peak with one (round 5).  Here
// the LDS slots of the (never staged) second piece hold the NEXT K-step's operands: the stage image, the fragment reads and the DMA count
// are the three-term kernel's, 16 matrix instructions per interval.  Same products in the same order: bit for bit SplitF16x1's results.
struct SplitF16x1D : SplitF16x1 {
  static constexpr int KDEEP = 2;
};

// split 8 values into NP fragments
template <class P>
__device__ __forceinline__ void split_frag(const float (&v)[8], float inv_scale, typename P::frag (&out)[P::NP]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    typename P::elem q[P::NP];
    P::split(v[e], inv_scale, q);
#pragma unroll
    for (int pc = 0; pc < P::NP; ++pc) out[pc][e] = q[pc];
  }
}

// scale of an operand from its bound pointer (NULL for the unscaled policy)
template <class P>
__device__ __forceinline__ float operand_scale(const float* bound) {
  if constexpr (P::SCALED) return mcd_scale_of_bound(*bound);
  return 1.f;
}

static inline int mcd_math_pieces(int math) { return (math == 3 || math == 1) ? 2 : 3; }  // MCDSEG_MATH_F16X3 / _F16X1 : MCDSEG_MATH_BF16X6
// the arithmetic whose STORAGE (companions, weight images, bounds) a math shares: F16X1 multiplies F16X3's operands
static inline int mcd_storage_math(int math) { return math == 1 ? 3 : math; }
static inline bool mcd_math_known(int math) { return math == 1 || math == 3 || math == 6; }
