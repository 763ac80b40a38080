// Shared host/device helpers for libmcdseg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "mcdseg.h"

void mcdseg_set_error(const char* fmt, ...);

#define MCD_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      mcdseg_set_error(__VA_ARGS__);          \
      return -22; /* -EINVAL */               \
    }                                         \
  } while (0)

#define MCD_LAUNCH_CHECK(name)                                                     \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      mcdseg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return -5; /* -EIO */                                                        \
    }                                                                              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---- GEMM padding rules shared by pack / fprop / dgrad / wgrad -------------------------------
// M tile: 32, 64 or 128 output channels per workgroup.
static inline int mcd_bm(int M) { return M <= 32 ? 32 : (M <= 64 ? 64 : 128); }
static inline int mcd_mp(int M) { return round_up(M, mcd_bm(M)); }
// K (contraction channels per tap) is padded to the K-step: 8 for thin inputs, else 16.
static inline int mcd_bk(int K) { return K <= 8 ? 8 : 16; }
static inline int mcd_kp(int K) { return round_up(K, mcd_bk(K)); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_half_sum(float v) {
  // sum over the 32 lanes that share (lane >> 5); every lane of the half ends with the total
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v = wave_half_sum(v);
  v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
