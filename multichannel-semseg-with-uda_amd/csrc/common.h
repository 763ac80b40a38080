// Shared host/device helpers for libmcdseg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "mcdseg.h"
#include "options.h"

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

// Butterfly steps of the wave reductions without the LDS crossbar: v + (v of lane ^ 1, 2, 4, 8) as DPP row operations -- quad
// permutes for 1 and 2; for 4 and 8 the half-row / row mirror, which pairs a lane with one that already holds the same value as
// lane ^ 4 / lane ^ 8 (after the earlier steps the sums are uniform within groups of 4 / 8 lanes, bit for bit: IEEE addition
// commutes) -- and one ds_swizzle for lane ^ 16.  Same sums in the same order as five __shfl_xor steps, i.e. the same bits; but
// __shfl_xor is a ds_bpermute (an LDS-crossbar round trip of ~100 cycles), and five DEPENDENT ones per sum made the fused
// BatchNorm statistics of a 256 x 256 convolution tile (128 sums per wave) cost 67 000 cycles -- 17 % of a 256-channel layer's kernel.
template <int CTRL>
__device__ __forceinline__ float mcd_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ float wave_half_sum(float v) {
  // sum over the 32 lanes that share (lane >> 5); every lane of the half ends with the total
  v += mcd_dpp<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += mcd_dpp<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += mcd_dpp<0x141>(v);  // row_half_mirror: lane -> 7 - lane (the other quad of the 8)
  v += mcd_dpp<0x140>(v);  // row_mirror: lane -> 15 - lane (the other 8 of the 16)
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // bit mode, xor 0x10: lane ^ 16
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

typedef int mcd_i32x4 __attribute__((ext_vector_type(4)));

// A raw buffer resource over [base, base + bytes) (what __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000) builds), as
// the four dwords an inline-asm operand can take.
__device__ __forceinline__ mcd_i32x4 mcd_raw_rsrc(const void* base, int bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  mcd_i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000;
  return r;
}

// LDS-DMA the compiler does not see.  Issued through the builtin, every LDS read that follows -- in every basic block -- gets a
// compiler-made `s_waitcnt vmcnt(0)` in front (it cannot tell the buffer being filled from the one being read), which drains
// the very prefetch (and, the counter being one for loads and stores, every store in flight); the caller places the waits.
// lds = the wave's destination (byte address in LDS, wave-uniform); lane l writes SIZE bytes at lds + l * SIZE.
// M0 is written without the compiler's knowledge (it is a reserved register: naming it as a clobber is refused as undefined behaviour),
// so a kernel that uses this must not ALSO use the builtin LDS-DMA, whose M0 set-up the compiler may hoist out of a loop.
template <int SIZE>
__device__ __forceinline__ void mcd_hidden_dma(mcd_i32x4 rs, unsigned lds, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (SIZE == 4)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rs) : "memory");
#else
  (void)rs; (void)lds; (void)voff;
#endif
}
