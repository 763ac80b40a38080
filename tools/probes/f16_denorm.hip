// Probe (development tool, not part of the library): do v_cvt_f16_f32 and v_mfma_f32_32x32x16_f16 keep fp16 subnormals?
// The f16x3 operand split relies on them for its absolute-error floor (2^-25 of the scale).
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/f16_denorm tools/probes/f16_denorm.hip && tools/probes/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(float* out, float tiny, float one) {
  const _Float16 t = (_Float16)tiny;  // 2^-20: subnormal in fp16 (min normal 2^-14)
  const _Float16 o = (_Float16)one;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = t;
    b[e] = o;
  }
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  f16v acc2;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc2, 0, 0, 0);  // subnormal x subnormal = 2^-40 each, x16 = 2^-36
  if (threadIdx.x == 0) {
    out[0] = (float)t;       // conversion round trip
    out[1] = acc[0];         // expect 16 * 2^-20 = 2^-16
    out[2] = acc2[0];        // expect 2^-36
    const _Float16 r = (_Float16)(tiny * 1.5f) - t;  // fp16 subtraction in the subnormal range
    out[3] = (float)r;       // expect 2^-21
  }
}

int main() {
  float* d;
  hipMalloc(&d, 16);
  probe<<<1, 64>>>(d, 9.5367431640625e-07f, 1.0f);
  float h[4];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("cvt(2^-20) -> %g (want 9.53674e-07)\nmfma tiny*1 sum16 -> %g (want 1.52588e-05)\nmfma tiny*tiny sum16 -> %g (want 1.45519e-11)\nf16 sub -> %g (want 4.76837e-07)\n",
         h[0], h[1], h[2], h[3]);
  const bool ok = h[0] == 9.5367431640625e-07f && h[1] == 1.52587890625e-05f && h[2] > 1.4e-11f && h[2] < 1.5e-11f;
  printf("%s\n", ok ? "F16_DENORMALS_KEPT" : "F16_DENORMALS_FLUSHED");
  return ok ? 0 : 1;
}
