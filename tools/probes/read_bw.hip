// Probe (development tool, not part of the library): what a kernel that ONLY READS reaches on this chip, next to one
// that reads and writes -- bn_bwd_reduce (two fp32 tensors in, a few KB out) runs at 4.2 TB/s of its bytes while its siblings that also
// write (bn_bwd_apply: two tensors in, one or two out) run at 5.7-5.9.  Three kernels over the same bytes: sum one buffer; sum two
// buffers (the reduce's shape); copy (half the bytes read, half written).  f4 per lane, 4 loads in flight per thread, grid-stride
// over 2048 workgroups of 256 threads (the reduce kernel's launch).  Sizes: 2 x 79 MB (a 256-channel layer of BASELINE config 2: what one
// bn_bwd_reduce launch reads), 2 x 315 MB, 2 x 1.26 GB.  "cold": a 1 GB buffer is streamed between repetitions (nothing of the
// operands left in the 256 MB Infinity Cache); "warm": back to back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/read_bw tools/probes/read_bw.hip && tools/probes/read_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sum1(const f4* __restrict__ a, size_t n4, float* out) {
  float s = 0.f;
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < n4; i += 4 * step) {
    const f4 q0 = __builtin_nontemporal_load(a + i), q1 = __builtin_nontemporal_load(a + i + step);
    const f4 q2 = __builtin_nontemporal_load(a + i + 2 * step), q3 = __builtin_nontemporal_load(a + i + 3 * step);
    s += (q0.x + q0.y + q0.z + q0.w) + (q1.x + q1.y + q1.z + q1.w) + (q2.x + q2.y + q2.z + q2.w) + (q3.x + q3.y + q3.z + q3.w);
  }
  for (; i < n4; i += step) {
    const f4 q = a[i];
    s += q.x + q.y + q.z + q.w;
  }
  if (s == 12345.678f) out[0] = s;  // (never: keeps the loads alive)
}

__global__ __launch_bounds__(256) void sum2(const f4* __restrict__ a, const f4* __restrict__ b, size_t n4, float* out) {
  float s = 0.f, d = 0.f;
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + step < n4; i += 2 * step) {
    const f4 q0 = __builtin_nontemporal_load(a + i), q1 = __builtin_nontemporal_load(a + i + step);
    const f4 r0 = __builtin_nontemporal_load(b + i), r1 = __builtin_nontemporal_load(b + i + step);
    s += (q0.x + q0.y + q0.z + q0.w) + (q1.x + q1.y + q1.z + q1.w);
    d += q0.x * r0.x + q0.y * r0.y + q0.z * r0.z + q0.w * r0.w + q1.x * r1.x + q1.y * r1.y + q1.z * r1.z + q1.w * r1.w;
  }
  for (; i < n4; i += step) {
    const f4 q = a[i], r = b[i];
    s += q.x + q.y + q.z + q.w;
    d += q.x * r.x + q.y * r.y + q.z * r.z + q.w * r.w;
  }
  if (s == 12345.678f && d == 1.f) out[0] = s;
}

__global__ __launch_bounds__(256) void copy1(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < n4; i += 4 * step) {
    const f4 q0 = __builtin_nontemporal_load(a + i), q1 = __builtin_nontemporal_load(a + i + step);
    const f4 q2 = __builtin_nontemporal_load(a + i + 2 * step), q3 = __builtin_nontemporal_load(a + i + 3 * step);
    __builtin_nontemporal_store(q0, b + i);
    __builtin_nontemporal_store(q1, b + i + step);
    __builtin_nontemporal_store(q2, b + i + 2 * step);
    __builtin_nontemporal_store(q3, b + i + 3 * step);
  }
  for (; i < n4; i += step) b[i] = a[i];
}

int main() {
  const size_t MAXB = (size_t)1260 << 20;
  f4 *a, *b, *flush;
  float* out;
  hipMalloc(&a, MAXB);
  hipMalloc(&b, MAXB);
  hipMalloc(&flush, (size_t)1 << 30);
  hipMalloc(&out, 256);
  hipMemset(a, 0, MAXB);
  hipMemset(b, 0, MAXB);
  hipMemset(flush, 0, (size_t)1 << 30);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t sizes[3] = {(size_t)79 << 20, (size_t)315 << 20, (size_t)1260 << 20};
  for (int grid : {2048, 8192}) {
    for (size_t bytes : sizes) {
      const size_t n4 = bytes / 16;
      for (int cold = 0; cold < 2; ++cold) {
        float t[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k) {
          const int reps = 12;
          float total = 0.f;
          for (int r = 0; r < reps + 2; ++r) {
            if (cold) hipLaunchKernelGGL(sum1, dim3(2048), dim3(256), 0, 0, flush, ((size_t)1 << 30) / 16, out);
            hipEventRecord(e0, 0);
            if (k == 0) hipLaunchKernelGGL(sum1, dim3(grid), dim3(256), 0, 0, a, 2 * n4 <= MAXB / 16 ? 2 * n4 : n4, out);  // the same bytes as sum2 where they fit
            if (k == 1) hipLaunchKernelGGL(sum2, dim3(grid), dim3(256), 0, 0, a, b, n4, out);
            if (k == 2) hipLaunchKernelGGL(copy1, dim3(grid), dim3(256), 0, 0, a, b, n4);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) total += ms;
          }
          t[k] = total / reps;
        }
        const double gb = 2.0 * bytes / 1e9;
        const double gb0 = (2 * n4 <= MAXB / 16 ? 2.0 : 1.0) * bytes / 1e9;
        printf("grid %4d  2 x %4zu MB  %s:  sum of one buffer %.3f ms = %.2f TB/s   sum of two %.3f ms = %.2f TB/s   copy %.3f ms = %.2f TB/s\n", grid,
               bytes >> 20, cold ? "cold" : "warm", t[0], gb0 / t[0], t[1], gb / t[1], t[2], gb / t[2]);
      }
    }
  }
  return 0;
}
