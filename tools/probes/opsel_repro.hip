// Stand-alone reproducer attempt for DESIGN.md section 5a (development tool, not part of the library; no torch, no libmcdseg).
//
// Claim under test: on gfx950 a packed-fp32 VALU instruction whose OP_SEL routes the HIGH dword of a 64-bit VGPR source into the
// LOW result lane -- `v_pk_add_f32 vdst, src0, src1 op_sel:[0,1]` -- now and then reads 0.0 instead of that dword for the lanes of
// one 16-lane pass WHEN A SECOND PROCESS COMPUTES ON THE SAME CUs.  In round 3 the compiler had formed exactly two such
// instructions in bn_bwd_apply_cb_v4_kernel (k1 = dbeta[c] / n allocated as the high half of a register pair); the library is now
// compiled without packed-fp32 formation in those files and a CPU test forbids the pattern in the disassembly.
//
// This probe isolates the instruction: every lane streams (g, xhat) pairs from memory like the BatchNorm backward does, forms
//   r = g - k          with k = (k_lo, k_hi) in one register pair and BOTH result lanes taking k_hi:  v_pk_add_f32 ... op_sel:[0,1]
// through inline assembly (so the exact encoding is under test, whatever the compiler would choose), and compares r bit for bit with
// values prepared on the host.  MODE 1 is the control: the same arithmetic with the low dword broadcast (op_sel_hi:[0,1] on the
// OTHER half, the form all other packed instructions of the library use).  Run it alone and as two concurrent processes
// (tools/probes/opsel_repro.sh); it prints how many lane results were checked and how many differed, and the first few offenders.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/opsel_repro tools/probes/opsel_repro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

struct Bad {
  unsigned iter, gid, lane_in_wave, which;
  float got, want;
};

template <int MODE>
__global__ __launch_bounds__(256) void probe(const v2f* __restrict__ g, const v2f* __restrict__ want, v2f* __restrict__ out, float k_lo, float k_hi,
                                             int n, int iters, unsigned long long* checked, unsigned long long* bad, Bad* log, int log_cap) {
  unsigned long long my_bad = 0, my_checked = 0;
  // k lives in ONE 64-bit register pair: (k_lo, k_hi); the instruction under test must take k_hi for BOTH result lanes
  v2f k = {-k_lo, -k_hi};
  asm volatile("" : "+v"(k));  // opaque: keep it a VGPR pair
  for (int it = 0; it < iters; ++it) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const v2f a = g[i];
      v2f r;
      if (MODE == 0)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(k));            // r.lo = a.lo + k.HI ; r.hi = a.hi + k.hi
      else
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(k));         // control: both lanes take k.LO
      const v2f w = want[i];
      const v2f ref = MODE == 0 ? w : v2f{a.x - k_lo, a.y - k_lo};
      my_checked += 2;
      if (__float_as_uint(r.x) != __float_as_uint(ref.x) || __float_as_uint(r.y) != __float_as_uint(ref.y)) {
        const unsigned long long slot = atomicAdd(bad, 1ull);
        ++my_bad;
        if (slot < (unsigned long long)log_cap)
          log[slot] = Bad{(unsigned)it, (unsigned)i, threadIdx.x & 63u, __float_as_uint(r.x) != __float_as_uint(ref.x) ? 0u : 1u,
                          __float_as_uint(r.x) != __float_as_uint(ref.x) ? r.x : r.y, __float_as_uint(r.x) != __float_as_uint(ref.x) ? ref.x : ref.y};
      }
      out[i] = r;
    }
  }
  (void)my_bad;
  atomicAdd(checked, my_checked);
}

int main(int argc, char** argv) {
  const int seconds = argc > 1 ? atoi(argv[1]) : 20;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const char* tag = argc > 3 ? argv[3] : "solo";
  const int iters = argc > 4 ? atoi(argv[4]) : 4;  // passes over the data per launch: 4 = 0.15 ms launches, 200 = 7 ms (longer than a time slice)
  const int n = 1 << 22;  // 32 MB of (g0, g1) pairs: streamed from memory every iteration
  const float k_lo = 3.0f, k_hi = 0.4375f;
  std::vector<v2f> hg(n), hw(n);
  srand(11);
  for (int i = 0; i < n; ++i) {
    hg[i] = v2f{(float)(rand() % 4096) / 64.f - 32.f, (float)(rand() % 4096) / 64.f - 32.f};  // exact in fp32, so are the sums
    hw[i] = v2f{hg[i].x - k_hi, hg[i].y - k_hi};
  }
  v2f *g, *want, *out;
  unsigned long long *checked, *bad;
  Bad* log;
  const int log_cap = 64;
  (void)hipMalloc(&g, n * sizeof(v2f));
  (void)hipMalloc(&want, n * sizeof(v2f));
  (void)hipMalloc(&out, n * sizeof(v2f));
  (void)hipMalloc(&checked, 8);
  (void)hipMalloc(&bad, 8);
  (void)hipMalloc(&log, log_cap * sizeof(Bad));
  (void)hipMemcpy(g, hg.data(), n * sizeof(v2f), hipMemcpyHostToDevice);
  (void)hipMemcpy(want, hw.data(), n * sizeof(v2f), hipMemcpyHostToDevice);
  (void)hipMemset(checked, 0, 8);
  (void)hipMemset(bad, 0, 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  double elapsed = 0;
  int launches = 0;
  while (elapsed < seconds) {
    (void)hipEventRecord(e0);
    for (int l = 0; l < (iters > 16 ? 2 : 20); ++l) {
      if (mode == 0)
        hipLaunchKernelGGL(probe<0>, dim3(2048), dim3(256), 0, 0, g, want, out, k_lo, k_hi, n, iters, checked, bad, log, log_cap);
      else
        hipLaunchKernelGGL(probe<1>, dim3(2048), dim3(256), 0, 0, g, want, out, k_lo, k_hi, n, iters, checked, bad, log, log_cap);
      ++launches;
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    elapsed += ms * 1e-3;
  }
  unsigned long long hc = 0, hb = 0;
  std::vector<Bad> hl(log_cap);
  (void)hipMemcpy(&hc, checked, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hl.data(), log, log_cap * sizeof(Bad), hipMemcpyDeviceToHost);
  printf("[%s] %s: %d launches, %.1f s, %llu lane results checked, expected 0 different, observed %llu\n", tag,
         mode == 0 ? "v_pk_add_f32 op_sel:[0,1] (high dword -> low lane)" : "control: v_pk_add_f32 op_sel_hi:[1,0] (low dword broadcast)", launches, elapsed,
         hc, hb);
  for (unsigned long long i = 0; i < hb && i < 8; ++i)
    printf("[%s]   iter %u element %u lane %u half %u: got %.6f want %.6f (got - want = %.6f; k_hi = %.4f)\n", tag, hl[i].iter, hl[i].gid, hl[i].lane_in_wave,
           hl[i].which, hl[i].got, hl[i].want, hl[i].got - hl[i].want, k_hi);
  return 0;
}
