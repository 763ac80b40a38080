// Probe (development tool, not part of the library): what a device-scope barrier between ALL workgroups of a launch
// costs on gfx950 -- the handshake a "conv + BatchNorm statistics + apply in one kernel" epilogue would need twice per layer (partial
// sums -> barrier -> a few workgroups finalize -> barrier -> every tile normalises its own accumulators).  G workgroups of 512 threads,
// one per CU (G <= 256: all co-resident on an otherwise idle chip), each: write a 1 KB partial row (what a tile's BatchNorm partial sums
// are), release, arrive at a counter, spin (BOUNDED: a workgroup gives up after SPIN_MAX polls and raises a flag -- the probe cannot hang
// the box), acquire, read what the OTHER workgroups wrote (the finalize step reads every row; here every workgroup reads 8 rows), repeat.
// Reports microseconds per barrier round trip, and the same loop without the barrier as the baseline.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/grid_barrier tools/probes/grid_barrier.hip && tools/probes/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int SPIN_MAX = 4000000;

__global__ __launch_bounds__(512) void barrier_loop(float* rows, unsigned* counter, unsigned* gave_up, float* sink, int rounds, int use_barrier) {
  const int g = blockIdx.x, G = gridDim.x, t = threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (t < 256) rows[(size_t)g * 256 + t] = (float)(r + g + t);  // this workgroup's partial row
    if (use_barrier) {
      __syncthreads();
      if (t == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope on a global pointer: the row is visible to the other XCDs' L2s
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(r + 1) * (unsigned)G;
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
          if (++spins > SPIN_MAX) {
            atomicExch(gave_up, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
      if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // somebody gave up: nobody waits any more
    }
    // the "finalize" side: read rows the others wrote (8 rows of 256 floats, spread over the grid)
    if (t < 256) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += __builtin_nontemporal_load(&rows[(size_t)((g + 1 + 31 * k) % G) * 256 + t]);
    }
  }
  if (t < 256) sink[(size_t)g * 256 + t] = acc;
}

int main() {
  float *rows, *sink;
  unsigned *counter, *gave_up;
  hipMalloc(&rows, 256 * 256 * sizeof(float));
  hipMalloc(&sink, 256 * 256 * sizeof(float));
  hipMalloc(&counter, sizeof(unsigned));
  hipMalloc(&gave_up, sizeof(unsigned));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int rounds = 200;
  for (int G : {64, 128, 240, 256}) {
    float ms[2] = {0.f, 0.f};
    unsigned flag = 0;
    for (int use = 0; use < 2; ++use) {
      for (int rep = 0; rep < 3; ++rep) {  // the last repetition is the one reported
        hipMemset(counter, 0, sizeof(unsigned));
        hipMemset(gave_up, 0, sizeof(unsigned));
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(512), 0, 0, rows, counter, gave_up, sink, rounds, use);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[use], e0, e1);
      }
      if (use) hipMemcpy(&flag, gave_up, sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    printf("G = %3d workgroups: %.2f us per round with the barrier, %.2f us without -> %.2f us per grid barrier%s\n", G, 1e3f * ms[1] / rounds,
           1e3f * ms[0] / rounds, 1e3f * (ms[1] - ms[0]) / rounds, flag ? "   (a workgroup GAVE UP: not all were co-resident)" : "");
  }
  return 0;
}
