// Probe (development tool, not part of the library): what does the K loop of the split convolution gain from an 8-wave
// PING-PONG workgroup (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md 8-phase template)?
//
//   A  the shipped structure (conv_gemm_split_kernel<SplitF16x3,4,2,2,2,false,true>): 4 waves, 256 x 128 tile, three LDS
//      stages, two independent workgroups per CU; per K-step (16 channels of one tap) a wave reads 12 fragments, issues 6
//      LDS-DMAs, multiplies (24 MFMA), waits with a counted vmcnt and meets its workgroup's barrier.
//   B  8 waves, 256 x 256 tile, one workgroup per CU; waves 0-3 and 4-7 (SIMD partners) run half a K-step apart: while one
//      group multiplies (24 MFMA between two barriers) the other reads its fragments and issues its 4 LDS-DMAs.
//
// Both move the real operand streams: the weight image slab by slab (L2-resident), the pixel operand as per-lane gathers of
// 16-byte units from a companion-shaped buffer with the 3 x 3 dilation-4 tap shifts and zero padding.  Random fp16 data.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/pingpong tools/probes/pingpong.hip && tools/probes/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// problem: N images of C channels, H x W pixels, M = 512 output channels, 3 x 3 taps, dilation D
constexpr int N = 16, C = 512, H = 60, W = 80, HW = H * W, M = 512, D = 4, TAPS = 9;
constexpr int KSTEPS = TAPS * (C / 16);
constexpr int NPIX = N * HW;

struct Args {
  const void* wimg;   // [kstep][plane 4 = piece*2+half][M][16 B]
  const void* cb;     // [piece 2][N][C/8][HW][16 B]
  float* out;         // [tiles][threads] checksum
  int wimg_bytes, cb_piece_bytes;
  long long cb_piece_stride;
  int ksteps;
};

#define DMA16(rs, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsptr), 16, voff, soff, 0, 0)

// ------------------------------------------------------------------------------------------------ structure A
__global__ __launch_bounds__(256, 2) void loop_a(Args p) {
  constexpr int BM = 256, BN = 128, NT = 256, NS = 3;
  constexpr int A_BYTES = 4 * BM * 16, B_BYTES = 4 * BN * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NS * A_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  const int m_tiles = M / BM, n_tiles = NPIX / BN;
  const int per_xcd = (n_tiles + 7) >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_m = slot % m_tiles, tile_n = xcd * per_xcd + slot / m_tiles;
  if (tile_n >= n_tiles) return;
  const int bj = t % BN, bh = __builtin_amdgcn_readfirstlane(t / BN);
  const int pix = tile_n * BN + bj, pn = pix / HW, rem = pix - pn * HW, py = rem / W, px = rem - py * W;
  unsigned valid = 0;
  for (int q = 0; q < TAPS; ++q) {
    const int sy = py + (q / 3 - 1) * D, sx = px + (q % 3 - 1) * D;
    valid |= (sy >= 0 && sy < H && sx >= 0 && sx < W ? 1u : 0u) << q;
  }
  const unsigned vbase = (unsigned)pn * (C / 8) * HW + py * W + px;
  __amdgpu_buffer_rsrc_t cb_rs[2];
  for (int pc = 0; pc < 2; ++pc)
    cb_rs[pc] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.cb + pc * p.cb_piece_stride), 0, p.cb_piece_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wimg, 0, p.wimg_bytes, 0x00020000);
  unsigned a_voff[4];
  for (int i = 0; i < 4; ++i) {
    const int id = t + i * NT, plane = id / BM, m = id - plane * BM;
    a_voff[i] = ((unsigned)plane * M + m) * 16u;
  }
  int l_tap = 0, l_c0 = 0, l_kstep = 0;
  auto issue = [&](int buf) {
    const int a_soff = (l_kstep * 4 * M + tile_m * BM) * 16;
    unsigned char* adst = As + buf * A_BYTES + wave * 64 * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) DMA16(w_rs, adst + i * NT * 16, a_voff[i], a_soff);
    const int rel = ((l_tap / 3) - 1) * D * W + ((l_tap % 3) - 1) * D;
    const unsigned voff = ((valid >> l_tap) & 1u) ? (vbase + (unsigned)rel) * 16u : 0x80000000u;
    const int wave_px = __builtin_amdgcn_readfirstlane(bj - lane);
    const int soff = ((l_c0 >> 3) + bh) * HW * 16;
    unsigned char* bdst = Bs + buf * B_BYTES + (bh * BN + wave_px) * 16;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) DMA16(cb_rs[pc], bdst + pc * 2 * BN * 16, voff, soff);
  };
  auto advance = [&]() {
    ++l_kstep;
    if (++l_tap == TAPS) {
      l_tap = 0;
      l_c0 += 16;
    }
  };
  f16v acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nsteps = p.ksteps;
  issue(0);
  advance();
  issue(1);
  asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  h8 fa[2][4], fb[2][2];
  int cur = 0, nxt2 = 2;
  for (int s = 0; s < nsteps; ++s) {
    const bool more2 = s + 2 < nsteps;
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * 128 + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * 64 + l31) * 16;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[pc][i] = *reinterpret_cast<const h8*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[pc][j] = *reinterpret_cast<const h8*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      advance();
      issue(nxt2);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm == 0 ? 1 : 0][i], fb[tm == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
    if (more2)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
    nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
  }
  float total = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) total += acc[i][j][r];
  p.out[(size_t)blockIdx.x * 512 + t] = total;
}

// ------------------------------------------------------------------------------------------------ structure B
// MODE bit 0: stagger the two wave groups by half a K-step (ping-pong); bit 1: s_setprio(1) around the matrix phase
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop_b(Args p) {
  constexpr int BM = 256, BN = 256, NT = 512, NS = 3;
  constexpr bool STAGGER = MODE & 1, PRIO = MODE & 2;
  constexpr int A_BYTES = 4 * BM * 16, B_BYTES = 4 * BN * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NS * A_BYTES;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 2, wn = wave & 3, l31 = lane & 31, lh = lane >> 5;
  const int m_tiles = M / BM, n_tiles = NPIX / BN;
  const int per_xcd = (n_tiles + 7) >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_m = slot % m_tiles, tile_n = xcd * per_xcd + slot / m_tiles;
  if (tile_n >= n_tiles) return;
  const int bj = t % BN, bh = __builtin_amdgcn_readfirstlane(t / BN);
  const int pix = tile_n * BN + bj, pn = pix / HW, rem = pix - pn * HW, py = rem / W, px = rem - py * W;
  unsigned valid = 0;
  for (int q = 0; q < TAPS; ++q) {
    const int sy = py + (q / 3 - 1) * D, sx = px + (q % 3 - 1) * D;
    valid |= (sy >= 0 && sy < H && sx >= 0 && sx < W ? 1u : 0u) << q;
  }
  const unsigned vbase = (unsigned)pn * (C / 8) * HW + py * W + px;
  __amdgpu_buffer_rsrc_t cb_rs[2];
  for (int pc = 0; pc < 2; ++pc)
    cb_rs[pc] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.cb + pc * p.cb_piece_stride), 0, p.cb_piece_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wimg, 0, p.wimg_bytes, 0x00020000);
  unsigned a_voff[2];
  for (int i = 0; i < 2; ++i) {
    const int id = t + i * NT, plane = id / BM, m = id - plane * BM;
    a_voff[i] = ((unsigned)plane * M + m) * 16u;
  }
  int l_tap = 0, l_c0 = 0, l_kstep = 0;
  auto issue = [&](int buf) {
    const int a_soff = (l_kstep * 4 * M + tile_m * BM) * 16;
    unsigned char* adst = As + buf * A_BYTES + wave * 64 * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) DMA16(w_rs, adst + i * NT * 16, a_voff[i], a_soff);
    const int rel = ((l_tap / 3) - 1) * D * W + ((l_tap % 3) - 1) * D;
    const unsigned voff = ((valid >> l_tap) & 1u) ? (vbase + (unsigned)rel) * 16u : 0x80000000u;
    const int wave_px = __builtin_amdgcn_readfirstlane(bj - lane);
    const int soff = ((l_c0 >> 3) + bh) * HW * 16;
    unsigned char* bdst = Bs + buf * B_BYTES + (bh * BN + wave_px) * 16;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) DMA16(cb_rs[pc], bdst + pc * 2 * BN * 16, voff, soff);
  };
  auto advance = [&]() {
    ++l_kstep;
    if (++l_tap == TAPS) {
      l_tap = 0;
      l_c0 += 16;
    }
  };
  f16v acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nsteps = p.ksteps;
  issue(0);
  advance();
  issue(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();
  h8 fa[2][4], fb[2][2];
  int cur = 0, nxt2 = 2;
  for (int s = 0; s < nsteps; ++s) {
    const bool more2 = s + 2 < nsteps;
    // ---- read phase: this step's fragments, the wave's share of step s+2's operands; step s+1's share must have landed
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * 128 + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * 64 + l31) * 16;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[pc][i] = *reinterpret_cast<const h8*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[pc][j] = *reinterpret_cast<const h8*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more2) {
      advance();
      issue(nxt2);
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix phase
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 3; ++tm)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tm == 0 ? 1 : 0][i], fb[tm == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
    nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();
  float total = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) total += acc[i][j][r];
  p.out[(size_t)blockIdx.x * 512 + t] = total;
}

template <class F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const size_t w_elems = (size_t)KSTEPS * 4 * M * 8, cb_piece_elems = (size_t)N * (C / 8) * HW * 8;
  std::vector<_Float16> h(1 << 22);
  srand(7);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f);
  _Float16 *w, *cb;
  float* out;
  (void)hipMalloc(&w, w_elems * 2);
  (void)hipMalloc(&cb, 2 * cb_piece_elems * 2);
  (void)hipMalloc(&out, (size_t)4096 * 512 * 4);
  for (size_t o = 0; o < w_elems; o += h.size()) (void)hipMemcpy(w + o, h.data(), std::min(h.size(), w_elems - o) * 2, hipMemcpyHostToDevice);
  const size_t chunk = h.size() - 977;  // shifted copies: no two chunks of the operand are identical
  for (size_t o = 0, k = 0; o < 2 * cb_piece_elems; o += chunk, ++k)
    (void)hipMemcpy(cb + o, h.data() + k % 977, std::min(chunk, 2 * cb_piece_elems - o) * 2, hipMemcpyHostToDevice);
  Args a{w, cb, out, (int)(w_elems * 2), (int)(cb_piece_elems * 2), (long long)(cb_piece_elems * 2), KSTEPS};
  const int grid_a = 8 * (((NPIX / 128 + 7) / 8) * (M / 256)), grid_b = 8 * (((NPIX / 256 + 7) / 8) * (M / 256));
  const double flop = 2.0 * NPIX * (double)M * C * TAPS;  // algorithmic; x3 executed
  auto la = [&]() { hipLaunchKernelGGL(loop_a, dim3(grid_a), dim3(256), 0, 0, a); };
  auto lb0 = [&]() { hipLaunchKernelGGL(loop_b<0>, dim3(grid_b), dim3(512), 0, 0, a); };
  auto lb1 = [&]() { hipLaunchKernelGGL(loop_b<1>, dim3(grid_b), dim3(512), 0, 0, a); };
  auto lb3 = [&]() { hipLaunchKernelGGL(loop_b<3>, dim3(grid_b), dim3(512), 0, 0, a); };
  printf("grids: A %d x 256 threads, B %d x 512 threads; %d K-steps; %.1f GFLOP algorithmic\n", grid_a, grid_b, KSTEPS, flop * 1e-9);
  time_ms(la, 600);  // warm the clocks
  time_ms(lb1, 600);
  for (int round = 0; round < 5; ++round) {
    const double ta = time_ms(la, 60), tb0 = time_ms(lb0, 60), tb1 = time_ms(lb1, 60), tb3 = time_ms(lb3, 60);
    auto fr = [&](double ms) { return 3.0 * flop / (ms * 1e-3) / 2.5e15; };
    printf("round %d: A %.4f ms (%.3f)   B same-phase %.4f (%.3f)   B ping-pong %.4f (%.3f)   B ping-pong+prio %.4f (%.3f)\n", round, ta,
           fr(ta), tb0, fr(tb0), tb1, fr(tb1), tb3, fr(tb3));
  }
  // full rounds only (512 tiles of B = 2 rounds of 256 CUs; 1024 of A = 2 rounds of 512 slots): the rate without the partial last round
  const int gb2 = 512, ga2 = 1024;
  auto la2 = [&]() { hipLaunchKernelGGL(loop_a, dim3(ga2), dim3(256), 0, 0, a); };
  auto lb2 = [&]() { hipLaunchKernelGGL(loop_b<3>, dim3(gb2), dim3(512), 0, 0, a); };
  auto lb2n = [&]() { hipLaunchKernelGGL(loop_b<1>, dim3(gb2), dim3(512), 0, 0, a); };
  for (int round = 0; round < 3; ++round) {
    const double ta = time_ms(la2, 60), tb = time_ms(lb2, 60), tbn = time_ms(lb2n, 60);
    const double f2 = flop * 512.0 / 600.0;
    printf("two full rounds: A %.4f ms (%.3f)   B ping-pong+prio %.4f (%.3f)   B ping-pong %.4f (%.3f)\n", ta, 3.0 * f2 / (ta * 1e-3) / 2.5e15, tb,
           3.0 * f2 / (tb * 1e-3) / 2.5e15, tbn, 3.0 * f2 / (tbn * 1e-3) / 2.5e15);
  }
  return 0;
}
