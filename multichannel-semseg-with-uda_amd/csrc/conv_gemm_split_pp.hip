// Implicit-GEMM convolution (forward and stride-1 data gradient) in the split arithmetic of conv_gemm_split.hip on a 256 x 256
// tile owned by EIGHT waves that run as two groups half a K-step apart ("ping-pong"): what the reference computes with
// nn.Conv2d (models/drn.py:21-23, 43-59).
//
// Why.  In conv_gemm_split_kernel a wave's K-step is a serial chain -- 12 fragment reads, 6 LDS-DMA issues with their scalar
// bookkeeping, 24 matrix instructions, a counted wait, the workgroup barrier -- and the two waves that share a SIMD belong to
// two independent workgroups, so nothing keeps one of them in its matrix phase while the other reads: measured matrix pipe
// busy 0.61, 49 % of the wave cycles waiting to issue (profiles/history/r03zb_pmc_sq.json).  Here the SIMD partners are waves w and
// w + 4 of ONE workgroup and the workgroup's barriers alternate their roles (MI355X_MICROARCH.md "Two waves per SIMD";
// cdna_hip_programming.md, the 8-phase GEMM template): between two barriers group 0 (waves 0-3, output rows 0-127) multiplies
// -- nothing but 24 MFMAs -- while group 1 (waves 4-7, rows 128-255) reads its fragments of the same K-step, issues its share
// of the LDS-DMAs for the K-step after next and waits for the share it issued one step earlier; then they swap.  The matrix
// pipe of every SIMD always has exactly one wave feeding it, and the read phase (12 x ds_read_b128, 4 DMAs) hides behind the
// partner's 768 matrix cycles.  The 256 x 256 tile also moves a third fewer operand bytes through L2 and LDS-DMA per FLOP than
// two 256 x 128 tiles.  tools/probes/pingpong.hip (512 -> 512, 3x3, dilation 4, N = 16, random data, two full rounds of
// workgroups): 0.594 of 2.5 PFLOP/s against 0.527 for the 4-wave structure; without the stagger the same 8-wave tile gets 0.44.
//
// One workgroup per CU (256 x 256: 96 KB of LDS, three stages of 16 + 16 KB), so a grid is worth whole rounds of CUs only.  The host
// (pp_wide / mcdseg_internal_conv_pp_pixels below, called from conv_gemm_split.hip::launch_math) plans a convolution either as ONE
// launch of the 256 x 320 (128 x 320) tile -- whose width makes BASELINE config 2's 76800 pixels 0.94 of a round -- or as whole rounds of
// 256 x 256 tiles plus the rest on 256 x 128 tiles, or leaves it to the 4-wave tiles.  K order, term order and the epilogue are those
// of conv_gemm_split_kernel, so outputs are bit for bit the same whichever kernel computes a pixel, and so are the BatchNorm partial
// rows of the 256- and 128-pixel tiles (one per 64 pixels; the 320-pixel tiles emit one per 160: the same moments, regrouped)
// (tests/test_kernels_gpu.py::test_conv_pingpong_tile, ::test_conv_pingpong_wide_tile_by_default).
//
// LDS timeline (stage of step s = s mod 3; "interval" = the time between two consecutive workgroup barriers; group g reads
// step s in interval 2s + g and multiplies it in interval 2s + g + 1):
//   * a group issues its share of step s+2's DMAs in its read phase of step s -- into the stage step s-1 occupied, whose last
//     readers (group 0 in interval 2s-2, group 1 in 2s-1) both waited lgkmcnt(0) before the barrier that closed their interval;
//   * at the end of its read phase of step s a group waits vmcnt(share) -- everything but the share just issued -- so its share
//     of step s+1 has landed before the barrier; group 1 passes that barrier at the end of interval 2s+1, and the first read of
//     step s+1 is group 0's in interval 2s+2.
#include <cstdlib>
#include <type_traits>

#include "conv_split_params.h"
#include "options.h"

namespace {

// WM x WN: 32 x 32 accumulator tiles per wave; QM x QN: the four waves of a group; the two groups split the rows.  Five shapes are built:
//   <4, 2, 1, 4>  256 x 256, waves of 128 x 64, ONE workgroup per CU (96 KB of LDS): the fewest operand bytes per FLOP;
//   <2, 2, 2, 2>  256 x 128, waves of 64 x 64 (<= 128 VGPRs), TWO workgroups per CU (2 x 72 KB): half the tile, so a partial last
//                 round of workgroups costs half as much, and one workgroup's epilogue hides behind the other's K loop.
//   <1, 5, 2, 2>  128 x 320, waves of 32 x 160: the same for the 128-channel layers.
//   <1, 5, 4, 1>  256 x 160, waves of 32 x 160: for HALF the pixels (N = 8: 38400 = 240 tiles of 160).
// (Round 5 built <1, 5, 1, 4> too -- 64 x 640 for the 64-channel layers, 307200 pixels = 480 tiles -- and measured it SLOWER than the
// 4-wave 64 x 256 tiles: forward 0.115 against 0.091 ms, data gradient 0.105 against 0.088, the step + 1.4 ms.  An ablation that moves
// the pixel operand once per channel chunk instead of once per tap (MCD_ABLATE & 64 below) leaves the 64-row layers unchanged: they pay
// per K-step -- barrier, counted wait, DMA look-ahead -- against 12 matrix instructions per wave, not per byte.  DESIGN.md section 4.1f.)
//   <2, 5, 2, 2>  256 x 320, waves of 64 x 160, one workgroup per CU (120 KB): the tile WIDTH is the knob against round quantisation --
//                 76800 pixels (BASELINE config 2, 1/8 resolution) are 300 tiles of 256 (1.17 rounds of 256 CUs: the last 44 tiles
//                 cost a whole round) but 240 tiles of 320 (0.94 of ONE round).  A wave's 160 pixels are one BatchNorm partial row.
template <class P, bool DGRAD, int WM, int WN, int QM, int QN>
__global__ __launch_bounds__(512, (WM * WN <= 4 ? 4 : 2)) void conv_gemm_split_pp_kernel(ConvSplitParams p) {
  static_assert(QM * QN == 4, "four waves per group");
  constexpr int WAVES_N = QN;
  constexpr int BM = 2 * QM * WM * 32, BN = QN * WN * 32, NT = 512, NS = 3;
  constexpr int NP = P::NP;          // pieces per operand
  constexpr int NPU = P::NPU;        // ... of which the policy multiplies (and the loop stages) the first NPU
  constexpr int NQ = 2 * NP;         // (piece, k-half) planes per K-step
  typedef typename P::frag frag;
  constexpr int A_UNITS = 2 * NPU * BM;            // 16-byte units of the weight slab per K-step (piece-major slab: the staged pieces are its head)
  constexpr int A_DMAS = (A_UNITS + NT - 1) / NT;  // per thread (units past A_UNITS: out of range, zeros past the staged planes)
  constexpr int B_UNITS = 2 * NPU * BN;            // 16-byte units of the pixel operand per K-step: [piece][k-half][pixel]
  constexpr int B_DMAS = (B_UNITS + NT - 1) / NT;  // per thread (units past B_UNITS: an out-of-range DMA that deposits zeros past the planes)
  // KDEEP K-steps of 16 channels per barrier interval (SplitF16x1D: 2): sub-step d of an interval lies A_SUB / B_SUB units behind
  // sub-step 0 in the stage -- for the one-term arithmetic with 256 / 128 pixels exactly where the second piece would lie
  constexpr int KDEEP = P::KDEEP;
  static_assert(KDEEP == 1 || NPU == 1, "deep K-steps use the slots of the pieces the policy does not stage");
  constexpr int A_SUB = 2 * NPU * BM > A_DMAS * NT ? 2 * NPU * BM : A_DMAS * NT;
  constexpr int B_SUB = 2 * NPU * BN > B_DMAS * NT ? 2 * NPU * BN : B_DMAS * NT;
  constexpr int DMA_PER_STEP = KDEEP * (A_DMAS + B_DMAS);
  constexpr int A_BYTES = KDEEP > 1 ? KDEEP * A_SUB * 16 : (NQ * BM > A_DMAS * NT ? NQ * BM : A_DMAS * NT) * 16;
  constexpr int B_BYTES = KDEEP > 1 ? KDEEP * B_SUB * 16 : (NQ * BN > B_DMAS * NT ? NQ * BN : B_DMAS * NT) * 16;
  // a pixel tile that divides the workgroup gives every thread ONE gather pixel for all its units; otherwise (320) one per unit
  constexpr bool SAMEPX = (NT % BN) == 0;
  constexpr int NPX = SAMEPX ? 1 : B_DMAS;
  // a pixel tile of 160: a wave's 64 units may straddle the two k-halves of a piece (never two pieces: 2 BN % 64 == 0), so the k-half
  // is the LANE's -- folded into its base offset -- and the scalar offset addresses the chunk's first channel group only
  constexpr bool LANEHALF = (BN % 64) != 0;
  static_assert(BM % 64 == 0 && (2 * BN) % 64 == 0 && (!LANEHALF || !SAMEPX),
                "every wave issues the same number of DMAs (the waits are counted); a wave's units share a piece");
  static_assert(NS * (A_BYTES + B_BYTES) <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;                   // [NS][piece][half][BM][16 B]
  unsigned char* Bs = smem + NS * A_BYTES;    // [NS][piece][half][BN][16 B]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp_id = wave >> 2;  // the group: SIMD partners are waves w and w + 4 (MI355X_MICROARCH.md item 9: split by wave >= 4, not by parity)
  const int wm = grp_id * QM + (wave & 3) / QN;  // wave row / column inside the tile
  const int wn = (wave & 3) % QN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int m_tiles = p.Mp / BM;
  const int n_tiles = p.tile_n1;  // this launch: pixel tiles tile_n0 .. tile_n1 - 1
  const int per_xcd = (n_tiles - p.tile_n0 + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int tile_m = slot % m_tiles;
  const int tile_n = p.tile_n0 + xcd * per_xcd + slot / m_tiles;
  if (tile_n >= n_tiles) return;

  // ---- this thread's gather pixels (they never change): unit i of the thread is unit t + i NT of the stage, plane (t + i NT) / BN =
  // piece * 2 + k-half, pixel (t + i NT) % BN of the tile -- plane and the wave's first pixel are wave-uniform (BN % 64 == 0)
  const int HWd = p.Hd * p.Wd;
  const int HWs = p.Hs * p.Ws;
  constexpr unsigned OOB = 0x80000000u;
  __amdgpu_buffer_rsrc_t cb_rs[NPU];  // one descriptor per piece of the companion (each below 2 GiB; the pieces of a batch slice are not adjacent)
#pragma unroll
  for (int pc = 0; pc < NPU; ++pc)
    cb_rs[pc] = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.src_cb + pc * p.cb_piece_stride), 0, p.cb_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wp_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.wp_bytes, 0x00020000);
  const int C8 = p.Cs >> 3;
  const int taps = p.KH * p.KW;  // <= 32 (host)
  // whether tap q of a pixel falls into the zero padding is one bit of a per-thread mask; the tap's address offset is the same for
  // every thread (scalar ALU): a K-step spends three vector instructions per gather pixel on addressing
  unsigned valid_mask[NPX], vbase[NPX];
  int khalf[NPX];
#pragma unroll
  for (int x = 0; x < NPX; ++x) {
    const int plane_l = (t + x * NT) / BN;  // (LANEHALF: this lane's plane; else the wave's)
    khalf[x] = LANEHALF ? (plane_l & 1) : 0;
    const int bj = (t + x * NT) % BN;
    const int pix = tile_n * BN + bj;
    const bool pv = pix < p.P;
    int pn = 0, py = 0, px = 0;
    if (pv) {
      pn = pix / HWd;
      const int rem = pix - pn * HWd;
      py = rem / p.Wd;
      px = rem - py * p.Wd;
    }
    const unsigned pix_base = (unsigned)pn * (unsigned)C8 * (unsigned)HWs;  // in 16-byte units
    unsigned vm = 0;
    for (int q = 0; q < taps; ++q) {
      const int ky = q / p.KW, kx = q - ky * p.KW;
      const int sy = DGRAD ? py + p.pad - ky * p.dil : py * p.stride + ky * p.dil - p.pad;
      const int sx = DGRAD ? px + p.pad - kx * p.dil : px * p.stride + kx * p.dil - p.pad;
      vm |= (pv && sy >= 0 && sy < p.Hs && sx >= 0 && sx < p.Ws ? 1u : 0u) << q;
    }
    valid_mask[x] = (LANEHALF && plane_l >= 2 * NPU) ? 0u : vm;  // (a unit past the staged planes)
    vbase[x] = DGRAD ? pix_base + (unsigned)(py * p.Ws + px) : pix_base + (unsigned)(py * p.stride * p.Ws + px * p.stride);
    if (LANEHALF) vbase[x] += (unsigned)(khalf[x] * HWs);
  }

  unsigned a_voff[A_DMAS];
#pragma unroll
  for (int i = 0; i < A_DMAS; ++i) {
    const int id = t + i * NT;
    const int plane = id / BM;
    const int m = id - plane * BM;
    a_voff[i] = id < A_UNITS ? ((unsigned)plane * (unsigned)p.Mp + (unsigned)m) * 16u : OOB;
  }

  // loader state: K order is channel-chunk outer, tap inner (the shifted re-reads of a 16-channel slab are back to back)
  int l_tap = 0, l_c0 = 0, l_ky = 0, l_kx = 0, l_kstep = 0;
  auto issue_sub = [&](int buf, int d) {  // this wave's share of one K-step: A_DMAS KB of the weight slab, NPU x 1 KB of gathered pixel units
#if defined(__HIP_DEVICE_COMPILE__)  // the LDS address space does not exist in the host pass of this translation unit
    const int a_soff = (l_kstep * NQ * p.Mp + tile_m * BM) * 16;
    unsigned char* adst = As + buf * A_BYTES + (d * A_SUB + wave * 64) * 16;
#pragma unroll
    for (int i = 0; i < A_DMAS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wp_rs, (__attribute__((address_space(3))) void*)(adst + i * NT * 16), 16, a_voff[i], a_soff, 0, 0);
    const int rel = DGRAD ? (p.pad - l_ky * p.dil) * p.Ws + (p.pad - l_kx * p.dil) : (l_ky * p.dil - p.pad) * p.Ws + (l_kx * p.dil - p.pad);
    unsigned voff[NPX];
#pragma unroll
    for (int x = 0; x < NPX; ++x) {
      bool ok = ((valid_mask[x] >> l_tap) & 1u) != 0;
      if (LANEHALF) ok = ok && (l_c0 >> 3) + khalf[x] < C8;  // (the ragged last chunk's second k-half)
      voff[x] = ok ? (vbase[x] + (unsigned)rel) * 16u : OOB;
    }
#if defined(MCD_ABLATE) && (MCD_ABLATE & 64)  // timing only (never shipped): the pixel operand moved for ONE tap of a channel chunk
    if (l_tap == 0)
#endif
#pragma unroll
    for (int i = 0; i < B_DMAS; ++i) {
      const int unit0 = wave * 64 + i * NT;  // this wave's first unit of the stage: wave-uniform, and so is its plane (LANEHALF: its piece)
      const int plane = LANEHALF ? 2 * (unit0 / (2 * BN)) : unit0 / BN;  // piece = plane / 2, k-half = plane % 2
      const int grp = (l_c0 >> 3) + (plane & 1);
      // (a ragged last chunk, or a unit past the staged pieces: the range check deposits zeros)
      const int soff = (grp < C8 && plane < 2 * NPU) ? grp * HWs * 16 : 0x7FFFFFFF;
      unsigned char* bdst = Bs + buf * B_BYTES + (d * B_SUB + unit0) * 16;
      const __amdgpu_buffer_rsrc_t rs = (NPU > 1 && plane >= 2) ? cb_rs[NPU - 1] : cb_rs[0];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)bdst, 16, voff[SAMEPX ? 0 : i], soff, 0, 0);
    }
#else
    (void)buf;
    (void)d;
#endif
  };
  auto advance = [&]() {
    ++l_kstep;
    ++l_tap;
    if (++l_kx == p.KW) {
      l_kx = 0;
      ++l_ky;
    }
    if (l_tap == taps) {
      l_tap = 0;
      l_kx = 0;
      l_ky = 0;
      l_c0 += 16;
    }
  };

  auto issue = [&](int buf) {  // one barrier interval's operands: KDEEP consecutive K-steps
    issue_sub(buf, 0);
    if constexpr (KDEEP > 1) {
#pragma unroll
      for (int d = 1; d < KDEEP; ++d) {
        advance();
        issue_sub(buf, d);
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = taps * (p.Kp / 16) / KDEEP;  // (KDEEP > 1: the host launches this instantiation for an even number of K-steps only)
  issue(0);
  if (nsteps > 1) {
    advance();
    issue(1);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (grp_id == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier behind group 0 from here to the end of the K loop

  frag fa[KDEEP > 1 ? KDEEP : NP][WM], fb[KDEEP > 1 ? KDEEP : NP][WN];
  int cur = 0, nxt2 = 2;
  for (int s = 0; s < nsteps; ++s) {
    // ---- read phase (the partner group multiplies meanwhile)
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * (32 * WM) + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * (32 * WN) + l31) * 16;
    if constexpr (KDEEP > 1) {
#pragma unroll
      for (int d = 0; d < KDEEP; ++d) {
#pragma unroll
        for (int i = 0; i < WM; ++i) fa[d][i] = *reinterpret_cast<const frag*>(a_base + (d * A_SUB + i * 32) * 16);
#pragma unroll
        for (int j = 0; j < WN; ++j) fb[d][j] = *reinterpret_cast<const frag*>(b_base + (d * B_SUB + j * 32) * 16);
      }
    } else {
#pragma unroll
    for (int pc = 0; pc < NPU; ++pc) {
#pragma unroll
      for (int i = 0; i < WM; ++i) fa[pc][i] = *reinterpret_cast<const frag*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < WN; ++j) fb[pc][j] = *reinterpret_cast<const frag*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < nsteps) {
      advance();
      issue(nxt2);
      // this wave's share of step s+1 has landed (the share of step s+2 stays in flight) and its fragment reads are back
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA_PER_STEP) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- matrix phase (the partner group reads meanwhile): the policy's cross terms, smallest first, term-major -- consecutive
    // instructions go to different accumulator tiles; same sums in the same order per tile as conv_gemm_split_kernel
    if constexpr (KDEEP > 1) {  // the interval's K-steps in K order (one term each): the sums of SplitF16x1's kernel, in its order
#pragma unroll
      for (int d = 0; d < KDEEP; ++d)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa[d][i], fb[d][j], acc[i][j]);
    } else {
#pragma unroll
    for (int tm = 0; tm < P::NTERMS; ++tm)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa[P::TA[tm]][i], fb[P::TB[tm]][j], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
    nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
  }
  if (grp_id == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last matrix-phase barrier

  // ---- epilogue (conv_gemm_split_kernel's, without the parity classes): acc[i][j][r] = D[row][col], row = (r&3) + 8*(r>>2) + 4*lh, col = l31
  if constexpr (P::SCALED) {
    const float osc = mcd_scale_of_bound(*p.src_bound) * mcd_scale_of_bound(*p.w_bound);  // exact power of two
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= osc;
  }
  const int m_wave = tile_m * BM + wm * (32 * WM);
  const int p_wave = tile_n * BN + wn * (32 * WN);
  if (!DGRAD && p.bias != nullptr) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (m < p.M) ? p.bias[m] : 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j][r] += bv;
      }
  }
  if (!DGRAD && p.ep_scale != nullptr) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float sc = (m < p.M) ? p.ep_scale[m] : 0.f;
        const float sh = (m < p.M) ? p.ep_shift[m] : 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j][r] = fmaf(acc[i][j][r], sc, sh);
      }
  }
  bool colv[WN];
  size_t dbase[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int pp = p_wave + j * 32 + l31;
    colv[j] = pp < p.P;
    int n = 0, rem = 0;
    if (colv[j]) {
      n = pp / HWd;
      rem = pp - n * HWd;
    }
    dbase[j] = (size_t)n * p.M * HWd + rem;
  }
  // ---- 16-bit channel-blocked output (the one-term arithmetic only: every other instantiation is compiled without this block).  The
  // registers r = 4 q .. 4 q + 3 of an accumulator tile are channels 8 q + 4 lh .. + 3 of one pixel: half of a 16-byte unit, so a
  // store instruction of the wave covers 32 whole units = 512 consecutive bytes (the fp32 planar form: two runs of 128 bytes).
  bool half_out = false;
  if constexpr (P::HALF_OUT) half_out = p.dst16 != nullptr;
  if constexpr (P::HALF_OUT) {
    if (half_out) {
      float inv_zs = 1.f;
      if (!DGRAD) {
        const float zb = (float)(taps * p.Cs) * (*p.src_bound) * (*p.w_bound);  // |z| <= taps Cs max|x| max|w|
        inv_zs = 1.f / mcd_scale_of_bound(zb);
        if (blockIdx.x == 0 && t == 0 && p.dst_bound != nullptr) *p.dst_bound = zb;
      }
      const int M8 = p.M >> 3;
      size_t ubase[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int pp = p_wave + j * 32 + l31;
        int n = 0, rem = 0;
        if (colv[j]) {
          n = pp / HWd;
          rem = pp - n * HWd;
        }
        ubase[j] = (size_t)n * M8 * HWd + rem;
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m0 = m_wave + i * 32 + 8 * q;
          if (m0 < p.M) {  // (M is a multiple of 8: host)
            const size_t gofs = (size_t)(m0 >> 3) * HWd;
            mcd_bf16x4 add[WN];
            if (DGRAD && p.ep_res16 != nullptr) {
#pragma unroll
              for (int j = 0; j < WN; ++j)
                if (colv[j]) add[j] = *reinterpret_cast<const mcd_bf16x4*>((const char*)p.ep_res16 + (ubase[j] + gofs) * 16 + lh * 8);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j)
              if (colv[j]) {
                char* dst = (char*)p.dst16 + (ubase[j] + gofs) * 16 + lh * 8;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * q + k];
                if (DGRAD) {
                  if (p.ep_res16 != nullptr) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += (float)add[j][k];
                  }
                  mcd_bf16x4 o;
#pragma unroll
                  for (int k = 0; k < 4; ++k) o[k] = (__bf16)v[k];
                  *reinterpret_cast<mcd_bf16x4*>(dst) = o;
                } else {
                  mcd_f16x4 o;
#pragma unroll
                  for (int k = 0; k < 4; ++k) o[k] = (_Float16)(v[k] * inv_zs);
#if defined(MCD_ABLATE) && (MCD_ABLATE & 256)  // timing only (never shipped): the 16-bit results of one pixel column in 32 only
                  if (l31 == 0)
#endif
                  *reinterpret_cast<mcd_f16x4*>(dst) = o;
                }
              }
          }
        }
    }
  }
  if (!half_out && DGRAD && p.ep_res != nullptr && p.ep_res_lds) {
    // ---- data gradient + addend, the addend's tile staged through LDS.  With one workgroup per CU nothing hides the latency of
    // epilogue loads, and the accumulators leave ~20 registers to keep them in flight (measured: matrix pipe busy 0.78 -> 0.69 with
    // the addend loaded value by value).  A pass stages 16 rows of every wave row block -- rows i * 32 + 16 h .. + 15, what the
    // accumulator registers r = 8 h .. 8 h + 7 hold -- as [row][BN] floats by 16-byte LDS-DMAs (a quad of pixels never straddles two
    // images: host), then every lane adds its values from LDS and stores.  Same sum, same order: bit for bit the direct form.
#if defined(__HIP_DEVICE_COMPILE__)
    // rows of a 32-row block per pass: 16 (the registers r = 8 h .. 8 h + 7), or 8 (r = 4 h .. 4 h + 3) where 16 do not fit the LDS
    constexpr int RPB = (2 * QM * 16 * BN * 4 <= NS * (A_BYTES + B_BYTES)) ? 16 : 8;
    constexpr int ROWS_PASS = 2 * QM * RPB, U_ROW = BN / 4, UNITS = ROWS_PASS * U_ROW, UPT = UNITS / NT;
    static_assert(UNITS % NT == 0 && ROWS_PASS * BN * 4 <= NS * (A_BYTES + B_BYTES), "a pass fits the LDS the K loop has left");
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_res, 0, p.ep_res_bytes, 0x00020000);
    const float* stage = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int h = 0; h < 32 / RPB; ++h) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the previous pass's (the K loop's) readers are done
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
          const int u = k * NT + t;
          const int row_l = u / U_ROW, c4 = u - row_l * U_ROW;
          const int m = tile_m * BM + (row_l / RPB) * (32 * WM) + i * 32 + h * RPB + (row_l % RPB);
          const int pp = tile_n * BN + c4 * 4;
          unsigned voff = OOB;
          if (m < p.M && pp < p.P) {
            const int n = pp / HWd;
            voff = ((unsigned)(n * p.M + m) * (unsigned)HWd + (unsigned)(pp - n * HWd)) * 4u;
          }
          __builtin_amdgcn_raw_ptr_buffer_load_lds(res_rs, (__attribute__((address_space(3))) void*)(smem + (k * NT + wave * 64) * 16), 16, voff, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < RPB / 2; ++rr) {
          const int r = h * (RPB / 2) + rr;
          const int e = (r & 3) + 8 * (r >> 2) + 4 * lh - h * RPB;  // row of the block inside this pass
          const int m = m_wave + i * 32 + h * RPB + e;
          const float* srow = stage + (wm * RPB + e) * BN + wn * (32 * WN) + l31;
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const float v = acc[i][j][r] + srow[j * 32];
            if (m < p.M && colv[j]) p.dst[dbase[j] + (size_t)m * HWd] = v;
          }
        }
      }
#endif
    return;
  }
  if (!half_out) {
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m_wave + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (colv[j]) {
            float v = acc[i][j][r];
            if (!DGRAD && p.ep_scale != nullptr) {
              if (p.ep_res != nullptr) v += p.ep_res[dbase[j] + (size_t)m * HWd];
              if (p.ep_relu) v = fmaxf(v, 0.f);
            }
            // data gradient: the other gradient of the same tensor (a residual block's shortcut), added as autograd would add it
            if (DGRAD && p.ep_res != nullptr) v += p.ep_res[dbase[j] + (size_t)m * HWd];
            p.dst[dbase[j] + (size_t)m * HWd] = v;
          }
      }
    }
  }
#if defined(MCD_ABLATE) && (MCD_ABLATE & 512)  // timing only (never shipped): no BatchNorm partial rows
  if (false) {
#else
  if (!DGRAD && p.stats != nullptr) {
#endif
    int cntw = p.P - p_wave;
    cntw = cntw < 0 ? 0 : (cntw > 32 * WN ? 32 * WN : cntw);
    const float inv = cntw > 0 ? 1.f / (float)cntw : 0.f;
    const size_t srow = ((size_t)(tile_n * WAVES_N + wn) * 3) * p.Mp;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      float my_mean = 0.f, my_m2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) v += colv[j] ? acc[i][j][r] : 0.f;
        const float mean = wave_half_sum(v) * inv;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const float d = acc[i][j][r] - mean;
          q += colv[j] ? d * d : 0.f;
        }
        q = wave_half_sum(q);
        if (l31 == r) {
          my_mean = mean;
          my_m2 = q;
        }
      }
      if (l31 < 16) {
        const int m = m_wave + i * 32 + (l31 & 3) + 8 * (l31 >> 2) + 4 * lh;
        p.stats[srow + m] = (float)cntw;
        p.stats[srow + p.Mp + m] = my_mean;
        p.stats[srow + 2 * (size_t)p.Mp + m] = my_m2;
      }
    }
  }
}

// Fewest whole rounds of 256 x 256 tiles for which the ping-pong kernels take a convolution.  Measured at BASELINE config 2's sizes
// (N = 16, 60 x 80, same box, ms forward / data gradient, 4-wave tiles -> ping-pong): 512 -> 512 (600 tiles = 2.3 rounds) 0.892 / 0.894 ->
// 0.816 / 0.800, 256 -> 512 0.476 / 0.477 -> 0.434 / 0.469; but 256 -> 256 (300 tiles = 1.2 rounds) 0.249 / 0.255 -> 0.253 / 0.246 and
// 128 -> 256 0.135 -> 0.145: with one round the 15 % of the tiles left over take a third of the time.  (Read per call: tests lower it.)
int pp_min_rounds() { return (int)mcd_opt(MCD_OPT_PP_MIN_ROUNDS); }

int compute_units() {  // one workgroup per CU: a launch is worth whole rounds of this many tiles
  // option PP_CUS: a test plans a small batch as if the chip had fewer CUs, which gives it the rounds -- and hence the
  // launch plan -- of a batch that many times larger (BASELINE config 5's N = 32 plan at N = 2 with 16 "CUs")
  if (mcd_opt(MCD_OPT_PP_CUS) > 0) return (int)mcd_opt(MCD_OPT_PP_CUS);
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;  // MI355X
    return cus;
  }();
  return n;
}

// MCDSEG_PINGPONG (development knob, read per call: a test runs one problem several ways): 0 the 4-wave tiles only; 1 whole rounds of
// 256 x 256 tiles + the rest on the 4-wave tiles; 2 the 256 x 128 ping-pong tile for everything; 3 (default) whole rounds of 256 x 256
// tiles + the rest on the 256 x 128 ping-pong tile
int pp_mode() { return (int)mcd_opt(MCD_OPT_PINGPONG); }

bool pp_applies(const ConvSplitParams& p, int math, bool dgrad, int bm = 256) {
  if (p.src_cb == nullptr || mcd_math_pieces(math) != 2 || (p.Mp % bm) != 0) return false;
  return p.KH * p.KW <= 32 && !(dgrad && p.stride != 1) && (p.Cs & 7) == 0;
}

template <class P, bool DGRAD, int WM, int WN, int QM, int QN>
void launch_tile(const ConvSplitParams& q, hipStream_t st) {
  constexpr int BM = 2 * QM * WM * 32;
  const dim3 grid((unsigned)(8 * ceil_div(q.tile_n1 - q.tile_n0, 8) * (q.Mp / BM)));
  hipLaunchKernelGGL((conv_gemm_split_pp_kernel<P, DGRAD, WM, WN, QM, QN>), grid, dim3(512), 0, st, q);
}

// SplitF16x1D (two K-steps per barrier interval) takes an even number of K-steps; not on the 256 x 128 tile, whose two workgroups per CU
// would no longer fit the LDS side by side (option PP_DEEP = 0: never)
template <int WM, int WN, int QM, int QN>
bool deep_applies(const ConvSplitParams& q, int math) {
  constexpr bool small = WM == 2 && WN == 2 && QM == 2 && QN == 2;
  return math == MCDSEG_MATH_F16X1 && !small && mcd_opt(MCD_OPT_PP_DEEP) != 0 && ((q.KH * q.KW * (q.Kp / 16)) & 1) == 0;
}

template <int WM, int WN, int QM, int QN>
void launch_math(const ConvSplitParams& q, int math, bool dgrad, hipStream_t st) {
  if constexpr (!(WM == 2 && WN == 2 && QM == 2 && QN == 2)) {
    if (deep_applies<WM, WN, QM, QN>(q, math)) {
      if (dgrad)
        launch_tile<SplitF16x1D, true, WM, WN, QM, QN>(q, st);
      else
        launch_tile<SplitF16x1D, false, WM, WN, QM, QN>(q, st);
      return;
    }
  }
  if (math == MCDSEG_MATH_F16X1) {
    if (dgrad)
      launch_tile<SplitF16x1, true, WM, WN, QM, QN>(q, st);
    else
      launch_tile<SplitF16x1, false, WM, WN, QM, QN>(q, st);
  } else {
    if (dgrad)
      launch_tile<SplitF16x3, true, WM, WN, QM, QN>(q, st);
    else
      launch_tile<SplitF16x3, false, WM, WN, QM, QN>(q, st);
  }
}

// The 256 x 320 tile takes a WHOLE convolution when that is the cheaper plan.  Costs in units of one 256 x 256 tile's time on a CU:
//   hybrid  = whole rounds of 256 x 256 tiles + the rest on 256 x 128 tiles (half a unit per workgroup, two per CU side by side)
//   wide    = ceil(tiles320 / CUs) rounds of 1.25
// and the hybrid plan exists only from pp_min_rounds() whole rounds on; without it the wide tile must fill its rounds to 80 %
// (MCDSEG_PP_WIDE_FILL, per cent; above 100 = never) to beat the 4-wave tiles.  Measured at BASELINE config 2 (76800 pixels, same
// box, ms forward / data gradient): 256 -> 256 (240 wide tiles = 0.94 round; 4-wave tiles 0.246 / 0.246) 0.205 / 0.200;
// 512 -> 512 (480 wide tiles, cost 2.5; hybrid cost 2.5, 0.793 / 0.773) 0.757 / 0.750 -- ties go to the single launch.
// MCDSEG_PINGPONG = 4 forces the wide tile wherever the kernel applies.
// With output rows a multiple of 128 only (the 128-channel layers) the same kernel runs as a 128 x 320 tile (waves of 32 x 160) under the
// fill rule alone: pp_wide returns 2.  Where the 320-pixel tile does not fill its rounds but a 160-pixel one does (256 rows at half of
// config 2's pixels: config 4's N = 8) the 256 x 160 tile takes the convolution: 3.
int pp_wide(const ConvSplitParams& p, int math, bool dgrad) {
  const int mode = pp_mode();
  if ((mode != 3 && mode != 4) || !pp_applies(p, math, dgrad, 128)) return 0;
  const int kind = (p.Mp % 256) == 0 ? 1 : 2;
  if (mode == 4) return kind;
  const int64_t fill = mcd_opt(MCD_OPT_PP_WIDE_FILL);
  if (fill > 100) return 0;
  const int64_t cus = compute_units();
  if (kind == 2) {
    if (mcd_opt(MCD_OPT_PP_WIDE128) == 0) return 0;  // development knob: 0 = the 128-row variant off
    const int64_t tiles = ceil_div64(p.P, 320) * (p.Mp / 128), rounds_w = ceil_div64(tiles, cus);
    return tiles * 100 >= fill * rounds_w * cus ? 2 : 0;
  }
  const int64_t m_tiles = p.Mp / 256;
  const int64_t tiles = ceil_div64(p.P, 320) * m_tiles, rounds_w = ceil_div64(tiles, cus);
  const int min_rounds = pp_min_rounds() > 1 ? pp_min_rounds() : 1;
  const int64_t rounds = (p.P / 256) * m_tiles / cus;
  if (rounds < min_rounds) {  // against the 4-wave tiles: the 320-pixel tile, else the 160-pixel one, where either fills its rounds
    if (tiles * 100 >= fill * rounds_w * cus) return 1;
    const int64_t tiles_h = ceil_div64(p.P, 160) * m_tiles, rounds_h = ceil_div64(tiles_h, cus);
    return tiles_h * 100 >= fill * rounds_h * cus ? 3 : 0;
  }
  int64_t n_pp = rounds * cus / m_tiles;  // (mcdseg_internal_conv_pp_pixels)
  if (n_pp > p.P / 256) n_pp = p.P / 256;
  const int64_t rest = ceil_div64(p.P - n_pp * 256, 128) * m_tiles;  // 256 x 128 workgroups
  // 4 x cost: hybrid = 4 rounds + 2 ceil(rest / CUs); wide = 5 rounds_w
  return 5 * rounds_w <= 4 * rounds + 2 * ceil_div64(rest, cus) ? 1 : 0;
}

}  // namespace

// 1 when the whole convolution runs on the 256 x 320 ping-pong tile (BatchNorm partial rows of 160 pixels)
int mcdseg_internal_conv_pp_wide(const ConvSplitParams& p, int math, bool dgrad) { return pp_wide(p, math, dgrad); }

// 1 when the ping-pong launches of this problem (all but the 256 x 128 tile's) run as SplitF16x1D (kernel names for profilers)
int mcdseg_internal_conv_pp_deep(const ConvSplitParams& p, int math) { return deep_applies<4, 2, 1, 4>(p, math) ? 1 : 0; }

// Pixels (a multiple of 256, counted from pixel 0) of this problem that the 256 x 256 ping-pong tile takes: whole rounds of one tile
// per CU; 0 when it does not apply (no pre-split operand, three-piece arithmetic, output rows not a multiple of 256, strided data
// gradient, more than 32 taps, less than one round of tiles).
int64_t mcdseg_internal_conv_pp_pixels(const ConvSplitParams& p, int math, bool dgrad) {
  const int mode = pp_mode();
  if (pp_wide(p, math, dgrad)) return p.P;  // (everything, on the 256 x 320 tile)
  if ((mode != 1 && mode != 3) || !pp_applies(p, math, dgrad)) return 0;
  const int64_t m_tiles = p.Mp / 256, n_full = p.P / 256, cus = compute_units();
  const int64_t rounds = n_full * m_tiles / cus;
  if (rounds < 1 || rounds < pp_min_rounds()) return 0;
  int64_t n_pp = rounds * cus / m_tiles;  // pixel tiles (all their row tiles) that make whole rounds
  if (n_pp > n_full) n_pp = n_full;
  return n_pp * 256;
}

// 1 when the pixels the 256 x 256 tile leaves (all of them when it takes none) run on the 256 x 128 ping-pong tile rather than on the
// 4-wave tiles of conv_gemm_split.hip
int mcdseg_internal_conv_pp_rest(const ConvSplitParams& p, int math, bool dgrad) {
  const int mode = pp_mode();
  if (!pp_applies(p, math, dgrad)) return 0;
  if (mode == 2) return 1;
  return mode == 3 && (p.P / 256) * (p.Mp / 256) / compute_units() >= (pp_min_rounds() > 1 ? pp_min_rounds() : 1) ? 1 : 0;
}

int mcdseg_internal_conv_pp_launch(const ConvSplitParams& p, int math, bool dgrad, int64_t pixels, hipStream_t st) {
  ConvSplitParams q = p;
  q.sub = 0;
  q.tile_n0 = 0;
  if (const int kind = pp_wide(p, math, dgrad)) {
    q.tile_n1 = ceil_div(p.P, kind == 3 ? 160 : 320);
    if (kind == 1)
      launch_math<2, 5, 2, 2>(q, math, dgrad, st);
    else if (kind == 2)
      launch_math<1, 5, 2, 2>(q, math, dgrad, st);
    else
      launch_math<1, 5, 4, 1>(q, math, dgrad, st);
    MCD_LAUNCH_CHECK("conv_gemm_split_pp (256 / 128 x 320, 256 x 160)");
    return 0;
  }
  q.tile_n1 = (int)(pixels / 256);
  launch_math<4, 2, 1, 4>(q, math, dgrad, st);
  MCD_LAUNCH_CHECK("conv_gemm_split_pp");
  return 0;
}

// the pixels from pix0 (a multiple of 256) to the end on the 256 x 128 ping-pong tile
int mcdseg_internal_conv_pp_rest_launch(const ConvSplitParams& p, int math, bool dgrad, int64_t pix0, hipStream_t st) {
  ConvSplitParams q = p;
  q.sub = 0;
  q.tile_n0 = (int)(pix0 / 128);
  q.tile_n1 = ceil_div(p.P, 128);
  launch_math<2, 2, 2, 2>(q, math, dgrad, st);
  MCD_LAUNCH_CHECK("conv_gemm_split_pp (256 x 128)");
  return 0;
}
