// Weight gradient of the 256-channel-and-wider convolutions from both pre-split companions, as an eight-wave PING-PONG kernel
// (the structure of conv_gemm_split_pp.hip) over a STREAM-K decomposition: what the reference gets from autograd for nn.Conv2d
// (models/drn.py:21-23; adapt_trainer.py:170-176).
//
// dW[tap][co][ci] = sum over pixels of dZ[co][pixel] * X[ci][pixel + shift(tap)].  A TILE is 256 (co) x 256 (ci) of one tap; its K
// dimension are the 16-pixel stages (16 consecutive pixels of one output row) of all images: KT = N * tiles_x * Ho K-steps.  The
// operands sit in LDS as in memory -- 16-byte units of 8 channels x 1 pixel, deposited by LDS-DMA -- and ds_read_b64_tr_b16 forms the
// MFMA fragments (8 pixels of one channel per lane), as in conv_wgrad_split_tr_kernel (conv_wgrad_split.hip).  What differs from that
// kernel's LDS image is the shape of a DMA: there one instruction moves 4 pixels of 16 channel groups (16 runs of 64 bytes); here it
// moves 16 PIXELS of 4 channel groups (lane = group * 16 + pixel): 4 runs of 256 contiguous bytes, a quarter of the memory requests
// for the same kilobyte.  The four groups of one instruction are four apart (d, d+4, d+8, d+12), and instruction d of a 128-channel block
// lands at d * (1 KB + 64 B): the four consecutive groups a transposing read touches then sit in four different instructions'
// images, 64 bytes apart modulo the 256-byte bank row -- conflict-free.
//
// Ping-pong: waves 0-3 (co rows 0-127) and 4-7 (rows 128-255) are SIMD partners half a K-step apart; between two workgroup
// barriers one group multiplies (24 MFMAs) while the other issues its 24 transposing reads of the same stage and its 4 LDS-DMAs
// for the stage after next.  One workgroup per CU (three stages of 32 KB).
//
// Stream-K: the weight gradient is summed over split slabs anyway, so the work need not be cut along tile borders.  The (tile, K-step)
// pairs, tile-major with the tap fastest (the nine taps of a channel block share their operands through L2), form one range that is cut
// into nwg equal pieces of L K-steps -- one workgroup per CU, every CU the same work, whatever the layer's tile count (conv_wgrad_split_tr
// at cfg2: 1 152 workgroups on 512 slots = 2.25 rounds).  A piece spans one or two tiles; each SEGMENT (piece x tile) writes its raw
// 256 x 256 accumulators to slab number  g / L + g / KT  (g = the segment's first flattened K-step; the numbers of a tile's segments
// are consecutive), and wgrad_reduce_sk_kernel sums a tile's slabs in K order in fp64 -- a fixed order, so the result is reproducible --
// and applies scale(x) * scale(dz).
#include <cstdlib>
#include <type_traits>

#include "split.h"

namespace {

struct WgradPpParams {
  const void* x_cb;
  const void* dy_cb;
  float* slab;
  int N, Cin, H, W, Cout, Ho, Wo;
  int KH, KW, stride, pad, dil;
  int co_tiles, ci_tiles;  // 256-channel tiles
  int tiles_x, tiles_y;    // 16 x 1 pixel stages of one image: ceil(Wo / 16) per row, Ho rows
  int kt;                  // K-steps of one tile: N * tiles_x * tiles_y
  int L, nwg;              // K-steps per workgroup (stream-K) or per slab (slab plan), workgroups
  int slabs;               // 0: stream-K pieces; > 0: the slab plan -- item i = (K slab i / tiles, tile i % tiles), workgroup w takes items w, w + nwg, ...
  int items;               // slabs * tiles
  int x_cb_bytes, dy_cb_bytes;                // ONE piece of each companion (this call's images)
  long long x_piece_stride, dy_piece_stride;  // bytes between the pieces (the companions' own batch: mcdseg_conv_desc.Ncb)
};

template <class P>
__global__ __launch_bounds__(512) void conv_wgrad_split_pp_kernel(WgradPpParams p) {
  static_assert(P::NP == 2, "two-piece policies");
  constexpr int WM = 4, WN = 2;
  constexpr int NP = P::NP;
  constexpr int NBLK = 2;               // 128-channel blocks per operand
  typedef typename P::frag frag;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  constexpr int NDMA = 4;               // DMA instructions per (operand, piece, block) and stage: 4 channel groups x 16 pixels each
  constexpr int DSTR = 1024 + 64;       // bytes from one instruction's image [group 4][pixel 16][16 B] to the next (64 B: the bank offset)
  constexpr int BLKB = NDMA * DSTR;     // one 128-channel block of one piece
  constexpr int UNIT = NBLK * BLKB;     // one piece of one operand
  constexpr int STAGE = 2 * NP * UNIT;  // [operand: dZ, X][piece]
  constexpr int NS = 3;
  static_assert(NS * STAGE <= 160 * 1024, "LDS");
  // One-term arithmetic, policy SplitF16x1D (the second piece of neither operand is ever moved): the idle piece-1 slots of the three
  // stages are three MORE stages -- logical stage c sits at (c % 3) * STAGE + (c / 3) * UNIT of both operands -- and a barrier interval
  // takes TWO K-steps: 24 transposing reads and 16 matrix instructions per wave and phase instead of 12 and 8, the LDS-DMAs four to five
  // K-steps ahead instead of two.  With one term a K-step is a third of the matrix work on half the bytes: the two barriers per step and
  // the short prefetch held the kernel at 0.26-0.38 of the pipe on every layer of BASELINE config 5, 1 x 1 or 3 x 3
  // (profiles/r06_wgrad_f16x1_shapes.txt).  The same stages in the same order: the same sums, bit for bit (SplitF16x1 keeps the
  // one-step loop: option WGRAD_PP_DEEP = 0, tests).
  constexpr bool DEEP = P::KDEEP == 2 && P::NPU == 1 && NP == 2;
  constexpr int PRO = DEEP ? 4 : 2;  // K-steps the prologue puts in flight
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp_id = wave >> 2;  // SIMD partners are waves w and w + 4
  const int wm = grp_id, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int T_ = p.KH * p.KW;

  // consecutive workgroup numbers on one XCD (the dispatcher deals blockIdx round-robin over the 8 XCDs; the grid is a multiple of 8)
  const int w_id = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (w_id >= p.nwg) return;
  const int tiles_all = p.co_tiles * p.ci_tiles * T_;
  const long long total = (long long)tiles_all * p.kt;
  long long g_next = (long long)w_id * p.L;
  long long g_end = g_next + p.L;
  if (g_end > total) g_end = total;
  int item = w_id;  // (slab plan)
  if (p.slabs > 0) {
    g_next = 0;
    g_end = item < p.items ? 1 : 0;
  }

  // ---- DMA role of this wave (wave-uniform): operand side (0 = dZ, the "A" rows; 1 = X, the "B" rows), piece, 128-channel block
  const int opnd = wave >> 2;
  const int piece = (wave >> 1) & 1;
  const int blk = wave & 1;
  const bool isx = opnd == 1;
  const int px = lane & 15;   // pixel of the stage
  const int cgl = lane >> 4;  // channel group d + 4 cgl of the wave's block, for DMA instruction d
  const int sH = isx ? p.H : p.Ho;
  const int sW = isx ? p.W : p.Wo;
  const int sS = isx ? p.stride : 1;
  const int sC8 = (isx ? p.Cin : p.Cout) >> 3;
  const int sHW = sH * sW;
  // the descriptor covers this wave's PIECE and starts `bias` bytes below it so that the SGPR offset (tile + tap shift) is never negative
  const int bias = p.pad * 16 + 16;
  const char* sptr = (const char*)(isx ? p.x_cb : p.dy_cb) + piece * (isx ? p.x_piece_stride : p.dy_piece_stride) - bias;
  const int sbytes = (isx ? p.x_cb_bytes : p.dy_cb_bytes) + bias;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sptr, 0, sbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int lane_x = px * sS;
  unsigned char* const unit_lds = smem + (opnd * NP + piece) * UNIT + blk * BLKB;
  const bool dma_on = piece < P::NPU;  // (a piece the policy never multiplies -- SplitF16x1's second -- is not moved)
  const int ntiles = p.tiles_x * p.tiles_y;

  // transposed-read address of this lane for the 32-row block i (channel groups 4i .. 4i+3) of a 128-channel block: 16-lane group
  // g = lane >> 4 covers rows 16 (g & 1) .. +15; lane 4q + pp of the group supplies pixel q of the quad, channels 4 pp .. 4 pp + 3 --
  // i.e. channel group j = 2 (g & 1) + (pp >> 1) of the four, which is row i of instruction j's image
  const int gl = lane & 15;
  const int tq = gl >> 2, tpp = gl & 3;
  const int trow = ((((lane >> 4) & 1) * 2 + (tpp >> 1)) * DSTR) + tq * 16 + 8 * (tpp & 1) + (2 * lh) * 64;  // (+ the k-half's first quad)
  const int a_lane = wm * BLKB + trow;                          // the wave's 128 dZ channels = block wm
  const int b_lane = (wn >> 1) * BLKB + (wn & 1) * 512 + trow;  // its 64 X channels = rows 2 (wn & 1), 2 (wn & 1) + 1 of block wn >> 1

  // The transposing reads are issued as inline assembly: for a compiler-visible one the wait-count pass puts an s_waitcnt vmcnt(0) in
  // front whenever an LDS-DMA may be pending (it cannot tell that the DMA targets another stage), which would drain the prefetch at the
  // head of every read phase.  They complete at the lgkmcnt(0) of the wait that closes the read phase; `settle` re-defines the registers
  // behind that wait so that no matrix instruction can be scheduled ahead of it.
  auto tr_read_at = [&](unsigned base, auto off_c) -> s16x4 {
    s16x4 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(decltype(off_c)::value) : "memory");
#else
    (void)base;
    v = s16x4{};
#endif
    return v;
  };
  auto frag_of = [&](unsigned base, auto lo_c, auto hi_c) -> frag {  // k slots 0-3 from this lane's first quad, 4-7 from its second
    const s16x4 lo = tr_read_at(base, lo_c);
    const s16x4 hi = tr_read_at(base, hi_c);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag, v);
  };
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#else
  const unsigned smem_lds = 0;
#endif

  while (g_next < g_end) {  // the segments of this workgroup's piece: the rest of the current tile's K range, at most
    int tile, k0, k1, slab_id;
    if (p.slabs > 0) {  // one item: the K slab item / tiles of tile item % tiles
      const int sl = item / tiles_all;
      tile = item - sl * tiles_all;
      k0 = sl * p.L;
      k1 = k0 + p.L < p.kt ? k0 + p.L : p.kt;
      slab_id = item;
      item += p.nwg;
      if (item >= p.items) g_next = g_end;
    } else {
      tile = (int)(g_next / p.kt);
      k0 = (int)(g_next - (long long)tile * p.kt);
      k1 = (g_end - (long long)tile * p.kt < p.kt) ? (int)(g_end - (long long)tile * p.kt) : p.kt;
      slab_id = (int)(g_next / p.L) + tile;
      g_next += k1 - k0;
    }
    const int tap = tile % T_;
    const int tile_ci = (tile / T_) % p.ci_tiles;
    const int tile_co = tile / (T_ * p.ci_tiles);
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int shy = isx ? ky * p.dil - p.pad : 0;
    const int shx = isx ? kx * p.dil - p.pad : 0;
    const int cg0 = (isx ? tile_ci : tile_co) * 32 + blk * 16;  // first channel group of this wave's block
    unsigned vconst[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) vconst[d] = (cg0 + d + 4 * cgl) < sC8 ? (unsigned)((d + 4 * cgl) * sHW + px * sS) * 16u : OOB;

    // loader state: K-step -> (image, output row, 16-pixel column block), advanced incrementally
    int l_n = k0 / ntiles;
    int l_ty = (k0 - l_n * ntiles) / p.tiles_x;
    int l_tx = k0 - l_n * ntiles - l_ty * p.tiles_x;
    auto stage_off = [](int c) { return DEEP ? (c < NS ? c * STAGE : (c - NS) * STAGE + UNIT) : c * STAGE; };
    auto issue = [&](int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (dma_on) {
        const int sbase = (l_n * sC8 + cg0) * sHW;  // 16-byte units inside the piece
        const int iy = l_ty * sS + shy;
        const int ux = l_tx * 16 * sS + shx;
        const bool colok = (unsigned)(ux + lane_x) < (unsigned)sW;
        const int soff = ((unsigned)iy < (unsigned)sH) ? (sbase + iy * sW + ux) * 16 + bias : 0x7FFFFFFF;
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(unit_lds + stage_off(stage) + d * DSTR), 16,
                                                   colok ? vconst[d] : OOB, soff, 0, 0);
      }
#else
      (void)stage;
#endif
    };
    auto advance = [&]() {
      if (++l_tx == p.tiles_x) {
        l_tx = 0;
        if (++l_ty == p.tiles_y) {
          l_ty = 0;
          ++l_n;
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

    const int nsteps = k1 - k0;
    issue(0);
#pragma unroll
    for (int a = 1; a < PRO; ++a)
      if (nsteps > a) {
        advance();
        issue(a);
      }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp_id == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier behind group 0 to the end of the K loop

    if constexpr (DEEP) {
      // ---- two K-steps per barrier interval: interval t = K-steps 2t, 2t + 1 = logical stages 2 dcur, 2 dcur + 1
      const int nint = (nsteps + 1) >> 1;
      frag fa2[2][WM], fb2[2][WN];
      int dcur = 0, dnxt = 2;
      for (int t = 0; t < nint; ++t) {
        const bool has2 = 2 * t + 1 < nsteps;  // (uniform: an odd slab ends on a single K-step; the second stage's reads are then unused)
        const unsigned off0 = (unsigned)stage_off(2 * dcur), off1 = (unsigned)stage_off(2 * dcur + 1);
        const unsigned a0 = smem_lds + off0 + (unsigned)a_lane, b0 = smem_lds + off0 + (unsigned)b_lane;
        const unsigned a1 = smem_lds + off1 + (unsigned)a_lane, b1 = smem_lds + off1 + (unsigned)b_lane;
#define MCD_A2(U, BASE, I) fa2[U][I] = frag_of(BASE, std::integral_constant<int, (I) * 256>{}, std::integral_constant<int, (I) * 256 + 64>{});
#define MCD_B2(U, BASE, J) \
        fb2[U][J] = frag_of(BASE, std::integral_constant<int, NP * UNIT + (J) * 256>{}, std::integral_constant<int, NP * UNIT + (J) * 256 + 64>{});
        MCD_A2(0, a0, 0) MCD_A2(0, a0, 1) MCD_A2(0, a0, 2) MCD_A2(0, a0, 3) MCD_B2(0, b0, 0) MCD_B2(0, b0, 1)
        MCD_A2(1, a1, 0) MCD_A2(1, a1, 1) MCD_A2(1, a1, 2) MCD_A2(1, a1, 3) MCD_B2(1, b1, 0) MCD_B2(1, b1, 1)
#undef MCD_A2
#undef MCD_B2
        __builtin_amdgcn_sched_barrier(0);
        const int nx = 2 * (t + 2);  // first K-step of the interval after next: into the stages the previous interval read
        if (nx + 1 < nsteps) {
          advance();
          issue(2 * dnxt);
          advance();
          issue(2 * dnxt + 1);
          // this wave's shares of the next interval's two stages have landed (those just issued stay in flight); its reads are back
          if (dma_on)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NDMA) : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          if (nx < nsteps) {
            advance();
            issue(2 * dnxt);
          }
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int i = 0; i < WM; ++i) asm volatile("" : "+v"(fa2[u][i]));
#pragma unroll
          for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(fb2[u][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase: K-step 2t into every tile, then K-step 2t + 1 -- per tile the order of the one-step loop
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa2[0][i], fb2[0][j], acc[i][j]);
        if (has2) {
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa2[1][i], fb2[1][j], acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        dcur = dcur == 2 ? 0 : dcur + 1;
        dnxt = dnxt == 2 ? 0 : dnxt + 1;
      }
    } else {
      frag fa[NP][WM], fb[NP][WN];
      int cur = 0, nxt2 = 2;
      for (int s = 0; s < nsteps; ++s) {
        // ---- read phase (the partner group multiplies meanwhile)
        const unsigned base_a = smem_lds + (unsigned)(stage_off(cur) + a_lane), base_b = smem_lds + (unsigned)(stage_off(cur) + b_lane);
        // (compile-time offsets: operand, piece, 32-row block, second quad)
#define MCD_A_FRAG(PC, I) \
        if constexpr ((PC) < P::NPU) \
          fa[PC][I] = frag_of(base_a, std::integral_constant<int, (PC) * UNIT + (I) * 256>{}, std::integral_constant<int, (PC) * UNIT + (I) * 256 + 64>{});
#define MCD_B_FRAG(PC, J) \
        if constexpr ((PC) < P::NPU) \
          fb[PC][J] = frag_of(base_b, std::integral_constant<int, (NP + (PC)) * UNIT + (J) * 256>{}, std::integral_constant<int, (NP + (PC)) * UNIT + (J) * 256 + 64>{});
        MCD_A_FRAG(0, 0) MCD_A_FRAG(0, 1) MCD_A_FRAG(0, 2) MCD_A_FRAG(0, 3) MCD_B_FRAG(0, 0) MCD_B_FRAG(0, 1)
        MCD_A_FRAG(1, 0) MCD_A_FRAG(1, 1) MCD_A_FRAG(1, 2) MCD_A_FRAG(1, 3) MCD_B_FRAG(1, 0) MCD_B_FRAG(1, 1)
#undef MCD_A_FRAG
#undef MCD_B_FRAG
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nsteps) {
          advance();
          issue(nxt2);
          // this wave's share of stage s+1 has landed (the share of stage s+2 stays in flight) and its fragment reads are back
          if (dma_on)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA) : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // (the reads above have completed: re-define their registers behind the wait)
#pragma unroll
        for (int pc = 0; pc < P::NPU; ++pc) {
#pragma unroll
          for (int i = 0; i < WM; ++i) asm volatile("" : "+v"(fa[pc][i]));
#pragma unroll
          for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(fb[pc][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix phase (the partner group reads meanwhile): term-major, the sums and their order per tile of conv_wgrad_split_tr_kernel
#pragma unroll
        for (int tm = 0; tm < P::NTERMS; ++tm)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa[P::TA[tm]][i], fb[P::TB[tm]][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
      }
    }
    if (grp_id == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last matrix-phase barrier: the groups are level again

    // ---- the segment's raw sums (scaled units): slab[slab_id][co 256][ci 256]
    float* out = p.slab + (size_t)slab_id * (256 * 256);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int j = 0; j < WN; ++j) out[row * 256 + wn * 64 + j * 32 + l31] = acc[i][j][r];
      }
  }
}

// dw[co][ci][tap] = scale * sum over the tile's slabs (in K order, fp64).  A block owns 64 consecutive ci of one co for all taps:
// slab reads are coalesced along ci, the T values of a (co, ci) pair leave through LDS as one contiguous run of dw [Cout][Cin][T].
__global__ __launch_bounds__(256) void wgrad_reduce_sk_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int T,
                                                             int ci_tiles, int kt, int L, int slabs, int tiles_all,
                                                             const float* __restrict__ x_bound, const float* __restrict__ dy_bound) {
  __shared__ float stage[64 * 33];
  const int co = blockIdx.y;
  const int ci0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // wave w sums taps w, w + 4, ...: four times the loads in flight per block
  const int ci = ci0 + lane;
  const double sc = x_bound != nullptr ? (double)mcd_scale_of_bound(*x_bound) * (double)mcd_scale_of_bound(*dy_bound) : 1.0;
  const int nci = Cin - ci0 < 64 ? Cin - ci0 : 64;
  for (int tap = wave; tap < T; tap += 4) {
    const int tile = ((co >> 8) * ci_tiles + (ci0 >> 8)) * T + tap;
    const long long g0 = (long long)tile * kt;
    // stream-K: the tile's consecutive slabs s0 .. s1; slab plan: slabs tile, tile + tiles_all, ... (K order either way)
    const int s0 = slabs > 0 ? 0 : (int)(g0 / L) + tile, s1 = slabs > 0 ? slabs - 1 : (int)((g0 + kt - 1) / L) + tile;
    double s = 0.0;
    if (ci < Cin) {
      const size_t SL = slabs > 0 ? (size_t)tiles_all * (256 * 256) : (size_t)(256 * 256);
      const float* src = slab + ((size_t)(slabs > 0 ? tile : s0) * 256 + (co & 255)) * 256 + (ci & 255);
      int k = s0;
      for (; k + 7 <= s1; k += 8, src += 8 * SL) {  // eight loads in flight, added in slab order
        const float v0 = src[0], v1 = src[SL], v2 = src[2 * SL], v3 = src[3 * SL];
        const float v4 = src[4 * SL], v5 = src[5 * SL], v6 = src[6 * SL], v7 = src[7 * SL];
        s += (double)v0;
        s += (double)v1;
        s += (double)v2;
        s += (double)v3;
        s += (double)v4;
        s += (double)v5;
        s += (double)v6;
        s += (double)v7;
      }
      for (; k + 3 <= s1; k += 4, src += 4 * SL) {  // four loads in flight, added in slab order
        const float v0 = src[0], v1 = src[SL], v2 = src[2 * SL], v3 = src[3 * SL];
        s += (double)v0;
        s += (double)v1;
        s += (double)v2;
        s += (double)v3;
      }
      for (; k <= s1; ++k, src += SL) s += (double)*src;
    }
    stage[lane * T + tap] = (float)(s * sc);
  }
  __syncthreads();
  float* out = dw + ((size_t)co * Cin + ci0) * T;
  for (int i = threadIdx.x; i < nci * T; i += 256) out[i] = stage[i];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same structure for the 128-CHANNEL layers (round 5): a tile is 128 (co) x 384 = the THREE taps of one
// kernel row (ky; kx = 0, 1, 2) x 128 (ci).  The three taps of a row contract the same dZ pixels against X at three column shifts, so
// one staged dZ block serves three times the products: per 16-pixel stage a wave issues 20 transposing reads for 18 matrix
// instructions where conv_wgrad_split_tr_kernel's 128 x 128 one-tap tile needs 16 for 12, and the nine taps are three tiles, not nine
// (three times fewer passes over dZ through L2).  The eight LDS-DMA roles of a stage are the same eight 128-channel blocks as above --
// (dZ, piece 0), (dZ, piece 1) and (X at shift kx, piece) for kx = 0, 1, 2 -- so the stage image, the DMA shape, the bank layout
// and the transposed-read addresses are the 256 x 256 kernel's.  Waves: the two groups split the 128 rows (64 each: two 32-row
// blocks), the four waves of a group the 384 columns (96 each: three 32-column blocks, which may lie in two taps).  K is cut by the
// slab plan only; a slab item writes its raw 128 x 384 sums, wgrad_reduce_rt_kernel adds a tile's slabs in K order in fp64.
template <class P>
__global__ __launch_bounds__(512) void conv_wgrad_split_pp3_kernel(WgradPpParams p) {
  static_assert(P::NP == 2, "two-piece policies");
  constexpr int WM = 2, WN = 3;
  constexpr int NP = P::NP;
  typedef typename P::frag frag;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  constexpr int NDMA = 4;
  constexpr int DSTR = 1024 + 64;
  constexpr int BLKB = NDMA * DSTR;     // one 128-channel block of one piece
  constexpr int STAGE = 8 * BLKB;       // [dZ piece 0][dZ piece 1][X piece 0: kx 0, 1, 2][X piece 1: kx 0, 1, 2]
  constexpr int NS = 3;
  static_assert(NS * STAGE <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp_id = wave >> 2;  // SIMD partners are waves w and w + 4
  const int wm = grp_id, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;

  const int w_id = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (w_id >= p.nwg) return;
  const int tiles_all = p.co_tiles * p.ci_tiles * p.KH;

  // ---- DMA role of this wave (wave-uniform): waves 0, 1 = dZ pieces 0, 1; wave 2 + i = X, piece i & 1, tap column kx = i >> 1
  const bool isx = wave >= 2;
  const int piece = isx ? ((wave - 2) & 1) : wave;
  const int kxs = isx ? ((wave - 2) >> 1) : 0;
  const int px = lane & 15;
  const int cgl = lane >> 4;
  const int sH = isx ? p.H : p.Ho;
  const int sW = isx ? p.W : p.Wo;
  const int sS = isx ? p.stride : 1;
  const int sC8 = (isx ? p.Cin : p.Cout) >> 3;
  const int sHW = sH * sW;
  const int bias = p.pad * 16 + 16;
  const char* sptr = (const char*)(isx ? p.x_cb : p.dy_cb) + piece * (isx ? p.x_piece_stride : p.dy_piece_stride) - bias;
  const int sbytes = (isx ? p.x_cb_bytes : p.dy_cb_bytes) + bias;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sptr, 0, sbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int lane_x = px * sS;
  unsigned char* const unit_lds = smem + (isx ? (2 + piece * 3 + kxs) : piece) * BLKB;
  const bool dma_on = piece < P::NPU;
  const int ntiles = p.tiles_x * p.tiles_y;

  const int gl = lane & 15;
  const int tq = gl >> 2, tpp = gl & 3;
  const int trow = ((((lane >> 4) & 1) * 2 + (tpp >> 1)) * DSTR) + tq * 16 + 8 * (tpp & 1) + (2 * lh) * 64;
  const int a_lane = (wm * 2) * 256 + trow;  // the wave's 64 dZ channels = 32-row blocks 2 wm, 2 wm + 1 of the block
  int b_lane[WN];                            // its 96 columns: 32-column block j lies in tap (column / 128), ci block (column % 128) / 32
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col0 = wn * 96 + j * 32;
    b_lane[j] = (2 + (col0 >> 7)) * BLKB + ((col0 & 127) >> 5) * 256 + trow;
  }

  auto tr_read_at = [&](unsigned base, auto off_c) -> s16x4 {
    s16x4 v;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(decltype(off_c)::value) : "memory");
#else
    (void)base;
    v = s16x4{};
#endif
    return v;
  };
  auto frag_of = [&](unsigned base, auto lo_c, auto hi_c) -> frag {
    const s16x4 lo = tr_read_at(base, lo_c);
    const s16x4 hi = tr_read_at(base, hi_c);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag, v);
  };
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#else
  const unsigned smem_lds = 0;
#endif

  for (int item = w_id; item < p.items; item += p.nwg) {  // slab plan: item = (K slab item / tiles, tile item % tiles)
    const int sl = item / tiles_all;
    const int tile = item - sl * tiles_all;
    const int k0 = sl * p.L;
    const int k1 = k0 + p.L < p.kt ? k0 + p.L : p.kt;
    const int ky = tile % p.KH;
    const int tile_ci = (tile / p.KH) % p.ci_tiles;
    const int tile_co = tile / (p.KH * p.ci_tiles);
    const int shy = isx ? ky * p.dil - p.pad : 0;
    const int shx = isx ? kxs * p.dil - p.pad : 0;
    const int cg0 = (isx ? tile_ci : tile_co) * 16;  // first channel group of this wave's 128-channel block
    unsigned vconst[NDMA];
#pragma unroll
    for (int d = 0; d < NDMA; ++d) vconst[d] = (cg0 + d + 4 * cgl) < sC8 ? (unsigned)((d + 4 * cgl) * sHW + px * sS) * 16u : OOB;

    int l_n = k0 / ntiles;
    int l_ty = (k0 - l_n * ntiles) / p.tiles_x;
    int l_tx = k0 - l_n * ntiles - l_ty * p.tiles_x;
    auto issue = [&](int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (dma_on) {
        const int sbase = (l_n * sC8 + cg0) * sHW;
        const int iy = l_ty * sS + shy;
        const int ux = l_tx * 16 * sS + shx;
        const bool colok = (unsigned)(ux + lane_x) < (unsigned)sW;
        const int soff = ((unsigned)iy < (unsigned)sH) ? (sbase + iy * sW + ux) * 16 + bias : 0x7FFFFFFF;
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(unit_lds + stage * STAGE + d * DSTR), 16,
                                                   colok ? vconst[d] : OOB, soff, 0, 0);
      }
#else
      (void)stage;
#endif
    };
    auto advance = [&]() {
      if (++l_tx == p.tiles_x) {
        l_tx = 0;
        if (++l_ty == p.tiles_y) {
          l_ty = 0;
          ++l_n;
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

    const int nsteps = k1 - k0;
    issue(0);
    if (nsteps > 1) {
      advance();
      issue(1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (grp_id == 1) __builtin_amdgcn_s_barrier();  // the stagger: group 1 runs one barrier behind group 0 to the end of the K loop

    frag fa[NP][WM], fb[NP][WN];
    int cur = 0, nxt2 = 2;
    for (int s = 0; s < nsteps; ++s) {
      // ---- read phase (the partner group multiplies meanwhile)
      const unsigned base_a = smem_lds + (unsigned)(cur * STAGE + a_lane);
      const unsigned base_b0 = smem_lds + (unsigned)(cur * STAGE + b_lane[0]), base_b1 = smem_lds + (unsigned)(cur * STAGE + b_lane[1]),
                     base_b2 = smem_lds + (unsigned)(cur * STAGE + b_lane[2]);
#define MCD_A3_FRAG(PC, I) \
      if constexpr ((PC) < P::NPU) \
        fa[PC][I] = frag_of(base_a, std::integral_constant<int, (PC) * BLKB + (I) * 256>{}, std::integral_constant<int, (PC) * BLKB + (I) * 256 + 64>{});
#define MCD_B3_FRAG(PC, J, BASE) \
      if constexpr ((PC) < P::NPU) \
        fb[PC][J] = frag_of(BASE, std::integral_constant<int, (PC) * 3 * BLKB>{}, std::integral_constant<int, (PC) * 3 * BLKB + 64>{});
      MCD_A3_FRAG(0, 0) MCD_A3_FRAG(0, 1) MCD_B3_FRAG(0, 0, base_b0) MCD_B3_FRAG(0, 1, base_b1) MCD_B3_FRAG(0, 2, base_b2)
      MCD_A3_FRAG(1, 0) MCD_A3_FRAG(1, 1) MCD_B3_FRAG(1, 0, base_b0) MCD_B3_FRAG(1, 1, base_b1) MCD_B3_FRAG(1, 2, base_b2)
#undef MCD_A3_FRAG
#undef MCD_B3_FRAG
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < nsteps) {
        advance();
        issue(nxt2);
        if (dma_on)
          asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NDMA) : "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
#pragma unroll
      for (int pc = 0; pc < P::NPU; ++pc) {
#pragma unroll
        for (int i = 0; i < WM; ++i) asm volatile("" : "+v"(fa[pc][i]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(fb[pc][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- matrix phase: term-major, the sums and their order per 32 x 32 tile of conv_wgrad_split_tr_kernel
#pragma unroll
      for (int tm = 0; tm < P::NTERMS; ++tm)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(fa[P::TA[tm]][i], fb[P::TB[tm]][j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      cur = cur == 2 ? 0 : cur + 1;
      nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    if (grp_id == 0) __builtin_amdgcn_s_barrier();  // pairs with group 1's last matrix-phase barrier: the groups are level again

    // ---- the item's raw sums (scaled units): slab[item][co 128][kx 3][ci 128]
    float* out = p.slab + (size_t)item * (128 * 384);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int j = 0; j < WN; ++j) out[row * 384 + wn * 96 + j * 32 + l31] = acc[i][j][r];
      }
  }
}

// dw[co][ci][tap] = scale * sum over the tile's slabs in K order (fp64); tile = (128 co, 128 ci, kernel row ky), slab item = slab * tiles +
// tile.  A block owns 64 consecutive ci of one co for all taps (wave w: taps w, w + 4, ...); the T values of a (co, ci) pair leave
// through LDS as one contiguous run of dw [Cout][Cin][T].
__global__ __launch_bounds__(256) void wgrad_reduce_rt_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int KH,
                                                             int ci_tiles, int slabs, int tiles_all, const float* __restrict__ x_bound,
                                                             const float* __restrict__ dy_bound) {
  __shared__ float stage[64 * 33];
  const int T = KH * 3;
  const int co = blockIdx.y;
  const int ci0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = ci0 + lane;
  const double sc = x_bound != nullptr ? (double)mcd_scale_of_bound(*x_bound) * (double)mcd_scale_of_bound(*dy_bound) : 1.0;
  const int nci = Cin - ci0 < 64 ? Cin - ci0 : 64;
  const size_t SL = (size_t)tiles_all * (128 * 384);
  for (int tap = wave; tap < T; tap += 4) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int tile = ((co >> 7) * ci_tiles + (ci0 >> 7)) * KH + ky;
    double s = 0.0;
    if (ci < Cin) {
      const float* src = slab + ((size_t)tile * 128 + (co & 127)) * 384 + kx * 128 + (ci & 127);
      int k = 0;
      for (; k + 8 <= slabs; k += 8, src += 8 * SL) {  // eight loads in flight, added in slab order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[u * SL];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (double)v[u];
      }
      for (; k < slabs; ++k, src += SL) s += (double)*src;
    }
    stage[lane * T + tap] = (float)(s * sc);
  }
  __syncthreads();
  float* out = dw + ((size_t)co * Cin + ci0) * T;
  for (int i = threadIdx.x; i < nci * T; i += 256) out[i] = stage[i];
}

int pp_compute_units() {
  if (mcd_opt(MCD_OPT_WGRAD_PP_CUS) > 0) return (int)mcd_opt(MCD_OPT_WGRAD_PP_CUS);  // development knob: plan the weight gradient for fewer CUs than the chip has
  if (mcd_opt(MCD_OPT_PP_CUS) > 0) return (int)mcd_opt(MCD_OPT_PP_CUS);              // development knob (shared with conv_gemm_split_pp.hip)
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;  // MI355X
    return cus;
  }();
  return n;
}

}  // namespace

// The stream-K plan of this geometry: workgroups and K-steps per workgroup; 0 workgroups when the kernel does not apply (two-piece
// arithmetic on both companions, 256-channel blocks on both sides, at most 32 taps, and enough K-steps that a workgroup's piece
// spans at most two tiles and is worth a launch).  MCDSEG_WGRAD_PP=0 turns it off (read per call: tests).
//
// Two decompositions (MCDSEG_WGRAD_PP: 1 = stream-K, 2 or unset = the slab plan; 0 = this kernel off):
//   stream-K   the flattened (tile, K-step) range in nwg equal pieces -- perfect balance, but the workgroups of an XCD sit at 32 different
//              K positions, so every tile streams its operands through the fabric by itself: measured 2 x FETCH_SIZE = 2.9 GB per launch
//              for 0.22 GB of algorithmic bytes (profiles/history/r04z_pmc_traffic.json), and the rate falls with the batch (430 TFLOP/s at
//              N = 8, 398 at 16, 382 at 32) as the operands outgrow the Infinity Cache;
//   slab plan  the K range is cut into `slabs` = floor(rounds * CUs / tiles) slabs and item i = (slab i / tiles, tile i % tiles): the
//              workgroups of one XCD hold consecutive items, i.e. the tiles (taps, channel blocks) of ONE slab, walk the same pixels
//              at the same time and share them through the XCD's L2.  252 of 256 CUs busy at BASELINE config 2 (36, 18 or 9 tiles).
int mcdseg_internal_wgrad_pp_plan(const mcdseg_conv_desc* d, int math, int* L, size_t* slab_floats, int* slabs_out) {
  if (mcd_opt(MCD_OPT_WGRAD_PP) == 0) return 0;
  const bool stream_k = mcd_opt(MCD_OPT_WGRAD_PP) == 1;
  if (slabs_out) *slabs_out = 0;
  if (mcd_storage_math(math) != MCDSEG_MATH_F16X3 || (d->Cin & 7) || (d->Cout & 7) || d->Cin < 129 || d->Cout < 129) return 0;
  const int T = d->KH * d->KW;
  if (T > 32 || d->pad > 128) return 0;
  const int64_t co_tiles = ceil_div(d->Cout, 256), ci_tiles = ceil_div(d->Cin, 256);
  if (co_tiles * 256 - d->Cout >= 128 || ci_tiles * 256 - d->Cin >= 128) return 0;  // (a half-empty tile: the 128-channel kernels are better)
  const int64_t kt = (int64_t)d->N * ceil_div(d->Wo, 16) * d->Ho;
  const int64_t tiles = co_tiles * ci_tiles * T, total = tiles * kt;
  int64_t nwg = pp_compute_units();
  if (total < nwg * 32) return 0;  // (less than 32 K-steps per CU: the launch is all prologue and slab traffic)
  if (!stream_k) {
    // rounds of items per workgroup: the fewest whose slabs keep >= 95 % of the CUs busy (else the best of up to four)
    int64_t best_r = 0, best_ls = 0, best_ns = 0;
    double best_eff = 0.0;
    for (int64_t r = 1; r <= 4; ++r) {
      int64_t ns = r * nwg / tiles;
      if (ns < 1) continue;
      if (ns > kt / 32) ns = kt / 32;  // (an item of less than 32 K-steps is all prologue)
      if (ns < 1) continue;
      const int64_t ls = ceil_div64(kt, ns);
      ns = ceil_div64(kt, ls);
      const double eff = (double)total / ((double)nwg * (double)(ceil_div64(ns * tiles, nwg) * ls));
      if (eff > best_eff + 0.02 || best_r == 0) best_r = r, best_ls = ls, best_ns = ns, best_eff = eff;
      if (best_eff >= 0.95) break;
    }
    if (best_r > 0 && best_eff >= 0.80) {
      const int64_t items = best_ns * tiles;
      const int64_t rounds = ceil_div64(items, nwg);
      if (L) *L = (int)best_ls;
      if (slabs_out) *slabs_out = (int)best_ns;
      if (slab_floats) *slab_floats = (size_t)items * 256 * 256;
      return (int)ceil_div64(items, rounds);
    }
  }
  const int64_t l = ceil_div64(total, nwg);
  if (l > kt) return 0;  // (a piece would span more than two tiles)
  nwg = ceil_div64(total, l);
  if (L) *L = (int)l;
  if (slab_floats) *slab_floats = (size_t)(nwg + tiles + 1) * 256 * 256;
  return (int)nwg;
}

int mcdseg_internal_wgrad_pp_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const float* x_bound, const void* dy_cb,
                                    const float* dy_bound, float* dw, float* slab, hipStream_t st) {
  WgradPpParams p;
  int L = 0, slabs = 0;
  const int nwg = mcdseg_internal_wgrad_pp_plan(d, math, &L, nullptr, &slabs);
  MCD_REQUIRE(nwg > 0, "conv_wgrad_split_pp: the geometry has no stream-K plan");
  const int64_t xb = (int64_t)d->N * d->Cin * d->H * d->W * 2, yb = (int64_t)d->N * d->Cout * d->Ho * d->Wo * 2;
  MCD_REQUIRE(xb + 4096 < (1ll << 31) && yb + 4096 < (1ll << 31) && (d->Ncb == 0 || d->Ncb >= d->N),
              "conv_wgrad_split_pp: pre-split operands need < 2 GiB per operand piece and Ncb >= N");
  p.x_cb = x_cb; p.dy_cb = dy_cb; p.slab = slab;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_tiles = ceil_div(d->Cout, 256); p.ci_tiles = ceil_div(d->Cin, 256);
  p.tiles_x = ceil_div(d->Wo, 16); p.tiles_y = d->Ho;
  p.kt = d->N * p.tiles_x * p.tiles_y;
  p.L = L; p.nwg = nwg;
  p.slabs = slabs; p.items = slabs * p.co_tiles * p.ci_tiles * d->KH * d->KW;
  p.x_cb_bytes = (int)xb; p.dy_cb_bytes = (int)yb;
  p.x_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cin * d->H * d->W * 2;  // (F16X1 reads piece 0 only)
  p.dy_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cout * d->Ho * d->Wo * 2;
  const dim3 grid((unsigned)(8 * ceil_div(nwg, 8)));
  if (math == MCDSEG_MATH_F16X1 && mcd_opt(MCD_OPT_WGRAD_PP_DEEP) != 0)
    hipLaunchKernelGGL(conv_wgrad_split_pp_kernel<SplitF16x1D>, grid, dim3(512), 0, st, p);  // six logical stages (see the kernel)
  else if (math == MCDSEG_MATH_F16X1)
    hipLaunchKernelGGL(conv_wgrad_split_pp_kernel<SplitF16x1>, grid, dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_split_pp_kernel<SplitF16x3>, grid, dim3(512), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_split_pp");
  const int T = d->KH * d->KW;
  MCD_REQUIRE(T <= 33, "conv_wgrad_split_pp: more than 33 taps");
  hipLaunchKernelGGL(wgrad_reduce_sk_kernel, dim3((unsigned)ceil_div(d->Cin, 64), (unsigned)d->Cout), dim3(256), 0, st, (const float*)slab, dw, d->Cout,
                     d->Cin, T, p.ci_tiles, p.kt, L, slabs, p.co_tiles * p.ci_tiles * T, x_bound, dy_bound);
  MCD_LAUNCH_CHECK("wgrad_reduce_sk");
  return 0;
}

// The slab plan of the row-of-taps kernel: workgroups (0 when it does not apply), K-steps per slab, slabs, slab floats.  It takes the
// 3 x KH kernels whose channel counts leave the 256 x 256 kernel out -- more than 64 channels on both sides, at most 128 on one -- when
// the 128-channel tiles are at least three quarters full.  MCDSEG_WGRAD_PP3=0 turns it off (read per call: tests).
int mcdseg_internal_wgrad_pp3_plan(const mcdseg_conv_desc* d, int math, int* L, size_t* slab_floats, int* slabs_out) {
  if (mcd_opt(MCD_OPT_WGRAD_PP3) == 0) return 0;
  if (slabs_out) *slabs_out = 0;
  if (mcd_storage_math(math) != MCDSEG_MATH_F16X3 || (d->Cin & 7) || (d->Cout & 7)) return 0;
  const int lo = d->Cin < d->Cout ? d->Cin : d->Cout;
  if (lo <= 64 || (d->Cin >= 129 && d->Cout >= 129)) return 0;
  if (d->KW != 3 || d->KH > 10 || d->pad > 128) return 0;
  const int64_t co_tiles = ceil_div(d->Cout, 128), ci_tiles = ceil_div(d->Cin, 128);
  if (4 * (int64_t)d->Cout * d->Cin < 3 * (co_tiles * 128) * (ci_tiles * 128)) return 0;  // (tiles less than three quarters full)
  const int64_t kt = (int64_t)d->N * ceil_div(d->Wo, 16) * d->Ho;
  const int64_t tiles = co_tiles * ci_tiles * d->KH, total = tiles * kt;
  const int64_t nwg = pp_compute_units();
  if (total < nwg * 32) return 0;
  int64_t best_r = 0, best_ls = 0, best_ns = 0;
  double best_eff = 0.0;
  for (int64_t r = 1; r <= 4; ++r) {
    int64_t ns = r * nwg / tiles;
    if (ns < 1) continue;
    if (ns > kt / 32) ns = kt / 32;
    if (ns < 1) continue;
    const int64_t ls = ceil_div64(kt, ns);
    ns = ceil_div64(kt, ls);
    const double eff = (double)total / ((double)nwg * (double)(ceil_div64(ns * tiles, nwg) * ls));
    if (eff > best_eff + 0.02 || best_r == 0) best_r = r, best_ls = ls, best_ns = ns, best_eff = eff;
    if (best_eff >= 0.95) break;
  }
  if (best_r == 0 || best_eff < 0.80) return 0;
  const int64_t items = best_ns * tiles;
  const int64_t rounds = ceil_div64(items, nwg);
  if (items * 128 * 384 >= (1ll << 31)) return 0;
  if (L) *L = (int)best_ls;
  if (slabs_out) *slabs_out = (int)best_ns;
  if (slab_floats) *slab_floats = (size_t)items * 128 * 384;
  return (int)ceil_div64(items, rounds);
}

int mcdseg_internal_wgrad_pp3_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const float* x_bound, const void* dy_cb,
                                     const float* dy_bound, float* dw, float* slab, hipStream_t st) {
  WgradPpParams p;
  int L = 0, slabs = 0;
  const int nwg = mcdseg_internal_wgrad_pp3_plan(d, math, &L, nullptr, &slabs);
  MCD_REQUIRE(nwg > 0 && slabs > 0, "conv_wgrad_split_pp3: the geometry has no plan");
  const int64_t xb = (int64_t)d->N * d->Cin * d->H * d->W * 2, yb = (int64_t)d->N * d->Cout * d->Ho * d->Wo * 2;
  MCD_REQUIRE(xb + 4096 < (1ll << 31) && yb + 4096 < (1ll << 31) && (d->Ncb == 0 || d->Ncb >= d->N),
              "conv_wgrad_split_pp3: pre-split operands need < 2 GiB per operand piece and Ncb >= N");
  p.x_cb = x_cb; p.dy_cb = dy_cb; p.slab = slab;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_tiles = ceil_div(d->Cout, 128); p.ci_tiles = ceil_div(d->Cin, 128);
  p.tiles_x = ceil_div(d->Wo, 16); p.tiles_y = d->Ho;
  p.kt = d->N * p.tiles_x * p.tiles_y;
  p.L = L; p.nwg = nwg;
  p.slabs = slabs; p.items = slabs * p.co_tiles * p.ci_tiles * d->KH;
  p.x_cb_bytes = (int)xb; p.dy_cb_bytes = (int)yb;
  p.x_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cin * d->H * d->W * 2;  // (F16X1 reads piece 0 only)
  p.dy_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cout * d->Ho * d->Wo * 2;
  const dim3 grid((unsigned)(8 * ceil_div(nwg, 8)));
  if (math == MCDSEG_MATH_F16X1)
    hipLaunchKernelGGL(conv_wgrad_split_pp3_kernel<SplitF16x1>, grid, dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_split_pp3_kernel<SplitF16x3>, grid, dim3(512), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_split_pp3");
  hipLaunchKernelGGL(wgrad_reduce_rt_kernel, dim3((unsigned)ceil_div(d->Cin, 64), (unsigned)d->Cout), dim3(256), 0, st, (const float*)slab, dw, d->Cout,
                     d->Cin, d->KH, p.ci_tiles, slabs, p.co_tiles * p.ci_tiles * d->KH, x_bound, dy_bound);
  MCD_LAUNCH_CHECK("wgrad_reduce_rt");
  return 0;
}
