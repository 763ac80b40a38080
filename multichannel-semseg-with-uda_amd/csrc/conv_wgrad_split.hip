// Weight-gradient convolution on the 16-bit matrix pipe with split fp32 operands (policies of split.h; see
// conv_gemm_split.hip for the arithmetic).  128 x 128 (co x ci) tiles only -- the layers that carry the FLOPs; thin
// layers stay on the f32 kernels of conv_wgrad.hip.  The slabs hold the sums in SCALED units for SplitF16x3; the
// fixed-order reduce of conv_wgrad.hip multiplies by scale(x) * scale(dy).
//
//   D[co][ci] (one tap) = sum_pix dY[co][pix] * X[ci][pix + shift(tap)]
//
// The contraction runs over pixels, 16 per K-step (one K=16 MFMA block).  A thread stages one pixel PAIR of four
// dY rows and four X rows: lanes run along the pixels of a row, so global reads stay pixel-contiguous; each pair is
// split into 16-bit pieces and lands as one 32-bit word in the fragment image [piece][k-half][row][8 x 16 bit], which the
// MFMA lanes read back as ds_read_b128 over 512 contiguous bytes per half-wave.  Slabs + fixed-order fp64 reduce
// as in conv_wgrad.hip (same plan, same workspace).
#include <cstdlib>
#include <type_traits>

#include "split.h"

namespace {

struct WgradSplitParams {
  const float* x;
  const float* dy;
  float* slab;
  int N, Cin, H, W, Cout, Ho, Wo;
  int KH, KW, stride, pad, dil;
  int co_p, ci_p;
  int chunk, chunks_per_img, splits;
  int x_bytes, dy_bytes;
  const float* x_bound;   // SplitF16x3: bound scalars of the two operands
  const float* dy_bound;
};

// a pixel pair of one row -> NP words, word pc = (piece pc of v0) | (piece pc of v1) << 16
template <class P>
__device__ __forceinline__ void split_pair(float v0, float v1, float inv_scale, unsigned (&w)[P::NP]) {
  typename P::elem q0[P::NP], q1[P::NP];
  P::split(v0, inv_scale, q0);
  P::split(v1, inv_scale, q1);
#pragma unroll
  for (int pc = 0; pc < P::NP; ++pc)
    w[pc] = (unsigned)__builtin_bit_cast(unsigned short, q0[pc]) | ((unsigned)__builtin_bit_cast(unsigned short, q1[pc]) << 16);
}

template <class P>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(WgradSplitParams p) {
  constexpr int BM = 128, BN = 128, BKP = 16, NT = 256;
  constexpr int WM = 2, WN = 2, WAVES_N = 2;
  constexpr int NP = P::NP;
  typedef typename P::frag frag;
  constexpr int A_BYTES = 2 * NP * BM * 16, B_BYTES = 2 * NP * BN * 16;  // [piece NP][k-half 2][row][16 B]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * A_BYTES;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ci_tiles = p.ci_p / BN;
  const int co_tiles = p.co_p / BM;
  const int T_ = p.KH * p.KW;
  const int per_split = co_tiles * ci_tiles * T_;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  int rem = slot % per_split;
  const int tap = rem % T_;
  rem /= T_;
  const int tile_ci = rem % ci_tiles;
  const int tile_co = rem / ci_tiles;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int ky = tap / p.KW;
  const int kx = tap - ky * p.KW;
  const int HoWo = p.Ho * p.Wo;
  const int HW = p.H * p.W;
  const int r_begin = chunk_id * p.chunk;
  int r_end = r_begin + p.chunk;
  if (r_end > HoWo) r_end = HoWo;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t dy_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int q = t & 7;      // pixel pair within the 16-pixel step
  const int row0 = t >> 3;  // rows row0 + 32 i, i < 4
  const int a_soff0 = (n * p.Cout + tile_co * BM) * HoWo * 4;
  const int b_soff0 = (n * p.Cin + tile_ci * BN) * HW * 4;
  const unsigned a_row = (unsigned)row0 * (unsigned)HoWo;
  const unsigned b_row = (unsigned)row0 * (unsigned)HW;
  const int lds_word = ((q >> 2) * BM + row0) * 16 + (q & 3) * 4;  // + piece*2*BM*16 + 32*i*16

  const float inv_a = 1.f / operand_scale<P>(p.dy_bound), inv_b = 1.f / operand_scale<P>(p.x_bound);
  float areg[4][2], breg[4][2];
  auto load_regs = [&](int r0) {
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = r0 + 2 * q + e;
      a_voff[e] = OOB;
      b_voff[e] = OOB;
      if (r < r_end) {
        a_voff[e] = (a_row + (unsigned)r) * 4u;
        const int oy = r / p.Wo;
        const int ox = r - oy * p.Wo;
        const int iy = oy * p.stride + ky * p.dil - p.pad;
        const int ix = ox * p.stride + kx * p.dil - p.pad;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) b_voff[e] = (b_row + (unsigned)(iy * p.W + ix)) * 4u;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        areg[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dy_rs, a_voff[e], a_soff0 + 32 * i * HoWo * 4, 0));
        breg[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_rs, b_voff[e], b_soff0 + 32 * i * HW * 4, 0));
      }
  };
  auto store_lds = [&](int buf) {
    unsigned char* a = As + buf * A_BYTES + lds_word;
    unsigned char* b = Bs + buf * B_BYTES + lds_word;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned w[NP];
      split_pair<P>(areg[i][0], areg[i][1], inv_a, w);
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<unsigned*>(a + (pc * 2 * BM + 32 * i) * 16) = w[pc];
      split_pair<P>(breg[i][0], breg[i][1], inv_b, w);
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<unsigned*>(b + (pc * 2 * BN + 32 * i) * 16) = w[pc];
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (r_end - r_begin + BKP - 1) / BKP;
  if (nsteps > 0) {
    load_regs(r_begin);
    store_lds(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const bool more = (s + 1) < nsteps;
    if (more) load_regs(r_begin + (s + 1) * BKP);
    const unsigned char* a_base = As + cur * A_BYTES + (lh * BM + wm * 64 + l31) * 16;
    const unsigned char* b_base = Bs + cur * B_BYTES + (lh * BN + wn * 64 + l31) * 16;
    frag a[NP][WM], b[NP][WN];
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
      for (int i = 0; i < WM; ++i) a[pc][i] = *reinterpret_cast<const frag*>(a_base + (pc * 2 * BM + i * 32) * 16);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[pc][j] = *reinterpret_cast<const frag*>(b_base + (pc * 2 * BN + j * 32) * 16);
    }
#pragma unroll
    for (int tm = 0; tm < P::NTERMS; ++tm)  // term-major: consecutive matrix instructions go to different accumulator tiles
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(a[P::TA[tm]][i], b[P::TB[tm]][j], acc[i][j]);
    if (more) store_lds(cur ^ 1);
    __syncthreads();
  }

  float* out = p.slab + ((size_t)split * T_ + tap) * p.co_p * p.ci_p;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_co * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = tile_ci * BN + wn * 64 + j * 32 + l31;
        out[(size_t)row * p.ci_p + col] = acc[i][j][r];
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Pre-split variant: both operands arrive in the channel-blocked layout their producers wrote
// ([piece NP][N][C/8][H*W][8 x 16 bit]: mcdseg_bn_bwd_apply_cb for dY, mcdseg_bn_apply_cb for X).  One 16-B load is
// 8 channels of one pixel; the MFMA wants 8 pixels of one channel per lane, so a thread loads the same 8-channel
// group at 8 pixels and transposes the 8x8 block of 16-bit values in registers (32 v_perm_b32) -- 0.5 VALU per value
// instead of ~11 for splitting an fp32 value, and no conversion at all.
//
// A K "super-step" is one 8 x 4 pixel tile of the output image (two K=16 MFMA depths).  The four lanes of a quad
// take four consecutive pixels (64 contiguous bytes), and a thread's 8 in-lane k slots are j -> (row j&3,
// column ps + 4*(j>>2)) of the tile; which pixel sits in which k slot is irrelevant as long as dY and X agree, and
// X simply adds the tap's shift to every address.  Everything about a load except the quad-lane column is
// wave-uniform: a wave stages one (operand, piece), the tile and tap offsets go into the SGPR offset, rows that fall
// into the padding get an out-of-range SGPR offset, columns an out-of-range VGPR offset (hardware returns 0).
//
// LDS image per operand [piece NP][k-octet 4][position 128][16 B] with position = c*16 + cg for channel cg*8 + c of
// the tile: transposed rows are written 16 B per lane with consecutive cg (conflict-free) and MFMA lane i reads
// position blockbase + i (conflict-free); the position -> channel permutation is undone when the slab is stored.
// Octet planes are padded by 32 B so the four quad lanes (four planes) of a write hit different banks.
// One LDS stage (16.25 KB per piece pair: 48.75 KB for three pieces, 32.5 KB for two), three workgroups per CU: one
// transposes/writes while the others multiply.  Staging units = (operand, piece): with two pieces each of the four waves
// owns exactly one; with three, the two piece-2 units alternate between the wave pairs from step to step.
struct WgradCbParams {
  const void* x_cb;
  const void* dy_cb;
  float* slab;
  int N, Cin, H, W, Cout, Ho, Wo;
  int KH, KW, stride, pad, dil;
  int co_p, ci_p;
  int tiles_x, tiles_y, tiles_per_chunk, chunks_per_img, splits;
  int x_cb_bytes, dy_cb_bytes;            // ONE piece of each companion (this call's images)
  long long x_piece_stride, dy_piece_stride;  // bytes between the pieces (the companions' own batch: mcdseg_conv_desc.Ncb)
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <class P>
__global__ __launch_bounds__(256, 3) void conv_wgrad_split_cb_kernel(WgradCbParams p) {
  constexpr int BM = 128, BN = 128;
  constexpr int WM = 2, WN = 2, WAVES_N = 2;
  constexpr int NP = P::NP;
  constexpr bool EXTRA = NP == 3;          // a third piece: its two staging units alternate between the wave pairs
  static_assert(NP == 2 || NP == 3, "two or three pieces");
  typedef typename P::frag frag;
  constexpr int PLANE = 130;               // 16-B units per k-octet plane (128 positions + 2 pad)
  constexpr int PIECE = 4 * PLANE;         // four octets per super-step
  constexpr int OP_BYTES = NP * PIECE * 16; // one operand
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ci_tiles = p.ci_p / BN;
  const int co_tiles = p.co_p / BM;
  const int T_ = p.KH * p.KW;
  const int per_split = co_tiles * ci_tiles * T_;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  int rem = slot % per_split;
  const int tap = rem % T_;
  rem /= T_;
  const int tile_ci = rem % ci_tiles;
  const int tile_co = rem / ci_tiles;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int ky = tap / p.KW;
  const int kx = tap - ky * p.KW;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int t_begin = chunk_id * p.tiles_per_chunk;
  int t_end = t_begin + p.tiles_per_chunk;
  if (t_end > ntiles) t_end = ntiles;

  // ---- staging role of this wave: operand (0 = dY rows, 1 = X rows) and piece are wave-uniform
  const int opnd = wave & 1;
  const int pieceA = wave >> 1;  // this wave's unit: piece 0 or 1 (a third piece alternates between the wave pairs)
  const int ps = lane & 3;
  const int cg = lane >> 2;
  // source geometry of the staged operand: X is gathered through the conv geometry, dY is its own output grid
  const int sH = opnd ? p.H : p.Ho;
  const int sW = opnd ? p.W : p.Wo;
  const int sS = opnd ? p.stride : 1;
  const int shy = opnd ? ky * p.dil - p.pad : 0;
  const int shx = opnd ? kx * p.dil - p.pad : 0;
  const int sC8 = (opnd ? p.Cin : p.Cout) >> 3;
  const int ctile = opnd ? tile_ci : tile_co;
  const int sHW = sH * sW;
  // ONE descriptor over all pieces here (a wave of this kernel stages more than one piece; the launcher admits the kernel only
  // while the pieces of the call -- their stride included -- stay below 2 GiB); it starts `bias` bytes below the tensor so that
  // the SGPR offset (tile + tap shift) is never negative
  const int pstride = (int)((opnd ? p.x_piece_stride : p.dy_piece_stride) >> 4);  // 16-B units between pieces
  const int bias = p.pad * 16 + 16;
  const char* sptr = (const char*)(opnd ? p.x_cb : p.dy_cb) - bias;
  const int sbytes = (int)((NP - 1) * (opnd ? p.x_piece_stride : p.dy_piece_stride)) + (opnd ? p.x_cb_bytes : p.dy_cb_bytes) + bias;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sptr, 0, sbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const unsigned vconst = (ctile * 16 + cg) < sC8 ? (unsigned)(cg * sHW + ps * sS) * 16u : OOB;
  const int sbase = (n * sC8 + ctile * 16) * sHW;  // 16-B units
  const int lane_x = ps * sS;

  u32x4 R[EXTRA ? 2 : 1][8];
  auto issue_loads = [&](int tt, int piece, u32x4 (&dst)[8]) {
    const int ty = tt / p.tiles_x;
    const int tx = tt - ty * p.tiles_x;
    unsigned voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ux = (tx * 8 + 4 * h) * sS + shx;  // uniform
      voff[h] = ((unsigned)(ux + lane_x) < (unsigned)sW) ? vconst : OOB;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int h = j >> 2;
      const int iy = (ty * 4 + (j & 3)) * sS + shy;
      const int ux = (tx * 8 + 4 * h) * sS + shx;
      const int soff = ((unsigned)iy < (unsigned)sH) ? (piece * pstride + sbase + iy * sW + ux) * 16 + bias : 0x7FFFFFFF;
      dst[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[h], soff, 0));
    }
  };
  // 8x8 transpose of 16-bit values + eight 16-B LDS writes (one per channel of the group)
  auto transpose_store = [&](int piece, const u32x4 (&src)[8]) {
    unsigned char* base = smem + opnd * OP_BYTES + ((piece * 4 + ps) * PLANE + cg) * 16;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      u32x4 o;
#pragma unroll
      for (int m = 0; m < 4; ++m)
        o[m] = __builtin_amdgcn_perm(src[2 * m + 1][c >> 1], src[2 * m][c >> 1], (c & 1) ? 0x07060302u : 0x05040100u);
      *reinterpret_cast<u32x4*>(base + c * 16 * 16) = o;
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = t_end - t_begin;
  if (nsteps > 0) {
    issue_loads(t_begin, pieceA, R[0]);
    if constexpr (EXTRA)
      if ((wave >> 1) == 0) issue_loads(t_begin, 2, R[EXTRA ? 1 : 0]);
  }
  for (int s = 0; s < nsteps; ++s) {
    const bool extra = EXTRA && (wave >> 1) == (s & 1);  // this wave pair also stages piece 2 of the step
    if (s > 0) __syncthreads();                          // all fragment reads of the previous step are done
    transpose_store(pieceA, R[0]);
    if constexpr (EXTRA)
      if (extra) transpose_store(2, R[EXTRA ? 1 : 0]);
    if (s + 1 < nsteps) {
      issue_loads(t_begin + s + 1, pieceA, R[0]);
      if constexpr (EXTRA)
        if ((wave >> 1) == ((s + 1) & 1)) issue_loads(t_begin + s + 1, 2, R[EXTRA ? 1 : 0]);
    }
    __syncthreads();
    const unsigned char* a_base = smem + (lh * PLANE + wm * 64 + l31) * 16;
    const unsigned char* b_base = smem + OP_BYTES + (lh * PLANE + wn * 64 + l31) * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag a[NP][WM], b[NP][WN];
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
          a[pc][i] = *reinterpret_cast<const frag*>(a_base + (pc * PIECE + 2 * kk * PLANE + i * 32) * 16);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          b[pc][j] = *reinterpret_cast<const frag*>(b_base + (pc * PIECE + 2 * kk * PLANE + j * 32) * 16);
      }
#pragma unroll
      for (int tm = 0; tm < P::NTERMS; ++tm)  // term-major
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[i][j] = P::mfma(a[P::TA[tm]][i], b[P::TB[tm]][j], acc[i][j]);
    }
  }

  // position -> channel: pos = c*16 + cg  <->  channel cg*8 + c
  float* out = p.slab + ((size_t)split * T_ + tap) * p.co_p * p.ci_p;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pa = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int row = tile_co * BM + (pa & 15) * 8 + (pa >> 4);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int pb = wn * 64 + j * 32 + l31;
        const int col = tile_ci * BN + (pb & 15) * 8 + (pb >> 4);
        out[(size_t)row * p.ci_p + col] = acc[i][j][r];
      }
    }
}


// re-define registers filled by inline-assembly LDS reads behind the wait that completes them (no instruction; an ordering edge)
template <class F, int A, int B>
__device__ __forceinline__ void mcd_settle(F (&f)[A][B], int used) {
#pragma unroll
  for (int a = 0; a < A; ++a)
#pragma unroll
    for (int b = 0; b < B; ++b)
      if (a < used) asm volatile("" : "+v"(f[a][b]));
}

// ---------------------------------------------------------------------------------------------------------------
// Pre-split operands, all-DMA variant (two-piece policies): the register transposition above is replaced by gfx950's
// transposing LDS read.  The operands stay in LDS exactly as they sit in memory -- 16-B units of 8 channels x 1 pixel --
// deposited by LDS-DMA (buffer_load ... lds: no staging registers, no v_perm, no ds_write), and the MFMA fragments (8 pixels
// of one channel per lane) are formed by ds_read_b64_tr_b16: per 16-lane group it reads a block of 4 rows (pixels) x 16
// columns (channels) and hands lane i column i -- the transpose is free.
//
// Stage = one 8 x R pixel tile of the output image = 2R quads of 4 consecutive pixels (quad j: row j % R, column half j / R);
// one DMA instruction moves one quad of 128 channels of one (operand, piece): lane = (channel group cg = lane >> 2, pixel
// ps = lane & 3), i.e. 16 runs of 64 contiguous bytes, and lands lane-linear as [cg 16][pixel 4][16 B] = 1 KB.  Everything
// about a DMA except the lane part is wave-uniform (tile, tap shift, padding rows via an out-of-range SGPR offset, padding
// columns / ragged channel groups via an out-of-range VGPR offset: the DMA deposits zeros).  MFMA lane (row l31, k-half lh)
// takes quads 4 kk + 2 lh and 4 kk + 2 lh + 1 of k-block kk; which pixel sits in which k slot is irrelevant as long as dY and
// X agree.  Transposed-read addresses of a 32-lane half cover 256 contiguous bytes: conflict-free.
//
//   <WM = 2, R = 2, NSTAGE = 3>   128 (co) x 128 (ci) tile, 16-pixel stages of 16 KB, three stages: 48 KB and 105 VGPRs, so
//                                 three workgroups share a CU (two 32-pixel stages -- 64 KB, two workgroups -- ran 10 % slower)
//   <WM = 4, R = 2, NSTAGE = 3>   256 (co) x 128 (ci) tile (each wave 128 x 64), 16-pixel stages of 24 KB, three stages: the
//                                 DMAs of tile s+2 fly while tile s multiplies -- the structure of the forward kernel's
//                                 256 x 128 configuration, for the layers with Cout a multiple of 256
//   <WM = 4, R = 2, NSTAGE = 3, TWO = true>   the 128-channel layers, TWO TAPS per workgroup (round 3; on request only, it measured slower --
//                                 DESIGN 4.1c): the operand roles are swapped -- the "A" side is X
//                                 at the shifts of taps 2 tp and 2 tp + 1 (two 128-channel blocks, exactly the big tile's two dY blocks),
//                                 the "B" side the ONE staged dY tile both taps share -- and the workgroup computes the transposed tile
//                                 [tap, ci][co] with the big tile's wave shape: 12 reads per 24 MFMAs, a quarter fewer DMA bytes.  The
//                                 pieces of the cross terms are swapped with the roles, so every product and its order are those of the
//                                 one-tap kernel: bit-identical slabs.
template <class P, int WM, int R, int NSTAGE, bool TWO>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_tr_kernel(WgradCbParams p) {
  static_assert(P::NP == 2, "two-piece policies");
  static_assert(!TWO || WM == 4, "two taps = two 128-row blocks");
  constexpr int WN = 2, WAVES_N = 2;
  constexpr int BM = 64 * WM, BN = 128;
  constexpr int NP = P::NP;
  constexpr int NQD = 2 * R;                 // quads per stage
  constexpr int KK = NQD / 4;                // K = 16 blocks per stage
  constexpr int AB = BM / 128;               // 128-channel blocks of the dY tile
  typedef typename P::frag frag;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  constexpr int QUAD = 1024;                 // bytes one DMA instruction deposits: [cg 16][pixel 4][16 B]
  constexpr int A_UNIT = NQD * AB * QUAD;    // one piece of the dY tile: [quad][cg 16 AB][pixel 4][16 B]
  constexpr int B_UNIT = NQD * QUAD;
  constexpr int STAGE = NP * (A_UNIT + B_UNIT);
  static_assert(NSTAGE * STAGE <= 80 * 1024, "two workgroups per CU");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ci_tiles = p.ci_p / BN;
  const int co_tiles = TWO ? p.co_p / 128 : p.co_p / BM;
  const int T_ = p.KH * p.KW;
  const int TG = TWO ? (T_ + 1) >> 1 : T_;  // tap groups: pairs or single taps
  const int per_split = co_tiles * ci_tiles * TG;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  int rem = slot % per_split;
  const int tg = rem % TG;
  rem /= TG;
  const int tile_ci = rem % ci_tiles;
  const int tile_co = rem / ci_tiles;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int t_begin = chunk_id * p.tiles_per_chunk;
  int t_end = t_begin + p.tiles_per_chunk;
  if (t_end > ntiles) t_end = ntiles;

  // ---- DMA role of this wave: (operand side, piece), wave-uniform.  Side 0 = the "A" rows (AB blocks of 128: dY channel blocks, or
  // with TWO the two taps of X), side 1 = the "B" rows (one block: X at the tap, or with TWO the dY tile)
  const int opnd = wave >> 1;
  const int piece = wave & 1;
  const bool isx = TWO ? (opnd == 0) : (opnd == 1);  // this wave stages X (gathered through the conv geometry) rather than dY
  const int ps = lane & 3;
  const int cg = lane >> 2;
  const int sH = isx ? p.H : p.Ho;
  const int sW = isx ? p.W : p.Wo;
  const int sS = isx ? p.stride : 1;
  int shy[AB], shx[AB];
  bool blk_ok[AB];
#pragma unroll
  for (int b = 0; b < AB; ++b) {
    const int tap_b = TWO ? 2 * tg + b : tg;
    const int ky = tap_b / p.KW, kx = tap_b - ky * p.KW;
    shy[b] = isx ? ky * p.dil - p.pad : 0;
    shx[b] = isx ? kx * p.dil - p.pad : 0;
    blk_ok[b] = tap_b < T_;
  }
  const int sC8 = (isx ? p.Cin : p.Cout) >> 3;
  const int cg0 = isx ? tile_ci * 16 : tile_co * (TWO ? 16 : 16 * AB);  // first channel group of the tile
  const int nblk = opnd ? 1 : AB;                                        // 128-row blocks this wave moves per quad
  const int sHW = sH * sW;
  // the descriptor covers this wave's PIECE and starts `bias` bytes below it so that the SGPR offset (tile + tap shift) is never negative
  const int bias = p.pad * 16 + 16;
  const char* sptr = (const char*)(isx ? p.x_cb : p.dy_cb) + piece * (isx ? p.x_piece_stride : p.dy_piece_stride) - bias;
  const int sbytes = (isx ? p.x_cb_bytes : p.dy_cb_bytes) + bias;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sptr, 0, sbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned vconst[AB];
#pragma unroll
  for (int b = 0; b < AB; ++b) {
    const int cb = TWO ? cg : 16 * b + cg;  // (two taps: both blocks are the same 128 channels)
    vconst[b] = ((cg0 + cb) < sC8 && blk_ok[b]) ? (unsigned)(cb * sHW + ps * sS) * 16u : OOB;
  }
  const int sbase = (n * sC8 + cg0) * sHW;  // 16-B units inside the piece
  const int lane_x = ps * sS;
  unsigned char* const unit_lds = smem + (opnd ? NP * A_UNIT + piece * B_UNIT : piece * A_UNIT);

  auto issue_dma = [&](int tt, int stage) {
    const int ty = tt / p.tiles_x;
    const int tx = tt - ty * p.tiles_x;
    bool colok[2][AB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int b = 0; b < AB; ++b) {
        const int ux = (tx * 8 + 4 * h) * sS + shx[b];  // uniform
        colok[h][b] = (unsigned)(ux + lane_x) < (unsigned)sW;
      }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < NQD; ++j) {  // quad j: row j % R of the tile, column half j / R
      const int h = j / R;
#pragma unroll
      for (int b = 0; b < AB; ++b)
        if (b < nblk) {
          const int iy = (ty * R + (j % R)) * sS + shy[b];
          const int ux = (tx * 8 + 4 * h) * sS + shx[b];
          const int soff = ((unsigned)iy < (unsigned)sH) ? (sbase + iy * sW + ux) * 16 + bias : 0x7FFFFFFF;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs, (__attribute__((address_space(3))) void*)(unit_lds + stage * STAGE + (j * (opnd ? 1 : AB) + b) * QUAD), 16,
              colok[h][b] ? vconst[b] : OOB, soff, 0, 0);
        }
    }
#else
    (void)colok;
    (void)stage;
#endif
  };
  // number of DMA instructions this wave issues per stage (wave-uniform): the counted wait below leaves one stage in flight
  // (a wave whose role is a piece the policy never multiplies -- SplitF16x1's second -- moves nothing)
  const bool dma_on = piece < P::NPU;
  const int dma_per_stage = dma_on ? NQD * nblk : 0;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read address of this lane inside a (piece, quad) image, for the 32-row block i of the wave's rows:
  // 16-lane group g = lane >> 4 covers rows 16 (g & 1) .. +15; lane 4q + pp of the group supplies pixel q, channels 4 pp .. 4 pp + 3
  const int gl = lane & 15;
  const int tq = gl >> 2, tpp = gl & 3;
  const int trow = ((((lane >> 4) & 1) * 2 + (tpp >> 1)) * 4 + tq) * 16 + 8 * (tpp & 1);
  const int a_lane = (wm * 4 * WM) * 64 + trow;  // the wave's first channel group: 4 WM groups per wave
  const int b_lane = (wn * 4 * WN) * 64 + trow;

  auto tr_read = [&](const unsigned char* addr) -> s16x4 {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
#else
    (void)addr;
    return s16x4{};
#endif
  };
  auto load_frag = [&](const unsigned char* unit, int quad_bytes, int quad0, int lane_off, int i) -> frag {
    const s16x4 lo = tr_read(unit + quad0 * quad_bytes + lane_off + i * 256);
    const s16x4 hi = tr_read(unit + (quad0 + 1) * quad_bytes + lane_off + i * 256);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag, v);
  };
  // Two fragment sets: while the matrix instructions of tile s run on one, the transposing reads of tile s+1 fill the other
  // (MCD_WGRAD_PIPE, default).  Without it a wave spends the LDS round trip of 24 reads at the head of every 16-pixel stage
  // before its first MFMA -- measured through the one-term policy: with a third of the matrix work the 256 x 128 kernel only
  // went from 0.98 to 0.78 ms, i.e. the stage's critical path was the read phase, not the matrix pipe.
  //   LDS ring of three stages, tile s in registers: at the top of step s tile s+1 has landed, tile s+2 is in flight; the step
  //   issues the DMAs of tile s+3 into the buffer tile s came from (every wave finished reading it before the last barrier),
  //   reads tile s+1's fragments, multiplies tile s, then waits for tile s+2 and its own reads and meets the barrier.
  static_assert(KK == 1, "one K = 16 block per stage");
  frag fa[2][NP][WM], fb[2][NP][WN];
  // The transposing reads are issued as inline assembly: for a compiler-visible LDS load the wait-count pass puts an
  // s_waitcnt vmcnt(0) in front whenever an LDS-DMA may be pending (it cannot tell that the DMA targets another stage), which
  // would drain the prefetch every step.  Their completion is the lgkmcnt(0) of the step's closing wait; `settle` then
  // re-defines the registers behind that wait so that no consumer can be scheduled ahead of it.
  typedef short s16x8 __attribute__((ext_vector_type(8)));
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
  auto read_frags = [&](int stage, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)(stage * STAGE);
#else
    const unsigned sbase = 0;
#endif
    const unsigned base_a = sbase + (unsigned)((2 * lh) * AB * QUAD + a_lane);
    const unsigned base_b = sbase + (unsigned)(NP * A_UNIT + (2 * lh) * QUAD + b_lane);
    auto frag_of = [&](unsigned base, auto lo_c, auto hi_c) -> frag {
      const s16x4 lo = tr_read_at(base, lo_c);
      const s16x4 hi = tr_read_at(base, hi_c);
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(frag, v);
    };
    // (compile-time offsets: piece, 32-row block, second quad)
#define MCD_A_FRAG(PC, I) \
    if constexpr ((PC) < P::NPU && (I) < WM) \
      fa[SET][PC][I] = frag_of(base_a, std::integral_constant<int, (PC) * A_UNIT + (I) * 256>{}, std::integral_constant<int, (PC) * A_UNIT + (I) * 256 + AB * QUAD>{});
#define MCD_B_FRAG(PC, J) \
    if constexpr ((PC) < P::NPU && (J) < WN) \
      fb[SET][PC][J] = frag_of(base_b, std::integral_constant<int, (PC) * B_UNIT + (J) * 256>{}, std::integral_constant<int, (PC) * B_UNIT + (J) * 256 + QUAD>{});
    MCD_A_FRAG(0, 0) MCD_A_FRAG(0, 1) MCD_A_FRAG(0, 2) MCD_A_FRAG(0, 3)
    MCD_B_FRAG(0, 0) MCD_B_FRAG(0, 1)
    MCD_A_FRAG(1, 0) MCD_A_FRAG(1, 1) MCD_A_FRAG(1, 2) MCD_A_FRAG(1, 3)
    MCD_B_FRAG(1, 0) MCD_B_FRAG(1, 1)
#undef MCD_A_FRAG
#undef MCD_B_FRAG
  };
  auto mfma_frags = [&](auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    // term-major (see conv_gemm_split.hip): consecutive matrix instructions go to different accumulator tiles
#pragma unroll
    for (int tm = 0; tm < P::NTERMS; ++tm)  // (roles swapped with TWO: the A side carries X, so the pieces swap too -- same products, same order)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          acc[i][j] = P::mfma(fa[SET][TWO ? P::TB[tm] : P::TA[tm]][i], fb[SET][TWO ? P::TA[tm] : P::TB[tm]][j], acc[i][j]);
  };
  // wait until at most `left` of this wave's DMAs are outstanding (left is wave-uniform: 0, NQD or NQD * AB), then barrier
  auto wait_barrier = [&](int left) {
    if (left == 0)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (left == NQD)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NQD) : "memory");
    else if (left == NQD * AB)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NQD * AB) : "memory");
    else if (left == 2 * NQD)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NQD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * NQD * AB) : "memory");
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  static_assert(NSTAGE == 3, "three LDS stages");

  const int nsteps = t_end - t_begin;
  // prologue: tiles 0, 1, 2 on their way; tile 0 into registers; tile 1 landed
  if (dma_on) {
    if (nsteps > 0) issue_dma(t_begin, 0);
    if (nsteps > 1) issue_dma(t_begin + 1, 1);
    if (nsteps > 2) issue_dma(t_begin + 2, 2);
  }
  wait_barrier(nsteps > 2 ? 2 * dma_per_stage : (nsteps > 1 ? dma_per_stage : 0));  // tile 0 landed
  if (nsteps > 0) read_frags(0, S0{});
  wait_barrier(nsteps > 2 ? dma_per_stage : 0);  // tile 1 landed, tile 0 read by every wave
  auto step = [&](int s, auto cur_c, auto nxt_c) {
    const bool more3 = s + 3 < nsteps;
    // (program order matters to the compiler's wait-count insertion: LDS reads placed behind an LDS-DMA issued in the same
    // iteration get an s_waitcnt vmcnt(0) in front -- it cannot tell that the DMA targets another buffer -- which would drain the
    // whole prefetch; reads first, then the DMAs, then the matrix instructions)
    mcd_settle(fa[decltype(cur_c)::value], P::NPU);
    mcd_settle(fb[decltype(cur_c)::value], P::NPU);
    if (s + 1 < nsteps) read_frags((s + 1) % 3, nxt_c);
    if (more3 && dma_on) issue_dma(t_begin + s + 3, s % 3);
    mfma_frags(cur_c);
    // tile s+2 must have landed (the DMAs of tile s+3, younger, stay in flight), this wave's reads of tile s+1 are complete
    wait_barrier(more3 ? dma_per_stage : 0);
  };
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    step(s, S0{}, S1{});
    step(s + 1, S1{}, S0{});
  }
  if (s < nsteps) step(s, S0{}, S1{});

  if constexpr (TWO) {  // transposed tile: rows = (tap wm, ci), columns = co -- turned through a wave-private LDS image so that the
    // slab, laid out [co][ci] like everyone else's, is written in 256-byte runs (the loop's last barrier released the stages)
    const int my_tap = 2 * tg + wm;
    constexpr int PITCH = 68;  // floats per co row of the image: 16-byte aligned rows, 64 ci + padding
    static_assert(4 * 64 * PITCH * 4 <= NSTAGE * STAGE, "four wave images fit the stage ring");
    float* img = reinterpret_cast<float*>(smem) + wave * 64 * PITCH;
    float* out = p.slab + ((size_t)split * T_ + (my_tap < T_ ? my_tap : 0)) * p.co_p * p.ci_p;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {  // 64 ci at a time
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci_l = i2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
          for (int j = 0; j < WN; ++j) img[(j * 32 + l31) * PITCH + ci_l] = acc[2 * hh + i2][j][r];
        }
      // (one wave: its LDS writes and reads are ordered by the LDS queue)
      if (my_tap < T_) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int co_l = q * 4 + (lane >> 4), c4 = (lane & 15) * 4;
          const float4 v = *reinterpret_cast<const float4*>(img + co_l * PITCH + c4);
          *reinterpret_cast<float4*>(out + (size_t)(tile_co * 128 + wn * 64 + co_l) * p.ci_p + tile_ci * BN + hh * 64 + c4) = v;
        }
      }
    }
    return;
  }
  float* out = p.slab + ((size_t)split * T_ + tg) * p.co_p * p.ci_p;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile_co * BM + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = tile_ci * BN + wn * 64 + j * 32 + l31;
        out[(size_t)row * p.ci_p + col] = acc[i][j][r];
      }
    }
}

}  // namespace

// launched by wgrad_impl (conv_wgrad.hip) when the 128x128 plan applies and a split arithmetic is requested
int mcdseg_internal_wgrad_split_launch(const mcdseg_conv_desc* d, int math, const float* x, const float* x_bound, const float* dy,
                                       const float* dy_bound, float* slab, int co_p, int ci_p, int chunk, int chunks_per_img, int splits,
                                       hipStream_t st) {
  WgradSplitParams p;
  p.x = x; p.dy = dy; p.slab = slab;
  p.x_bound = x_bound; p.dy_bound = dy_bound;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_p = co_p; p.ci_p = ci_p; p.chunk = chunk; p.chunks_per_img = chunks_per_img; p.splits = splits;
  p.x_bytes = (int)((int64_t)d->N * d->Cin * d->H * d->W * 4);
  p.dy_bytes = (int)((int64_t)d->N * d->Cout * d->Ho * d->Wo * 4);
  const int64_t per_split = (int64_t)(co_p / 128) * (ci_p / 128) * d->KH * d->KW;
  const int64_t nwg = 8 * ceil_div64(splits, 8) * per_split;
  if (nwg >= (1ll << 31)) {
    mcdseg_set_error("conv_wgrad_split: grid too large");
    return -22;
  }
  if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(conv_wgrad_split_kernel<SplitF16x3>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (math == MCDSEG_MATH_F16X1)
    hipLaunchKernelGGL(conv_wgrad_split_kernel<SplitF16x1>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_split_kernel<SplitBf16x6>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_split");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// 64-channel layers (min(Cin, Cout) in (32, 64]: the plan's 64 x 64 tiles).  A 64 x 64 output tile alone leaves the matrix pipe
// starved (6 MFMAs per wave and K-block), so a workgroup computes the tiles of TWO taps from one staged dY tile: output
// 64 (co) x 128 (tap pair x 64 ci), wave (wm, wn) = 32 co x the 64 ci of tap 2 tp + wn.  Same staging as
// conv_wgrad_split_tr_kernel -- LDS-DMA of 16-B units, transposing LDS reads -- except that a DMA instruction's upper half-wave
// carries something else than its lower half: for dY (8 channel groups) the NEXT quad, for X the SECOND tap's shift; the
// address (and the padding test) is therefore per lane.
template <class P>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_tr64_kernel(WgradCbParams p) {
  static_assert(P::NP == 2, "two-piece policies");
  constexpr int R = 4, NSTAGE = 2;
  constexpr int BM = 64, BNC = 64;           // co tile, ci tile (x 2 taps = 128 columns)
  constexpr int NP = P::NP;
  constexpr int NQD = 2 * R, KK = NQD / 4;
  typedef typename P::frag frag;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  constexpr int AQ = 512, BQ = 1024;         // bytes per quad image: [cg 8][pixel 4][16 B] / [tap 2][cg 8][pixel 4][16 B]
  constexpr int A_UNIT = NQD * AQ, B_UNIT = NQD * BQ;
  constexpr int STAGE = NP * (A_UNIT + B_UNIT);  // 24 KB
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int ci_tiles = p.ci_p / BNC;
  const int co_tiles = p.co_p / BM;
  const int T_ = p.KH * p.KW;
  const int TP = (T_ + 1) >> 1;  // tap pairs
  const int per_split = co_tiles * ci_tiles * TP;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int split = (slot / per_split) * 8 + xcd;
  if (split >= p.splits) return;
  int rem = slot % per_split;
  const int tp = rem % TP;
  rem /= TP;
  const int tile_ci = rem % ci_tiles;
  const int tile_co = rem / ci_tiles;
  const int n = split / p.chunks_per_img;
  const int chunk_id = split - n * p.chunks_per_img;
  const int ntiles = p.tiles_x * p.tiles_y;
  const int t_begin = chunk_id * p.tiles_per_chunk;
  int t_end = t_begin + p.tiles_per_chunk;
  if (t_end > ntiles) t_end = ntiles;

  // ---- DMA role of this wave: (operand, piece); per lane: half-wave selector, channel group, pixel
  const int opnd = wave >> 1;   // 0 = dY, 1 = X
  const int piece = wave & 1;
  const int ps = lane & 3;
  const int cg = (lane >> 2) & 7;
  const int hw_sel = lane >> 5;  // dY: second quad of the pair; X: second tap of the pair
  const int sH = opnd ? p.H : p.Ho;
  const int sW = opnd ? p.W : p.Wo;
  const int sS = opnd ? p.stride : 1;
  const int tap = 2 * tp + (opnd ? hw_sel : 0);
  const bool tap_ok = tap < T_;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int shy = opnd ? ky * p.dil - p.pad : 0;  // per lane for X
  const int shx = opnd ? kx * p.dil - p.pad : 0;
  const int sC8 = (opnd ? p.Cin : p.Cout) >> 3;
  const int cg0 = (opnd ? tile_ci : tile_co) * 8;
  const int sHW = sH * sW;
  const char* sptr = (const char*)(opnd ? p.x_cb : p.dy_cb) + piece * (opnd ? p.x_piece_stride : p.dy_piece_stride);  // this wave's piece
  const int sbytes = opnd ? p.x_cb_bytes : p.dy_cb_bytes;
  // (LDS-DMA the compiler does not see -- mcd_hidden_dma: through the builtin, the transposed reads of fragments kk >= 1 of the CURRENT
  // stage wait vmcnt(0) for the next stage's DMA issued a moment before them: no look-ahead at all)
  const mcd_i32x4 rs = mcd_raw_rsrc(sptr, sbytes);
  constexpr unsigned OOB = 0x80000000u;
  const bool cg_ok = (cg0 + cg) < sC8 && tap_ok;
  const int lbase = (n * sC8 + cg0 + cg) * sHW;  // 16-B units inside the piece, per lane
  unsigned char* const unit_lds = smem + (opnd ? NP * A_UNIT + piece * B_UNIT : piece * A_UNIT);
  const unsigned unit_lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)unit_lds;

  auto issue_dma = [&](int tt, int stage) {
    const int ty = tt / p.tiles_x;
    const int tx = tt - ty * p.tiles_x;
#if defined(__HIP_DEVICE_COMPILE__)
    if (opnd == 0) {  // dY: instruction i carries quads 2 i (lower half-wave) and 2 i + 1 (upper)
#pragma unroll
      for (int i = 0; i < NQD / 2; ++i) {
        const int j = 2 * i + hw_sel;
        const int iy = ty * R + (j % R);
        const int ix = tx * 8 + 4 * (j / R) + ps;
        const bool ok = cg_ok && iy < sH && ix < sW;
        const unsigned voff = ok ? (unsigned)(lbase + iy * sW + ix) * 16u : OOB;
        mcd_hidden_dma<16>(rs, __builtin_amdgcn_readfirstlane(unit_lds_addr + (unsigned)(stage * STAGE + i * 2 * AQ)), voff);
      }
    } else {  // X: instruction j carries quad j of tap 2 tp (lower half-wave) and of tap 2 tp + 1 (upper)
#pragma unroll
      for (int j = 0; j < NQD; ++j) {
        const int iy = (ty * R + (j % R)) * sS + shy;
        const int ix = (tx * 8 + 4 * (j / R) + ps) * sS + shx;
        const bool ok = cg_ok && (unsigned)iy < (unsigned)sH && (unsigned)ix < (unsigned)sW;
        const unsigned voff = ok ? (unsigned)(lbase + iy * sW + ix) * 16u : OOB;
        mcd_hidden_dma<16>(rs, __builtin_amdgcn_readfirstlane(unit_lds_addr + (unsigned)(stage * STAGE + j * BQ)), voff);
      }
    }
#else
    (void)ty;
    (void)tx;
    (void)stage;
#endif
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int gl = lane & 15;
  const int tq = gl >> 2, tpp = gl & 3;
  const int trow = ((((lane >> 4) & 1) * 2 + (tpp >> 1)) * 4 + tq) * 16 + 8 * (tpp & 1);
  const int a_lane = (wm * 4) * 64 + trow;  // 32 co = 4 channel groups of the 8 in a dY quad
  const int b_lane = (wn * 8) * 64 + trow;  // tap wn: its 8 channel groups of the 16 in an X quad

  auto tr_read = [&](const unsigned char* addr) -> s16x4 {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
#else
    (void)addr;
    return s16x4{};
#endif
  };
  auto load_frag = [&](const unsigned char* unit, int quad_bytes, int quad0, int lane_off) -> frag {
    const s16x4 lo = tr_read(unit + quad0 * quad_bytes + lane_off);
    const s16x4 hi = tr_read(unit + (quad0 + 1) * quad_bytes + lane_off);
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag, v);
  };
  frag fa[NP], fb[NP][2];
  auto read_frags = [&](const unsigned char* st, int kk) {
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      fa[pc] = load_frag(st + pc * A_UNIT, AQ, 4 * kk + 2 * lh, a_lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[pc][j] = load_frag(st + NP * A_UNIT + pc * B_UNIT, BQ, 4 * kk + 2 * lh, b_lane + j * 256);
    }
  };
  auto mfma_frags = [&]() {
#pragma unroll
    for (int tm = 0; tm < P::NTERMS; ++tm)  // term-major: the two tiles alternate
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = P::mfma(fa[P::TA[tm]], fb[P::TB[tm]][j], acc[j]);
  };

  const int nsteps = t_end - t_begin;
  if (nsteps > 0) issue_dma(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    const unsigned char* st = smem + cur * STAGE;
    read_frags(st, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < nsteps) issue_dma(t_begin + s + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags();
#pragma unroll
    for (int kk = 1; kk < KK; ++kk) {
      read_frags(st, kk);
      mfma_frags();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  const int my_tap = 2 * tp + wn;
  if (my_tap >= T_) return;
  float* out = p.slab + ((size_t)split * T_ + my_tap) * p.co_p * p.ci_p;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tile_co * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = tile_ci * BNC + j * 32 + l31;
      out[(size_t)row * p.ci_p + col] = acc[j][r];
    }
  }
}

// which pre-split kernel runs: 0 = register-transposing (three-piece policy, or MCDSEG_WGRAD_TR=0), 1 = transposed-read
// 128x128 tiles, 2 = transposed-read 256x128 tiles
int mcdseg_internal_wgrad_cb_variant(const mcdseg_conv_desc* d, int math, int co_p, int ci_p, int splits) {
  // development / test knobs (options.h), read per call so that a test can run one problem on
  // both tile shapes: WGRAD_TR=0 = register-transposing kernel for the two-piece policy too, MCDSEG_WGRAD_BIG=0 = 128 x 128
  // tiles only
  const bool use_tr = mcd_opt(MCD_OPT_WGRAD_TR) != 0;
  const bool use_big = mcd_opt(MCD_OPT_WGRAD_BIG) != 0;
  if (!(mcd_storage_math(math) == MCDSEG_MATH_F16X3 && use_tr)) return 0;
  // 256 x 128 tiles (16-pixel stages) for the layers whose padded Cout is a multiple of 256 -- the plan was made for 128-row
  // tiles, so the number of workgroups halves; taken only while that still fills the chip twice over
  const bool big = use_big && (co_p % 256) == 0 && (int64_t)(co_p / 256) * (ci_p / 128) * d->KH * d->KW * splits >= 1024;
  if (big) return 2;
  // the remaining 128-row layers with more than one tap: two taps per workgroup sharing the staged dY tile -- built, bit-identical,
  // and SLOWER at the benchmark's sizes (256 -> 256 at N = 16: 0.309-0.349 ms against 0.286: an odd tap count idles a tenth of the waves,
  // half as many workgroups, a transposed epilogue), so it runs only on request (MCDSEG_WGRAD_TWOTAP=1; tests)
  return (d->KH * d->KW > 1 && mcd_opt(MCD_OPT_WGRAD_TWOTAP) != 0) ? 3 : 1;
}

// pre-split operands (see conv_wgrad_split_cb_kernel); chunks_per_img / splits come from the shared plan, the pixel range
// of a chunk is expressed in 8x4 output tiles
int mcdseg_internal_wgrad_split_cb_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb, float* slab,
                                          int co_p, int ci_p, int chunks_per_img, int splits, hipStream_t st) {
  WgradCbParams p;
  p.x_cb = x_cb; p.dy_cb = dy_cb; p.slab = slab;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_p = co_p; p.ci_p = ci_p; p.chunks_per_img = chunks_per_img; p.splits = splits;
  const int variant = mcdseg_internal_wgrad_cb_variant(d, math, co_p, ci_p, splits);
  const bool tr = variant >= 1, big = variant == 2, two = variant == 3;
  {
    const int64_t npc = mcd_math_pieces(math), nb = d->Ncb ? d->Ncb : d->N;
    const int64_t xall = (npc - 1) * nb * d->Cin * d->H * d->W * 2 + (int64_t)d->N * d->Cin * d->H * d->W * 2;
    const int64_t yall = (npc - 1) * nb * d->Cout * d->Ho * d->Wo * 2 + (int64_t)d->N * d->Cout * d->Ho * d->Wo * 2;
    if (!tr && (xall + 4096 >= (1ll << 31) || yall + 4096 >= (1ll << 31))) {
      mcdseg_set_error("conv_wgrad_split: the register-transposing kernel needs all pieces of an operand below 2 GiB; split the batch further");
      return -22;
    }
  }
  const int rows = tr ? 2 : 4;  // pixel rows of a stage tile (the transposing-read kernels run 16-pixel stages)
  p.tiles_x = ceil_div(d->Wo, 8);
  p.tiles_y = ceil_div(d->Ho, rows);
  p.tiles_per_chunk = ceil_div(p.tiles_x * p.tiles_y, chunks_per_img);
  // per PIECE of this call's images (the pieces of a batch slice are not adjacent: mcdseg_conv_desc.Ncb)
  const int64_t xb = (int64_t)d->N * d->Cin * d->H * d->W * 2, yb = (int64_t)d->N * d->Cout * d->Ho * d->Wo * 2;
  if ((d->Cin & 7) || (d->Cout & 7) || xb + 4096 >= (1ll << 31) || yb + 4096 >= (1ll << 31) || d->pad > 128 || (d->Ncb != 0 && d->Ncb < d->N)) {
    mcdseg_set_error("conv_wgrad_split: pre-split operands need channel counts divisible by 8, < 2 GiB per operand piece and Ncb >= N");
    return -22;
  }
  p.x_cb_bytes = (int)xb;
  p.dy_cb_bytes = (int)yb;
  p.x_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cin * d->H * d->W * 2;  // (F16X1 reads piece 0 only)
  p.dy_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cout * d->Ho * d->Wo * 2;
  const int64_t per_split = (int64_t)(co_p / (big ? 256 : 128)) * (ci_p / 128) * (two ? (d->KH * d->KW + 1) / 2 : d->KH * d->KW);
  const int64_t nwg = 8 * ceil_div64(splits, 8) * per_split;
  if (nwg >= (1ll << 31)) {
    mcdseg_set_error("conv_wgrad_split: grid too large");
    return -22;
  }
  const bool one = math == MCDSEG_MATH_F16X1;  // the same staging, one term
  if (big && one)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x1, 4, 2, 3, false>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (big)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x3, 4, 2, 3, false>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (two && one)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x1, 4, 2, 3, true>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (two)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x3, 4, 2, 3, true>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (tr && one)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x1, 2, 2, 3, false>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (tr)
    hipLaunchKernelGGL((conv_wgrad_split_tr_kernel<SplitF16x3, 2, 2, 3, false>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (one)
    hipLaunchKernelGGL(conv_wgrad_split_cb_kernel<SplitF16x1>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else if (math == MCDSEG_MATH_F16X3)
    hipLaunchKernelGGL(conv_wgrad_split_cb_kernel<SplitF16x3>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_split_cb_kernel<SplitBf16x6>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_split_cb");
  return 0;
}

// the plan's 64 x 64 tiles from both companions (two-piece policies): see conv_wgrad_split_tr64_kernel
int mcdseg_internal_wgrad_split_tr64_launch(const mcdseg_conv_desc* d, int math, const void* x_cb, const void* dy_cb, float* slab, int co_p,
                                            int ci_p, int chunks_per_img, int splits, hipStream_t st) {
  WgradCbParams p;
  p.x_cb = x_cb; p.dy_cb = dy_cb; p.slab = slab;
  p.N = d->N; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.Ho = d->Ho; p.Wo = d->Wo;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.co_p = co_p; p.ci_p = ci_p; p.chunks_per_img = chunks_per_img; p.splits = splits;
  p.tiles_x = ceil_div(d->Wo, 8);
  p.tiles_y = ceil_div(d->Ho, 4);
  p.tiles_per_chunk = ceil_div(p.tiles_x * p.tiles_y, chunks_per_img);
  const int64_t xb = (int64_t)d->N * d->Cin * d->H * d->W * 2, yb = (int64_t)d->N * d->Cout * d->Ho * d->Wo * 2;  // per piece
  if (mcd_storage_math(math) != MCDSEG_MATH_F16X3 || (d->Cin & 7) || (d->Cout & 7) || (co_p & 63) || (ci_p & 63) || xb >= (1ll << 31) || yb >= (1ll << 31) ||
      (d->Ncb != 0 && d->Ncb < d->N)) {
    mcdseg_set_error("conv_wgrad_split: the 64-tile pre-split plan needs f16x3, channel counts divisible by 8, < 2 GiB per operand piece and Ncb >= N");
    return -22;
  }
  p.x_cb_bytes = (int)xb;
  p.dy_cb_bytes = (int)yb;
  p.x_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cin * d->H * d->W * 2;  // (F16X1 reads piece 0 only)
  p.dy_piece_stride = math == MCDSEG_MATH_F16X1 ? 0 : (long long)(d->Ncb ? d->Ncb : d->N) * d->Cout * d->Ho * d->Wo * 2;
  const int64_t per_split = (int64_t)(co_p / 64) * (ci_p / 64) * ((d->KH * d->KW + 1) / 2);
  const int64_t nwg = 8 * ceil_div64(splits, 8) * per_split;
  if (nwg >= (1ll << 31)) {
    mcdseg_set_error("conv_wgrad_split: grid too large");
    return -22;
  }
  if (math == MCDSEG_MATH_F16X1)
    hipLaunchKernelGGL(conv_wgrad_split_tr64_kernel<SplitF16x1>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(conv_wgrad_split_tr64_kernel<SplitF16x3>, dim3((unsigned)nwg), dim3(256), 0, st, p);
  MCD_LAUNCH_CHECK("conv_wgrad_split_tr64");
  return 0;
}
