// Kernel parameters of the split-arithmetic implicit-GEMM convolutions: conv_gemm_split.hip (4-wave tiles) and
// conv_gemm_split_pp.hip (8-wave ping-pong tile).  Host code of conv_gemm_split.hip fills one block and hands it to both.
#pragma once
#include "split.h"

struct ConvSplitParams {
  const float* src;
  const void* src_cb;  // optional pre-split operand: [piece NP][N][Cs/8][Hs*Ws][8 x 16 bit] (PRESPLIT kernels)
  const void* wp;      // packed 16-bit weight image
  const float* src_bound;  // SplitF16x3: device scalars, upper bounds of |src| and |w| (scale = mcd_scale_of_bound)
  const float* w_bound;
  const float* bias;
  float* dst;
  float* stats;
  const float* ep_scale;
  const float* ep_shift;
  const float* ep_res;
  int ep_relu;
  int N;
  int Cs, Hs, Ws;
  int M, Hd, Wd;
  int Mp, Kp;
  int KH, KW, stride, pad, dil;
  int P;
  int src_bytes, wp_bytes, cb_bytes;  // cb_bytes: ONE piece of the companion (this call's images)
  long long cb_piece_stride;          // bytes from piece p to piece p + 1 (the companion's own batch may be larger: mcdseg_conv_desc.Ncb)
  // stride-2 dgrad in parity classes: output pixels (y%2, x%2) = class receive only the taps of matching parity, so a
  // tile holds pixels of ONE class and its K loop visits that class's taps only (1, 2, 2, 4 of 9 for a 3x3 kernel)
  int sub;           // 1 when the class ordering is active
  int cls_tile0[5];  // first tile of class c (c = 2*ry + rx), cls_tile0[4] = number of tiles
  // A launch may cover a WINDOW of the pixel tiles (the host gives whole rounds of 256 x 256 tiles to the ping-pong kernel and
  // the rest to the 4-wave tiles): this launch's tiles are tile_n0 .. tile_n1 - 1 of the launch's own pixel-tile size.  Outputs
  // and BatchNorm partial rows (one per 64 pixels, numbered over the whole tensor) are those of a single launch.
  int tile_n0, tile_n1;
  // data gradient + addend (ep_res with DGRAD): the ping-pong kernel may stage the addend's tile through LDS (free after the K loop)
  // by LDS-DMA instead of loading it value by value into the few registers its accumulators leave: set by the host when a pixel
  // quad never straddles two images and the tensor is 16-byte aligned
  int ep_res_lds;
  unsigned ep_res_bytes;
  // 16-bit channel-blocked output (SplitF16x1 only; round 6: BASELINE config 5's 2-byte activation storage).  When dst16 is set the
  // kernel writes NO fp32 dst: [N][M/8][Hd*Wd][8 x 16 bit] -- the companions' unit layout, one 16-byte unit per pixel and 8-channel
  // group -- as fp16 of z / scale(dst_bound) in the forward pass (the kernel itself writes dst_bound = taps * Cs * src_bound * w_bound,
  // an upper bound of |z| known before any z exists) and as bf16 in the data gradient (gradients span more binades than a per-tensor
  // scale covers, and need no bound); ep_res16: the data gradient's addend in the same bf16 layout.
  void* dst16;
  float* dst_bound;
  const void* ep_res16;
};

typedef _Float16 mcd_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mcd_bf16x4 __attribute__((ext_vector_type(4)));
