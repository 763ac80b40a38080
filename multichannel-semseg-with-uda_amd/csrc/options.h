// Plan / development options of libmcdseg: ONE table, set through the ABI (mcdseg_set_option), never read from the process environment.
// The host side (mcdseg/_lib.py) translates the MCDSEG_<NAME> environment variables into mcdseg_set_option calls once, when it loads the
// library; tests set options through the same call.  Values are plain integers; a kernel launcher reads the current value at every call.
#pragma once
#include <stdint.h>

#define MCD_OPTIONS(X)                                                                                                       \
  X(BN_STATS_ONE, 1024)       /* BatchNorm statistics in ONE launch up to this many partial rows (0 = never) */              \
  X(BN_REVERSE, 1)            /* the four-pixel BatchNorm apply kernels walk their tensors back to front */                  \
  X(BN_V4, 1)                 /* four-pixel forms of the companion-writing BatchNorm apply kernels */                        \
  X(BIGTILE_MIN_SLOTS, 1024)  /* tile slots from which the 4-wave 256 x 128 tile is preferred */                             \
  X(WIDETILE_MIN_SLOTS, 1024) /* ... and the 128 x 256 one */                                                                \
  X(DGRAD_INTERLEAVE, 1)      /* stride-2 data gradient: 0 classes in turn, 1 interleaved (+ row classes for 1x1), 2 row classes */ \
  X(DGRAD_ADD_LDS, 1)         /* data gradient + addend: the addend's tile staged through LDS (ping-pong kernel) */          \
  X(PACK_BLOCKS, 192)         /* workgroups per (convolution, image) of the table-driven weight pack */                      \
  X(PP_MIN_ROUNDS, 2)         /* fewest whole rounds of 256 x 256 tiles for which the ping-pong kernels take a convolution */ \
  X(PP_CUS, 0)                /* plan for this many compute units (0 = the device's): a small batch gets a large batch's plan */ \
  X(PINGPONG, 3)              /* 0 4-wave tiles only; 1 / 3 whole rounds of 256 x 256 + rest on 4-wave / 256 x 128; 2 256 x 128 for all; 4 wide tile forced */ \
  X(PP_WIDE_FILL, 80)         /* per cent of its rounds the 320-pixel tile must fill (> 100 = never) */                      \
  X(PP_WIDE128, 1)            /* the 128 x 320 variant */                                                                    \
  X(PP_DEEP, 1)               /* one-term arithmetic: two K-steps per barrier interval of the ping-pong kernels (SplitF16x1D) */ \
  X(THIN_WINDOW, 1)           /* LDS-window kernels of the thin 3x3 layers */                                                \
  X(WGRAD_WGS, 1024)          /* target workgroup count of the 128 / 64-tile weight-gradient plans */                        \
  X(WGRAD_THIN_TR, 1)         /* thin layers' window weight gradient */                                                      \
  X(WGRAD_TR64, 1)            /* 64 x 64 weight-gradient plan from both companions */                                        \
  X(WGRAD_TR, 1)              /* transposed-read weight-gradient kernels (0 = register-transposing) */                       \
  X(WGRAD_BIG, 1)             /* 256 x 128 weight-gradient tiles */                                                          \
  X(WGRAD_TWOTAP, 0)          /* two taps per workgroup (slower at the benchmark's sizes; tests) */                          \
  X(WGRAD_PP_CUS, 0)          /* weight gradient planned for this many CUs (0 = PP_CUS, else the device's) */                \
  X(WGRAD_PP, 2)              /* ping-pong weight gradient: 0 off, 1 stream-K, 2 the slab plan */                            \
  X(WGRAD_PP3, 1)             /* row-of-taps ping-pong weight gradient */                                                    \
  X(WGRAD_PP_DEEP, 1)         /* one-term arithmetic: six logical LDS stages in the ping-pong weight gradient (SplitF16x1D) */ \
  X(UP8_LOSS_DMA, 1)          /* fused up-sampler + loss: the LDS-DMA kernel (0 = register-staged) */                        \
  X(UP8_BAND_ROWS, -1)        /* up-sampler backward: 0 = two separate kernels, n = rows per band, -1 = the plan's own */

enum McdOpt {
#define X(name, def) MCD_OPT_##name,
  MCD_OPTIONS(X)
#undef X
      MCD_OPT_COUNT
};

int64_t mcd_opt(McdOpt which);
