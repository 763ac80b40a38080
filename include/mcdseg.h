/*
 * mcdseg.h -- C ABI of libmcdseg.so, the MI355X (gfx950) kernels behind the MCD hot path.
 *
 * The reference (LittleWat/multichannel-semseg-with-uda) has no FFI layer: its hot path is a chain
 * of stock ATen operators launched from Python.  Each entry point below replaces one such operator
 * call site (cited as file:line of the reference); the Python host side in
 * multichannel-semseg-with-uda_amd/mcdseg/ binds them with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *   - tensors are fp32, NCHW, contiguous; labels are int64;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     never synchronises; scratch space is passed in by the caller (sizes from the *_bytes /
 *     *_count helpers);
 *   - return value 0 = launched; <0 = rejected (bad argument, unsupported shape, launch error) and
 *     mcdseg_last_error() holds the reason (thread-local string).
 */
#ifndef MCDSEG_H
#define MCDSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the declarations below are exported */
#pragma GCC visibility push(default)

#define MCDSEG_VERSION 101

int mcdseg_version(void);
const char* mcdseg_last_error(void);

/* Data parallelism (mcdseg_allreduce(buf, count, comm, stream) over RCCL): what the reference gets from
 * torch.nn.DataParallel's gradient reduction (/root/reference/models/model_util.py:283-284).  A communicator of the library's own:
 * rank 0 draws an id (mcdseg_comm_unique_id: MCDSEG_COMM_ID_BYTES bytes) and hands it to the other ranks by any means (the host side:
 * a broadcast over the process group it already has), every rank calls mcdseg_comm_init (collective); mcdseg_allreduce sums `count`
 * fp32 values in place over the ranks, enqueued on `stream` -- asynchronous, no host synchronisation, like every other entry point.
 * RCCL is bound at run time (the librccl the process has loaded already -- PyTorch's -- else librccl.so.1 on the loader's path):
 * the library does not link it, and returns -38 (ENOSYS) where there is none.  csrc/comm.hip. */
#define MCDSEG_COMM_ID_BYTES 128
int mcdseg_comm_unique_id(void* id128);
int mcdseg_comm_init(void** comm, int32_t nranks, const void* id128, int32_t rank);
int mcdseg_comm_destroy(void* comm);
int mcdseg_allreduce(float* buf, int64_t count, void* comm, void* stream);

/* Plan / development options (tile choices, launch plans, kernel forms: csrc/options.h lists them with their defaults).  The library
 * never reads the process environment: whoever wants another plan says so through this call.  The table is process-wide (a backward
 * pass runs on other threads than the forward pass that built its graph) and read at every launch; a name may carry the "MCDSEG_"
 * prefix of the environment variable the Python host side translates (mcdseg/_lib.py).  Unknown names are rejected. */
int32_t mcdseg_option_count(void);
const char* mcdseg_option_name(int32_t index);
int mcdseg_set_option(const char* name, int64_t value);
int mcdseg_get_option(const char* name, int64_t* value, int64_t* default_value);

/* ------------------------------------------------------------------------------------------------
 * Convolution  (nn.Conv2d call sites: models/drn.py:21-23,127,177,199; models/dilated_fcn.py:227)
 * groups = 1, square kernel, symmetric stride / padding / dilation.
 * ---------------------------------------------------------------------------------------------- */
typedef struct mcdseg_conv_desc {
  int32_t N, Cin, H, W;      /* input  [N,Cin,H,W]                      */
  int32_t Cout, KH, KW;      /* weight [Cout,Cin,KH,KW]                 */
  int32_t stride, pad, dil;
  int32_t Ho, Wo;            /* output [N,Cout,Ho,Wo]                   */
  int32_t Ncb;               /* split operators: batch size of the tensor the pre-split companions (x_cb / dy_cb) were WRITTEN for,
                              * when this call covers only N of its images (a launch addresses < 2 GiB per operand piece, so larger
                              * tensors are processed in slices along N): the companion layout is [piece][Ncb][C/8][H*W][8], the
                              * pointer passed is that of the slice's first image in piece 0, and piece p of the slice lies
                              * p * Ncb * (C/8) * H*W * 16 bytes behind it.  0 = N (the companion belongs to exactly this batch). */
} mcdseg_conv_desc;

/* Padded GEMM dims of the packed weight images: fprop image is [KH*KW][Kp_f][Mp_f] with M = Cout,
 * K = Cin; dgrad image is [KH*KW][Kp_d][Mp_d] with M = Cin, K = Cout. */
int mcdseg_conv_packed_dims(const mcdseg_conv_desc* d, int32_t* Mp_f, int32_t* Kp_f, int32_t* Mp_d, int32_t* Kp_d);
/* w [Cout,Cin,KH,KW] -> fprop image and/or dgrad image (either destination may be NULL). */
int mcdseg_conv_pack_weights(const mcdseg_conv_desc* d, const float* w, float* wp_fprop, float* wp_dgrad, void* stream);

/* Number of per-channel partial-statistics rows conv_fprop writes (each row is 3*Mp_f floats:
 * count, mean, M2 of one wave's slice of pixels). */
int64_t mcdseg_conv_stat_rows(const mcdseg_conv_desc* d);
/* y = conv(x, w) (+ bias).  If stat_partials != NULL also emits the train-mode BatchNorm partials of y
 * (fused epilogue; nn.BatchNorm2d call sites models/drn.py:34,38,129,179,202). */
int mcdseg_conv_fprop(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* bias,
                      float* y, float* stat_partials, void* stream);
/* Inference form (eval-mode BatchNorm folded into the epilogue, adapt_tester.py:87-104):
 * y = act(scale[c] * conv(x, w) + shift[c] (+ residual)), act = ReLU if relu != 0.  No bn_apply pass. */
int mcdseg_conv_fprop_affine(const mcdseg_conv_desc* d, const float* x, const float* wp_fprop, const float* scale,
                             const float* shift, const float* residual, int32_t relu, float* y, void* stream);
/* dx = conv_transpose(dy, w)   (autograd of the same call sites) */
int mcdseg_conv_dgrad(const mcdseg_conv_desc* d, const float* dy, const float* wp_dgrad, float* dx, void* stream);
/* ---- split-precision variants: the same operators with the fp32 operands carried through the 16-bit matrix pipe as a
 * sum of exact piece products (csrc/split.h).  `math` selects the arithmetic:
 *   MCDSEG_MATH_F16X3   x = s (h1 + h2): two fp16 pieces and a power-of-two scale s per tensor, three cross terms on
 *                       v_mfma_f32_32x32x16_f16 -- 5.3x the f32 MFMA rate, fp32-grade against the reference's fp64 gradients;
 *   MCDSEG_MATH_BF16X6  x = a1 + a2 + a3: three bf16 pieces, the six largest cross terms on v_mfma_f32_32x32x16_bf16
 *                       (2.67x the f32 MFMA rate; no scale needed).
 * *_bound arguments (F16X3 only, NULL otherwise): one device float holding an UPPER BOUND of |tensor| -- written by the
 * producer of the tensor (mcdseg_bn_stats_finalize, mcdseg_bn_bwd_reduce, mcdseg_conv_split_pack_weights) or measured
 * with mcdseg_absmax; every kernel derives the scale 2^(ceil(log2 bound) - 15) from it, so no finite value overflows fp16.
 * Weight images are 16-bit, layout [k-step][piece][k-half 2][Mp][8]; sizes from *_packed_bytes.
 *   MCDSEG_MATH_F16X1   reduced precision: F16X3's operands (same images, companions, bounds -- every producer treats it as
 *                       F16X3) multiplied with the leading term only, h1 h1': one MFMA instead of three, operands rounded to
 *                       fp16's 11 significant bits.  The thin full-resolution layers' window kernels keep three terms. */
#define MCDSEG_MATH_F16X3 3
#define MCDSEG_MATH_BF16X6 6
#define MCDSEG_MATH_F16X1 1
int mcdseg_conv_split_packed_bytes(const mcdseg_conv_desc* d, int32_t math, int64_t* fprop_bytes, int64_t* dgrad_bytes);
/* 1 when the forward of this geometry runs as the direct (LDS-tiled) convolution of the network stem (7x7, stride 1,
 * pad 3, Cin <= 8, Cout <= 16; models/drn.py:126-131) -- the one case where the split path takes fewer than 16
 * contraction channels (always in the bf16x6 arithmetic).  mcdseg_conv_split_stat_rows = rows of the BN partial-statistics
 * buffer mcdseg_conv_split_fprop writes (differs from mcdseg_conv_stat_rows for the direct kernel). */
int32_t mcdseg_conv_split_direct_ok(const mcdseg_conv_desc* d);
int64_t mcdseg_conv_split_stat_rows(const mcdseg_conv_desc* d);
/* The thin full-resolution 3x3 layers (16 contraction channels, 16 / 32 outputs; models/drn.py:195-205 "_make_conv_layers") run,
 * with MCDSEG_MATH_F16X3 and a pre-split operand, on an LDS-window kernel whose partial-statistics rows differ (one per tile of
 * 8 (4) x 32 pixels and wave): mcdseg_conv_split_stat_rows_for gives the row count mcdseg_conv_split_fprop will write for this
 * (math, x_cb != NULL) combination; mcdseg_conv_split_window_ok tells whether that kernel takes the forward (dgrad = 0) or the
 * data gradient (dgrad = 1, stride 1 only).  It has no bias / affine epilogue: calling mcdseg_conv_split_fprop with a companion AND a
 * bias for such a geometry is rejected. */
int64_t mcdseg_conv_split_stat_rows_for(const mcdseg_conv_desc* d, int32_t math, int32_t presplit);
int32_t mcdseg_conv_split_window_ok(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad);
/* Workgroup tile mcdseg_conv_split_fprop / _dgrad take for M output rows (Cout forward, Cin for the data gradient) and P output
 * pixels, as the decimal digits WM WN WAVES_M WAVES_N of conv_gemm_split_kernel<P, WM, WN, WAVES_M, WAVES_N, ...>: 4222 = 256 x 128,
 * 4214 = 128 x 256, 2222 = 128 x 128, 2214 = 64 x 256, 1214 = 32 x 256 (rows x pixels).  For profilers and the benchmark's
 * per-kernel accounting; never needed to call the operators. */
int32_t mcdseg_conv_split_tile_config(int32_t M, int64_t P, int32_t presplit);
/* bound[0] = max |x[i]| (exact, order-independent; non-finite data gives a non-finite bound) */
int mcdseg_absmax(const float* x, int64_t n, float* bound, void* stream);
/* w [Cout,Cin,KH,KW] -> fprop and/or dgrad image; F16X3 first measures w_bound = max |w| (device float, written here) */
int mcdseg_conv_split_pack_weights(const mcdseg_conv_desc* d, int32_t math, const float* w, void* wp_fprop, void* wp_dgrad,
                                   float* w_bound, void* stream);
/* The same for n convolutions in two launches (every image of a model after an optimizer step).  Device tables: ptrs[4*e ..] =
 * {w, wp_fprop or 0, wp_dgrad or 0, &bounds[e]} as 64-bit addresses, dims[4*e ..] = {Cout, Cin, KH*KW, 0}; bounds: n device
 * floats, written here (F16X3; may be NULL for BF16X6, the table's bound addresses are then ignored).  The stem's direct-kernel
 * forward image is not produced by this call (pass 0 for it and use mcdseg_conv_split_pack_weights). */
int mcdseg_conv_split_pack_weights_multi(const int64_t* ptrs, const int32_t* dims, int32_t n, int32_t math, float* bounds,
                                         void* stream);
/* x_cb / dy_cb (may be NULL): the gathered operand already split by its producer into the channel-blocked layout
 * [piece][N][C/8][H*W][8 x 16 bit] (mcdseg_bn_apply_cb / mcdseg_bn_bwd_apply_cb / mcdseg_split_cb, same `math` and the same
 * bound scalar); C must be divisible by 8.  With it the K loop does no conversion and moves 16 B per piece, pixel and
 * 8-channel group straight into LDS instead of gathering 8 dwords and splitting them. */
int mcdseg_conv_split_fprop(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                            const void* wp_fprop, const float* w_bound, const float* bias, float* y, float* stat_partials,
                            void* stream);
int mcdseg_conv_split_fprop_affine(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                                   const void* wp_fprop, const float* w_bound, const float* scale, const float* shift,
                                   const float* residual, int32_t relu, float* y, void* stream);
int mcdseg_conv_split_dgrad(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                            const void* wp_dgrad, const float* w_bound, float* dx, void* stream);
/* dx = data gradient + addend: the element-wise add autograd performs when the convolution's input has a second consumer (the
 * shortcut of a residual block: models/drn.py:43-59 BasicBlock, :79-100 Bottleneck), folded into the epilogue -- same sum, same
 * order, same bits; 80 such adds per MCD step at BASELINE config 2.  `addend`: fp32, dx's shape, not dx itself.  `part` as in
 * mcdseg_conv_split_dgrad_part.  Not for geometries of the thin-layer window kernel (mcdseg_conv_split_window_ok(d, math, 1, 1)). */
int mcdseg_conv_split_dgrad_add(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                                const void* wp_dgrad, const float* w_bound, const float* addend, float* dx, int32_t part, void* stream);
/* One convolution may run as TWO launches: whole rounds of 256 x 256 tiles (one per CU) on the 8-wave ping-pong kernel
 * (csrc/conv_gemm_split_pp.hip), the remaining pixels on the 4-wave tiles.  mcdseg_conv_split_parts returns the number of output
 * pixels (a multiple of 256, counted from pixel 0 of the flattened (n, y, x) order) the first launch takes -- 0 when the geometry
 * runs as one launch.  The _part entry points are mcdseg_conv_split_fprop / _dgrad with `part` = 0 (the whole convolution, what
 * those do), 1 (only the ping-pong launch) or 2 (only the rest); parts 1 + 2 write exactly what part 0 writes.  They exist so that
 * a profiler or bench.py can bracket each kernel with its own HIP events.  Reference: nn.Conv2d, models/drn.py:21-23. */
int64_t mcdseg_conv_split_parts(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad);
/* 1 when "the rest" (part 2; the whole convolution when mcdseg_conv_split_parts returns 0) runs on the ping-pong kernel's 256 x 128
 * tile, 0 when it runs on the 4-wave tiles -- the kernel name a profiler will see. */
int32_t mcdseg_conv_split_rest_pingpong(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad);
/* 1 (2: its 128 x 320 form, for output rows that are a multiple of 128 only; 3: its 256 x 160 form, where only 160-pixel tiles fill
 * their rounds -- partial rows of 160 pixels as well, one per tile) when the WHOLE convolution runs as one launch of the
 * ping-pong kernel's 256 x 320 tile (chosen when fewer than two rounds of
 * 256 x 256 tiles exist but tiles 320 pixels wide fill their rounds to 80 %: 76800 pixels x 256 channels = 240 tiles on 256 CUs);
 * mcdseg_conv_split_parts then returns every pixel, and the BatchNorm partial rows are one per 160 pixels
 * (mcdseg_conv_split_stat_rows_for counts them). */
int32_t mcdseg_conv_split_wide_pingpong(const mcdseg_conv_desc* d, int32_t math, int32_t presplit, int32_t dgrad);
int mcdseg_conv_split_fprop_part(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                                 const void* wp_fprop, const float* w_bound, const float* bias, float* y, float* stat_partials,
                                 int32_t part, void* stream);
int mcdseg_conv_split_dgrad_part(const mcdseg_conv_desc* d, int32_t math, const float* dy, const void* dy_cb, const float* dy_bound,
                                 const void* wp_dgrad, const float* w_bound, float* dx, int32_t part, void* stream);
/* same workspace as mcdseg_conv_wgrad; 128x128-tile layers run on the split path, thin layers on the f32 kernels.
 * x_cb / dy_cb (may be NULL; used only when BOTH are given): the pre-split companions of x and dy in the layout above --
 * the kernel then transposes 8x8 blocks of 16-bit pieces in registers instead of splitting fp32 values.  x / dy may be NULL
 * only when that path applies (both companions, channel counts divisible by 8, min(Cin,Cout) > 64, not the thin-input plan). */
int mcdseg_conv_split_wgrad(const mcdseg_conv_desc* d, int32_t math, const float* x, const void* x_cb, const float* x_bound,
                            const float* dy, const void* dy_cb, const float* dy_bound, float* dw, void* workspace,
                            size_t workspace_bytes, void* stream);
/* which kernel the two weight-gradient entry points launch for a geometry (math = 0: mcdseg_conv_wgrad): 0..3 the f32 plans
 * (128x128, 64x64, 32x32 tiles, tap-packed thin inputs), 10 split arithmetic from fp32 operands, 11 / 12 / 13 from both
 * pre-split companions (register-transposing, transposed-read 128x128, transposed-read 256x128), 14 the 64-channel tap pairs,
 * 15 the thin-layer window kernel, 16 two taps per workgroup, 17 the eight-wave ping-pong kernel over a stream-K decomposition
 * (256-channel blocks on both sides; csrc/conv_wgrad_split_pp.hip), 18 its row-of-taps form for 3 x KH kernels with at most 128
 * channels on one side (tiles of 128 co x three taps x 128 ci).  For profilers and bench.py's per-kernel accounting. */
int32_t mcdseg_conv_wgrad_variant(const mcdseg_conv_desc* d, int32_t math, int32_t presplit);
/* 1 when ONE launch of mcdseg_conv_wgrad (math = 0) / mcdseg_conv_split_wgrad (presplit: both companions are passed) can address
 * this descriptor's operands with its 32-bit buffer offsets: (N*C + 128 channels of slack) planes below 2 GiB for the kernels that
 * read the fp32 tensors, N*C planes for the plans that read the companions (variants 11..18).  The host cuts larger batches along
 * N -- the reference has no such limit (nn.Conv2d, models/drn.py:21-23).  Host-side arithmetic, no launch. */
int32_t mcdseg_conv_wgrad_fits(const mcdseg_conv_desc* d, int32_t math, int32_t presplit);
/* dw = x (*) dy ; split over pixels into slabs in `workspace`, then reduced in a fixed order. */
size_t mcdseg_conv_wgrad_workspace_bytes(const mcdseg_conv_desc* d);
int mcdseg_conv_wgrad(const mcdseg_conv_desc* d, const float* x, const float* dy, float* dw,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (train: batch statistics, eps 1e-5, momentum 0.1; eval: running statistics)
 * fused with ReLU and the residual add of BasicBlock/Bottleneck (models/drn.py:43-59, 80-100)
 * ---------------------------------------------------------------------------------------------- */
/* Merge the conv epilogue partials -> mean[C], rstd[C]; update running_mean/var (unbiased var) and
 * num_batches_tracked when those pointers are non-NULL -- running_updates times (>= 1): one launch may stand for several identical
 * forward passes of the reference's schedule (solvers/solver.py; each update is rounded to fp32 as a separate pass would).
 * workspace: 8-byte aligned scratch.
 * y_bound (may be NULL): receives an upper bound of |y| for the tensor mcdseg_bn_apply(_cb) is about to write from these
 * statistics, y = act(gamma*xhat + beta (+ residual)): max_c(|gamma_c| sqrt(n-1) + |beta_c|) + res_bound[0], using
 * Samuelson's inequality |xhat| <= sqrt(n-1) for a batch of n values normalised by their own mean and biased variance
 * (gamma, beta required then; res_bound = bound scalar of the residual tensor or NULL). */
size_t mcdseg_bn_stats_workspace_bytes(int64_t rows, int32_t C);
int mcdseg_bn_stats_finalize(const float* stat_partials, int64_t rows, int32_t C, int32_t Mp,
                             float* mean, float* rstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float momentum, float eps,
                             const float* gamma, const float* beta, const float* res_bound, float* y_bound,
                             int32_t running_updates, void* workspace, size_t workspace_bytes, void* stream);
/* eval mode: mean = running_mean, rstd = 1/sqrt(running_var+eps) */
int mcdseg_bn_eval_stats(const float* running_mean, const float* running_var, int32_t C, float eps,
                         float* mean, float* rstd, void* stream);
/* eval-mode BN as an affine map: scale = gamma/sqrt(running_var+eps), shift = beta + (conv_bias - running_mean)*scale
 * (conv_bias may be NULL) -- feeds mcdseg_conv_fprop_affine */
int mcdseg_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                          const float* conv_bias, int32_t C, float eps, float* scale, float* shift, void* stream);
/* y = act(gamma*(z-mean)*rstd + beta (+ residual)), act = ReLU if relu != 0 */
int mcdseg_bn_apply(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const float* residual, float* y, int32_t N, int32_t C, int32_t HW, int32_t relu, void* stream);
/* Same as mcdseg_bn_apply / mcdseg_bn_bwd_apply, additionally emitting the split of the produced tensor (pieces of `math`,
 * scale from the bound scalar for MCDSEG_MATH_F16X3; NULL for BF16X6) in the channel-blocked layout
 * [piece][N][C/8][HW][8 x 16 bit] consumed by the split convolutions; C must be divisible by 8.  mcdseg_bn_bwd_apply_cb
 * accepts dz == NULL (only the split companion is written) for layers whose input and weight gradients both read the
 * companion.  mcdseg_split_cb is the split alone (fp32 NCHW -> companion) for operands no fused BN group produced,
 * mcdseg_unsplit_cb its inverse (value = scale * sum of the pieces: exact for BF16X6, the 22 leading bits for F16X3).
 * Compact activation storage (the fp32 activation is never written; what travels is the companion): mcdseg_bn_apply_cb accepts
 * y == NULL and the residual as a companion (res_cb, res_bound) instead of fp32; the backward kernels accept y == NULL with y_cb,
 * reading the ReLU mask from the companion's leading piece. */
int mcdseg_split_cb(const float* x, void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW, void* stream);
/* The same for a channel count that is not a multiple of 8 (the 6-channel RGB+HHA network input, adapt_trainer.py:157-160): the
 * companion has ceil(C/8) channel groups -- [piece][N][ceil(C/8)][H*W][8 x 16 bit] -- with zeros in the missing channels.  Read
 * by the weight-gradient kernel of the 7x7 stem (models/drn.py:126-131) only. */
int mcdseg_split_cb_padded(const float* x, void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW,
                           void* stream);
int mcdseg_unsplit_cb(const void* x_cb, const float* x_bound, int32_t math, int32_t N, int32_t C, int32_t HW, float* x, void* stream);
int mcdseg_bn_apply_cb(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* residual, const void* res_cb, const float* res_bound, float* y, void* y_cb,
                       const float* y_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t relu, void* stream);
int mcdseg_bn_bwd_apply_cb(const float* dy, const float* y, const void* y_cb, const float* z, const float* mean, const float* rstd,
                           const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres,
                           void* dz_cb, const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t relu,
                           int32_t train, void* stream);
/* The backward pair for a ReLU group WITHOUT a residual branch (models/drn.py:43-47 bn1; the conv-BN-ReLU chains of
 * _make_conv_layers): the mask y > 0 is recomputed from z -- y > 0 <=> fma(z, gamma rstd, beta - mean gamma rstd) > 0, which is the
 * forward kernels' own expression, bit for bit -- so y is not read at all: 8 instead of 12 bytes per element in the reduce, 12
 * instead of 16 in the apply.  Same results as mcdseg_bn_bwd_reduce / mcdseg_bn_bwd_apply_cb with relu = 1 and the group's fp32 y. */
int mcdseg_bn_bwd_reduce_zmask(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, float* dgamma, float* dbeta, float* dz_bound, int32_t train, int32_t N, int32_t C,
                               int32_t HW, void* workspace, size_t workspace_bytes, void* stream);
int mcdseg_bn_bwd_apply_cb_zmask(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, const float* dgamma, const float* dbeta, float* dz, void* dz_cb,
                                 const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t train, void* stream);
/* The same economy for a ReLU group WITH a residual branch (models/drn.py:48-57: out = relu(bn2(conv2(out)) + residual)), whose mask
 * cannot be recomputed from z: the forward apply kernel also writes the mask as a bit-plane -- per (image, channel) and block of 256
 * pixels four 64-bit words, bit l of word j = (y > 0) of pixel 256 blk + 4 l + j; mcdseg_bn_relu_mask_bytes gives its size, 0 when the
 * geometry has none (HW not a multiple of 4) -- and the backward pair reads 1 bit per element where mcdseg_bn_bwd_reduce /
 * mcdseg_bn_bwd_apply_cb read the 32 of the fp32 y: 8 instead of 12 and 16 instead of 20 bytes per element.  y > 0 is evaluated once, on
 * the value that was stored: the same mask, the same results bit for bit.  Tensors 16-byte aligned, the mask 8-byte aligned. */
size_t mcdseg_bn_relu_mask_bytes(int32_t N, int32_t C, int32_t HW);
int mcdseg_bn_apply_cb_mask(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            const float* residual, float* y, void* y_cb, const float* y_bound, void* relu_mask, int32_t math, int32_t N,
                            int32_t C, int32_t HW, void* stream);
int mcdseg_bn_bwd_reduce_mask(const float* dy, const void* relu_mask, const float* z, const float* mean, const float* rstd,
                              const float* gamma, float* dgamma, float* dbeta, float* dz_bound, int32_t train, int32_t N, int32_t C,
                              int32_t HW, void* workspace, size_t workspace_bytes, void* stream);
int mcdseg_bn_bwd_apply_cb_mask(const float* dy, const void* relu_mask, const float* z, const float* mean, const float* rstd,
                                const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres, void* dz_cb,
                                const float* dz_bound, int32_t math, int32_t N, int32_t C, int32_t HW, int32_t train, void* stream);
/* Backward.  dy is the gradient w.r.t. y; y (the saved forward output) -- or, when y == NULL, its companion y_cb of
 * arithmetic `math` -- supplies the ReLU mask when relu != 0.  reduce: dgamma[c] = sum dy_m*xhat, dbeta[c] = sum dy_m.
 * With z == NULL only dbeta is produced (used for the conv bias gradient, models/dilated_fcn.py:227).
 * dz_bound (may be NULL; needs z and gamma): receives an upper bound of |dz| for the tensor mcdseg_bn_bwd_apply(_cb) will
 * write: max_c |gamma_c rstd_c| (max|dy_m| + |dbeta_c|/n + sqrt(n-1) |dgamma_c|/n) in train mode, max_c |gamma_c rstd_c|
 * max|dy_m| in eval mode (train selects). */
size_t mcdseg_bn_bwd_workspace_bytes(int32_t N, int32_t C, int32_t HW);
int mcdseg_bn_bwd_reduce(const float* dy, const float* y, const void* y_cb, int32_t math, const float* z, const float* mean,
                         const float* rstd, float* dgamma, float* dbeta, const float* gamma, float* dz_bound, int32_t train,
                         int32_t N, int32_t C, int32_t HW, int32_t relu,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dz = gamma*rstd*(dy_m - dbeta/n - xhat*dgamma/n) (train) or gamma*rstd*dy_m (eval);
 * dres (optional) = dy_m, the gradient of the residual branch. */
int mcdseg_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean, const float* rstd,
                        const float* gamma, const float* dgamma, const float* dbeta, float* dz, float* dres,
                        int32_t N, int32_t C, int32_t HW, int32_t relu, int32_t train, void* stream);

/* ------------------------------------------------------------------------------------------------
 * x8 learned up-sampler: depthwise ConvTranspose2d(C,C,16,stride 8,pad 4,groups=C,bias=False)
 * (models/dilated_fcn.py:357-366, 479-491).  x [N,C,Hi,Wi] -> y [N,C,8Hi,8Wi], w [C,1,16,16].
 * With x2/w2 != NULL computes y = up(x,w) + up(x2,w2) (ScoreFusion + AddFusion, models/fusion.py:24-29).
 * ---------------------------------------------------------------------------------------------- */
int mcdseg_up8_fwd(const float* x, const float* w, const float* x2, const float* w2, float* y,
                   int32_t N, int32_t C, int32_t Hi, int32_t Wi, void* stream);
int mcdseg_up8_bwd_input(const float* dy, const float* w, float* dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                         void* stream);
size_t mcdseg_up8_bwd_weight_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi);
int mcdseg_up8_bwd_weight(const float* dy, const float* x, float* dw, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Both backward passes of the up-sampler (either may be skipped: dx == NULL or dw == NULL) from ONE staged read of dy -- the
 * backward of DRNSegPixelClassifier.up (models/dilated_fcn.py:357-366) as autograd runs it in adapt_trainer.py:172-212.  Same
 * results as mcdseg_up8_bwd_input / mcdseg_up8_bwd_weight up to the order of the fp32 sums; falls back to those two kernels
 * when a dy row band does not fit the LDS ring (Wi > 192). */
size_t mcdseg_up8_bwd_workspace_bytes(int32_t N, int32_t C, int32_t Hi, int32_t Wi);
int mcdseg_up8_bwd(const float* dy, const float* w, const float* x, float* dx, float* dw, int32_t N, int32_t C, int32_t Hi,
                   int32_t Wi, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused per-pixel softmax -> weighted CE (both heads) -> L1 discrepancy, forward + gradient
 * (loss.py:7-13 CrossEntropyLoss2d, loss.py:93-100 Diff2d; three-step use adapt_trainer.py:163-212)
 *
 *   losses[0] = CE(z1,labels)  losses[1] = CE(z2,labels)  losses[2] = mean|softmax(z1)-softmax(z2)|
 *   losses[3] = sum_i w[y_i]
 *   g1 = ce_coef * dCE1/dz1 + diff_coef * dDiff/dz1,   g2 likewise for z2.
 * Any of z2 / labels / g1 / g2 may be NULL (single-head CE, discrepancy only, loss values only).
 * wsum_in (device scalar, may be NULL) overrides the CE normaliser sum_i w[y_i]: under data parallelism the
 * caller passes the all-reduced sum over all shards so that averaged per-rank gradients equal the gradient
 * of the reference's single global weighted mean (nn.DataParallel gathers logits before the loss).
 * ---------------------------------------------------------------------------------------------- */
size_t mcdseg_loss_workspace_bytes(int32_t N, int32_t HW);
int mcdseg_softmax_ce_l1(const float* z1, const float* z2, const int64_t* labels, const float* class_weight,
                         int64_t ignore_index, float ce_coef, float diff_coef, const float* wsum_in,
                         float* g1, float* g2, float* losses,
                         int32_t N, int32_t C, int32_t HW, void* workspace, size_t workspace_bytes, void* stream);
/* The same losses and gradients for z_k = up8(s_k, w_k) (the x8 up-sampler above) WITHOUT the full-resolution logits ever being
 * stored: the MCD classifiers are exactly that up-sampler (models/dilated_fcn.py:357-366; F1(G(x)), F2(G(x)) of
 * adapt_trainer.py:163-212), so the kernel forms each pixel's logits from the [N,C,Hi,Wi] score maps on the fly.  s1 and s2 may
 * be the same tensor (both heads read the generator's scores); s2 / w2 NULL = single head.  g1 / g2 are [N,C,8Hi,8Wi], to be
 * fed to mcdseg_up8_bwd_input / mcdseg_up8_bwd_weight; they equal the two-pass result bit for bit. */
size_t mcdseg_up8_loss_workspace_bytes(int32_t N, int32_t Hi, int32_t Wi);
int mcdseg_up8_softmax_ce_l1(const float* s1, const float* w1, const float* s2, const float* w2, const int64_t* labels,
                             const float* class_weight, int64_t ignore_index, float ce_coef, float diff_coef, const float* wsum_in,
                             float* g1, float* g2, float* losses, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                             void* workspace, size_t workspace_bytes, void* stream);
/* out[0] = sum_i w[labels_i] over P pixels (ignore_index and out-of-range labels contribute 0) */
size_t mcdseg_label_weight_sum_workspace_bytes(int64_t P);
int mcdseg_label_weight_sum(const int64_t* labels, const float* class_weight, int64_t ignore_index, int32_t C, int64_t P,
                            float* out, void* workspace, size_t workspace_bytes, void* stream);
/* Inference tail (adapt_tester.py:101-124; util.py:44-48): o = z1, or (z1+z2)/2 when z2 != NULL;
 * labels[n,hw] = argmax_{c < C_used} o (uint8; the background channel is excluded by C_used = C-1 unless it was trained);
 * entropy[0] = -mean_{n,c,hw} p log(p + 1e-6), p = softmax over all C classes. */
size_t mcdseg_predict_workspace_bytes(int32_t N, int32_t HW);
int mcdseg_predict_labels(const float* z1, const float* z2, uint8_t* labels, float* entropy, int32_t N, int32_t C,
                          int32_t C_used, int32_t HW, void* workspace, size_t workspace_bytes, void* stream);
/* buf[i] *= *scale (device scalar) -- applies autograd's upstream scalar without a host sync */
int mcdseg_scale_by_device_scalar(float* buf, const float* scale, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multitask decoder (BASELINE config 4; models/dilated_fcn.py:661-739)
 *   bilinear x8 up-sampling with align_corners = False (nn.Upsample(scale_factor=8, mode='bilinear'), :676)
 *   loss[0] = mean((pred-target)^2), grad = 2 (pred-target) / n            (F.mse_loss, :712-714)
 * ---------------------------------------------------------------------------------------------- */
int mcdseg_bilinear8_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t Hi, int32_t Wi, void* stream);
int mcdseg_bilinear8_bwd(const float* dy, float* dx, int32_t N, int32_t C, int32_t Hi, int32_t Wi, void* stream);
size_t mcdseg_mse_workspace_bytes(int64_t n);
int mcdseg_mse(const float* pred, const float* target, float* grad, float* loss, int64_t n,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MFNet late fusion beyond the plain sum (models/fusion.py:6-50) and its loss (loss.py:16-30)
 *   gate_mix:    out = x1*s + x2*(1-s), s = sigmoid(g)   (GateFusion.forward :19-22; g = 1x1 conv of cat(x1,x2))
 *                backward: dx1 = dy*s, dx2 = dy*(1-s), dg = dy*(x1-x2)*s*(1-s).  n = element count, multiple of 4.
 *   softmax_ch:  softmax over C of NCHW (F.softmax of ScoreGateFusion :13-15); backward dx = y*(dy - sum_c dy*y). C <= 64.
 *   prob_nll:    ProbCrossEntropyLoss2d: NLLLoss2d(weight)(log(p), labels).  loss[0] = weighted mean, loss[1] = weighted
 *                sum, loss[2] = sum of weights; grad (may be NULL) = d loss[size_average ? 0 : 1] / dp, dense NCHW.
 * ---------------------------------------------------------------------------------------------- */
int mcdseg_gate_mix_fwd(const float* x1, const float* x2, const float* g, float* out, int64_t n, void* stream);
int mcdseg_gate_mix_bwd(const float* dy, const float* x1, const float* x2, const float* g, float* dx1, float* dx2, float* dg,
                        int64_t n, void* stream);
int mcdseg_softmax_ch_fwd(const float* x, float* y, int32_t N, int32_t C, int32_t HW, void* stream);
int mcdseg_softmax_ch_bwd(const float* dy, const float* y, float* dx, int32_t N, int32_t C, int32_t HW, void* stream);
size_t mcdseg_prob_nll_workspace_bytes(int32_t N, int32_t HW);
int mcdseg_prob_nll(const float* p, const int64_t* labels, const float* weight, int64_t ignore_index, int32_t size_average,
                    float* grad, float* loss, int32_t N, int32_t C, int32_t HW, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Scale(img_shape, Image.BILINEAR) / Scale(img_shape, Image.NEAREST) in front of the two transforms below (transform.py:303,
 * 320; torchvision's Scale = PIL.Image.resize) on uint8 batches: src [N,H,W,C] -> dst [N,OH,OW,C] (bilinear; Pillow's 8-bit
 * ImagingResample, bit for bit) and src [N,H,W] -> dst [N,OH,OW] (nearest; ImagingScaleAffine, for label maps).  workspace: 4-byte
 * aligned scratch of mcdseg_resize_workspace_bytes (coefficient / index tables built on the device + the image between the passes). */
size_t mcdseg_resize_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW);
int mcdseg_resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW,
                              void* workspace, size_t workspace_bytes, void* stream);
int mcdseg_resize_nearest_u8(const uint8_t* src, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW,
                             void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Either side of the step: input transform and evaluation histogram
 *   normalize_u8:   ToTensor() + Normalize(mean,std) (transform.py:302-315): src uint8 [N,H,W,Cs] (HWC) ->
 *                   dst fp32 [N,C,H,W], channels [c_off, c_off+Cs): ((u/255) - mean[c]) / std[c], IEEE division.
 *                   mean / std: device arrays of Cs floats.  RGB and HHA = two calls with c_off 0 and 3.
 *   relabel_u8:     ToLabel() + ReLabel(olabel -> nlabel) (transform.py:21-48, 319-325): uint8 -> int64
 *   confusion_hist: fast_hist (eval.py:21-23): hist[n*gt + pred] += 1 for 0 <= gt < n (pred outside [0,n) is
 *                   skipped, where numpy would fail on the reshape); hist is int64 [n*n], ACCUMULATED into.
 * ---------------------------------------------------------------------------------------------- */
int mcdseg_normalize_u8(const uint8_t* src, float* dst, const float* mean, const float* std, int32_t N, int32_t H, int32_t W,
                        int32_t Cs, int32_t C, int32_t c_off, void* stream);
int mcdseg_relabel_u8(const uint8_t* src, int64_t* dst, int64_t count, int32_t olabel, int32_t nlabel, void* stream);
int mcdseg_confusion_hist(const int64_t* gt, const int64_t* pred, int64_t count, int32_t n, int64_t* hist, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 2-byte activation storage (round 6): BASELINE config 5 ("drn_d_105 ... bf16 ... HBM-bound stress"; the network is the reference's
 * Bottleneck trunk, models/drn.py:62-100, 344-348, trained by adapt_trainer.py:155-220).  In the one-term arithmetic MCDSEG_MATH_F16X1
 * a fused conv + BatchNorm (+ residual) + ReLU group inside a trunk keeps ONE 16-bit value per element of every tensor it moves, all in
 * the companions' unit layout [N][C/8][H*W][8 x 16 bit] (one 16-byte unit = 8 channels of one pixel):
 *   z16    the convolution's output: fp16 of z / scale(z_bound), z_bound = KH KW Cin x_bound w_bound written by the convolution (|z| cannot
 *          exceed it; scale(b) = 2^(e-15), 2^e >= b, as for every F16X3 / F16X1 tensor).  fp16 rather than bf16: 11 significant bits
 *          against 8 where the per-tensor scale keeps the values in range, which a bound known before the tensor is written does;
 *   y_cb   the activation: the leading piece of the F16X3 companion, alone -- it IS the next convolution's operand;
 *   dy16   the gradient arriving from the consumers: bf16 -- gradients span more binades than one scale covers and have no a-priori
 *          bound; written by mcdseg_conv_split_dgrad_half (+ addend16: the other gradient of a residual block's input, same layout);
 *   dz_cb  the leading piece of dz's companion (scale from dz_bound, as in mcdseg_bn_bwd_reduce); dres16: dy16 under the ReLU mask.
 * mean / rstd / running statistics come from the convolution's fp32 accumulators through mcdseg_bn_stats_finalize as always; all
 * BatchNorm arithmetic is fp32.  mask_kind: 0 no ReLU; 2 y > 0 recomputed from z16 (a group without residual: the forward kernel's own
 * expression on the same stored z, so the same mask); 4 from y_cb (a group with residual).
 * F16X1 reads the LEADING piece of every companion only (piece stride 0), so one- and two-piece companions mix freely.
 * ---------------------------------------------------------------------------------------------- */
int32_t mcdseg_conv_split_half_ok(const mcdseg_conv_desc* d, int32_t math, int32_t dgrad);
/* Which kernels a call will launch, for profilers and bench.py's per-kernel tables (the host side derives rocprofv3's kernel names from
 * these instead of restating the dispatch): mcdseg_up8_loss_variant = the class count of mcdseg_up8_softmax_ce_l1's LDS-DMA instantiation
 * (16, 24, 41, 48) or minus that of the register-staged kernel; mcdseg_conv_wgrad_thin_tr_config = the template arguments
 * <cin8, mt, ntl, tr> of the thin layers' window weight gradient as cin8 * 1000000 + mt * 10000 + ntl * 100 + tr (0: not that kernel). */
int32_t mcdseg_up8_loss_variant(int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t labelled);
int32_t mcdseg_conv_wgrad_thin_tr_config(const mcdseg_conv_desc* d);
/* 1 when the 8-wave ping-pong launches of this convolution run with two K-steps of 16 channels per barrier interval (MCDSEG_MATH_F16X1 with
 * an even number of K-steps; kernel policy SplitF16x1D in profiles): the same products in the same order as the one-step kernel. */
int32_t mcdseg_conv_split_pp_deep(const mcdseg_conv_desc* d, int32_t math, int32_t dgrad);
int mcdseg_conv_split_fprop_half(const mcdseg_conv_desc* d, int32_t math, const void* x_cb, const float* x_bound, const void* wp_fprop,
                                 const float* w_bound, void* z16, float* z_bound, float* stat_partials, int32_t part, void* stream);
int mcdseg_conv_split_dgrad_half(const mcdseg_conv_desc* d, int32_t math, const void* dy_cb, const float* dy_bound, const void* wp_dgrad,
                                 const float* w_bound, const void* addend16, void* dx16, int32_t part, void* stream);
int mcdseg_bn_apply_half(const void* z16, const float* z_bound, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         const void* res_cb, const float* res_bound, void* y_cb, const float* y_bound, int32_t N, int32_t C, int32_t HW,
                         int32_t relu, void* stream);
size_t mcdseg_bn_bwd_half_workspace_bytes(int32_t N, int32_t C, int32_t HW);
int mcdseg_bn_bwd_reduce_half(const void* dy16, const void* y_cb, const void* z16, const float* z_bound, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dz_bound,
                              int32_t mask_kind, int32_t train, int32_t N, int32_t C, int32_t HW, void* workspace, size_t workspace_bytes,
                              void* stream);
int mcdseg_bn_bwd_apply_half(const void* dy16, const void* y_cb, const void* z16, const float* z_bound, const float* mean,
                             const float* rstd, const float* gamma, const float* beta, const float* dgamma, const float* dbeta, void* dz_cb,
                             const float* dz_bound, void* dres16, int32_t mask_kind, int32_t train, int32_t N, int32_t C, int32_t HW,
                             void* stream);
/* fp32 NCHW <-> bf16 units: a gradient crossing the border of the 2-byte chain through a kernel without the 16-bit epilogue */
int mcdseg_pack_bf16_units(const float* x, void* out16, int32_t N, int32_t C, int32_t HW, void* stream);
int mcdseg_unpack_bf16_units(const void* in16, float* x, int32_t N, int32_t C, int32_t HW, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SGD with momentum + weight decay on flat buffers (torch.optim.SGD via models/model_util.py:289-292)
 *   d = g*grad_scale + wd*p ; v = mu*v + d ; p -= lr*v        (v starts at 0, so the first v = d)
 * ---------------------------------------------------------------------------------------------- */
int mcdseg_sgd_momentum_flat(float* p, const float* g, float* v, int64_t n, float lr, float momentum,
                             float weight_decay, float grad_scale, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MCDSEG_H */
