/*
 * libpixelhip -- C-ABI of the MI355X (gfx950) kernels behind the PixelSSL `sseg` training step.
 *
 * This header is the drop-in boundary: plain pointers and sizes, no torch
 * types.  The reference (ZHKKKe/PixelSSL) has no native component, so there is no existing FFI
 * to replace; each entry point below names the torch operator call site(s) of the reference it
 * stands in for (file:line relative to the reference root), and INTEGRATION.md shows the ctypes
 * stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - return 0 (PXL_OK) on success, a negative PXL_ERR_* otherwise; pxl_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in
 *     the shipped host code); the library never allocates or frees device memory.  Scratch space
 *     is passed in explicitly and sized by the matching *_workspace / *_bytes query.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All work is enqueued
 *     asynchronously on it; no entry point synchronises the device or the stream.
 *   - activations are NHWC ("channels last") with an explicit channel pitch that is a multiple of
 *     16 bytes; dtype selects fp32 (exact-parity mode, v_mfma_f32_32x32x2_f32) or bf16 with fp32
 *     accumulation (throughput mode, v_mfma_f32_32x32x16_bf16).  Parameters and their gradients
 *     are fp32 in [K][taps][C] order == the memory order of a channels_last OIHW torch tensor.
 *   - re-entrant per (device, stream); one host thread per rank in the shipped design.
 */
#ifndef PIXELHIP_H
#define PIXELHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXL_VERSION 100

#define PXL_F32 0
#define PXL_BF16 1

#define PXL_OK 0
#define PXL_ERR_ARG (-1)
#define PXL_ERR_HIP (-2)
#define PXL_ERR_UNSUPPORTED (-3)
#define PXL_ERR_WORKSPACE (-4)

const char* pxl_last_error(void);
int pxl_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Convolution as implicit GEMM                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Geometry of one gather-GEMM.  Input coordinate of output pixel (oy,ox) under tap t:
 *     ny = oy*out_stride + dy[t],  iy = ny / div   (tap contributes iff ny % div == 0, 0 <= iy < Hi)
 * forward conv (k, stride s, dilation d, pad p): out_stride = s, div = 1, dy = r*d - p
 * data gradient of the same conv:                out_stride = 1, div = s, dy = p - r*d, transposed weights
 * ASPP (sum of 4 dilated 3x3): 36 taps in one launch. */
typedef struct pxl_conv_desc {
  int32_t dtype;        /* PXL_F32 | PXL_BF16 */
  int32_t B, Hi, Wi;    /* input tensor */
  int32_t Cin;          /* input channel pitch (multiple of 16 B) == K extent per tap */
  int32_t Ho, Wo;       /* output tensor */
  int32_t Cout;         /* output channel pitch */
  int32_t Kreal;        /* rows of the weight matrix (real output channels, <= Cout) */
  int32_t ntaps;        /* 1..64 */
  int32_t out_stride;
  int32_t div;          /* 1 or 2 */
  int32_t relu_in;      /* prologue: relu after the input affine */
  int32_t tile_cfg;     /* -1 = heuristic; otherwise forces a tile configuration (tests/tuning) */
  int32_t stats_rep;    /* replicas of the [2*Kreal] statistics vector (0/1 = one); tile row % stats_rep */
  int32_t split_k;      /* 0 = heuristic, 1 = off, >1 = forced number of K slices (needs workspace) */
  int16_t dy[64];
  int16_t dx[64];
} pxl_conv_desc;

/* out[m][n] = sum_{t,c} act(in)[m,t,c] * w[n][t][c]  (+ bias[n]) (+ addend[m][n]);
 * act(x) = relu?(x*in_scale[c] + in_shift[c]) applied to in-bounds taps only (zero padding stays 0);
 * stats (optional, [stats_rep][2*Kreal] fp32, caller-zeroed): per-channel sum and sum of squares of
 * `out` taken from the fp32 accumulators, spread over stats_rep replicas (fold with pxl_bn_finalize).
 * workspace (optional, >= M*Cout*4 bytes): enables split-K for launches that cannot fill the chip
 * (long reductions with few output tiles, e.g. the 36-tap ASPP head); only without stats/addend.
 * Replaces: nn.Conv2d forward/backward-data at task/sseg/module/backbone/resnet.py:18-23,69,89,106,
 * module/deeplab_v2.py:76,81-85; the fused prologue/epilogue replaces SynchronizedBatchNorm2d +
 * nn.ReLU at resnet.py:33-41 (sync_batchnorm/batchnorm.py:48-78). */
int pxl_conv_igemm(const pxl_conv_desc* desc, const void* in, const void* w, void* out,
                   const float* in_scale, const float* in_shift, const float* bias,
                   const void* addend, float* stats, void* workspace, size_t ws_bytes, void* stream);

/* Forward convolution whose input is relu?(bn(y)) of the previous convolution's RAW output y, applied to the input tiles
 * as they land in LDS ("BN-apply on load"): one launch instead of pxl_bn_finalize + pxl_bn_apply_fwd + pxl_conv_igemm and no
 * materialised activation tensor.  bin (struct below): the INPUT BatchNorm -- statistics [nrep][2*Cin] (or the running
 * statistics when training == 0), affine parameters; workgroup 0 writes bin->coef [4*Cin] and updates the running
 * statistics exactly as pxl_bn_finalize does.  Cin <= 512, Cin % 64 == 0.  Bit-identical to the three-launch path (the
 * transformed tile is rounded to bf16 like the materialised tensor).  Replaces SynchronizedBatchNorm2d + nn.ReLU + the
 * next nn.Conv2d of a Bottleneck (resnet.py:33-41).  z (optional; 1x1 / stride-1 convolutions only): the activated tensor
 * relu?(bn(y)), written on the way by the workgroups of output-channel tile 0 (what this convolution's weight gradient
 * reads).  PXL_ERR_UNSUPPORTED: use the three launches. */
struct pxl_bn_fin;
/* Diagnostics (tools/cbench.cpp; never on the product path): the LDS-DMA launch of pxl_conv_igemm built with cycle stamps.
 * trace = [workgroups][72] uint32: words 0..63 s_memtime stamps of wave 0 (kernel entry, prologue issued, end of each of the
 * first 52 K steps, loop drained, tile staged, read-back passes, statistics, everything acknowledged), 64 = stamp count,
 * 65 = HW_ID, 66 = XCC_ID, 67..70 = s_memrealtime at entry / exit, 71 = K steps.  Plain bf16 operands only. */
int pxl_conv_dma_trace(const pxl_conv_desc* desc, const void* in, const void* w, void* out, const float* bias,
                       float* stats, unsigned* trace, void* stream);
int pxl_conv_dma_bnin(const pxl_conv_desc* desc, const void* y, const void* w, void* out, const float* bias, float* stats,
                      const struct pxl_bn_fin* bin, int bin_relu, void* z, void* stream);

/* Data gradient with the BatchNorm-backward reduction of its OUTPUT fused into the epilogue (LDS-DMA kernel only):
 * din = dgrad(dy) (+ addend) and bn_sums[0..C) += sum_m gd, bn_sums[C..2C) += sum_m gd * xhat over the tensor just
 * written, gd = din * (bn_relu ? scale*bn_y + shift > 0 : 1), xhat = (bn_y - mean) * rstd from bn_coef [4C]; C =
 * d->Kreal == d->Cout.  Stands in for pxl_conv_igemm + pxl_bn_bwd_reduce (one pass over (din, bn_y) less).  Returns
 * PXL_ERR_UNSUPPORTED when the descriptor cannot run on the LDS-DMA kernel. */
int pxl_conv_dgrad_bnreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                            const void* bn_y, const float* bn_coef, int bn_relu, float* bn_sums, void* stream);

/* Data gradient that COMPLETES the gradient of a residual join's output (Bottleneck `out += residual; relu(out)`,
 * resnet.py:43-48), with the join's backward fused into the epilogue: g = dgrad(dy) (+ addend); din = g * (join_out > 0)
 * -- the masked gradient is what both branches of the join receive -- and bn_sums[0..C) += sum_m din, bn_sums[C..2C) +=
 * sum_m din * xhat(bn_y) for the main branch's last BatchNorm (bn3, no ReLU of its own).  Stands in for
 * pxl_residual_bwd_reduce (3 tensor reads + 2 writes in a launch of its own).  LDS-DMA kernel only, else
 * PXL_ERR_UNSUPPORTED. */
int pxl_conv_dgrad_joinreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                              const void* join_out, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream);
/* ... with the join's ReLU mask as ONE BIT per element -- join_bits [M][C / 8] bytes, bit e of a byte = channel 8 * chunk + e is
 * positive, written by pxl_residual_fwd_bits / pxl_residual_finalize_fwd_bits -- instead of the join output: identical results,
 * 1/16 of that operand's bytes (17.8 -> 1.1 MB per ResNet-101 stage-3 join).  bf16 only. */
int pxl_conv_dgrad_joinreduce_bits(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                   const void* join_bits, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream);

/* dw[k][t][c] += sum_m dy[m][k] * act(in)[m,t,c]   (fp32, atomically accumulated: zero dw first
 * unless accumulating).  `desc` describes the FORWARD conv (in = its input, Ho/Wo/Cout = dy).
 * creal = real input channels (<= Cin pitch), dw_cpitch = channel pitch of dw.
 * Replaces: the weight-gradient half of autograd for the same nn.Conv2d call sites. */
int pxl_conv_wgrad(const pxl_conv_desc* desc, const void* in, const float* in_scale,
                   const float* in_shift, const void* dy, float* dw, int creal, int dw_cpitch,
                   void* stream);

/* master fp32 w[K][T][C] -> wf[K][T_total][Cp] (forward operand, taps placed at t_off) and, if wt != NULL,
 * wt[C][T_total][Kp] (data-gradient operand), both in `dtype`, zero padded. */
int pxl_pack_weights(int dtype, const float* w, int K, int T, int C, void* wf, int Cp, int T_total,
                     int t_off, void* wt, int Kp, void* stream);

/* batched form: params + src_off (floats) -> packed + wf_off / wt_off (bytes; wt_off < 0 = no dgrad operand) */
typedef struct pxl_pack_item {
  int64_t src_off;      /* floats, into params */
  int64_t wf_off;       /* bytes, into packed  */
  int64_t wt_off;       /* bytes, into packed; -1 = none */
  int32_t K, T, C;      /* master tensor [K][T][C] */
  int32_t Cp, T_total, t_off, Kp;
} pxl_pack_item;
int pxl_pack_weights_batched(int dtype, const float* params, void* packed, const pxl_pack_item* items, int n,
                             void* stream);

/* layout conversion at the API edge: NCHW fp32 <-> NHWC engine dtype (channel pitch Cp >= C) */
int pxl_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W, int Cp, void* stream);
int pxl_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int H, int W, int Cp, void* stream);
/* input pipeline on the device (task/sseg/data.py:150-182 Normalize + ToTensor): uint8 image crops [B,H,W,C] ->
 * (x / 255 - mean[c]) / std[c] as fp32 NCHW with numpy's rounding (float32 division, then double-precision subtract and
 * divide each rounded to float32); mean / std: device pointers to C doubles.  pxl_u8_to_f32: label maps, the byte
 * `marker` (< 0: none) becoming `marker_value` (the stand-in of the reference's -1 "unlabeled" plane, data.py:104-105). */
int pxl_normalize_u8(int B, int C, long HW, const unsigned char* src, const double* mean, const double* stdv, float* dst, void* stream);
int pxl_u8_to_f32(long n, const unsigned char* src, float* dst, int marker, float marker_value, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* BatchNorm (training-mode, cross-device statistics)                                          */
/* ------------------------------------------------------------------------------------------ */

/* stats [nrep][2C] (sum, sumsq over `count` elements per channel; for SyncBN the caller folds the
 * replicas with pxl_bn_fold_replicas, all-reduces the [2C] vector and passes nrep = 1)
 * -> coef [4C] = mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; updates running stats
 * (momentum, unbiased variance).  training=0: coef from the running statistics.
 * clamp_var=1 selects the reference's multi-device formula clamp(var,eps)^-1/2
 * (sync_batchnorm/batchnorm.py:125) instead of (var+eps)^-1/2 (F.batch_norm, :50-53). */
int pxl_bn_finalize(int C, const float* stats, int nrep, float count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, int training,
                    int clamp_var, float* coef, void* stream);
/* z = relu?(y*scale + shift) with (scale, shift) = coef[2C..4C): the activated tensor.  The bf16 engine
 * materialises it once per BN so that the LDS-DMA contraction kernels read plain operands
 * (SynchronizedBatchNorm2d + nn.ReLU at resnet.py:33-41). */
int pxl_bn_apply_fwd(int dtype, long M, int C, const void* y, const float* coef, int relu, void* z, void* stream);
/* buf[0][i] = sum_r buf[r][i], i < n */
int pxl_bn_fold_replicas(int n, int nrep, float* buf, void* stream);
/* sums [nrep][2C] (caller-zeroed) += sum dz', sum dz'*xhat with dz' = dz * (relu ? bn(y) > 0 : 1) */
int pxl_bn_bwd_reduce(int dtype, int M, int C, const void* dz, const void* y, const float* coef, int relu,
                      float* sums, int nrep, void* stream);
/* bcoef [2C] = fold(sums) / count (zeros when training == 0: eval-mode BN is a fixed affine);
 * dgamma += sum dz'*xhat ; dbeta += sum dz' */
/* One BatchNorm's finalize inputs, for the kernels that fold pxl_bn_finalize into their prologue: every block derives the
 * (scale, shift) of its own channel group from `stats` ([nrep][2C] sums) -- or from the running statistics when
 * training == 0 -- and the blocks of row group 0 write coef [4C] (mean, rstd, scale, shift: what backward reads) and
 * update the running statistics, exactly as pxl_bn_finalize does. */
typedef struct pxl_bn_fin {
  const float* stats; int32_t nrep; float count;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;       /* may be NULL in training mode (no running-stat update) */
  float momentum, eps;
  int32_t training, clamp_var;
  float* coef;
} pxl_bn_fin;
/* Forward convolution (LDS-DMA kernel) whose LAST workgroup finalizes the BatchNorm of its output from the statistics the
 * launch itself accumulated: stats [desc->stats_rep][2*Kreal] and *counter (uint32) caller-zeroed; fin as above (its
 * stats / nrep fields are ignored, training must be 1).  Replaces pxl_conv_igemm + pxl_bn_finalize for one rank (with
 * Sync-BN the statistics are all-reduced between the two).  PXL_ERR_UNSUPPORTED: use pxl_conv_igemm + pxl_bn_finalize. */
int pxl_conv_dma_finalize(const pxl_conv_desc* desc, const void* in, const void* w, void* out, const float* bias,
                          float* stats, const pxl_bn_fin* fin, unsigned* counter, void* stream);
/* z = relu?(bn(y)) with the finalize folded in (replaces pxl_bn_finalize + pxl_bn_apply_fwd) */
int pxl_bn_finalize_apply_fwd(int dtype, long M, int C, const void* y, const pxl_bn_fin* fin, int relu, void* z, void* stream);
/* out = relu(bn_y(y) + (rfin ? bn_r(res) : res)) with both finalizes folded in (replaces up to two pxl_bn_finalize +
 * pxl_residual_fwd); rfin == NULL: identity shortcut */
int pxl_residual_finalize_fwd(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                              const pxl_bn_fin* rfin, void* out, void* stream);
/* ... + `bits` (may be NULL): the ReLU mask of `out`, one byte per 16-byte chunk ([M][C / 8] for bf16), for
 * pxl_conv_dgrad_joinreduce_bits */
int pxl_residual_finalize_fwd_bits(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                                   const pxl_bn_fin* rfin, void* out, void* bits, void* stream);

/* Backward of the bottleneck join out = relu(bn3(y) + res) (resnet.py:44-48) fused with bn3's reduction: g = dout *
 * (out > 0) -> g (and g2 = the residual branch's copy, may be NULL); sums[0..C) += sum_m g, sums[C..2C) += sum_m g*xhat */
int pxl_residual_bwd_reduce(int dtype, int M, int C, const void* dout, const void* out, const void* y, const float* coef,
                            void* g, void* g2, float* sums, void* stream);
int pxl_bn_bwd_finalize(int C, const float* sums, int nrep, float count, float* dgamma, float* dbeta,
                        float* bcoef, int training, void* stream);
/* dgamma += sums[C..2C), dbeta += sums[0..C).  Multi-rank training takes the affine gradients from the LOCAL
 * sums (before the Sync-BN all-reduce): the rank-average of local gradients is the global-batch gradient. */
int pxl_bn_param_grad(int C, const float* sums, float* dgamma, float* dbeta, void* stream);
/* finalize + apply in one launch: sums is ONE [2C] vector (replicas folded / all-reduced by the caller),
 * count = elements per channel over all devices; also accumulates dgamma += sum dz'*xhat, dbeta += sum dz'. */
int pxl_bn_bwd_apply_fused(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                           const float* sums, float count, int training, int relu, float* dgamma, float* dbeta,
                           void* dy, void* stream);
/* dy = scale * (dz' - bcoef0 - xhat*bcoef1)  -- gradient w.r.t. the raw conv output (in place allowed) */
int pxl_bn_bwd_apply(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                     const float* bcoef, int relu, void* dy, void* stream);

/* IBNorm + LeakyReLU of the GCT flaw detector (ssl_gct.py:588-607, 567-585), y NHWC [B][HW][C]: channels [0,nb) are
 * batch-normalised (affine, Sync-BN), channels [nb,C) instance-normalised (no affine).  Forward: stats -> fold ->
 * [all-reduce bn (2*nb floats) for Sync-BN] -> coef -> apply.  Backward: bwd_reduce -> fold (+ dgamma/dbeta from the
 * LOCAL sums) -> [all-reduce] -> bwd_apply.  sums / bsums: [B][2][C] fp32, bn: [2*nb], coef: [B][4][C]. */
int pxl_ibn_stats(int dtype, int B, int HW, int C, const void* y, float* sums, void* stream);
/* the same onto CALLER-ZEROED sums (the executor zeroes the sums of all IBNorm layers of a pass with one memset) */
int pxl_ibn_stats_acc(int dtype, int B, int HW, int C, const void* y, float* sums, void* stream);
int pxl_ibn_fold(int B, int C, int nb, const float* sums, float* bn, float* dgamma, float* dbeta, void* stream);
/* bn = the folded (and, multi-rank, all-reduced) batch sums of the BN half; NULL: folded from `sums` inside the kernel */
int pxl_ibn_coef(int B, int C, int nb, int HW, float count_bn, const float* sums, const float* bn, const float* gamma,
                 const float* beta, float* running_mean, float* running_var, float momentum, float eps, int training,
                 int clamp_var, float* coef, void* stream);
int pxl_ibn_apply_fwd(int dtype, int B, int HW, int C, const void* y, const float* coef, float slope, void* out,
                      void* stream);
int pxl_ibn_bwd_reduce(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef, float slope,
                       float* bsums, void* stream);
int pxl_ibn_bwd_reduce_acc(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef, float slope,
                           float* bsums, void* stream);       /* onto caller-zeroed sums */
int pxl_ibn_bwd_apply(int dtype, int B, int HW, int C, int nb, const void* dout, const void* y, const float* coef,
                      const float* bsums, const float* bn, float count_bn, int training, float slope, void* dy,
                      void* stream);

/* ------------------------------------------------------------------------------------------ */
/* ResNet trunk element-wise / pooling                                                          */
/* ------------------------------------------------------------------------------------------ */

/* out = relu( y*ycoef.scale+ycoef.shift + (rcoef ? res*rcoef.scale+rcoef.shift : res) )  resnet.py:44-48 */
int pxl_residual_fwd(int dtype, long M, int C, const void* y, const float* ycoef, const void* res,
                     const float* rcoef, void* out, void* stream);
int pxl_residual_fwd_bits(int dtype, long M, int C, const void* y, const float* ycoef, const void* res,
                          const float* rcoef, void* out, void* bits, void* stream);       /* see pxl_residual_finalize_fwd_bits */
/* nn.LeakyReLU(slope) forward / backward of the FCDiscriminator and FlawDetector stacks (ssl_adv.py:474,480-485;
 * ssl_gct.py:567-585): y = x > 0 ? x : slope*x ; dx = x > 0 ? dy : slope*dy */
int pxl_leaky_fwd(int dtype, long n, const void* x, float slope, void* y, void* stream);
int pxl_leaky_bwd(int dtype, long n, const void* dy, const void* x, float slope, void* dx, void* stream);
/* g = dout * (out > 0), optionally duplicated into g2 (the residual branch's gradient) */
int pxl_relu_mask(int dtype, long n, const void* dout, const void* out, void* g, void* g2, void* stream);
/* out[c] += sum_m x[m][c] for c < Creal (bias gradient; x NHWC with channel pitch Cp <= 256) */
int pxl_colsum(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream);
/* out = a + b + c + d (NULL operands are skipped): summed ASPP bias */
int pxl_vec_sum4(int n, float* out, const float* a, const float* b, const float* c, const float* d, void* stream);
int pxl_add_inplace(int dtype, long n, void* a, const void* b, void* stream);
/* MaxPool2d(3,2,1) of relu(bn(y)) (coef may be NULL = raw input); idx = argmax code 0..8.  resnet.py:71-73 */
int pxl_maxpool3x3s2_fwd(int dtype, int B, int Hi, int Wi, int C, const void* y, const float* coef, void* out,
                         uint8_t* idx, void* stream);
int pxl_maxpool3x3s2_bwd(int dtype, int B, int Hi, int Wi, int C, const void* dp, const uint8_t* idx, void* dz,
                         void* stream);

/* ------------------------------------------------------------------------------------------ */
/* PSPNet head data movement (task/sseg/module/_pspnet.py:41-55, 88-102), NHWC                  */
/* ------------------------------------------------------------------------------------------ */

/* nn.AdaptiveAvgPool2d(bin): in [B][H][W][Cp] -> out [B][bin][bin][Cp] (torch windows floor/ceil).  bwd: din (+)= ... */
int pxl_adaptive_avgpool_fwd(int dtype, int B, int H, int W, int Cp, int bin, const void* in, void* out, void* stream);
int pxl_adaptive_avgpool_bwd(int dtype, int B, int H, int W, int Cp, int bin, const void* dout, void* din,
                             int accumulate, void* stream);
/* F.interpolate(bilinear, align_corners=False) of relu?(in*scale+shift) ([B][h][w][Cp_in], C channels; coef = the BN's
 * [4*Cp_in] coefficients or NULL) written into channels [c_off, c_off+C) of out [B][H][W][Cp_out] (the torch.cat slot);
 * bwd: din = gradient wrt the activated low-resolution input from the same slice of dout */
int pxl_upsample_slice_fwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* in, const float* coef,
                           int relu, int H, int W, void* out, int Cp_out, int c_off, void* stream);
int pxl_upsample_slice_bwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* dout, int H, int W,
                           int Cp_out, int c_off, void* din, void* stream);
/* dst[m][d_off + c] (+)= src[m][s_off + c], c < C (channel-slice copy between NHWC tensors of different pitch) */
int pxl_slice_copy(int dtype, long M, int C, const void* src, int Cp_src, int s_off, void* dst, int Cp_dst, int d_off,
                   int accumulate, void* stream);
/* nn.PixelShuffle(2) of relu(in): in [B][h][w][Cp_in] holding 4*C channels -> out [B][2h][2w][Cp_out] holding C */
int pxl_pixshuf_relu_fwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* in, void* out, int Cp_out,
                         void* stream);
int pxl_pixshuf_relu_bwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* dout, int Cp_out,
                         const void* in, void* din, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* SSLCCT auxiliary-decoder perturbations (pixelssl/ssl_algorithm/ssl_cct.py:535-745), NCHW fp32   */
/* ------------------------------------------------------------------------------------------ */

/* out = x * mask[b][p] * cscale[b][c] * (1 + noise[c][p]) + add_scale * add (NULL factors skipped): x [B][C][HW].
 * Covers Dropout2d (cscale), G-Cutout / Con-Msk / Obj-Msk / F-Drop (mask), F-Noise (noise), I-VAT (add); the same call
 * with x = dout and add = NULL is the backward. */
int pxl_latent_perturb(int B, int C, long HW, const float* x, const float* mask, const float* cscale, const float* noise,
                       const float* add, float add_scale, float* out, void* stream);
/* mask[b][i][j] = (argmax_c pred[b][:][..] > 0) (invert: 1 - ...) sampled like F.interpolate(mode='nearest') from
 * [H][W] to [h][w]: the context / object masks of ssl_cct.py:664-676 */
int pxl_fg_mask_nearest(int B, int C, int H, int W, const float* pred, int h, int w, int invert, float* mask, void* stream);
/* att[b][p] = mean_c x[b][c][p] ; mask[b][p] = att[b][p] < max_p(att[b]) * u   (feature_dropout, ssl_cct.py:718-724) */
int pxl_chan_mean(int B, int C, long HW, const float* x, float* att, void* stream);
int pxl_fdrop_mask(int B, long HW, const float* att, float u, float* mask, void* stream);
/* out[b] = scale * x[b] / (||x[b]||_2 + 1e-8)   (_l2_normalize, ssl_cct.py:577-581; scale = eps gives r_adv) */
int pxl_l2_normalize_persample(int B, long n, const float* x, float scale, float* norm2 /* B floats of scratch */, float* out,
                               void* stream);
/* out = (a - b) * scale: d KL(softmax(pred_hat) || pred) / d pred_hat with scale = 1/B (batchmean) */
int pxl_sub_scale(long n, const float* a, const float* b, float scale, float* out, void* stream);
/* HOST routine (no GPU): what G-Cutout's cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + the `> 50 points`
 * filter yield (ssl_cct.py:627-636): bounding boxes {min_x, max_x, min_y, max_y} of the external contours of the
 * binary image mask[H][W] (host memory) whose border polygon has more than min_vertices vertices; *nboxes = number
 * found (only the first max_boxes are written). */
int pxl_external_contour_boxes_host(const uint8_t* mask, int H, int W, int min_vertices, int* boxes, int max_boxes,
                                    int* nboxes);

/* ------------------------------------------------------------------------------------------ */
/* Head tail + losses                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* F.interpolate(bilinear, align_corners) + F.softmax(dim=1): low NHWC [B][h][w][Cp] ->
 * logits / prob NCHW fp32 [B][C][H][W] (prob may be NULL).  deeplab_v2.py:32, task/sseg/model.py:62 (align_corners =
 * 1); ssl_cct.py:483 interpolates the auxiliary predictions with the default align_corners = 0 */
int pxl_upsample_softmax_fwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners,
                             const void* low, float* logits, float* prob, void* stream);
size_t pxl_upsample_bwd_workspace(int B, int w, int C, int H);
/* adjoint: dlow = U^T (dlogits + softmax_bwd(dprob, prob)); dlogits or dprob may be NULL */
int pxl_upsample_softmax_bwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners,
                             const float* dlogits, const float* dprob, const float* prob, void* dlow, void* workspace,
                             size_t ws_bytes, void* stream);

/* The training seam without full-resolution planes: everything the reference computes between the low-resolution logits
 * and their gradient -- F.interpolate (deeplab_v2.py:32), CommonSSEGCriterion on the labeled samples of the student and
 * of the teacher (task/sseg/criterion.py:24-38, ssl_mt.py:166-176), nn.MSELoss between the two predictions
 * (ssl_mt.py:179-184) and autograd's backward through all of them -- as one pass over the labels.
 * s_low / t_low: NHWC [B][h][w][Cp] in the engine dtype (t_low NULL: student-only, SupOnly); gt: float class ids
 * [n_ce][H][W]; ce_weight = d(loss)/d(per-sample CE), mse_weight = d(loss)/d(MSE mean).  Writes dlow [B][h][w][Cp] =
 * d(loss)/d(s_low) and sums[2*B+1] (zeroed here): student CE per sample, teacher CE per sample, MSE mean over samples
 * [mse_lo, mse_hi).  workspace: pxl_upsample_bwd_workspace().  PXL_ERR_UNSUPPORTED when a row does not fit the LDS
 * staging (pxl_head_loss_lds_bytes() > 64 KiB) */
size_t pxl_head_loss_lds_bytes(int w, int C, int W);
int pxl_head_loss(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                  const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                  float mse_weight, void* dlow, void* workspace, size_t ws_bytes, float* sums, void* stream);

/* CommonSSEGCriterion (task/sseg/criterion.py:24-38): loss[n] = mean over ALL HW pixels of CE with
 * ignore_index (ignored pixels add 0); gt holds class ids as float32. */
int pxl_ce_fwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index, float* loss,
               void* stream);
int pxl_ce_bwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index, const float* gout,
               float* dlogits, void* stream);

/* d(task + consistency)/d(logits) of a mean-teacher style step in one pass (ssl_mt.py:166-196: CE on the labeled
 * samples' `pred`, MSE between the student's and the teacher's `pred`): samples [0, n_ce) get pxl_ce_bwd's term (gt
 * [n_ce][HW], g_ce [n_ce]), samples [mse_lo, mse_hi) get pxl_mse_bwd's term against target ([N][C][HW], indexed like
 * logits; g_mse [1]; mean over (mse_hi - mse_lo)*C*HW elements); everything else is zero.  Each term is bit-identical to its own
 * kernel; where both apply the sum is within 1 ulp of autograd's.  g_ce / g_mse may be NULL (term absent). */
int pxl_ce_mse_bwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index, int n_ce,
                   const float* g_ce, const float* target, int mse_lo, int mse_hi, const float* g_mse,
                   float* dlogits, void* stream);
/* FCDiscriminatorCriterion (ssl_adv.py:496-503) fused with ssladv_preprocess_fcd_criterion (task/sseg/func.py:
 * 137-157): x = discriminator logits [B][1][HW], task_gt = float labels [B][1][HW] or NULL (unlabeled: nothing
 * masked), target = 1 (real) / 0 (fake).  Pixels whose label is ignore_index get x := 0, t := 0 and still count in
 * the mean (they add log 2 each), exactly like the reference's mask-then-BCE; loss[b] = mean over HW. */
int pxl_bce_logits_masked_fwd(int B, long HW, const float* x, const float* task_gt, int ignore_index, float target,
                              float* loss, void* stream);
int pxl_bce_logits_masked_bwd(int B, long HW, const float* x, const float* task_gt, int ignore_index, float target,
                              const float* gout, float* dx, void* stream);
/* CutMix mask-and-mix (ssl_cutmix.py:193-201 teacher softmax halves, :424-430 input images):
 * out[b][c] = mask[b]*a[b][c] + (1-mask[b])*b[b][c], mask [B][1][HW]; if count != NULL it receives the number of
 * pixels whose mixed max over channels exceeds `threshold` (confidence = count / (B*HW), ssl_cutmix.py:200). */
int pxl_cutmix_mix(int B, int C, long HW, const float* mask, const float* a, const float* b, float* out,
                   float threshold, float* count, void* stream);
/* ssladv_preprocess_fcd_criterion as two tensors (task/sseg/func.py:137-157): pred_out = x * m, gt_out = target * m,
 * m = (task_gt == NULL || task_gt != ignore_index); either output may be NULL.  d pred_out / d x = m: call again with
 * x = the incoming gradient and gt_out = NULL for the backward.  total = number of elements. */
int pxl_fcd_prepare(long total, const float* x, const float* task_gt, int ignore_index, float target, float* pred_out,
                    float* gt_out, void* stream);
/* FCDiscriminatorCriterion on plain tensors (ssl_adv.py:496-503): loss[b] = mean BCEWithLogits(x[b], t[b]) */
int pxl_bce_logits_fwd(int B, long HW, const float* x, const float* t, float* loss, void* stream);
int pxl_bce_logits_bwd(int B, long HW, const float* x, const float* t, const float* gout, float* dx, void* stream);
/* F.softmax(x, dim=1) of an NCHW fp32 prediction (task/sseg/model.py:59-65, func.py:216-220) and its backward */
int pxl_softmax_nchw_fwd(int N, int C, long HW, const float* x, float* p, void* stream);
int pxl_softmax_nchw_bwd(int N, int C, long HW, const float* p, const float* dp, float* dx, void* stream);
/* Validation metrics (task/sseg/func.py:36-80, numpy on the host in the reference).  pred NCHW fp32 [N,C,HW] (any
 * activation: only its channel arg-max is used, np.argmax semantics), gt float class ids [N,HW]:
 * cm[g*C + a] += #pixels with label g (0 <= g < C; 255 / -1 excluded) whose arg-max is a.  cm: C*C int64, accumulated. */
int pxl_confusion_matrix(int N, int C, long HW, const float* pred, const float* gt, long long* cm, void* stream);
/* out[n][pix] = arg-max over the C channels (visualisation, func.py:116-119) */
int pxl_argmax_u8(int N, int C, long HW, const float* pred, unsigned char* out, void* stream);
/* nn.MSELoss() (ssl_mt.py:115,182-184): out[0] = mean((a-b)^2); da = 2(a-b)/n * gout[0] */
int pxl_mse_fwd(long n, const float* a, const float* b, float* out, void* stream);
int pxl_mse_bwd(long n, const float* a, const float* b, const float* gout, float* da, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* GCT flaw-map pipeline (fp32 NCHW, single-channel maps [B][1][H][W])                           */
/* ------------------------------------------------------------------------------------------ */

/* out = mu * sum_c |onehot(gt)[c] - pred[c]| with the one-hot taken from float class ids (ignore_index and ids
 * outside [0,C) give an all-zero one-hot): sslgct_prepare_task_gt_for_fdgt (task/sseg/func.py:179-192) fused
 * with FDGTGenerator.forward's abs/sum (ssl_algorithm/ssl_gct.py:715-716). */
int pxl_absdiff_chansum(int B, int C, long HW, const float* pred, const float* gt, int ignore_index, float mu,
                        float* out, void* stream);
/* ... against a dense target gt [B][C][HW] (the explicit one-hot of the reference's task hook): FDGTGenerator's
 * `torch.sum(torch.abs(gt - pred), dim=1, keepdim=True) * mu`, ssl_gct.py:703. */
int pxl_absdiff_chansum_dense(int B, int C, long HW, const float* pred, const float* gt, float mu, float* out,
                              void* stream);
/* explicit one-hot [B][C][HW] with ignored pixels all-zero (task/sseg/func.py:159-168, 179-192) */
int pxl_onehot_ignore(int B, int C, long HW, const float* gt, int ignore_index, float* out, void* stream);
/* GaussianBlurLayer (nn/module/gaussian_blur.py:31-61) of single-channel maps: separable evaluation of the rank-1
 * k x k kernel, `taps` = the k 1-D weights (device), reflect padding k/2; tmp = scratch of the same size. */
int pxl_gauss_sep_reflect(int B, int H, int W, const float* x, const float* taps, int k, float* tmp, float* out,
                          void* stream);
/* x *= (x >= 0) in place: FlawmapHandler mutates its argument (ssl_gct.py:643-645) */
int pxl_clamp_min0_inplace(long n, float* x, void* stream);
/* nn.ReflectionPad2d(1) + nn.MaxPool2d(3, 1) (ssl_gct.py:707-710) */
int pxl_dilate3_reflect(int B, int H, int W, const float* x, float* out, void* stream);
/* out = ((max > clip ? x : 0) - min) / (max - min + 1e-9) per sample; mm = [B][2] scratch (min, max).
 * clip = -INFINITY: FDGT normalisation (ssl_gct.py:722-725); clip = 0.1: FlawmapHandler (:648-655). */
int pxl_minmax_norm_persample(int B, long HW, const float* x, float clip_threshold, float* mm, float* out,
                              void* stream);
/* GaussianNoiseLayer.forward (pixelssl/nn/module/gaussian_noise.py:18-40), in place on x [B][n] fp32: per-sample min-max
 * normalise, add `noise`, clip to [0, 1], de-normalise; mm = [B][2] scratch */
int pxl_gaussian_noise_apply(int B, long n, float* x, const float* noise, float* mm, void* stream);
/* SSLS4L._batch_prehandle + _rotate_tensor (pixelssl/ssl_algorithm/ssl_s4l.py:296-355): src [B][C][N][N] (src_kind 0 = float32,
 * 1 = int64, 2 = uint8) -> dst fp32 [2B][C][N][N] = the batch followed by one rotated copy per sample; angles = B HOST
 * ints, 0 = copy, 1 = transpose + flip(W), 2 = flip(W) + flip(H), 3 = transpose + flip(H).  1 <= B <= 128. */
int pxl_rotate_append(int src_kind, int B, int C, int N, const void* src, const int* angles, float* dst, void* stream);
/* DCGTGenerator.forward (ssl_gct.py:668-689): l_fm / r_fm are updated IN PLACE (fm <= thr ? fm : 1) */
int pxl_dcgt(int B, int C, long HW, const float* l_pred, const float* r_pred, float* l_fm, float* r_fm,
             float threshold, float* l_gt, float* r_gt, float* both_bad, void* stream);
/* flaw-correction constraint (ssl_gct.py:429-439): out[0] = mean(mask * x^2) over all n elements (x = flaw map,
 * ground truth zero, mask = both_bad); dx = 2 * mask * x / n * gout[0] */
int pxl_masked_sq_mean_fwd(long n, const float* x, const float* mask, float* out, void* stream);
int pxl_masked_sq_mean_bwd(long n, const float* x, const float* mask, const float* gout, float* dx, void* stream);
/* FlawDetectorCriterion (ssl_gct.py:617-621): loss[b] = mean over (C,H,W) of (a-g)^2; da = 2(a-g)/n * gout[b] */
int pxl_mse_persample_fwd(int B, long n, const float* a, const float* g, float* loss, void* stream);
int pxl_mse_persample_bwd(int B, long n, const float* a, const float* g, const float* gout, float* da, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Optimizer / EMA over flat fp32 buffers                                                       */
/* ------------------------------------------------------------------------------------------ */

/* torch.optim.SGD(momentum, weight_decay) semantics, pixelssl/nn/optimizer.py:57-75 */
int pxl_sgd_step(long n, float* p, const float* g, float* buf, float lr, float momentum, float weight_decay,
                 int first_step, void* stream);
/* the full torch.optim.SGD update incl. dampening and nesterov (pixelssl/nn/optimizer.py:57-75 passes both through) */
int pxl_sgd_step_general(long n, float* p, const float* g, float* buf, float lr, float momentum, float dampening,
                         float weight_decay, int nesterov, int first_step, void* stream);
/* torch.optim.Adam(lr, betas, eps) without weight decay over a flat buffer (ssl_adv.py:101-102, discriminator);
 * `step` counts from 1 (bias correction) */
int pxl_adam_step(long n, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int step, void* stream);
/* ... with torch.optim.Adam's L2 weight decay (g += wd * p; pixelssl/nn/optimizer.py:103-122 passes --weight-decay) */
int pxl_adam_step_wd(long n, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, void* stream);
/* SSLMT._update_ema_variables, pixelssl/ssl_algorithm/ssl_mt.py:359-363 */
int pxl_ema_update(long n, float* teacher, const float* student, float alpha, void* stream);
int pxl_scale_inplace(long n, float* x, float a, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Network executor (see net.h section below)                                                   */
/* ------------------------------------------------------------------------------------------ */

/* Layer program of one segmentation network (DeepLab-v2 trunk+head today).  The host describes the
 * network once; forward/backward then run the whole launch sequence from C++ with no per-layer
 * host round trip.  Offsets index the caller's flat fp32 parameter / gradient / running-stat
 * buffers (in floats). */
#define PXL_OP_INPUT 0      /* NCHW fp32 image -> NHWC tensor                        */
#define PXL_OP_CONV 1       /* conv (+ optional BN statistics of the output)         */
#define PXL_OP_MAXPOOL 2    /* 3x3/s2/p1 max-pool of relu(bn(in))                    */
#define PXL_OP_RESIDUAL 3   /* out = relu(bn(in0) + (bn(in1) | in1))                 */
#define PXL_OP_HEAD 4       /* upsample + softmax of the low-res logits -> outputs (stride = 1: align_corners=True,
                               stride = 0: align_corners=False)                                    */
#define PXL_OP_ACT 5        /* out = LeakyReLU(in0, slope), or LeakyReLU(bn_in0(in0), slope) (ssl_s4l.py:384-388) */
#define PXL_OP_IBN 6        /* out = LeakyReLU(IBNorm(in0), slope); bn_out = BN half  */
#define PXL_OP_AVGPOOL 7    /* out = AdaptiveAvgPool2d(kh)(in0)                      */
#define PXL_OP_CONCAT 8     /* out = new tensor of cout channels; in0 -> channels [0, cin)           */
#define PXL_OP_UPCAT 9      /* bilinear(relu(bn(in0))) -> channels [c_off, c_off+cin) of `out` (made by CONCAT) */
#define PXL_OP_PIXSHUF 10   /* out = PixelShuffle(2)(relu(in0)); cout = cin / 4      */

typedef struct pxl_op {
  int32_t kind;
  int32_t in0, in1, out;       /* tensor ids (-1 = none); HEAD: in0 = low-res logits, in1 = latent   */
  int32_t bn_in0, bn_in1;      /* BN ids applied to in0/in1 on load (conv/maxpool add the ReLU); HEAD: bn_in1 >= 0
                                  makes the latent relu(bn(in1)) (PSPNet) instead of the raw tensor  */
  int32_t bn_out;              /* BN id whose statistics this conv produces (-1 = none)              */
  int32_t ngroups;             /* tap groups: 1 for plain convs, 4 for the ASPP sum                  */
  int32_t w_off[4];            /* weight offset (floats) of each tap group in the parameter buffer   */
  int32_t b_off[4];            /* bias offset of each tap group (-1 = no bias)                       */
  int32_t dil[4];              /* dilation of each tap group                                         */
  int32_t pads[4];             /* zero padding of each tap group                                     */
  int32_t cin, cout;           /* real channels                                                      */
  int32_t kh, kw, stride;
  int32_t need_dgrad;          /* 0 for the stem (the image needs no gradient)                       */
  float slope;                 /* PXL_OP_ACT: negative slope of the LeakyReLU                        */
  int32_t c_off;               /* PXL_OP_UPCAT: first channel of the slice written in `out`          */
} pxl_op;

typedef struct pxl_bn_desc {
  int32_t C;
  int32_t gamma_off, beta_off;      /* in the parameter buffer  */
  int32_t rmean_off, rvar_off;      /* in the running-stat buffer */
  float eps, momentum;
} pxl_bn_desc;

typedef struct pxl_net pxl_net;

/* cross-device statistics hook (SyncBN): called between a conv's statistics epilogue and the
 * matching finalize, must all-reduce(sum) `n` floats at `buf` on `stream`.  NULL = single device. */
typedef int (*pxl_allreduce_fn)(void* user, float* buf, int n, void* stream);

int pxl_net_create(int dtype, int num_classes, const pxl_op* ops, int nops, const pxl_bn_desc* bns, int nbns,
                   int ntensors, pxl_net** out);
void pxl_net_destroy(pxl_net* net);
/* plan buffers for a batch of B images of H x W; must be called before the *_bytes queries */
int pxl_net_plan(pxl_net* net, int B, int H, int W);
/* same, with the HEAD's output size given separately (SSLCCT auxiliary decoders: input = the 33x33 latent, output =
 * the 513x513 image size, or the decoder's own 264x264 inside I-VAT) */
int pxl_net_plan_out(pxl_net* net, int B, int H, int W, int Hout, int Wout);
size_t pxl_net_packed_bytes(const pxl_net* net);     /* persistent packed-weight buffer            */
size_t pxl_net_arena_bytes(const pxl_net* net);      /* activations saved between fwd and bwd      */
size_t pxl_net_scratch_bytes(const pxl_net* net);    /* backward gradient buffers                  */
int pxl_net_set_sync(pxl_net* net, pxl_allreduce_fn fn, void* user, int world_size);
/* Gradient exchange overlapped with the backward pass (replaces nn.DataParallel's reduction, pixelssl/nn/func.py:54-62):
 * pxl_net_backward all-reduces (fn: in-place sum on the given stream; then x 1/world_size) the flat gradient buffer
 * `grads[0, total_floats)` in contiguous buckets of >= bucket_floats floats, each as soon as every kernel writing into it
 * has been issued, on its own communication stream; the caller's stream waits for the last bucket before the call's
 * work is considered done.  bucket_floats = 0: one exchange at the end.  fn = NULL: off.  Use a communicator of its own
 * for fn (RCCL runs the collectives of one communicator in issue order, a bucket must not queue behind Sync-BN). */
int pxl_net_set_grad_sync(pxl_net* net, pxl_allreduce_fn fn, void* user, int world_size, long bucket_floats, long total_floats);
/* Parameter update pipelined behind the backward pass.  The reference steps the optimizer after `loss.backward()` has
 * returned (ssl_mt.py:198-204: backward, optimizer.step, EMA update), i.e. SGD over 44.6 M parameters, the EMA and the
 * re-packing of the kernel-layout weights sit between two iterations with nothing beside them.  With a hook installed,
 * pxl_net_backward hands every bucket grads[lo, hi) of the flat gradient buffer to `fn` as soon as every kernel writing
 * into it has been issued (after the all-reduce of pxl_net_set_grad_sync when both are set), on the communication stream,
 * in descending address order; the host enqueues optimizer step / EMA / pxl_net_pack_range of that slice on `stream`.
 * The call's stream waits for the last bucket.  bucket_floats: minimum bucket size; tail_floats > 0: one more boundary
 * where at most that many floats of parameters remain below (a small last bucket: the next forward waits for it).
 * Same update arithmetic, element by element, as one step over the whole buffer.  fn = NULL: off. */
typedef int (*pxl_update_fn)(void* user, long lo, long hi, void* stream);
/* One convolution weight whose kernel (forward) layout is its master layout (channels_last, Cin % 32 == 0): floats
 * params[off, off + n) -> bf16 at packed + s_pk (student) / + t_pk (the same op of a second network; -1: none). */
typedef struct pxl_upd_seg { int64_t off, n, s_pk, t_pk; } pxl_upd_seg;
/* the segments of `net` (and of `teacher`, a network with the same program, or NULL), sorted by off; -> count (>= 0), or < 0 */
int pxl_net_update_segments(pxl_net* net, pxl_net* teacher, pxl_upd_seg* out, int cap);
/* Fused update of params[lo, hi): momentum SGD -> EMA into `t` -> both networks' bf16 forward copies -> gradient zeroed, one
 * pass (ssl_mt.py:198-204 + 359-363 + the re-packing pxl_net_pack does); same arithmetic as pxl_sgd_step / pxl_ema_update /
 * pxl_net_pack element for element.  The weights the segments do not cover (stem patches, multi-rate heads, the transposed
 * data-gradient copies) are packed by pxl_net_pack_range(which = 2 | 4). */
int pxl_sgd_ema_pack(long lo, long hi, float* p, float* g, float* buf, float* t, int nruns, const long* run_start,
                     const float* run_lr, const float* const* run_lr_dev, float momentum, float weight_decay, float alpha,
                     const float* alpha_dev, const pxl_upd_seg* segs, int nseg, void* s_packed, void* t_packed, int zero_grad,
                     void* stream);
int pxl_net_set_update_hook(pxl_net* net, pxl_update_fn fn, void* user, long bucket_floats, long tail_floats, long total_floats);
int pxl_net_update_buckets(const pxl_net* net);      /* buckets the last backward handed to the hook */
int pxl_net_grad_buckets(const pxl_net* net);        /* buckets issued by the last pxl_net_backward */
/* autotune: time every tile configuration of every contraction on the planned shapes and keep the
 * fastest (call once after plan + pack; clobbers arena / scratch / grads contents; synchronises) */
int pxl_net_tune(pxl_net* net, const float* params, const void* packed, float* grads, void* arena,
                 size_t arena_bytes, void* scratch, size_t scratch_bytes, void* stream);
/* repack params -> packed (call after every parameter update) */
int pxl_net_pack(pxl_net* net, const float* params, void* packed, void* stream);
/* the same in two independent halves: which bit 0 = forward operand layout + biases (read by pxl_net_forward), bit 1 =
 * transposed data-gradient layout (first read by pxl_net_backward: pack it on a side stream next to the forward) */
/* pxl_net_pack_parts for the convolutions whose master weights lie in params[lo, hi) (a bucket of the pipelined update);
 * which: 1 forward layouts, 2 transposed data-gradient layouts, 4 forward layouts of the convolutions pxl_net_update_segments
 * does NOT list only (the rest was written by pxl_sgd_ema_pack) */
int pxl_net_pack_range(pxl_net* net, const float* params, void* packed, int which, long lo, long hi, void* stream);
int pxl_net_pack_parts(pxl_net* net, const float* params, void* packed, int which, void* stream);
/* x NCHW fp32 [B,3,H,W] -> logits/prob NCHW fp32 [B,classes,H,W]; training selects batch statistics
 * (+ running-stat update) vs running statistics; the arena keeps what backward needs. */
int pxl_net_forward(pxl_net* net, const float* params, const void* packed, float* running, const float* x,
                    float* logits, float* prob, void* arena, size_t arena_bytes, int training, void* stream);
/* latent (backbone feature, NCHW fp32 [B,2048,h,w]) of the last forward held in `arena` */
int pxl_net_latent(pxl_net* net, const void* arena, float* latent, void* stream);
int pxl_net_latent_shape(const pxl_net* net, int* C, int* h, int* w);
/* inspection (parity tests): forward tensor `tensor` of the pass held in `arena` as NCHW fp32 [B,C,h,w]; bn >= 0 applies
 * that BatchNorm's affine first (the pre-activation whose sign its ReLU decides -- what a torch forward hook on the
 * reference's BN module would see, e.g. resnet.py:34-42).  tmp: pxl_net_tensor_bytes() of device scratch (bn >= 0) */
int pxl_net_read_tensor(pxl_net* net, const void* arena, int tensor, int bn, void* tmp, float* out, void* stream);
size_t pxl_net_tensor_bytes(const pxl_net* net, int tensor);
int pxl_net_tensor_shape(const pxl_net* net, int tensor, int* C, int* h, int* w);
/* accumulates parameter gradients into `grads` (same layout as params; caller zeroes when needed);
 * `training` must equal the flag of the forward pass that filled `arena` */
int pxl_net_backward(pxl_net* net, const float* params, const void* packed, const float* dlogits,
                     const float* dprob, const float* prob, float* grads, void* arena, size_t arena_bytes,
                     void* scratch, size_t scratch_bytes, int training, void* stream);

/* Deferred head: pxl_net_forward with logits == NULL skips the up-sampling / soft-max op; pxl_net_head_loss evaluates the
 * losses of one (teacher == NULL) or two forward passes on their low-resolution logits (pxl_head_loss) and writes
 * d(loss)/d(low-res logits) into the student's gradient slot; pxl_net_backward_low is pxl_net_backward starting from
 * that slot.  Together they stand in for deeplab_v2.py:32 + task/sseg/criterion.py:24-38 + ssl_mt.py:166-196 + the
 * autograd backward through them when no plugin reads the full-resolution predictions. */
int pxl_net_head_loss_supported(const pxl_net* net);
int pxl_net_head_forward(pxl_net* net, const void* arena, float* logits, float* prob, void* stream);   /* the skipped op, on demand */
int pxl_net_head_loss(pxl_net* net, const void* arena, const pxl_net* teacher, const void* t_arena, const float* gt,
                      int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight, float mse_weight, void* scratch,
                      size_t scratch_bytes, float* sums, void* stream);
int pxl_net_backward_low(pxl_net* net, const float* params, const void* packed, float* grads, void* arena,
                         size_t arena_bytes, void* scratch, size_t scratch_bytes, int training, void* stream);

/* Consistency seam of one SSLCCT auxiliary decoder: everything the reference runs between the decoder's own-resolution logits
 * and their gradient -- F.interpolate(bilinear, align_corners=False) to the main prediction's size, the channel soft-max
 * (sslcct_activate_ad_preds), nn.MSELoss against the detached soft-max of the main decoder (ssl_cct.py:482-484) and autograd's
 * backward through the three -- without the full-resolution planes.  low: NHWC [B][h][w][Cp] in the engine dtype; target: NCHW
 * fp32 [B][C][H][W].  pxl_cons_head_fwd writes loss[0] (the MSE mean; ordered != 0: folded in row order, bit-reproducible) and
 * parks the row-reduced gradient for a UNIT incoming gradient in `workspace` (pxl_cons_head_workspace bytes);
 * pxl_cons_head_bwd writes d(low) = gout[0] * that gradient (gout: device scalar, the incoming gradient of the scalar loss).
 * PXL_ERR_UNSUPPORTED when an output row does not fit the LDS staging (pxl_cons_head_lds_bytes > 64 KiB).
 * pxl_net_cons_head_*: the same on the low-resolution logits of a pxl_net_forward that ran with logits == NULL; the backward
 * half leaves d(low) where pxl_net_backward_low starts from. */
size_t pxl_cons_head_lds_bytes(int w, int C, int W);
size_t pxl_cons_head_workspace(int B, int w, int C, int H);
int pxl_cons_head_fwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* low,
                      const float* target, void* workspace, size_t ws_bytes, float* loss, int ordered, void* stream);
int pxl_cons_head_bwd(int dtype, int B, int h, int w, int Cp, int C, int H, int align_corners, const void* workspace,
                      size_t ws_bytes, const float* gout, void* dlow, void* stream);
int pxl_net_cons_head_supported(const pxl_net* net);
int pxl_net_cons_head_fwd(pxl_net* net, const void* arena, const float* target, void* scratch, size_t scratch_bytes, float* loss,
                          void* stream);
int pxl_net_cons_head_bwd(pxl_net* net, void* scratch, size_t scratch_bytes, const float* gout, void* stream);

/* Seed the gradient of the latent tensor before pxl_net_backward (auxiliary decoders that consume the latent outside
 * this program, SSLCCT): dlatent NCHW fp32 [B,C,h,w]; consumed (and cleared) by the next backward */
int pxl_net_seed_latent_grad(pxl_net* net, void* scratch, size_t scratch_bytes, const float* dlatent, void* stream);

/* gradient w.r.t. the network input of the last backward (NCHW fp32 [B,Cin,H,W]); the first convolution of the
 * program must have need_dgrad = 1 (discriminator / flaw detector: the input is the task model's softmax) */
int pxl_net_input_grad(pxl_net* net, const void* scratch, float* dx, void* stream);

/* Concatenation on load (FlawDetector.forward: torch.cat((inp, pred), dim=1), ssl_gct.py:578): the NEXT pxl_net_forward
 * gathers its input from `nparts` (<= 4) NCHW fp32 tensors of chans[k] channels each instead of `x` (which may then be
 * NULL); cleared by that pass.  pxl_net_input_grad_parts writes the input gradient as one NCHW tensor per part
 * (dsts[k] == NULL: not needed) -- what autograd's cat-backward + .contiguous() would produce. */
int pxl_net_set_input_parts(pxl_net* net, int nparts, const float* const* srcs, const int* chans);
/* Stem im2col patches of ONE input tensor shared by two networks with the same stem geometry (Mean Teacher without input noise
 * feeds student and teacher the same tensor, ssl_mt.py:340-348: 203 MB of patches per network at 8 x 513 x 513).
 * pxl_net_make_patches writes `net`'s patches for x into `arena` NOW (its next pxl_net_forward on that arena does not write them
 * again) and returns their address / size; pxl_net_borrow_patches makes the NEXT pxl_net_forward of `net` (and the backward of
 * that pass) read its stem operand from `patches` instead of producing its own -- the caller orders the lender's launch before
 * the borrower's pass and keeps the lender's arena alive.  patches == NULL disarms.  PXL_ERR_UNSUPPORTED: no patch-mode stem. */
int pxl_net_make_patches(pxl_net* net, const float* x, void* arena, size_t arena_bytes, void** patches, size_t* bytes, void* stream);
int pxl_net_borrow_patches(pxl_net* net, const void* patches, size_t bytes);
int pxl_net_input_grad_parts(pxl_net* net, const void* scratch, int nparts, float* const* dsts, const int* chans,
                             void* stream);
/* Stem patches: im2col of a few-channel NCHW fp32 input, P[(b, oy, ox)][(ky*kw + kx)*C + c] (zero outside the image and
 * for k >= kh*kw*C; row pitch Kp, a multiple of 8) in the engine dtype.  With them the 7x7 / stride-2 / 3-channel stem
 * (resnet.py:100-101 `conv1`) is a 1x1 convolution over Kp channels for the LDS-DMA kernels, forward and weight gradient;
 * the k order equals the master weight layout [Cout][kh][kw][C].  pxl_net uses it for a first convolution that needs no
 * data gradient (bf16 engine; PXL_STEM_PATCHES=0 keeps the gathering kernel). */
int pxl_stem_patches(int dtype, const float* x, void* P, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                     int Ho, int Wo, int Kp, void* stream);
/* the two data movements behind them (NCHW fp32 parts <-> NHWC engine tensor with channel pitch Cp) */
int pxl_nchw_parts_to_nhwc(int dtype, int nparts, const float* const* srcs, const int* chans, void* y, int B, int H,
                           int W, int Cp, void* stream);
int pxl_nhwc_to_nchw_parts(int dtype, const void* x, int nparts, float* const* dsts, const int* chans, int B, int H,
                           int W, int Cp, void* stream);
/* enable = 0: pxl_net_pack skips the transposed (data-gradient) weight copies -- networks that only run forward (the
 * Mean-Teacher teacher); pxl_net_backward then refuses to run */
int pxl_net_set_pack_dgrad(pxl_net* net, int enable);
/* Forward pass of TWO networks that run the same program (Mean Teacher's student || teacher -- ssl_mt.py:131-170 runs them one
 * after the other --, GCT's l || r task models, ssl_gct.py:176-230) in lockstep on ONE stream, every convolution of the pair as
 * ONE launch where the two match (same geometry, tile configuration and operand combination).  Per network the arguments,
 * semantics and results are exactly those of pxl_net_forward (logits* NULL: the HEAD op is skipped); nothing is shared between
 * the passes.  pxl_net_tune_pair selects the tile configuration of every paired launch by timing it (arenas are clobbered:
 * warm-up only); pxl_net_pairs(n0) = convolutions the last paired pass issued as one launch. */
int pxl_net_forward_pair(pxl_net* n0, pxl_net* n1, const float* params0, const float* params1, const void* packed0,
                         const void* packed1, float* running0, float* running1, const float* x0, const float* x1,
                         float* logits0, float* prob0, float* logits1, float* prob1, void* arena0, void* arena1,
                         size_t arena_bytes0, size_t arena_bytes1, int training0, int training1, void* stream);
int pxl_net_tune_pair(pxl_net* n0, pxl_net* n1, const float* params0, const float* params1, const void* packed0,
                      const void* packed1, void* arena0, void* arena1, size_t arena_bytes0, size_t arena_bytes1, void* stream);
/* Forward passes that follow update the BatchNorm running statistics as if each had run `times` times on the same batch
 * (momentum 1 - (1-m)^times).  For a caller that would run the SAME forward twice with unchanged weights -- SSLGCT's step-0
 * no-grad pass and its step-1 pass (ssl_gct.py:196-200, 403) -- and runs it once.  times = 1: the plain update. */
int pxl_net_set_bn_repeat(pxl_net* net, int times);
/* enable = 1: pxl_net_tune times every forward tile candidate with TWO copies of the launch in flight on two streams -- for a
 * network whose forward runs beside a copy of itself (SSLMT's student || teacher, ssl_mt.py:166-180).  Before the first pass. */
int pxl_net_set_tune_dual(pxl_net* net, int enable);
int pxl_net_pairs(const pxl_net* n);
/* Sync-BN statistics exchanges the last paired pass issued for BOTH networks in one launch (pxl_peer_allreduce_fold). */
int pxl_net_pair_syncs(const pxl_net* n);

/* enable = 0: backward skips every parameter gradient (a frozen discriminator only relays dL/dinput) */
int pxl_net_set_wgrad(pxl_net* net, int enable);

/* Tuning aid: target number of thread blocks of the row-streaming kernels (key: 0 = bn_bwd_reduce, 1 = bn_bwd_apply_fused,
 * 2 = residual_fwd, 3 = bn_apply_fwd); tools/eltwise_bench.py sweeps it, the defaults are the measured optimum */
#define PXL_TUNE_COUNT 5      /* key 4 = widest column group (16-byte chunks of one row per block) of those kernels */
int pxl_tune_set(int key, int value);

/* ------------------------------------------------------------------------------------------ */
/* RCCL communicator driven from C (one per process / GPU); librccl is resolved at run time     */
/* ------------------------------------------------------------------------------------------ */
typedef struct pxl_comm pxl_comm;
int pxl_comm_available(void);                                  /* 1 if librccl could be loaded */
int pxl_comm_unique_id(void* id128);                           /* rank 0: 128-byte id to broadcast to every rank */
int pxl_comm_init(const void* id128, int rank, int world, pxl_comm** out);      /* collective */
void pxl_comm_destroy(pxl_comm* comm);
/* in-place all-reduce(sum) of n floats at device pointer buf, enqueued on `stream` */
int pxl_comm_allreduce_sum(pxl_comm* comm, float* buf, long n, void* stream);
/* the same with the pxl_allreduce_fn signature: pxl_net_set_sync(net, pxl_comm_allreduce_hook, comm, world) makes the
 * Sync-BN statistics exchange a direct RCCL call from the executor (no host-language callback in the loop) */
int pxl_comm_allreduce_hook(void* user, float* buf, int n, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Hardware-queue-aware stream placement (csrc/streams.hip): the reference leaves concurrency to   */
/* nn.DataParallel's threads; this engine overlaps ROLES on HIP streams, and two streams overlap   */
/* only if they sit on different hardware queues -- the pool hands out streams proven to do so.    */
/* ------------------------------------------------------------------------------------------ */
#define PXL_STREAM_SIDE 0     /* a second network next to the main stream: MT teacher, GCT r model, AdvSSL discriminator update */
#define PXL_STREAM_WGRAD 1    /* weight gradients next to the data-gradient chain */
#define PXL_STREAM_AUX 2      /* weight packing, gradient-bucket exchange, a third chain */
int pxl_stream_pool_init(void* main_stream, int* nqueues);      /* idempotent; PXL_STREAM_POOL=0 disables placement */
void* pxl_stream_role(int role);                                /* NULL: placement off / not initialised / single queue */
int pxl_stream_pool_probes(void);

/* ------------------------------------------------------------------------------------------ */
/* One-shot all-reduce of small vectors over peer-mapped buffers (csrc/peer.hip): what replaces   */
/* the per-BatchNorm master/slave exchange of sync_batchnorm/comm.py:59-137 for N > 1.            */
/* ------------------------------------------------------------------------------------------ */
typedef struct pxl_peer pxl_peer;
/* allocates this rank's exchange buffer (2 slot sets x world x slot_floats + flags); timeout_ms bounds every wait */
int pxl_peer_create(int rank, int world, int slot_floats, int timeout_ms, pxl_peer** out);
int pxl_peer_handle(pxl_peer* peer, void* handle64);           /* 64-byte HIP IPC handle of the buffer: gather over ranks */
int pxl_peer_open(pxl_peer* peer, const void* handles);        /* [world][64] handles in rank order; maps every peer */
void pxl_peer_destroy(pxl_peer* peer);
/* in-place all-reduce(sum) of n floats at device pointer buf, enqueued on `stream`: one kernel per slot_floats floats.
 * Every rank issues the same call sequence on a context; one context per concurrently running network. */
int pxl_peer_allreduce_sum(pxl_peer* peer, float* buf, long n, void* stream);
/* Sync-BN statistics in ONE launch (sync_batchnorm/batchnorm.py:56-78: sum, ssum of the local batch -> master -> broadcast):
 * buf0, and buf1 when given (the same BatchNorm of a second network: the student || teacher pass of pxl_net_forward_pair), hold
 * nrep replicas [nrep][n] of the local sums; the replicas are folded, both vectors exchanged together, the all-reduced sums
 * land in replica 0 of each.  Stands in for pxl_bn_fold_replicas + pxl_peer_allreduce_sum per network. */
int pxl_peer_allreduce_fold(pxl_peer* peer, float* buf0, float* buf1, long n, int nrep, void* stream);
/* BatchNorm backward of a multi-rank pass in ONE launch (sync_batchnorm/batchnorm.py's backward through _sync_master): sums =
 * [sum(dz) | sum(dz * xhat)] of the local batch; dbeta += sums[0..C), dgamma += sums[C..2C) from the LOCAL values (either may be
 * NULL), then sums <- all-reduce(sums).  Stands in for pxl_bn_param_grad + pxl_peer_allreduce_sum. */
int pxl_peer_allreduce_bnbwd(pxl_peer* peer, float* sums, int C, float* dgamma, float* dbeta, void* stream);
int pxl_peer_allreduce_hook(void* user, float* buf, int n, void* stream);      /* pxl_allreduce_fn signature */
/* *status = 0, or k > 0: an exchange gave up waiting for rank k-1 (its result is invalid).  Synchronises the device. */
int pxl_peer_status(pxl_peer* peer, int* status);
/* The same word read from mapped host memory, WITHOUT synchronising the device: the exchange that times out writes it there
 * as well, so the host can look at it at every optimizer step (dist.poll_peers).  -1: no host mirror on this context.
 * A timed-out exchange returns NaN in every element whose peer word did not arrive (never a stale word): the step that used
 * it is detectably invalid.  Reference: sync_batchnorm/comm.py:59-137 blocks forever on a dead replica thread. */
int pxl_peer_status_nosync(const pxl_peer* peer);
/* exchanges issued on this context so far (identical on every rank by construction; bench.py: exchanges per step) */
long pxl_peer_exchanges(const pxl_peer* peer);

/* Measurement aid (bench.py roofline leg): bracket every contraction launch of this net with HIP
 * events on the launch stream.  kind 0 = implicit-GEMM conv (forward + data gradient), 1 = weight
 * gradient.  read() synchronises on the recorded events, returns the summed kernel time, the number
 * of launches and their ALGORITHMIC flops (2*MAC of the real, unpadded problem), and resets. */
int pxl_net_profile(pxl_net* net, int enable);
int pxl_net_profile_read(pxl_net* net, int kind, double* ms, long* launches, double* flops);
/* algorithmic operand bytes (activations in + weights + output, each once) of the launches stamped since the last
 * call, per kind; resets the counter */
int pxl_net_profile_bytes(pxl_net* net, int kind, double* bytes);

/* ------------------------------------------------------------------------------------------ */
/* Per-step scalars in device memory (captured training steps)                                  */
/* ------------------------------------------------------------------------------------------ */
/* A training iteration whose launches are replayed from a hipGraph must not carry values that change from step to step in
 * its kernel arguments.  The three that do in the reference's loops -- the learning rate per parameter group
 * (nn/lrer.py:143-179), the EMA coefficient (ssl_mt.py:359-363) and the ramped consistency weight (nn/func.py:44-52,
 * ssl_mt.py:190-196) -- live in a small device block that pxl_hyper_set fills EAGERLY right before the graph launch; the
 * *_hp entry points are the same kernels reading them from there (bit-identical arithmetic). */
/* dst[0..n) <- vals[0..n), n <= 32; `vals` is a HOST array copied into the launch arguments at enqueue time (no staging buffer) */
int pxl_hyper_set(float* dst, const float* vals, int n, void* stream);
int pxl_sgd_step_hp(long n, float* p, const float* g, float* buf, const float* lr_dev, float momentum, float weight_decay,
                    int first_step, void* stream);
int pxl_ema_update_hp(long n, float* teacher, const float* student, const float* alpha_dev, void* stream);
int pxl_head_loss_hp(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                     const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                     const float* mse_weight_dev, void* dlow, void* workspace, size_t ws_bytes, float* sums, void* stream);
int pxl_net_head_loss_hp(pxl_net* net, const void* arena, const pxl_net* teacher, const void* t_arena, const float* gt,
                         int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight, const float* mse_weight_dev,
                         void* scratch, size_t scratch_bytes, float* sums, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Bit-reproducible reductions (the executor's PXL_DETERMINISTIC=1 mode)                        */
/* ------------------------------------------------------------------------------------------ */
/* The reference's reductions (F.batch_norm, conv weight gradients, bias gradients, the criterion) run in whatever order the
 * library picks; the default kernels here combine partial sums with fp32 atomics, whose order differs from launch to launch.
 * These variants fix the order (one add per destination, or partials folded in index order), for parity runs:
 *   - pxl_conv_dgrad_bnreduce / _joinreduce take desc->stats_rep replicas of bn_sums ([stats_rep][2C]; tile row t adds into
 *     replica t % stats_rep), pxl_bn_bwd_reduce uses replica = row group when nrep >= #row groups (<= 256);
 *     pxl_bn_fold_replicas folds them in index order;
 *   - pxl_conv_wgrad: desc->split_k = 1 = ONE pixel split (every element of dw receives one add);
 *   - pxl_colsum_ordered: one block per column slab;  pxl_residual_bwd_reduce_rep: pxl_residual_bwd_reduce with `nrep` replicas;
 *   - pxl_head_loss_ex: pxl_head_loss / _hp (mse_weight_dev != NULL) with the kernel chosen (kernel_choice -1 auto, 0 row-wise,
 *     1 cell-wise) and, ordered != 0, the row-wise kernel with its loss sums folded in row order (workspace:
 *     pxl_upsample_bwd_workspace + B * H * 12 bytes). */
int pxl_colsum_ordered(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream);
int pxl_residual_bwd_reduce_rep(int dtype, int M, int C, const void* dout, const void* out, const void* y, const float* coef,
                                void* g, void* g2, float* sums, int nrep, void* stream);
int pxl_head_loss_ex(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                     const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                     float mse_weight, const float* mse_weight_dev, int kernel_choice, int ordered, void* dlow, void* workspace,
                     size_t ws_bytes, float* sums, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Seams between the translation units of the library                                          */
/* ------------------------------------------------------------------------------------------ */
/* Exported because csrc/net.cpp, tools/cbench.cpp and the kernel tests reach them across object files; a host that drives the
 * library through the entry points above never needs them.  They stand in for the same reference call sites as the
 * dispatchers that forward to them (pxl_conv_igemm / pxl_conv_wgrad: backbone/resnet.py:30-50, deeplab_v2.py:81-85). */
/* 1 when the LDS-DMA kernel (conv_dma_kernel.h: tiles HBM/L2 -> LDS by buffer_load ... lds) can run this launch: plain
 * operands (no in_scale / in_shift prologue), Cin % 64 == 0 (bf16) / % 32 == 0 (fp32), 16-byte-aligned pitches. */
int pxl_conv_dma_eligible(const pxl_conv_desc* desc, const float* in_scale, const void* workspace);
/* the LDS-DMA launch itself (pxl_conv_igemm forwards here when eligible); arguments as pxl_conv_igemm without the prologue */
int pxl_conv_dma(const pxl_conv_desc* desc, const void* in, const void* w, void* out, const float* bias, const void* addend,
                 float* stats, void* workspace, size_t ws_bytes, void* stream);
/* split-K epilogue shared by both convolution kernels: out[m][n] = T(ws[m][n] + bias[n]) for n < Kreal, 0 above */
int pxl_splitk_finish(int dtype, long total, int Cout, int Kreal, const float* ws, const float* bias, void* out, void* stream);
/* The LDS-DMA convolution as fp32 partial sums: K in `slices` equal slices (1 = unsplit), slice s stores its [M][Cout] tile at
 * ws + s * M * Cout floats; no finish pass, nothing rounded -- the caller consumes the slabs (pxl_aspp_col2im).  bf16 only;
 * PXL_ERR_UNSUPPORTED when the tile configuration cannot stage fp32 sums, K is not divisible or the workspace is too small. */
int pxl_conv_dma_slabs(const pxl_conv_desc* desc, const void* in, const void* w, float* ws, size_t ws_bytes, int slices, void* stream);
/* Multi-rate head (DeepLab-v2 ASPP classifier, deeplab_v2.py:76-85) as ONE GEMM (csrc/aspp.hip): P = X . Wp^T with a column
 * j = g * GP + t_local * cout + c per (dilation group g, tap, class c), GP = cout * tpg rounded up to 64.
 * pxl_aspp_pack: Wp [ngroups * GP][Cp] and / or its transpose Wd [Cp][ngroups * GP] from the master weights params + w_off[g]
 * ([cout][kh][kw][Cin] each; padding rows / columns zero);
 * pxl_aspp_col2im: out[b,y,x,c] = bias[c] + sum_t sum_slabs P[(b, y + dy_t, x + dx_t)][j(t, c)] (fp32 sums, rounded once);
 * pxl_aspp_dp_gather: dP[(b,y',x')][j(t, c)] = dOut[b, y' - dy_t, x' - dx_t, c], zero outside and in the padding columns;
 * pxl_aspp_dw_scatter: master gradient row (c * tpg + t) of group g += row j(t, c) of the GEMM's weight gradient tmp [J][Cpin].
 * dy / dx: the ngroups * tpg tap offsets of the forward convolution (input minus output position). */
int pxl_aspp_pack(int dtype, const float* params, const long* w_off, int ngroups, int GP, int cout, int tpg, int Cin, int Cp,
                  void* Wp, void* Wd, void* stream);
int pxl_aspp_col2im(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                    const int16_t* dx, const float* P, int nslab, size_t slab_floats, const float* bias, void* out, int Cp, void* stream);
int pxl_aspp_dp_gather(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                       const int16_t* dx, const void* dout, int Cp, void* dP, void* stream);
int pxl_aspp_dw_scatter(const float* tmp, int ngroups, int GP, int cout, int tpg, int Cin, int Cpin, float* grads, const long* w_off,
                        void* stream);
/* ... over nslab partial-sum slabs ws[nslab][total] written side by side (the LDS-DMA kernel's split-K when the workspace holds
 * one slab per K slice: plain stores instead of fp32 atomics, no pre-zeroing, a fixed summation order) */
int pxl_splitk_finish_slabs(int dtype, long total, int Cout, int Kreal, int nslab, const float* ws, const float* bias, void* out,
                            void* stream);
/* weight-gradient twins of the two above (pxl_conv_wgrad forwards here when eligible) */
int pxl_conv_wgrad_dma_eligible(const pxl_conv_desc* desc, const float* in_scale);
int pxl_conv_wgrad_dma(const pxl_conv_desc* desc, const void* in, const void* dy, float* dw, int creal, int dw_cpitch, void* stream);
/* Paired launches (pxl_net_forward_pair; ssl_mt.py:166-180, the two forwards of one iteration): between begin(&slot) and end()
 * the calling THREAD's next LDS-DMA convolution is recorded into *slot instead of launched; pxl_dma_launch_captured issues what
 * two brackets recorded -- ONE launch (gridDim.z = 2) when geometry and tile agree, else one after the other -- and frees the
 * slots.  -> 1 paired, 0 issued separately, < 0 error.  pxl_elt_pair_begin / _end: the same bracket for the finalize-folding
 * row-streaming kernels (residual join, BN + ReLU) of the two networks. */
void pxl_dma_capture_begin(void** slot);
void pxl_dma_capture_end(void);
int pxl_dma_launch_captured(void* slot0, void* slot1);
void pxl_elt_pair_begin(void);
int pxl_elt_pair_end(void);

#ifdef __cplusplus
}
#endif
#endif /* PIXELHIP_H */
