// LDS-DMA implicit-GEMM convolution for gfx950: C-ABI entry points and tile-configuration dispatch.  The kernel itself is
// conv_dma_kernel.h; its tile configurations are instantiated in conv_dma_a .. d.hip (parallel compilation) and here (12..15,
// the timing ablations).
#include "conv_dma_kernel.h"

using namespace pxl_dma;

// 1 if the descriptor / operand combination can run on the LDS-DMA kernel
extern "C" int pxl_conv_dma_eligible(const pxl_conv_desc* d, const float* in_scale, const void* workspace) {
  (void)workspace;
  if (d->dtype != PXL_BF16 || in_scale != nullptr) return 0;
  if (d->Cin % 64 != 0 || d->Cout % 8 != 0 || (d->div != 1 && d->div != 2)) return 0;
  if (d->div == 2 && d->out_stride != 1) return 0;
  if (d->div == 2) {
    // data gradient of a stride-2 convolution: the kernel walks ALL taps and lets the parity test zero the 3 of 4 that do
    // not apply to a pixel.  Measured a win for the 1x1 / 3x3 strided convolutions of the ResNet (125 -> 48 us) and still
    // slightly ahead of the gathering kernel for the 4x4 stacks of the discriminators (AdvSSL 31.5 vs 32.4 ms, GCT 54.0 vs
    // 55.4 ms in the same call).  PXL_DMA_STRIDED_DGRAD=<max taps> restricts it.
    static const int max_taps = getenv("PXL_DMA_STRIDED_DGRAD") ? atoi(getenv("PXL_DMA_STRIDED_DGRAD")) : 64;
    if (d->ntaps > max_taps) return 0;
  }
  if ((long)d->Kreal * d->ntaps * d->Cin * 2 >= (1L << 31)) return 0;
  if ((long)d->B * d->Ho * d->Wo >= (1L << 24)) return 0;      // float-reciprocal pixel decomposition in the prologue
  return 1;
}

// tile configurations 8..: 8 = 128x128, 9 = 128(pixels)x64, 10 = 64x128, 11 = 64x64 with a 3-stage LDS ring;
// 12..15 the same with 4 stages, 16..19 with 2 stages (more blocks per CU);
// 20..27: tall tiles 96 / 160 / 192 / 256 (pixels) x 128 with ONE wave column (WM = 1, WN = 4): M = 8 * 33 * 33 = 8712 is
// 68.06 tiles of 128 and 136.1 tiles of 64 -- every ResNet-101 stage-3/4 launch ends with a nearly empty last wave of
// workgroups (274 on 256 CUs).  96-row tiles give 91 x N/128 workgroups (182 for N = 256: one wave at 1.17x the tile
// traffic instead of two), 160 / 192 rows fit N = 512 / 2048 the same way.  20..23 = 3 stages, 24..27 = 2 stages;
// 28..35: 8-wave workgroups (two waves per SIMD): 28 = 64x128, 29 = 128x64, 30 / 31 = 128x128 as 2x4 / 4x2 waves (2 stages),
// 32 = 64x128, 33 = 128x128 (3 stages), 34 = 256x128, 35 = 128x256 (2 stages)
namespace { int conv_dma_launch(const pxl_conv_desc* d, DmaArgs& a, int sk, size_t ws_bytes, void* stream); }

extern "C" int pxl_conv_dma(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                            const void* addend, float* stats, void* workspace, size_t ws_bytes, void* stream) {
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = addend; a.stats = stats;
  a.ws = reinterpret_cast<float*>(workspace);
  a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  const int sk = d->split_k;
  return conv_dma_launch(d, a, sk, ws_bytes, stream);
}

// Timeline probe of pxl_conv_dma (tools/cbench.cpp, not on the product path): the same launch built with cycle stamps;
// trace = [workgroups][72] uint32 (layout: TRACE_WORDS above).  Plain operands only (no split-K, no BN-on-load).
extern "C" int pxl_conv_dma_trace(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                  float* stats, unsigned* trace, void* stream) {
  PXL_REQUIRE(d && in && w && out && trace, "conv_dma_trace: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_trace: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = trace;
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Forward convolution with batch statistics AND the BatchNorm finalize: `stats` ([stats_rep][2*Kreal], caller-zeroed)
// receives the sums as in pxl_conv_igemm, and the workgroup that finishes last computes what pxl_bn_finalize would
// (fin->coef, running statistics) from them -- `counter` is one caller-zeroed uint32 per launch.  fin->stats / nrep are
// ignored (the launch's own stats / stats_rep are used).  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the
// descriptor or the launch would split K.
extern "C" int pxl_conv_dma_finalize(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                     float* stats, const pxl_bn_fin* fin, unsigned* counter, void* stream) {
  PXL_REQUIRE(d && in && w && out && stats && fin && fin->coef && counter && fin->count > 0.f, "conv_dma_finalize: bad argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || (d->tile_cfg >= 0 && d->tile_cfg < 8) || !fin->training)
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_finalize: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin = *fin; a.fin_counter = counter;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Forward convolution whose input is relu?(bn(y)) of the previous convolution's RAW output y, applied to the tiles as they
// land in LDS: stands in for pxl_bn_finalize + pxl_bn_apply_fwd + pxl_conv_igemm (no materialised activation, one launch
// instead of two or three).  `bin`: the input BatchNorm (statistics [nrep][2*Cin] or running statistics, affine
// parameters, coef [4*Cin] written by workgroup 0, running statistics updated there); d->Cin == the BatchNorm's channel
// count <= 512, a multiple of 64.  z (optional, 1x1 / stride-1 convolutions only): the activated tensor, written by the
// workgroups of output-channel tile 0.  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the descriptor.
extern "C" int pxl_conv_dma_bnin(const pxl_conv_desc* d, const void* y, const void* w, void* out, const float* bias,
                                 float* stats, const pxl_bn_fin* bin, int bin_relu, void* z, void* stream) {
  PXL_REQUIRE(d && y && w && out && bin && bin->coef && bin->count > 0.f, "conv_dma_bnin: bad argument");
  PXL_REQUIRE(bin->training ? (bin->stats != nullptr && bin->nrep >= 1) : (bin->running_mean && bin->running_var),
              "conv_dma_bnin: missing statistics");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->div != 1 || d->Cin > 512 || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: descriptor is not eligible for the BN-on-load kernel");
  DmaArgs a;
  a.in = y; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin = *bin; a.bin_relu = bin_relu; a.bin_z = z; a.trace = nullptr;
  if (z != nullptr) {          // the activated tensor can only be written by a kernel that walks every input pixel exactly once per tile row
    bool plain = d->ntaps == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->out_stride == 1 && d->Ho == d->Hi && d->Wo == d->Wi;
    if (!plain) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: z output needs a 1x1 / stride-1 convolution");
  }
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Data gradient with the BatchNorm-backward reduction of its output fused into the epilogue: din = dgrad(dy) (+ addend)
// and bn_sums[0..C) += sum_m gd, bn_sums[C..2C) += sum_m gd * xhat over the tensor just written (C = d->Kreal ==
// d->Cout: an unpadded channel count), gd = din * (bn_relu ? bn_coef.scale * bn_y + bn_coef.shift > 0 : 1).  Replaces
// a separate pxl_bn_bwd_reduce pass over (din, bn_y).  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the
// descriptor (the caller then uses pxl_conv_igemm + pxl_bn_bwd_reduce).
extern "C" int pxl_conv_dgrad_bnreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                       const void* bn_y, const float* bn_coef, int bn_relu, float* bn_sums, void* stream) {
  PXL_REQUIRE(d && dy && wt && din && bn_y && bn_coef && bn_sums, "conv_dgrad_bnreduce: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->Kreal != d->Cout || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dgrad_bnreduce: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = dy; a.w = wt; a.out = din; a.bias = nullptr; a.addend = addend; a.stats = bn_sums;
  a.ws = nullptr; a.nk_per = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = bn_relu; a.bn_mask = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  q.stats_rep = 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}

// Data gradient that completes the gradient of a residual join's output, with the join's backward fused into the
// epilogue: g = dgrad(dy) (+ addend); din = g * (join_out > 0) -- the ReLU after the join -- and bn_sums[0..C) +=
// sum_m din, bn_sums[C..2C) += sum_m din * xhat(bn_y) for the main branch's last BatchNorm (the one without a ReLU of its
// own).  Replaces pxl_residual_bwd_reduce (3 tensor reads + 2 writes) by two extra reads in this epilogue.
extern "C" int pxl_conv_dgrad_joinreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                         const void* join_out, const void* bn_y, const float* bn_coef, float* bn_sums,
                                         void* stream) {
  PXL_REQUIRE(d && dy && wt && din && join_out && bn_y && bn_coef && bn_sums, "conv_dgrad_joinreduce: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->Kreal != d->Cout || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dgrad_joinreduce: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = dy; a.w = wt; a.out = din; a.bias = nullptr; a.addend = addend; a.stats = bn_sums;
  a.ws = nullptr; a.nk_per = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = 0; a.bn_mask = join_out;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  q.stats_rep = 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}

namespace {
int conv_dma_launch(const pxl_conv_desc* d, DmaArgs& a, int sk, size_t ws_bytes, void* stream) {
  a.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.Kreal = d->Kreal;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.div_shift = d->div == 2 ? 1 : 0;
  a.M = d->B * d->Ho * d->Wo;
  a.Ktot = d->ntaps * d->Cin;
  a.nk = a.Ktot / 64;
  a.tiles_m = a.tiles_n = 0;
  a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->Cin * 2);
  a.w_bytes = (unsigned)((size_t)d->Kreal * a.Ktot * 2);
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // taps at offset (0,0) only and every output pixel inside the input: no bounds logic at all
  bool gather = false;
  for (int t = 0; t < d->ntaps; ++t) gather |= d->dy[t] != 0 || d->dx[t] != 0;
  gather |= d->ntaps != 1;
  gather |= d->div != 1;
  gather |= (d->Ho - 1) * d->out_stride >= d->Hi || (d->Wo - 1) * d->out_stride >= d->Wi;
  int cfg = d->tile_cfg;
  if (cfg < 8) {
    const long t128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
    if (a.Cout <= 64) cfg = 9;
    else if (t128 >= 384) cfg = 8;
    else cfg = 9;
  }
  if (cfg >= 8 && cfg <= 11) return pxl_dma_launch_a(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 16 && cfg <= 19) return pxl_dma_launch_b(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 20 && cfg <= 23) return pxl_dma_launch_c(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 24 && cfg <= 27) return pxl_dma_launch_d(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 28 && cfg <= 31) return pxl_dma_launch_e(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 32 && cfg <= 35) return pxl_dma_launch_f(cfg, a, gather, sk, ws_bytes, s);
  switch (cfg) {
    case 12: return launch_dma<128, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 13: return launch_dma<128, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 14: return launch_dma<64, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 15: return launch_dma<64, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    // timing ablations (garbage results): 64x128 3-stage gather kernel, 100 + ABL bits
    case 100: return launch_abl<64, 128, 3, 0>(a, s);
    case 101: return launch_abl<64, 128, 3, 1>(a, s);
    case 102: return launch_abl<64, 128, 3, 2>(a, s);
    case 103: return launch_abl<64, 128, 3, 3>(a, s);
    case 104: return launch_abl<64, 128, 3, 4>(a, s);
    case 106: return launch_abl<64, 128, 3, 6>(a, s);
    case 107: return launch_abl<64, 128, 3, 7>(a, s);
    case 115: return launch_abl<64, 128, 3, 15>(a, s);
    case 116: return launch_abl<64, 128, 3, 16>(a, s);
    case 131: return launch_abl<64, 128, 3, 31>(a, s);
    case 132: return launch_abl<64, 128, 3, 32>(a, s);
    case 147: return launch_abl<64, 128, 3, 47>(a, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
}  // namespace
