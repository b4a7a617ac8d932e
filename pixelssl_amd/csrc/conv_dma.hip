// LDS-DMA implicit-GEMM convolution for gfx950: C-ABI entry points and tile-configuration dispatch.  The kernel itself is
// conv_dma_kernel.h; its tile configurations are instantiated in conv_dma_a .. d.hip (parallel compilation) and here (12..15,
// the timing ablations).
#include <cstring>
#include <new>

#include "conv_dma_kernel.h"

using namespace pxl_dma;

// 1 if the descriptor / operand combination can run on the LDS-DMA kernel
extern "C" int pxl_conv_dma_eligible(const pxl_conv_desc* d, const float* in_scale, const void* workspace) {
  (void)workspace;
  if ((d->dtype != PXL_BF16 && d->dtype != PXL_F32) || in_scale != nullptr) return 0;
  // fp32 operands: conv_dma_f32.hip (PXL_F32_DMA=0 keeps the fp32 engine on the generic kernels, for A/B runs)
  static const bool f32_on = getenv("PXL_F32_DMA") == nullptr || getenv("PXL_F32_DMA")[0] != '0';
  const long es = d->dtype == PXL_F32 ? 4 : 2;
  if (d->dtype == PXL_F32 && !f32_on) return 0;
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
  if ((long)d->Kreal * d->ntaps * d->Cin * es >= (1L << 31)) return 0;
  if ((long)d->B * d->Ho * d->Wo >= (1L << 24)) return 0;      // float-reciprocal pixel decomposition in the prologue
  if (d->dtype == PXL_F32) {
    // 32-bit byte offsets (int arithmetic in the prologue, bit 31 = "out of range")
    if ((long)d->B * d->Hi * d->Wi * d->Cin * es >= (1L << 31)) return 0;
    if ((long)d->B * d->Ho * d->Wo * d->Cout * es >= (1L << 32) - 256) return 0;
  }
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
int pxl_dma_f32_launch(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s);   // conv_dma_f32.hip

extern "C" int pxl_conv_dma(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                            const void* addend, float* stats, void* workspace, size_t ws_bytes, void* stream) {
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = addend; a.stats = stats;
  a.ws = reinterpret_cast<float*>(workspace);
  a.nk_per = 0; a.raw_slabs = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr; a.mask_bits = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  const int sk = d->split_k;
  return conv_dma_launch(d, a, sk, ws_bytes, stream);
}

// The convolution as fp32 partial sums: K in `slices` equal slices (1 = unsplit), slice s stores its [M][Cout] tile at ws + s * M *
// Cout floats with plain stores; NO finish pass, nothing rounded -- the caller consumes the slabs (aspp.hip: pxl_aspp_col2im sums
// them together with the taps).  bf16 LDS-DMA kernel only; PXL_ERR_UNSUPPORTED otherwise (tile without fp32 staging, K not
// divisible, workspace too small).
extern "C" int pxl_conv_dma_slabs(const pxl_conv_desc* d, const void* in, const void* w, float* ws, size_t ws_bytes, int slices,
                                  void* stream) {
  PXL_REQUIRE(d && in && w && ws && slices >= 1, "conv_dma_slabs: bad argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->dtype != PXL_BF16 || (d->tile_cfg >= 0 && d->tile_cfg < 8) || d->div != 1)
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_slabs: descriptor is not eligible for the bf16 LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = nullptr; a.bias = nullptr; a.addend = nullptr; a.stats = nullptr;
  a.ws = ws; a.nk_per = 0; a.raw_slabs = 1;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr; a.mask_bits = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  return conv_dma_launch(d, a, slices, ws_bytes, stream);
}

// Timeline probe of pxl_conv_dma (tools/cbench.cpp, not on the product path): the same launch built with cycle stamps;
// trace = [workgroups][72] uint32 (layout: TRACE_WORDS above).  Plain operands only (no split-K, no BN-on-load).
extern "C" int pxl_conv_dma_trace(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                  float* stats, unsigned* trace, void* stream) {
  PXL_REQUIRE(d && in && w && out && trace, "conv_dma_trace: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->dtype != PXL_BF16 || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_trace: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0; a.raw_slabs = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr; a.mask_bits = 0;
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
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->dtype != PXL_BF16 || (d->tile_cfg >= 0 && d->tile_cfg < 8) || !fin->training)
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_finalize: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0; a.raw_slabs = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr; a.mask_bits = 0;
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
  if (d->dtype == PXL_F32) {       // fp32 (conv_dma_f32.hip): 1x1 / stride-1 launches only
    const bool plain1 = d->ntaps == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->out_stride == 1 && d->Ho == d->Hi && d->Wo == d->Wi;
    if (!plain1) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: the fp32 kernel applies the BatchNorm on load for 1x1 / stride-1 convolutions");
  }
  DmaArgs a;
  a.in = y; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0; a.raw_slabs = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr; a.mask_bits = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin = *bin; a.bin_relu = bin_relu; a.bin_z = z; a.trace = nullptr;
  if (z != nullptr) {          // the activated tensor can only be written by a kernel that walks every input pixel exactly once per tile row
    bool plain = d->ntaps == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->out_stride == 1 && d->Ho == d->Hi && d->Wo == d->Wi;
    // (the halo-tile kernel, tile configurations 40..43, visits every pixel once as the CENTRE of a slab and writes z from there)
    const bool halo = d->tile_cfg >= 40 && d->tile_cfg <= 43 && d->out_stride == 1 && d->Ho == d->Hi && d->Wo == d->Wi;
    if (!plain && !halo) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: z output needs a 1x1 / stride-1 convolution");
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
  a.ws = nullptr; a.nk_per = 0; a.raw_slabs = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = bn_relu; a.bn_mask = nullptr; a.mask_bits = 0;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  // bn_sums holds d->stats_rep replicas [stats_rep][2C] (<= 1: one vector); tile row t adds into replica t % stats_rep.  With at
  // least as many replicas as tile rows every (replica, channel) receives one add per launch: bit-reproducible (PXL_DETERMINISTIC)
  q.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}

// Data gradient that completes the gradient of a residual join's output, with the join's backward fused into the
// epilogue: g = dgrad(dy) (+ addend); din = g * (join_out > 0) -- the ReLU after the join -- and bn_sums[0..C) +=
// sum_m din, bn_sums[C..2C) += sum_m din * xhat(bn_y) for the main branch's last BatchNorm (the one without a ReLU of its
// own).  Replaces pxl_residual_bwd_reduce (3 tensor reads + 2 writes) by two extra reads in this epilogue.
namespace { int joinreduce_impl(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend, const void* join_out,
                                int mask_bits, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream); }
extern "C" int pxl_conv_dgrad_joinreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                         const void* join_out, const void* bn_y, const float* bn_coef, float* bn_sums,
                                         void* stream) {
  return joinreduce_impl(d, dy, wt, din, addend, join_out, 0, bn_y, bn_coef, bn_sums, stream);
}
// ... with the join's ReLU mask as the bit plane pxl_residual_fwd_bits wrote ([M][C / 8] bytes) instead of the join output itself:
// the same results, 1/16 of that operand's bytes.  bf16 LDS-DMA kernel only.
extern "C" int pxl_conv_dgrad_joinreduce_bits(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                              const void* join_bits, const void* bn_y, const float* bn_coef, float* bn_sums,
                                              void* stream) {
  PXL_REQUIRE(d && d->dtype == PXL_BF16 && d->Cout % 8 == 0, "conv_dgrad_joinreduce_bits: bf16 launches with 8-channel chunks only");
  return joinreduce_impl(d, dy, wt, din, addend, join_bits, 1, bn_y, bn_coef, bn_sums, stream);
}
namespace {
int joinreduce_impl(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend, const void* join_out,
                    int mask_bits, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream) {
  PXL_REQUIRE(d && dy && wt && din && join_out && bn_y && bn_coef && bn_sums, "conv_dgrad_joinreduce: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->Kreal != d->Cout || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dgrad_joinreduce: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = dy; a.w = wt; a.out = din; a.bias = nullptr; a.addend = addend; a.stats = bn_sums;
  a.ws = nullptr; a.nk_per = 0; a.raw_slabs = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = 0; a.bn_mask = join_out; a.mask_bits = mask_bits;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  // bn_sums holds d->stats_rep replicas [stats_rep][2C] (<= 1: one vector); tile row t adds into replica t % stats_rep.  With at
  // least as many replicas as tile rows every (replica, channel) receives one add per launch: bit-reproducible (PXL_DETERMINISTIC)
  q.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}
}  // namespace

namespace {
// PXL_S2_CLASSES=0: stride-2 data gradients as ONE launch that walks every tap (rounds 1-3), for A/B runs
bool s2_classes_on() {
  static const bool on = getenv("PXL_S2_CLASSES") == nullptr || getenv("PXL_S2_CLASSES")[0] != '0';
  return on;
}
// (LDS bytes of a tile configuration: the coefficient table of a BN-on-load launch must fit next to the ring -- checked before
// a launch is captured for pairing, where it could no longer fall back)
size_t dma_cfg_ring_bytes(int cfg) {
  static const int BMs[] = {128, 128, 64, 64};                     // 8..19: 128x128, 128x64, 64x128, 64x64
  static const int BNs[] = {128, 64, 128, 64};
  if (cfg >= 8 && cfg <= 19) { const int st = cfg < 12 ? 3 : (cfg < 16 ? 4 : 2); return (size_t)st * (BMs[cfg & 3] + BNs[cfg & 3]) * 128; }
  if (cfg >= 20 && cfg <= 27) { static const int T[] = {96, 160, 192, 128}; return (size_t)(cfg < 24 ? 3 : 2) * (T[cfg & 3] + 128) * 128; }
  switch (cfg) {
    case 28: case 29: return 2 * 192 * 128;
    case 30: case 31: return 2 * 256 * 128;
    case 32: return 3 * 192 * 128;
    case 33: return 3 * 256 * 128;
    case 34: case 35: return 2 * 384 * 128;
    default: return 0;
  }
}

int dma_dispatch(int cfg, const DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups) {
  if (cfg >= 8 && cfg <= 11) return pxl_dma_launch_a(cfg, a, gather, sk, ws_bytes, s, groups);
  if (cfg >= 16 && cfg <= 19) return pxl_dma_launch_b(cfg, a, gather, sk, ws_bytes, s, groups);
  if (cfg >= 20 && cfg <= 23) return pxl_dma_launch_c(cfg, a, gather, sk, ws_bytes, s, groups);
  if (cfg >= 24 && cfg <= 27) return pxl_dma_launch_d(cfg, a, gather, sk, ws_bytes, s, groups);
  if (cfg >= 28 && cfg <= 31) return pxl_dma_launch_e(cfg, a, gather, sk, ws_bytes, s, groups);
  if (cfg >= 32 && cfg <= 35) return pxl_dma_launch_f(cfg, a, gather, sk, ws_bytes, s, groups);
  switch (cfg) {
    case 12: return launch_dma<128, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s, groups);
    case 13: return launch_dma<128, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s, groups);
    case 14: return launch_dma<64, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s, groups);
    case 15: return launch_dma<64, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s, groups);
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

// ---- paired launches: the executor runs two networks with the same program in lockstep (pxl_net_forward_pair) and brackets
// the convolution of each with pxl_dma_capture_begin / _end: inside the bracket a launch that CAN be one half of a pair is
// recorded instead of issued; pxl_dma_launch_captured then issues both halves as ONE launch (gridDim.z = 2) when they
// match, else one after the other.
struct DmaCapture {
  bool armed = false, held = false;
  DmaArgs a; int cfg = 0; bool gather = false; int sk = 1; size_t ws_bytes = 0; hipStream_t stream = nullptr;
};
thread_local DmaCapture* tl_capture = nullptr;

int conv_dma_launch(const pxl_conv_desc* d, DmaArgs& a, int sk, size_t ws_bytes, void* stream) {
  a.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.Kreal = d->Kreal;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.div_shift = d->div == 2 ? 1 : 0;
  a.M = d->B * d->Ho * d->Wo;
  const bool f32 = d->dtype == PXL_F32;
  const size_t es = f32 ? 4 : 2;
  a.Ktot = d->ntaps * d->Cin;
  a.nk = a.Ktot / (f32 ? 32 : 64);            // a K step is one 128-byte row segment
  a.tiles_m = a.tiles_n = 0;
  a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->Cin * es);
  a.w_bytes = (unsigned)((size_t)d->Kreal * a.Ktot * es);
  std::memset(&a.g1, 0, sizeof(a.g1));
  for (int t = 0; t < d->ntaps; ++t)
    if (d->dy[t] < -2047 || d->dy[t] > 2047 || d->dx[t] < -2047 || d->dx[t] > 2047)
      return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: tap offset (%d, %d) outside the 12-bit fields of the tap list", (int)d->dy[t], (int)d->dx[t]);
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? tap_encode(t, d->dy[t], d->dx[t]) : 0;
  a.sub_mul = 1; a.sub_py = a.sub_px = 0; a.out_H = d->Ho; a.out_W = d->Wo;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // taps at offset (0,0) only and every output pixel inside the input: no bounds logic at all
  bool gather = false;
  for (int t = 0; t < d->ntaps; ++t) gather |= d->dy[t] != 0 || d->dx[t] != 0;
  gather |= d->ntaps != 1;
  gather |= d->div != 1;
  gather |= (d->Ho - 1) * d->out_stride >= d->Hi || (d->Wo - 1) * d->out_stride >= d->Wi;
  int cfg = d->tile_cfg;
  if (f32) {
    // conv_dma_f32.hip: plain launches only (the fp32 engine materialises its activations and finalizes on its own)
    if (a.fin.coef != nullptr || a.trace != nullptr)
      return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: in-kernel finalize / trace are bf16 launches");
    if (cfg < 8) cfg = a.Cout <= 64 ? 17 : 18;        // MFMA-bound: the small 2-stage tiles (3 workgroups per CU)
    if (cfg >= 12 && cfg < 16) cfg -= 4;
    if (cfg >= 20) cfg = 16 + (cfg & 3);
  } else if (cfg < 8) {
    const long t128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
    if (a.Cout <= 64) cfg = 9;
    else if (t128 >= 384) cfg = 8;
    else cfg = 9;
  }
  // (bf16 1x1 stride-2: three read-back-only launches cost what the one zero-filled K loop costs -- 45.6 vs 43.0 us, 36.5 vs 38.2:
  // kept as one launch; fp32: 283 -> 127 us)
  if (d->div == 2 && s2_classes_on() && (f32 || d->ntaps >= 4)) {
    // Data gradient of a stride-2 convolution, one launch per output-parity class.  An output pixel (y, x) only receives the
    // taps with (y + dy) and (x + dx) even: 1 / 2 / 2 / 4 of a 3x3's nine, 4 each of a 4x4's sixteen.  The single-launch form
    // walks EVERY tap for every pixel and lets the parity test zero 3 of 4 (harmless while the bf16 loop waits for the DMA
    // path anyway, 4x the MFMA time in fp32 and on the large 4x4 detector layers); here each class is its own sub-grid launch
    // over exactly its taps, writing rows (2 oy + py, 2 ox + px) of the same output tensor.
    int count[4] = {0, 0, 0, 0};
    for (int t = 0; t < d->ntaps; ++t) ++count[(d->dy[t] & 1) * 2 + (d->dx[t] & 1)];
    // class (py, px) takes the taps with dy parity == py and dx parity == px.  A class without taps (three of the four of a 1x1
    // stride-2 convolution) is a launch with a K loop of ZERO steps: the read-back pass alone (zeros + addend, the fused sums)
    if (a.ws == nullptr) {
      const int ksteps_per_tap = a.Cin / (f32 ? 32 : 64);
      for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
          const int Hs = (d->Ho - py + 1) / 2, Ws = (d->Wo - px + 1) / 2;
          if (Hs <= 0 || Ws <= 0) continue;
          DmaArgs c = a;
          c.Ho = Hs; c.Wo = Ws; c.M = d->B * Hs * Ws;
          c.sub_mul = 2; c.sub_py = py; c.sub_px = px;
          int k = 0;
          for (int t = 0; t < d->ntaps; ++t)
            if ((d->dy[t] & 1) == py && (d->dx[t] & 1) == px) c.taps[k++] = tap_encode(t, d->dy[t], d->dx[t]);
          for (int t = k; t < 64; ++t) c.taps[t] = 0;
          c.nk = k * ksteps_per_tap;
          if (k == 0) { c.taps[0] = a.taps[0]; k = 1; }        // (the prologue still issues its first tiles: any valid tap)
          c.ntaps = k;
          const int rc = f32 ? pxl_dma_f32_launch(cfg, c, true, 1, 0, s) : dma_dispatch(cfg, c, true, 1, 0, s, 1);
          if (rc != PXL_OK) return rc;
        }
      return PXL_OK;
    }
  }
  if (f32) return pxl_dma_f32_launch(cfg, a, gather, sk, ws_bytes, s);
  if (cfg >= 40 && cfg <= 43) {      // halo-tile kernel (never split-K, never paired)
    if (a.raw_slabs) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_slabs: not a halo-tile launch");
    return pxl_halo_launch(cfg, a, s);
  }
  DmaCapture* cap = tl_capture;
  if (cap != nullptr && cap->armed && !cap->held && !a.raw_slabs && cfg >= 8 && cfg <= 35 && a.addend == nullptr && a.bn_y == nullptr &&
      a.fin.coef == nullptr && a.trace == nullptr && (a.ws == nullptr || sk == 1 || a.stats != nullptr)) {
    // would this launch split K on its own?  (launch_dma's rule: few tiles and a long reduction, only with a workspace and no
    // statistics) -- those stay single launches
    const bool may_split = a.ws != nullptr && a.stats == nullptr && sk != 1;
    const size_t ring = dma_cfg_ring_bytes(cfg) + (a.bin.coef != nullptr ? (size_t)a.Cin * 8 : 0);
    if (!may_split && ring > 0 && ring <= 156 * 1024) {
      cap->a = a; cap->cfg = cfg; cap->gather = gather; cap->sk = 1; cap->ws_bytes = 0; cap->stream = s;
      cap->a.ws = nullptr;
      cap->held = true;
      return PXL_OK;
    }
  }
  return dma_dispatch(cfg, a, gather, sk, ws_bytes, s, 1);
}
}  // namespace

// Capture bracket of the paired forward pass (csrc/net.cpp: pxl_net_forward_pair).  *slot receives an opaque record (heap,
// released by pxl_dma_launch_captured); a bracket that captured nothing leaves it untouched.
extern "C" void pxl_dma_capture_begin(void** slot) {
  DmaCapture* c = new (std::nothrow) DmaCapture();
  if (c != nullptr) c->armed = true;
  *slot = c;
  tl_capture = c;
}
extern "C" void pxl_dma_capture_end(void) {
  if (tl_capture != nullptr) tl_capture->armed = false;
  tl_capture = nullptr;
}
// Issue what two brackets recorded: ONE launch of both networks' convolution when both were recorded with the same tile
// configuration / geometry / operand combination, else each on its own.  Frees the records.  -> 1 paired, 0 separate, < 0 error
extern "C" int pxl_dma_launch_captured(void* slot0, void* slot1) {
  DmaCapture* c0 = static_cast<DmaCapture*>(slot0);
  DmaCapture* c1 = static_cast<DmaCapture*>(slot1);
  int rc = PXL_OK, paired = 0;
  const bool h0 = c0 != nullptr && c0->held, h1 = c1 != nullptr && c1->held;
  if (h0 && h1) {
    const DmaArgs &x = c0->a, &y = c1->a;
    const bool same = c0->cfg == c1->cfg && c0->gather == c1->gather && c0->stream == c1->stream && x.B == y.B && x.Hi == y.Hi &&
                      x.Wi == y.Wi && x.Cin == y.Cin && x.Ho == y.Ho && x.Wo == y.Wo && x.Cout == y.Cout && x.Kreal == y.Kreal &&
                      x.ntaps == y.ntaps && x.so == y.so && x.div_shift == y.div_shift && x.stats_rep == y.stats_rep &&
                      (x.bias != nullptr) == (y.bias != nullptr) && (x.stats != nullptr) == (y.stats != nullptr) &&
                      (x.bin.coef != nullptr) == (y.bin.coef != nullptr) && x.bin_relu == y.bin_relu &&
                      std::memcmp(x.taps, y.taps, sizeof(x.taps)) == 0 &&          // (bin_z: per network, the kernel tests its own pointer)

                      (x.bin.coef == nullptr || (x.bin.nrep == y.bin.nrep && x.bin.count == y.bin.count && x.bin.training == y.bin.training &&
                                                 x.bin.momentum == y.bin.momentum && x.bin.eps == y.bin.eps && x.bin.clamp_var == y.bin.clamp_var &&
                                                 (x.bin.running_mean != nullptr) == (y.bin.running_mean != nullptr) &&
                                                 (x.bin.gamma != nullptr) == (y.bin.gamma != nullptr) && (x.bin.beta != nullptr) == (y.bin.beta != nullptr)));
    if (same) {
      DmaArgs p = x;
      p.g1.in = y.in; p.g1.w = y.w; p.g1.out = y.out; p.g1.bias = y.bias; p.g1.stats = y.stats;
      p.g1.bin_stats = y.bin.stats; p.g1.bin_gamma = y.bin.gamma; p.g1.bin_beta = y.bin.beta;
      p.g1.bin_rmean = y.bin.running_mean; p.g1.bin_rvar = y.bin.running_var; p.g1.bin_coef = y.bin.coef; p.g1.bin_z = y.bin_z;
      rc = dma_dispatch(c0->cfg, p, c0->gather, 1, 0, c0->stream, 2);
      paired = 1;
    }
  }
  if (!paired) {
    if (h0) rc = dma_dispatch(c0->cfg, c0->a, c0->gather, 1, 0, c0->stream, 1);
    if (h1 && rc == PXL_OK) rc = dma_dispatch(c1->cfg, c1->a, c1->gather, 1, 0, c1->stream, 1);
  }
  delete c0;
  delete c1;
  return rc != PXL_OK ? rc : paired;
}
