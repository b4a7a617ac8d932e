// Halo-tile convolution (conv_halo_kernel.h): host side.  Tile configurations 40..43 of the LDS-DMA entry points
// (pxl_conv_igemm / pxl_conv_dma_finalize / pxl_conv_dma_bnin / pxl_conv_dgrad_bnreduce): 40 = 128x128 and 41 = 128(pixels)x64
// with eight waves, 42 / 43 the same with four; every other launch parameter -- slab size, LDS bytes -- follows from the geometry.
#include <algorithm>
#include <cstring>

#include "conv_halo_kernel.h"

using namespace pxl_halo;

namespace {

template <int BM, int BN, int WM, int WN, bool BNIN, int EM>
int launch_one(dim3 grid, size_t smem, hipStream_t stream, const HaloArgs& p) {
  static bool raised = false;
  if (!raised) {
    PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<BM, BN, WM, WN, 3, BNIN, EM>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    raised = true;
  }
  hipLaunchKernelGGL((conv_halo_kernel<BM, BN, WM, WN, 3, BNIN, EM>), grid, dim3(WM * WN * 64), smem, stream, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_halo(const DmaArgs& a, int d, hipStream_t stream) {
  constexpr int NW = WM * WN, NSTW = 3;
  constexpr int TP = BN * 2 + 16, RPP = NW * 64 / (BN / 8);
  HaloArgs p;
  std::memset(&p, 0, sizeof(p));
  p.in = a.in; p.w = a.w; p.out = a.out; p.bias = a.bias; p.addend = a.addend; p.stats = a.stats; p.stats_rep = a.stats_rep;
  p.bn_y = a.bn_y; p.bn_coef = a.bn_coef; p.bn_relu = a.bn_relu; p.bn_mask = a.bn_mask;
  p.B = a.B; p.H = a.Hi; p.W = a.Wi; p.Cin = a.Cin; p.Cout = a.Cout; p.Kreal = a.Kreal;
  p.ntaps = a.ntaps; p.Ktot = a.Ktot;
  p.Hp = a.Hi + d; p.Wp = a.Wi + d;
  const long mv = (long)a.B * p.Hp * p.Wp;
  p.hoff = d * p.Wp + d;
  p.AG = cdiv(BM + 2 * p.hoff, 8);
  p.nslab = a.Cin / 64;
  if (mv + BM + p.hoff >= (1L << 24)) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_halo: padded grid too large");
  p.Mv = (int)mv;
  p.tiles_m = cdiv(p.Mv, BM); p.tiles_n = cdiv(a.Cout, BN);
  p.fin = a.fin; p.fin_counter = a.fin_counter;
  p.bin = a.bin; p.bin_relu = a.bin_relu; p.bin_z = a.bin_z;
  p.in_bytes = a.in_bytes; p.w_bytes = a.w_bytes;
  p.out_bytes = (unsigned)((size_t)a.B * a.Hi * a.Wi * a.Cout * 2);
  for (int t = 0; t < 16; ++t) p.taps[t] = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    const int tp = a.taps[t];
    const int dy = tap_dy(tp), dx = tap_dx(tp);           // (host: the __device__ helpers are plain bit arithmetic)
    p.taps[t] = (int)((tap_wt(tp) << 24) | (unsigned)((dy + d) * p.Wp + (dx + d)));
  }
  // the next slab (AG pieces) is issued underneath the ntaps steps of the current one, one piece per wave and step
  if (p.nslab > 1 && p.AG > a.ntaps * NW) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_halo: slab too large for this tile");
  const bool bnin = a.bin.coef != nullptr;
  size_t smem = (size_t)(p.nslab > 1 ? 2 : 1) * p.AG * 1024 + (size_t)NSTW * BN * 128 + (bnin ? (size_t)a.Cin * 8 : 0);
  smem = std::max(smem, (size_t)BM * TP + (size_t)RPP * 2 * BN * 4);      // the staged output tile + the statistics partials reuse it
  if (smem > 156 * 1024) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_halo: slab + weight ring exceed the LDS");
  const bool has_stats = a.stats != nullptr, has_bnr = has_stats && a.bn_y != nullptr, has_mask = has_bnr && a.bn_mask != nullptr;
  const int em = (a.addend ? 1 : 0) | (a.bias ? 2 : 0) | (has_stats ? 4 : 0) | (has_bnr ? 8 : 0) | (has_mask ? 16 : 0) |
                 ((has_bnr && !has_mask && a.bn_relu) ? 32 : 0);
  const dim3 g(p.tiles_m * p.tiles_n);
#define PXL_H(E) case E: return launch_one<BM, BN, WM, WN, false, E>(g, smem, stream, p);
#define PXL_HB(E) case E: return launch_one<BM, BN, WM, WN, true, E>(g, smem, stream, p);
  if (!bnin) switch (em) { PXL_H(0) PXL_H(4) PXL_H(12) PXL_H(13) PXL_H(44) PXL_H(45) default: break; }
  else switch (em) { PXL_HB(0) PXL_HB(4) default: break; }
#undef PXL_H
#undef PXL_HB
  return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_halo: operand combination %d is not instantiated", em);
}

}  // namespace

// 1 if the launch described by `a` (conv_dma.hip: conv_dma_launch has filled the geometry) can run on the halo-tile kernel:
// a "same" stride-1 multi-tap convolution over 64-channel slabs, no split-K / sub-grid / paired / traced launch
int pxl_halo_reach(const pxl_dma::DmaArgs& a) {
  if (a.ntaps < 2 || a.ntaps > 16 || a.so != 1 || a.div_shift != 0 || a.Ho != a.Hi || a.Wo != a.Wi || a.sub_mul != 1) return 0;
  if (a.Cin % 64 != 0 || a.trace != nullptr) return 0;
  int d = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    const int tp = a.taps[t];
    const int dy = tap_dy(tp), dx = tap_dx(tp);
    d = std::max(d, std::max(dy < 0 ? -dy : dy, dx < 0 ? -dx : dx));
  }
  return d;
}

int pxl_halo_launch(int cfg, const pxl_dma::DmaArgs& a, hipStream_t s) {
  const int d = pxl_halo_reach(a);
  if (d < 1) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_halo: not a same-size stride-1 multi-tap convolution");
  switch (cfg) {
    case 40: return launch_halo<128, 128, 2, 4>(a, d, s);
    case 41: return launch_halo<128, 64, 4, 2>(a, d, s);
    case 42: return launch_halo<128, 128, 2, 2>(a, d, s);
    case 43: return launch_halo<128, 64, 2, 2>(a, d, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_halo: unknown tile config %d", cfg);
  }
}
