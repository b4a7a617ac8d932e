// LDS-DMA convolution, tile configurations 32..35: 8-wave workgroups (see conv_dma_kernel.h / conv_dma.hip)
#include "conv_dma_kernel.h"

int pxl_dma_launch_f(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups) {
  using namespace pxl_dma;
  switch (cfg) {
    case 32: return launch_dma<64, 128, 2, 4, 3>(a, gather, sk, ws_bytes, s, groups);
    case 33: return launch_dma<128, 128, 2, 4, 3>(a, gather, sk, ws_bytes, s, groups);
    case 34: return launch_dma<256, 128, 4, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    case 35: return launch_dma<128, 256, 2, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
