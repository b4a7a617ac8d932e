// LDS-DMA convolution, tile configurations 24..27 (see conv_dma_kernel.h / conv_dma.hip)
#include "conv_dma_kernel.h"

int pxl_dma_launch_d(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups) {
  using namespace pxl_dma;
  switch (cfg) {
    case 24: return launch_dma<96, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    case 25: return launch_dma<160, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    case 26: return launch_dma<192, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    case 27: return launch_dma<128, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
