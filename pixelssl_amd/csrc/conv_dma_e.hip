// LDS-DMA convolution, tile configurations 28..31: 8-wave workgroups (see conv_dma_kernel.h / conv_dma.hip)
#include "conv_dma_kernel.h"

int pxl_dma_launch_e(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups) {
  using namespace pxl_dma;
  switch (cfg) {
    case 28: return launch_dma<64, 128, 2, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    case 29: return launch_dma<128, 64, 4, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    case 30: return launch_dma<128, 128, 2, 4, 2>(a, gather, sk, ws_bytes, s, groups);
    case 31: return launch_dma<128, 128, 4, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
