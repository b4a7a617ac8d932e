// LDS-DMA convolution, tile configurations 16..19 (see conv_dma_kernel.h / conv_dma.hip)
#include "conv_dma_kernel.h"

int pxl_dma_launch_b(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups) {
  using namespace pxl_dma;
  switch (cfg) {
    case 16: return launch_dma<128, 128, 2, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    case 17: return launch_dma<128, 64, 2, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    case 18: return launch_dma<64, 128, 2, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    case 19: return launch_dma<64, 64, 2, 2, 2>(a, gather, sk, ws_bytes, s, groups);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
