// Validation metrics of the sseg task on the device (task/sseg/func.py:36-80): channel arg-max of the activated
// prediction and the confusion matrix  cm[gt][pred] += 1  over the pixels with 0 <= gt < num_classes.  Integer work,
// HBM-bound (one pass over C planes of the prediction); bit-exact against numpy: np.argmax semantics (first maximum,
// NaN counts as the maximum), `(gt >= 0) & (gt < C)` mask, `gt.astype(int)` truncation.
// The reference moves the whole prediction to the host and runs numpy per batch; here only the C*C counters leave HBM.
#include "common.h"

namespace {

constexpr int MAXC = 32;

__device__ __forceinline__ int argmax_channels(const float* __restrict__ p, int C, size_t HW, size_t pix) {
  float best = p[pix];
  int idx = 0;
  for (int c = 1; c < C; ++c) {
    const float v = p[(size_t)c * HW + pix];
    if (v > best || (v != v && best == best)) { best = v; idx = c; }
  }
  return idx;
}

// grid (blocks over pixels, N); each block keeps a C*C histogram in LDS and flushes the non-zero bins with 64-bit atomics
__global__ __launch_bounds__(256) void confusion_kernel(int C, long HW, const float* __restrict__ pred,
                                                        const float* __restrict__ gt, unsigned long long* __restrict__ cm) {
  __shared__ unsigned int hist[MAXC * MAXC];
  for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0u;
  __syncthreads();
  const int n = blockIdx.y;
  const float* p = pred + (size_t)n * C * HW;
  const float* g = gt + (size_t)n * HW;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < HW; pix += (long)gridDim.x * 256) {
    const float gv = g[pix];
    if (!(gv >= 0.f && gv < (float)C)) continue;          // also rejects NaN labels
    const int label = (int)gv;                            // astype('int'): truncation
    const int a = argmax_channels(p, C, (size_t)HW, (size_t)pix);
    atomicAdd(&hist[label * C + a], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (hist[i]) atomicAdd(cm + i, (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(256) void argmax_u8_kernel(int C, long HW, const float* __restrict__ pred,
                                                        unsigned char* __restrict__ out) {
  const int n = blockIdx.y;
  const float* p = pred + (size_t)n * C * HW;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < HW; pix += (long)gridDim.x * 256)
    out[(size_t)n * HW + pix] = (unsigned char)argmax_channels(p, C, (size_t)HW, (size_t)pix);
}

inline int pix_blocks(long HW) {
  long b = (HW + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" int pxl_confusion_matrix(int N, int C, long HW, const float* pred, const float* gt, long long* cm, void* stream) {
  PXL_REQUIRE(pred && gt && cm && N > 0 && HW > 0, "confusion_matrix: bad argument");
  PXL_REQUIRE(C >= 1 && C <= MAXC, "confusion_matrix: %d classes (max %d)", C, MAXC);
  hipLaunchKernelGGL(confusion_kernel, dim3(pix_blocks(HW), N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW,
                     pred, gt, reinterpret_cast<unsigned long long*>(cm));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_argmax_u8(int N, int C, long HW, const float* pred, unsigned char* out, void* stream) {
  PXL_REQUIRE(pred && out && N > 0 && HW > 0 && C >= 1 && C <= 256, "argmax_u8: bad argument");
  hipLaunchKernelGGL(argmax_u8_kernel, dim3(pix_blocks(HW), N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW,
                     pred, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
