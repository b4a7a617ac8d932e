// Multi-tensor optimizer / EMA updates over the flat fp32 parameter buffers (one launch per
// learning-rate group instead of 320 per-tensor launches).  HBM-bound, float4 per lane.
//   * SGD with momentum + weight decay: torch.optim.SGD semantics (pixelssl/nn/optimizer.py:57-75)
//   * EMA teacher update (ssl_mt.py:359-363)
#include <cmath>

#include "common.h"

namespace {

// lr_dev != NULL: the learning rate is read from device memory (a per-step hyper-parameter block, pxl_hyper_set): the launch
// then carries nothing that changes from step to step and can be replayed from a captured hipGraph
__global__ __launch_bounds__(256) void sgd_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, float lr, float momentum,
                                                  float wd, int first, const float* __restrict__ lr_dev) {
  if (lr_dev != nullptr) lr = lr_dev[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    const float d = g[i] + wd * pv;
    const float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}

// the full torch.optim.SGD update (torch/optim/sgd.py _single_tensor_sgd): dampening scales the gradient entering the
// momentum buffer (not on the first step, where the buffer is a copy of d), nesterov steps along d + momentum * buf
__global__ __launch_bounds__(256) void sgd_general_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ buf, float lr, float momentum, float dampening,
                                                          float wd, int nesterov, int first) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pv = p[i];
    float d = g[i] + wd * pv;
    if (momentum != 0.f) {
      const float b = first ? d : momentum * buf[i] + (1.f - dampening) * d;
      buf[i] = b;
      d = nesterov ? d + momentum * b : b;
    }
    p[i] = pv - lr * d;
  }
}

__global__ __launch_bounds__(256) void ema_kernel(long n, float* __restrict__ t, const float* __restrict__ s,
                                                  float alpha, const float* __restrict__ alpha_dev) {
  if (alpha_dev != nullptr) alpha = alpha_dev[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    t[i] = t[i] * alpha + (1.f - alpha) * s[i];
}

// torch.optim.Adam (no weight decay, no amsgrad): the FC discriminator / flaw detector optimizer (ssl_adv.py:101-102)
__global__ __launch_bounds__(256) void adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float step_size,
                                                   float beta1, float beta2, float eps, float inv_sqrt_bc2, float wd) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = wd != 0.f ? g[i] + wd * p[i] : g[i];        // L2 weight decay folded into the gradient (torch.optim.Adam)
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

__global__ __launch_bounds__(256) void scale_kernel(long n, float* __restrict__ x, float a) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int pxl_sgd_step(long n, float* p, const float* g, float* buf, float lr, float momentum,
                            float weight_decay, int first_step, void* stream) {
  PXL_REQUIRE(p && g && buf && n > 0, "sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g,
                     buf, lr, momentum, weight_decay, first_step, (const float*)nullptr);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

// pxl_sgd_step with the learning rate in DEVICE memory (*lr_dev, written by pxl_hyper_set before the step): same arithmetic
extern "C" int pxl_sgd_step_hp(long n, float* p, const float* g, float* buf, const float* lr_dev, float momentum,
                               float weight_decay, int first_step, void* stream) {
  PXL_REQUIRE(p && g && buf && lr_dev && n > 0, "sgd_step_hp: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g,
                     buf, 0.f, momentum, weight_decay, first_step, lr_dev);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
struct HyperVals { float v[32]; };
__global__ void hyper_set_kernel(float* __restrict__ dst, HyperVals h, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = h.v[threadIdx.x];
}
}  // namespace

// dst[0..n) <- vals[0..n) (n <= 32), `vals` a HOST array whose content is copied into the launch's arguments at enqueue time:
// no staging buffer that a host running several steps ahead of the device could overwrite.  The per-step scalars of a captured
// training step (learning rates, EMA coefficient, ramp-up weight: lrer.py:143-179, ssl_mt.py:359-363, nn/func.py:44-52) go through
// here, eagerly, right before the graph launch; the captured kernels read them from `dst`.
extern "C" int pxl_hyper_set(float* dst, const float* vals, int n, void* stream) {
  PXL_REQUIRE(dst && vals && n >= 1 && n <= 32, "hyper_set: bad argument (1 <= n <= 32)");
  HyperVals h;
  for (int i = 0; i < 32; ++i) h.v[i] = i < n ? vals[i] : 0.f;
  hipLaunchKernelGGL(hyper_set_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), dst, h, n);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_sgd_step_general(long n, float* p, const float* g, float* buf, float lr, float momentum, float dampening,
                                    float weight_decay, int nesterov, int first_step, void* stream) {
  PXL_REQUIRE(p && g && buf && n > 0, "sgd_step_general: bad argument");
  PXL_REQUIRE(!nesterov || (momentum > 0.f && dampening == 0.f), "sgd_step_general: nesterov needs momentum > 0 and dampening = 0");
  hipLaunchKernelGGL(sgd_general_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g, buf,
                     lr, momentum, dampening, weight_decay, nesterov, first_step);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ema_update(long n, float* teacher, const float* student, float alpha, void* stream) {
  PXL_REQUIRE(teacher && student && n > 0, "ema_update: bad argument");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n,
                     teacher, student, alpha, (const float*)nullptr);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

// pxl_ema_update with the coefficient in DEVICE memory (*alpha_dev): same arithmetic
extern "C" int pxl_ema_update_hp(long n, float* teacher, const float* student, const float* alpha_dev, void* stream) {
  PXL_REQUIRE(teacher && student && alpha_dev && n > 0, "ema_update_hp: bad argument");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n,
                     teacher, student, 0.f, alpha_dev);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_adam_step_wd(long n, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, void* stream);

extern "C" int pxl_adam_step(long n, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float lr,
                             float beta1, float beta2, float eps, int step, void* stream) {
  return pxl_adam_step_wd(n, p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, 0.f, step, stream);
}

extern "C" int pxl_adam_step_wd(long n, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, void* stream) {
  PXL_REQUIRE(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad argument (step counts from 1)");
  // torch: denom = sqrt(v) / sqrt(1 - beta2^t) + eps ; p -= lr / (1 - beta1^t) * m / denom
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g,
                     exp_avg, exp_avg_sq, (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), weight_decay);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_scale_inplace(long n, float* x, float a, void* stream) {
  PXL_REQUIRE(x && n > 0, "scale_inplace: bad argument");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, x, a);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
