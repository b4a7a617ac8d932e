// Multi-tensor optimizer / EMA updates over the flat fp32 parameter buffers (one launch per
// learning-rate group instead of 320 per-tensor launches).  HBM-bound, float4 per lane.
//   * SGD with momentum + weight decay: torch.optim.SGD semantics (pixelssl/nn/optimizer.py:57-75)
//   * EMA teacher update (ssl_mt.py:359-363)
#include <cmath>

#include "common.h"

namespace {

// the update arithmetic, shared by the stand-alone kernels and the fused update kernel (one source = one contraction choice)
// Written with explicit roundings -- the forms hipcc chose for the stand-alone kernels of rounds 1-4 (three fused multiply-adds
// for SGD; two rounded products and a sum for the EMA, as torch's `mul_().add_()`): left to the compiler, the fused kernel
// contracted the EMA into a multiply + fma and differed from the stand-alone kernel in the last bit.
__device__ __forceinline__ float sgd_elem(float pv, float gv, float bv, float lr, float momentum, float wd, int first, float* b_out) {
  const float d = __fmaf_rn(wd, pv, gv);
  const float b = first ? d : __fmaf_rn(momentum, bv, d);
  *b_out = b;
  return __fmaf_rn(-lr, b, pv);
}
__device__ __forceinline__ float ema_elem(float tv, float sv, float alpha) {
#pragma clang fp contract(off)
  const float a = tv * alpha;
  const float b = (1.f - alpha) * sv;
  return a + b;
}

// lr_dev != NULL: the learning rate is read from device memory (a per-step hyper-parameter block, pxl_hyper_set): the launch
// then carries nothing that changes from step to step and can be replayed from a captured hipGraph
__global__ __launch_bounds__(256) void sgd_kernel(long n, float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, float lr, float momentum,
                                                  float wd, int first, const float* __restrict__ lr_dev) {
  if (lr_dev != nullptr) lr = lr_dev[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float b;
    p[i] = sgd_elem(p[i], g[i], buf[i], lr, momentum, wd, first, &b);
    buf[i] = b;
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
    t[i] = ema_elem(t[i], s[i], alpha);
}

// Fused parameter update of params[lo, hi): SGD (per-run learning rates) -> EMA into the teacher (optional) -> the bf16
// forward-layout copies of both networks' convolution weights (for every convolution whose kernel layout IS the master
// layout, channels_last with Cin % 32 == 0: a cast), and the consumed gradient zeroed -- one pass, 32 B per parameter,
// instead of SGD (20 B) + EMA (12 B) + two packing passes (6 B each) + a memset (4 B).  Same arithmetic as the stand-alone
// kernels (sgd_elem / ema_elem / pack_bf2), element for element.
struct UpdRuns { long start[8]; float lr[8]; const float* lr_dev[8]; int n; };
__global__ __launch_bounds__(256) void sgd_ema_pack_kernel(long lo, long hi, float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ buf, float* __restrict__ t, const UpdRuns runs,
                                                           float momentum, float wd, float alpha, const float* __restrict__ alpha_dev,
                                                           const pxl_upd_seg* __restrict__ segs, int nseg,
                                                           unsigned char* __restrict__ s_pk, unsigned char* __restrict__ t_pk, int zero_grad) {
  if (alpha_dev != nullptr) alpha = alpha_dev[0];
  float lrs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) lrs[k] = k < runs.n ? (runs.lr_dev[k] != nullptr ? runs.lr_dev[k][0] : runs.lr[k]) : 0.f;
  // the segment table in LDS (the binary search below ran ~7 DEPENDENT global loads per float4: 3.8 TB/s), and every block a
  // CONTIGUOUS chunk of the range so that consecutive iterations of a thread stay inside one segment (the search result is kept)
  __shared__ long s_off[512], s_n[512], s_spk[512], s_tpk[512];
  const int ns = nseg < 512 ? nseg : 512;
  for (int k = threadIdx.x; k < ns; k += 256) { s_off[k] = segs[k].off; s_n[k] = segs[k].n; s_spk[k] = segs[k].s_pk; s_tpk[k] = segs[k].t_pk; }
  __syncthreads();
  const long per = (((hi - lo) / 4 + gridDim.x - 1) / gridDim.x + 255) / 256 * 256 * 4;      // elements per block, a multiple of 1024
  const long b0 = lo + (long)blockIdx.x * per, b1 = b0 + per < hi ? b0 + per : hi;
  int cur = 0;
  for (long i = b0 + (long)threadIdx.x * 4; i < b1; i += 1024) {
    float lr = lrs[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < runs.n && i >= runs.start[k]) lr = lrs[k];
    const float4 pv = *reinterpret_cast<const float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i),
                 bv = *reinterpret_cast<const float4*>(buf + i);
    float4 pn, bn;
    pn.x = sgd_elem(pv.x, gv.x, bv.x, lr, momentum, wd, 0, &bn.x);
    pn.y = sgd_elem(pv.y, gv.y, bv.y, lr, momentum, wd, 0, &bn.y);
    pn.z = sgd_elem(pv.z, gv.z, bv.z, lr, momentum, wd, 0, &bn.z);
    pn.w = sgd_elem(pv.w, gv.w, bv.w, lr, momentum, wd, 0, &bn.w);
    *reinterpret_cast<float4*>(p + i) = pn;
    *reinterpret_cast<float4*>(buf + i) = bn;
    if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tn = pn;
    if (t != nullptr) {
      const float4 tv = *reinterpret_cast<const float4*>(t + i);
      tn.x = ema_elem(tv.x, pn.x, alpha); tn.y = ema_elem(tv.y, pn.y, alpha);
      tn.z = ema_elem(tv.z, pn.z, alpha); tn.w = ema_elem(tv.w, pn.w, alpha);
      *reinterpret_cast<float4*>(t + i) = tn;
    }
    if (ns > 0) {
      if (!(i >= s_off[cur] && (cur + 1 >= ns || i < s_off[cur + 1]))) {
        int a = 0, b = ns - 1;
        while (a < b) {                          // last segment with off <= i
          const int mid = (a + b + 1) >> 1;
          if (s_off[mid] <= i) a = mid; else b = mid - 1;
        }
        cur = a;
      }
      const long so = s_off[cur];
      if (i >= so && i < so + s_n[cur]) {
        const long e = (i - so) * 2;
        if (s_pk != nullptr && s_spk[cur] >= 0)
          *reinterpret_cast<uint2*>(s_pk + s_spk[cur] + e) = make_uint2(pack_bf2(pn.x, pn.y), pack_bf2(pn.z, pn.w));
        if (t != nullptr && t_pk != nullptr && s_tpk[cur] >= 0)
          *reinterpret_cast<uint2*>(t_pk + s_tpk[cur] + e) = make_uint2(pack_bf2(tn.x, tn.y), pack_bf2(tn.z, tn.w));
      }
    }
  }
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

// see sgd_ema_pack_kernel.  run_start / run_lr / run_lr_dev: nruns <= 8 learning-rate runs of the flat buffer in ascending order
// (run k covers [run_start[k], run_start[k + 1])); run_lr_dev[k] != NULL: that rate is read from device memory.  t / t_packed NULL:
// no teacher.  segs: DEVICE array of nseg segments sorted by `off` (pxl_net_update_segments).  lo, hi, every segment: multiples of 4.
extern "C" int pxl_sgd_ema_pack(long lo, long hi, float* p, float* g, float* buf, float* t, int nruns, const long* run_start,
                                const float* run_lr, const float* const* run_lr_dev, float momentum, float weight_decay, float alpha,
                                const float* alpha_dev, const pxl_upd_seg* segs, int nseg, void* s_packed, void* t_packed,
                                int zero_grad, void* stream) {
  PXL_REQUIRE(p && g && buf && hi > lo && lo >= 0 && (lo & 3) == 0 && (hi & 3) == 0, "sgd_ema_pack: bad range");
  PXL_REQUIRE(nruns >= 1 && nruns <= 8 && run_start && run_lr, "sgd_ema_pack: 1..8 learning-rate runs");
  PXL_REQUIRE(nseg == 0 || segs != nullptr, "sgd_ema_pack: null segment table");
  PXL_REQUIRE(nseg <= 512, "sgd_ema_pack: at most 512 segments");
  UpdRuns r;
  for (int k = 0; k < 8; ++k) {
    r.start[k] = k < nruns ? run_start[k] : 0;
    r.lr[k] = k < nruns ? run_lr[k] : 0.f;
    r.lr_dev[k] = (k < nruns && run_lr_dev != nullptr) ? run_lr_dev[k] : nullptr;
  }
  r.n = nruns;
  long blocks = ((hi - lo) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sgd_ema_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), lo, hi, p, g, buf,
                     t, r, momentum, weight_decay, alpha, alpha_dev, segs, nseg, reinterpret_cast<unsigned char*>(s_packed),
                     reinterpret_cast<unsigned char*>(t_packed), zero_grad);
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
