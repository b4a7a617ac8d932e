// Per-pixel losses of the SSL algorithms on full-resolution NCHW fp32 predictions (HBM-bound;
// consecutive lanes walk the pixel axis so each channel plane is read in coalesced lines, the
// per-sample / global reductions use wave shuffles + one atomic per block).
//   * cross-entropy with ignore_index, mean over ALL H*W pixels per sample (task/sseg/criterion.py:24-38)
//   * mean-squared error (nn.MSELoss(), ssl_mt.py:115,182-184)
#include "common.h"

// PXL_DETERMINISTIC=1 (read per call: these entry points have no executor to remember it): the loss VALUES -- per-sample sums that
// the default launches combine with one fp32 atomic per block or wave -- come from ONE block (one wave for the wave-level kernels)
// per sample, i.e. one add in a fixed order.  Gradients never depended on these sums; this makes the logged numbers bit-reproducible.
static inline bool pxl_det_now() { const char* e = getenv("PXL_DETERMINISTIC"); return e != nullptr && e[0] == '1'; }

namespace {

constexpr int MAXC = 32;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;   // valid on thread 0
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(int C, int HW, const float* __restrict__ logits,
                                                     const float* __restrict__ gt, int ignore_index,
                                                     float* __restrict__ loss) {
  __shared__ float red[4];
  const int n = blockIdx.y;
  const float* lg = logits + (size_t)n * C * HW;
  float acc = 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    // label and channel planes are requested together (the label used to gate the plane loads: two dependent HBM
    // round trips per pixel)
    const float lab_f = gt[(size_t)n * HW + p];
    float v[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) v[c] = lg[(size_t)c * HW + p];
    const int label = (int)lab_f;
    if (label == ignore_index || label < 0 || label >= C) continue;
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) mx = fmaxf(mx, v[c]);
    float sum = 0.f, pick = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { sum += expf(v[c] - mx); if (c == label) pick = v[c]; }
    acc += (mx + logf(sum)) - pick;
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss + n, tot / (float)HW);
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(int C, int HW, const float* __restrict__ logits,
                                                     const float* __restrict__ gt, int ignore_index,
                                                     const float* __restrict__ gout, float* __restrict__ dlogits) {
  const int n = blockIdx.y;
  const float* lg = logits + (size_t)n * C * HW;
  float* dl = dlogits + (size_t)n * C * HW;
  const float gs = gout[n] / (float)HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const int label = (int)gt[(size_t)n * HW + p];
    const bool valid = !(label == ignore_index || label < 0 || label >= C);
    float v[MAXC];
    float mx = -INFINITY;
    if (valid) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { v[c] = lg[(size_t)c * HW + p]; mx = fmaxf(mx, v[c]); }
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
      const float inv = gs / sum;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) dl[(size_t)c * HW + p] = __fmaf_rn(v[c], inv, c == label ? -gs : 0.f);   // (one rounding, spelled out: pxl_ce_mse_bwd must match bit for bit)
    } else {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) dl[(size_t)c * HW + p] = 0.f;
    }
  }
}

// d(task loss + consistency loss)/d(logits) in ONE pass over the [N][C][HW] logits of a mean-teacher style step:
// samples n < n_ce carry the cross-entropy term of ce_bwd_kernel (gt holds n_ce maps), samples mse_lo <= n < mse_hi the
// term s * (logits - target), s = g_mse[0] * two_inv_n, of mse_bwd_kernel; every element is written exactly once (zeros
// where neither applies).  Same expressions as the two separate kernels followed by autograd's add (each term bit-identical; their sum within 1 ulp).
__global__ __launch_bounds__(256) void ce_mse_bwd_kernel(int C, int HW, const float* __restrict__ logits,
                                                         const float* __restrict__ gt, int ignore_index, int n_ce,
                                                         const float* __restrict__ g_ce, const float* __restrict__ target,
                                                         int mse_lo, int mse_hi, const float* __restrict__ g_mse,
                                                         float two_inv_n, float* __restrict__ dlogits) {
  const int n = blockIdx.y;
  const float* lg = logits + (size_t)n * C * HW;
  float* dl = dlogits + (size_t)n * C * HW;
  const bool ce = n < n_ce && g_ce != nullptr;
  const bool mse = n >= mse_lo && n < mse_hi && g_mse != nullptr;
  const float* tg = mse ? target + (size_t)n * C * HW : nullptr;
  const float gs = ce ? g_ce[n] / (float)HW : 0.f;
  const float ms = mse ? g_mse[0] * two_inv_n : 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    bool valid = false;
    int label = -1;
    if (ce) {
      label = (int)gt[(size_t)n * HW + p];
      valid = !(label == ignore_index || label < 0 || label >= C);
    }
    float v[MAXC], x[MAXC];
    if (valid || mse) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) x[c] = lg[(size_t)c * HW + p];
    }
    float inv = 0.f;
    if (valid) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) mx = fmaxf(mx, x[c]);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { v[c] = expf(x[c] - mx); sum += v[c]; }
      inv = gs / sum;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        float g = valid ? __fmaf_rn(v[c], inv, c == label ? -gs : 0.f) : 0.f;
        if (mse) {
          // explicit roundings: no fma contraction across the two terms
          const float m = __fmul_rn(ms, x[c] - tg[(size_t)c * HW + p]);
          g = ce ? __fadd_rn(g, m) : m;
        }
        dl[(size_t)c * HW + p] = g;
      }
  }
}

__global__ __launch_bounds__(256) void mse_fwd_kernel(long n4, long n, const float* __restrict__ a,
                                                      const float* __restrict__ b, float inv_n,
                                                      float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
    acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    acc += d * d;
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, tot * inv_n);
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(long n4, long n, const float* __restrict__ a,
                                                      const float* __restrict__ b,
                                                      const float* __restrict__ gout, float two_inv_n,
                                                      float* __restrict__ da) {
  const float s = gout[0] * two_inv_n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(da)[i] = make_float4(s * (x.x - y.x), s * (x.y - y.y), s * (x.z - y.z), s * (x.w - y.w));
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    da[i] = s * (a[i] - b[i]);
}

}  // namespace

extern "C" int pxl_ce_fwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index,
                          float* loss, void* stream) {
  PXL_REQUIRE(logits && gt && loss && N > 0, "ce_fwd: bad argument");
  PXL_REQUIRE(C >= 1 && C <= MAXC, "ce_fwd: C=%d unsupported (max %d)", C, MAXC);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float) * N, s));
  int gx = cdiv(HW, 256);
  if (gx > 512) gx = 512;
  if (pxl_det_now()) gx = 1;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(gx, N), dim3(256), 0, s, C, HW, logits, gt, ignore_index, loss);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ce_mse_bwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index, int n_ce,
                              const float* g_ce, const float* target, int mse_lo, int mse_hi, const float* g_mse,
                              float* dlogits, void* stream) {
  PXL_REQUIRE(logits && dlogits && N > 0 && n_ce >= 0 && n_ce <= N, "ce_mse_bwd: bad argument");
  PXL_REQUIRE(C >= 1 && C <= MAXC, "ce_mse_bwd: C=%d unsupported (max %d)", C, MAXC);
  PXL_REQUIRE(g_ce == nullptr || n_ce == 0 || gt != nullptr, "ce_mse_bwd: cross-entropy term without label maps");
  PXL_REQUIRE(mse_lo >= 0 && mse_lo <= mse_hi && mse_hi <= N, "ce_mse_bwd: bad consistency range [%d, %d)", mse_lo, mse_hi);
  PXL_REQUIRE(g_mse == nullptr || mse_hi == mse_lo || target != nullptr, "ce_mse_bwd: consistency term without a target");
  int gx = cdiv(HW, 256);
  if (gx > 1024) gx = 1024;
  const long cnt = (long)(mse_hi - mse_lo) * C * HW;
  hipLaunchKernelGGL(ce_mse_bwd_kernel, dim3(gx, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW, logits, gt,
                     ignore_index, n_ce, g_ce, target, mse_lo, mse_hi, g_mse, cnt > 0 ? 2.0f / (float)cnt : 0.f, dlogits);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ce_bwd(int N, int C, int HW, const float* logits, const float* gt, int ignore_index,
                          const float* gout, float* dlogits, void* stream) {
  PXL_REQUIRE(logits && gt && gout && dlogits && N > 0, "ce_bwd: bad argument");
  PXL_REQUIRE(C >= 1 && C <= MAXC, "ce_bwd: C=%d unsupported (max %d)", C, MAXC);
  int gx = cdiv(HW, 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(gx, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW,
                     logits, gt, ignore_index, gout, dlogits);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
// FCDiscriminatorCriterion (ssl_adv.py:496-503) fused with ssladv_preprocess_fcd_criterion (task/sseg/func.py:137-157):
// per pixel m = (task_gt == NULL || task_gt != ignore), x' = x*m, t' = target*m (target = 1 real / 0 fake);
// loss[b] = mean over ALL pixels of BCEWithLogits(x', t')  (masked pixels contribute log 2 like in the reference);
// dx = m * (sigmoid(x') - t') / HW * gout[b]
__global__ void bce_masked_fwd_kernel(long HW, const float* __restrict__ x, const float* __restrict__ task_gt,
                                      int ignore, float target, float* __restrict__ loss) {
  const int b = blockIdx.y;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float m = (task_gt == nullptr || (int)task_gt[b * HW + i] != ignore) ? 1.f : 0.f;
    const float xv = x[b * HW + i] * m, t = target * m;
    s += fmaxf(xv, 0.f) - xv * t + log1pf(expf(-fabsf(xv)));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss + b, s / (float)HW);
}
__global__ void bce_masked_bwd_kernel(long HW, const float* __restrict__ x, const float* __restrict__ task_gt,
                                      int ignore, float target, const float* __restrict__ gout,
                                      float* __restrict__ dx) {
  const int b = blockIdx.y;
  const float k = gout[b] / (float)HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float m = (task_gt == nullptr || (int)task_gt[b * HW + i] != ignore) ? 1.f : 0.f;
    const float xv = x[b * HW + i] * m, t = target * m;
    dx[b * HW + i] = m * (1.f / (1.f + expf(-xv)) - t) * k;
  }
}
}  // namespace

namespace {
// CutMix (ssl_cutmix.py:193-201, 424-430): out = mask*a + (1-mask)*b with a box mask [B][1][HW] broadcast over C;
// optionally counts the pixels whose mixed maximum over channels exceeds `thr` (the confidence numerator).
__global__ void cutmix_mix_kernel(int B, int C, long HW, const float* __restrict__ mask, const float* __restrict__ a,
                                  const float* __restrict__ b, float* __restrict__ out, float thr,
                                  float* __restrict__ count) {
  const long total = (long)B * HW;
  float cnt = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long bb = i / HW, p = i - bb * HW;
    const float m = mask[i];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
      const long o = (bb * C + c) * HW + p;
      const float v = m * a[o] + (1.f - m) * b[o];
      out[o] = v;
      mx = fmaxf(mx, v);
    }
    cnt += mx > thr ? 1.f : 0.f;
  }
  if (count != nullptr) {
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt != 0.f) atomicAdd(count, cnt);
  }
}
}  // namespace

extern "C" int pxl_cutmix_mix(int B, int C, long HW, const float* mask, const float* a, const float* b, float* out,
                              float threshold, float* count, void* stream) {
  PXL_REQUIRE(mask && a && b && out && B > 0 && C > 0 && HW > 0, "cutmix_mix: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (count) PXL_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(float), s));
  long g = ((long)B * HW + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(cutmix_mix_kernel, dim3((int)g), dim3(256), 0, s, B, C, HW, mask, a, b, out, threshold, count);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bce_logits_masked_fwd(int B, long HW, const float* x, const float* task_gt, int ignore_index,
                                         float target, float* loss, void* stream) {
  PXL_REQUIRE(x && loss && B > 0 && HW > 0, "bce_logits_masked_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(loss, 0, (size_t)B * sizeof(float), s));
  const bool det = pxl_det_now();
  const int gx = det ? 1 : (int)((HW + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(bce_masked_fwd_kernel, dim3(gx, B), dim3(det ? 64 : 256), 0, s, HW, x, task_gt, ignore_index, target, loss);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bce_logits_masked_bwd(int B, long HW, const float* x, const float* task_gt, int ignore_index,
                                         float target, const float* gout, float* dx, void* stream) {
  PXL_REQUIRE(x && gout && dx && B > 0 && HW > 0, "bce_logits_masked_bwd: bad argument");
  const int gx = (int)((HW + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(bce_masked_bwd_kernel, dim3(gx, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), HW, x,
                     task_gt, ignore_index, target, gout, dx);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
// ssladv_preprocess_fcd_criterion as the reference returns it (task/sseg/func.py:137-157): two TENSORS,
// pred_out = x * m and gt_out = target * m with m = (task_gt == NULL || task_gt != ignore).  One pass; backward of the
// first output is dx = dout * m.
__global__ void fcd_prepare_kernel(long total, const float* __restrict__ x, const float* __restrict__ task_gt, int ignore,
                                   float target, float* __restrict__ pred_out, float* __restrict__ gt_out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float m = (task_gt == nullptr || (int)task_gt[i] != ignore) ? 1.f : 0.f;
    if (pred_out) pred_out[i] = x[i] * m;
    if (gt_out) gt_out[i] = target * m;
  }
}
// FCDiscriminatorCriterion on plain tensors (ssl_adv.py:496-503): loss[b] = mean_i BCEWithLogits(x, t)
__global__ void bce_logits_fwd_kernel(long HW, const float* __restrict__ x, const float* __restrict__ t,
                                      float* __restrict__ loss) {
  const int b = blockIdx.y;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float xv = x[b * HW + i], tv = t[b * HW + i];
    s += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(loss + b, s / (float)HW);
}
__global__ void bce_logits_bwd_kernel(long HW, const float* __restrict__ x, const float* __restrict__ t,
                                      const float* __restrict__ gout, float* __restrict__ dx) {
  const int b = blockIdx.y;
  const float k = gout[b] / (float)HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x)
    dx[b * HW + i] = (1.f / (1.f + expf(-x[b * HW + i])) - t[b * HW + i]) * k;
}
}  // namespace

extern "C" int pxl_fcd_prepare(long total, const float* x, const float* task_gt, int ignore_index, float target,
                               float* pred_out, float* gt_out, void* stream) {
  PXL_REQUIRE(x && (pred_out || gt_out) && total > 0, "fcd_prepare: bad argument");
  long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(fcd_prepare_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), total, x, task_gt,
                     ignore_index, target, pred_out, gt_out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bce_logits_fwd(int B, long HW, const float* x, const float* t, float* loss, void* stream) {
  PXL_REQUIRE(x && t && loss && B > 0 && HW > 0, "bce_logits_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(loss, 0, (size_t)B * sizeof(float), s));
  const bool det = pxl_det_now();
  const int gx = det ? 1 : (int)((HW + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(bce_logits_fwd_kernel, dim3(gx, B), dim3(det ? 64 : 256), 0, s, HW, x, t, loss);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bce_logits_bwd(int B, long HW, const float* x, const float* t, const float* gout, float* dx, void* stream) {
  PXL_REQUIRE(x && t && gout && dx && B > 0 && HW > 0, "bce_logits_bwd: bad argument");
  const int gx = (int)((HW + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(bce_logits_bwd_kernel, dim3(gx, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), HW, x, t, gout, dx);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
// F.softmax(x, dim=1) on NCHW fp32 (the activation of the sseg task: task/sseg/model.py:59-65, func.py:216-220) and its
// backward dx = p * (dp - sum_c dp * p); one thread per pixel, channel planes read in coalesced lines.
__global__ __launch_bounds__(256) void softmax_nchw_fwd_kernel(int C, long HW, const float* __restrict__ x, float* __restrict__ p) {
  const int n = blockIdx.y;
  const float* xs = x + (size_t)n * C * HW;
  float* ps = p + (size_t)n * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    float v[MAXC], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = xs[(size_t)c * HW + i]; mx = fmaxf(mx, v[c]); }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) ps[(size_t)c * HW + i] = v[c] * inv;
  }
}
__global__ __launch_bounds__(256) void softmax_nchw_bwd_kernel(int C, long HW, const float* __restrict__ p,
                                                               const float* __restrict__ dp, float* __restrict__ dx) {
  const int n = blockIdx.y;
  const size_t base = (size_t)n * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    float pv[MAXC], gv[MAXC], dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) { pv[c] = p[base + (size_t)c * HW + i]; gv[c] = dp[base + (size_t)c * HW + i]; dot += pv[c] * gv[c]; }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) dx[base + (size_t)c * HW + i] = pv[c] * (gv[c] - dot);
  }
}
}  // namespace

extern "C" int pxl_softmax_nchw_fwd(int N, int C, long HW, const float* x, float* p, void* stream) {
  PXL_REQUIRE(x && p && N > 0 && HW > 0 && C >= 1 && C <= MAXC, "softmax_nchw_fwd: bad argument (C <= %d)", MAXC);
  long g = (HW + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(softmax_nchw_fwd_kernel, dim3((int)g, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW, x, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_softmax_nchw_bwd(int N, int C, long HW, const float* p, const float* dp, float* dx, void* stream) {
  PXL_REQUIRE(p && dp && dx && N > 0 && HW > 0 && C >= 1 && C <= MAXC, "softmax_nchw_bwd: bad argument (C <= %d)", MAXC);
  long g = (HW + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(softmax_nchw_bwd_kernel, dim3((int)g, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C, HW, p, dp, dx);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_mse_fwd(long n, const float* a, const float* b, float* out, void* stream) {
  PXL_REQUIRE(a && b && out && n > 0, "mse_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), s));
  // float4 path needs 16-byte aligned operands (a batch slice of an odd-sized plane is not); the
  // scalar tail loop of the kernel covers everything when n4 == 0
  const bool vec = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
  const long n4 = vec ? n / 4 : 0;
  long g = ((vec ? n4 : n) + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1 || pxl_det_now()) g = 1;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3((int)g), dim3(256), 0, s, n4, n, a, b, 1.0f / (float)n, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_mse_bwd(long n, const float* a, const float* b, const float* gout, float* da, void* stream) {
  PXL_REQUIRE(a && b && gout && da && n > 0, "mse_bwd: bad argument");
  const bool vec = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)da % 16 == 0);
  const long n4 = vec ? n / 4 : 0;
  long g = ((vec ? n4 : n) + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n4, n,
                     a, b, gout, 2.0f / (float)n, da);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
