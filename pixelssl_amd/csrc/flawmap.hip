// GCT flaw-map pipeline, fp32 NCHW like the task model's outputs.
//   FDGTGenerator      ssl_gct.py:692-728   |onehot - softmax| summed over classes * mu -> blur -> nu x (dilate, reblur)
//                                           -> per-sample min-max normalisation
//   FlawmapHandler     ssl_gct.py:624-657   in-place clamp >= 0 -> blur -> zero if max <= 0.1 -> min-max normalisation
//   DCGTGenerator      ssl_gct.py:660-689   dynamic-consistency pseudo ground truth (element-wise)
//   FlawDetectorCriterion ssl_gct.py:610-621  per-sample MSE
// The reference's GaussianBlurLayer (nn/module/gaussian_blur.py) is a dense k x k depthwise convolution (k = 33 / 65 /
// 129 at 513 x 513) whose kernel is scipy's separable gaussian filter of a delta: rank 1, so it is evaluated as a
// row pass + a column pass over the reflect-padded map (2k instead of k*k MACs per pixel).  All of it is HBM/L2-bound
// single-channel work: coalesced along x, wavefront shuffles + one atomic per block for the per-sample reductions.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect101(int i, int n) {      // torch ReflectionPad2d: -1 -> 1, n -> n-2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// out[b][y][x] = mu * sum_c |onehot(gt)[c] - pred[b][c][y][x]| ; gt: float class ids, ignore -> all-zero one-hot,
// ids outside [0, C) (e.g. -1 = unlabeled) -> all-zero one-hot (task/sseg/func.py:179-192)
__global__ void absdiff_chansum_kernel(int B, int C, long HW, const float* __restrict__ pred,
                                       const float* __restrict__ gt, int ignore, float mu, float* __restrict__ out) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const int lab = (int)gt[i];
    const bool valid = lab != ignore && lab >= 0 && lab < C;
    const float* pp = pred + b * C * HW + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
      const float oh = (valid && c == lab) ? 1.f : 0.f;
      s += fabsf(oh - pp[(long)c * HW]);
    }
    out[i] = s * mu;
  }
}

// the same against a dense target [B][C][HW] (the one-hot tensor the reference's task hook builds, or a regression target):
// torch.sum(torch.abs(gt - pred), dim=1) * mu with torch's channel order of summation (ssl_gct.py:703)
__global__ void absdiff_chansum_dense_kernel(int B, int C, long HW, const float* __restrict__ pred,
                                             const float* __restrict__ gt, float mu, float* __restrict__ out) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float* pp = pred + b * C * HW + p;
    const float* gg = gt + b * C * HW + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += fabsf(gg[(long)c * HW] - pp[(long)c * HW]);
    out[i] = s * mu;
  }
}

// explicit one-hot tensor [B][C][HW] (ssladv / sslgct task hooks): 1 at the label's class, 0 for ignored pixels
__global__ void onehot_kernel(int B, int C, long HW, const float* __restrict__ gt, int ignore, float* __restrict__ out) {
  const long total = (long)B * C * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % HW;
    const int c = (int)((i / HW) % C);
    const long b = i / (HW * C);
    const int lab = (int)gt[b * HW + p];
    out[i] = (lab != ignore && lab == c) ? 1.f : 0.f;
  }
}

// one pass of the separable blur along x (dir = 0) or y (dir = 1), reflect padding
__global__ void blur_pass_kernel(int B, int H, int W, const float* __restrict__ in, const float* __restrict__ taps,
                                 int k, int dir, float* __restrict__ out) {
  const long total = (long)B * H * W;
  const int r = k / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const long base = (i / ((long)W * H)) * (long)W * H;
    float s = 0.f;
    if (dir == 0) {
      const float* row = in + base + (long)y * W;
      for (int t = 0; t < k; ++t) s += taps[t] * row[reflect101(x + t - r, W)];
    } else {
      const float* col = in + base + x;
      for (int t = 0; t < k; ++t) s += taps[t] * col[(long)reflect101(y + t - r, H) * W];
    }
    out[i] = s;
  }
}

// x *= (x >= 0)   (FlawmapHandler mutates its argument, ssl_gct.py:643-645)
__global__ void clamp0_kernel(long n, float* __restrict__ x) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    x[i] = v >= 0.f ? v : 0.f;
  }
}

// 3 x 3 max filter with reflect padding 1 (nn.ReflectionPad2d(1) + nn.MaxPool2d(3, 1))
__global__ void dilate3_kernel(int B, int H, int W, const float* __restrict__ in, float* __restrict__ out) {
  const long total = (long)B * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const float* img = in + (i / ((long)W * H)) * (long)W * H;
    float m = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) m = fmaxf(m, img[(long)reflect101(y + dy, H) * W + reflect101(x + dx, W)]);
    out[i] = m;
  }
}

__device__ __forceinline__ void atomic_maxf(float* a, float v) {      // IEEE order trick, valid for any sign
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_minf(float* a, float v) {
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(a), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned*>(a), __float_as_uint(v));
}

// mm[b] = (min, max) over the sample; mm must be initialised to (+inf, -inf)
__global__ void minmax_kernel(long HW, const float* __restrict__ x, float* __restrict__ mm) {
  const int b = blockIdx.y;
  const float* xb = x + b * HW;
  float lo = INFINITY, hi = -INFINITY;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const float v = xb[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  hi = wave_max(hi);
  lo = -wave_max(-lo);
  // one atomic pair per BLOCK: same-address atomics serialize at ~10 ns each, and with one pair per wave of ~130 blocks per
  // sample the 16 result words took 74 us to settle
  __shared__ float rlo[4], rhi[4];
  if ((threadIdx.x & 63) == 0) { rlo[threadIdx.x >> 6] = lo; rhi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w) { lo = fminf(lo, rlo[w]); hi = fmaxf(hi, rhi[w]); }
    atomic_minf(mm + 2 * b, lo);
    atomic_maxf(mm + 2 * b + 1, hi);
  }
}
__global__ void minmax_init_kernel(int B, float* __restrict__ mm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { mm[2 * i] = INFINITY; mm[2 * i + 1] = -INFINITY; }
}

// out = ((max > clip ? x : 0) - min) / (max - min + 1e-9)     clip = -inf for the FDGT normalisation
__global__ void minmax_norm_kernel(long HW, const float* __restrict__ x, const float* __restrict__ mm, float clip,
                                   float* __restrict__ out) {
  const int b = blockIdx.y;
  const float lo = mm[2 * b], hi = mm[2 * b + 1];
  const float keep = hi > clip ? 1.f : 0.f;
  const float den = hi - lo + 1e-9f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x)
    out[b * HW + i] = (x[b * HW + i] * keep - lo) / den;
}

// GaussianNoiseLayer.forward (pixelssl/nn/module/gaussian_noise.py:18-40), in place on x: per-sample min-max normalise to
// [0, 1], add the noise, clip to [0, 1] (the reference's mask arithmetic), de-normalise.  Same operation order as the
// reference's in-place chain: sub_(min).div_(range) ; add_(noise) ; clip ; mul_(range).add_(min).
__global__ void gaussian_noise_kernel(long n, float* __restrict__ x, const float* __restrict__ noise,
                                      const float* __restrict__ mm) {
  const int b = blockIdx.y;
  const float lo = mm[2 * b], hi = mm[2 * b + 1];
  const float range = hi - lo + 1e-9f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = (x[b * n + i] - lo) / range;
    v = v + noise[b * n + i];
    const float ub = v > 1.f ? 1.f : 0.f;
    v = v * (1.f - ub) + ub;
    const float lb = v < 0.f ? 1.f : 0.f;
    v = v * (1.f - lb);
    x[b * n + i] = v * range + lo;
  }
}

// DCGTGenerator: fm := fm <= thr ? fm : 1 (in place); mask_l = r_fm >= l_fm; gt_l = mask_l ? l_pred : r_pred
__global__ void dcgt_kernel(int B, int C, long HW, const float* __restrict__ lp, const float* __restrict__ rp,
                            float* __restrict__ lf, float* __restrict__ rf, float thr, float* __restrict__ lgt,
                            float* __restrict__ rgt, float* __restrict__ both_bad) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    const float l0 = lf[i], r0 = rf[i];
    const bool lb = l0 > thr, rb = r0 > thr;
    both_bad[i] = (lb && rb) ? 1.f : 0.f;
    const float l1 = l0 * (l0 <= thr ? 1.f : 0.f) + (lb ? 1.f : 0.f);
    const float r1 = r0 * (r0 <= thr ? 1.f : 0.f) + (rb ? 1.f : 0.f);
    lf[i] = l1;
    rf[i] = r1;
    const float lm = r1 >= l1 ? 1.f : 0.f, rm = l1 >= r1 ? 1.f : 0.f;
    for (int c = 0; c < C; ++c) {
      const long o = (b * C + c) * HW + p;
      const float a = lp[o], bb = rp[o];
      lgt[o] = lm * a + (1.f - lm) * bb;
      rgt[o] = rm * bb + (1.f - rm) * a;
    }
  }
}

// PXL_DETERMINISTIC=1 (read per call): the two loss sums below run as ONE block per output whose wave sums are folded in wave
// order and stored once -- the same bits on every run (the default: many blocks, one atomic per wave, any order)
static inline bool flaw_det_now() { const char* e = getenv("PXL_DETERMINISTIC"); return e != nullptr && e[0] == '1'; }
__device__ __forceinline__ void block_add_ordered(float s, float scale, float* dst, bool ordered) {
  s = wave_sum(s);
  if (!ordered) {
    if ((threadIdx.x & 63) == 0) atomicAdd(dst, s * scale);
    return;
  }
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *dst += (((ws[0] + ws[1]) + ws[2]) + ws[3]) * scale;          // (one block per output: no one else writes it)
}

// loss[b] = mean_i (a - g)^2 ; da = 2 (a - g) / n * gout[b]
__global__ __launch_bounds__(256) void mse_ps_fwd_kernel(long n, const float* __restrict__ a, const float* __restrict__ g,
                                                         float* __restrict__ loss, int ordered) {
  const int b = blockIdx.y;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = a[b * n + i] - g[b * n + i];
    s += d * d;
  }
  block_add_ordered(s, 1.f / (float)n, loss + b, ordered != 0);
}
__global__ void mse_ps_bwd_kernel(long n, const float* __restrict__ a, const float* __restrict__ g,
                                  const float* __restrict__ gout, float* __restrict__ da) {
  const int b = blockIdx.y;
  const float k = 2.f * gout[b] / (float)n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    da[b * n + i] = k * (a[b * n + i] - g[b * n + i]);
}

__global__ __launch_bounds__(256) void masked_sq_fwd_kernel(long n, const float* __restrict__ x, const float* __restrict__ mask,
                                                            float* __restrict__ out, int ordered) {
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    s += mask[i] * x[i] * x[i];
  block_add_ordered(s, 1.f / (float)n, out, ordered != 0);
}
__global__ void masked_sq_bwd_kernel(long n, const float* __restrict__ x, const float* __restrict__ mask,
                                     const float* __restrict__ gout, float* __restrict__ dx) {
  const float k = 2.f * gout[0] / (float)n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = k * mask[i] * x[i];
}

inline int g1(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int pxl_absdiff_chansum(int B, int C, long HW, const float* pred, const float* gt, int ignore_index,
                                   float mu, float* out, void* stream) {
  PXL_REQUIRE(pred && gt && out && B > 0 && C > 0 && HW > 0, "absdiff_chansum: bad argument");
  hipLaunchKernelGGL(absdiff_chansum_kernel, dim3(g1((long)B * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     B, C, HW, pred, gt, ignore_index, mu, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_absdiff_chansum_dense(int B, int C, long HW, const float* pred, const float* gt, float mu, float* out,
                                         void* stream) {
  PXL_REQUIRE(pred && gt && out && B > 0 && C > 0 && HW > 0, "absdiff_chansum_dense: bad argument");
  hipLaunchKernelGGL(absdiff_chansum_dense_kernel, dim3(g1((long)B * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     B, C, HW, pred, gt, mu, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_onehot_ignore(int B, int C, long HW, const float* gt, int ignore_index, float* out, void* stream) {
  PXL_REQUIRE(gt && out && B > 0 && C > 0 && HW > 0, "onehot_ignore: bad argument");
  hipLaunchKernelGGL(onehot_kernel, dim3(g1((long)B * C * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C,
                     HW, gt, ignore_index, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_gauss_sep_reflect(int B, int H, int W, const float* x, const float* taps, int k, float* tmp,
                                     float* out, void* stream) {
  PXL_REQUIRE(x && taps && tmp && out && B > 0 && k >= 1 && (k & 1), "gauss_sep_reflect: bad argument (odd k)");
  PXL_REQUIRE(k / 2 < H && k / 2 < W, "gauss_sep_reflect: reflect padding %d needs a larger map (%d x %d)", k / 2, H, W);
  PXL_REQUIRE(tmp != x && tmp != out, "gauss_sep_reflect: tmp must not alias x / out");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(blur_pass_kernel, dim3(g1(n)), dim3(256), 0, s, B, H, W, x, taps, k, 0, tmp);
  PXL_LAUNCH_CHECK();
  hipLaunchKernelGGL(blur_pass_kernel, dim3(g1(n)), dim3(256), 0, s, B, H, W, tmp, taps, k, 1, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_clamp_min0_inplace(long n, float* x, void* stream) {
  PXL_REQUIRE(x && n > 0, "clamp_min0_inplace: bad argument");
  hipLaunchKernelGGL(clamp0_kernel, dim3(g1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, x);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_dilate3_reflect(int B, int H, int W, const float* x, float* out, void* stream) {
  PXL_REQUIRE(x && out && x != out && B > 0 && H > 1 && W > 1, "dilate3_reflect: bad argument");
  hipLaunchKernelGGL(dilate3_kernel, dim3(g1((long)B * H * W)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, H,
                     W, x, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_minmax_norm_persample(int B, long HW, const float* x, float clip_threshold, float* mm, float* out,
                                         void* stream) {
  PXL_REQUIRE(x && mm && out && B > 0 && HW > 0, "minmax_norm_persample: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(minmax_init_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, B, mm);
  const int gx = (int)((HW + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(minmax_kernel, dim3(gx > 32 ? 32 : gx, B), dim3(256), 0, s, HW, x, mm);
  hipLaunchKernelGGL(minmax_norm_kernel, dim3(gx, B), dim3(256), 0, s, HW, x, mm, clip_threshold, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_gaussian_noise_apply(int B, long n, float* x, const float* noise, float* mm, void* stream) {
  PXL_REQUIRE(x && noise && mm && B > 0 && n > 0, "gaussian_noise_apply: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(minmax_init_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, B, mm);
  const int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(minmax_kernel, dim3(gx > 32 ? 32 : gx, B), dim3(256), 0, s, n, x, mm);
  hipLaunchKernelGGL(gaussian_noise_kernel, dim3(gx, B), dim3(256), 0, s, n, x, noise, mm);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_dcgt(int B, int C, long HW, const float* l_pred, const float* r_pred, float* l_fm, float* r_fm,
                        float threshold, float* l_gt, float* r_gt, float* both_bad, void* stream) {
  PXL_REQUIRE(l_pred && r_pred && l_fm && r_fm && l_gt && r_gt && both_bad, "dcgt: null argument");
  hipLaunchKernelGGL(dcgt_kernel, dim3(g1((long)B * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C, HW,
                     l_pred, r_pred, l_fm, r_fm, threshold, l_gt, r_gt, both_bad);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_mse_persample_fwd(int B, long n, const float* a, const float* g, float* loss, void* stream) {
  PXL_REQUIRE(a && g && loss && B > 0 && n > 0, "mse_persample_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(loss, 0, (size_t)B * sizeof(float), s));
  const bool det = flaw_det_now();
  const int gx = det ? 1 : (int)((n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(mse_ps_fwd_kernel, dim3(gx, B), dim3(256), 0, s, n, a, g, loss, det ? 1 : 0);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_mse_persample_bwd(int B, long n, const float* a, const float* g, const float* gout, float* da,
                                     void* stream) {
  PXL_REQUIRE(a && g && gout && da && B > 0 && n > 0, "mse_persample_bwd: bad argument");
  const int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(mse_ps_bwd_kernel, dim3(gx, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, a, g, gout, da);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_masked_sq_mean_fwd(long n, const float* x, const float* mask, float* out, void* stream) {
  PXL_REQUIRE(x && mask && out && n > 0, "masked_sq_mean_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), s));
  const bool det = flaw_det_now();
  long g = det ? 1 : (n + 256 * 8 - 1) / (256 * 8);
  hipLaunchKernelGGL(masked_sq_fwd_kernel, dim3((int)(g > 2048 ? 2048 : g)), dim3(256), 0, s, n, x, mask, out, det ? 1 : 0);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_masked_sq_mean_bwd(long n, const float* x, const float* mask, const float* gout, float* dx, void* stream) {
  PXL_REQUIRE(x && mask && gout && dx && n > 0, "masked_sq_mean_bwd: bad argument");
  hipLaunchKernelGGL(masked_sq_bwd_kernel, dim3(g1(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, x, mask, gout, dx);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
