// SSLCCT auxiliary-decoder perturbations on the encoder latent (NCHW fp32 [B][C][h*w], ~1M elements: HBM/launch
// bound, one fused kernel per perturbation) and the guidance masks they use.
//   reference: pixelssl/ssl_algorithm/ssl_cct.py:535-745 (VAT / DropOut / CutOut / Con-Msk / Obj-Msk / F-Drop / F-Noise)
#include <cstdlib>
#include "common.h"

namespace {

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// out = x * mask[b][p] * cscale[b][c] * (1 + noise[c][p]) + add_scale * add[b][c][p]   (NULL factors are skipped)
__global__ __launch_bounds__(256) void perturb_kernel(int B, int C, int HW, const float* __restrict__ x,
                                                      const float* __restrict__ mask, const float* __restrict__ cscale,
                                                      const float* __restrict__ noise, const float* __restrict__ add,
                                                      float add_scale, float* __restrict__ out) {
  const long total = (long)B * C * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long)HW * C));
    float v = x[i];
    if (mask) v *= mask[(size_t)b * HW + p];
    if (cscale) v *= cscale[(size_t)b * C + c];
    if (noise) v *= 1.f + noise[(size_t)c * HW + p];
    if (add) v += add_scale * add[i];
    out[i] = v;
  }
}

// mask[b][i][j] = (argmax_c pred[b][:][yi][xj] > 0) ^ invert, (yi, xj) = F.interpolate(mode='nearest') source of (i, j)
__global__ __launch_bounds__(256) void fg_mask_kernel(int B, int C, int H, int W, const float* __restrict__ pred, int h, int w,
                                                      int invert, float* __restrict__ mask) {
  const long total = (long)B * h * w;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
    const int ys = min((int)floorf((float)y * sy), H - 1), xs = min((int)floorf((float)x * sx), W - 1);
    const float* p = pred + (size_t)b * C * H * W + (size_t)ys * W + xs;
    const float bg = p[0];
    bool fg = false;
    for (int c = 1; c < C; ++c) fg = fg || (p[(size_t)c * H * W] > bg);      // torch.argmax returns the first maximum
    mask[i] = (fg != (invert != 0)) ? 1.f : 0.f;
  }
}

// att[b][p] = mean_c x[b][c][p].  A block = 32 pixels x 8 channel lanes: lane k sums channels k, k+8, ... of its pixel, the
// eight partial sums are folded in lane order through LDS (deterministic).  (One thread per pixel over all 512 channels
// left 17 blocks on the GPU: 243 us for F-Drop's 4 x 512 x 33 x 33 latent.)
__global__ __launch_bounds__(256) void chan_mean_kernel(int B, int C, int HW, const float* __restrict__ x, float* __restrict__ att) {
  __shared__ float red[8][33];
  const int px = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int b = blockIdx.y;
  const int p = blockIdx.x * 32 + px;
  float s = 0.f;
  if (p < HW) {
    const float* q = x + (size_t)b * C * HW + p;
    for (int c = cl; c < C; c += 8) s += q[(size_t)c * HW];
  }
  red[cl][px] = s;
  __syncthreads();
  if (cl == 0 && p < HW) {
    float t = red[0][px];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][px];
    att[(size_t)b * HW + p] = t / (float)C;
  }
}

// one block per sample: mask[b][p] = att[b][p] < max_p(att[b]) * u
__global__ __launch_bounds__(256) void fdrop_mask_kernel(int HW, const float* __restrict__ att, float u, float* __restrict__ mask) {
  __shared__ float red[4];
  const float* a = att + (size_t)blockIdx.x * HW;
  float m = -INFINITY;
  for (int p = threadIdx.x; p < HW; p += 256) m = fmaxf(m, a[p]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float thr = m * u;
  for (int p = threadIdx.x; p < HW; p += 256) mask[(size_t)blockIdx.x * HW + p] = a[p] < thr ? 1.f : 0.f;
}

// out[b] = scale * x[b] / (||x[b]||_2 + 1e-8) in two passes over many blocks: squared-norm partials -> norm2[b] (fp32
// atomics, caller-provided, zeroed here), then the scaling.  (One block per sample took 552 us for I-VAT's 4 x 512 x 33 x 33
// direction: 4 blocks on 256 CUs.)
__global__ __launch_bounds__(256) void l2_zero_kernel(int B, float* __restrict__ norm2) {
  if (threadIdx.x < B) norm2[threadIdx.x] = 0.f;
}
__global__ __launch_bounds__(256) void l2_norm2_kernel(long n, const float* __restrict__ x, float* __restrict__ norm2) {
  __shared__ float red[4];
  const float* a = x + (size_t)blockIdx.y * n;
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) s += a[i] * a[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(norm2 + blockIdx.y, red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void l2_scale_kernel(long n, const float* __restrict__ x, const float* __restrict__ norm2, float scale,
                                                       float* __restrict__ out) {
  const float inv = scale / (sqrtf(norm2[blockIdx.y]) + 1e-8f);
  const size_t base = (size_t)blockIdx.y * n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[base + i] = x[base + i] * inv;
}

__global__ __launch_bounds__(256) void sub_scale_kernel(long n, const float* __restrict__ a, const float* __restrict__ b, float s,
                                                        float* __restrict__ out) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = (a[i] - b[i]) * s;
}

}  // namespace

extern "C" int pxl_latent_perturb(int B, int C, long HW, const float* x, const float* mask, const float* cscale,
                                  const float* noise, const float* add, float add_scale, float* out, void* stream) {
  PXL_REQUIRE(x && out && B > 0 && C > 0 && HW > 0, "latent_perturb: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  perturb_kernel<<<grid_for((long)B * C * HW), 256, 0, s>>>(B, C, (int)HW, x, mask, cscale, noise, add, add_scale, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_fg_mask_nearest(int B, int C, int H, int W, const float* pred, int h, int w, int invert, float* mask,
                                   void* stream) {
  PXL_REQUIRE(pred && mask && B > 0 && C > 1 && H > 0 && W > 0 && h > 0 && w > 0, "fg_mask_nearest: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  fg_mask_kernel<<<grid_for((long)B * h * w), 256, 0, s>>>(B, C, H, W, pred, h, w, invert, mask);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_chan_mean(int B, int C, long HW, const float* x, float* att, void* stream) {
  PXL_REQUIRE(x && att && B > 0 && C > 0 && HW > 0, "chan_mean: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  chan_mean_kernel<<<dim3((unsigned)((HW + 31) / 32), B), 256, 0, s>>>(B, C, (int)HW, x, att);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_fdrop_mask(int B, long HW, const float* att, float u, float* mask, void* stream) {
  PXL_REQUIRE(att && mask && B > 0 && HW > 0, "fdrop_mask: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  fdrop_mask_kernel<<<B, 256, 0, s>>>((int)HW, att, u, mask);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_l2_normalize_persample(int B, long n, const float* x, float scale, float* norm2, float* out, void* stream) {
  PXL_REQUIRE(x && out && norm2 && B > 0 && B <= 256 && n > 0, "l2_normalize_persample: bad argument (1 <= B <= 256, norm2 = B floats of scratch)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (gx > 512) gx = 512;
  // PXL_DETERMINISTIC=1: ONE block per sample for the squared norm (its four wave sums are added in wave order, one add onto the
  // zeroed slot): I-VAT's direction, and with it every figure of a CCT step, is then the same on every run
  const char* de = getenv("PXL_DETERMINISTIC");
  const int gxn = (de != nullptr && de[0] == '1') ? 1 : gx;
  l2_zero_kernel<<<1, 256, 0, s>>>(B, norm2);
  l2_norm2_kernel<<<dim3(gxn, B), 256, 0, s>>>(n, x, norm2);
  l2_scale_kernel<<<dim3(gx, B), 256, 0, s>>>(n, x, norm2, scale, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_sub_scale(long n, const float* a, const float* b, float scale, float* out, void* stream) {
  PXL_REQUIRE(a && b && out && n > 0, "sub_scale: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  sub_scale_kernel<<<grid_for(n), 256, 0, s>>>(n, a, b, scale, out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
