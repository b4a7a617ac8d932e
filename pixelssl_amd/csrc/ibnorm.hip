// IBNorm + LeakyReLU of the GCT flaw detector (pixelssl/ssl_algorithm/ssl_gct.py:588-607, 567-585), NHWC.
//   channels [0, nb)  : SynchronizedBatchNorm2d(affine) -- statistics over (B, H, W) (and all ranks)
//   channels [nb, C)  : InstanceNorm2d(affine=False)    -- statistics over (H, W) of each sample
// One set of kernels serves both halves: statistics are accumulated per (sample, channel); the BN half is folded over
// the samples (-> [2*nb] vector that the Sync-BN hook all-reduces) and every (sample, channel) gets a coefficient row
// (mean, rstd, scale, shift).  The activation z = LeakyReLU(y*scale + shift) is materialised for the LDS-DMA
// convolution that consumes it.  HBM-bound row streaming with the column-group geometry of common.h.
#include "common.h"

namespace {

// sums[b][0][c] += sum_hw v1 ; sums[b][1][c] += sum_hw v2, per sample b = blockIdx.z.
//   MODE 0 (forward):  v1 = y, v2 = y^2
//   MODE 1 (backward): v1 = dz', v2 = dz' * xhat   with dz' = dout * (pre > 0 ? 1 : slope), pre = y*scale + shift
template <typename T, int MODE>
__global__ __launch_bounds__(256) void ibn_reduce_kernel(int HW, int C, const T* __restrict__ y,
                                                         const T* __restrict__ dout, const float* __restrict__ coef,
                                                         float slope, float* __restrict__ sums, int rows_per_group) {
  constexpr int EPC = Elem<T>::EPC;
  const ColGeom g = col_geom(C, EPC);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  const int b = blockIdx.z;
  __shared__ float red[256 * 2 * EPC];
  float a1[EPC], a2[EPC], mean[EPC], rstd[EPC], sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) a1[e] = a2[e] = 0.f;
  if (MODE == 1) {
    const float* cf = coef + (size_t)b * 4 * C + cc * EPC;
    load_cvec<EPC>(cf, mean); load_cvec<EPC>(cf + C, rstd); load_cvec<EPC>(cf + 2 * C, sc); load_cvec<EPC>(cf + 3 * C, sh);
  }
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(HW, m_begin + rows_per_group);
  const size_t base = (size_t)b * HW * C;
  for (int m = m_begin + rlane; m < m_end; m += g.rl) {
    const size_t o = base + (size_t)m * C + cc * EPC;
    float fy[EPC];
    Chunk<T>::unpack(*reinterpret_cast<const uint4*>(y + o), fy);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { a1[e] += fy[e]; a2[e] += fy[e] * fy[e]; }
    } else {
      float fd[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(dout + o), fd);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float dz = (fy[e] * sc[e] + sh[e] > 0.f) ? fd[e] : fd[e] * slope;
        a1[e] += dz;
        a2[e] += dz * (fy[e] - mean[e]) * rstd[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[(rlane * g.cg + ccol) * 2 * EPC + e] = a1[e];
    red[(rlane * g.cg + ccol) * 2 * EPC + EPC + e] = a2[e];
  }
  __syncthreads();
  const int nout = g.cg * 2 * EPC;
  if ((int)threadIdx.x < nout) {
    const int col = threadIdx.x / (2 * EPC), w = threadIdx.x % (2 * EPC);
    float v = 0.f;
    for (int r = 0; r < g.rl; ++r) v += red[(r * g.cg + col) * 2 * EPC + w];
    const int c = (blockIdx.x * g.cg + col) * EPC + (w % EPC);
    atomicAdd(sums + ((size_t)b * 2 + w / EPC) * C + c, v);
  }
}

// bn[0..nb) = sum_b sums[b][0][c], bn[nb..2nb) = sum_b sums[b][1][c]; backward: also dgamma += bn[nb+c], dbeta += bn[c]
__global__ void ibn_fold_kernel(int B, int C, int nb, const float* __restrict__ sums, float* __restrict__ bn,
                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nb) return;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) { s1 += sums[((size_t)b * 2) * C + c]; s2 += sums[((size_t)b * 2 + 1) * C + c]; }
  bn[c] = s1;
  bn[nb + c] = s2;
  if (dgamma) dgamma[c] += s2;
  if (dbeta) dbeta[c] += s1;
}

// coef[b][0..4][c] = mean, rstd, scale, shift
__global__ void ibn_coef_kernel(int B, int C, int nb, float hw, float count_bn, const float* __restrict__ sums,
                                const float* __restrict__ bn, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                float momentum, float eps, int training, int clamp_var, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= C) return;
  float mean, var, g = 1.f, bt = 0.f;
  bool clamp = false;
  if (c < nb) {
    if (training) {
      float s1, s2;
      if (bn != nullptr) { s1 = bn[c]; s2 = bn[nb + c]; }
      else {                      // single rank: the fold over the samples (ibn_fold_kernel) in place, same order
        s1 = s2 = 0.f;
        for (int q = 0; q < B; ++q) { s1 += sums[((size_t)q * 2) * C + c]; s2 += sums[((size_t)q * 2 + 1) * C + c]; }
      }
      mean = s1 / count_bn;
      var = fmaxf(s2 / count_bn - mean * mean, 0.f);
      if (b == 0 && rmean != nullptr) {
        const float unbiased = count_bn > 1.f ? var * count_bn / (count_bn - 1.f) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
      }
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    g = gamma[c];
    bt = beta[c];
    clamp = clamp_var != 0;
  } else {
    mean = sums[((size_t)b * 2) * C + c] / hw;
    var = fmaxf(sums[((size_t)b * 2 + 1) * C + c] / hw - mean * mean, 0.f);
  }
  const float rstd = clamp ? rsqrtf(fmaxf(var, eps)) : rsqrtf(var + eps);
  float* cf = coef + (size_t)b * 4 * C + c;
  cf[0] = mean;
  cf[C] = rstd;
  cf[2 * C] = g * rstd;
  cf[3 * C] = bt - mean * g * rstd;
}

// out = LeakyReLU(y*scale[b][c] + shift[b][c])
template <typename T>
__global__ __launch_bounds__(256) void ibn_apply_fwd_kernel(int HW, int C, const T* __restrict__ y,
                                                            const float* __restrict__ coef, float slope,
                                                            T* __restrict__ out, int rows_per_group) {
  constexpr int EPC = Elem<T>::EPC;
  const ColGeom g = col_geom(C, EPC);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  const int b = blockIdx.z;
  float sc[EPC], sh[EPC];
  load_cvec<EPC>(coef + (size_t)b * 4 * C + 2 * C + cc * EPC, sc);
  load_cvec<EPC>(coef + (size_t)b * 4 * C + 3 * C + cc * EPC, sh);
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(HW, m_begin + rows_per_group);
  const size_t base = (size_t)b * HW * C;
  for (int m = m_begin + rlane; m < m_end; m += g.rl) {
    const size_t o = base + (size_t)m * C + cc * EPC;
    float f[EPC];
    Chunk<T>::unpack(*reinterpret_cast<const uint4*>(y + o), f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float v = f[e] * sc[e] + sh[e];
      f[e] = v > 0.f ? v : v * slope;
    }
    *reinterpret_cast<uint4*>(out + o) = Chunk<T>::pack(f);
  }
}

// dy = scale * (dz' - m1 - xhat*m2): BN half m = bn / count_bn (0 in eval mode), IN half m = bsums[b] / hw
template <typename T>
__global__ __launch_bounds__(256) void ibn_bwd_apply_kernel(int HW, int C, int nb, const T* __restrict__ dout,
                                                            const T* __restrict__ y, const float* __restrict__ coef,
                                                            const float* __restrict__ bsums,
                                                            const float* __restrict__ bn, float inv_count_bn,
                                                            float inv_hw, int training, float slope,
                                                            T* __restrict__ dy, int rows_per_group) {
  constexpr int EPC = Elem<T>::EPC;
  const ColGeom g = col_geom(C, EPC);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  const int b = blockIdx.z;
  float mean[EPC], rstd[EPC], sc[EPC], sh[EPC], m1[EPC], m2[EPC];
  {
    const float* cf = coef + (size_t)b * 4 * C + cc * EPC;
    load_cvec<EPC>(cf, mean); load_cvec<EPC>(cf + C, rstd); load_cvec<EPC>(cf + 2 * C, sc); load_cvec<EPC>(cf + 3 * C, sh);
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = cc * EPC + e;
    if (c < nb) {
      m1[e] = training ? bn[c] * inv_count_bn : 0.f;
      m2[e] = training ? bn[nb + c] * inv_count_bn : 0.f;
    } else {
      m1[e] = bsums[((size_t)b * 2) * C + c] * inv_hw;
      m2[e] = bsums[((size_t)b * 2 + 1) * C + c] * inv_hw;
    }
  }
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(HW, m_begin + rows_per_group);
  const size_t base = (size_t)b * HW * C;
  for (int m = m_begin + rlane; m < m_end; m += g.rl) {
    const size_t o = base + (size_t)m * C + cc * EPC;
    float fd[EPC], fy[EPC], v[EPC];
    Chunk<T>::unpack(*reinterpret_cast<const uint4*>(dout + o), fd);
    Chunk<T>::unpack(*reinterpret_cast<const uint4*>(y + o), fy);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float dz = (fy[e] * sc[e] + sh[e] > 0.f) ? fd[e] : fd[e] * slope;
      v[e] = sc[e] * (dz - m1[e] - (fy[e] - mean[e]) * rstd[e] * m2[e]);
    }
    *reinterpret_cast<uint4*>(dy + o) = Chunk<T>::pack(v);
  }
}

inline bool ok_dtype(int dtype) { return dtype == PXL_F32 || dtype == PXL_BF16; }

}  // namespace

// PXL_DETERMINISTIC=1 (read per call: these entry points are also used without an executor): the REDUCING launches take one row
// group per (sample, channel group), so every sum receives exactly one add onto its zeroed slot -- the same bits on every run
// (the default spreads a sample's rows over several blocks whose atomics land in any order)
static inline bool ibn_det_now() { const char* e = getenv("PXL_DETERMINISTIC"); return e != nullptr && e[0] == '1'; }
#define IBN_GEOM(target, reducing)                              \
  const int epc = dtype == PXL_F32 ? 4 : 8;                     \
  PXL_REQUIRE(C % epc == 0, "ibnorm: C=%d must be a multiple of %d", C, epc); \
  const ColGeom g = col_geom(C, epc);                           \
  const int rpg = ((reducing) && ibn_det_now()) ? (HW + g.rl - 1) / g.rl * g.rl : rows_per_group(HW, g, (target) / B > 0 ? (target) / B : 1); \
  const dim3 grid(g.ncg, cdiv(HW, rpg), B);                     \
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

static int ibn_stats_impl(int dtype, int B, int HW, int C, const void* y, float* sums, int zero, void* stream);
extern "C" int pxl_ibn_stats(int dtype, int B, int HW, int C, const void* y, float* sums, void* stream) {
  return ibn_stats_impl(dtype, B, HW, C, y, sums, 1, stream);
}
// the same onto CALLER-ZEROED sums (the executor zeroes the sums of all IBNorm layers of a pass with one memset)
extern "C" int pxl_ibn_stats_acc(int dtype, int B, int HW, int C, const void* y, float* sums, void* stream) {
  return ibn_stats_impl(dtype, B, HW, C, y, sums, 0, stream);
}
static int ibn_stats_impl(int dtype, int B, int HW, int C, const void* y, float* sums, int zero, void* stream) {
  PXL_REQUIRE(y && sums && B > 0 && HW > 0 && ok_dtype(dtype), "ibn_stats: bad argument");
  IBN_GEOM(1024, true)
  if (zero) PXL_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)B * 2 * C * sizeof(float), s));
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((ibn_reduce_kernel<float, 0>), grid, dim3(256), 0, s, HW, C, (const float*)y, nullptr, nullptr, 0.f, sums, rpg);
  else
    hipLaunchKernelGGL((ibn_reduce_kernel<bf16_t, 0>), grid, dim3(256), 0, s, HW, C, (const bf16_t*)y, nullptr, nullptr, 0.f, sums, rpg);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ibn_fold(int B, int C, int nb, const float* sums, float* bn, float* dgamma, float* dbeta, void* stream) {
  PXL_REQUIRE(sums && bn && B > 0 && nb >= 1 && nb <= C, "ibn_fold: bad argument");
  hipLaunchKernelGGL(ibn_fold_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C, nb,
                     sums, bn, dgamma, dbeta);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ibn_coef(int B, int C, int nb, int HW, float count_bn, const float* sums, const float* bn,
                            const float* gamma, const float* beta, float* rmean, float* rvar, float momentum, float eps,
                            int training, int clamp_var, float* coef, void* stream) {
  PXL_REQUIRE(sums && gamma && beta && coef && B > 0, "ibn_coef: bad argument");      // bn NULL: batch sums folded from `sums` here
  PXL_REQUIRE(training || (rmean && rvar), "ibn_coef: eval mode needs running statistics");
  hipLaunchKernelGGL(ibn_coef_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), B, C, nb,
                     (float)HW, count_bn, sums, bn, gamma, beta, rmean, rvar, momentum, eps, training, clamp_var, coef);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ibn_apply_fwd(int dtype, int B, int HW, int C, const void* y, const float* coef, float slope, void* out,
                                 void* stream) {
  PXL_REQUIRE(y && coef && out && B > 0 && ok_dtype(dtype), "ibn_apply_fwd: bad argument");
  IBN_GEOM(2048, false)
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(ibn_apply_fwd_kernel<float>, grid, dim3(256), 0, s, HW, C, (const float*)y, coef, slope, (float*)out, rpg);
  else
    hipLaunchKernelGGL(ibn_apply_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, HW, C, (const bf16_t*)y, coef, slope, (bf16_t*)out, rpg);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

static int ibn_bwd_reduce_impl(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef, float slope,
                               float* bsums, int zero, void* stream);
extern "C" int pxl_ibn_bwd_reduce(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef,
                                  float slope, float* bsums, void* stream) {
  return ibn_bwd_reduce_impl(dtype, B, HW, C, dout, y, coef, slope, bsums, 1, stream);
}
extern "C" int pxl_ibn_bwd_reduce_acc(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef,
                                      float slope, float* bsums, void* stream) {
  return ibn_bwd_reduce_impl(dtype, B, HW, C, dout, y, coef, slope, bsums, 0, stream);
}
static int ibn_bwd_reduce_impl(int dtype, int B, int HW, int C, const void* dout, const void* y, const float* coef, float slope,
                               float* bsums, int zero, void* stream) {
  PXL_REQUIRE(dout && y && coef && bsums && B > 0 && ok_dtype(dtype), "ibn_bwd_reduce: bad argument");
  IBN_GEOM(1024, true)
  if (zero) PXL_CHECK_HIP(hipMemsetAsync(bsums, 0, (size_t)B * 2 * C * sizeof(float), s));
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((ibn_reduce_kernel<float, 1>), grid, dim3(256), 0, s, HW, C, (const float*)y, (const float*)dout, coef, slope, bsums, rpg);
  else
    hipLaunchKernelGGL((ibn_reduce_kernel<bf16_t, 1>), grid, dim3(256), 0, s, HW, C, (const bf16_t*)y, (const bf16_t*)dout, coef, slope, bsums, rpg);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_ibn_bwd_apply(int dtype, int B, int HW, int C, int nb, const void* dout, const void* y,
                                 const float* coef, const float* bsums, const float* bn, float count_bn, int training,
                                 float slope, void* dy, void* stream) {
  PXL_REQUIRE(dout && y && coef && bsums && bn && dy && B > 0 && ok_dtype(dtype), "ibn_bwd_apply: bad argument");
  IBN_GEOM(2048, false)
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(ibn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, HW, C, nb, (const float*)dout, (const float*)y, coef,
                       bsums, bn, 1.f / count_bn, 1.f / (float)HW, training, slope, (float*)dy, rpg);
  else
    hipLaunchKernelGGL(ibn_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, s, HW, C, nb, (const bf16_t*)dout, (const bf16_t*)y, coef,
                       bsums, bn, 1.f / count_bn, 1.f / (float)HW, training, slope, (bf16_t*)dy, rpg);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
