// BatchNorm statistics plumbing around the fused convolutions (training-mode SyncBN semantics,
// reference: pixelssl/nn/module/third_party/sync_batchnorm/batchnorm.py:48-78, 113-125).
//
// Forward: the producing conv's epilogue accumulates per-channel [sum, sum^2] in fp32; an
// optional RCCL all-reduce of that [2C] buffer (multi-GPU) happens between the conv and
// pxl_bn_finalize, which turns it into the affine (scale, shift) that the *consumer* kernels
// apply on load.  Backward: a reduce pass (sum dz', sum dz'*xhat), a tiny finalize, and an apply
// pass producing the gradient w.r.t. the raw conv output.  All passes are HBM-bound: 16 B per
// lane along channels (NHWC), wave-level register accumulation, one atomic per channel per
// block.
#include "common.h"

namespace {

// coef layout [4][C]: mean, rstd, scale (= gamma*rstd), shift (= beta - mean*scale)
__global__ void bn_finalize_kernel(int C, const float* __restrict__ stats, float count, float total_count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                   float eps, int training, int clamp_var, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    mean = stats[c] / total_count;
    var = stats[C + c] / total_count - mean * mean;
    if (var < 0.f) var = 0.f;
    if (rmean != nullptr) {
      const float unbiased = total_count > 1.f ? var * total_count / (total_count - 1.f) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  // single-device path: (var+eps)^-1/2 ; multi-device path of the reference: clamp(var,eps)^-1/2
  const float rstd = clamp_var ? rsqrtf(fmaxf(var, eps)) : rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  coef[c] = mean;
  coef[C + c] = rstd;
  coef[2 * C + c] = scale;
  coef[3 * C + c] = b - mean * scale;
}

// sums[0..C) += sum_m dzh ; sums[C..2C) += sum_m dzh * xhat     dzh = dz * (relu ? z>0 : 1)
// block = 256 threads = (C/EPC chunks) x rows ; grid-stride over row slabs.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int M, int C, const T* __restrict__ dz,
                                                            const T* __restrict__ y,
                                                            const float* __restrict__ coef, int relu,
                                                            float* __restrict__ sums, int rows_per_block) {
  constexpr int EPC = Elem<T>::EPC;
  const int nchunk = C / EPC;                     // chunks per row
  const int cpb = min(nchunk, 256);               // chunk columns handled per block pass
  const int rpb = 256 / cpb;                      // rows in flight per pass
  const int ccol = threadIdx.x % cpb;
  const int rrow = threadIdx.x / cpb;
  extern __shared__ float red[];                  // [rpb][cpb*EPC*2]
  const int m_begin = blockIdx.x * rows_per_block;
  const int m_end = min(M, m_begin + rows_per_block);
  for (int cc0 = 0; cc0 < nchunk; cc0 += cpb) {
    const int cc = cc0 + ccol;
    float a1[EPC], a2[EPC], mean[EPC], rstd[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    const bool active = cc < nchunk && rrow < rpb;
    if (active) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = cc * EPC + e;
        mean[e] = coef[c]; rstd[e] = coef[C + c]; sc[e] = coef[2 * C + c]; sh[e] = coef[3 * C + c];
      }
      for (int m = m_begin + rrow; m < m_end; m += rpb) {
        const uint4 vd = *reinterpret_cast<const uint4*>(dz + (size_t)m * C + cc * EPC);
        const uint4 vy = *reinterpret_cast<const uint4*>(y + (size_t)m * C + cc * EPC);
        float fd[EPC], fy[EPC];
        Chunk<T>::unpack(vd, fd);
        Chunk<T>::unpack(vy, fy);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float g = fd[e];
          if (relu && !(fy[e] * sc[e] + sh[e] > 0.f)) g = 0.f;
          a1[e] += g;
          a2[e] += g * (fy[e] - mean[e]) * rstd[e];
        }
      }
    }
    // reduce over the rpb row-threads through LDS
    if (rrow < rpb && ccol < cpb) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        red[(rrow * cpb + ccol) * EPC * 2 + e] = a1[e];
        red[(rrow * cpb + ccol) * EPC * 2 + EPC + e] = a2[e];
      }
    }
    __syncthreads();
    if (rrow == 0 && cc < nchunk) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < rpb; ++r) {
          s1 += red[(r * cpb + ccol) * EPC * 2 + e];
          s2 += red[(r * cpb + ccol) * EPC * 2 + EPC + e];
        }
        atomicAdd(sums + cc * EPC + e, s1);
        atomicAdd(sums + C + cc * EPC + e, s2);
      }
    }
    __syncthreads();
  }
}

// bcoef[0..C) = sum dzh / count ; bcoef[C..2C) = sum dzh*xhat / count ; accumulates dgamma, dbeta.
__global__ void bn_bwd_finalize_kernel(int C, const float* __restrict__ sums, float total_count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ bcoef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s1 = sums[c], s2 = sums[C + c];
  bcoef[c] = s1 / total_count;
  bcoef[C + c] = s2 / total_count;
  if (dgamma) dgamma[c] += s2;
  if (dbeta) dbeta[c] += s1;
}

// dy = scale * (dzh - c1 - xhat * c2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long nchunks, int C, const T* __restrict__ dz,
                                                           const T* __restrict__ y,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ bcoef, int relu,
                                                           T* __restrict__ dy) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cpr) * EPC;
    const uint4 vd = reinterpret_cast<const uint4*>(dz)[i];
    const uint4 vy = reinterpret_cast<const uint4*>(y)[i];
    float fd[EPC], fy[EPC], o[EPC];
    Chunk<T>::unpack(vd, fd);
    Chunk<T>::unpack(vy, fy);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = c0 + e;
      const float mean = coef[c], rstd = coef[C + c], sc = coef[2 * C + c], sh = coef[3 * C + c];
      float g = fd[e];
      if (relu && !(fy[e] * sc + sh > 0.f)) g = 0.f;
      o[e] = sc * (g - bcoef[c] - (fy[e] - mean) * rstd * bcoef[C + c]);
    }
    reinterpret_cast<uint4*>(dy)[i] = Chunk<T>::pack(o);
  }
}

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int pxl_bn_finalize(int C, const float* stats, float count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int training, int clamp_var, float* coef,
                               void* stream) {
  PXL_REQUIRE(C > 0 && coef, "bn_finalize: bad argument");
  PXL_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     C, stats, count, count, gamma, beta, running_mean, running_var, momentum, eps, training,
                     clamp_var, coef);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_bwd_reduce(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                                 int relu, float* sums, void* stream) {
  PXL_REQUIRE(dz && y && coef && sums && M > 0, "bn_bwd_reduce: bad argument");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_bwd_reduce: C=%d must be a multiple of %d", C, epc);
  const int nchunk = C / epc;
  const int cpb = nchunk < 256 ? nchunk : 256;
  const int rpb = 256 / cpb;
  int blocks = cdiv(M, rpb * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const int rows_per_block = cdiv(M, blocks);
  blocks = cdiv(M, rows_per_block);
  const size_t smem = (size_t)rpb * cpb * epc * 2 * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(256), smem, s, M, C, (const float*)dz,
                       (const float*)y, coef, relu, sums, rows_per_block);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(256), smem, s, M, C, (const bf16_t*)dz,
                       (const bf16_t*)y, coef, relu, sums, rows_per_block);
  else
    return pxl_set_error(PXL_ERR_ARG, "bn_bwd_reduce: bad dtype %d", dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_bwd_finalize(int C, const float* sums, float count, float* dgamma, float* dbeta,
                                   float* bcoef, void* stream) {
  PXL_REQUIRE(C > 0 && sums && bcoef, "bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), C, sums, count, dgamma, dbeta, bcoef);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_bwd_apply(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                                const float* bcoef, int relu, void* dy, void* stream) {
  PXL_REQUIRE(dz && y && coef && bcoef && dy && M > 0, "bn_bwd_apply: bad argument");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_bwd_apply: C=%d must be a multiple of %d", C, epc);
  const long nchunks = (long)M * (C / epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, C,
                       (const float*)dz, (const float*)y, coef, bcoef, relu, (float*)dy);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, C,
                       (const bf16_t*)dz, (const bf16_t*)y, coef, bcoef, relu, (bf16_t*)dy);
  else
    return pxl_set_error(PXL_ERR_ARG, "bn_bwd_apply: bad dtype %d", dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
