// BatchNorm statistics plumbing around the fused convolutions (training-mode SyncBN semantics,
// reference: pixelssl/nn/module/third_party/sync_batchnorm/batchnorm.py:48-78, 113-125).
//
// Forward: the producing conv's epilogue accumulates per-channel [sum, sum^2] in fp32 into one of
// `nrep` replicas of the [2C] statistics vector (replica = tile row % nrep: spreads same-address
// atomics, measured 23 ns each when 2000 deep); pxl_bn_finalize folds the replicas and turns them
// into the affine (scale, shift) that the CONSUMER kernels apply on load.  For SyncBN the caller
// folds first (pxl_bn_fold_replicas), all-reduces the [2C] vector over RCCL, then finalizes.
// Backward: a reduce pass (sum dz', sum dz'*xhat) with the same replica scheme, a tiny finalize,
// and an apply pass producing the gradient w.r.t. the raw conv output.  All passes are HBM-bound:
// 16 B per lane along channels (NHWC); every thread keeps its channel chunk fixed and walks rows so
// the per-channel coefficients live in registers.
#include "common.h"

namespace {

// coef layout [4][C]: mean, rstd, scale (= gamma*rstd), shift (= beta - mean*scale)
__global__ void bn_finalize_kernel(int C, const float* __restrict__ stats, int nrep, float total_count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                   float eps, int training, int clamp_var, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    float s1 = 0.f, s2 = 0.f;
    if (nrep > 16) {
      // many replicas (PXL_DETERMINISTIC: one per 64 pixel rows, thousands for the first layers): a sequential fp32 sum of
      // that length costs accuracy in var = E[x^2] - mean^2 (513 x 513 logits 1.5e-3 off the reference against 5.8e-4 with
      // four replicas); fp64 partial sums are exact to fp32 here and still a fixed order
      double d1 = 0.0, d2 = 0.0;
      for (int r = 0; r < nrep; ++r) { d1 += (double)stats[(size_t)r * 2 * C + c]; d2 += (double)stats[(size_t)r * 2 * C + C + c]; }
      s1 = (float)d1; s2 = (float)d2;
    } else {
#pragma unroll 8
      for (int r = 0; r < nrep; ++r) { s1 += stats[(size_t)r * 2 * C + c]; s2 += stats[(size_t)r * 2 * C + C + c]; }
    }
    mean = s1 / total_count;
    var = s2 / total_count - mean * mean;
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

// dgamma += sums[C..2C), dbeta += sums[0..C): the LOCAL backward sums (multi-rank: taken before the all-reduce, so
// that the rank-average of the parameter gradients is the global-batch gradient)
__global__ void bn_param_grad_kernel(int C, const float* __restrict__ sums, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dgamma) dgamma[c] += sums[C + c];
  if (dbeta) dbeta[c] += sums[c];
}

// buf[0][i] = sum_r buf[r][i]  (i < n)
__global__ void fold_replicas_kernel(int n, int nrep, float* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (nrep > 16) {           // (see bn_finalize_kernel: long folds in fp64, same fixed order)
    double d = 0.0;
    for (int r = 0; r < nrep; ++r) d += (double)buf[(size_t)r * n + i];
    buf[i] = (float)d;
    return;
  }
  float s = 0.f;
  for (int r = 0; r < nrep; ++r) s += buf[(size_t)r * n + i];
  buf[i] = s;
}

// sums[rep][0..C) += sum_m dzh ; sums[rep][C..2C) += sum_m dzh * xhat     dzh = dz * (relu ? z>0 : 1)
// One LDS reduction over the row lanes and ONE atomic per (channel, sum) per block: 2*8*CG atomics per block
// instead of 2*C (the previous row-major blocks issued 2 M atomics for an 8712 x 1024 tensor).
// MASK: the incoming gradient is first passed through the ReLU of a residual join, dz = dout * (mout > 0), and that
// masked gradient is also WRITTEN (g, and g2 = the identity branch's copy): relu_mask + reduce in one pass.
template <typename T, bool MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int M, int C, const T* __restrict__ dz,
                                                            const T* __restrict__ y,
                                                            const float* __restrict__ coef, int relu,
                                                            float* __restrict__ sums, int nrep, int rows_per_group,
                                                            int cgmax, const T* __restrict__ mout, T* __restrict__ g,
                                                            T* __restrict__ g2) {
  constexpr int EPC = Elem<T>::EPC;
  const ColGeom geo = col_geom(C, EPC, cgmax);
  const ColGeom& gq = geo;
  const int ccol = threadIdx.x % gq.cg, rlane = threadIdx.x / gq.cg;
  const int cc = blockIdx.x * gq.cg + ccol;
  __shared__ float red[256 * 2 * EPC];            // [rl][cg][2*EPC]
  float mean[EPC], rstd[EPC], sc[EPC], sh[EPC], a1[EPC], a2[EPC];
  load_cvec<EPC>(coef + cc * EPC, mean);
  load_cvec<EPC>(coef + C + cc * EPC, rstd);
  load_cvec<EPC>(coef + 2 * C + cc * EPC, sc);
  load_cvec<EPC>(coef + 3 * C + cc * EPC, sh);
#pragma unroll
  for (int e = 0; e < EPC; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(M, m_begin + rows_per_group);
  auto accum = [&](const uint4& vd, const uint4& vy, size_t o) {
    float fd[EPC], fy[EPC];
    Chunk<T>::unpack(vd, fd);
    Chunk<T>::unpack(vy, fy);
    if constexpr (MASK) {
      float fo[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(mout + o), fo);
#pragma unroll
      for (int e = 0; e < EPC; ++e) fd[e] = fo[e] > 0.f ? fd[e] : 0.f;
      const uint4 v = Chunk<T>::pack(fd);
      *reinterpret_cast<uint4*>(g + o) = v;
      if (g2 != nullptr) *reinterpret_cast<uint4*>(g2 + o) = v;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float gd = fd[e];
      if (relu && !(fy[e] * sc[e] + sh[e] > 0.f)) gd = 0.f;
      a1[e] += gd;
      a2[e] += gd * (fy[e] - mean[e]) * rstd[e];
    }
  };
  int m = m_begin + rlane;
  for (; m + gq.rl < m_end; m += 2 * gq.rl) {        // two rows in flight per thread
    const size_t o0 = (size_t)m * C + cc * EPC, o1 = (size_t)(m + gq.rl) * C + cc * EPC;
    const uint4 d0 = *reinterpret_cast<const uint4*>(dz + o0), y0 = *reinterpret_cast<const uint4*>(y + o0);
    const uint4 d1 = *reinterpret_cast<const uint4*>(dz + o1), y1 = *reinterpret_cast<const uint4*>(y + o1);
    accum(d0, y0, o0);
    accum(d1, y1, o1);
  }
  if (m < m_end) {
    const size_t o0 = (size_t)m * C + cc * EPC;
    accum(*reinterpret_cast<const uint4*>(dz + o0), *reinterpret_cast<const uint4*>(y + o0), o0);
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[(rlane * gq.cg + ccol) * 2 * EPC + e] = a1[e];
    red[(rlane * gq.cg + ccol) * 2 * EPC + EPC + e] = a2[e];
  }
  __syncthreads();
  // thread t < cg * 2 * EPC sums one (column, which, element) over the row lanes
  const int nout = gq.cg * 2 * EPC;
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int col = o / (2 * EPC), w = o % (2 * EPC);
    float v = 0.f;
    for (int r = 0; r < gq.rl; ++r) v += red[(r * gq.cg + col) * 2 * EPC + w];
    // enough replicas for one per row group: every (replica, channel) receives exactly ONE add (bit-reproducible sums once the
    // replicas are folded in index order: PXL_DETERMINISTIC); fewer: spread the contention
    float* rep = sums + (size_t)(nrep >= (int)gridDim.y ? blockIdx.y : (blockIdx.y + blockIdx.x) % nrep) * 2 * C;
    const int c = (blockIdx.x * gq.cg + col) * EPC + (w % EPC);
    atomicAdd(rep + (w / EPC) * C + c, v);
  }
}

// bcoef[0..C) = sum dzh / count ; bcoef[C..2C) = sum dzh*xhat / count ; accumulates dgamma, dbeta.
__global__ void bn_bwd_finalize_kernel(int C, const float* __restrict__ sums, int nrep, float total_count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ bcoef, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int r = 0; r < nrep; ++r) { s1 += sums[(size_t)r * 2 * C + c]; s2 += sums[(size_t)r * 2 * C + C + c]; }
  // eval-mode BN (running statistics, freeze_bn) is a fixed affine: no batch-mean terms in dy
  bcoef[c] = training ? s1 / total_count : 0.f;
  bcoef[C + c] = training ? s2 / total_count : 0.f;
  if (dgamma) dgamma[c] += s2;
  if (dbeta) dbeta[c] += s1;
}

// dy = scale * (dzh - c1 - xhat * c2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int C, const T* __restrict__ dz,
                                                           const T* __restrict__ y,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ bcoef, int relu,
                                                           T* __restrict__ dy) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const int cpb = cpr < 256 ? cpr : 256;
  const int rpb = 256 / cpb;
  const int ccol = threadIdx.x % cpb, rlane = threadIdx.x / cpb;
  for (int c0 = 0; c0 < cpr; c0 += cpb) {
    const int cc = c0 + ccol;
    if (cc >= cpr || rlane >= rpb) continue;
    float mean[EPC], rstd[EPC], sc[EPC], sh[EPC], b1[EPC], b2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = cc * EPC + e;
      mean[e] = coef[c]; rstd[e] = coef[C + c]; sc[e] = coef[2 * C + c]; sh[e] = coef[3 * C + c];
      b1[e] = bcoef[c]; b2[e] = bcoef[C + c];
    }
    for (int m = blockIdx.x * rpb + rlane; m < M; m += gridDim.x * rpb) {
      const size_t o = (size_t)m * C + cc * EPC;
      float fd[EPC], fy[EPC], v[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(dz + o), fd);
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(y + o), fy);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float g = fd[e];
        if (relu && !(fy[e] * sc[e] + sh[e] > 0.f)) g = 0.f;
        v[e] = sc[e] * (g - b1[e] - (fy[e] - mean[e]) * rstd[e] * b2[e]);
      }
      *reinterpret_cast<uint4*>(dy + o) = Chunk<T>::pack(v);
    }
  }
}

// dy = scale * (dzh - c1 - xhat*c2) with (c1, c2) = sums / count computed in the kernel's prologue (sums must be
// a single [2C] vector: fold + all-reduce happen before): removes the bn_bwd_finalize launch per layer.  The
// blocks of row group 0 also accumulate dgamma / dbeta.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(int M, int C, const T* dz,
                                                                 const T* __restrict__ y,
                                                                 const float* __restrict__ coef,
                                                                 const float* __restrict__ sums, float inv_count,
                                                                 int training, int relu, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, T* dy,
                                                                 int rows_per_group, int cgmax) {
  constexpr int EPC = Elem<T>::EPC;
  const ColGeom g = col_geom(C, EPC, cgmax);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  // dy = sc*(gd - b1 - (y - mean)*rstd*b2) = sc*gd - k0 - y*k1 with k1 = sc*rstd*b2, k0 = sc*b1 - mean*k1
  float sc[EPC], sh[EPC], k0[EPC], k1[EPC];
  {
    float mean[EPC], rstd[EPC], s1[EPC], s2[EPC];
    load_cvec<EPC>(coef + cc * EPC, mean);
    load_cvec<EPC>(coef + C + cc * EPC, rstd);
    load_cvec<EPC>(coef + 2 * C + cc * EPC, sc);
    load_cvec<EPC>(coef + 3 * C + cc * EPC, sh);
    load_cvec<EPC>(sums + cc * EPC, s1);
    load_cvec<EPC>(sums + C + cc * EPC, s2);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float b1 = training ? s1[e] * inv_count : 0.f;      // eval-mode BN is a fixed affine: no batch-mean terms
      const float b2 = training ? s2[e] * inv_count : 0.f;
      k1[e] = sc[e] * rstd[e] * b2;
      k0[e] = sc[e] * b1 - mean[e] * k1[e];
    }
    if (blockIdx.y == 0 && rlane == 0) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        if (dgamma) dgamma[cc * EPC + e] += s2[e];
        if (dbeta) dbeta[cc * EPC + e] += s1[e];
      }
    }
  }
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(M, m_begin + rows_per_group);
  auto one = [&](const uint4& vd, const uint4& vy) -> uint4 {
    float fd[EPC], fy[EPC], v[EPC];
    Chunk<T>::unpack(vd, fd);
    Chunk<T>::unpack(vy, fy);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float gd = fd[e];
      if (relu && !(fy[e] * sc[e] + sh[e] > 0.f)) gd = 0.f;
      v[e] = sc[e] * gd - k0[e] - fy[e] * k1[e];
    }
    return Chunk<T>::pack(v);
  };
  const size_t col = (size_t)cc * EPC, step = (size_t)g.rl * C;
  int m = m_begin + rlane;
  for (; m + 3 * g.rl < m_end; m += 4 * g.rl) {           // four rows (8 x 16-byte loads) in flight per thread
    const size_t o0 = (size_t)m * C + col, o1 = o0 + step, o2 = o1 + step, o3 = o2 + step;
    const uint4 d0 = *reinterpret_cast<const uint4*>(dz + o0), y0 = *reinterpret_cast<const uint4*>(y + o0);
    const uint4 d1 = *reinterpret_cast<const uint4*>(dz + o1), y1 = *reinterpret_cast<const uint4*>(y + o1);
    const uint4 d2 = *reinterpret_cast<const uint4*>(dz + o2), y2 = *reinterpret_cast<const uint4*>(y + o2);
    const uint4 d3 = *reinterpret_cast<const uint4*>(dz + o3), y3 = *reinterpret_cast<const uint4*>(y + o3);
    *reinterpret_cast<uint4*>(dy + o0) = one(d0, y0);
    *reinterpret_cast<uint4*>(dy + o1) = one(d1, y1);
    *reinterpret_cast<uint4*>(dy + o2) = one(d2, y2);
    *reinterpret_cast<uint4*>(dy + o3) = one(d3, y3);
  }
  for (; m < m_end; m += g.rl) {
    const size_t o0 = (size_t)m * C + col;
    *reinterpret_cast<uint4*>(dy + o0) = one(*reinterpret_cast<const uint4*>(dz + o0), *reinterpret_cast<const uint4*>(y + o0));
  }
}

inline int row_grid(long M, int C, int epc, int cap) {
  const int cpr = C / epc;
  const int rpb = 256 / (cpr < 256 ? cpr : 256);
  long g = (M + rpb - 1) / rpb;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int pxl_bn_finalize(int C, const float* stats, int nrep, float count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int training, int clamp_var, float* coef,
                               void* stream) {
  PXL_REQUIRE(C > 0 && coef && nrep >= 1, "bn_finalize: bad argument");
  PXL_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     C, stats, nrep, count, gamma, beta, running_mean, running_var, momentum, eps, training,
                     clamp_var, coef);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_param_grad(int C, const float* sums, float* dgamma, float* dbeta, void* stream) {
  PXL_REQUIRE(C > 0 && sums, "bn_param_grad: bad argument");
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), C,
                     sums, dgamma, dbeta);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_fold_replicas(int n, int nrep, float* buf, void* stream) {
  PXL_REQUIRE(n > 0 && nrep >= 1 && buf, "bn_fold_replicas: bad argument");
  if (nrep == 1) return PXL_OK;
  hipLaunchKernelGGL(fold_replicas_kernel, dim3(cdiv(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     n, nrep, buf);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
int launch_bwd_reduce(int dtype, int M, int C, const void* dz, const void* y, const float* coef, int relu, float* sums,
                      int nrep, const void* mout, void* g, void* g2, void* stream) {
  const int epc = dtype == PXL_F32 ? 4 : 8;
  const int cgmax = pxl_tune_get(4);
  const ColGeom geo = col_geom(C, epc, cgmax);
  // every row group ends in one atomic per channel: cap the row groups at 256 (narrow tensors have 1-2 column groups)
  const int rpg = rows_per_group(M, geo, min(pxl_tune_get(0), 256 * geo.ncg));
  const dim3 grid(geo.ncg, cdiv(M, rpg));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32) {
    if (mout) hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, true>), grid, dim3(256), 0, s, M, C, (const float*)dz, (const float*)y,
                                 coef, relu, sums, nrep, rpg, cgmax, (const float*)mout, (float*)g, (float*)g2);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, false>), grid, dim3(256), 0, s, M, C, (const float*)dz, (const float*)y,
                            coef, relu, sums, nrep, rpg, cgmax, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
  } else {
    if (mout) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, true>), grid, dim3(256), 0, s, M, C, (const bf16_t*)dz,
                                 (const bf16_t*)y, coef, relu, sums, nrep, rpg, cgmax, (const bf16_t*)mout, (bf16_t*)g, (bf16_t*)g2);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, false>), grid, dim3(256), 0, s, M, C, (const bf16_t*)dz,
                            (const bf16_t*)y, coef, relu, sums, nrep, rpg, cgmax, (const bf16_t*)nullptr, (bf16_t*)nullptr,
                            (bf16_t*)nullptr);
  }
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
}  // namespace

extern "C" int pxl_bn_bwd_reduce(int dtype, int M, int C, const void* dz, const void* y, const float* coef, int relu,
                                 float* sums, int nrep, void* stream) {
  PXL_REQUIRE(dz && y && coef && sums && M > 0 && nrep >= 1, "bn_bwd_reduce: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "bn_bwd_reduce: bad dtype %d", dtype);
  PXL_REQUIRE(C % (dtype == PXL_F32 ? 4 : 8) == 0, "bn_bwd_reduce: C=%d is not 16-byte aligned", C);
  return launch_bwd_reduce(dtype, M, C, dz, y, coef, relu, sums, nrep, nullptr, nullptr, nullptr, stream);
}

// Backward of the bottleneck join out = relu(bn3(y) + res) fused with bn3's reduction: g = dout * (out > 0) is written
// to `g` (and `g2`, the residual branch's copy, may be NULL) and sums[0..C) += sum g, sums[C..2C) += sum g * xhat(y).
extern "C" int pxl_residual_bwd_reduce_rep(int dtype, int M, int C, const void* dout, const void* out, const void* y,
                                           const float* coef, void* g, void* g2, float* sums, int nrep, void* stream) {
  PXL_REQUIRE(dout && out && y && coef && g && sums && M > 0 && nrep >= 1, "residual_bwd_reduce: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "residual_bwd_reduce: bad dtype %d", dtype);
  PXL_REQUIRE(C % (dtype == PXL_F32 ? 4 : 8) == 0, "residual_bwd_reduce: C=%d is not 16-byte aligned", C);
  return launch_bwd_reduce(dtype, M, C, dout, y, coef, 0, sums, nrep, out, g, g2, stream);
}

extern "C" int pxl_residual_bwd_reduce(int dtype, int M, int C, const void* dout, const void* out, const void* y,
                                       const float* coef, void* g, void* g2, float* sums, void* stream) {
  return pxl_residual_bwd_reduce_rep(dtype, M, C, dout, out, y, coef, g, g2, sums, 1, stream);
}

extern "C" int pxl_bn_bwd_apply_fused(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                                      const float* sums, float count, int training, int relu, float* dgamma,
                                      float* dbeta, void* dy, void* stream) {
  PXL_REQUIRE(dz && y && coef && sums && dy && M > 0 && count > 0.f, "bn_bwd_apply_fused: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "bn_bwd_apply_fused: bad dtype %d", dtype);
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_bwd_apply_fused: C=%d must be a multiple of %d", C, epc);
  const int cgmax = pxl_tune_get(4);
  const ColGeom g = col_geom(C, epc, cgmax);
  const int rpg = rows_per_group(M, g, pxl_tune_get(1));
  const dim3 grid(g.ncg, cdiv(M, rpg));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<float>, grid, dim3(256), 0, s, M, C, (const float*)dz, (const float*)y,
                       coef, sums, 1.f / count, training, relu, dgamma, dbeta, (float*)dy, rpg, cgmax);
  else
    hipLaunchKernelGGL(bn_bwd_apply_fused_kernel<bf16_t>, grid, dim3(256), 0, s, M, C, (const bf16_t*)dz,
                       (const bf16_t*)y, coef, sums, 1.f / count, training, relu, dgamma, dbeta, (bf16_t*)dy, rpg, cgmax);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_bwd_finalize(int C, const float* sums, int nrep, float count, float* dgamma, float* dbeta,
                                   float* bcoef, int training, void* stream) {
  PXL_REQUIRE(C > 0 && sums && bcoef && nrep >= 1, "bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), C, sums, nrep, count, dgamma, dbeta, bcoef, training);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_bwd_apply(int dtype, int M, int C, const void* dz, const void* y, const float* coef,
                                const float* bcoef, int relu, void* dy, void* stream) {
  PXL_REQUIRE(dz && y && coef && bcoef && dy && M > 0, "bn_bwd_apply: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "bn_bwd_apply: bad dtype %d", dtype);
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_bwd_apply: C=%d must be a multiple of %d", C, epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int blocks = row_grid(M, C, epc, 2048);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(blocks), dim3(256), 0, s, M, C, (const float*)dz,
                       (const float*)y, coef, bcoef, relu, (float*)dy);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, M, C, (const bf16_t*)dz,
                       (const bf16_t*)y, coef, bcoef, relu, (bf16_t*)dy);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
