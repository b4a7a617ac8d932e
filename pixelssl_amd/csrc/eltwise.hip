// HBM-bound element-wise / pooling kernels of the ResNet trunk (NHWC, 16 B per lane).
//   * residual join:   out = relu( bn3(y3) + (bnd(yd) | identity) )      resnet.py:44-48
//   * its backward mask: g = dout * (out > 0)
//   * stem max-pool 3x3/s2/p1 with the stem BN+ReLU fused into the load   resnet.py:71-73
#include "common.h"

namespace {

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

// BN finalize folded into a consumer's prologue: the block derives (scale, shift) of channels [c0, c0 + nch), nch <= 128,
// into lsc / lsh (LDS); `writer` blocks also store coef [4C] and update the running statistics (same arithmetic as
// bn_finalize_kernel in bn.hip).  Ends with a __syncthreads().
__device__ __forceinline__ void block_bn_finalize(const pxl_bn_fin& f, int C, int c0, int nch, bool writer, float* lsum,
                                                  float* lsc, float* lsh) {
  if (f.training) {
    for (int t = threadIdx.x; t < 2 * nch; t += 256) {      // (which, channel): coalesced over the channel index
      const int w = t / nch, j = t - w * nch;
      float s = 0.f;
      for (int r = 0; r < f.nrep; ++r) s += f.stats[(size_t)r * 2 * C + (size_t)w * C + c0 + j];
      lsum[t] = s;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nch; j += 256) {
    const int c = c0 + j;
    float mean, var;
    if (f.training) {
      mean = lsum[j] / f.count;
      var = lsum[nch + j] / f.count - mean * mean;
      if (var < 0.f) var = 0.f;
      if (writer && f.running_mean != nullptr) {
        const float unbiased = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
      }
    } else {
      mean = f.running_mean[c];
      var = f.running_var[c];
    }
    const float rstd = f.clamp_var ? rsqrtf(fmaxf(var, f.eps)) : rsqrtf(var + f.eps);
    const float ga = f.gamma ? f.gamma[c] : 1.f, be = f.beta ? f.beta[c] : 0.f;
    const float scale = ga * rstd, shift = be - mean * scale;
    lsc[j] = scale;
    lsh[j] = shift;
    if (writer) {
      f.coef[c] = mean; f.coef[C + c] = rstd; f.coef[2 * C + c] = scale; f.coef[3 * C + c] = shift;
    }
  }
  __syncthreads();
}

// Per-channel-affine element-wise kernels: column-group blocks (common.h: col_geom), the channel chunk is FIXED
// per thread so the coefficients are loaded once into registers, two rows in flight per thread.  FIN: the BN finalize
// of the operand(s) is folded into the prologue (yfin / rfin) instead of reading ready-made coefficients.
// second operand set of a PAIRED launch (gridDim.z == 2: the same op of a second network, pxl_net_forward_pair)
struct EltSecond { const void* y; const void* res; void* out; pxl_bn_fin yfin, rfin; unsigned char* bits; };

// one bit per element of a packed 16-byte chunk: element e of Chunk<T>::unpack's order is non-zero (the join's post-ReLU output
// is >= 0, so "non-zero" IS its ReLU mask: what the fused join backward of conv_dma_kernel.h needs from that tensor)
template <typename T> __device__ __forceinline__ unsigned nz_bits(const uint4& v);
template <> __device__ __forceinline__ unsigned nz_bits<bf16_t>(const uint4& v) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
  unsigned b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    b |= ((w[k] & 0x7fffu) != 0u ? 1u : 0u) << (2 * k);
    b |= ((w[k] & 0x7fff0000u) != 0u ? 1u : 0u) << (2 * k + 1);
  }
  return b;
}
template <> __device__ __forceinline__ unsigned nz_bits<float>(const uint4& v) {
  return ((v.x & 0x7fffffffu) != 0u ? 1u : 0u) | ((v.y & 0x7fffffffu) != 0u ? 2u : 0u) | ((v.z & 0x7fffffffu) != 0u ? 4u : 0u) |
         ((v.w & 0x7fffffffu) != 0u ? 8u : 0u);
}

// FIN kernels: request the first rows before the finalize prologue?  (PXL_ELT_PREFETCH; measured per workload)
static bool elt_prefetch() {
  static const bool on = [] { const char* e = getenv("PXL_ELT_PREFETCH"); return e != nullptr && e[0] == '1'; }();
  return on;
}

template <typename T, bool FIN, int PFN = 0>
__global__ __launch_bounds__(256) void residual_fwd_kernel(int M, int C, const T* y,
                                                           const float* __restrict__ ycoef,
                                                           const T* res,
                                                           const float* __restrict__ rcoef,
                                                           T* out, int rows_per_group, int cgmax,
                                                           pxl_bn_fin yfin, pxl_bn_fin rfin, int has_rfin, EltSecond sec,
                                                           unsigned char* bits) {
  constexpr int EPC = Elem<T>::EPC;
  if (blockIdx.z != 0) {        // paired launch (pxl_net_forward_pair): the same join of the second network
    y = static_cast<const T*>(sec.y); res = static_cast<const T*>(sec.res); out = static_cast<T*>(sec.out);
    yfin = sec.yfin; rfin = sec.rfin; bits = sec.bits;
  }
  const ColGeom g = col_geom(C, EPC, cgmax);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  // bits (optional): the ReLU mask of the output, one BYTE per 16-byte chunk ([M][C / EPC]): the fused join backward reads 1/16
  // of what the tensor itself costs (17.8 MB -> 1.1 MB per stage-3 join)
  const int cpr = C / EPC;
  auto put = [&](size_t o, int row, const uint4& r) {
    *reinterpret_cast<uint4*>(out + o) = r;
    if (bits != nullptr) bits[(size_t)row * cpr + cc] = (unsigned char)nz_bits<T>(r);
  };
  // FIN: the first rows of both operands are requested BEFORE the finalize prologue (statistics loads + two barriers):
  // the launch is latency-bound (a few rows per thread), so the two round trips overlap instead of adding up
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(M, m_begin + rows_per_group);
  const size_t col = (size_t)cc * EPC, step = (size_t)g.rl * C;
  int m = m_begin + rlane;
  constexpr int PF = FIN ? PFN : 0;
  uint4 pa[PF > 0 ? PF : 1], pb[PF > 0 ? PF : 1];
  if constexpr (FIN) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if (m + k * g.rl < m_end) {
        pa[k] = *reinterpret_cast<const uint4*>(y + (size_t)(m + k * g.rl) * C + col);
        pb[k] = *reinterpret_cast<const uint4*>(res + (size_t)(m + k * g.rl) * C + col);
      }
    }
  }
  float ys[EPC], yb[EPC], rs[EPC], rb[EPC];
  if constexpr (FIN) {
    __shared__ float lsum[256], lsc[128], lsh[128];
    const int nch = g.cg * EPC, c0 = blockIdx.x * nch;
    block_bn_finalize(yfin, C, c0, nch, blockIdx.y == 0, lsum, lsc, lsh);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { ys[e] = lsc[ccol * EPC + e]; yb[e] = lsh[ccol * EPC + e]; }
    if (has_rfin) {
      __syncthreads();
      block_bn_finalize(rfin, C, c0, nch, blockIdx.y == 0, lsum, lsc, lsh);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { rs[e] = lsc[ccol * EPC + e]; rb[e] = lsh[ccol * EPC + e]; }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { rs[e] = 1.f; rb[e] = 0.f; }
    }
  } else {
    load_cvec<EPC>(ycoef + 2 * C + cc * EPC, ys);
    load_cvec<EPC>(ycoef + 3 * C + cc * EPC, yb);
    if (rcoef) {
      load_cvec<EPC>(rcoef + 2 * C + cc * EPC, rs);
      load_cvec<EPC>(rcoef + 3 * C + cc * EPC, rb);
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { rs[e] = 1.f; rb[e] = 0.f; }
    }
  }
  auto one = [&](const uint4& vy, const uint4& vr) -> uint4 {
    float fy[EPC], fr[EPC], v[EPC];
    Chunk<T>::unpack(vy, fy);
    Chunk<T>::unpack(vr, fr);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = fmaxf(fy[e] * ys[e] + yb[e] + (fr[e] * rs[e] + rb[e]), 0.f);
    return Chunk<T>::pack(v);
  };
  if constexpr (FIN) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (m + k * g.rl < m_end) put((size_t)(m + k * g.rl) * C + col, m + k * g.rl, one(pa[k], pb[k]));
    m += PF * g.rl;
  }
  for (; m + g.rl < m_end; m += 2 * g.rl) {
    const size_t o0 = (size_t)m * C + col, o1 = o0 + step;
    const uint4 a0 = *reinterpret_cast<const uint4*>(y + o0), b0 = *reinterpret_cast<const uint4*>(res + o0);
    const uint4 a1 = *reinterpret_cast<const uint4*>(y + o1), b1 = *reinterpret_cast<const uint4*>(res + o1);
    put(o0, m, one(a0, b0));
    put(o1, m + g.rl, one(a1, b1));
  }
  if (m < m_end) {
    const size_t o0 = (size_t)m * C + col;
    put(o0, m, one(*reinterpret_cast<const uint4*>(y + o0), *reinterpret_cast<const uint4*>(res + o0)));
  }
}

// z = relu?(y*scale + shift): the activated tensor the LDS-DMA convolutions (conv_dma.hip, wgrad) read directly
template <typename T, bool FIN, int PFN = 0>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(int M, int C, const T* y,
                                                           const float* __restrict__ coef, int relu,
                                                           T* z, int rows_per_group, int cgmax,
                                                           pxl_bn_fin fin, EltSecond sec) {
  constexpr int EPC = Elem<T>::EPC;
  if (blockIdx.z != 0) { y = static_cast<const T*>(sec.y); z = static_cast<T*>(sec.out); fin = sec.yfin; }
  const ColGeom g = col_geom(C, EPC, cgmax);
  const int ccol = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
  const int cc = blockIdx.x * g.cg + ccol;
  const int m_begin = blockIdx.y * rows_per_group;
  const int m_end = min(M, m_begin + rows_per_group);
  const size_t col = (size_t)cc * EPC, step = (size_t)g.rl * C;
  int m = m_begin + rlane;
  // FIN: first rows requested before the finalize prologue (see residual_fwd_kernel)
  constexpr int PF = FIN ? PFN : 0;
  uint4 pa[PF > 0 ? PF : 1];
  if constexpr (FIN) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (m + k * g.rl < m_end) pa[k] = *reinterpret_cast<const uint4*>(y + (size_t)(m + k * g.rl) * C + col);
  }
  float sc[EPC], sh[EPC];
  if constexpr (FIN) {
    __shared__ float lsum[256], lsc[128], lsh[128];
    const int nch = g.cg * EPC;
    block_bn_finalize(fin, C, blockIdx.x * nch, nch, blockIdx.y == 0, lsum, lsc, lsh);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = lsc[ccol * EPC + e]; sh[e] = lsh[ccol * EPC + e]; }
  } else {
    load_cvec<EPC>(coef + 2 * C + cc * EPC, sc);
    load_cvec<EPC>(coef + 3 * C + cc * EPC, sh);
  }
  auto one = [&](const uint4& vy) -> uint4 {
    float f[EPC];
    Chunk<T>::unpack(vy, f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float v = f[e] * sc[e] + sh[e];
      f[e] = relu ? fmaxf(v, 0.f) : v;
    }
    return Chunk<T>::pack(f);
  };
  if constexpr (FIN) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (m + k * g.rl < m_end) *reinterpret_cast<uint4*>(z + (size_t)(m + k * g.rl) * C + col) = one(pa[k]);
    m += PF * g.rl;
  }
  for (; m + g.rl < m_end; m += 2 * g.rl) {
    const size_t o0 = (size_t)m * C + col, o1 = o0 + step;
    const uint4 a0 = *reinterpret_cast<const uint4*>(y + o0);
    const uint4 a1 = *reinterpret_cast<const uint4*>(y + o1);
    *reinterpret_cast<uint4*>(z + o0) = one(a0);
    *reinterpret_cast<uint4*>(z + o1) = one(a1);
  }
  if (m < m_end) {
    const size_t o0 = (size_t)m * C + col;
    *reinterpret_cast<uint4*>(z + o0) = one(*reinterpret_cast<const uint4*>(y + o0));
  }
}

// LeakyReLU of the discriminator / flaw-detector conv stacks (ssl_adv.py:474-487, ssl_gct.py:567-585)
template <typename T>
__global__ __launch_bounds__(256) void leaky_fwd_kernel(long nchunks, const T* __restrict__ x, float slope,
                                                        T* __restrict__ y) {
  constexpr int EPC = Elem<T>::EPC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[EPC];
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
    for (int e = 0; e < EPC; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * slope;
    reinterpret_cast<uint4*>(y)[i] = Chunk<T>::pack(f);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void leaky_bwd_kernel(long nchunks, const T* __restrict__ dy,
                                                        const T* __restrict__ x, float slope, T* __restrict__ dx) {
  constexpr int EPC = Elem<T>::EPC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float fd[EPC], fx[EPC];
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(dy)[i], fd);
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(x)[i], fx);
#pragma unroll
    for (int e = 0; e < EPC; ++e) fd[e] = fx[e] > 0.f ? fd[e] : fd[e] * slope;
    reinterpret_cast<uint4*>(dx)[i] = Chunk<T>::pack(fd);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(long nchunks, const T* __restrict__ dout,
                                                        const T* __restrict__ out, T* __restrict__ g,
                                                        T* __restrict__ g2) {
  constexpr int EPC = Elem<T>::EPC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float fd[EPC], fo[EPC];
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(dout)[i], fd);
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(out)[i], fo);
#pragma unroll
    for (int e = 0; e < EPC; ++e) fd[e] = fo[e] > 0.f ? fd[e] : 0.f;
    const uint4 v = Chunk<T>::pack(fd);
    reinterpret_cast<uint4*>(g)[i] = v;
    if (g2 != nullptr) reinterpret_cast<uint4*>(g2)[i] = v;
  }
}

// out[c] += sum_m x[m][c]  for c < Creal (bias gradients: ASPP head, discriminator / flaw-detector convolutions).
// blockIdx.y selects a 256-column slab (pitches > 256); inside a slab `cw` columns x 256/cw row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int M, int Cp, int Creal, const T* __restrict__ x,
                                                     float* __restrict__ out, int rows_per_block) {
  // 16-byte chunks: `cg` chunk columns x 256/cg row lanes per block (blockIdx.y selects a slab of 256 chunk columns for very
  // wide rows); the row lanes are folded through LDS, one atomic per channel and block.  (The first version read one ELEMENT
  // per lane -- 2-byte loads for bf16 -- and reached ~1 TB/s: 71 us per bias gradient of GCT's flaw detector.)
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float red[256 * EPC];
  const int nch = Cp / EPC;
  const int ch0 = blockIdx.y * 256;
  const int cg = min(256, nch - ch0);
  const int rl = 256 / cg;
  const int ccol = threadIdx.x % cg, rlane = threadIdx.x / cg;
  const int m_end = min(M, (int)(blockIdx.x + 1) * rows_per_block);
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  if (rlane < rl) {
    const T* __restrict__ col = x + (size_t)(ch0 + ccol) * EPC;
    for (int m = blockIdx.x * rows_per_block + rlane; m < m_end; m += rl) {
      float f[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(col + (size_t)m * Cp), f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[(rlane * cg + ccol) * EPC + e] = acc[e];      // (rlane * cg + ccol == threadIdx.x when rlane < rl)
  __syncthreads();
  for (int j = threadIdx.x; j < cg * EPC; j += 256) {
    float v = 0.f;
    for (int r = 0; r < rl; ++r) v += red[r * cg * EPC + j];
    const int c = ch0 * EPC + j;
    if (c < Creal) atomicAdd(out + c, v);
  }
}

__global__ void vec_sum4_kernel(int n, float* __restrict__ out, const float* a, const float* b, const float* c,
                                const float* d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f) + (c ? c[i] : 0.f) + (d ? d[i] : 0.f);
}

// a += b (gradient accumulation for multi-consumer tensors)
template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(long nchunks, T* __restrict__ a, const T* __restrict__ b) {
  constexpr int EPC = Elem<T>::EPC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float fa[EPC], fb[EPC];
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(a)[i], fa);
    Chunk<T>::unpack(reinterpret_cast<const uint4*>(b)[i], fb);
#pragma unroll
    for (int e = 0; e < EPC; ++e) fa[e] += fb[e];
    reinterpret_cast<uint4*>(a)[i] = Chunk<T>::pack(fa);
  }
}

// out[b][py][px][c] = max over 3x3 window (stride 2, pad 1) of relu(y*scale+shift); idx = first argmax
// in row-major window order (PyTorch's tie rule: strict '>' keeps the first maximum).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                          const T* __restrict__ y,
                                                          const float* __restrict__ coef,
                                                          T* __restrict__ out, uint8_t* __restrict__ idx) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long total = (long)B * Ho * Wo * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    long r = i / cpr;
    const int px = (int)(r % Wo); r /= Wo;
    const int py = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float sc[EPC], sh[EPC], best[EPC];
    int bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sc[e] = coef ? coef[2 * C + cc * EPC + e] : 1.f;
      sh[e] = coef ? coef[3 * C + cc * EPC + e] : 0.f;
      best[e] = -INFINITY;
      bi[e] = 0;
    }
#pragma unroll
    for (int wy = 0; wy < 3; ++wy) {
#pragma unroll
      for (int wx = 0; wx < 3; ++wx) {
        const int iy = py * 2 - 1 + wy, ix = px * 2 - 1 + wx;
        if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
          float f[EPC];
          Chunk<T>::unpack(*reinterpret_cast<const uint4*>(y + ((size_t)(b * Hi + iy) * Wi + ix) * C + cc * EPC), f);
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float z = f[e] * sc[e] + sh[e];
            if (coef) z = fmaxf(z, 0.f);
            if (z > best[e]) { best[e] = z; bi[e] = wy * 3 + wx; }
          }
        }
      }
    }
    *reinterpret_cast<uint4*>(out + ((size_t)(b * Ho + py) * Wo + px) * C + cc * EPC) = Chunk<T>::pack(best);
    uint8_t* ip = idx + ((size_t)(b * Ho + py) * Wo + px) * C + cc * EPC;
#pragma unroll
    for (int e = 0; e < EPC; ++e) ip[e] = (uint8_t)bi[e];
  }
}

// dz[b][iy][ix][c] = sum over the (<=2x2) windows containing (iy,ix) whose argmax is (iy,ix) of dp
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(int B, int Hi, int Wi, int C, int Ho, int Wo,
                                                          const T* __restrict__ dp,
                                                          const uint8_t* __restrict__ idx, T* __restrict__ dz) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long total = (long)B * Hi * Wi * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    long r = i / cpr;
    const int ix = (int)(r % Wi); r /= Wi;
    const int iy = (int)(r % Hi);
    const int b = (int)(r / Hi);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    // windows containing iy: py*2-1 <= iy <= py*2+1  ->  py in [iy/2, (iy+1)/2]
    for (int py = iy >> 1; py <= ((iy + 1) >> 1); ++py) {
      if (py < 0 || py >= Ho) continue;
      const int wy = iy - (py * 2 - 1);
      if (wy < 0 || wy > 2) continue;
      for (int px = ix >> 1; px <= ((ix + 1) >> 1); ++px) {
        if (px < 0 || px >= Wo) continue;
        const int wx = ix - (px * 2 - 1);
        if (wx < 0 || wx > 2) continue;
        const int code = wy * 3 + wx;
        const size_t o = ((size_t)(b * Ho + py) * Wo + px) * C + cc * EPC;
        float f[EPC];
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(dp + o), f);
        const uint8_t* ip = idx + o;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
          if (ip[e] == code) acc[e] += f[e];
      }
    }
    *reinterpret_cast<uint4*>(dz + ((size_t)(b * Hi + iy) * Wi + ix) * C + cc * EPC) = Chunk<T>::pack(acc);
  }
}

}  // namespace

template <typename T> static const T* cp(const void* p) { return reinterpret_cast<const T*>(p); }
template <typename T> static T* mp(void* p) { return reinterpret_cast<T*>(p); }

inline int row_grid(long M, int C, int epc) {
  const int cpr = C / epc;
  const int rpb = 256 / (cpr < 256 ? cpr : 256);
  long g = (M + rpb - 1) / rpb;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int pxl_residual_fwd_bits(int dtype, long M, int C, const void* y, const float* ycoef, const void* res,
                                     const float* rcoef, void* out, void* bits, void* stream);
extern "C" int pxl_residual_fwd(int dtype, long M, int C, const void* y, const float* ycoef, const void* res,
                                const float* rcoef, void* out, void* stream) {
  return pxl_residual_fwd_bits(dtype, M, C, y, ycoef, res, rcoef, out, nullptr, stream);
}
// ... + the ReLU mask of `out` as one byte per 16-byte chunk (bits [M][C / (8 bf16 | 4 fp32)], may be NULL): conv_dma_kernel.h's
// fused join backward reads it instead of the tensor
extern "C" int pxl_residual_fwd_bits(int dtype, long M, int C, const void* y, const float* ycoef, const void* res,
                                     const float* rcoef, void* out, void* bits, void* stream) {
  PXL_REQUIRE(y && ycoef && res && out, "residual_fwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "residual_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "residual_fwd: C=%d must be a multiple of %d", C, epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int cgmax = pxl_tune_get(4);
  const ColGeom g = col_geom(C, epc, cgmax);
  const int rpg = rows_per_group((int)M, g, pxl_tune_get(2));
  const dim3 grid(g.ncg, cdiv((int)M, rpg));
  const pxl_bn_fin none = {};
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((residual_fwd_kernel<float, false>), grid, dim3(256), 0, s, (int)M, C,
                       cp<float>(y), ycoef, cp<float>(res), rcoef, mp<float>(out), rpg, cgmax, none, none, 0, EltSecond{},
                       reinterpret_cast<unsigned char*>(bits));
  else
    hipLaunchKernelGGL((residual_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, (int)M, C,
                       cp<bf16_t>(y), ycoef, cp<bf16_t>(res), rcoef, mp<bf16_t>(out), rpg, cgmax, none, none, 0, EltSecond{},
                       reinterpret_cast<unsigned char*>(bits));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
inline bool fin_ok(const pxl_bn_fin* f) {
  return f && f->coef && f->count > 0.f && (f->training ? (f->stats != nullptr && f->nrep >= 1) : (f->running_mean && f->running_var));
}
}  // namespace

// ---- paired launches of the two finalize-folding kernels (pxl_net_forward_pair).  Inside a pxl_elt_pair_begin / _end
// bracket the FIRST eligible call is held back; a second call of the same kind and shape is issued together with it as one
// launch (gridDim.z = 2); anything else (or the end of the bracket) issues the held call on its own.
namespace {
struct EltHeld {
  bool armed = false, held = false;
  int kind = 0, dtype = 0, C = 0, relu = 0, has_rfin = 0; long M = 0; hipStream_t s = nullptr;
  const void* y = nullptr; const void* res = nullptr; void* out = nullptr; pxl_bn_fin yfin{}, rfin{};
  unsigned char* bits = nullptr;
};
thread_local EltHeld tl_elt;

int launch_residual_fin(int dtype, long M, int C, const void* y, const pxl_bn_fin& yfin, const void* res, const pxl_bn_fin& rf, int has_rfin,
                        void* out, hipStream_t s, const EltSecond* sec, unsigned char* bits = nullptr) {
  const int epc = dtype == PXL_F32 ? 4 : 8;
  const int cgmax = 16;                         // 16 chunks x EPC <= 128 channels per block (LDS coefficient arrays)
  const ColGeom g = col_geom(C, epc, cgmax);
  const int rpg = rows_per_group((int)M, g, pxl_tune_get(2));
  const dim3 grid(g.ncg, cdiv((int)M, rpg), sec ? 2 : 1);
  const EltSecond e2 = sec ? *sec : EltSecond{};
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((elt_prefetch() ? residual_fwd_kernel<float, true, 4> : residual_fwd_kernel<float, true, 0>), grid, dim3(256), 0, s, (int)M, C, cp<float>(y), nullptr,
                       cp<float>(res), nullptr, mp<float>(out), rpg, cgmax, yfin, rf, has_rfin, e2, bits);
  else
    hipLaunchKernelGGL((elt_prefetch() ? residual_fwd_kernel<bf16_t, true, 4> : residual_fwd_kernel<bf16_t, true, 0>), grid, dim3(256), 0, s, (int)M, C, cp<bf16_t>(y), nullptr,
                       cp<bf16_t>(res), nullptr, mp<bf16_t>(out), rpg, cgmax, yfin, rf, has_rfin, e2, bits);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
int launch_bn_fin_apply(int dtype, long M, int C, const void* y, const pxl_bn_fin& fin, int relu, void* z, hipStream_t s, const EltSecond* sec) {
  const int epc = dtype == PXL_F32 ? 4 : 8;
  const int cgmax = 16;
  const ColGeom g = col_geom(C, epc, cgmax);
  const int rpg = rows_per_group((int)M, g, pxl_tune_get(3));
  const dim3 grid(g.ncg, cdiv((int)M, rpg), sec ? 2 : 1);
  const EltSecond e2 = sec ? *sec : EltSecond{};
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((elt_prefetch() ? bn_apply_fwd_kernel<float, true, 4> : bn_apply_fwd_kernel<float, true, 0>), grid, dim3(256), 0, s, (int)M, C, cp<float>(y), nullptr, relu,
                       mp<float>(z), rpg, cgmax, fin, e2);
  else
    hipLaunchKernelGGL((elt_prefetch() ? bn_apply_fwd_kernel<bf16_t, true, 4> : bn_apply_fwd_kernel<bf16_t, true, 0>), grid, dim3(256), 0, s, (int)M, C, cp<bf16_t>(y), nullptr, relu,
                       mp<bf16_t>(z), rpg, cgmax, fin, e2);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
int issue_held() {
  EltHeld& h = tl_elt;
  if (!h.held) return PXL_OK;
  h.held = false;
  return h.kind == 1 ? launch_residual_fin(h.dtype, h.M, h.C, h.y, h.yfin, h.res, h.rfin, h.has_rfin, h.out, h.s, nullptr, h.bits)
                     : launch_bn_fin_apply(h.dtype, h.M, h.C, h.y, h.yfin, h.relu, h.out, h.s, nullptr);
}
bool same_fin_shape(const pxl_bn_fin& a, const pxl_bn_fin& b) {
  return a.nrep == b.nrep && a.count == b.count && a.momentum == b.momentum && a.eps == b.eps && a.training == b.training &&
         a.clamp_var == b.clamp_var;
}
}  // namespace

extern "C" void pxl_elt_pair_begin(void) { tl_elt.armed = true; tl_elt.held = false; }
extern "C" int pxl_elt_pair_end(void) { tl_elt.armed = false; return issue_held(); }

extern "C" int pxl_residual_finalize_fwd_bits(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                                              const pxl_bn_fin* rfin, void* out, void* bits, void* stream);
extern "C" int pxl_residual_finalize_fwd(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                                         const pxl_bn_fin* rfin, void* out, void* stream) {
  return pxl_residual_finalize_fwd_bits(dtype, M, C, y, yfin, res, rfin, out, nullptr, stream);
}
extern "C" int pxl_residual_finalize_fwd_bits(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                                              const pxl_bn_fin* rfin, void* out, void* bits_, void* stream) {
  unsigned char* bits = reinterpret_cast<unsigned char*>(bits_);
  PXL_REQUIRE(y && res && out && fin_ok(yfin) && (rfin == nullptr || fin_ok(rfin)), "residual_finalize_fwd: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "residual_finalize_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "residual_finalize_fwd: C=%d must be a multiple of %d", C, epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const pxl_bn_fin rf = rfin ? *rfin : pxl_bn_fin{};
  EltHeld& h = tl_elt;
  if (h.armed) {
    if (h.held && h.kind == 1 && h.dtype == dtype && h.M == M && h.C == C && h.s == s && h.has_rfin == (rfin ? 1 : 0) &&
        same_fin_shape(h.yfin, *yfin) && (!rfin || same_fin_shape(h.rfin, rf))) {
      h.held = false;
      const EltSecond sec{y, res, out, *yfin, rf, bits};
      return launch_residual_fin(dtype, M, C, h.y, h.yfin, h.res, h.rfin, h.has_rfin, h.out, s, &sec, h.bits);
    }
    const int rc = issue_held();
    if (rc != PXL_OK) return rc;
    h.held = true; h.kind = 1; h.dtype = dtype; h.M = M; h.C = C; h.s = s; h.has_rfin = rfin ? 1 : 0;
    h.y = y; h.res = res; h.out = out; h.yfin = *yfin; h.rfin = rf; h.bits = bits;
    return PXL_OK;
  }
  return launch_residual_fin(dtype, M, C, y, *yfin, res, rf, rfin ? 1 : 0, out, s, nullptr, bits);
}

extern "C" int pxl_bn_apply_fwd(int dtype, long M, int C, const void* y, const float* coef, int relu, void* z,
                                void* stream) {
  PXL_REQUIRE(y && coef && z, "bn_apply_fwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "bn_apply_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_apply_fwd: C=%d must be a multiple of %d", C, epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int cgmax = pxl_tune_get(4);
  const ColGeom g = col_geom(C, epc, cgmax);
  const int rpg = rows_per_group((int)M, g, pxl_tune_get(3));
  const dim3 grid(g.ncg, cdiv((int)M, rpg));
  const pxl_bn_fin none = {};
  if (dtype == PXL_F32)
    hipLaunchKernelGGL((bn_apply_fwd_kernel<float, false>), grid, dim3(256), 0, s, (int)M, C, cp<float>(y),
                       coef, relu, mp<float>(z), rpg, cgmax, none, EltSecond{});
  else
    hipLaunchKernelGGL((bn_apply_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, (int)M, C,
                       cp<bf16_t>(y), coef, relu, mp<bf16_t>(z), rpg, cgmax, none, EltSecond{});
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_bn_finalize_apply_fwd(int dtype, long M, int C, const void* y, const pxl_bn_fin* fin, int relu, void* z,
                                         void* stream) {
  PXL_REQUIRE(y && z && fin_ok(fin), "bn_finalize_apply_fwd: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "bn_finalize_apply_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "bn_finalize_apply_fwd: C=%d must be a multiple of %d", C, epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  EltHeld& h = tl_elt;
  if (h.armed) {
    if (h.held && h.kind == 2 && h.dtype == dtype && h.M == M && h.C == C && h.s == s && h.relu == relu && same_fin_shape(h.yfin, *fin)) {
      h.held = false;
      const EltSecond sec{y, nullptr, z, *fin, pxl_bn_fin{}, nullptr};
      return launch_bn_fin_apply(dtype, M, C, h.y, h.yfin, relu, h.out, s, &sec);
    }
    const int rc = issue_held();
    if (rc != PXL_OK) return rc;
    h.held = true; h.kind = 2; h.dtype = dtype; h.M = M; h.C = C; h.s = s; h.relu = relu; h.y = y; h.out = z; h.yfin = *fin;
    return PXL_OK;
  }
  return launch_bn_fin_apply(dtype, M, C, y, *fin, relu, z, s, nullptr);
}

extern "C" int pxl_leaky_fwd(int dtype, long n, const void* x, float slope, void* y, void* stream) {
  PXL_REQUIRE(x && y && n > 0, "leaky_fwd: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "leaky_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(n % epc == 0, "leaky_fwd: n must be a multiple of %d", epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long nchunks = n / epc;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(leaky_fwd_kernel<float>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, cp<float>(x), slope, mp<float>(y));
  else
    hipLaunchKernelGGL(leaky_fwd_kernel<bf16_t>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, cp<bf16_t>(x), slope, mp<bf16_t>(y));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_leaky_bwd(int dtype, long n, const void* dy, const void* x, float slope, void* dx, void* stream) {
  PXL_REQUIRE(dy && x && dx && n > 0, "leaky_bwd: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "leaky_bwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(n % epc == 0, "leaky_bwd: n must be a multiple of %d", epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long nchunks = n / epc;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(leaky_bwd_kernel<float>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, cp<float>(dy), cp<float>(x), slope, mp<float>(dx));
  else
    hipLaunchKernelGGL(leaky_bwd_kernel<bf16_t>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks, cp<bf16_t>(dy), cp<bf16_t>(x), slope, mp<bf16_t>(dx));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_relu_mask(int dtype, long n, const void* dout, const void* out, void* g, void* g2,
                             void* stream) {
  PXL_REQUIRE(dout && out && g, "relu_mask: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "relu_mask: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(n % epc == 0, "relu_mask: n must be a multiple of %d", epc);
  const long nchunks = n / epc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks,
                       cp<float>(dout), cp<float>(out), mp<float>(g), mp<float>(g2));
  else
    hipLaunchKernelGGL(relu_mask_kernel<bf16_t>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks,
                       cp<bf16_t>(dout), cp<bf16_t>(out), mp<bf16_t>(g), mp<bf16_t>(g2));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

namespace {
int colsum_impl(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream, bool one_block);
}
extern "C" int pxl_colsum(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream) {
  return colsum_impl(dtype, M, Cp, Creal, x, out, stream, false);
}
// the same with ONE block per 256-chunk column slab: every out[c] receives one add (bit-reproducible; PXL_DETERMINISTIC)
extern "C" int pxl_colsum_ordered(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream) {
  return colsum_impl(dtype, M, Cp, Creal, x, out, stream, true);
}
namespace {
int colsum_impl(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream, bool one_block) {
  PXL_REQUIRE(x && out && M > 0, "colsum: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "colsum: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(Cp >= epc && Cp % epc == 0 && Creal <= Cp, "colsum: channel pitch %d must be a multiple of %d (16-byte chunks)", Cp, epc);
  const int nch = Cp / epc;
  const int rl = 256 / (nch < 256 ? nch : 256);            // rows a block reads per pass
  int blocks = cdiv(M, rl * 8);                            // >= 8 passes per block
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1 || one_block) blocks = 1;
  const int rpb = cdiv(M, blocks);
  blocks = cdiv(M, rpb);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks, cdiv(nch, 256)), dim3(256), 0, s, M, Cp, Creal, cp<float>(x), out, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(blocks, cdiv(nch, 256)), dim3(256), 0, s, M, Cp, Creal, cp<bf16_t>(x), out, rpb);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
}  // namespace

extern "C" int pxl_vec_sum4(int n, float* out, const float* a, const float* b, const float* c, const float* d,
                            void* stream) {
  PXL_REQUIRE(out && n > 0, "vec_sum4: bad argument");
  hipLaunchKernelGGL(vec_sum4_kernel, dim3(cdiv(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n,
                     out, a, b, c, d);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_add_inplace(int dtype, long n, void* a, const void* b, void* stream) {
  PXL_REQUIRE(a && b, "add_inplace: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "add_inplace: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(n % epc == 0, "add_inplace: n must be a multiple of %d", epc);
  const long nchunks = n / epc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks,
                       mp<float>(a), cp<float>(b));
  else
    hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(grid_for(nchunks)), dim3(256), 0, s, nchunks,
                       mp<bf16_t>(a), cp<bf16_t>(b));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_maxpool3x3s2_fwd(int dtype, int B, int Hi, int Wi, int C, const void* y, const float* coef,
                                    void* out, uint8_t* idx, void* stream) {
  PXL_REQUIRE(y && out && idx, "maxpool_fwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "maxpool_fwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "maxpool_fwd: C=%d must be a multiple of %d", C, epc);
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, B, Hi, Wi, C, Ho, Wo,
                       cp<float>(y), coef, mp<float>(out), idx);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, B, Hi, Wi, C, Ho, Wo,
                       cp<bf16_t>(y), coef, mp<bf16_t>(out), idx);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_maxpool3x3s2_bwd(int dtype, int B, int Hi, int Wi, int C, const void* dp, const uint8_t* idx,
                                    void* dz, void* stream) {
  PXL_REQUIRE(dp && idx && dz, "maxpool_bwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "maxpool_bwd: bad dtype");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0, "maxpool_bwd: C=%d must be a multiple of %d", C, epc);
  const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
  const long total = (long)B * Hi * Wi * (C / epc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, B, Hi, Wi, C, Ho, Wo,
                       cp<float>(dp), idx, mp<float>(dz));
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, B, Hi, Wi, C, Ho, Wo,
                       cp<bf16_t>(dp), idx, mp<bf16_t>(dz));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
