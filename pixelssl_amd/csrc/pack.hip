// Weight packing and layout conversion kernels (HBM-bound, 16 B/lane where the layout allows).
#include "common.h"

namespace {

// master fp32 [K][T][C]  ->  fwd  [K][T_total][Cp]  (taps placed at t_off, channels zero-padded)
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, int K, int Tn, int C,
                                T* __restrict__ wf, int Cp, int T_total, int t_off) {
  const long total = (long)K * Tn * Cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const long kt = i / Cp;
    const int t = (int)(kt % Tn);
    const int k = (int)(kt / Tn);
    const float v = c < C ? w[((long)k * Tn + t) * C + c] : 0.f;
    wf[((long)k * T_total + t_off + t) * Cp + c] = from_f<T>(v);
  }
}

// master fp32 [K][T][C]  ->  dgrad [C][T_total][Kp] (transposed per tap through a 32x32 LDS tile,
// out channels zero-padded).  grid = (ceil(C/32), ceil(Kp/32), T)
template <typename T>
__global__ void pack_t_kernel(const float* __restrict__ w, int K, int Tn, int C,
                              T* __restrict__ wt, int Kp, int T_total, int t_off) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int c0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, c = c0 + tx;
    tile[r][tx] = (k < K && c < C) ? w[((long)k * Tn + t) * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, k = k0 + tx;
    if (c < C && k < Kp) wt[((long)c * T_total + t_off + t) * Kp + k] = from_f<T>(tile[tx][r]);
  }
}

// ---- batched variants: one launch packs up to 64 weight tensors (the executor repacks ~110 convs after
// every optimizer / EMA step; per-tensor launches cost ~430 launches per step)
struct PackBatch {
  int n;
  pxl_pack_item it[64];
  int start[65];        // first block of each item
};

template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_batched_kernel(const float* __restrict__ params,
                                                               unsigned char* __restrict__ packed, const PackBatch b) {
  int lo = 0, hi = b.n - 1;
  while (lo < hi) {                       // last item with start <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (b.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pxl_pack_item it = b.it[lo];
  const float* __restrict__ w = params + it.src_off;
  T* __restrict__ wf = reinterpret_cast<T*>(packed + it.wf_off);
  const long total = (long)it.K * it.T * it.Cp;
  // no channel padding, no tap offset (every convolution with Cin % 32 == 0): the pack is a contiguous cast -- 8 elements
  // per thread as two 16-byte loads and one (bf16) or two (fp32) 16-byte stores.  The element-wise path below ran at
  // 1.4 TB/s (index div/mod per element, 2-byte stores) and the forward pass waits for it after every optimizer step.
  if (it.C == it.Cp && it.T == it.T_total && it.t_off == 0 && (total & 7) == 0 && (it.src_off & 3) == 0 &&
      (it.wf_off & 15) == 0) {
    const long i = (((long)blockIdx.x - b.start[lo]) * 256 + threadIdx.x) * 8;
    if (i < total) {
      const float4 a = *reinterpret_cast<const float4*>(w + i), c4 = *reinterpret_cast<const float4*>(w + i + 4);
      const float f[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(wf + i) = Chunk<bf16_t>::pack(f);
      } else {
        *reinterpret_cast<float4*>(wf + i) = a;
        *reinterpret_cast<float4*>(wf + i + 4) = c4;
      }
    }
    return;
  }
  const long i0 = ((long)blockIdx.x - b.start[lo]) * 2048 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const long i = i0 + r * 256;
    if (i >= total) break;
    const int c = (int)(i % it.Cp);
    const long kt = i / it.Cp;
    const int t = (int)(kt % it.T);
    const int k = (int)(kt / it.T);
    const float v = c < it.C ? w[((long)k * it.T + t) * it.C + c] : 0.f;
    wf[((long)k * it.T_total + it.t_off + t) * it.Cp + c] = from_f<T>(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_t_batched_kernel(const float* __restrict__ params,
                                                             unsigned char* __restrict__ packed, const PackBatch b) {
  __shared__ float tile[32][33];
  int lo = 0, hi = b.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const pxl_pack_item it = b.it[lo];
  const float* __restrict__ w = params + it.src_off;
  T* __restrict__ wt = reinterpret_cast<T*>(packed + it.wt_off);
  int rem = (int)blockIdx.x - b.start[lo];
  const int tc = (it.C + 31) / 32, tk = (it.Kp + 31) / 32;
  const int c0 = (rem % tc) * 32; rem /= tc;
  const int k0 = (rem % tk) * 32;
  const int t = rem / tk;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, c = c0 + tx;
    tile[r][tx] = (k < it.K && c < it.C) ? w[((long)k * it.T + t) * it.C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, k = k0 + tx;
    if (c < it.C && k < it.Kp) wt[((long)c * it.T_total + it.t_off + t) * it.Kp + k] = from_f<T>(tile[tx][r]);
  }
}

// x NCHW fp32 [B][C][H][W] (the channels optionally gathered from up to 4 tensors: concatenation along C on load, no torch.cat
// copy) -> NHWC [B][H][W][Cp] (zero padded channels).  One thread per (pixel, 16-byte channel chunk), pixel fastest: a wave
// reads 64 consecutive pixels of one channel plane per load (unit stride) and writes one 16-byte chunk per lane.  (The first
// version ran one thread per PIXEL over all Cp channels: fine for the 3-channel image, 246 us for CCT's 512-channel latent
// -- 17 blocks of 2-byte stores -- and 274 us for GCT's 24-channel flaw-detector input.)
struct NchwParts { const float* src[4]; int chans[4]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void nchw_parts_to_nhwc_kernel(const NchwParts ps, T* __restrict__ y, int B, long HW, int Cp) {
  constexpr int EPC = Elem<T>::EPC;
  const int nch = Cp / EPC;
  const long total = (long)B * HW * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const long p = i % HW;
    const long r = i / HW;
    const int chunk = (int)(r % nch);
    const long b = r / nch;
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      int c = chunk * EPC + e, k = 0;
      while (k < ps.n && c >= ps.chans[k]) { c -= ps.chans[k]; ++k; }
      v[e] = k < ps.n ? ps.src[k][(b * ps.chans[k] + c) * HW + p] : 0.f;
    }
    *reinterpret_cast<uint4*>(y + (b * HW + p) * Cp + chunk * EPC) = Chunk<T>::pack(v);
  }
}

// Stem patches (im2col of a few-channel input, once per forward): x NCHW fp32 [B][C][H][W] -> P [B*Ho*Wo][Kp] in the
// engine dtype, P[m][(ky*kw + kx)*C + c] = x[b][c][oy*stride - pad + ky][ox*stride - pad + kx] (0 outside the image and
// for k >= kh*kw*C).  The 7x7 / stride-2 / 3-channel stem then IS a 1x1 convolution over Kp = 192 channels for the
// LDS-DMA kernels (forward and weight gradient); the k order equals the master weight layout [Cout][kh][kw][C].
template <typename T>
__global__ __launch_bounds__(256) void stem_patches_kernel(const float* __restrict__ x, T* __restrict__ P, int B, int C, int H,
                                                           int W, int kh, int kw, int stride, int pad, int Ho, int Wo, int Kp) {
  const int K = kh * kw * C;
  const int chunks = Kp / 8;
  const long total = (long)B * Ho * Wo * chunks;
  const long plane = (long)H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    const long m = i / chunks;
    const int ox = (int)(m % Wo);
    const int oy = (int)((m / Wo) % Ho);
    const long b = m / ((long)Wo * Ho);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    float v[8];
    int k = ch * 8;
    int tap = k / C, c = k - tap * C;
    int ky = tap / kw, kx = tap - ky * kw;
#pragma unroll
    for (int e = 0; e < 8; ++e, ++k) {
      float f = 0.f;
      if (k < K) {
        const int iy = iy0 + ky, ix = ix0 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) f = x[(b * C + c) * plane + (long)iy * W + ix];
      }
      v[e] = f;
      if (++c == C) { c = 0; if (++kx == kw) { kx = 0; ++ky; } }
    }
    T* dst = P + m * Kp + ch * 8;
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(dst) = Chunk<bf16_t>::pack(v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = from_f<T>(v[e]);
    }
  }
}
// (Two LDS-transposing variants of this kernel -- lane <-> pixel gathers into a [pixels][Kp] tile, 8 and 24 loads in
// flight -- were measured at 365-400 us against 134-146 us for this direct one, both networks running it at once; the
// direct form stays.)

// NHWC [B][H][W][Cp] (first C channels) -> NCHW fp32, one tensor per part (NULL = that part is not wanted).  One thread per
// (pixel, 16-byte chunk), pixel fastest: every lane reads its chunk with one 16-byte load and the wave writes 64 consecutive
// pixels of each channel plane (unit stride).
struct NchwOutParts { float* dst[4]; int chans[4]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_parts_kernel(const T* __restrict__ x, const NchwOutParts ps, int B, int C, long HW,
                                                                 int Cp) {
  constexpr int EPC = Elem<T>::EPC;
  const int nch = (C + EPC - 1) / EPC;
  const long total = (long)B * HW * nch;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const long p = i % HW;
    const long r = i / HW;
    const int chunk = (int)(r % nch);
    const long b = r / nch;
    float v[EPC];
    Chunk<T>::unpack(*reinterpret_cast<const uint4*>(x + (b * HW + p) * Cp + chunk * EPC), v);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      int c = chunk * EPC + e, k = 0;
      if (c >= C) break;
      while (k < ps.n && c >= ps.chans[k]) { c -= ps.chans[k]; ++k; }
      if (k < ps.n && ps.dst[k] != nullptr) ps.dst[k][(b * ps.chans[k] + c) * HW + p] = v[e];
    }
  }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f2bf(x[i]);
}

inline int grid_for(long n, int block = 256) {
  long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

template <typename T>
static int pack_weights_impl(const float* w, int K, int Tn, int C, T* wf, int Cp, int T_total, int t_off,
                             T* wt, int Kp, hipStream_t s) {
  hipLaunchKernelGGL(pack_fwd_kernel<T>, dim3(grid_for((long)K * Tn * Cp)), dim3(256), 0, s, w, K, Tn, C,
                     wf, Cp, T_total, t_off);
  PXL_LAUNCH_CHECK();
  if (wt != nullptr) {
    hipLaunchKernelGGL(pack_t_kernel<T>, dim3(cdiv(C, 32), cdiv(Kp, 32), Tn), dim3(256), 0, s, w, K, Tn, C,
                       wt, Kp, T_total, t_off);
    PXL_LAUNCH_CHECK();
  }
  return PXL_OK;
}

extern "C" int pxl_pack_weights(int dtype, const float* w, int K, int T, int C, void* wf, int Cp,
                                int T_total, int t_off, void* wt, int Kp, void* stream) {
  PXL_REQUIRE(w && wf, "pack_weights: null argument");
  PXL_REQUIRE(Cp >= C && (wt == nullptr || Kp >= K) && t_off + T <= T_total, "pack_weights: bad padding");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    return pack_weights_impl<float>(w, K, T, C, (float*)wf, Cp, T_total, t_off, (float*)wt, Kp, s);
  if (dtype == PXL_BF16)
    return pack_weights_impl<bf16_t>(w, K, T, C, (bf16_t*)wf, Cp, T_total, t_off, (bf16_t*)wt, Kp, s);
  return pxl_set_error(PXL_ERR_ARG, "pack_weights: bad dtype %d", dtype);
}

extern "C" int pxl_pack_weights_batched(int dtype, const float* params, void* packed, const pxl_pack_item* items,
                                        int n, void* stream) {
  PXL_REQUIRE(params && packed && items && n >= 0, "pack_weights_batched: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "pack_weights_batched: bad dtype %d", dtype);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  unsigned char* pk = reinterpret_cast<unsigned char*>(packed);
  for (int base = 0; base < n; base += 64) {
    const int cnt = n - base < 64 ? n - base : 64;
    PackBatch f, t;
    f.n = 0; t.n = 0;
    int fb = 0, tb = 0;
    for (int i = 0; i < cnt; ++i) {
      const pxl_pack_item& it = items[base + i];
      PXL_REQUIRE(it.Cp >= it.C && it.t_off + it.T <= it.T_total, "pack_weights_batched: bad padding in item %d", base + i);
      if (it.wf_off >= 0) {              // (wf_off < 0: only the transposed copy of this tensor is wanted)
        f.it[f.n] = it;
        f.start[f.n] = fb;
        fb += (int)(((long)it.K * it.T * it.Cp + 2047) / 2048);
        ++f.n;
      }
      if (it.wt_off >= 0) {
        PXL_REQUIRE(it.Kp >= it.K, "pack_weights_batched: bad Kp in item %d", base + i);
        t.it[t.n] = it;
        t.start[t.n] = tb;
        tb += ((it.C + 31) / 32) * ((it.Kp + 31) / 32) * it.T;
        ++t.n;
      }
    }
    f.start[f.n] = fb;
    t.start[t.n] = tb;
    if (fb > 0) {
      if (dtype == PXL_F32) hipLaunchKernelGGL(pack_fwd_batched_kernel<float>, dim3(fb), dim3(256), 0, s, params, pk, f);
      else hipLaunchKernelGGL(pack_fwd_batched_kernel<bf16_t>, dim3(fb), dim3(256), 0, s, params, pk, f);
      PXL_LAUNCH_CHECK();
    }
    if (tb > 0) {
      if (dtype == PXL_F32) hipLaunchKernelGGL(pack_t_batched_kernel<float>, dim3(tb), dim3(256), 0, s, params, pk, t);
      else hipLaunchKernelGGL(pack_t_batched_kernel<bf16_t>, dim3(tb), dim3(256), 0, s, params, pk, t);
      PXL_LAUNCH_CHECK();
    }
  }
  return PXL_OK;
}

namespace {
int launch_to_nhwc(int dtype, const NchwParts& ps, void* y, int B, int H, int W, int Cp, hipStream_t s, const char* who) {
  const long HW = (long)H * W;
  const int epc = dtype == PXL_F32 ? 4 : 8;
  if (Cp % epc != 0) return pxl_set_error(PXL_ERR_ARG, "%s: channel pitch %d is not a multiple of %d (16-byte chunks)", who, Cp, epc);
  const long total = (long)B * HW * (Cp / epc);
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(nchw_parts_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, s, ps, (float*)y, B, HW, Cp);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(nchw_parts_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, ps, (bf16_t*)y, B, HW, Cp);
  else
    return pxl_set_error(PXL_ERR_ARG, "%s: bad dtype %d", who, dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
int launch_to_nchw(int dtype, const void* x, const NchwOutParts& ps, int B, int C, int H, int W, int Cp, hipStream_t s,
                   const char* who) {
  const long HW = (long)H * W;
  const int epc = dtype == PXL_F32 ? 4 : 8;
  if (Cp % epc != 0) return pxl_set_error(PXL_ERR_ARG, "%s: channel pitch %d is not a multiple of %d (16-byte chunks)", who, Cp, epc);
  const long total = (long)B * HW * ((C + epc - 1) / epc);
  const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_parts_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ps, B, C, HW, Cp);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_parts_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ps, B, C, HW, Cp);
  else
    return pxl_set_error(PXL_ERR_ARG, "%s: bad dtype %d", who, dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
}  // namespace

extern "C" int pxl_nchw_to_nhwc(int dtype, const float* x, void* y, int B, int C, int H, int W, int Cp,
                                void* stream) {
  PXL_REQUIRE(x && y && Cp >= C && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
  NchwParts ps;
  ps.n = 1;
  for (int k = 0; k < 4; ++k) { ps.src[k] = k == 0 ? x : nullptr; ps.chans[k] = k == 0 ? C : 0; }
  return launch_to_nhwc(dtype, ps, y, B, H, W, Cp, reinterpret_cast<hipStream_t>(stream), "nchw_to_nhwc");
}

extern "C" int pxl_stem_patches(int dtype, const float* x, void* P, int B, int C, int H, int W, int kh, int kw, int stride,
                                int pad, int Ho, int Wo, int Kp, void* stream) {
  PXL_REQUIRE(x && P && B > 0 && C > 0 && kh > 0 && kw > 0 && stride > 0 && Ho > 0 && Wo > 0, "stem_patches: bad argument");
  PXL_REQUIRE(Kp % 8 == 0 && Kp >= kh * kw * C, "stem_patches: pitch %d does not hold %d x %d x %d", Kp, kh, kw, C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * Ho * Wo * (Kp / 8);
  int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(stem_patches_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float*)P, B, C, H, W, kh, kw, stride, pad, Ho,
                       Wo, Kp);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(stem_patches_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t*)P, B, C, H, W, kh, kw, stride, pad,
                       Ho, Wo, Kp);
  else
    return pxl_set_error(PXL_ERR_ARG, "stem_patches: bad dtype %d", dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_nchw_parts_to_nhwc(int dtype, int nparts, const float* const* srcs, const int* chans, void* y, int B,
                                     int H, int W, int Cp, void* stream) {
  PXL_REQUIRE(srcs && chans && y && nparts >= 1 && nparts <= 4 && B > 0 && H > 0 && W > 0, "nchw_parts_to_nhwc: bad argument");
  NchwParts ps;
  ps.n = nparts;
  int C = 0;
  for (int k = 0; k < 4; ++k) {
    ps.src[k] = k < nparts ? srcs[k] : nullptr;
    ps.chans[k] = k < nparts ? chans[k] : 0;
    if (k < nparts) { PXL_REQUIRE(srcs[k] && chans[k] > 0, "nchw_parts_to_nhwc: empty part %d", k); C += chans[k]; }
  }
  PXL_REQUIRE(Cp >= C, "nchw_parts_to_nhwc: %d channels do not fit the pitch %d", C, Cp);
  return launch_to_nhwc(dtype, ps, y, B, H, W, Cp, reinterpret_cast<hipStream_t>(stream), "nchw_parts_to_nhwc");
}

extern "C" int pxl_nhwc_to_nchw_parts(int dtype, const void* x, int nparts, float* const* dsts, const int* chans, int B,
                                      int H, int W, int Cp, void* stream) {
  PXL_REQUIRE(x && dsts && chans && nparts >= 1 && nparts <= 4 && B > 0 && H > 0 && W > 0, "nhwc_to_nchw_parts: bad argument");
  NchwOutParts ps;
  ps.n = nparts;
  int C = 0;
  for (int k = 0; k < 4; ++k) {
    ps.dst[k] = k < nparts ? dsts[k] : nullptr;
    ps.chans[k] = k < nparts ? chans[k] : 0;
    if (k < nparts) { PXL_REQUIRE(chans[k] > 0, "nhwc_to_nchw_parts: empty part %d", k); C += chans[k]; }
  }
  PXL_REQUIRE(Cp >= C, "nhwc_to_nchw_parts: %d channels exceed the pitch %d", C, Cp);
  return launch_to_nchw(dtype, x, ps, B, C, H, W, Cp, reinterpret_cast<hipStream_t>(stream), "nhwc_to_nchw_parts");
}

extern "C" int pxl_nhwc_to_nchw(int dtype, const void* x, float* y, int B, int C, int H, int W, int Cp,
                                void* stream) {
  PXL_REQUIRE(x && y && Cp >= C && B > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad argument");
  NchwOutParts ps;
  ps.n = 1;
  for (int k = 0; k < 4; ++k) { ps.dst[k] = k == 0 ? y : nullptr; ps.chans[k] = k == 0 ? C : 0; }
  return launch_to_nchw(dtype, x, ps, B, C, H, W, Cp, reinterpret_cast<hipStream_t>(stream), "nhwc_to_nchw");
}

namespace {
// Input pipeline, device side (task/sseg/data.py:150-182 `Normalize` + `ToTensor`): uint8 HWC crops -> normalised fp32 NCHW.
// numpy semantics of the reference, bit for bit: `img /= 255.0` is a float32 division; `img -= mean` and `img /= std`
// take float64 tuples, i.e. compute in double and round to float32 after each step.
__global__ __launch_bounds__(256) void normalize_u8_kernel(long HW, int C, const unsigned char* __restrict__ src,
                                                          const double* __restrict__ mean, const double* __restrict__ stdv,
                                                          float* __restrict__ dst) {
  const int b = blockIdx.y;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
    for (int c = 0; c < C; ++c) {
      float v = (float)src[((size_t)b * HW + i) * C + c] / 255.0f;
      v = (float)((double)v - mean[c]);
      v = (float)((double)v / stdv[c]);
      dst[((size_t)b * C + c) * HW + i] = v;
    }
  }
}
__global__ __launch_bounds__(256) void u8_to_f32_kernel(long n, const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                        int marker, float marker_value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int v = src[i];
    dst[i] = v == marker ? marker_value : (float)v;
  }
}
}  // namespace

extern "C" int pxl_normalize_u8(int B, int C, long HW, const unsigned char* src, const double* mean, const double* stdv,
                                float* dst, void* stream) {
  PXL_REQUIRE(src && mean && stdv && dst && B > 0 && C > 0 && C <= 4 && HW > 0, "normalize_u8: bad argument");
  long g = (HW + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3((int)g, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), HW, C, src, mean,
                     stdv, dst);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_u8_to_f32(long n, const unsigned char* src, float* dst, int marker, float marker_value, void* stream) {
  PXL_REQUIRE(src && dst && n > 0, "u8_to_f32: bad argument");
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, src, dst, marker,
                     marker_value);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
