// PSPNet head data movement (NHWC, HBM-bound): adaptive average pooling to a bin grid, bilinear (align_corners=False)
// up-sampling of the pyramid stages straight into their channel slice of the concatenated tensor, channel-slice
// copies, and the sub-pixel decoder's ReLU + PixelShuffle(2).
//   reference: task/sseg/module/_pspnet.py:88-102 (_PSPModule), :41-55 (PixelShuffle block)
#include "common.h"

namespace {

// torch's adaptive pooling windows: [floor(i*H/bin), ceil((i+1)*H/bin))
__host__ __device__ inline int bin_lo(int i, int H, int bin) { return (i * H) / bin; }
__host__ __device__ inline int bin_hi(int i, int H, int bin) { return ((i + 1) * H + bin - 1) / bin; }

// F.interpolate(mode='bilinear', align_corners=False) source coordinate of output index `o`
__device__ __forceinline__ void src_coord(int o, int in, int out, int& i0, int& i1, float& lam) {
  float s = ((float)o + 0.5f) * ((float)in / (float)out) - 0.5f;
  s = fmaxf(s, 0.f);
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = min(i0 + 1, in - 1);
  lam = s - (float)i0;
}

constexpr int RED_Y = 16;   // row slices per block in the window reductions

// block = 64 channel lanes x RED_Y row slices; blockIdx.x = (b, i, j), blockIdx.y = 64-channel group.
// out[b,i,j,c] = mean over the window.
template <typename T>
__global__ __launch_bounds__(64 * RED_Y) void avgpool_fwd_kernel(int H, int W, int Cp, int bin, const T* __restrict__ in,
                                                                 T* __restrict__ out) {
  __shared__ float part[RED_Y][64];
  const int bij = blockIdx.x, j = bij % bin, i = (bij / bin) % bin, b = bij / (bin * bin);
  const int c = blockIdx.y * 64 + threadIdx.x;
  const int y0 = bin_lo(i, H, bin), y1 = bin_hi(i, H, bin), x0 = bin_lo(j, W, bin), x1 = bin_hi(j, W, bin);
  const int ww = x1 - x0, npx = (y1 - y0) * ww;
  float acc = 0.f;
  if (c < Cp)
    for (int p = threadIdx.y; p < npx; p += RED_Y) {
      const int y = y0 + p / ww, x = x0 + p % ww;
      acc += to_f(in[((size_t)(b * H + y) * W + x) * Cp + c]);
    }
  part[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && c < Cp) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RED_Y; ++r) s += part[r][threadIdx.x];
    out[(size_t)bij * Cp + c] = from_f<T>(s / (float)npx);
  }
}

// din[b,y,x,c] (+)= sum over the windows containing (y,x) of dout[b,i,j,c] / area(i,j); one 16-byte chunk per thread
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(int B, int H, int W, int Cp, int bin, const T* __restrict__ dout,
                                                          T* __restrict__ din, int accumulate) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = Cp / EPC;
  const long total = (long)B * H * W * cpr;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
    const int cc = (int)(t % cpr);
    const long pix = t / cpr;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc[EPC];
    if (accumulate) Chunk<T>::unpack(*reinterpret_cast<const uint4*>(din + pix * Cp + cc * EPC), acc);
    else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    }
    for (int i = 0; i < bin; ++i) {
      const int y0 = bin_lo(i, H, bin), y1 = bin_hi(i, H, bin);
      if (y < y0 || y >= y1) continue;
      for (int j = 0; j < bin; ++j) {
        const int x0 = bin_lo(j, W, bin), x1 = bin_hi(j, W, bin);
        if (x < x0 || x >= x1) continue;
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        float g[EPC];
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(dout + ((size_t)(b * bin + i) * bin + j) * Cp + cc * EPC), g);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += g[e] * inv;
      }
    }
    *reinterpret_cast<uint4*>(din + pix * Cp + cc * EPC) = Chunk<T>::pack(acc);
  }
}

// out[b,y,x,c_off+c] = bilinear(relu?(in*scale+shift)) : one 16-byte chunk of the slice per thread
template <typename T>
__global__ __launch_bounds__(256) void upslice_fwd_kernel(int B, int h, int w, int Cpi, int C, const T* __restrict__ in,
                                                          const float* __restrict__ coef, int relu, int H, int W,
                                                          T* __restrict__ out, int Cpo, int c_off) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long total = (long)B * H * W * cpr;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
    const int cc = (int)(t % cpr);
    const long pix = t / cpr;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    int y0, y1, x0, x1; float ly, lx;
    src_coord(y, h, H, y0, y1, ly);
    src_coord(x, w, W, x0, x1, lx);
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sc[e] = coef ? coef[2 * Cpi + cc * EPC + e] : 1.f;
      sh[e] = coef ? coef[3 * Cpi + cc * EPC + e] : 0.f;
    }
    auto tap = [&](int yy, int xx, float* f) {
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(in + ((size_t)(b * h + yy) * w + xx) * Cpi + cc * EPC), f);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float v = f[e] * sc[e] + sh[e];
        f[e] = relu ? fmaxf(v, 0.f) : v;
      }
    };
    float a[EPC], bb[EPC], c2[EPC], d[EPC], o[EPC];
    tap(y0, x0, a); tap(y0, x1, bb); tap(y1, x0, c2); tap(y1, x1, d);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float top = a[e] + lx * (bb[e] - a[e]), bot = c2[e] + lx * (d[e] - c2[e]);
      o[e] = top + ly * (bot - top);
    }
    *reinterpret_cast<uint4*>(out + pix * Cpo + c_off + cc * EPC) = Chunk<T>::pack(o);
  }
}

// gradient of the up-sampling wrt the (activated) low-resolution stage output:
// din[b,i,j,c] = sum_{y,x} wy(y,i) wx(x,j) dout[b,y,x,c_off+c]; same block geometry as the pooling reduction
template <typename T>
__global__ __launch_bounds__(64 * RED_Y) void upslice_bwd_kernel(int h, int w, int Cpi, int C, const T* __restrict__ dout,
                                                                 int H, int W, int Cpo, int c_off, T* __restrict__ din) {
  __shared__ float part[RED_Y][64];
  const int bij = blockIdx.x, j = bij % w, i = (bij / w) % h, b = bij / (h * w);
  const int c = blockIdx.y * 64 + threadIdx.x;
  float acc = 0.f;
  if (c < C)
    for (int p = threadIdx.y; p < H * W; p += RED_Y) {
      const int y = p / W, x = p % W;
      int y0, y1, x0, x1; float ly, lx;
      src_coord(y, h, H, y0, y1, ly);
      float wy = 0.f;
      if (y0 == i) wy += 1.f - ly;
      if (y1 == i) wy += ly;
      if (wy == 0.f) continue;
      src_coord(x, w, W, x0, x1, lx);
      float wx = 0.f;
      if (x0 == j) wx += 1.f - lx;
      if (x1 == j) wx += lx;
      if (wx == 0.f) continue;
      acc += wy * wx * to_f(dout[((size_t)(b * H + y) * W + x) * Cpo + c_off + c]);
    }
  part[threadIdx.y][threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.y == 0 && c < Cpi) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < RED_Y; ++r) s += part[r][threadIdx.x];
    din[(size_t)bij * Cpi + c] = from_f<T>(c < C ? s : 0.f);
  }
}

// dst[m, d_off + c] = src[m, s_off + c] (accumulate: +=) for c < C, 16-byte chunks
template <typename T>
__global__ __launch_bounds__(256) void slice_copy_kernel(long M, int C, const T* __restrict__ src, int Cps, int s_off,
                                                         T* __restrict__ dst, int Cpd, int d_off, int accumulate) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long total = M * cpr;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
    const int cc = (int)(t % cpr);
    const long m = t / cpr;
    uint4 v = *reinterpret_cast<const uint4*>(src + m * Cps + s_off + cc * EPC);
    T* q = dst + m * Cpd + d_off + cc * EPC;
    if (accumulate) {
      float a[EPC], g[EPC];
      Chunk<T>::unpack(v, g);
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(q), a);
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[e] += g[e];
      v = Chunk<T>::pack(a);
    }
    *reinterpret_cast<uint4*>(q) = v;
  }
}

// out[b, 2y+dy, 2x+dx, c] = relu(in[b, y, x, 4c + 2dy + dx]); pad channels of `out` are zeroed
template <typename T>
__global__ __launch_bounds__(256) void pixshuf_fwd_kernel(int B, int h, int w, int Cpi, int C, const T* __restrict__ in,
                                                          T* __restrict__ out, int Cpo) {
  const long total = (long)B * 2 * h * 2 * w * Cpo;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
    const int c = (int)(t % Cpo);
    const long pix = t / Cpo;
    const int X = (int)(pix % (2 * w)), Y = (int)((pix / (2 * w)) % (2 * h)), b = (int)(pix / (4L * w * h));
    float v = 0.f;
    if (c < C) v = fmaxf(to_f(in[((size_t)(b * h + (Y >> 1)) * w + (X >> 1)) * Cpi + 4 * c + 2 * (Y & 1) + (X & 1)]), 0.f);
    out[t] = from_f<T>(v);
  }
}

// din[b, y, x, k] = in > 0 ? dout[b, 2y + (k>>1&1), 2x + (k&1), k>>2] : 0 for k < 4C; pad channels zeroed
template <typename T>
__global__ __launch_bounds__(256) void pixshuf_bwd_kernel(int B, int h, int w, int Cpi, int C, const T* __restrict__ dout,
                                                          int Cpo, const T* __restrict__ in, T* __restrict__ din) {
  const long total = (long)B * h * w * Cpi;
  for (long t = blockIdx.x * 256L + threadIdx.x; t < total; t += gridDim.x * 256L) {
    const int k = (int)(t % Cpi);
    const long pix = t / Cpi;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((long)w * h));
    float v = 0.f;
    if (k < 4 * C && to_f(in[t]) > 0.f)
      v = to_f(dout[((size_t)(b * 2 * h + 2 * y + ((k >> 1) & 1)) * (2 * w) + 2 * x + (k & 1)) * Cpo + (k >> 2)]);
    din[t] = from_f<T>(v);
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define PXL_DISPATCH(dtype, CALL)                                                   \
  do {                                                                              \
    if ((dtype) == PXL_F32) { using T = float; CALL; }                              \
    else if ((dtype) == PXL_BF16) { using T = bf16_t; CALL; }                       \
    else return pxl_set_error(PXL_ERR_ARG, "bad dtype %d", (int)(dtype));           \
  } while (0)

extern "C" int pxl_adaptive_avgpool_fwd(int dtype, int B, int H, int W, int Cp, int bin, const void* in, void* out,
                                        void* stream) {
  PXL_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && Cp > 0 && bin > 0, "adaptive_avgpool_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(B * bin * bin, cdiv(Cp, 64)), block(64, RED_Y);
  PXL_DISPATCH(dtype, (avgpool_fwd_kernel<T><<<grid, block, 0, s>>>(H, W, Cp, bin, (const T*)in, (T*)out)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_adaptive_avgpool_bwd(int dtype, int B, int H, int W, int Cp, int bin, const void* dout, void* din,
                                        int accumulate, void* stream) {
  PXL_REQUIRE(dout && din && B > 0 && H > 0 && W > 0 && Cp > 0 && bin > 0, "adaptive_avgpool_bwd: bad argument");
  PXL_REQUIRE(Cp % (dtype == PXL_F32 ? 4 : 8) == 0, "adaptive_avgpool_bwd: channel pitch %d is not 16-byte aligned", Cp);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * H * W * (Cp / (dtype == PXL_F32 ? 4 : 8));
  PXL_DISPATCH(dtype, (avgpool_bwd_kernel<T><<<grid_for(n), 256, 0, s>>>(B, H, W, Cp, bin, (const T*)dout, (T*)din, accumulate)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_upsample_slice_fwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* in, const float* coef,
                                      int relu, int H, int W, void* out, int Cp_out, int c_off, void* stream) {
  PXL_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_slice_fwd: bad argument");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0 && Cp_in % epc == 0 && Cp_out % epc == 0 && c_off % epc == 0 && c_off + C <= Cp_out && C <= Cp_in,
              "upsample_slice_fwd: slice [%d, %d) of pitch %d / input pitch %d is not 16-byte aligned", c_off, c_off + C, Cp_out, Cp_in);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * H * W * (C / epc);
  PXL_DISPATCH(dtype, (upslice_fwd_kernel<T><<<grid_for(n), 256, 0, s>>>(B, h, w, Cp_in, C, (const T*)in, coef, relu, H, W,
                                                                         (T*)out, Cp_out, c_off)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_upsample_slice_bwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* dout, int H, int W,
                                      int Cp_out, int c_off, void* din, void* stream) {
  PXL_REQUIRE(dout && din && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && c_off + C <= Cp_out && C <= Cp_in,
              "upsample_slice_bwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(B * h * w, cdiv(Cp_in, 64)), block(64, RED_Y);
  PXL_DISPATCH(dtype, (upslice_bwd_kernel<T><<<grid, block, 0, s>>>(h, w, Cp_in, C, (const T*)dout, H, W, Cp_out, c_off, (T*)din)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_slice_copy(int dtype, long M, int C, const void* src, int Cp_src, int s_off, void* dst, int Cp_dst,
                              int d_off, int accumulate, void* stream) {
  PXL_REQUIRE(src && dst && M > 0 && C > 0, "slice_copy: bad argument");
  const int epc = dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(C % epc == 0 && Cp_src % epc == 0 && Cp_dst % epc == 0 && s_off % epc == 0 && d_off % epc == 0 &&
              s_off + C <= Cp_src && d_off + C <= Cp_dst, "slice_copy: slices are not 16-byte aligned / out of range");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = M * (C / epc);
  PXL_DISPATCH(dtype, (slice_copy_kernel<T><<<grid_for(n), 256, 0, s>>>(M, C, (const T*)src, Cp_src, s_off, (T*)dst, Cp_dst,
                                                                        d_off, accumulate)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_pixshuf_relu_fwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* in, void* out, int Cp_out,
                                    void* stream) {
  PXL_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && C > 0 && 4 * C <= Cp_in && C <= Cp_out, "pixshuf_relu_fwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * 4 * h * w * Cp_out;
  PXL_DISPATCH(dtype, (pixshuf_fwd_kernel<T><<<grid_for(n), 256, 0, s>>>(B, h, w, Cp_in, C, (const T*)in, (T*)out, Cp_out)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_pixshuf_relu_bwd(int dtype, int B, int h, int w, int Cp_in, int C, const void* dout, int Cp_out,
                                    const void* in, void* din, void* stream) {
  PXL_REQUIRE(dout && in && din && B > 0 && h > 0 && w > 0 && C > 0 && 4 * C <= Cp_in && C <= Cp_out,
              "pixshuf_relu_bwd: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)B * h * w * Cp_in;
  PXL_DISPATCH(dtype, (pixshuf_bwd_kernel<T><<<grid_for(n), 256, 0, s>>>(B, h, w, Cp_in, C, (const T*)dout, Cp_out, (const T*)in,
                                                                         (T*)din)));
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
