// Multi-rate head (DeepLab-v2 ASPP classifier, deeplab_v2.py:76-85: out = sum_g conv3x3(x; dilation d_g) + bias_g, 2048 -> 21
// channels, d = 6 / 12 / 18 / 24) as ONE dense GEMM plus two data-movement kernels.
//
// As a convolution the head is a terrible contraction: 21 output channels fill a third of a 64-wide tile, half of the 36 taps
// of a 33 x 33 map fall into the zero padding, and a tap-per-K-step kernel re-reads the 35.7 MB input once per tap (round 5/6:
// 117-141 us forward, 98-139 us data gradient, 75-113 us weight gradient at 8 x 33 x 33).  But
//
//     out[b, y, x, c] = sum_t  < X[b, y + dy_t, x + dx_t, :], W[t, c, :] >  =  sum_t  P[(b, y + dy_t, x + dx_t)][t, c]
//     with  P = X . Wp^T,   Wp[(t, c)][:] = W[t, c, :]                       (M x 756 = a plain GEMM over K = 2048)
//
// so the forward pass is a 1x1 "convolution" with 756 output channels on the LDS-DMA kernel -- X read ONCE, no padding work, no
// idle tile columns -- followed by a gather-sum of P (col2im).  The column order is j = g * GP + t_local * Cout + c with GP = 192
// (189 rounded to the 64-channel granule of the DMA kernels): TAP-major, so the Cout values one (pixel, tap) pair contributes
// are contiguous (class-major -- the master weight layout [Cout][kh][kw][Cin] -- made col2im read one float per 64-byte line:
// 24 us for 27 MB); pxl_aspp_pack writes Wp and its transpose (the data-gradient operand) in that order, pxl_aspp_dw_scatter
// maps the GEMM's weight gradient back to the master rows.  Backward: dP[(b, y', x')][t, c] = dOut[b, y' - dy_t, x' - dx_t, c] (zero outside)
// is gathered once (13 MB), dX = dP . Wp and dWp = dP^T . X are plain GEMMs again.  P stays fp32 (the bf16 engine's GEMM leaves
// fp32 partial-sum slabs, pxl_conv_dma_slabs), so the 36-term sum rounds once like the convolution did.
#include "common.h"

namespace {
struct AsppGeo {
  int B, H, W;
  int J, GP, ngroups, cout, tpg;     // columns of P (pitch), columns per group, groups, classes, taps per group
  int dy[64], dx[64];                // per global tap t = g * tpg + t_local
};

// out[m][c] (c < cout; padded channels zero) = bias[c] + sum over taps and slabs of P; one thread per (pixel, class)
template <typename T>
__global__ __launch_bounds__(256) void aspp_col2im_kernel(const AsppGeo g, const float* __restrict__ P, int nslab, size_t slab,
                                                          const float* __restrict__ bias, T* __restrict__ out, int Cp) {
  // tap offsets in LDS: indexed from the argument block they cost a scalar load + wait per tap and thread
  __shared__ int s_dy[64], s_dx[64];
  if (threadIdx.x < 64) { s_dy[threadIdx.x] = g.dy[threadIdx.x]; s_dx[threadIdx.x] = g.dx[threadIdx.x]; }
  __syncthreads();
  const long total = (long)g.B * g.H * g.W * Cp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int c = (int)(i % Cp);
    const long m = i / Cp;
    float v = 0.f;
    if (c < g.cout) {
      const int x = (int)(m % g.W), y = (int)((m / g.W) % g.H);
      const long b = m / ((long)g.W * g.H);
      const size_t img = (size_t)b * g.H * g.W;
      v = bias != nullptr ? bias[c] : 0.f;
      for (int grp = 0; grp < g.ngroups; ++grp) {
        const float* Pg = P + grp * g.GP + c;
        // (independent, predicated loads: a `continue` per padding tap left one dependent load chain per thread)
#pragma unroll 3
        for (int tl = 0; tl < g.tpg; ++tl) {
          const int t = grp * g.tpg + tl;
          const int yy = y + s_dy[t], xx = x + s_dx[t];
          const bool in = (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;
          const size_t o = in ? (img + (size_t)yy * g.W + xx) * g.J + tl * g.cout : 0;
          float sv = Pg[o];
          for (int k = 1; k < nslab; ++k) sv += Pg[o + k * slab];      // (slabs in index order: the same bits on every run)
          v += in ? sv : 0.f;
        }
      }
    }
    out[i] = from_f<T>(v);
  }
}

// dP[m'][j] for the 8 columns of one 16-byte chunk; j = grp * GP + tl * cout + c -> dOut[(y' - dy_t, x' - dx_t)][c], zero outside / padding
template <typename T>
__global__ __launch_bounds__(256) void aspp_dp_gather_kernel(const AsppGeo g, const T* __restrict__ dout, int Cp, T* __restrict__ dP) {
  constexpr int EPC = Elem<T>::EPC;
  const int chunks = g.J / EPC;
  const long total = (long)g.B * g.H * g.W * chunks;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int ch = (int)(i % chunks);
    const long m = i / chunks;
    const int x = (int)(m % g.W), y = (int)((m / g.W) % g.H);
    const long b = m / ((long)g.W * g.H);
    float v[EPC];
    int j0 = ch * EPC;
    int grp = j0 / g.GP, r = j0 - grp * g.GP;
    int tl = r / g.cout, c = r - tl * g.cout;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float f = 0.f;
      if (tl < g.tpg) {                                   // (tl >= tpg: the padding columns of the group)
        const int t = grp * g.tpg + tl;
        const int yy = y - g.dy[t], xx = x - g.dx[t];
        if ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W)
          f = to_f(dout[((size_t)(b * g.H + yy) * g.W + xx) * Cp + c]);
      }
      v[e] = f;
      if (++r == g.GP) { r = 0; ++grp; tl = 0; c = 0; }
      else if (++c == g.cout) { c = 0; ++tl; }
    }
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<uint4*>(dP + (size_t)m * g.J + ch * EPC) = Chunk<bf16_t>::pack(v);
    } else {
      *reinterpret_cast<float4*>(dP + (size_t)m * g.J + ch * EPC) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

struct AsppOffs { long w_off[4]; };
// master gradient row (c * tpg + tl) of group grp += row grp * GP + tl * cout + c of the GEMM's weight gradient [J][Cpin] (fp32);
// one thread per 4 consecutive input channels
__global__ __launch_bounds__(256) void aspp_dw_scatter_kernel(const float* __restrict__ tmp, int ngroups, int GP, int cout, int tpg,
                                                              int Cin, int Cpin, float* __restrict__ grads, const AsppOffs o) {
  const int q4 = Cin / 4;
  const long per = (long)cout * tpg * q4;
  const long total = per * ngroups;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int grp = (int)(i / per);
    const long q = i - grp * per;
    const int r = (int)(q / q4), k = (int)(q - (long)r * q4) * 4;          // r = master row c * tpg + tl
    const int c = r / tpg, tl = r - c * tpg;
    const float4 v = *reinterpret_cast<const float4*>(tmp + ((size_t)grp * GP + tl * cout + c) * Cpin + k);
    float* dst = grads + o.w_off[grp] + (long)r * Cin + k;                // (parameter offsets are not 16-byte aligned in general)
    dst[0] += v.x; dst[1] += v.y; dst[2] += v.z; dst[3] += v.w;
  }
}

// Wp[(grp * GP + tl * cout + c)][k] = T(w_grp[c][tl][k]) and Wd[k][the same column] (either may be NULL); padding rows / columns
// are written as zeros.  One block per 32 x 32 (column j, channel k) tile: read with k fastest, the transpose through LDS
template <typename T>
__global__ __launch_bounds__(256) void aspp_pack_kernel(const float* __restrict__ params, const AsppOffs o, int ngroups, int GP, int cout,
                                                        int tpg, int Cin, int Cp, T* __restrict__ Wp, T* __restrict__ Wd) {
  __shared__ float tile[32][33];
  const int J = ngroups * GP;
  const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int rr = ty; rr < 32; rr += 8) {
    const int j = j0 + rr, k = k0 + tx;
    const int grp = j / GP, r = j - grp * GP;
    float v = 0.f;
    if (j < J && r < cout * tpg && k < Cin) {
      const int tl = r / cout, c = r - tl * cout;
      v = params[o.w_off[grp] + ((long)c * tpg + tl) * Cin + k];
    }
    tile[rr][tx] = v;
    if (Wp != nullptr && j < J && k < Cp) Wp[(size_t)j * Cp + k] = from_f<T>(v);
  }
  __syncthreads();
  if (Wd != nullptr)
    for (int rr = ty; rr < 32; rr += 8) {
      const int k = k0 + rr, j = j0 + tx;
      if (k < Cp && j < J) Wd[(size_t)k * J + j] = from_f<T>(tile[tx][rr]);
    }
}

int fill_geo(AsppGeo& g, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy, const int16_t* dx) {
  if (ngroups < 1 || ngroups > 4 || ngroups * tpg > 64 || cout * tpg > GP || ngroups * GP > J) return 1;
  g.B = B; g.H = H; g.W = W; g.J = J; g.GP = GP; g.ngroups = ngroups; g.cout = cout; g.tpg = tpg;
  for (int t = 0; t < 64; ++t) { g.dy[t] = t < ngroups * tpg ? dy[t] : 0; g.dx[t] = t < ngroups * tpg ? dx[t] : 0; }
  return 0;
}
inline int grid_of(long total) { long b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }
}  // namespace

extern "C" int pxl_aspp_col2im(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                               const int16_t* dx, const float* P, int nslab, size_t slab_floats, const float* bias, void* out,
                               int Cp, void* stream) {
  AsppGeo g;
  PXL_REQUIRE(P && out && dy && dx && nslab >= 1 && Cp >= cout && fill_geo(g, B, H, W, J, GP, ngroups, cout, tpg, dy, dx) == 0,
              "aspp_col2im: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * H * W * Cp;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(aspp_col2im_kernel<float>, dim3(grid_of(total)), dim3(256), 0, s, g, P, nslab, slab_floats, bias, (float*)out, Cp);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(aspp_col2im_kernel<bf16_t>, dim3(grid_of(total)), dim3(256), 0, s, g, P, nslab, slab_floats, bias, (bf16_t*)out, Cp);
  else
    return pxl_set_error(PXL_ERR_ARG, "aspp_col2im: bad dtype %d", dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_aspp_dp_gather(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                                  const int16_t* dx, const void* dout, int Cp, void* dP, void* stream) {
  AsppGeo g;
  PXL_REQUIRE(dout && dP && dy && dx && Cp >= cout && J % 8 == 0 && fill_geo(g, B, H, W, J, GP, ngroups, cout, tpg, dy, dx) == 0,
              "aspp_dp_gather: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32) {
    const long total = (long)B * H * W * (J / 4);
    hipLaunchKernelGGL(aspp_dp_gather_kernel<float>, dim3(grid_of(total)), dim3(256), 0, s, g, (const float*)dout, Cp, (float*)dP);
  } else if (dtype == PXL_BF16) {
    const long total = (long)B * H * W * (J / 8);
    hipLaunchKernelGGL(aspp_dp_gather_kernel<bf16_t>, dim3(grid_of(total)), dim3(256), 0, s, g, (const bf16_t*)dout, Cp, (bf16_t*)dP);
  } else {
    return pxl_set_error(PXL_ERR_ARG, "aspp_dp_gather: bad dtype %d", dtype);
  }
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_aspp_dw_scatter(const float* tmp, int ngroups, int GP, int cout, int tpg, int Cin, int Cpin, float* grads,
                                   const long* w_off, void* stream) {
  PXL_REQUIRE(tmp && grads && w_off && ngroups >= 1 && ngroups <= 4 && cout * tpg <= GP && Cin <= Cpin && Cin % 4 == 0 && Cpin % 4 == 0,
              "aspp_dw_scatter: bad argument");
  AsppOffs o;
  for (int g = 0; g < 4; ++g) o.w_off[g] = g < ngroups ? w_off[g] : 0;
  const long total = (long)cout * tpg * (Cin / 4) * ngroups;
  hipLaunchKernelGGL(aspp_dw_scatter_kernel, dim3(grid_of(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tmp, ngroups, GP,
                     cout, tpg, Cin, Cpin, grads, o);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_aspp_pack(int dtype, const float* params, const long* w_off, int ngroups, int GP, int cout, int tpg, int Cin, int Cp,
                             void* Wp, void* Wd, void* stream) {
  PXL_REQUIRE(params && w_off && (Wp || Wd) && ngroups >= 1 && ngroups <= 4 && cout * tpg <= GP && Cin <= Cp, "aspp_pack: bad argument");
  AsppOffs o;
  for (int g = 0; g < 4; ++g) o.w_off[g] = g < ngroups ? w_off[g] : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((Cp + 31) / 32, (ngroups * GP + 31) / 32);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(aspp_pack_kernel<float>, grid, dim3(256), 0, s, params, o, ngroups, GP, cout, tpg, Cin, Cp, (float*)Wp, (float*)Wd);
  else if (dtype == PXL_BF16)
    hipLaunchKernelGGL(aspp_pack_kernel<bf16_t>, grid, dim3(256), 0, s, params, o, ngroups, GP, cout, tpg, Cin, Cp, (bf16_t*)Wp, (bf16_t*)Wd);
  else
    return pxl_set_error(PXL_ERR_ARG, "aspp_pack: bad dtype %d", dtype);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
