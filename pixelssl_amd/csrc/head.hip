// Segmentation head tail: bilinear up-sampling (align_corners=True, or False for SSLCCT's auxiliary decoders) of the low-resolution logits
// to the input size, fused with the channel soft-max (deeplab_v2.py:32, task/sseg/model.py:62),
// and its backward (adjoint of the interpolation, with the soft-max Jacobian folded in).
//
// Layouts: low-res logits NHWC [B][h][w][Cp] (Cp = padded channel pitch, real C <= 32) in the
// engine dtype; full-res outputs NCHW fp32 [B][C][H][W] (what the plugin API hands to the SSL
// algorithms).  HBM-bound: the forward writes 2*C*H*W floats per image; consecutive lanes walk x
// so every channel-plane store is a coalesced 256-byte line.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int MAXC = 32;

// PyTorch area_pixel_compute_source_index: align_corners=True: src = scale * dst with scale = (in-1)/(out-1);
// align_corners=False (the F.interpolate default, used on SSLCCT's auxiliary predictions): src = max(0,
// scale * (dst + 0.5) - 0.5) with scale = in/out.  `align` selects the formula.
__device__ __forceinline__ void src_coord(int o, float scale, int align, int in_size, int& i0, int& i1, float& l1) {
  const float src = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// The source index i0 of src_coord is monotone in the destination index: start(v) = the first destination index whose i0 is >= v
// (start(in_size) = out_size), found by bisection.  The destinations that touch source index v are then two contiguous runs --
// [start(v-1), start(v)) through their i1 (weight l1) and [start(v), start(v+1)) through their i0 (weight 1 - l1; at the clamped last
// index i1 == i0, so the weight is (1 - l1) + l1) -- which is what the adjoint passes below walk instead of testing every candidate
// of a padded interval.
__device__ __forceinline__ int first_dst_ge(int v, float scale, int align, int in_size, int out_size) {
  if (v <= 0) return 0;
  if (v >= in_size) return out_size;
  int lo = 0, hi = out_size;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    int a0, a1; float l;
    src_coord(mid, scale, align, in_size, a0, a1, l);
    if (a0 >= v) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// sum over the destinations x that touch source index x0 of weight(x) * g[x * gs], in ascending x; cs = the start() table in LDS,
// l1 = the interpolation weights of the destinations (LDS).  Loads of four terms are issued together.
__device__ __forceinline__ float adjoint_run_lds(const float* __restrict__ g, int gs, const float* __restrict__ l1, const int* __restrict__ cs,
                                                 int x0, int in_size) {
  const int a = cs[x0 > 0 ? x0 - 1 : 0], m = cs[x0], e = cs[x0 + 1];
  const bool last = x0 == in_size - 1;
  float acc = 0.f;
  for (int x = (x0 > 0 ? a : m); x < e; x += 4) {
    float lv[4], gv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x + k < e ? x + k : e - 1;
      lv[k] = l1[xx];
      gv[k] = g[(size_t)xx * gs];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (x + k < e) {
        float wgt = 0.f;
        if (x + k >= m) { wgt += 1.f - lv[k]; if (last) wgt += lv[k]; } else wgt += lv[k];
        acc += wgt * gv[k];
      }
    }
  }
  return acc;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_softmax_fwd_kernel(int B, int h, int w, int Cp, int C, int H,
                                                                   int W, float sy, float sx, int align,
                                                                   const T* __restrict__ low,
                                                                   float* __restrict__ logits,
                                                                   float* __restrict__ prob) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  if (x >= W) return;
  int y0, y1, x0, x1;
  float ly, lx;
  src_coord(y, sy, align, h, y0, y1, ly);
  src_coord(x, sx, align, w, x0, x1, lx);
  const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
  const T* p00 = low + ((size_t)(b * h + y0) * w + x0) * Cp;
  const T* p01 = low + ((size_t)(b * h + y0) * w + x1) * Cp;
  const T* p10 = low + ((size_t)(b * h + y1) * w + x0) * Cp;
  const T* p11 = low + ((size_t)(b * h + y1) * w + x1) * Cp;
  // the four corner vectors as 16-byte chunks (the pitch is a multiple of the chunk): 4 x ceil(C / EPC) loads per pixel instead
  // of 4 x C element loads -- at the 264 -> 513 resize of CCT's decoders neighbouring pixels share almost no corner, and the
  // element loads made this kernel 4 x slower than its writes
  constexpr int EPC = Elem<T>::EPC;
  float v[MAXC];
  float mx = -INFINITY;
#pragma unroll
  for (int q = 0; q < MAXC / EPC; ++q) {
    if (q * EPC < C) {
      float a[EPC], bq[EPC], cq[EPC], d[EPC];
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p00 + q * EPC), a);
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p01 + q * EPC), bq);
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p10 + q * EPC), cq);
      Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p11 + q * EPC), d);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = q * EPC + e;
        v[c] = w00 * a[e] + w01 * bq[e] + w10 * cq[e] + w11 * d[e];
        if (c < C) mx = fmaxf(mx, v[c]);
      }
    }
  }
  const size_t plane = (size_t)H * W;
  const size_t o = (size_t)b * C * plane + (size_t)y * W + x;
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (c < C) {
      logits[o + c * plane] = v[c];
      v[c] = __expf(v[c] - mx);
      sum += v[c];
    }
  }
  if (prob != nullptr) {
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) prob[o + c * plane] = v[c] * inv;
  }
}

// pass 1 (one block per full-res row): G = dlogits + softmax_bwd(dprob, prob) ; reduce along x onto
// the w low-res columns:  tmp[b][y][x0][c] = sum_x wx(x,x0) * G[b][c][y][x]
__global__ __launch_bounds__(320) void upsample_bwd_rows_kernel(int C, int H, int W, int w, float sx, int align,
                                                                const float* __restrict__ dlogits,
                                                                const float* __restrict__ dprob,
                                                                const float* __restrict__ prob,
                                                                float* __restrict__ tmp) {
  extern __shared__ float g[];   // [C][W+1], then l1 per full-res column [W] and the start() table [w + 1]
  const int y = blockIdx.x, b = blockIdx.y;
  const size_t plane = (size_t)H * W;
  const size_t base = (size_t)b * C * plane + (size_t)y * W;
  const int ld = W + 1;
  float* cl1 = g + (size_t)C * ld;
  int* cs = reinterpret_cast<int*>(cl1 + W);
  for (int v = threadIdx.x; v <= w; v += blockDim.x) cs[v] = first_dst_ge(v, sx, align, w, W);
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int i0, i1;
    float l1;
    src_coord(x, sx, align, w, i0, i1, l1);      // once per column (the reduction below used to redo it 2 * C times)
    cl1[x] = l1;
    // every channel plane load of this column is issued before the first use (a loop over the run-time C kept one HBM
    // round trip per channel in flight: 179 us for 177 MB)
    float dl[MAXC], dp[MAXC], pr[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      dl[c] = (c < C && dlogits != nullptr) ? dlogits[base + c * plane + x] : 0.f;
      if (dprob != nullptr) {
        dp[c] = c < C ? dprob[base + c * plane + x] : 0.f;
        pr[c] = c < C ? prob[base + c * plane + x] : 0.f;
      }
    }
    float dot = 0.f;
    if (dprob != nullptr) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) dot += dp[c] * pr[c];
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        float v = dl[c];
        if (dprob != nullptr) v += pr[c] * (dp[c] - dot);
        g[c * ld + x] = v;
      }
  }
  __syncthreads();
  // each output (x0, c): the full-res x whose i0 is x0 contribute (1 - l), those whose i0 is x0 - 1 contribute l
  for (int o = threadIdx.x; o < w * C; o += blockDim.x) {
    const int c = o % C, x0 = o / C;
    tmp[(((size_t)b * H + y) * w + x0) * C + c] = adjoint_run_lds(g + c * ld, 1, cl1, cs, x0, w);
  }
}

// pass 2: dlow[b][y0][x0][c] = sum_y wy(y,y0) * tmp[b][y][x0][c]   (written in the engine dtype, padded
// channels zeroed)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_cols_kernel(int B, int h, int w, int Cp, int C, int H,
                                                                float sy, int align, const float* __restrict__ tmp,
                                                                T* __restrict__ dlow, const float* __restrict__ scale_dev) {
  extern __shared__ float cols_lds[];      // l1 per full-res row [H], then the start() table [h + 1]
  float* rl1 = cols_lds;
  int* rs = reinterpret_cast<int*>(rl1 + H);
  for (int v = threadIdx.x; v <= h; v += blockDim.x) rs[v] = first_dst_ge(v, sy, align, h, H);
  for (int y = threadIdx.x; y < H; y += blockDim.x) {
    int i0, i1;
    float l1;
    src_coord(y, sy, align, h, i0, i1, l1);
    rl1[y] = l1;
  }
  __syncthreads();
  const float scale = scale_dev != nullptr ? scale_dev[0] : 1.f;     // (pxl_cons_head_bwd: the incoming gradient of a scalar loss)
  const long total = (long)B * h * w * Cp;
  const size_t ystride = (size_t)w * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    long r = i / Cp;
    const int x0 = (int)(r % w); r /= w;
    const int y0 = (int)(r % h);
    const int b = (int)(r / h);
    float acc = 0.f;
    if (c < C) {
      // the rows that touch y0: [start(y0 - 1), start(y0)) through i1, [start(y0), start(y0 + 1)) through i0; eight loads in flight
      // (four outputs per thread and trip on top of that were measured: no faster)
      const int m = rs[y0], e = rs[y0 + 1];
      const bool last = y0 == h - 1;
      const float* col = tmp + ((size_t)b * H * w + x0) * C + c;
      for (int y = (y0 > 0 ? rs[y0 - 1] : m); y < e; y += 8) {
        float tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tv[k] = col[(size_t)(y + k < e ? y + k : e - 1) * ystride];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (y + k < e) {
            const float l1 = rl1[y + k];
            float wgt = 0.f;
            if (y + k >= m) { wgt += 1.f - l1; if (last) wgt += l1; } else wgt += l1;
            if (wgt != 0.f) acc += wgt * tv[k];
          }
        }
      }
    }
    dlow[i] = from_f<T>(scale_dev != nullptr ? acc * scale : acc);
  }
}


// ---- training seam of the segmentation step without the full-resolution planes -------------------------------------
// The reference materialises logits = upsample(low) [B][C][H][W] (deeplab_v2.py:32), softmax(logits)
// (task/sseg/model.py:62), then the criterion (task/sseg/criterion.py:24-38), the consistency term (ssl_mt.py:179-184)
// and autograd's backward read / write those planes again: at 8 x 21 x 513 x 513 that is 177 MB per plane and seven
// launches between the last forward convolution and the first data gradient.  When no plugin reads the planes, everything
// between the low-resolution logits and their gradient is a function of (student low-res logits, teacher low-res logits,
// labels): this kernel evaluates it per full-resolution row in registers / LDS and never writes a plane.
//   per pixel:  z_s = bilinear(s_low), z_t = bilinear(t_low)                           (same expression as the forward kernel)
//               n <  n_ce          : CE(z_s, label), CE(z_t, label) with ignore_index, summed / (H*W) per sample
//               mse_lo <= n < mse_hi: sum (z_s - z_t)^2
//               G = ce_scale * (softmax(z_s) - onehot) [valid label]  +  mse_scale * (z_s - z_t)
//   then the x-reduction of upsample_bwd_rows_kernel on G -> tmp[b][y][x0][c]; upsample_bwd_cols_kernel finishes d(low).
// One block per (row y, sample b).  LDS: G [C][W+1], the column tables, and the two low-res rows of both networks.
template <typename T>
__global__ __launch_bounds__(256) void head_loss_rows_kernel(int C, int Cp, int h, int w, int H, int W, float sy, float sx,
                                                             int align, const T* __restrict__ s_low,
                                                             const T* __restrict__ t_low, const float* __restrict__ gt,
                                                             int ignore_index, int n_ce, int mse_lo, int mse_hi,
                                                             float ce_scale, float mse_scale, float inv_hw, float inv_mse_n,
                                                             float* __restrict__ tmp, float* __restrict__ sums, int B,
                                                             const float* __restrict__ mse_w_dev, float* __restrict__ rowpart) {
  if (mse_w_dev != nullptr) mse_scale *= mse_w_dev[0];      // (consistency weight in device memory: pxl_head_loss_hp)
  extern __shared__ float g[];   // [C][W+1] | the start() table [w + 1 <= 2 W] | cl1[W] | srow[2][w][C] | trow[2][w][C]
  __shared__ float red[4];
  const int y = blockIdx.x, b = blockIdx.y;
  const int ld = W + 1;
  int* cs = reinterpret_cast<int*>(g + (size_t)C * ld);
  float* cl1 = reinterpret_cast<float*>(cs + 2 * W);
  float* srow = cl1 + W;
  for (int v = threadIdx.x; v <= w; v += blockDim.x) cs[v] = first_dst_ge(v, sx, align, w, W);
  float* trow = srow + 2 * w * C;
  int y0, y1;
  float ly;
  src_coord(y, sy, align, h, y0, y1, ly);
  const bool has_t = t_low != nullptr;
  for (int i = threadIdx.x; i < 2 * w * C; i += blockDim.x) {
    const int c = i % C, xx = (i / C) % w, r = i / (C * w);
    const size_t o = ((size_t)(b * h + (r ? y1 : y0)) * w + xx) * Cp + c;
    srow[i] = to_f(s_low[o]);
    if (has_t) trow[i] = to_f(t_low[o]);
  }
  __syncthreads();
  const bool ce = b < n_ce;
  const bool mse = has_t && b >= mse_lo && b < mse_hi;
  float acc_s = 0.f, acc_t = 0.f, acc_m = 0.f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int x0, x1;
    float lx;
    src_coord(x, sx, align, w, x0, x1, lx);
    cl1[x] = lx;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const float* s00 = srow + x0 * C; const float* s01 = srow + x1 * C;
    const float* s10 = srow + (w + x0) * C; const float* s11 = srow + (w + x1) * C;
    const float* t00 = trow + x0 * C; const float* t01 = trow + x1 * C;
    const float* t10 = trow + (w + x0) * C; const float* t11 = trow + (w + x1) * C;
    int label = -1;
    bool valid = false;
    if (ce) {
      label = (int)gt[((size_t)b * H + y) * W + x];
      valid = !(label == ignore_index || label < 0 || label >= C);
    }
    float zs[MAXC], zt[MAXC];
    float mxs = -INFINITY, mxt = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < C) {
        zs[c] = w00 * s00[c] + w01 * s01[c] + w10 * s10[c] + w11 * s11[c];
        mxs = fmaxf(mxs, zs[c]);
        if (has_t) {
          zt[c] = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
          mxt = fmaxf(mxt, zt[c]);
        }
      }
    }
    float es[MAXC];
    float inv = 0.f;
    if (valid) {
      float sum = 0.f, pick = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) { es[c] = expf(zs[c] - mxs); sum += es[c]; if (c == label) pick = zs[c]; }
      acc_s += (mxs + logf(sum)) - pick;
      inv = ce_scale / sum;
      if (has_t) {
        float tsum = 0.f, tpick = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < C) { tsum += expf(zt[c] - mxt); if (c == label) tpick = zt[c]; }
        acc_t += (mxt + logf(tsum)) - tpick;
      }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        float gv = valid ? __fmaf_rn(es[c], inv, c == label ? -ce_scale : 0.f) : 0.f;
        if (mse) {
          const float d = zs[c] - zt[c];
          acc_m += d * d;
          const float m = __fmul_rn(mse_scale, d);
          gv = ce ? __fadd_rn(gv, m) : m;
        }
        g[c * ld + x] = gv;
      }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < w * C; o += blockDim.x) {
    const int c = o % C, x0 = o / C;
    tmp[(((size_t)b * H + y) * w + x0) * C + c] = adjoint_run_lds(g + c * ld, 1, cl1, cs, x0, w);
  }
  // loss sums: one atomic per block and quantity (as ce_fwd_kernel / mse_fwd_kernel do) -- or, rowpart != NULL, one plain
  // store per block into rowpart[b][y][3], summed in row order by head_loss_rows_finish_kernel (bit-reproducible values)
  float v = wave_sum(acc_s);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const part = rowpart != nullptr ? rowpart + ((size_t)b * H + y) * 3 : nullptr;
  __syncthreads();
  if (part != nullptr && threadIdx.x < 3) part[threadIdx.x] = 0.f;
  __syncthreads();
  if (ce) {
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = (red[0] + red[1] + red[2] + red[3]) * inv_hw;
      if (part != nullptr) part[0] = t; else atomicAdd(sums + b, t);
    }
    __syncthreads();
    if (has_t) {
      v = wave_sum(acc_t);
      if (lane == 0) red[wave] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        const float t = (red[0] + red[1] + red[2] + red[3]) * inv_hw;
        if (part != nullptr) part[1] = t; else atomicAdd(sums + B + b, t);
      }
      __syncthreads();
    }
  }
  if (mse) {
    v = wave_sum(acc_m);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float t = (red[0] + red[1] + red[2] + red[3]) * inv_mse_n;
      if (part != nullptr) part[2] = t; else atomicAdd(sums + 2 * B, t);
    }
  }
}

// sums[b] = sum_y rowpart[b][y][0], sums[B + b] = sum_y rowpart[b][y][1], sums[2B] = sum_b sum_y rowpart[b][y][2], every sum in
// index order by ONE thread (B * H <= a few thousand terms: microseconds; the point is the fixed order)
__global__ void head_loss_rows_finish_kernel(int B, int H, const float* __restrict__ rowpart, float* __restrict__ sums) {
  const int t = threadIdx.x;
  if (t < 2 * B) {
    const int b = t % B, q = t / B;
    float v = 0.f;
    for (int y = 0; y < H; ++y) v += rowpart[((size_t)b * H + y) * 3 + q];
    sums[q * B + b] = v;
  } else if (t == 2 * B) {
    float v = 0.f;
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y) v += rowpart[((size_t)b * H + y) * 3 + 2];
    sums[2 * B] = v;
  }
}

// ---- the same seam, cell-wise (large up-sampling factors: DeepLab's 33 x 33 -> 513 x 513) ----------------------------------
// One WAVE per low-resolution cell (b, i, j) = the full-resolution pixels whose bilinear footprint is the four corners
// (i, j) .. (i+1, j+1): 16 x 16 pixels at scale 16.  The corner values of both networks sit in LDS (broadcast reads); a
// lane owns ONE pixel column of the cell (its x weight is a constant) and walks the rows, everything of a pixel stays in
// registers; it accumulates A0 = sum (1 - ly) * G and A1 = sum ly * G (2 * C registers), the x weights and the sum over
// the lanes are applied once per cell by a butterfly, and 4 * C fp32 atomics per cell go into dacc [B][h][w][C] (every
// element receives at most four addends).  The row-wise kernel above spends its time in two barriers and an LDS image
// of the row per 513 pixels; this one has no barrier in the pixel loop and ~8x the waves in flight.  Loss partial sums go
// to part[cell][3] and are reduced in a fixed order by head_loss_finish_kernel (no contended atomics, deterministic
// losses).  C is a template parameter: with a run-time channel count the accumulators spill.
// fp32 engine (the parity mode): library expf / logf; bf16 engine: v_exp_f32 / v_log_f32 (~1e-6 relative)
template <typename T> __device__ __forceinline__ float head_exp(float x) {
  if constexpr (sizeof(T) == 4) return expf(x); else return __expf(x);
}
template <typename T> __device__ __forceinline__ float head_log(float x) {
  if constexpr (sizeof(T) == 4) return logf(x); else return __logf(x);
}

template <typename T, int C>
__global__ __launch_bounds__(256) void head_loss_cells_kernel(int Cp, int h, int w, int H, int W, float sy, float sx,
                                                              int align, const T* __restrict__ s_low,
                                                              const T* __restrict__ t_low, const float* __restrict__ gt,
                                                              int ignore_index, int n_ce, int mse_lo, int mse_hi,
                                                              float ce_scale, float mse_scale, float* __restrict__ dacc,
                                                              float* __restrict__ part, int B, const float* __restrict__ mse_w_dev) {
  if (mse_w_dev != nullptr) mse_scale *= mse_w_dev[0];      // (consistency weight in device memory: pxl_head_loss_hp)
  __shared__ float corner[4][2][4 * C];      // [wave][student | teacher][corner k = 2 * dy + dx][c]
  __shared__ float tr[4][(2 * C + 1) * 65];    // [wave][A0 rows | A1 rows | lx][lane], pitch 65
  __shared__ float lsum[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long cell = (long)blockIdx.x * 4 + wave;
  const long ncell = (long)B * h * w;
  float acc_s = 0.f, acc_t = 0.f, acc_m = 0.f;
  int b = 0, i = 0, j = 0, i1 = 0, j1 = 0;
  bool live = cell < ncell;
  int ya = 0, yb = 0, xa = 0, xb = 0;
  if (live) {
    j = (int)(cell % w);
    i = (int)((cell / w) % h);
    b = (int)(cell / ((long)w * h));
    i1 = i + (i < h - 1 ? 1 : 0);
    j1 = j + (j < w - 1 ? 1 : 0);
    // pixel range of the cell: floor(src) is monotone in the destination index
    auto first_ge = [&](int target, float scale, int in_size, int out_size) {
      int lo = 0, hi = out_size;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        int a0, a1; float l;
        src_coord(mid, scale, align, in_size, a0, a1, l);
        if (a0 >= target) hi = mid; else lo = mid + 1;
      }
      return lo;
    };
    ya = first_ge(i, sy, h, H); yb = first_ge(i + 1, sy, h, H);
    xa = first_ge(j, sx, w, W); xb = first_ge(j + 1, sx, w, W);
    live = yb > ya && xb > xa;
  }
  const bool has_t = t_low != nullptr;
  if (live) {
    for (int e = lane; e < 4 * C; e += 64) {
      const int k = e / C, c = e - k * C;
      const size_t o = ((size_t)(b * h + ((k >> 1) ? i1 : i)) * w + ((k & 1) ? j1 : j)) * Cp + c;
      corner[wave][0][e] = to_f(s_low[o]);
      if (has_t) corner[wave][1][e] = to_f(t_low[o]);
    }
  }
  __syncthreads();
  float A0[C], A1[C];
#pragma unroll
  for (int c = 0; c < C; ++c) A0[c] = A1[c] = 0.f;
  float lx = 0.f;
  if (live) {
    const bool ce = b < n_ce;
    const bool mse = has_t && b >= mse_lo && b < mse_hi;
    const float ms = mse ? mse_scale : 0.f, msq = mse ? 1.f : 0.f;
    // lane -> (row group, column): columns padded to a power of two <= 64 (the host guarantees xb - xa <= 64)
    const int nx = xb - xa;
    int nxp = 1;
    while (nxp < nx) nxp <<= 1;
    const int xl = lane & (nxp - 1), rg = lane / nxp, nrg = 64 / nxp;
    const int x = xa + xl;
    const bool col = xl < nx;
    int a0, a1;
    src_coord(col ? x : xa, sx, align, w, a0, a1, lx);
    for (int y = ya + rg; y < yb && col; y += nrg) {
      // the corner values are re-read from LDS (broadcast) for every pixel: hoisted out of the loop they would cost
      // 8 * C registers -- the empty asm makes the offset opaque to that optimisation
      int coff = 0;
      asm volatile("" : "+v"(coff));
      const float* cs = corner[wave][0] + coff;
      const float* ct = corner[wave][1] + coff;
      float ly;
      src_coord(y, sy, align, h, a0, a1, ly);
      const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
      int label = -1;
      bool valid = false;
      if (ce) {
        label = (int)gt[((size_t)b * H + y) * W + x];
        valid = !(label == ignore_index || label < 0 || label >= C);
      }
      // straight-line per-pixel code with scalar switches (ce / mse are uniform over the wave): conditional in-place
      // updates of the channel arrays made the register allocator keep two copies of them
      float zs[C], gm[C];
      float mxs = -INFINITY, mxt = -INFINITY;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        zs[c] = w00 * cs[c] + w01 * cs[C + c] + w10 * cs[2 * C + c] + w11 * cs[3 * C + c];
        mxs = fmaxf(mxs, zs[c]);
      }
      float tsum = 0.f, tpick = 0.f;
      if (has_t) {                                   // teacher: max, then exp-sum + the consistency part of G in one sweep
        float zt[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
          zt[c] = w00 * ct[c] + w01 * ct[C + c] + w10 * ct[2 * C + c] + w11 * ct[3 * C + c];
          mxt = fmaxf(mxt, zt[c]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (ce) tsum += head_exp<T>(zt[c] - mxt);          // (ce is uniform over the wave: unlabeled samples skip the exps)
          if (c == label) tpick = zt[c];
          const float d = zs[c] - zt[c];
          acc_m += msq * (d * d);
          gm[c] = __fmul_rn(ms, d);
        }
        if (valid) acc_t += (mxt + head_log<T>(tsum)) - tpick;
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) gm[c] = 0.f;
      }
      // fast exp / log (v_exp_f32 / v_log_f32, ~1e-6 relative): the library versions are ~15 instructions each and were
      // two thirds of this kernel's time; the soft-max of the forward kernel uses the same intrinsic
      float sum = 1.f, pick = 0.f;
      if (ce) {
        sum = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { if (c == label) pick = zs[c]; zs[c] = head_exp<T>(zs[c] - mxs); sum += zs[c]; }
      }
      if (valid) acc_s += (mxs + head_log<T>(sum)) - pick;
      const float inv = valid ? ce_scale / sum : 0.f;
      const float hot = valid ? -ce_scale : 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float gce = __fmaf_rn(zs[c], inv, c == label ? hot : 0.f);      // 0 for an invalid / unlabeled pixel
        const float gv = ce ? (mse ? __fadd_rn(gce, gm[c]) : gce) : gm[c];
        A0[c] += (1.f - ly) * gv;
        A1[c] += ly * gv;
      }
    }
  }
  // x weights + sum over the lanes through LDS (a register butterfly over 4 * C values needs more registers than the
  // pixel loop): every lane parks its 2 * C sums and its x weight, then lane e < 4 * C forms corner value e =
  // sum_lanes (dx ? lx : 1 - lx) * A_dy[c] in lane order (deterministic).  Row pitch 65: conflict-free column walks.
  if (live) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      tr[wave][c * 65 + lane] = A0[c];
      tr[wave][(C + c) * 65 + lane] = A1[c];
    }
    tr[wave][2 * C * 65 + lane] = lx;
    acc_s = wave_sum(acc_s); acc_t = wave_sum(acc_t); acc_m = wave_sum(acc_m);
  }
  if (lane == 0) { lsum[wave][0] = acc_s; lsum[wave][1] = acc_t; lsum[wave][2] = acc_m; }
  __syncthreads();
  if (live) {
    for (int e = lane; e < 4 * C; e += 64) {
      const int k = e / C, c = e - k * C;
      const float* row = tr[wave] + ((k >> 1) * C + c) * 65;
      const float* lxs = tr[wave] + 2 * C * 65;
      float v = 0.f;
      for (int l = 0; l < 64; ++l) v += ((k & 1) ? lxs[l] : 1.f - lxs[l]) * row[l];
      const size_t o = ((size_t)(b * h + ((k >> 1) ? i1 : i)) * w + ((k & 1) ? j1 : j)) * C + c;
      atomicAdd(dacc + o, v);
    }
  }
  if (threadIdx.x < 12) {        // per-cell loss partials (an empty / out-of-range cell writes zeros)
    const int wv = threadIdx.x / 3, q = threadIdx.x - wv * 3;
    part[((size_t)blockIdx.x * 4 + wv) * 3 + q] = lsum[wv][q];
  }
}

// d(low) fp32 accumulator -> engine dtype NHWC with zeroed channel padding; blocks >= gridDim.x - B reduce the per-cell
// loss partials of one sample each, in cell order (deterministic): sums[b], sums[B + b] (x 1/HW); the first of them also
// sums[2B] = MSE mean over every sample
template <typename T>
__global__ __launch_bounds__(256) void head_loss_finish_kernel(int B, int h, int w, int C, int Cp, const float* __restrict__ dacc,
                                                               T* __restrict__ dlow, const float* __restrict__ part,
                                                               float inv_hw, float inv_mse_n, int n_ce, int has_t,
                                                               float* __restrict__ sums) {
  __shared__ float red[3][4];
  const int nconv = gridDim.x - B;
  if ((int)blockIdx.x < nconv) {
    const long total = (long)B * h * w * Cp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)nconv * blockDim.x) {
      const int c = (int)(i % Cp);
      const long px = i / Cp;
      dlow[i] = from_f<T>(c < C ? dacc[px * C + c] : 0.f);
    }
    return;
  }
  const int b = blockIdx.x - nconv;
  const long per = (long)h * w;
  float a[3] = {0.f, 0.f, 0.f};
  for (long k = threadIdx.x; k < per; k += blockDim.x) {      // fixed assignment of cells to threads: deterministic
    const float* p = part + ((long)b * per + k) * 3;
    a[0] += p[0]; a[1] += p[1]; a[2] += p[2];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float v = wave_sum(a[q]);
    if (lane == 0) red[q][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const float s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (b < n_ce) { sums[b] = s0 * inv_hw; if (has_t) sums[B + b] = s1 * inv_hw; }
  }
  // the MSE mean: ONE block adds every sample's cell partials, sample by sample with the same per-sample reduction as above
  // -> a fixed order, bit-identical from run to run (B contended atomicAdds used to decide the last bits)
  if (b == 0) {
    float total = 0.f;
    for (int q = 0; q < B; ++q) {
      float a2 = 0.f;
      for (long k = threadIdx.x; k < per; k += blockDim.x) a2 += part[((long)q * per + k) * 3 + 2];
      const float v = wave_sum(a2);
      __syncthreads();
      if (lane == 0) red[2][wave] = v;
      __syncthreads();
      total += red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
    if (threadIdx.x == 0) sums[2 * B] = total * inv_mse_n;
  }
}

}  // namespace

extern "C" int pxl_upsample_softmax_fwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners,
                                        const void* low, float* logits, float* prob, void* stream) {
  PXL_REQUIRE(low && logits, "upsample_softmax_fwd: null argument");
  PXL_REQUIRE(C >= 1 && C <= MAXC && C <= Cp, "upsample_softmax_fwd: C=%d unsupported (max %d)", C, MAXC);
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "upsample_softmax_fwd: bad dtype");
  {
    const int epc = dtype == PXL_F32 ? 4 : 8;             // the corner vectors are read as 16-byte chunks
    PXL_REQUIRE(Cp % epc == 0 && (C + epc - 1) / epc * epc <= Cp, "upsample_softmax_fwd: pitch %d does not hold whole 16-byte chunks of %d channels", Cp, C);
  }
  const int align = align_corners ? 1 : 0;
  const float sy = align ? (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f) : (float)h / (float)H;
  const float sx = align ? (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f) : (float)w / (float)W;
  dim3 grid(cdiv(W, 256), H, B);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(upsample_softmax_fwd_kernel<float>, grid, dim3(256), 0, s, B, h, w, Cp, C, H, W, sy, sx, align,
                       (const float*)low, logits, prob);
  else
    hipLaunchKernelGGL(upsample_softmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, B, h, w, Cp, C, H, W, sy, sx, align,
                       (const bf16_t*)low, logits, prob);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" size_t pxl_upsample_bwd_workspace(int B, int w, int C, int H) {
  return (size_t)B * H * w * C * sizeof(float);
}

extern "C" int pxl_upsample_softmax_bwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners,
                                        const float* dlogits, const float* dprob, const float* prob,
                                        void* dlow, void* workspace, size_t ws_bytes, void* stream) {
  PXL_REQUIRE(dlow && workspace, "upsample_softmax_bwd: null argument");
  PXL_REQUIRE(dlogits || dprob, "upsample_softmax_bwd: no incoming gradient");
  PXL_REQUIRE(dprob == nullptr || prob != nullptr, "upsample_softmax_bwd: dprob needs prob");
  PXL_REQUIRE(C >= 1 && C <= MAXC && C <= Cp, "upsample_softmax_bwd: C=%d unsupported", C);
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "upsample_softmax_bwd: bad dtype");
  if (ws_bytes < pxl_upsample_bwd_workspace(B, w, C, H))
    return pxl_set_error(PXL_ERR_WORKSPACE, "upsample_softmax_bwd: workspace too small");
  // (1 x 1 -> 1 x 1 is the identity resize of a pooled classifier head, ssl_s4l.py:389-391; align_corners needs H, W > 1)
  PXL_REQUIRE(H >= 1 && W >= 1 && h >= 1 && w >= 1 && (!align_corners || (H > 1 && W > 1)), "upsample_softmax_bwd: degenerate sizes");
  const int align = align_corners ? 1 : 0;
  const float sy = align ? (float)(h - 1) / (float)(H - 1) : (float)h / (float)H;
  const float sx = align ? (float)(w - 1) / (float)(W - 1) : (float)w / (float)W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t smem = (size_t)C * (W + 1) * sizeof(float) + (size_t)(W + w + 1) * sizeof(float);      // G, l1 per column, the start() table
  PXL_REQUIRE(smem <= 64 * 1024, "upsample_softmax_bwd: row too wide for LDS staging (W=%d)", W);
  // threads per row: W = 513 in two even passes (320 + 193) instead of 256 + 256 + 1
  const int rows_threads = (W > 256 && W <= 640) ? 320 : 256;
  hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3(H, B), dim3(rows_threads), smem, s, C, H, W, w, sx, align, dlogits, dprob,
                     prob, (float*)workspace);
  PXL_LAUNCH_CHECK();
  const long total = (long)B * h * w * Cp;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<float>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (float*)dlow, nullptr);
  else
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<bf16_t>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (bf16_t*)dlow, nullptr);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}


// LDS bytes of head_loss_rows_kernel; the fused seam runs when this fits the default 64 KiB dynamic limit
extern "C" size_t pxl_head_loss_lds_bytes(int w, int C, int W) {
  return (size_t)C * (W + 1) * sizeof(float) + (size_t)W * 3 * sizeof(float) + (size_t)4 * w * C * sizeof(float);
}

// Fused training seam (see head_loss_rows_kernel).  s_low / t_low: low-resolution logits NHWC [B][h][w][Cp] in the engine
// dtype (t_low NULL: no teacher -> no teacher CE, no consistency term); gt: float class ids [n_ce][H][W]; ce_weight =
// d(final loss)/d(per-sample CE) (1 / n_ce for torch.mean over the labeled samples), mse_weight = d(final loss)/d(MSE
// mean) (ramp * cons_scale).  Outputs: dlow [B][h][w][Cp] (engine dtype) = d(final loss)/d(s_low); sums [2*B + 1]
// fp32, zeroed here: per-sample student CE, per-sample teacher CE, the MSE mean over samples [mse_lo, mse_hi).
// workspace: pxl_upsample_bwd_workspace(B, w, C, H) bytes.
namespace {
// kernel_choice: -1 = PXL_HEAD_LOSS_CELLS / the heuristic, 0 = row-wise kernel, 1 = cell-wise kernel.  ordered: bit-reproducible
// results -- the row-wise kernel (its gradient goes through plain stores) with the loss sums folded in row order; needs
// B * H * 12 bytes of workspace beyond pxl_upsample_bwd_workspace (PXL_ERR_WORKSPACE otherwise).
int head_loss_impl(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                   const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi,
                   float ce_weight, float mse_weight, const float* mse_w_dev, void* dlow, void* workspace, size_t ws_bytes, float* sums,
                   void* stream, int kernel_choice = -1, bool ordered = false) {
  PXL_REQUIRE(s_low && dlow && workspace && sums, "head_loss: null argument");
  PXL_REQUIRE(n_ce == 0 || gt != nullptr, "head_loss: labels missing");
  PXL_REQUIRE(C >= 1 && C <= MAXC && C <= Cp, "head_loss: C=%d unsupported (max %d)", C, MAXC);
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "head_loss: bad dtype");
  PXL_REQUIRE(n_ce >= 0 && n_ce <= B && 0 <= mse_lo && mse_lo <= mse_hi && mse_hi <= B, "head_loss: bad sample ranges");
  PXL_REQUIRE(H >= 1 && W >= 1 && h >= 1 && w >= 1 && (!align_corners || (H > 1 && W > 1)), "head_loss: degenerate sizes");
  if (ws_bytes < pxl_upsample_bwd_workspace(B, w, C, H)) return pxl_set_error(PXL_ERR_WORKSPACE, "head_loss: workspace too small");
  const size_t smem = pxl_head_loss_lds_bytes(w, C, W);
  if (smem > 64 * 1024) return pxl_set_error(PXL_ERR_UNSUPPORTED, "head_loss: row too wide for LDS staging (W=%d)", W);
  if (w + 1 > 2 * W) return pxl_set_error(PXL_ERR_UNSUPPORTED, "head_loss: a head that shrinks its input by more than 2 (w=%d -> W=%d)", w, W);
  const int align = align_corners ? 1 : 0;
  const float sy = align ? (float)(h - 1) / (float)(H - 1) : (float)h / (float)H;
  const float sx = align ? (float)(w - 1) / (float)(W - 1) : (float)w / (float)W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  PXL_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)(2 * B + 1) * sizeof(float), s));
  const float hw = (float)H * (float)W;
  // cell-wise kernel for large up-sampling factors (DeepLab: x16), row-wise kernel otherwise (PSPNet's head resizes a map
  // of nearly the output size: a "cell" there is a pixel or two).  PXL_HEAD_LOSS_CELLS=0/1 forces either.
  const char* fc_env = getenv("PXL_HEAD_LOSS_CELLS");          // (read per call: the tests cover both kernels)
  const int force_cells = (ordered || kernel_choice == 0) ? 0 : (kernel_choice == 1 ? 1 : (fc_env ? atoi(fc_env) : -1));
  const size_t base_ws = pxl_upsample_bwd_workspace(B, w, C, H);
  if (ordered && ws_bytes < base_ws + (size_t)B * H * 3 * sizeof(float))
    return pxl_set_error(PXL_ERR_WORKSPACE, "head_loss: the ordered variant needs %zu more bytes of workspace", (size_t)B * H * 3 * sizeof(float));
  float* const rowpart = ordered ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + base_ws) : nullptr;
  const long ncell = (long)B * h * w;
  const size_t cells_ws = (size_t)ncell * C * sizeof(float) + (size_t)((ncell + 3) / 4 * 4) * 3 * sizeof(float);
  const float xscale = w > 1 ? (float)(W - 1) / (float)(w - 1) : (float)W;     // widest cell ~ ceil(scale) + 1 pixels
  const bool cells = (force_cells >= 0 ? force_cells != 0 : ((long)H * W >= 36L * h * w)) && cells_ws <= ws_bytes && C == 21 &&
                     xscale <= 60.f;
  if (cells) {
    const double mse_n_ = (double)(mse_hi - mse_lo) * C * (double)H * W;
    const float ce_scale_ = ce_weight / hw;
    const float mse_scale_ = mse_hi > mse_lo ? (float)(2.0 / mse_n_) * mse_weight : 0.f;
    const float inv_mse_n_ = mse_hi > mse_lo ? (float)(1.0 / mse_n_) : 0.f;
    float* dacc = reinterpret_cast<float*>(workspace);
    float* part = dacc + ncell * C;
    PXL_CHECK_HIP(hipMemsetAsync(dacc, 0, (size_t)ncell * C * sizeof(float), s));
    const int nblk = (int)((ncell + 3) / 4);
    const int nconv = 64;
    if (dtype == PXL_F32) {
      hipLaunchKernelGGL((head_loss_cells_kernel<float, 21>), dim3(nblk), dim3(256), 0, s, Cp, h, w, H, W, sy, sx, align,
                         (const float*)s_low, (const float*)t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_scale_, mse_scale_,
                         dacc, part, B, mse_w_dev);
      hipLaunchKernelGGL(head_loss_finish_kernel<float>, dim3(nconv + B), dim3(256), 0, s, B, h, w, C, Cp, dacc, (float*)dlow, part,
                         1.f / hw, inv_mse_n_, n_ce, t_low != nullptr ? 1 : 0, sums);
    } else {
      hipLaunchKernelGGL((head_loss_cells_kernel<bf16_t, 21>), dim3(nblk), dim3(256), 0, s, Cp, h, w, H, W, sy, sx, align,
                         (const bf16_t*)s_low, (const bf16_t*)t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_scale_, mse_scale_,
                         dacc, part, B, mse_w_dev);
      hipLaunchKernelGGL(head_loss_finish_kernel<bf16_t>, dim3(nconv + B), dim3(256), 0, s, B, h, w, C, Cp, dacc, (bf16_t*)dlow, part,
                         1.f / hw, inv_mse_n_, n_ce, t_low != nullptr ? 1 : 0, sums);
    }
    PXL_LAUNCH_CHECK();
    return PXL_OK;
  }
  const float ce_scale = ce_weight / hw;                                         // = g_ce[n] / HW of ce_bwd_kernel
  const double mse_n = (double)(mse_hi - mse_lo) * C * (double)H * W;
  const float mse_scale = mse_hi > mse_lo ? (float)(2.0 / mse_n) * mse_weight : 0.f;      // = g_mse * two_inv_n of mse_bwd_kernel
  const float inv_mse_n = mse_hi > mse_lo ? (float)(1.0 / mse_n) : 0.f;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(head_loss_rows_kernel<float>, dim3(H, B), dim3(256), smem, s, C, Cp, h, w, H, W, sy, sx, align,
                       (const float*)s_low, (const float*)t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_scale, mse_scale,
                       1.f / hw, inv_mse_n, (float*)workspace, sums, B, mse_w_dev, rowpart);
  else
    hipLaunchKernelGGL(head_loss_rows_kernel<bf16_t>, dim3(H, B), dim3(256), smem, s, C, Cp, h, w, H, W, sy, sx, align,
                       (const bf16_t*)s_low, (const bf16_t*)t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_scale, mse_scale,
                       1.f / hw, inv_mse_n, (float*)workspace, sums, B, mse_w_dev, rowpart);
  PXL_LAUNCH_CHECK();
  if (rowpart != nullptr) {
    hipLaunchKernelGGL(head_loss_rows_finish_kernel, dim3(1), dim3(64 * ((2 * B + 1 + 63) / 64)), 0, s, B, H, rowpart, sums);
    PXL_LAUNCH_CHECK();
  }
  const long total = (long)B * h * w * Cp;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<float>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (float*)dlow, nullptr);
  else
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<bf16_t>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (bf16_t*)dlow, nullptr);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
}  // namespace

extern "C" int pxl_head_loss(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                             const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi,
                             float ce_weight, float mse_weight, void* dlow, void* workspace, size_t ws_bytes, float* sums,
                             void* stream) {
  return head_loss_impl(dtype, B, h, w, Cp, C, H, W, align_corners, s_low, t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_weight,
                        mse_weight, nullptr, dlow, workspace, ws_bytes, sums, stream);
}

// pxl_head_loss / pxl_head_loss_hp (mse_weight_dev != NULL) with the kernel chosen by the caller and, `ordered` != 0, bit-reproducible
// results (see head_loss_impl): the executor's PXL_DETERMINISTIC mode
extern "C" int pxl_head_loss_ex(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                                const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi,
                                float ce_weight, float mse_weight, const float* mse_weight_dev, int kernel_choice, int ordered,
                                void* dlow, void* workspace, size_t ws_bytes, float* sums, void* stream) {
  return head_loss_impl(dtype, B, h, w, Cp, C, H, W, align_corners, s_low, t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_weight,
                        mse_weight_dev != nullptr ? 1.0f : mse_weight, mse_weight_dev, dlow, workspace, ws_bytes, sums, stream,
                        kernel_choice, ordered != 0);
}

// pxl_head_loss with the consistency weight d(final loss)/d(MSE mean) in DEVICE memory (*mse_weight_dev, a per-step scalar:
// ramp-up x cons_scale, ssl_mt.py:190-196): the launch arguments no longer change from step to step (hipGraph replay)
extern "C" int pxl_head_loss_hp(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                                const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi,
                                float ce_weight, const float* mse_weight_dev, void* dlow, void* workspace, size_t ws_bytes,
                                float* sums, void* stream) {
  PXL_REQUIRE(mse_weight_dev != nullptr, "head_loss_hp: null weight pointer");
  return head_loss_impl(dtype, B, h, w, Cp, C, H, W, align_corners, s_low, t_low, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_weight,
                        1.0f, mse_weight_dev, dlow, workspace, ws_bytes, sums, stream);
}


namespace {
// ---- consistency seam of one SSLCCT auxiliary decoder ---------------------------------------------------------------------
// The reference resizes every auxiliary prediction to the size of the main prediction (F.interpolate, bilinear,
// align_corners=False), applies the channel soft-max (task/sseg/func.py sslcct_activate_ad_preds) and takes nn.MSELoss against
// the detached soft-max of the main decoder (ssl_cct.py:482-484); autograd then walks back through the MSE, the soft-max and
// the resize.  Per decoder at 4 x 21 x 513 x 513 that is five launches over 88 MB planes (up-sampling + soft-max 44 us, MSE
// forward 51, MSE backward 47, the two adjoint passes 129 + 67).  Everything between the decoder's own-resolution logits and
// their gradient is a function of (low, target): this kernel evaluates it per output row in registers / LDS --
//   per pixel:  z = bilinear(low), p = softmax(z), d = p - target, loss += d^2,
//               dp = (2 / n) d,   G = p (dp - sum_c dp p)                               (the same expressions as the separate kernels)
//   then the x-reduction of upsample_bwd_rows_kernel on G -> tmp[b][y][x0][c]
// -- for a UNIT incoming gradient; the loss is a scalar, so pxl_cons_head_bwd only scales: upsample_bwd_cols_kernel finishes
// d(low) and multiplies by the incoming gradient read from device memory.  One block per (row y, sample b); LDS: G [C][W+1]
// and the column tables (the low-resolution corners come from global memory as 16-byte chunks, like the forward kernel).
// CT: the class count as a compile-time constant (21: the sseg workloads; registers for 21 instead of MAXC channels and no guards)
// or 0 = run-time C <= MAXC.
template <typename T, int CT>
__global__ __launch_bounds__(320) void cons_rows_kernel(int Crt, int Cp, int h, int w, int H, int W, float sy, float sx, int align,
                                                        const T* __restrict__ low, const float* __restrict__ target,
                                                        float two_inv_n, float inv_n, float* __restrict__ tmp,
                                                        float* __restrict__ loss, float* __restrict__ rowpart) {
  constexpr int CC = CT ? CT : MAXC;
  const int C = CT ? CT : Crt;
  extern __shared__ float g[];   // [C][W+1] | l1 per column [W] | the start() table [w + 1]
  __shared__ float red[8];
  const int y = blockIdx.x, b = blockIdx.y;
  const int ld = W + 1;
  float* cl1 = g + (size_t)C * ld;
  int* cs = reinterpret_cast<int*>(cl1 + W);
  for (int v = threadIdx.x; v <= w; v += blockDim.x) cs[v] = first_dst_ge(v, sx, align, w, W);
  int y0, y1;
  float ly;
  src_coord(y, sy, align, h, y0, y1, ly);
  const size_t plane = (size_t)H * W;
  const size_t base = (size_t)b * C * plane + (size_t)y * W;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int NQ = (CC + EPC - 1) / EPC;
  float acc = 0.f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    int x0, x1;
    float lx;
    src_coord(x, sx, align, w, x0, x1, lx);
    cl1[x] = lx;
    float t[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) t[c] = (CT || c < C) ? target[base + c * plane + x] : 0.f;      // (every plane load in flight before the first use)
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const T* p00 = low + ((size_t)(b * h + y0) * w + x0) * Cp;
    const T* p01 = low + ((size_t)(b * h + y0) * w + x1) * Cp;
    const T* p10 = low + ((size_t)(b * h + y1) * w + x0) * Cp;
    const T* p11 = low + ((size_t)(b * h + y1) * w + x1) * Cp;
    float v[NQ * EPC];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (CT || q * EPC < C) {
        float a[EPC], bq[EPC], cq[EPC], d[EPC];
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p00 + q * EPC), a);
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p01 + q * EPC), bq);
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p10 + q * EPC), cq);
        Chunk<T>::unpack(*reinterpret_cast<const uint4*>(p11 + q * EPC), d);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const int c = q * EPC + e;
          v[c] = w00 * a[e] + w01 * bq[e] + w10 * cq[e] + w11 * d[e];          // (upsample_softmax_fwd_kernel's expression)
          if (c < C) mx = fmaxf(mx, v[c]);
        }
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c)
      if (CT || c < C) { v[c] = __expf(v[c] - mx); sum += v[c]; }
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c)
      if (CT || c < C) {
        v[c] = v[c] * inv;                       // p
        const float d = v[c] - t[c];
        acc += d * d;
        t[c] = two_inv_n * d;                    // dp (mse_bwd_kernel with a unit incoming gradient)
        dot += t[c] * v[c];
      }
#pragma unroll
    for (int c = 0; c < CC; ++c)
      if (CT || c < C) g[c * ld + x] = v[c] * (t[c] - dot);       // soft-max Jacobian (upsample_bwd_rows_kernel's expression)
  }
  __syncthreads();
  for (int o = threadIdx.x; o < w * C; o += blockDim.x) {
    const int c = o % C, x0 = o / C;
    tmp[(((size_t)b * H + y) * w + x0) * C + c] = adjoint_run_lds(g + c * ld, 1, cl1, cs, x0, w);
  }
  // the loss: one atomic per block -- or, rowpart != NULL, one plain store per block, folded in row order afterwards
  const float vs = wave_sum(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = vs;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += red[k];
    tot *= inv_n;
    if (rowpart != nullptr) rowpart[(size_t)b * H + y] = tot; else atomicAdd(loss, tot);
  }
}

__global__ void cons_rows_finish_kernel(int n, const float* __restrict__ rowpart, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float a = 0.f;
    for (int i = 0; i < n; ++i) a += rowpart[i];
    loss[0] = a;
  }
}
}  // namespace

extern "C" size_t pxl_cons_head_lds_bytes(int w, int C, int W) { return (size_t)C * (W + 1) * sizeof(float) + (size_t)(W + w + 1) * sizeof(float); }
extern "C" size_t pxl_cons_head_workspace(int B, int w, int C, int H) { return pxl_upsample_bwd_workspace(B, w, C, H) + (size_t)B * H * sizeof(float); }

// low: NHWC [B][h][w][Cp] in the engine dtype (the decoder's own-resolution logits); target: NCHW fp32 [B][C][H][W] (the main
// decoder's soft-max, detached).  Writes loss[0] = mean((softmax(resize(low)) - target)^2) and, into `workspace`
// (pxl_cons_head_workspace bytes), the row-reduced gradient for a unit incoming gradient -- consumed by pxl_cons_head_bwd.
// ordered != 0: the loss is folded in row order (bit-reproducible).
extern "C" int pxl_cons_head_fwd(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* low,
                                 const float* target, void* workspace, size_t ws_bytes, float* loss, int ordered, void* stream) {
  PXL_REQUIRE(low && target && workspace && loss, "cons_head_fwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "cons_head_fwd: bad dtype");
  PXL_REQUIRE(C >= 1 && C <= MAXC && C <= Cp && Cp % 8 == 0, "cons_head_fwd: C=%d / pitch %d unsupported (max %d, pitch a multiple of 8)", C, Cp, MAXC);
  PXL_REQUIRE(B >= 1 && H >= 1 && W >= 1 && h >= 1 && w >= 1 && (!align_corners || (H > 1 && W > 1)), "cons_head_fwd: degenerate sizes");
  if (ws_bytes < pxl_cons_head_workspace(B, w, C, H)) return pxl_set_error(PXL_ERR_WORKSPACE, "cons_head_fwd: workspace too small");
  const size_t smem = pxl_cons_head_lds_bytes(w, C, W);
  if (smem > 64 * 1024) return pxl_set_error(PXL_ERR_UNSUPPORTED, "cons_head_fwd: row too wide for LDS staging (W=%d)", W);
  const int align = align_corners ? 1 : 0;
  const float sy = align ? (float)(h - 1) / (float)(H - 1) : (float)h / (float)H;
  const float sx = align ? (float)(w - 1) / (float)(W - 1) : (float)w / (float)W;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const double n = (double)B * C * (double)H * W;
  float* rowpart = ordered ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + pxl_upsample_bwd_workspace(B, w, C, H)) : nullptr;
  if (!ordered) PXL_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), s));
  const int threads = (W > 256 && W <= 640) ? 320 : 256;
  const float s2 = (float)(2.0 / n), s1 = (float)(1.0 / n);
  if (dtype == PXL_F32 && C == 21)
    hipLaunchKernelGGL((cons_rows_kernel<float, 21>), dim3(H, B), dim3(threads), smem, s, C, Cp, h, w, H, W, sy, sx, align, (const float*)low,
                       target, s2, s1, (float*)workspace, loss, rowpart);
  else if (dtype == PXL_F32)
    hipLaunchKernelGGL((cons_rows_kernel<float, 0>), dim3(H, B), dim3(threads), smem, s, C, Cp, h, w, H, W, sy, sx, align, (const float*)low,
                       target, s2, s1, (float*)workspace, loss, rowpart);
  else if (C == 21)
    hipLaunchKernelGGL((cons_rows_kernel<bf16_t, 21>), dim3(H, B), dim3(threads), smem, s, C, Cp, h, w, H, W, sy, sx, align, (const bf16_t*)low,
                       target, s2, s1, (float*)workspace, loss, rowpart);
  else
    hipLaunchKernelGGL((cons_rows_kernel<bf16_t, 0>), dim3(H, B), dim3(threads), smem, s, C, Cp, h, w, H, W, sy, sx, align, (const bf16_t*)low,
                       target, s2, s1, (float*)workspace, loss, rowpart);
  PXL_LAUNCH_CHECK();
  if (ordered) {
    hipLaunchKernelGGL(cons_rows_finish_kernel, dim3(1), dim3(64), 0, s, B * H, rowpart, loss);
    PXL_LAUNCH_CHECK();
  }
  return PXL_OK;
}

// d(low) [B][h][w][Cp] (engine dtype) = gout[0] * (the gradient pxl_cons_head_fwd left in `workspace`); gout: device scalar
extern "C" int pxl_cons_head_bwd(int dtype, int B, int h, int w, int Cp, int C, int H, int align_corners, const void* workspace,
                                 size_t ws_bytes, const float* gout, void* dlow, void* stream) {
  PXL_REQUIRE(workspace && gout && dlow, "cons_head_bwd: null argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "cons_head_bwd: bad dtype");
  PXL_REQUIRE(C >= 1 && C <= MAXC && C <= Cp, "cons_head_bwd: C=%d unsupported (max %d)", C, MAXC);
  if (ws_bytes < pxl_upsample_bwd_workspace(B, w, C, H)) return pxl_set_error(PXL_ERR_WORKSPACE, "cons_head_bwd: workspace too small");
  const int align = align_corners ? 1 : 0;
  const float sy = align ? (float)(h - 1) / (float)(H - 1) : (float)h / (float)H;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = (long)B * h * w * Cp;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<float>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (float*)dlow, gout);
  else
    hipLaunchKernelGGL(upsample_bwd_cols_kernel<bf16_t>, dim3(grid), dim3(256), (size_t)(H + h + 1) * sizeof(float), s, B, h, w, Cp, C, H, sy, align,
                       (const float*)workspace, (bf16_t*)dlow, gout);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
