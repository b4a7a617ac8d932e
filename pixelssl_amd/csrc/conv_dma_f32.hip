// LDS-DMA implicit-GEMM convolution for gfx950, fp32 operands (the parity-meeting engine): the pipeline of
// conv_dma_kernel.h with 4-byte elements and the fp32 MFMA.
//
//   out[m][n] = sum_{t,c} in[pix(m,t)][c] * w[n][t][c]        m = (b,oy,ox), n = out channel
//
// What carries over unchanged, byte for byte: tile rows are 128 bytes (here 32 channels of ONE tap per K step), tiles go
// HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` with out-of-range lanes as zero padding, the NST-deep ring with counted
// vmcnt + one raw barrier per step, the 16-byte XOR swizzle on the source side, the swapped MFMA roles (A = weights,
// B = activations) and the staged, coalesced epilogue.  A lane's 16-byte fragment read is 4 fp32 values of k-granule
// 2*kk + (lane >> 5): element e of both operands feeds `v_mfma_f32_32x32x2_f32` number (kk, e), whose two k slots are the
// two lane halves -- over kk = 0..3, e = 0..3 every one of the 32 channels of the step is multiplied exactly once.
//
// What differs: the fp32 MFMA runs at 1/16 of the bf16 rate (256 flop/clk/CU), so a K step is 2048-4096 MFMA cycles per wave
// against ~800 cycles of DMA -- the loop is MFMA-bound, small tiles suffice, and the epilogue's operand combination is a
// run-time flag set (the ~3500-cycle dispatch that mattered for 12 us bf16 workgroups is 2 % of these).  No paired launches, no
// in-kernel finalize.  BNIN (round 6, 1x1 / stride-1 launches): the A operand is the RAW output of the previous convolution and
// relu?(bn(y)) is applied to the pieces in LDS by the lane that DMA'd them, exactly as in conv_dma_kernel.h -- in a loop that
// waits for the MFMAs the transform is free, and the parity engine sheds its 66 pxl_bn_apply_fwd launches of 35 us.
#include "conv_dma_kernel.h"

namespace pxl_dma {

constexpr int f32_lds_bytes(int BM, int BN, int NST) {
  const int ring = NST * (BM + BN) * 128;
  const int stage = BM * (BN * 4 + 16) + (256 / (BN / 4)) * 2 * BN * 4;       // staged tile + statistics partials
  return ring > stage ? ring : stage;
}

template <int BM, int BN, int NST, bool GATHER, bool BNIN = false>
__global__ __launch_bounds__(256, ((BM / 64) * (BN / 64) <= 2 ? 3 : 2)) void conv_dma_f32_kernel(const DmaArgs p) {
  static_assert(!(BNIN && GATHER), "BN-on-load: 1x1 / stride-1 launches");
  constexpr int WM = 2, WN = 2, NW = 4, NT = 256;
  constexpr int TMI = BM / WM / 32, TNI = BN / WN / 32;
  constexpr int LA = BM / (8 * NW), LB = BN / (8 * NW);
  constexpr int SB = (BM + BN) * 128;
  constexpr int ES = 4;                       // bytes per element
  constexpr int TP = BN * ES + 16;            // staging row pitch
  constexpr int TPR = BN / 4;                 // threads per output row on the read-back pass (4 channels = 16 bytes each)
  constexpr int RPP = NT / TPR;
  constexpr int NPASS = BM / RPP;
  static_assert(TMI >= 1 && TNI >= 1 && TMI <= 2 && TNI <= 2 && BM % RPP == 0, "tile");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  kernarg_touch<sizeof(DmaArgs)>();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- loader coordinates (conv_dma_kernel.h): DMA instruction g = wave + 4*q covers tile rows 8g .. 8g+7
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned voffA[LA], voffB[LB];
  int a_pix[LA], a_iy[LA], a_ix[LA], a_img[LA];
  const int HoWo = p.Ho * p.Wo;
  const bool flat = !GATHER && p.so == 1 && p.Ho == p.Hi && p.Wo == p.Wi;
  const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
#pragma unroll
  for (int q = 0; q < LA; ++q) {
    const int row = (wave + NW * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const bool in = m < p.M;
    if (flat) {
      a_iy[q] = 0; a_ix[q] = 0; a_img[q] = 0;
      a_pix[q] = m * (int)(p.Cin * ES) + chunk * 16;
    } else {
      int b, r, oy, ox;
      fast_divmod(m, HoWo, inv_howo, b, r);
      fast_divmod(r, p.Wo, inv_wo, oy, ox);
      oy = oy * p.sub_mul + p.sub_py;                  // (sub-grid launches: the pixel this row stands for; else * 1 + 0)
      ox = ox * p.sub_mul + p.sub_px;
      a_iy[q] = in ? oy * p.so : -(1 << 20);
      a_ix[q] = in ? ox * p.so : 0;
      a_img[q] = b * p.Hi * p.Wi * p.Cin * ES + chunk * 16;
      a_pix[q] = a_img[q] + (a_iy[q] * p.Wi + a_ix[q]) * p.Cin * ES;
    }
    voffA[q] = in ? (unsigned)a_pix[q] : OOB;
  }
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int row = (wave + NW * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int n = n0 + row;
    voffB[q] = n < p.Kreal ? (unsigned)(n * p.Ktot * ES + chunk * 16) : OOB;
  }

  const unsigned cin_bytes = (unsigned)p.Cin * ES;
  const int ks_begin = blockIdx.y * p.nk_per;
  const int nk_here = min(p.nk, ks_begin + p.nk_per) - ks_begin;
  int ld_t = (int)(((unsigned)ks_begin * 128u) / cin_bytes);
  unsigned kcb = (unsigned)ks_begin * 128u - (unsigned)ld_t * cin_bytes, kwb = (unsigned)ks_begin * 128u;
  auto set_tap = [&](int t) {
    if constexpr (GATHER) {
      const int tp = p.taps[min(t, p.ntaps - 1)];
      const int dy = tap_dy(tp), dx = tap_dx(tp);
      kwb = tap_wt(tp) * cin_bytes + kcb;            // this tap's slice of the weight row (a launch may walk a subset of the taps)
      const int tapoff = (dy * p.Wi + dx) * p.Cin * ES;
      if (p.div_shift == 0) {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
          const int iy = a_iy[q] + dy, ix = a_ix[q] + dx;
          const bool ok = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
          voffA[q] = ok ? (unsigned)(a_pix[q] + tapoff) : OOB;
        }
      } else {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
          const int ny = a_iy[q] + dy, nx = a_ix[q] + dx;
          const int iy = ny >> 1, ix = nx >> 1;
          const bool ok = ((ny | nx) & 1) == 0 && ny >= 0 && nx >= 0 && iy < p.Hi && ix < p.Wi;
          voffA[q] = ok ? (unsigned)(a_img[q] + (iy * p.Wi + ix) * p.Cin * ES) : OOB;
        }
      }
    }
  };
  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * SB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LA; ++q) dma16(r_in, sa + q * NW * 1024, voffA[q], kcb);
    unsigned char* sb = smem + stage * SB + BM * 128 + wave * 1024;
#pragma unroll
    for (int q = 0; q < LB; ++q) dma16(r_w, sb + q * NW * 1024, voffB[q], kwb);
    kwb += 128;
    kcb += 128;
    if (kcb == cin_bytes) {
      kcb = 0;
      ++ld_t;
      set_tap(ld_t);
    }
  };

  f32x16 acc[TNI][TMI];
#pragma unroll
  for (int j = 0; j < TNI; ++j)
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fx = (frow >> 1) & 7;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned f = frow * 128 + (((2 * kk + fhalf) ^ fx) << 4);
    aoff[kk] = f + wm * TMI * 4096;
    boff[kk] = f + wn * TNI * 4096;
  }

  set_tap(ld_t);
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);
  // ---- BNIN: (scale, shift) of every input channel -> LDS table behind the ring / staging area (conv_dma_kernel.h)
  unsigned ckc = 0;                                   // channel offset of the tile being consumed
  const unsigned tab0 = lds0 + (unsigned)f32_lds_bytes(BM, BN, NST);
  const int lchunk = (lane & 7) ^ ((wave * 4 + ((lane >> 3) >> 1)) & 7);     // the lane's (q-independent) source chunk
  if constexpr (BNIN) {
    float* tab = reinterpret_cast<float*>(smem + f32_lds_bytes(BM, BN, NST));
    const pxl_bn_fin& f = p.bin;
    const int C = p.Cin;
    for (int c = tid; c < C; c += NT) {
      float mean, var;
      if (f.training) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < f.nrep; ++r) { s1 += f.stats[(size_t)r * 2 * C + c]; s2 += f.stats[(size_t)r * 2 * C + C + c]; }
        mean = s1 / f.count;
        var = s2 / f.count - mean * mean;
        if (var < 0.f) var = 0.f;
        if (blockIdx.x == 0 && blockIdx.y == 0 && f.running_mean != nullptr) {
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
      tab[c] = scale;
      tab[C + c] = shift;
      if (blockIdx.x == 0 && blockIdx.y == 0) { f.coef[c] = mean; f.coef[C + c] = rstd; f.coef[2 * C + c] = scale; f.coef[3 * C + c] = shift; }
    }
    __syncthreads();
    ckc = ((unsigned)ks_begin * 32u) % (unsigned)C;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);          // retire the scalar argument loads before the loop (conv_dma_kernel.h)
  int st_c = 0, st_l = NST - 1;
  for (int ks = 0; ks < nk_here; ++ks) {
    wait_vmcnt<(NST - 2) * (LA + LB)>();
    if constexpr (BNIN) {
      // relu?(scale * y + shift) on the pieces THIS lane has DMA'd, in place (fp32: no rounding); rows past M stay zero
      const unsigned pa = lds0 + st_c * SB + wave * 1024 + lane * 16;
      const unsigned tb = tab0 + (ckc + (unsigned)lchunk * 4u) * 4u;
      u32x4 cf[4], dd[LA];
      cf[0] = lds_read128<0>(tb);
      cf[1] = lds_read128<0>(tb + (unsigned)p.Cin * 4u);
      cf[2] = cf[0]; cf[3] = cf[1];
      XformLoad<0, LA, NW * 1024>::run(dd, pa);
      wait_xform<LA>(dd, cf);
#pragma unroll
      for (int q = 0; q < LA; ++q) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = __uint_as_float(dd[q][e]) * __uint_as_float(cf[0][e]) + __uint_as_float(cf[1][e]);
          r[e] = __float_as_uint(p.bin_relu ? fmaxf(v, 0.f) : v);
        }
        dd[q] = voffA[q] != OOB ? r : dd[q];
      }
      XformStore<0, LA, NW * 1024>::run(dd, pa);
      if (p.bin_z != nullptr && tn == 0) {
        const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(p.bin_z, 0, p.in_bytes, 0x00020000);
#pragma unroll
        for (int q = 0; q < LA; ++q)
          __builtin_amdgcn_raw_buffer_store_b128(dd[q], r_z, (int)voffA[q], (int)(ckc * 4u), 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ckc += 32;
      if (ckc == (unsigned)p.Cin) ckc = 0;
    }
    __builtin_amdgcn_s_barrier();
    issue(st_l);
    const unsigned sbase = lds0 + st_c * SB;
    u32x4 fa[4][TMI], fw[4][TNI];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FragLoad<0, TMI, 4096, 0>::run(fa[kk], sbase + aoff[kk]);
      FragLoad<0, TNI, 4096, BM * 128>::run(fw[kk], sbase + boff[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      constexpr int PER = TMI + TNI;
      if (kk == 0) wait_chunk<(3 * PER > 15 ? 15 : 3 * PER)>(fa[0], fw[0], acc);
      if (kk == 1) wait_chunk<(2 * PER > 15 ? 15 : 2 * PER)>(fa[1], fw[1], acc);
      if (kk == 2) wait_chunk<(1 * PER > 15 ? 15 : 1 * PER)>(fa[2], fw[2], acc);
      if (kk == 3) wait_chunk<0>(fa[3], fw[3], acc);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < TNI; ++j)
#pragma unroll
          for (int i = 0; i < TMI; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fw[kk][j][e]), __uint_as_float(fa[kk][i][e]),
                                                             acc[j][i], 0, 0, 0);
    }
    st_c = st_c + 1 == NST ? 0 : st_c + 1;
    st_l = st_l + 1 == NST ? 0 : st_l + 1;
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  if (p.ws != nullptr) {
    // split-K: fp32 partial sums of this K slice -> workspace; bias happens in the finish kernel
#pragma unroll
    for (int j = 0; j < TNI; ++j)
#pragma unroll
      for (int i = 0; i < TMI; ++i) {
        const int m = m0 + (wm * TMI + i) * 32 + frow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wn * TNI + j) * 32 + 8 * (r >> 2) + 4 * fhalf + (r & 3);
          if (m < p.M && n < p.Kreal) atomicAdd(p.ws + (size_t)m * p.Cout + n, acc[j][i][r]);
        }
      }
    return;
  }
  // ---- epilogue 1: accumulators -> fp32 tile T[m][n] in LDS (column (lane & 31) = pixel, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  // = channel: a lane's accumulator quads are 4 consecutive channels = one 16-byte write)
  unsigned char* T = smem;
#pragma unroll
  for (int j = 0; j < TNI; ++j)
#pragma unroll
    for (int i = 0; i < TMI; ++i) {
      const int ml = (wm * TMI + i) * 32 + frow;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = (wn * TNI + j) * 32 + 8 * g + 4 * fhalf;
        *reinterpret_cast<float4*>(T + ml * TP + nl * ES) =
            make_float4(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
      }
    }
  __syncthreads();

  // ---- epilogue 2: coalesced read-back, bias / addend / statistics / BatchNorm-backward sums, 16-byte stores.  Flag
  // meaning as in conv_dma_kernel.h's epi_passes (run-time here).
  const int ec = tid % TPR, er = tid / TPR;
  const int n = n0 + ec * 4;
  const bool ncol = n < p.Cout;
  const bool has_bias = p.bias != nullptr, has_add = p.addend != nullptr, has_stats = p.stats != nullptr;
  const bool has_bnr = has_stats && p.bn_y != nullptr, has_mask = has_bnr && p.bn_mask != nullptr;
  const bool bn_relu = has_bnr && !has_mask && p.bn_relu;
  float bv[4] = {0.f, 0.f, 0.f, 0.f}, bn_mean[4] = {0.f, 0.f, 0.f, 0.f}, bn_rstd[4] = {0.f, 0.f, 0.f, 0.f},
        bn_sc[4] = {0.f, 0.f, 0.f, 0.f}, bn_sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = n + e < p.Kreal ? p.bias[n + e] : 0.f;
  }
  if (has_bnr && ncol) {
    load_cvec<4>(p.bn_coef + n, bn_mean);
    load_cvec<4>(p.bn_coef + p.Cout + n, bn_rstd);
    if (bn_relu) {
      load_cvec<4>(p.bn_coef + 2 * p.Cout + n, bn_sc);
      load_cvec<4>(p.bn_coef + 3 * p.Cout + n, bn_sh);
    }
  }
  const unsigned out_bytes = p.sub_mul != 1 ? (unsigned)((size_t)p.B * p.out_H * p.out_W * p.Cout * ES) : (unsigned)((size_t)p.M * p.Cout * ES);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.addend), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bny = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_y), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_mask), 0, out_bytes, 0x00020000);
  constexpr unsigned EOOB = 0xffffff00u;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int PG = 2;            // passes per group: the group's loads are issued before its first store (one vmcnt counter)
  static_assert(NPASS % PG == 0, "passes per group");
#pragma unroll 1
  for (int g0 = 0; g0 < NPASS; g0 += PG) {
    unsigned vo[PG];
    u32x4 xa[PG], xy[PG], xm[PG];
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int m = m0 + (g0 + q) * RPP + er;
      size_t pix = (size_t)m;
      if (p.sub_mul != 1) {               // row of a sub-grid launch -> pixel of the full tensor
        const int b = m / HoWo, r = m - b * HoWo;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        pix = ((size_t)b * p.out_H + oy * p.sub_mul + p.sub_py) * p.out_W + ox * p.sub_mul + p.sub_px;
      }
      vo[q] = (m < p.M && ncol) ? (unsigned)((pix * p.Cout + n) * ES) : EOOB;
    }
    if (has_add) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xa[q] = __builtin_amdgcn_raw_buffer_load_b128(r_add, (int)vo[q], 0, 0);
    }
    if (has_bnr) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xy[q] = __builtin_amdgcn_raw_buffer_load_b128(r_bny, (int)vo[q], 0, 0);
    }
    if (has_mask) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xm[q] = __builtin_amdgcn_raw_buffer_load_b128(r_msk, (int)vo[q], 0, 0);
    }
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const float4 tv = *reinterpret_cast<const float4*>(T + ((g0 + q) * RPP + er) * TP + ec * 16);
      float f[4] = {tv.x, tv.y, tv.z, tv.w};
      if (has_add) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] += __uint_as_float(xa[q][e]);
      }
      if (has_bias) {
        const bool ok = vo[q] != EOOB;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = ok ? f[e] + bv[e] : 0.f;          // (rows past M stay out of the statistics)
      }
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gd = __uint_as_float(xm[q][e]) > 0.f ? f[e] : 0.f;
          f[e] = gd;
          s1[e] += gd;
          s2[e] += gd * (__uint_as_float(xy[q][e]) - bn_mean[e]) * bn_rstd[e];
        }
      } else if (has_bnr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float fy = __uint_as_float(xy[q][e]);
          float gd = f[e];
          if (bn_relu && !(fy * bn_sc[e] + bn_sh[e] > 0.f)) gd = 0.f;
          s1[e] += gd;
          s2[e] += gd * (fy - bn_mean[e]) * bn_rstd[e];
        }
      } else if (has_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += f[e];
          s2[e] += f[e] * f[e];
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                                                   __float_as_uint(f[3])}, r_out, (int)vo[q], 0, 0);
    }
  }
  if (has_stats) {
    // reduce over the RPP threads that share a channel chunk through LDS (behind the staged tile), then one atomic per
    // (sum, channel) and workgroup
    float* red = reinterpret_cast<float*>(smem + BM * TP);           // [RPP rows][2][BN]
    float* mine = red + er * 2 * BN + ec * 4;
    *reinterpret_cast<float4*>(mine) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(mine + BN) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    __syncthreads();
    if (tid < 2 * BN) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) v += red[r * 2 * BN + tid];
      const int which = tid / BN, c = tid % BN;
      if (n0 + c < p.Kreal) {
        float* rep = p.stats + (size_t)(tm % p.stats_rep) * 2 * p.Kreal;
        atomicAdd(rep + which * p.Kreal + n0 + c, v);
      }
    }
  }
}

template <int BM, int BN, int NST>
int launch_dma_f32(const DmaArgs& a, bool gather, int want_split, size_t ws_bytes, hipStream_t stream) {
  DmaArgs p = a;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const int grid = p.tiles_m * p.tiles_n;
  int splitk = 1;
  const bool can_split = p.ws != nullptr && p.stats == nullptr && p.addend == nullptr &&
                         ws_bytes >= (size_t)p.M * p.Cout * sizeof(float);
  if (can_split) {
    if (want_split > 1) splitk = want_split;
    else if (want_split <= 0 && grid < 200 && p.nk >= 128) splitk = min(cdiv(768, grid), p.nk / 32);
    if (splitk > p.nk) splitk = p.nk;
    if (splitk < 1) splitk = 1;
  }
  p.nk_per = p.nk > 0 ? cdiv(p.nk, splitk) : 0;
  splitk = p.nk > 0 ? cdiv(p.nk, p.nk_per) : 1;          // (nk == 0: a read-back-only launch of a tap-less parity class)
  if (splitk > 1) PXL_CHECK_HIP(hipMemsetAsync(p.ws, 0, (size_t)p.M * p.Cout * sizeof(float), stream));
  else p.ws = nullptr;
  constexpr size_t smem = (size_t)f32_lds_bytes(BM, BN, NST);
  static_assert(smem <= 156 * 1024, "LDS");
  if (p.bin.coef != nullptr) {
    // BN-apply on load: the 1x1 / stride-1 kernel with the coefficient table behind its ring
    const size_t smem_b = smem + (size_t)p.Cin * 8;
    if (gather || splitk > 1 || smem_b > 156 * 1024)
      return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma (fp32): BN-on-load needs a 1x1 / stride-1 launch whose table fits the LDS");
    static bool raised_b = false;
    if (!raised_b) {
      PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_f32_kernel<BM, BN, NST, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      raised_b = true;
    }
    hipLaunchKernelGGL((conv_dma_f32_kernel<BM, BN, NST, false, true>), dim3(grid, 1), dim3(256), smem_b, stream, p);
    PXL_LAUNCH_CHECK();
    return PXL_OK;
  }
  static bool raised[2] = {false, false};
  if (!raised[gather ? 1 : 0]) {
    if (gather) PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_f32_kernel<BM, BN, NST, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    else PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_f32_kernel<BM, BN, NST, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    raised[gather ? 1 : 0] = true;
  }
  const dim3 g(grid, splitk), b(256);
  if (gather) hipLaunchKernelGGL((conv_dma_f32_kernel<BM, BN, NST, true>), g, b, smem, stream, p);
  else hipLaunchKernelGGL((conv_dma_f32_kernel<BM, BN, NST, false>), g, b, smem, stream, p);
  PXL_LAUNCH_CHECK();
  if (splitk > 1)
    return pxl_splitk_finish(PXL_F32, (long)p.M * p.Cout, p.Cout, p.Kreal, p.ws, p.bias, p.out, stream);
  return PXL_OK;
}

}  // namespace pxl_dma

// tile configurations of the fp32 kernel (the numbers of the bf16 2x2-wave families): 8 = 128x128, 9 = 128(pixels)x64,
// 10 = 64x128, 11 = 64x64 with a 3-stage ring, 16..19 the same with 2 stages
int pxl_dma_f32_launch(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s) {
  using namespace pxl_dma;
  switch (cfg) {
    case 8: return launch_dma_f32<128, 128, 3>(a, gather, sk, ws_bytes, s);
    case 9: return launch_dma_f32<128, 64, 3>(a, gather, sk, ws_bytes, s);
    case 10: return launch_dma_f32<64, 128, 3>(a, gather, sk, ws_bytes, s);
    case 11: return launch_dma_f32<64, 64, 3>(a, gather, sk, ws_bytes, s);
    case 16: return launch_dma_f32<128, 128, 2>(a, gather, sk, ws_bytes, s);
    case 17: return launch_dma_f32<128, 64, 2>(a, gather, sk, ws_bytes, s);
    case 18: return launch_dma_f32<64, 128, 2>(a, gather, sk, ws_bytes, s);
    case 19: return launch_dma_f32<64, 64, 2>(a, gather, sk, ws_bytes, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma (fp32): unknown tile config %d", cfg);
  }
}
