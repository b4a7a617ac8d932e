// Weight-gradient of the implicit-GEMM convolution (gfx950), fp32 accumulate into the
// master-gradient layout [K][taps][C] (== channels_last OIHW), split over pixels.
//
//   dW[k][t][c] += sum_m dY[m][k] * A(m,t,c)        A = act(in) gathered like the forward
//
// GEMM view: rows = out channels k, columns = (tap, in channel), reduction = pixels m.
// Both operands arrive pixel-major from HBM (NHWC), i.e. the reduction index is the
// *slow* one, so the MFMA fragments need a transpose:
//   * fp32 (v_mfma_f32_32x32x2_f32): one value per lane -> LDS image [pixel][channel],
//     conflict-free ds_read_b32, no transpose needed;
//   * bf16 (v_mfma_f32_32x32x16_bf16): 8 reduction elements per lane -> each loader
//     thread takes 4 consecutive pixels x 8 channels, transposes 4x8 in registers and
//     writes an LDS image [pixel/4][channel][4 pixels] (8-channel blocks padded 64->80 B
//     so the 16-byte stores of 8 neighbouring lanes hit distinct banks); a fragment is two
//     ds_read_b64.  The reduction order is permuted identically for both operands.
// Pixels are split across blockIdx.y; partial tiles are combined with fp32 atomics
// (one red per element per split), which also implements gradient accumulation.
#include "common.h"

namespace {

struct WgradArgs {
  const void* in;
  const void* dy;
  float* dw;
  const float* in_scale;
  const float* in_shift;
  int B, Hi, Wi, Cin;
  int Ho, Wo, Cout;
  int Kreal, Creal, dw_cpitch;
  int ntaps, so, relu_in;
  int M, m_per_split;
  int Ng;          // ntaps*Cin (GEMM columns)
  float inv_wo, inv_howo;
  int tiles_k, tiles_c;
  int taps[64];
};

template <typename T> struct WLayout;
template <> struct WLayout<float> {
  static constexpr int BKP = 16;   // pixels per K step
  // bytes of one operand tile with NCH channels
  static constexpr int tile_bytes(int nch) { return BKP * nch * 4; }
};
template <> struct WLayout<bf16_t> {
  static constexpr int BKP = 32;
  static constexpr int tile_bytes(int nch) { return (BKP / 4) * (nch / 8) * 80; }
};

// store one loader task (4 consecutive pixels x one 16-byte channel chunk) into the LDS image
template <typename T, int NCH>
__device__ __forceinline__ void store_task(unsigned char* base, int pq, int cc, const uint4 (&v)[4]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint4*>(base + ((pq * 4 + j) * NCH) * 4 + cc * 16) = v[j];
  } else {
    // v[j] = 8 channels of pixel j; emit per channel pair [c0: p0 p1 p2 p3 | c1: p0 p1 p2 p3]
    unsigned char* dst = base + (pq * (NCH / 8) + cc) * 80;
    const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&v[0]);
    const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&v[1]);
    const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&v[2]);
    const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&v[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // dword j holds channels 2j (lo) and 2j+1 (hi)
      uint4 o;
      o.x = (w0[j] & 0xffffu) | (w1[j] << 16);
      o.y = (w2[j] & 0xffffu) | (w3[j] << 16);
      o.z = (w0[j] >> 16) | (w1[j] & 0xffff0000u);
      o.w = (w2[j] >> 16) | (w3[j] & 0xffff0000u);
      *reinterpret_cast<uint4*>(dst + j * 16) = o;
    }
  }
}

template <typename T, int BMK, int BNC, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  constexpr int PF = 3;                       // register prefetch stages
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKP = WLayout<T>::BKP;
  constexpr int PQ = BKP / 4;                 // pixel quads per K step
  constexpr int CA = BMK / EPC, CB = BNC / EPC;
  constexpr int NA = CA * PQ, NB = CB * PQ;   // loader tasks per K step
  constexpr int NT = (NA + NB + 255) / 256;
  constexpr int TM = BMK / (32 * WM), TN = BNC / (32 * WN);
  constexpr int ABYTES = WLayout<T>::tile_bytes(BMK);
  constexpr int BBYTES = WLayout<T>::tile_bytes(BNC);
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                    // [2][ABYTES]
  unsigned char* sB = smem + 2 * ABYTES;       // [2][BBYTES]
  float* sAff = reinterpret_cast<float*>(smem + 2 * (ABYTES + BBYTES));

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int tile = blockIdx.x;
  const int tk = tile / p.tiles_c, tc = tile % p.tiles_c;
  const int k0 = tk * BMK, j0 = tc * BNC;
  const int m_begin = blockIdx.y * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int nks = (m_end - m_begin + BKP - 1) / BKP;

  const bool has_aff = p.in_scale != nullptr;
  if (has_aff) {
    for (int i = tid; i < p.Cin; i += 256) {
      sAff[i] = p.in_scale[i];
      sAff[p.Cin + i] = p.in_shift[i];
    }
  }

  const T* __restrict__ gin = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ gdy = reinterpret_cast<const T*>(p.dy);
  const int HoWo = p.Ho * p.Wo;

  // ---- loader task state
  int t_kind[NT], t_pq[NT], t_cc[NT];          // 0 = dY, 1 = act, 2 = idle
  int t_col[NT];                               // dY: channel offset ; act: channel c
  int t_dy[NT], t_dx[NT];                      // act: tap offsets
  bool t_colok[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int task = tid + 256 * i;
    t_dy[i] = t_dx[i] = 0;
    if (task < NA) {
      t_kind[i] = 0;
      t_pq[i] = task / CA;
      t_cc[i] = task % CA;
      t_col[i] = k0 + t_cc[i] * EPC;
      t_colok[i] = t_col[i] < p.Cout;
    } else if (task < NA + NB) {
      const int u = task - NA;
      t_kind[i] = 1;
      t_pq[i] = u / CB;
      t_cc[i] = u % CB;
      const int j = j0 + t_cc[i] * EPC;
      t_colok[i] = j < p.Ng;
      const int t = t_colok[i] ? j / p.Cin : 0;
      t_col[i] = j - t * p.Cin;
      const int tp = p.taps[t];
      t_dy[i] = tp >> 16;
      t_dx[i] = (int)(short)(tp & 0xffff);
    } else {
      t_kind[i] = 2;
      t_pq[i] = t_cc[i] = t_col[i] = 0;
      t_colok[i] = false;
    }
  }

  // PF register stages in flight (see conv_igemm.hip); the fused activation of the `in` operand is
  // applied when a stage is written to LDS so the loads are never waited for at issue time.
  uint4 stage[PF][NT][4];
  int st_ok[PF][NT];

  // Loads are unconditional (invalid lanes read element 0 and are zeroed at store time): a branch around a
  // load makes hipcc emit vmcnt(0) behind it and serialises the whole stage (see conv_igemm.hip).
  auto issue_loads = [&](int ks, uint4 (&q)[NT][4], int (&okm)[NT]) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int mq = m_begin + ks * BKP + t_pq[i] * 4;
      okm[i] = 0;
      if (t_kind[i] == 0) {                     // wave-uniform for the 128x128 tile (tasks 0..127 / 128..255)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mq + j;
          const bool ok = t_colok[i] && m < m_end;
          q[i][j] = *reinterpret_cast<const uint4*>(gdy + (ok ? (size_t)m * p.Cout + t_col[i] : (size_t)0));
          okm[i] |= (ok ? 1 : 0) << j;
        }
      } else if (t_kind[i] == 1) {
        int b, r, oy, ox;
        fast_divmod(mq, HoWo, p.inv_howo, b, r);
        fast_divmod(r, p.Wo, p.inv_wo, oy, ox);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mq + j;
          const int iy = oy * p.so + t_dy[i], ix = ox * p.so + t_dx[i];
          const bool ok = t_colok[i] && m < m_end && ((unsigned)iy < (unsigned)p.Hi) &&
                          ((unsigned)ix < (unsigned)p.Wi);
          const size_t off = ((size_t)((b * p.Hi + iy) * p.Wi + ix)) * p.Cin + t_col[i];
          q[i][j] = *reinterpret_cast<const uint4*>(gin + (ok ? off : (size_t)0));
          okm[i] |= (ok ? 1 : 0) << j;
          // next pixel (branch-free wrap)
          ++ox;
          const bool wx = ox == p.Wo;
          ox = wx ? 0 : ox;
          oy = wx ? oy + 1 : oy;
          const bool wy = oy == p.Ho;
          oy = wy ? 0 : oy;
          b = wy ? b + 1 : b;
        }
      }
    }
  };

  auto store_stage = [&](int buf, uint4 (&q)[NT][4], int (&okm)[NT]) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      if (t_kind[i] == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!((okm[i] >> j) & 1)) q[i][j] = zero;
        store_task<T, BMK>(sA + buf * ABYTES, t_pq[i], t_cc[i], q[i]);
      } else if (t_kind[i] == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint4 v = q[i][j];
          if (has_aff) {
            float f[EPC];
            Chunk<T>::unpack(v, f);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              float z = f[e] * sAff[t_col[i] + e] + sAff[p.Cin + t_col[i] + e];
              f[e] = p.relu_in ? fmaxf(z, 0.f) : z;
            }
            v = Chunk<T>::pack(f);
          }
          q[i][j] = ((okm[i] >> j) & 1) ? v : zero;
        }
        store_task<T, BNC>(sB + buf * BBYTES, t_pq[i], t_cc[i], q[i]);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();
  // tiles past nks are all-zero (m >= m_end), so the loop below runs without conditionals for a trip count
  // rounded up to a multiple of PF (see conv_igemm.hip)
  issue_loads(0, stage[0], st_ok[0]);
  store_stage(0, stage[0], st_ok[0]);
#pragma unroll
  for (int u = 1; u <= PF; ++u) issue_loads(u, stage[u % PF], st_ok[u % PF]);
  __syncthreads();

  const int frow = lane & 31, fhalf = lane >> 5;
  const int nsteps = (nks + PF - 1) / PF * PF;
  for (int ks0 = 0; ks0 < nsteps; ks0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ks = ks0 + u;
      const int buf = ks & 1;
      const unsigned char* a = sA + buf * ABYTES;
      const unsigned char* b = sB + buf * BBYTES;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < BKP / 2; ++e) {
          const int pix = 2 * e + fhalf;
          float fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[i] = *reinterpret_cast<const float*>(a + (pix * BMK + (wm * TM + i) * 32 + frow) * 4);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[j] = *reinterpret_cast<const float*>(b + (pix * BNC + (wn * TN + j) * 32 + frow) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < BKP / 16; ++kk) {
          const int q0 = 4 * kk + 2 * fhalf;
          uint4 fa[TM], fb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int ch = (wm * TM + i) * 32 + frow;
            const uint2 lo = *reinterpret_cast<const uint2*>(a + (q0 * (BMK / 8) + (ch >> 3)) * 80 + (ch & 7) * 8);
            const uint2 hi = *reinterpret_cast<const uint2*>(a + ((q0 + 1) * (BMK / 8) + (ch >> 3)) * 80 + (ch & 7) * 8);
            fa[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int ch = (wn * TN + j) * 32 + frow;
            const uint2 lo = *reinterpret_cast<const uint2*>(b + (q0 * (BNC / 8) + (ch >> 3)) * 80 + (ch & 7) * 8);
            const uint2 hi = *reinterpret_cast<const uint2*>(b + ((q0 + 1) * (BNC / 8) + (ch >> 3)) * 80 + (ch & 7) * 8);
            fb[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]),
                                                                  __builtin_bit_cast(bf16x8, fb[j]),
                                                                  acc[i][j], 0, 0, 0);
        }
      }
      const int nx = (u + 1) % PF;      // static after unrolling
      store_stage(buf ^ 1, stage[nx], st_ok[nx]);
      issue_loads(ks + 1 + PF, stage[nx], st_ok[nx]);
      __syncthreads();
    }
  }

  // ---- epilogue: rows = out channel k (A operand index), cols = (tap, c)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = j0 + (wn * TN + j) * 32 + frow;
    if (col >= p.Ng) continue;
    const int t = col / p.Cin;
    const int c = col - t * p.Cin;
    if (c >= p.Creal) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (k < p.Kreal)
          atomicAdd(p.dw + ((size_t)k * p.ntaps + t) * p.dw_cpitch + c, acc[i][j][r]);
      }
    }
  }
}

template <typename T, int BMK, int BNC, int WM, int WN>
int launch_cfg(const WgradArgs& a, hipStream_t stream, int splits_hint) {
  WgradArgs p = a;
  constexpr int BKP = WLayout<T>::BKP;
  p.tiles_k = cdiv(p.Kreal, BMK);
  p.tiles_c = cdiv(p.Ng, BNC);
  const int tiles = p.tiles_k * p.tiles_c;
  int splits = splits_hint > 0 ? splits_hint : 1024 / tiles;
  const int max_splits = cdiv(p.M, BKP * 8);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = cdiv(p.M, splits);
  mps = cdiv(mps, BKP) * BKP;
  splits = cdiv(p.M, mps);
  p.m_per_split = mps;
  const size_t smem = 2 * (WLayout<T>::tile_bytes(BMK) + WLayout<T>::tile_bytes(BNC)) +
                      (p.in_scale ? 2 * (size_t)p.Cin * 4 : 0);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, BMK, BNC, WM, WN>), dim3(tiles, splits), dim3(256), smem,
                     stream, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

template <typename T>
int launch_wgrad(const WgradArgs& a, int force_cfg, hipStream_t stream, int splits_hint) {
  int cfg = force_cfg;
  if (cfg < 0) {
    if (a.Kreal <= 32) cfg = 2;
    else if (a.Kreal <= 64 || a.Ng <= 64) cfg = 1;
    else cfg = 0;
  }
  switch (cfg) {
    case 0: return launch_cfg<T, 128, 128, 2, 2>(a, stream, splits_hint);
    case 1: return launch_cfg<T, 64, 64, 2, 2>(a, stream, splits_hint);
    case 2: return launch_cfg<T, 32, 128, 1, 4>(a, stream, splits_hint);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_wgrad: unknown tile config %d", cfg);
  }
}

}  // namespace

extern "C" int pxl_conv_wgrad_dma_eligible(const pxl_conv_desc* d, const float* in_scale);
extern "C" int pxl_conv_wgrad_dma(const pxl_conv_desc* d, const void* in, const void* dy, float* dw, int creal,
                                  int dw_cpitch, void* stream);

extern "C" int pxl_conv_wgrad(const pxl_conv_desc* d, const void* in, const float* in_scale,
                              const float* in_shift, const void* dy, float* dw, int creal,
                              int dw_cpitch, void* stream) {
  PXL_REQUIRE(d && in && dy && dw, "conv_wgrad: null argument");
  PXL_REQUIRE(d->dtype == PXL_F32 || d->dtype == PXL_BF16, "conv_wgrad: bad dtype %d", d->dtype);
  const int epc = d->dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(d->Cin % epc == 0 && d->Cout % epc == 0, "conv_wgrad: channel pitches (%d,%d) must be multiples of %d",
              d->Cin, d->Cout, epc);
  PXL_REQUIRE(d->div == 1, "conv_wgrad: describes the forward conv (div must be 1)");
  PXL_REQUIRE(d->ntaps >= 1 && d->ntaps <= 64, "conv_wgrad: ntaps %d out of range", d->ntaps);
  PXL_REQUIRE(creal >= 1 && creal <= d->Cin && dw_cpitch >= creal, "conv_wgrad: bad creal/dw_cpitch");
  PXL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_wgrad: scale/shift must come together");
  // plain bf16 operands with Cin % 64 == 0 take the LDS-DMA kernel (conv_wgrad_dma.hip); tile_cfg 0..2 forces this one
  if (d->tile_cfg < 0 || d->tile_cfg >= 8) {
    if (pxl_conv_wgrad_dma_eligible(d, in_scale)) return pxl_conv_wgrad_dma(d, in, dy, dw, creal, dw_cpitch, stream);
    PXL_REQUIRE(d->tile_cfg < 8, "conv_wgrad: tile config %d needs plain bf16 operands with Cin %% 64 == 0", d->tile_cfg);
  }
  WgradArgs a;
  a.in = in; a.dy = dy; a.dw = dw; a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.Kreal = d->Kreal; a.Creal = creal; a.dw_cpitch = dw_cpitch;
  a.ntaps = d->ntaps; a.so = d->out_stride; a.relu_in = d->relu_in;
  a.M = d->B * d->Ho * d->Wo; a.m_per_split = 0;
  a.Ng = d->ntaps * d->Cin; a.tiles_k = a.tiles_c = 0;
  a.inv_wo = 1.0f / (float)d->Wo; a.inv_howo = 1.0f / (float)(d->Ho * d->Wo);
  PXL_REQUIRE((long)d->B * d->Ho * d->Wo < (1L << 24), "conv_wgrad: more than 2^24 output pixels");
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int hint = d->split_k > 0 ? d->split_k : 0;       // 1: one add per element of dw (PXL_DETERMINISTIC)
  if (d->dtype == PXL_F32) return launch_wgrad<float>(a, d->tile_cfg, s, hint);
  return launch_wgrad<bf16_t>(a, d->tile_cfg, s, hint);
}
