// LDS-DMA weight gradient for gfx950, fp32 operands (the parity-meeting engine), fp32 accumulate into the master-gradient
// layout.
//
//   dW[n][t][c] += sum_m dY[m][n] * Z[pix(m,t)][c]          Z = plain (already activated) input, NHWC
//
// conv_wgrad_dma.hip's pipeline (tiles HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds`, rows = pixels, out-of-range lanes =
// zero padding / ragged tail; NST-stage ring, counted vmcnt, one raw barrier per step, pixel splits over blockIdx.y combined
// with fp32 atomics) with 4-byte elements.  What the element size changes:
//   * the fp32 MFMA `v_mfma_f32_32x32x2_f32` takes ONE value per lane and operand: lane l holds reduction slot l >> 5 of row /
//     column l & 31.  With pixel-major tiles that is a plain `ds_read_b32` -- 32 lanes read 32 consecutive channels of pixel
//     2p, the other 32 those of pixel 2p + 1 -- so there is no transposing read and no swizzle (a 128-byte run per half-wave
//     is conflict-free);
//   * the loop is MFMA-bound (64 cycles per MFMA against one 2-cycle LDS read per operand): reads of the next three pixel
//     pairs are kept in flight while the MFMAs of the current one issue (lgkmcnt is a 4-bit counter).
#include "common.h"

namespace {

struct WFArgs {
  const void* in;
  const void* dy;
  float* dw;
  int B, Hi, Wi, Cin;
  int Ho, Wo, Cout;
  int Kreal, Creal, dw_cpitch;
  int ntaps, so;
  int M, m_per_split;
  int tiles_n, tiles_c, ctiles_per_tap;
  int step_i, step_q, step_r;   // BKP = step_i * Ho*Wo + step_q * Wo + step_r
  unsigned in_bytes, dy_bytes;
  int taps[64];
};

constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// (inline asm: a C++ LDS load makes hipcc drain the LDS-DMA ring first -- see conv_dma_kernel.h)
template <int OFF> __device__ __forceinline__ unsigned lds_read32(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// `s_waitcnt lgkmcnt(N)` threaded through the operands of ONE pixel pair and the accumulators (conv_dma_kernel.h: wait_chunk)
template <int N, int TN, int TC>
__device__ __forceinline__ void wait_pair(unsigned (&a)[TN], unsigned (&b)[TC], f32x16 (&acc)[TN][TC]) {
  if constexpr (TN == 1 && TC == 1)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(b[0]), "+a"(acc[0][0]) : "n"(N));
  else if constexpr (TN == 2 && TC == 1)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+a"(acc[0][0]), "+a"(acc[1][0]) : "n"(N));
  else if constexpr (TN == 1 && TC == 2)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]), "+a"(acc[0][0]), "+a"(acc[0][1]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]),
                   "+a"(acc[1][1])
                 : "n"(N));
}

// reads of pixel pair KP: one value per 32-channel tile and operand (tiles are 128 bytes apart in a row)
template <int KP, int TN, int TC, int RBN, int RBC>
__device__ __forceinline__ void load_pair(unsigned (&a)[TN], unsigned (&b)[TC], unsigned aaddr, unsigned baddr) {
  a[0] = lds_read32<KP * 2 * RBN>(aaddr);
  if constexpr (TN == 2) a[TN - 1] = lds_read32<KP * 2 * RBN + 128>(aaddr);
  b[0] = lds_read32<KP * 2 * RBC>(baddr);
  if constexpr (TC == 2) b[TC - 1] = lds_read32<KP * 2 * RBC + 128>(baddr);
}
template <int KP, int NP, int D, int TN, int TC, int RBN, int RBC> struct PairLoop {
  static __device__ __forceinline__ void run(unsigned (&fa)[NP][TN], unsigned (&fb)[NP][TC], f32x16 (&acc)[TN][TC], unsigned aaddr,
                                             unsigned baddr) {
    if constexpr (KP + D < NP) load_pair<KP + D, TN, TC, RBN, RBC>(fa[KP + D], fb[KP + D], aaddr, baddr);
    constexpr int ahead = NP - 1 - KP < D ? NP - 1 - KP : D;
    wait_pair<ahead * (TN + TC), TN, TC>(fa[KP], fb[KP], acc);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TC; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[KP][i]), __uint_as_float(fb[KP][j]), acc[i][j], 0, 0, 0);
    if constexpr (KP + 1 < NP) PairLoop<KP + 1, NP, D, TN, TC, RBN, RBC>::run(fa, fb, acc, aaddr, baddr);
  }
};

// tile: BN out channels x BC in channels (one tap), reduction step BKP pixels, 4 waves as 2 x 2
template <int BN, int BC, int BKP, int NST, bool GATHER>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_f32_kernel(const WFArgs p) {
  constexpr int ES = 4;
  constexpr int RBN = BN * ES, RBC = BC * ES;           // row bytes of the dY / Z tiles
  constexpr int AB = BKP * RBN, BB = BKP * RBC;         // tile bytes
  constexpr int SB = AB + BB;
  constexpr int LA = AB / 4096, LB = BB / 4096;         // DMA instructions per wave per step
  constexpr int RPA = 1024 / RBN, RPB = 1024 / RBC;     // pixel rows per DMA instruction
  static_assert((BN == 128 || BN == 64) && (BC == 128 || BC == 64) && (BKP == 32 || BKP == 64) && LA >= 1 && LB >= 1, "tile");
  constexpr int TN = BN / 64, TC = BC / 64;             // 32-wide MFMA tiles per wave (2 x 2 waves)
  constexpr int NP = BKP / 2;                           // pixel pairs (= MFMAs per accumulator tile) per step
  constexpr int DEPTH = 3;                              // pixel pairs of LDS reads kept in flight: 3 * (TN + TC) <= 12 < 16

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  kernarg_touch<sizeof(WFArgs)>();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;

  // XCD-aware remap of the whole 2-D grid: one XCD runs consecutive (split, tile) pairs = every output tile of the same pixel
  // slice (conv_wgrad_dma.hip)
  const int lid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int tile = lid % gridDim.x;
  const int split = lid / gridDim.x;
  const int tn = tile / p.tiles_c, tcg = tile % p.tiles_c;
  const int tap = tcg / p.ctiles_per_tap;
  const int c0 = (tcg % p.ctiles_per_tap) * BC;
  const int n0 = tn * BN;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int nks = (m_end - m_begin + BKP - 1) / BKP;

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);

  // ---- dY loader: instruction g = wave + 4q covers pixel rows RPA*g .. ; the walk over pixels is the soffset
  unsigned voffA[LA];
#pragma unroll
  for (int q = 0; q < LA; ++q) {
    const int row = (wave + 4 * q) * RPA + lane / (RBN / 16);
    const int chunk = lane % (RBN / 16);
    const int n = n0 + chunk * 4;
    voffA[q] = n < p.Cout ? (unsigned)(((m_begin + row) * p.Cout + n) * ES) : OOB;
  }
  // ---- Z loader: per DMA row the INPUT coordinates of tap (0,0) and the byte offset of that pixel; advancing by BKP output
  // pixels is add + two conditional wraps
  const int tp = p.taps[tap];
  const int tdy = tp >> 16, tdx = (int)(short)(tp & 0xffff);
  int ziy[LB], zix[LB], zoff[LB];
  unsigned voffB[LB];
  const int HoWo = p.Ho * p.Wo;
  const int tapoff = (tdy * p.Wi + tdx) * p.Cin * ES;
  int hos = p.Ho * p.so, wos = p.Wo * p.so, s_so = p.so, s_hi = p.Hi, s_wi = p.Wi;
  auto z_voff = [&](int q) -> unsigned {
    if constexpr (GATHER) {
      const bool ok = ((unsigned)(ziy[q] + tdy) < (unsigned)s_hi) && ((unsigned)(zix[q] + tdx) < (unsigned)s_wi);
      return ok ? (unsigned)(zoff[q] + tapoff) : OOB;
    } else {
      return (unsigned)zoff[q];
    }
  };
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int row = (wave + 4 * q) * RPB + lane / (RBC / 16);
    const int chunk = lane % (RBC / 16);
    const int m = m_begin + row;
    const int b = m / HoWo;
    const int r = m - b * HoWo;
    const int oy = r / p.Wo;
    ziy[q] = oy * p.so;
    zix[q] = (r - oy * p.Wo) * p.so;
    // pixels past M land in image B (or later): past the end of the tensor => zero-filled by the descriptor
    zoff[q] = ((b * p.Hi + ziy[q]) * p.Wi + zix[q]) * p.Cin * ES + (c0 + chunk * 4) * ES;
    voffB[q] = z_voff(q);
  }
  int adv_off = (p.step_i * p.Hi * p.Wi + p.step_q * p.so * p.Wi + p.step_r * p.so) * p.Cin * ES;   // +BKP pixels, no wrap
  int adv_y = p.step_q * p.so, adv_x = p.step_r * p.so;
  int wrap_x = (p.so * p.Wi - wos) * p.Cin * ES;                           // extra when ox wraps
  int wrap_y = (p.Hi - hos) * p.Wi * p.Cin * ES;                           // extra when oy wraps
  unsigned sdy = 0;                      // dY soffset
  unsigned dy_step = (unsigned)(BKP * p.Cout * ES);
  int zstep = BKP * p.Cin * ES;
  // pin the loop's scalars in SGPRs (conv_wgrad_dma.hip: a kernel-argument re-load inside the loop would drain the LDS reads)
  asm volatile("" : "+s"(hos), "+s"(wos), "+s"(s_so), "+s"(s_hi), "+s"(s_wi), "+s"(adv_off), "+s"(adv_y), "+s"(adv_x));
  asm volatile("" : "+s"(wrap_x), "+s"(wrap_y), "+s"(dy_step), "+s"(zstep));
  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * SB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LA; ++q) dma16(r_dy, sa + q * 4096, voffA[q], sdy);
    unsigned char* sb = smem + stage * SB + AB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LB; ++q) dma16(r_in, sb + q * 4096, voffB[q], 0);
    sdy += dy_step;
  };
  auto advance = [&]() {
#pragma unroll
    for (int q = 0; q < LB; ++q) {
      if constexpr (GATHER) {
        int ix = zix[q] + adv_x, iy = ziy[q] + adv_y, off = zoff[q] + adv_off;
        const bool wx = ix >= wos;
        ix = wx ? ix - wos : ix;
        iy = wx ? iy + s_so : iy;
        off = wx ? off + wrap_x : off;
        const bool wy = iy >= hos;
        iy = wy ? iy - hos : iy;
        off = wy ? off + wrap_y : off;
        zix[q] = ix; ziy[q] = iy; zoff[q] = off;
        voffB[q] = z_voff(q);
      } else {
        zoff[q] += zstep;               // 1x1 / stride 1: input pixel == output pixel
        voffB[q] = (unsigned)zoff[q];
      }
    }
  };

  f32x16 acc[TN][TC];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment addresses: lane = 32 h + i reads pixel 2p + h, channel (wave's half tile) + i
  const int fi = lane & 31, fh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const unsigned aaddr0 = fh * RBN + (wn * (BN / 2) + fi) * ES;
  const unsigned baddr0 = AB + fh * RBC + (wc * (BC / 2) + fi) * ES;

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    issue(s);
    advance();
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);          // retire the scalar argument loads before the loop
  int st_c = 0, st_l = NST - 1;
  for (int ks = 0; ks < nks; ++ks) {
    wait_vmcnt<(NST - 2) * (LA + LB)>();
    __builtin_amdgcn_s_barrier();
    issue(st_l);
    const unsigned sbase = lds0 + st_c * SB;
    unsigned fa[NP][TN], fb[NP][TC];
    load_pair<0, TN, TC, RBN, RBC>(fa[0], fb[0], sbase + aaddr0, sbase + baddr0);
    load_pair<1, TN, TC, RBN, RBC>(fa[1], fb[1], sbase + aaddr0, sbase + baddr0);
    load_pair<2, TN, TC, RBN, RBC>(fa[2], fb[2], sbase + aaddr0, sbase + baddr0);
    PairLoop<0, NP, DEPTH, TN, TC, RBN, RBC>::run(fa, fb, acc, sbase + aaddr0, sbase + baddr0);
    advance();
    st_c = st_c + 1 == NST ? 0 : st_c + 1;
    st_l = st_l + 1 == NST ? 0 : st_l + 1;
  }
  wait_vmcnt<0>();

  // ---- epilogue: D[n][c], column (lane & 31) = in channel, rows (r&3) + 8*(r>>2) + 4*(lane>>5) = out channel
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int c = c0 + wc * (BC / 2) + j * 32 + fi;
    if (c >= p.Creal) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (n < p.Kreal) atomicAdd(p.dw + ((size_t)n * p.ntaps + tap) * p.dw_cpitch + c, acc[i][j][r]);
      }
    }
  }
}

template <int BN, int BC, int BKP, int NST>
int launch_wf(const WFArgs& a, bool gather, int splits_hint, hipStream_t stream) {
  WFArgs p = a;
  p.tiles_n = cdiv(p.Kreal, BN);
  p.ctiles_per_tap = p.Cin / BC;
  p.tiles_c = p.ntaps * p.ctiles_per_tap;
  const int tiles = p.tiles_n * p.tiles_c;
  // pixel splits: enough workgroups to fill the chip twice over; every split costs one fp32 atomic per output element, so
  // never cut the reduction into pieces shorter than 8 steps
  int splits = splits_hint > 0 ? splits_hint : (512 + tiles / 2) / tiles;
  const int max_splits = cdiv(p.M, BKP * 8);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int mps = cdiv(cdiv(p.M, splits), BKP) * BKP;
  splits = cdiv(p.M, mps);
  p.m_per_split = mps;
  p.step_i = BKP / (p.Ho * p.Wo);
  p.step_q = (BKP % (p.Ho * p.Wo)) / p.Wo;
  p.step_r = (BKP % (p.Ho * p.Wo)) % p.Wo;
  constexpr size_t smem = (size_t)NST * BKP * (BN + BC) * 4;
  static_assert(smem <= 156 * 1024, "LDS");
  static bool raised[2] = {false, false};
  if (!raised[gather ? 1 : 0]) {
    if (gather) PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_f32_kernel<BN, BC, BKP, NST, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    else PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_f32_kernel<BN, BC, BKP, NST, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    raised[gather ? 1 : 0] = true;
  }
  if (gather)
    hipLaunchKernelGGL((conv_wgrad_dma_f32_kernel<BN, BC, BKP, NST, true>), dim3(tiles, splits), dim3(256), smem, stream, p);
  else
    hipLaunchKernelGGL((conv_wgrad_dma_f32_kernel<BN, BC, BKP, NST, false>), dim3(tiles, splits), dim3(256), smem, stream, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

}  // namespace

// fp32 leg of pxl_conv_wgrad_dma (conv_wgrad_dma.hip dispatches here).  tile_cfg (the bf16 numbers): 8 / 9 = 128x128 tile
// with a 3 / 2-stage ring of 32-pixel steps, 10 / 11 = 64x64 with 3 / 2 stages of 64-pixel steps, 12 = 128(out) x 64(in),
// 13 = 64 x 128 (3 stages of 32 pixels)
int pxl_conv_wgrad_dma_f32(const pxl_conv_desc* d, const void* in, const void* dy, float* dw, int creal, int dw_cpitch,
                           void* stream) {
  WFArgs a;
  a.in = in; a.dy = dy; a.dw = dw;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.Kreal = d->Kreal; a.Creal = creal; a.dw_cpitch = dw_cpitch;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.M = d->B * d->Ho * d->Wo; a.m_per_split = 0;
  a.tiles_n = a.tiles_c = a.ctiles_per_tap = 0;
  a.step_i = a.step_q = a.step_r = 0;
  a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->Cin * 4);
  a.dy_bytes = (unsigned)((size_t)a.M * d->Cout * 4);
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool gather = d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0 || d->out_stride != 1 || d->Hi != d->Ho || d->Wi != d->Wo;
  int cfg = d->tile_cfg;
  if (cfg < 8) {
    const long t128 = (long)cdiv(a.Kreal, 128) * d->ntaps * (d->Cin / 128);
    cfg = t128 >= 64 ? 9 : 11;
  }
  if (d->Cin % 128 != 0 && (cfg == 8 || cfg == 9 || cfg == 13)) cfg = 11;    // 128-channel column tiles need Cin % 128 == 0
  const int hint = d->split_k > 0 ? d->split_k : 0;     // 1: one add per element of dw (PXL_DETERMINISTIC)
  switch (cfg) {
    case 8: return launch_wf<128, 128, 32, 3>(a, gather, hint, s);
    case 9: return launch_wf<128, 128, 32, 2>(a, gather, hint, s);
    case 10: return launch_wf<64, 64, 64, 3>(a, gather, hint, s);
    case 11: return launch_wf<64, 64, 64, 2>(a, gather, hint, s);
    case 12: return launch_wf<128, 64, 32, 3>(a, gather, hint, s);
    case 13: return launch_wf<64, 128, 32, 3>(a, gather, hint, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_wgrad_dma (fp32): unknown tile config %d", cfg);
  }
}
