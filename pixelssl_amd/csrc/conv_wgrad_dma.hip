// LDS-DMA weight gradient for gfx950 (bf16 operands, fp32 accumulate into the master-gradient layout).
//
//   dW[n][t][c] += sum_m dY[m][n] * Z[pix(m,t)][c]          Z = plain (already activated) input, NHWC
//
// GEMM view: rows = out channels n, columns = (tap t, in channel c), reduction = output pixels m, split
// over blockIdx.y and combined with fp32 atomics.  Both operands arrive pixel-major (the reduction index is
// the slow one).  conv_wgrad.hip transposes 4x8 blocks in registers on the way into LDS; here
//   * the tiles go HBM/L2 -> LDS untouched (`buffer_load_dwordx4 ... lds`, rows = pixels), rows past the
//     end of the tensor and zero-padding taps are out-of-range lanes that the DMA zero-fills;
//   * the MFMA fragments (8 reduction elements of ONE channel per lane) are read with the gfx950 transposing
//     LDS load `ds_read_b64_tr_b16` (lane map probed in tools/probes/probe_tr.hip: within a 16-lane group
//     lane j receives 16-bit slot j&3 of lanes (j>>2)+4e, e = 0..3), two reads per 32x16 fragment;
//   * the 16-byte chunk index is XOR-swizzled with the pixel row so that the 4 pixel rows touched by one
//     transposing read hit distinct banks (source-side swizzle, LDS image stays lane-linear).
// Pipeline: conv_dma.hip's (NST-stage ring, counted vmcnt, raw s_barrier, asm fragment reads).
#include <cstdlib>
#include "common.h"

namespace {

struct WDmaArgs {
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds, 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// see conv_dma.hip: the wait is threaded through the fragments of its k-chunk and the accumulators
template <int N, int TN, int TC>
__device__ __forceinline__ void wait_chunk(u32x4 (&a)[TN], u32x4 (&b)[TC], f32x16 (&acc)[TN][TC]) {
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

// row-dependent XOR of the 16-byte chunk index: RB = bytes per tile row
template <int RB> __device__ __forceinline__ int row_swz(int row) {
  if constexpr (RB == 256) return 4 * (row & 3);
  else return 4 * ((row >> 1) & 1);
}

// tile: BN out channels x BC in channels (one tap), reduction step BKP = 64 pixels, 4 waves as 2 x 2
template <int BN, int BC, int NST, bool GATHER>
__global__ __launch_bounds__(256) void conv_wgrad_dma_kernel(const WDmaArgs p) {
  constexpr int BKP = 64;
  constexpr int RBN = BN * 2, RBC = BC * 2;            // row bytes of the dY / Z tiles
  constexpr int AB = BKP * RBN, BB = BKP * RBC;         // tile bytes
  constexpr int SB = AB + BB;
  constexpr int LA = AB / 4096, LB = BB / 4096;         // DMA instructions per wave per step
  constexpr int RPA = 1024 / RBN, RPB = 1024 / RBC;     // pixel rows per DMA instruction
  static_assert((BN == 128 || BN == 64) && (BC == 128 || BC == 64), "tile");
  constexpr int TN = BN / 64, TC = BC / 64;             // 32-wide MFMA tiles per wave (2 x 2 waves)
  constexpr int RPK = 2 * (TN + TC);                    // transposing reads per k-chunk

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  kernarg_touch<sizeof(WDmaArgs)>();        // every argument line in one scalar-cache round trip (common.h)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;

  // workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest): remap the WHOLE 2-D grid so that one
  // XCD (one L2) runs consecutive (split, tile) pairs = every output tile of the same pixel slice; the slice's dY and Z
  // rows (m_per_split x (Cout + Cin) x 2 B, 1-3 MB) are then fetched from HBM once per XCD instead of once per tile
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
    const int slot = lane % (RBN / 16);
    const int chunk = slot ^ row_swz<RBN>(row);
    const int n = n0 + chunk * 8;
    voffA[q] = n < p.Cout ? (unsigned)(((m_begin + row) * p.Cout + n) * 2) : OOB;
  }
  // ---- Z loader: per DMA row the INPUT coordinates of tap (0,0) (iy0 = oy*stride, ix0 = ox*stride) and the byte
  // offset of that pixel; advancing by BKP output pixels is add + two conditional wraps, no multiplies
  const int tp = p.taps[tap];
  const int tdy = tp >> 16, tdx = (int)(short)(tp & 0xffff);
  int ziy[LB], zix[LB], zoff[LB];
  unsigned voffB[LB];
  const int HoWo = p.Ho * p.Wo;
  const int tapoff = (tdy * p.Wi + tdx) * p.Cin * 2;
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
    const int slot = lane % (RBC / 16);
    const int chunk = slot ^ row_swz<RBC>(row);
    const int m = m_begin + row;
    const int b = m / HoWo;
    const int r = m - b * HoWo;
    const int oy = r / p.Wo;
    ziy[q] = oy * p.so;
    zix[q] = (r - oy * p.Wo) * p.so;
    // pixels past M land in image B (or later): past the end of the tensor => zero-filled by the descriptor
    zoff[q] = ((b * p.Hi + ziy[q]) * p.Wi + zix[q]) * p.Cin * 2 + (c0 + chunk * 8) * 2;
    voffB[q] = z_voff(q);
  }
  int adv_off = (p.step_i * p.Hi * p.Wi + p.step_q * p.so * p.Wi + p.step_r * p.so) * p.Cin * 2;   // +BKP pixels, no wrap
  int adv_y = p.step_q * p.so, adv_x = p.step_r * p.so;
  int wrap_x = (p.so * p.Wi - wos) * p.Cin * 2;                           // extra when ox wraps
  int wrap_y = (p.Hi - hos) * p.Wi * p.Cin * 2;                           // extra when oy wraps
  unsigned sdy = 0;                      // dY soffset
  unsigned dy_step = (unsigned)(BKP * p.Cout * 2);
  int zstep = BKP * p.Cin * 2;
  // Pin every scalar the K loop uses in an SGPR here.  Otherwise hipcc re-loads kernel arguments right before
  // the loop and, not seeing the asm waits, puts its own `s_waitcnt lgkmcnt(0)` at their first use INSIDE the
  // loop, which drains the 16 transposing reads in flight every step.
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
  // advance the Z rows by BKP output pixels (placed AFTER the MFMAs of a step so that it overlaps them)
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

  // ---- fragment addresses (per operand, per 32-channel tile): lane = 32g + 16h + i
  const int fi = lane & 15, fh = (lane >> 4) & 1, fg = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  unsigned aaddr[TN], baddr[TC];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int cb = (wn * (BN / 2) + 32 * t + 16 * fh) / 8;            // chunk index of the 16-channel group
    const int chunk = (cb ^ row_swz<RBN>(fi >> 2)) + ((fi >> 1) & 1);
    aaddr[t] = (8 * fg + (fi >> 2)) * RBN + chunk * 16 + (fi & 1) * 8;
  }
#pragma unroll
  for (int t = 0; t < TC; ++t) {
    const int cb = (wc * (BC / 2) + 32 * t + 16 * fh) / 8;
    const int chunk = (cb ^ row_swz<RBC>(fi >> 2)) + ((fi >> 1) & 1);
    baddr[t] = AB + (8 * fg + (fi >> 2)) * RBC + chunk * 16 + (fi & 1) * 8;
  }

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) {
    issue(s);
    advance();
  }
  // retire every scalar (kernel-argument) load the compiler still counts as outstanding: its own
  // `s_waitcnt lgkmcnt(0)` at the first use would otherwise land inside the loop and drain the LDS reads
  __builtin_amdgcn_s_waitcnt(0xc07f);
  int st_c = 0, st_l = NST - 1;
  for (int ks = 0; ks < nks; ++ks) {
    wait_vmcnt<(NST - 2) * (LA + LB)>();
    __builtin_amdgcn_s_barrier();
    issue(st_l);
    const unsigned sbase = lds0 + st_c * SB;
    u32x4 fa[4][TN], fb[4][TC];
#define PXL_TR_PAIR(dst, addr, RB, KK)                                                      \
  {                                                                                         \
    const u32x2 lo = lds_read_tr<(16 * (KK)) * (RB)>(addr);                                  \
    const u32x2 hi = lds_read_tr<(16 * (KK) + 4) * (RB)>(addr);                              \
    dst = u32x4{lo.x, lo.y, hi.x, hi.y};                                                    \
  }
#define PXL_TR_CHUNK(KK)                                                       \
  PXL_TR_PAIR(fa[KK][0], sbase + aaddr[0], RBN, KK)                            \
  if constexpr (TN == 2) PXL_TR_PAIR(fa[KK][TN - 1], sbase + aaddr[TN - 1], RBN, KK) \
  PXL_TR_PAIR(fb[KK][0], sbase + baddr[0], RBC, KK)                            \
  if constexpr (TC == 2) PXL_TR_PAIR(fb[KK][TC - 1], sbase + baddr[TC - 1], RBC, KK)
#define PXL_MFMA_CHUNK(KK)                                                                                   \
  _Pragma("unroll") for (int i = 0; i < TN; ++i) _Pragma("unroll") for (int j = 0; j < TC; ++j)              \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[KK][i]),             \
                                                          __builtin_bit_cast(bf16x8, fb[KK][j]), acc[i][j], 0, 0, 0);
    // lgkmcnt is a 4-bit counter: at most two k-chunks (<= 16 transposing reads) are kept in flight
    PXL_TR_CHUNK(0)
    PXL_TR_CHUNK(1)
    wait_chunk<RPK>(fa[0], fb[0], acc);
    PXL_MFMA_CHUNK(0)
    PXL_TR_CHUNK(2)
    wait_chunk<RPK>(fa[1], fb[1], acc);
    PXL_MFMA_CHUNK(1)
    PXL_TR_CHUNK(3)
    wait_chunk<RPK>(fa[2], fb[2], acc);
    PXL_MFMA_CHUNK(2)
    wait_chunk<0>(fa[3], fb[3], acc);
    PXL_MFMA_CHUNK(3)
#undef PXL_MFMA_CHUNK
#undef PXL_TR_CHUNK
#undef PXL_TR_PAIR
    advance();
    st_c = st_c + 1 == NST ? 0 : st_c + 1;
    st_l = st_l + 1 == NST ? 0 : st_l + 1;
  }
  wait_vmcnt<0>();

  // ---- epilogue: D[n][c], column (lane & 31) = in channel, rows (r&3) + 8*(r>>2) + 4*(lane>>5) = out channel
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int c = c0 + wc * (BC / 2) + j * 32 + frow;
    if (c >= p.Creal) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (n < p.Kreal) atomicAdd(p.dw + ((size_t)n * p.ntaps + tap) * p.dw_cpitch + c, acc[i][j][r]);
      }
    }
  }
}

template <int BN, int BC, int NST>
int launch_wdma(const WDmaArgs& a, bool gather, int splits_hint, hipStream_t stream) {
  WDmaArgs p = a;
  constexpr int BKP = 64;
  p.tiles_n = cdiv(p.Kreal, BN);
  p.ctiles_per_tap = p.Cin / BC;
  p.tiles_c = p.ntaps * p.ctiles_per_tap;
  const int tiles = p.tiles_n * p.tiles_c;
  // pixel splits: enough blocks to fill the chip twice over, but every split costs one fp32 atomic per output
  // element, so never cut the reduction into pieces shorter than 8 steps
  int splits = splits_hint > 0 ? splits_hint : (512 + tiles / 2) / tiles;
  const int max_splits = cdiv(p.M, BKP * 8);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = cdiv(cdiv(p.M, splits), BKP) * BKP;
  splits = cdiv(p.M, mps);
  p.m_per_split = mps;
  p.step_i = BKP / (p.Ho * p.Wo);
  p.step_q = (BKP % (p.Ho * p.Wo)) / p.Wo;
  p.step_r = (BKP % (p.Ho * p.Wo)) % p.Wo;
  constexpr size_t smem = (size_t)NST * BKP * (BN + BC) * 2;
  static bool raised[2] = {false, false};
  if (!raised[gather ? 1 : 0]) {
    if (gather) PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<BN, BC, NST, true>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_kernel<BN, BC, NST, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    raised[gather ? 1 : 0] = true;
  }
  if (gather)
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<BN, BC, NST, true>), dim3(tiles, splits), dim3(256), smem, stream, p);
  else
    hipLaunchKernelGGL((conv_wgrad_dma_kernel<BN, BC, NST, false>), dim3(tiles, splits), dim3(256), smem, stream, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

}  // namespace

int pxl_conv_wgrad_dma_f32(const pxl_conv_desc* d, const void* in, const void* dy, float* dw, int creal, int dw_cpitch,
                           void* stream);      // conv_wgrad_dma_f32.hip

extern "C" int pxl_conv_wgrad_dma_eligible(const pxl_conv_desc* d, const float* in_scale) {
  if ((d->dtype != PXL_BF16 && d->dtype != PXL_F32) || in_scale != nullptr) return 0;
  // fp32 operands: conv_wgrad_dma_f32.hip (PXL_F32_DMA=0 keeps the fp32 engine on the generic kernels, for A/B runs)
  static const bool f32_on = getenv("PXL_F32_DMA") == nullptr || getenv("PXL_F32_DMA")[0] != '0';
  if (d->dtype == PXL_F32 && !f32_on) return 0;
  const long es = d->dtype == PXL_F32 ? 4 : 2;
  if (d->Cin % 64 != 0 || d->Cout % 8 != 0 || d->div != 1) return 0;
  if ((long)d->B * d->Hi * d->Wi * d->Cin * es * 2 >= (1L << 32)) return 0;    // one image of slack for the ragged tail
  if ((long)(d->B * d->Ho * d->Wo + 64) * d->Cout * es >= (1L << 31)) return 0;
  return 1;
}

// tile_cfg 8/9 = 128x128 tile (3/2-stage ring), 10/11 = 64x64, 12 = 128(out) x 64(in), 13 = 64 x 128
extern "C" int pxl_conv_wgrad_dma(const pxl_conv_desc* d, const void* in, const void* dy, float* dw, int creal,
                                  int dw_cpitch, void* stream) {
  if (d->dtype == PXL_F32) return pxl_conv_wgrad_dma_f32(d, in, dy, dw, creal, dw_cpitch, stream);
  WDmaArgs a;
  a.in = in; a.dy = dy; a.dw = dw;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
  a.Kreal = d->Kreal; a.Creal = creal; a.dw_cpitch = dw_cpitch;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.M = d->B * d->Ho * d->Wo; a.m_per_split = 0;
  a.tiles_n = a.tiles_c = a.ctiles_per_tap = 0;
  a.step_i = a.step_q = a.step_r = 0;
  a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->Cin * 2);
  a.dy_bytes = (unsigned)((size_t)a.M * d->Cout * 2);
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  bool gather = d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0 || d->out_stride != 1 || d->Hi != d->Ho || d->Wi != d->Wo;
  int cfg = d->tile_cfg;
  if (cfg < 8) {
    // few output tiles (short reductions dominate the ResNet shapes at B*H*W = 8712): prefer the small tile
    const long t128 = (long)cdiv(a.Kreal, 128) * d->ntaps * (d->Cin / 128);
    cfg = t128 >= 96 ? 8 : 10;
  }
  if (d->Cin % 128 != 0 && (cfg == 8 || cfg == 9 || cfg == 13)) cfg = 10;    // 128-channel column tiles need Cin % 128 == 0
  // d->split_k > 0 forces the number of pixel splits; 1 = every element of dw receives ONE add (bit-reproducible: PXL_DETERMINISTIC)
  const int hint = d->split_k > 0 ? d->split_k : 0;
  switch (cfg) {
    case 8: return launch_wdma<128, 128, 3>(a, gather, hint, s);
    case 9: return launch_wdma<128, 128, 2>(a, gather, hint, s);
    case 10: return launch_wdma<64, 64, 3>(a, gather, hint, s);
    case 11: return launch_wdma<64, 64, 2>(a, gather, hint, s);
    case 12: return launch_wdma<128, 64, 3>(a, gather, hint, s);
    case 13: return launch_wdma<64, 128, 3>(a, gather, hint, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_wgrad_dma: unknown tile config %d", cfg);
  }
}
