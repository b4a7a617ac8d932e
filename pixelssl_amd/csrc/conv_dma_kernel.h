// LDS-DMA implicit-GEMM convolution for gfx950 (bf16, NHWC, channel pitch % 64 == 0).
//
//   out[m][n] = sum_{t,c} in[pix(m,t)][c] * w[n][t][c]        m = (b,oy,ox), n = out channel
//
// This is the fast path behind pxl_conv_igemm for plain (already activated) bf16 inputs: forward
// 1x1 / 3x3 / atrous / strided convolutions and the data gradients of stride-1 convolutions.
// Differences to conv_igemm.hip (the generic kernel: operands that are not plain, channel pitches that are not multiples of 64;
// fp32 operands have their own build of this pipeline in conv_dma_f32.hip):
//   * tiles go HBM/L2 -> LDS directly (`buffer_load_dwordx4 ... lds`), 1 KiB per wave-instruction,
//     no register staging and no per-element VALU in the K loop.  Zero padding, ragged M and padded
//     output channels are lanes whose buffer offset is out of range: the buffer descriptor makes
//     the DMA write zeros (probed on MI355X: tools/probes/probe_tr.hip).
//   * a K step is 64 channels (128-byte rows) of ONE tap; the per-lane gather offset only changes
//     at a tap boundary, the walk inside a tap is the scalar soffset.
//   * NST-deep LDS ring, one raw s_barrier per K step, counted `s_waitcnt vmcnt(N)` so that NST-2
//     tiles stay in flight across the barrier (a __syncthreads() would drain them).
//   * 16-byte XOR swizzle applied on the SOURCE side (lane -> chunk) and on the fragment read, LDS
//     image stays lane-linear as the DMA requires.
//   * MFMA roles are swapped (A = weights, B = activations) so that a lane's accumulator quads are
//     4 consecutive output channels: the epilogue packs them, stages the tile through LDS
//     (ds_write_b64) and stores full 16-byte row segments; bias / addend / BN statistics are applied
//     on that coalesced read-back pass.
#pragma once
#include <cstdlib>
#include "common.h"

namespace pxl_dma {

struct DmaArgs {
  const void* in;
  const void* w;
  void* out;
  const float* bias;
  const void* addend;
  float* stats;
  int stats_rep;
  // data-gradient launches: fuse the BatchNorm-backward reduction of the tensor being written.  With bn_y set, `stats`
  // ([2*Kreal], one replica) receives sum(gd) and sum(gd * xhat), gd = dz * (relu ? bn(y) > 0 : 1), instead of the
  // forward statistics (sum, sum of squares)
  const void* bn_y;
  const float* bn_coef;
  int bn_relu;
  // ... of a residual join: the tensor being written is d(join output); gd = din * (bn_mask > 0) (bn_mask = the join's
  // post-ReLU output), the MASKED gradient is what gets stored, bn_y / bn_coef belong to the main branch's last BN
  const void* bn_mask;
  int mask_bits;     // bn_mask is the join's ReLU mask as one BYTE per 8 channels ([M][Cout / 8], pxl_residual_fwd_bits), not the tensor
  int B, Hi, Wi, Cin;
  int Ho, Wo, Cout, Kreal;
  int ntaps, so;
  int div_shift;     // data gradient of a stride-2 convolution: source pixel = (oy + dy, ox + dx) / 2 where both are even
  int M, Ktot, nk;
  int tiles_m, tiles_n;
  float* ws;         // split-K: pre-zeroed fp32 [M][Cout] accumulation buffer (blockIdx.y = K slice), else nullptr
  int nk_per;        // K steps per slice
  // split-K by CHANNEL slice (multi-tap gather launches): slice blockIdx.y walks EVERY tap over input channels
  // [kc_per * y, kc_per * (y + 1)) (bytes of the pixel's channel vector, a multiple of 128) instead of a contiguous run of
  // (tap, channel) steps.  A slice of taps re-reads the whole input once per tap from HBM -- the 36-tap ASPP forward moved
  // 673 MB per launch for 36 MB of input (profiles/r05_traffic.json) --; a channel slice reads its 1/S of every pixel's
  // channels 36 times, from the L2 of the XCD that holds the image.  0: contiguous K slices.
  unsigned kc_per;
  // split-K into SLABS: slice blockIdx.y stores its fp32 partial tile at ws + y * ws_slab (elements; 0: one shared buffer that the
  // slices add into with atomics).  The uncoalesced atomics of the shared buffer ran at ~20 per ns (tools/cbench --splitk: 130 us
  // for an 8712 x 256 tile grid); slabs are plain coalesced stores, need no memset and sum in a fixed order
  size_t ws_slab;
  // pxl_conv_dma_slabs: the caller wants the fp32 partial-sum slabs THEMSELVES (ws[slices][M][Cout]; one slab when K is not split)
  // -- no finish kernel, nothing written to `out`; it sums them in a pass of its own (the multi-rate head's col2im, aspp.hip)
  int raw_slabs;
  // forward with batch statistics: the LAST workgroup to finish turns the completed [sum, sumsq] into the BatchNorm
  // coefficients (what pxl_bn_finalize does), so no finalize launch and no replica reduction in the consumers
  pxl_bn_fin fin;    // fin.coef == nullptr: off
  unsigned* fin_counter;
  // BNIN kernels: the A operand is the RAW output y of the previous convolution and relu?(bn(y)) is applied to the tile
  // after it has landed in LDS (no materialised activation tensor, no pxl_bn_apply_fwd launch); `bin` describes that
  // BatchNorm -- every workgroup derives (scale, shift) of all Cin channels from its statistics in the prologue, workgroup 0
  // also writes bin.coef and updates the running statistics (what pxl_bn_finalize does)
  pxl_bn_fin bin;
  int bin_relu;
  void* bin_z;       // optional: the workgroups of output-channel tile 0 also write the activated tile to this tensor (the
                     // weight gradient of this convolution reads it); only for convolutions without a gather (1x1, stride 1)
  unsigned in_bytes, w_bytes;
  // PAIRED launch (gridDim.z == 2): the same convolution of a SECOND network with the same geometry -- student || teacher
  // of Mean Teacher, the l || r task models of GCT -- as workgroups blockIdx.z == 1 of ONE launch: twice the tiles per launch
  // (M = 8712 per network leaves 256 CUs with 138 - 274 tiles), half the launches.  Forward launches only (no addend /
  // BatchNorm-backward operands, no split-K, no in-kernel finalize).
  struct Second {
    const void* in; const void* w; void* out; const float* bias; float* stats;
    const float* bin_stats; const float* bin_gamma; const float* bin_beta; float* bin_rmean; float* bin_rvar; float* bin_coef;
    void* bin_z;
  } g1;
  unsigned* trace;   // TRACE kernels (tools/cbench): [workgroup][TRACE_WORDS] cycle stamps of wave 0, else unused
  // Output sub-grid (data gradient of a stride-2 convolution, one launch per output-parity class): row m of this launch is
  // pixel (b, oy, ox) of a [B][Ho][Wo] SUB-grid and stands for pixel (oy * sub_mul + sub_py, ox * sub_mul + sub_px) of the
  // [B][out_H][out_W] tensor -- for the gather (which taps of which source pixel) and for every row address of the read-back
  // pass (output, addend, BN input, join mask).  sub_mul == 1: off.
  int sub_mul, sub_py, sub_px, out_H, out_W;
  int taps[64];      // (weight tap index << 24) | ((dy & 0xfff) << 12) | (dx & 0xfff): a launch may walk a SUBSET of the taps
};
__host__ __device__ inline int tap_encode(int wt, int dy, int dx) { return (int)(((unsigned)wt << 24) | (((unsigned)dy & 0xfffu) << 12) | ((unsigned)dx & 0xfffu)); }
__host__ __device__ __forceinline__ int tap_dy(int tp) { return (tp << 8) >> 20; }
__host__ __device__ __forceinline__ int tap_dx(int tp) { return (tp << 20) >> 20; }
__host__ __device__ __forceinline__ unsigned tap_wt(int tp) { return (unsigned)tp >> 24; }

// timeline probe (tools/cbench.cpp; never on the product path): per workgroup, words 0..63 = s_memtime stamps of wave 0
// (0 entry, 1 prologue issued, 2 + k = end of K step k (first 52), then loop drained / tile staged / pass 0 / pass 1 / stores
// issued / statistics parked + barrier / atomics issued / everything acknowledged),
// 64 = number of stamps, 65 = HW_ID, 66 = XCC_ID, 67/68 = s_memrealtime (100 MHz) at entry, 69/70 at exit, 71 = nk
constexpr int TRACE_WORDS = 72;

constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

// cache policy of the tile DMAs (the builtin's aux immediate: 1 = sc0, 2 = nt, 16 = sc1): compile-time constants, 0 in the product
// build; tools/build_alt.sh builds variants of the library with other values (DMA_AUX_A=2 ...) for A/B runs
constexpr int DMA_AUX_A = 0;
constexpr int DMA_AUX_W = 0;
template <int AUX = 0>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds, 16, (int)voff, (int)soff, 0, AUX);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Fragment reads are inline asm on purpose: for a C++ LDS load hipcc's waitcnt pass assumes it may alias
// every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of the first ds_read of each K step, which
// drains the ring.  The ordering that is actually needed (this wave's counted vmcnt + the barrier) is
// written out in the loop below.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// `s_waitcnt lgkmcnt(N)` with the fragments of ONE k-chunk and the accumulators threaded through it.  To the
// compiler an MFMA is pure register code, which it schedules freely around a bare wait statement (a "memory"
// clobber does not order it).  "+v"(fragments): the MFMAs of this k-chunk cannot move above the wait that makes
// their operands valid.  "+a"(accumulators): the MFMAs of the previous k-chunk cannot sink below it, so they
// overlap the LDS reads that are still outstanding.  The statement touches none of these registers.
template <int N, int TMI, int TNI>
__device__ __forceinline__ void wait_chunk(u32x4 (&fa)[TMI], u32x4 (&fw)[TNI], f32x16 (&acc)[TNI][TMI]) {
  if constexpr (TMI == 1 && TNI == 1)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0]), "+v"(fw[0]), "+a"(acc[0][0]) : "n"(N));
  else if constexpr (TMI == 2 && TNI == 1)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fw[0]), "+a"(acc[0][0]), "+a"(acc[0][1]) : "n"(N));
  else if constexpr (TMI == 1 && TNI == 2)
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(fa[0]), "+v"(fw[0]), "+v"(fw[1]), "+a"(acc[0][0]), "+a"(acc[1][0]) : "n"(N));
  else if constexpr (TMI == 2 && TNI == 2)
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fw[0]), "+v"(fw[1]), "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]),
                   "+a"(acc[1][1])
                 : "n"(N));
  // tall tiles (one wave column: WM = 1, WN = 4): TNI = 1, TMI = 3 .. 6 pixel tiles per wave
  else if constexpr (TMI == 3 && TNI == 1)
    asm volatile("s_waitcnt lgkmcnt(%7)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fw[0]), "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2])
                 : "n"(N));
  else if constexpr (TMI == 4 && TNI == 1)
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fw[0]), "+a"(acc[0][0]), "+a"(acc[0][1]),
                   "+a"(acc[0][2]), "+a"(acc[0][3])
                 : "n"(N));
  else if constexpr (TMI == 5 && TNI == 1)
    asm volatile("s_waitcnt lgkmcnt(%11)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fw[0]), "+a"(acc[0][0]),
                   "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4])
                 : "n"(N));
  else {
    static_assert(TMI == 6 && TNI == 1, "wait_chunk: unsupported wave tile");
    asm volatile("s_waitcnt lgkmcnt(%13)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fw[0]), "+a"(acc[0][0]),
                   "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5])
                 : "n"(N));
  }
}
template <int OFF> __device__ __forceinline__ void lds_write128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// BN-on-load: the LA pieces a lane has DMA'd itself + the 4 coefficient vectors, all LDS reads waited for at once
template <int LA> __device__ __forceinline__ void wait_xform(u32x4 (&d)[LA], u32x4 (&c)[4]) {
  if constexpr (LA == 1)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else if constexpr (LA == 2)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else if constexpr (LA == 3)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else if constexpr (LA == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else if constexpr (LA == 5)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else {
    static_assert(LA == 6, "wait_xform: unsupported tile height");
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]),
                   "+v"(c[3]));
  }
}
template <int I, int N, int STRIDE> struct XformLoad {
  static __device__ __forceinline__ void run(u32x4 (&d)[N], unsigned addr) {
    d[I] = lds_read128<I * STRIDE>(addr);
    if constexpr (I + 1 < N) XformLoad<I + 1, N, STRIDE>::run(d, addr);
  }
};
template <int I, int N, int STRIDE> struct XformStore {
  static __device__ __forceinline__ void run(const u32x4 (&d)[N], unsigned addr) {
    lds_write128<I * STRIDE>(addr, d[I]);
    if constexpr (I + 1 < N) XformStore<I + 1, N, STRIDE>::run(d, addr);
  }
};

template <int I, int N, int STRIDE, int BASE> struct FragLoad {
  static __device__ __forceinline__ void run(u32x4 (&f)[N], unsigned addr) {
    f[I] = lds_read128<BASE + I * STRIDE>(addr);
    if constexpr (I + 1 < N) FragLoad<I + 1, N, STRIDE, BASE>::run(f, addr);
  }
};

// ---- epilogue read-back passes ------------------------------------------------------------------------------------
struct EpiCtx {
  const unsigned char* T;                 // staged bf16 tile, row pitch TP
  int er, ec, m0, M, Cout, n; bool ncol;
  __amdgpu_buffer_rsrc_t r_out, r_add, r_bny, r_msk;
  const float* bias; const float* bn_coef; int Kreal;
  int sub_mul, sub_py, sub_px, sub_hw, sub_w, out_H, out_W;      // output sub-grid (DmaArgs): generic (EM < 0) read-back only
  // VIRT (conv_halo_kernel.h): row m is position (b, yy, xx) of a PADDED [B][vH + d][vW + d] grid; positions with yy >= vH or
  // xx >= vW are padding -- computed, never stored, and kept out of every sum
  int vH, vW, vHpWp, vWp; float vinv_hpwp, vinv_wp;
};
// Two rules, both from tools/cbench --trace (3.1-5.6 us of a 12 us workgroup sat in the round-3 loop):
//  * straight-line code: out-of-tile rows / padded channel chunks are out-of-range BUFFER offsets (loads return 0, stores
//    are dropped) and the operand combination EM is a compile-time constant -- with one wave per SIMD every instruction of
//    this tail costs ~5 cycles and a scalar branch ~20, and the run-time-flag version executed ~1000 of them per workgroup;
//  * every load of a group of <= 4 passes is issued before the group's first store.  gfx950 counts loads and stores on ONE
//    counter (vmcnt) and hipcc treats a counter with both kinds pending as unordered: a load waited for after a store is
//    waited for with `vmcnt(0)`, i.e. together with the store's acknowledgement (a full round trip per pass).
template <int EM, int NPASS, int RPP, int TP, bool VIRT = false, typename STAMP>
__device__ __forceinline__ void epi_passes(STAMP&& stamp, const EpiCtx& c, float (&s1)[8], float (&s2)[8], int rt_mode = 0) {
  const int em = EM >= 0 ? EM : rt_mode;
  const bool has_add = em & 1, has_bias = em & 2, has_stats = em & 4, has_bnr = em & 8, has_mask = em & 16, bn_relu = em & 32;
  const bool mask_is_bits = em & 64;       // the join mask arrives as one byte per 8-channel chunk (1/16 of the tensor's bytes)
  // per-channel operands of THIS combination only (loaded here, not ahead of the dispatch: 40 registers held across the
  // switch were what pushed the 64 x 128 kernels over the 3-workgroups-per-CU register budget)
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bn_mean[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bn_rstd[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
        bn_sc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bn_sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (has_bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = c.n + e < c.Kreal ? c.bias[c.n + e] : 0.f;
  }
  if (has_bnr && c.ncol) {
    load_cvec<8>(c.bn_coef + c.n, bn_mean);
    load_cvec<8>(c.bn_coef + c.Cout + c.n, bn_rstd);
    if (bn_relu) {
      load_cvec<8>(c.bn_coef + 2 * c.Cout + c.n, bn_sc);
      load_cvec<8>(c.bn_coef + 3 * c.Cout + c.n, bn_sh);
    }
  }
  constexpr unsigned EOOB = 0xffffff00u;
  // passes per group: 4 when a pass loads at most one operand, 2 otherwise (12 live VGPRs per pass of a join data gradient:
  // with groups of 4 the whole kernel went from 134 to 180 registers = from 3 to 2 workgroups per CU, and the launches with
  // several workgroups per CU lost 20-29 % -- tools/r04_ab.sh)
  constexpr int NLD = EM < 0 ? 3 : ((EM & 1) ? 1 : 0) + ((EM & 8) ? 1 : 0) + ((EM & 16) ? 1 : 0);
  constexpr int PGW = (NLD == 0 && NPASS <= 4) ? 4 : 2;
  constexpr int PG = NPASS < PGW ? NPASS : PGW;
  // (tall tiles, 6-12 passes: a real loop over the groups, so that the tail does not set the kernel's register allocation)
#pragma unroll 1
  for (int g0 = 0; g0 < NPASS; g0 += PG) {
    unsigned vo[PG];
    u32x4 xa[PG], xy[PG], xm[PG];
    unsigned xb[PG];
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int m = c.m0 + (g0 + q) * RPP + c.er;
      size_t pix = (size_t)m;
      if constexpr (EM < 0) {
        if (c.sub_mul != 1) {               // row of a sub-grid launch -> pixel of the full tensor
          const int b = m / c.sub_hw, r = m - b * c.sub_hw;
          const int oy = r / c.sub_w, ox = r - oy * c.sub_w;
          pix = ((size_t)b * c.out_H + oy * c.sub_mul + c.sub_py) * c.out_W + ox * c.sub_mul + c.sub_px;
        }
      }
      bool real = true;
      if constexpr (VIRT) {
        int b, r, yy, xx;
        fast_divmod(m, c.vHpWp, c.vinv_hpwp, b, r);
        fast_divmod(r, c.vWp, c.vinv_wp, yy, xx);
        real = yy < c.vH && xx < c.vW;
        pix = ((size_t)b * c.vH + yy) * c.vW + xx;
      }
      vo[q] = (g0 + q < NPASS && m < c.M && c.ncol && real) ? (unsigned)((pix * c.Cout + c.n) * 2) : EOOB;
    }
    if (has_add) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xa[q] = __builtin_amdgcn_raw_buffer_load_b128(c.r_add, (int)vo[q], 0, 0);
    }
    if (has_bnr) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xy[q] = __builtin_amdgcn_raw_buffer_load_b128(c.r_bny, (int)vo[q], 0, 0);
    }
    if (has_mask && mask_is_bits) {
      // byte (pix * Cout + n) / 8 = vo / 16; an out-of-tile row stays out of range (EOOB >> 4 is far beyond the bit plane)
#pragma unroll
      for (int q = 0; q < PG; ++q) xb[q] = __builtin_amdgcn_raw_buffer_load_b8(c.r_msk, (int)(vo[q] >> 4), 0, 0);
    } else if (has_mask) {
#pragma unroll
      for (int q = 0; q < PG; ++q) xm[q] = __builtin_amdgcn_raw_buffer_load_b128(c.r_msk, (int)vo[q], 0, 0);
    }
    uint4 tv[PG];
#pragma unroll
    for (int q = 0; q < PG; ++q)
      if (g0 + q < NPASS) tv[q] = *reinterpret_cast<const uint4*>(c.T + ((g0 + q) * RPP + c.er) * TP + c.ec * 16);
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int ps = g0 + q;
      if (ps >= NPASS) break;
      uint4 v = tv[q];
      if constexpr (VIRT) {
        if (vo[q] == EOOB) v = make_uint4(0u, 0u, 0u, 0u);       // a padding position holds a real (meaningless) accumulator
      }
      if (has_bias || has_add || has_stats) {
        float f[8];
        Chunk<bf16_t>::unpack(v, f);
        if (has_bias || has_add) {
          if (has_add) {
            float ad[8];
            Chunk<bf16_t>::unpack(make_uint4(xa[q][0], xa[q][1], xa[q][2], xa[q][3]), ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += ad[e];
          }
          if (has_bias) {
            const bool ok = vo[q] != EOOB;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = ok ? f[e] + bv[e] : 0.f;     // (rows past M stay out of the statistics)
          }
          v = Chunk<bf16_t>::pack(f);
          if (has_stats) Chunk<bf16_t>::unpack(v, f);     // statistics of the stored (rounded) values
        }
        if (has_mask) {
          float fy[8], fm[8];
          Chunk<bf16_t>::unpack(make_uint4(xy[q][0], xy[q][1], xy[q][2], xy[q][3]), fy);
          if (mask_is_bits) {
#pragma unroll
            for (int e = 0; e < 8; ++e) fm[e] = (xb[q] >> e) & 1u ? 1.f : 0.f;
          } else {
            Chunk<bf16_t>::unpack(make_uint4(xm[q][0], xm[q][1], xm[q][2], xm[q][3]), fm);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float gd = fm[e] > 0.f ? f[e] : 0.f;
            f[e] = gd;
            s1[e] += gd;
            s2[e] += gd * (fy[e] - bn_mean[e]) * bn_rstd[e];
          }
          v = Chunk<bf16_t>::pack(f);
        } else if (has_bnr) {
          float fy[8];
          Chunk<bf16_t>::unpack(make_uint4(xy[q][0], xy[q][1], xy[q][2], xy[q][3]), fy);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gd = f[e];
            if (bn_relu && !(fy[e] * bn_sc[e] + bn_sh[e] > 0.f)) gd = 0.f;
            s1[e] += gd;
            s2[e] += gd * (fy[e] - bn_mean[e]) * bn_rstd[e];
          }
        } else if (has_stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += f[e];
            s2[e] += f[e] * f[e];
          }
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, c.r_out, (int)vo[q], 0, 0);
      if (g0 == 0 && q < 2) stamp();          // (TRACE builds: the first two passes; a no-op otherwise)
    }
  }
}

// BM x BN output tile (pixels x channels), 4 or 8 waves as WM x WN, NST LDS stages, GATHER = taps / padding logic.
// Eight waves (two per SIMD) exist for the DMA issue rate: a `buffer_load ... lds` costs its wave 100-180 cycles of issue
// (MI355X_MICROARCH.md), four waves x 6 pieces per K step of a 64 x 128 tile is the ~800 cycles per step that
// tools/cbench --trace measures, and tools/cbench --floor shows a CU reaching ~40 B/clk only with more waves issuing.
// ABL: timing ablations for tools/conv_bench.py (results are garbage): 1 = no DMA in the loop, 2 = no MFMA,
// 4 = no fragment reads, 8 = no barrier.  0 in every product instantiation.
// workgroups (= waves per SIMD) the register allocation must leave room for: small tiles live on co-residency (a CU with 3-4
// workgroups in flight keeps ~38 B/clk of DMA going, one alone ~28: tools/cbench --floor)
constexpr int dma_occupancy(int acc_tiles) { return acc_tiles <= 2 ? 3 : 2; }
// ... as waves per SIMD (the launch-bounds unit): 8-wave workgroups put two waves on every SIMD
constexpr int dma_waves_per_simd(int nw, int acc_tiles) { return nw == 4 ? dma_occupancy(acc_tiles) : (acc_tiles <= 1 ? 4 : 2); }

template <int BM, int BN, int WM, int WN, int NST, bool GATHER, int ABL = 0, bool BNIN = false, bool TRACE = false, int EM = -1>
__global__ __launch_bounds__(WM * WN * 64, dma_waves_per_simd(WM * WN, (BM / WM / 32) * (BN / WN / 32))) void conv_dma_kernel(const DmaArgs p) {
  constexpr int TMI = BM / WM / 32;          // 32-pixel tiles per wave
  constexpr int TNI = BN / WN / 32;          // 32-channel tiles per wave
  constexpr int NW = WM * WN;                // waves: 4, or 8 = two per SIMD (half the DMA instructions per wave and step)
  constexpr int NT = NW * 64;
  constexpr int LA = BM / (8 * NW), LB = BN / (8 * NW);  // DMA instructions per wave per K step
  constexpr int SB = (BM + BN) * 128;        // bytes per stage
  constexpr int TP = BN * 2 + 16;            // epilogue staging row pitch (bank-conflict-free ds_write_b64)
  constexpr int TPR = BN / 8;                // threads per output row on the read-back pass
  constexpr int RPP = NT / TPR;              // rows per pass
  constexpr int NPASS = BM / RPP;
  static_assert((NW == 4 || NW == 8) && LA >= 1 && LB >= 1 && TMI >= 1 && TNI >= 1 && ((TMI <= 2 && TNI <= 2) || (TNI == 1 && TMI <= 6)), "tile");
  static_assert(BM % RPP == 0, "epilogue rows per pass must divide the tile");
  constexpr int RED_BYTES = RPP * 2 * BN * 4;                   // statistics partials [RPP rows][2][BN] fp32
  constexpr bool RED_ALIAS = BM * TP + RED_BYTES > NST * SB;    // no room behind the staged tile: reuse it (one more barrier)
  static_assert(NST * SB >= BM * TP && NST * SB >= RED_BYTES, "epilogue staging / statistics partials must fit in the ring");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  // Pull every 64-byte line of the kernel-argument segment into the scalar cache in the SAME round trip as the compiler's
  // first argument loads.  hipcc fetches DmaArgs in 3-4 dependent batches (tile index -> geometry -> pointers -> taps),
  // each a cold miss of 0.4-0.5 us right after the launch: tools/cbench --trace showed 1.6-2.8 us between kernel entry and
  // the first DMA.
  kernarg_touch<sizeof(DmaArgs)>();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  unsigned tsv = 0;      // TRACE: lane i = stamp i
  int tsi = 0;
  unsigned long long rt0 = 0;
  auto stamp = [&]() {
    if constexpr (TRACE) {
      if (tsi < 64) {
        const unsigned now = (unsigned)__builtin_readcyclecounter();
        tsv = lane == tsi ? now : tsv;
      }
      ++tsi;
    }
  };
  if constexpr (TRACE) rt0 = wall_clock64();
  stamp();

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // operand set of this workgroup's network (paired launches: blockIdx.z == 1 takes DmaArgs::g1; scalar selects)
  const bool second = blockIdx.z != 0;
  const void* const a_in = second ? p.g1.in : p.in;
  const void* const a_w = second ? p.g1.w : p.w;
  void* const a_out = second ? p.g1.out : p.out;
  const float* const a_bias = second ? p.g1.bias : p.bias;
  float* const a_stats = second ? p.g1.stats : p.stats;
  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a_in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a_w), 0, p.w_bytes, 0x00020000);

  // ---- loader coordinates: DMA instruction g = wave + 4*q covers tile rows 8g .. 8g+7, lane -> (row, 16-byte slot)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned voffA[LA], voffB[LB];
  int a_pix[LA], a_iy[LA], a_ix[LA], a_img[LA];
  const int HoWo = p.Ho * p.Wo;
  // (branch-free: one wave per SIMD executes this prologue at ~5 cycles per instruction with nothing to overlap it, and a
  // 1x1 / stride-1 launch -- half of the network's convolutions -- needs no pixel decomposition at all)
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
      a_pix[q] = m * (int)(p.Cin * 2) + chunk * 16;
    } else {
      int b, r, oy, ox;
      fast_divmod(m, HoWo, inv_howo, b, r);            // m < 2^24 (pxl_conv_dma_eligible)
      fast_divmod(r, p.Wo, inv_wo, oy, ox);
      oy = oy * p.sub_mul + p.sub_py;                  // (sub-grid launches: the pixel this row stands for; else * 1 + 0)
      ox = ox * p.sub_mul + p.sub_px;
      a_iy[q] = in ? oy * p.so : -(1 << 20);
      a_ix[q] = in ? ox * p.so : 0;
      a_img[q] = b * p.Hi * p.Wi * p.Cin * 2 + chunk * 16;
      a_pix[q] = a_img[q] + (a_iy[q] * p.Wi + a_ix[q]) * p.Cin * 2;
    }
    voffA[q] = in ? (unsigned)a_pix[q] : OOB;
  }
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int row = (wave + NW * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int n = n0 + row;
    voffB[q] = n < p.Kreal ? (unsigned)(n * p.Ktot * 2 + chunk * 16) : OOB;
  }

  // ---- load cursor: tap t, byte offset kcb inside the pixel's channel vector, kwb inside the weight row
  const unsigned cin_bytes = (unsigned)p.Cin * 2;
  const bool by_chan = GATHER && p.kc_per != 0;                 // (block-uniform)
  const int ks_begin = by_chan ? 0 : blockIdx.y * p.nk_per;     // split-K slice (whole range when gridDim.y == 1)
  const unsigned c_lo = by_chan ? blockIdx.y * p.kc_per : 0u;
  const unsigned c_hi = by_chan ? min(cin_bytes, c_lo + p.kc_per) : cin_bytes;
  const int nk_here = by_chan ? (int)((c_hi - c_lo) >> 7) * (int)(((unsigned)p.nk * 128u) / cin_bytes)
                              : min(p.nk, ks_begin + p.nk_per) - ks_begin;
  int ld_t = by_chan ? 0 : (int)(((unsigned)ks_begin * 128u) / cin_bytes);
  unsigned kcb = by_chan ? c_lo : (unsigned)ks_begin * 128u - (unsigned)ld_t * cin_bytes, kwb = (unsigned)ks_begin * 128u;
  auto set_tap = [&](int t) {
    if constexpr (GATHER) {
      const int tp = p.taps[min(t, p.ntaps - 1)];
      const int dy = tap_dy(tp), dx = tap_dx(tp);
      kwb = tap_wt(tp) * cin_bytes + kcb;            // this tap's slice of the weight row (a launch may walk a subset of the taps)
      const int tapoff = (dy * p.Wi + dx) * p.Cin * 2;
      if (p.div_shift == 0) {
#pragma unroll
        for (int q = 0; q < LA; ++q) {
          const int iy = a_iy[q] + dy, ix = a_ix[q] + dx;
          const bool ok = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
          voffA[q] = ok ? (unsigned)(a_pix[q] + tapoff) : OOB;
        }
      } else {
        // stride-2 data gradient: only the (pixel, tap) pairs whose source coordinate is even exist; the others are
        // out-of-range lanes (zero fill) like padding -- 3/4 of a 3x3 tap set, but the tile still streams by DMA
#pragma unroll
        for (int q = 0; q < LA; ++q) {
          const int ny = a_iy[q] + dy, nx = a_ix[q] + dx;
          const int iy = ny >> 1, ix = nx >> 1;
          const bool ok = ((ny | nx) & 1) == 0 && ny >= 0 && nx >= 0 && iy < p.Hi && ix < p.Wi;
          voffA[q] = ok ? (unsigned)(a_img[q] + (iy * p.Wi + ix) * p.Cin * 2) : OOB;
        }
      }
    }
  };
  unsigned vm = 0;                 // BNIN: 8 bits per ring stage, bit q = piece q of this lane was in range (not zero-filled)
  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * SB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LA; ++q) dma16<DMA_AUX_A>(r_in, sa + q * NW * 1024, voffA[q], kcb);
    if constexpr (BNIN) {
      unsigned bits = 0;
#pragma unroll
      for (int q = 0; q < LA; ++q) bits |= (voffA[q] != OOB ? 1u : 0u) << q;
      vm = (vm & ~(0xffu << (8 * stage))) | (bits << (8 * stage));
    }
    unsigned char* sb = smem + stage * SB + BM * 128 + wave * 1024;
#pragma unroll
    for (int q = 0; q < LB; ++q) dma16<DMA_AUX_W>(r_w, sb + q * NW * 1024, voffB[q], kwb);
    kwb += 128;
    kcb += 128;
    if (kcb == c_hi) {           // block-uniform: next tap (channel-sliced split-K: back to the slice's first channel)
      kcb = c_lo;
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

  // fragment read offsets: row (lane & 31) of a 32-row tile, 16-byte chunk 2*kk + (lane >> 5), swizzled
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
  // ---- prologue: NST-1 tiles in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);

  // ---- BNIN: (scale, shift) of every input channel -> LDS table behind the ring (the tiles of the prologue are in flight)
  unsigned ckc = 0;                                   // channel offset of the tile being consumed
  const unsigned tab0 = lds0 + NST * SB;              // [Cin] scale, [Cin] shift (fp32)
  const int lchunk = (lane & 7) ^ ((wave * 4 + ((lane >> 3) >> 1)) & 7);     // the lane's (q-independent) source chunk
  if constexpr (BNIN) {
    float* tab = reinterpret_cast<float*>(smem + NST * SB);
    pxl_bn_fin f = p.bin;
    if (second) {
      f.stats = p.g1.bin_stats; f.gamma = p.g1.bin_gamma; f.beta = p.g1.bin_beta;
      f.running_mean = p.g1.bin_rmean; f.running_var = p.g1.bin_rvar; f.coef = p.g1.bin_coef;
    }
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
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        f.coef[c] = mean; f.coef[C + c] = rstd; f.coef[2 * C + c] = scale; f.coef[3 * C + c] = shift;
      }
    }
    __syncthreads();
    ckc = ((unsigned)ks_begin * 64u) % (unsigned)C;
  }
  // retire every scalar (kernel-argument) load the compiler still counts as outstanding: its own
  // `s_waitcnt lgkmcnt(0)` at the first use would otherwise land inside the loop and drain the LDS reads
  __builtin_amdgcn_s_waitcnt(0xc07f);
  stamp();
  int st_c = 0;               // stage being multiplied
  int st_l = NST - 1;         // stage being filled
  for (int ks = 0; ks < nk_here; ++ks) {
    wait_vmcnt<(NST - 2) * (LA + LB)>();      // this wave's share of tile ks has landed
    if constexpr (BNIN) {
      // relu?(scale * y + shift) on the pieces THIS lane has DMA'd (visible to the issuing wave after its vmcnt wait, no
      // barrier needed), in place, rounded to bf16 like the materialised tensor was; zero-filled pieces (padding taps,
      // rows past M) stay zero.  All LDS traffic is inline asm: see the note on the fragment reads.
      const unsigned pa = lds0 + st_c * SB + wave * 1024 + lane * 16;
      const unsigned tb = tab0 + (ckc + (unsigned)lchunk * 8u) * 4u;
      const unsigned tb2 = tb + (unsigned)p.Cin * 4u;
      u32x4 cf[4], dd[LA];
      cf[0] = lds_read128<0>(tb);  cf[1] = lds_read128<16>(tb);
      cf[2] = lds_read128<0>(tb2); cf[3] = lds_read128<16>(tb2);
      XformLoad<0, LA, NW * 1024>::run(dd, pa);
      wait_xform<LA>(dd, cf);
      const unsigned bits = (vm >> (8 * st_c)) & 0xffu;
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = __uint_as_float(cf[0][e]); sc[4 + e] = __uint_as_float(cf[1][e]);
        sh[e] = __uint_as_float(cf[2][e]); sh[4 + e] = __uint_as_float(cf[3][e]);
      }
#pragma unroll
      for (int q = 0; q < LA; ++q) {
        float f[8];
        Chunk<bf16_t>::unpack(make_uint4(dd[q][0], dd[q][1], dd[q][2], dd[q][3]), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = f[e] * sc[e] + sh[e];
          f[e] = p.bin_relu ? fmaxf(v, 0.f) : v;
        }
        const uint4 r = Chunk<bf16_t>::pack(f);
        const bool real = (bits >> q) & 1u;
        dd[q] = real ? u32x4{r.x, r.y, r.z, r.w} : dd[q];
      }
      XformStore<0, LA, NW * 1024>::run(dd, pa);
      if constexpr (!GATHER) {
        // materialise z = relu(bn(y)) for the weight gradient: one workgroup per pixel tile writes what it transformed (the
        // same bytes, the same offsets as the source; zero-filled lanes are out of range for the store as well)
        void* const a_z = second ? p.g1.bin_z : p.bin_z;
        if (a_z != nullptr && tn == 0) {
          const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(a_z, 0, p.in_bytes, 0x00020000);
#pragma unroll
          for (int q = 0; q < LA; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(dd[q], r_z, (int)voffA[q], (int)(ckc * 2u), 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ckc += 64;
      if (ckc == (unsigned)p.Cin) ckc = 0;
    }
    if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();   // ... everyone's has; stage st_l is no longer being read
    if constexpr (!(ABL & 1)) issue(st_l);
    // all 4*(TMI+TNI) fragment reads of the step are issued up front (LDS returns in order), the MFMAs of
    // k-chunk kk start as soon as its own reads are back: lgkmcnt counts the reads still outstanding
    const unsigned sbase = lds0 + st_c * SB;
    u32x4 fa[4][TMI], fw[4][TNI];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (ABL & 4) {
#pragma unroll
        for (int i = 0; i < TMI; ++i) fa[kk][i] = u32x4{sbase, sbase, sbase, sbase};
#pragma unroll
        for (int j = 0; j < TNI; ++j) fw[kk][j] = u32x4{sbase, sbase, sbase, sbase};
      } else {
        FragLoad<0, TMI, 4096, 0>::run(fa[kk], sbase + aoff[kk]);
        FragLoad<0, TNI, 4096, BM * 128>::run(fw[kk], sbase + boff[kk]);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // (lgkmcnt is a 4-bit counter on gfx950: a wait for "more than 15 outstanding" is a wait for 15)
      constexpr int PER = TMI + TNI;
      if (kk == 0) wait_chunk<(3 * PER > 15 ? 15 : 3 * PER)>(fa[0], fw[0], acc);
      if (kk == 1) wait_chunk<(2 * PER > 15 ? 15 : 2 * PER)>(fa[1], fw[1], acc);
      if (kk == 2) wait_chunk<(1 * PER > 15 ? 15 : 1 * PER)>(fa[2], fw[2], acc);
      if (kk == 3) wait_chunk<0>(fa[3], fw[3], acc);
      if constexpr (!(ABL & 2)) {
#pragma unroll
        for (int j = 0; j < TNI; ++j)
#pragma unroll
          for (int i = 0; i < TMI; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[kk][j]),
                                                                __builtin_bit_cast(bf16x8, fa[kk][i]), acc[j][i], 0, 0, 0);
      }
    }
    st_c = st_c + 1 == NST ? 0 : st_c + 1;
    st_l = st_l + 1 == NST ? 0 : st_l + 1;
    if constexpr (TRACE) { if (ks < 52) stamp(); }
  }
  wait_vmcnt<0>();                 // the tail DMAs (tiles past nk) must not land in the staging area
  __builtin_amdgcn_s_barrier();
  stamp();

  if constexpr (ABL & 16) return;
  if (EM == -2 || (EM == -1 && p.ws != nullptr)) {
    // split-K: fp32 partial sums of this K slice -> workspace; bias / rounding happen in the finish kernel.
    // The accumulator layout puts a wave's 64 lanes on 32 DIFFERENT rows: an atomic straight from the registers touches 64 cache
    // lines per instruction (tools/cbench --splitk: 130 us for the 2.2 M atomics of an 8712 x 256 slab, 20 per ns).  Staged
    // through LDS as an fp32 tile, a wave adds 4 whole row segments of 16 lanes x 16 bytes per instruction instead.
    constexpr int TPF = BN * 4 + 16;                               // fp32 staging row pitch
    if constexpr ((size_t)BM * TPF <= (size_t)NST * SB) {
      float* T32 = reinterpret_cast<float*>(smem);
#pragma unroll
      for (int j = 0; j < TNI; ++j)
#pragma unroll
        for (int i = 0; i < TMI; ++i) {
          const int ml = (wm * TMI + i) * 32 + frow;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = (wn * TNI + j) * 32 + 8 * g + 4 * fhalf;
            *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(T32) + ml * TPF + nl * 4) =
                make_float4(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
          }
        }
      __syncthreads();
      constexpr int TPR4 = BN / 4;                                 // threads per row (4 floats each)
      constexpr int RPP4 = NT / TPR4;
      const int c4 = tid % TPR4, r4 = tid / TPR4;
      const int n = n0 + 4 * c4;
#pragma unroll 4
      for (int ps = 0; ps < BM / RPP4; ++ps) {
        const int ml = ps * RPP4 + r4, m = m0 + ml;
        if (p.ws_slab != 0) {
          if (m < p.M && n < p.Cout) {
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(T32) + ml * TPF + c4 * 16);
            *reinterpret_cast<float4*>(p.ws + (size_t)blockIdx.y * p.ws_slab + (size_t)m * p.Cout + n) = v;
          }
        } else if (m < p.M && n < p.Kreal) {
          const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(T32) + ml * TPF + c4 * 16);
          float* dst = p.ws + (size_t)m * p.Cout + n;
          atomicAdd(dst, v.x);
          if (n + 1 < p.Kreal) atomicAdd(dst + 1, v.y);
          if (n + 2 < p.Kreal) atomicAdd(dst + 2, v.z);
          if (n + 3 < p.Kreal) atomicAdd(dst + 3, v.w);
        }
      }
    } else {
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
    }
    return;
  }
  // ---- epilogue 1: accumulators -> bf16 tile T[m][n] in LDS.  C/D layout of the 32x32 MFMA with swapped
  // roles: column (lane & 31) = pixel, rows (r&3) + 8*(r>>2) + 4*(lane>>5) = channel
  unsigned char* T = smem;
#pragma unroll
  for (int j = 0; j < TNI; ++j)
#pragma unroll
    for (int i = 0; i < TMI; ++i) {
      const int ml = (wm * TMI + i) * 32 + frow;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = (wn * TNI + j) * 32 + 8 * g + 4 * fhalf;
        uint2 v;
        v.x = pack_bf2(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1]);
        v.y = pack_bf2(acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
        *reinterpret_cast<uint2*>(T + ml * TP + nl * 2) = v;
      }
    }
  __syncthreads();
  stamp();

  // ---- epilogue 2: coalesced read-back, bias / addend / statistics, 16-byte stores
  const int ec = tid % TPR;                  // 8-channel chunk of this thread
  const int er = tid / TPR;
  const int n = n0 + ec * 8;
  const bool ncol = n < p.Cout;
  // EM >= 0: the operand combination is a compile-time constant of this instantiation (the host picks it, launch_dma below).
  // The round-4 timeline probe measured the run-time dispatch between the read-back variants at ~3500 cycles per workgroup
  // (a ten-way scalar branch into cold instruction-cache lines): 4100 -> 660 cycles for the passes once it was gone.
  const bool has_bias = EM >= 0 ? bool(EM & 2) : a_bias != nullptr;
  const bool has_add = EM >= 0 ? bool(EM & 1) : p.addend != nullptr;
  const bool has_stats = EM >= 0 ? bool(EM & 4) : (a_stats != nullptr && !(ABL & 32));
  const bool has_bnr = EM >= 0 ? bool(EM & 8) : (has_stats && p.bn_y != nullptr);
  const bool has_mask = EM >= 0 ? bool(EM & 16) : (has_bnr && p.bn_mask != nullptr);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  // Read-back passes: one straight-line specialisation per operand combination (epi_passes above).  emode bits: 1 addend, 2 bias, 4 statistics, 8 BatchNorm-backward sums, 16 join mask, 32 ReLU mask of the BN.
  {
    EpiCtx c;
    c.T = T; c.er = er; c.ec = ec; c.m0 = m0; c.M = p.M; c.Cout = p.Cout; c.n = n; c.ncol = ncol;
    const unsigned out_bytes = p.sub_mul != 1 ? (unsigned)((size_t)p.B * p.out_H * p.out_W * p.Cout * 2) : (unsigned)((size_t)p.M * p.Cout * 2);
    c.sub_mul = p.sub_mul; c.sub_py = p.sub_py; c.sub_px = p.sub_px; c.sub_hw = HoWo; c.sub_w = p.Wo; c.out_H = p.out_H; c.out_W = p.out_W;
    c.r_out = __builtin_amdgcn_make_buffer_rsrc(a_out, 0, out_bytes, 0x00020000);
    c.r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.addend), 0, out_bytes, 0x00020000);
    c.r_bny = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_y), 0, out_bytes, 0x00020000);
    c.r_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_mask), 0, p.mask_bits ? out_bytes / 16 : out_bytes, 0x00020000);
    c.bias = a_bias; c.bn_coef = p.bn_coef; c.Kreal = p.Kreal;
    const int emode = (has_add ? 1 : 0) | (has_bias ? 2 : 0) | (has_stats ? 4 : 0) | (has_bnr ? 8 : 0) | (has_mask ? 16 : 0) |
                      ((has_bnr && !has_mask && p.bn_relu) ? 32 : 0) | ((has_mask && p.mask_bits) ? 64 : 0);
    if constexpr (EM >= 0) epi_passes<EM, NPASS, RPP, TP>(stamp, c, s1, s2);
    else epi_passes<-1, NPASS, RPP, TP>(stamp, c, s1, s2, emode);           // uncommon combinations: run-time flags
  }
  stamp();
  if (has_stats) {
    // reduce over the RPP threads that share a channel chunk (one per row of a pass), through LDS: every thread parks its
    // 16 partial sums, one thread per (sum, channel) adds the RPP rows.  (The round-3 form -- 32 cross-lane shuffles, each an
    // LDS permute with its own wait, then a 4-wave LDS pass -- took 1.0 us of a 12 us workgroup: tools/cbench --trace.)
    if constexpr (RED_ALIAS) __syncthreads();                  // every thread is done reading the staged tile
    float* red = reinterpret_cast<float*>(smem + (RED_ALIAS ? 0 : BM * TP));     // [RPP rows][2][BN]
    {
      float* mine = red + er * 2 * BN + ec * 8;
      *reinterpret_cast<float4*>(mine) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(mine + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
      *reinterpret_cast<float4*>(mine + BN) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      *reinterpret_cast<float4*>(mine + BN + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
    }
    __syncthreads();
    stamp();
    if (tid < 2 * BN) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) v += red[r * 2 * BN + tid];
      const int which = tid / BN, c = tid % BN;
      if (n0 + c < p.Kreal) {
        float* rep = a_stats + (size_t)(tm % p.stats_rep) * 2 * p.Kreal;
        atomicAdd(rep + which * p.Kreal + n0 + c, v);
      }
    }
  } else {
    stamp();
  }
  stamp();
  if (p.fin.coef != nullptr && has_stats && !has_bnr) {
    // last-block-done.  The statistics are device-scope atomics (performed at the memory side, never cached): a thread's
    // `s_waitcnt vmcnt(0)` means its atomics have been performed, the barrier extends that to the block, and only then
    // does thread 0 draw the ticket.  The block holding the last ticket reads every replica with agent-scope loads (past
    // its L1; no other block of this launch ever read those lines, so no L2 holds them).  No __threadfence(): a full
    // agent fence per block writes back the L2 and made the whole step 1.8x slower when it was tried here.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int mine = 0;
    if (tid == 0) mine = atomicAdd(p.fin_counter, 1u) == (unsigned)(gridDim.x * gridDim.y) - 1u;
    const int is_last = __syncthreads_or(mine);
    if (is_last) {
      const int C = p.Kreal;
      for (int c = tid; c < C; c += NT) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < p.stats_rep; ++r) {
          s1 += __hip_atomic_load(p.stats + (size_t)r * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s2 += __hip_atomic_load(p.stats + (size_t)r * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float mean = s1 / p.fin.count;
        float var = s2 / p.fin.count - mean * mean;
        if (var < 0.f) var = 0.f;
        if (p.fin.running_mean != nullptr) {
          const float unbiased = p.fin.count > 1.f ? var * p.fin.count / (p.fin.count - 1.f) : var;
          p.fin.running_mean[c] = (1.f - p.fin.momentum) * p.fin.running_mean[c] + p.fin.momentum * mean;
          p.fin.running_var[c] = (1.f - p.fin.momentum) * p.fin.running_var[c] + p.fin.momentum * unbiased;
        }
        const float rstd = p.fin.clamp_var ? rsqrtf(fmaxf(var, p.fin.eps)) : rsqrtf(var + p.fin.eps);
        const float g = p.fin.gamma ? p.fin.gamma[c] : 1.f, b = p.fin.beta ? p.fin.beta[c] : 0.f;
        const float scale = g * rstd;
        p.fin.coef[c] = mean;
        p.fin.coef[C + c] = rstd;
        p.fin.coef[2 * C + c] = scale;
        p.fin.coef[3 * C + c] = b - mean * scale;
      }
    }
  }
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile's stores have been performed
    stamp();
    if (wave == 0 && p.trace != nullptr) {
      unsigned* t = p.trace + (size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * TRACE_WORDS;
      t[lane] = tsv;
      if (lane == 0) {
        const unsigned long long rt1 = wall_clock64();
        t[64] = (unsigned)tsi;
        t[65] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        t[66] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        t[67] = (unsigned)rt0; t[68] = (unsigned)(rt0 >> 32);
        t[69] = (unsigned)rt1; t[70] = (unsigned)(rt1 >> 32);
        t[71] = (unsigned)nk_here;
      }
    }
  }
}

template <int BM, int BN, int NST, int ABL>
int launch_abl(const DmaArgs& a, hipStream_t stream) {
  DmaArgs p = a;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  p.ws = nullptr;
  p.ws_slab = 0;
  p.raw_slabs = 0;
  p.nk_per = p.nk;
  p.kc_per = 0;
  p.fin.coef = nullptr;
  p.bin.coef = nullptr; p.bin_z = nullptr; p.trace = nullptr;
  constexpr size_t smem = (size_t)NST * (BM + BN) * 128;
  PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, 2, 2, NST, true, ABL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
  hipLaunchKernelGGL((conv_dma_kernel<BM, BN, 2, 2, NST, true, ABL>), dim3(p.tiles_m * p.tiles_n), dim3(256), smem, stream, p);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_splitk_finish(int dtype, long total, int Cout, int Kreal, const float* ws, const float* bias, void* out,
                                 void* stream);
extern "C" int pxl_splitk_finish_slabs(int dtype, long total, int Cout, int Kreal, int nslab, const float* ws, const float* bias,
                                       void* out, void* stream);

template <int BM, int BN, int WM, int WN, int NST, bool GATHER, bool BNIN, bool TRACE, int EM>
int launch_one(dim3 grid, dim3 block, size_t smem, hipStream_t stream, const DmaArgs& p) {
  static bool raised = false;
  if (!raised) {
    PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, GATHER, 0, BNIN, TRACE, EM>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    raised = true;
  }
  const long t0 = pxlht::on ? pxlht::now() : 0;
  hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, GATHER, 0, BNIN, TRACE, EM>), grid, block, smem, stream, p);
  PXL_LAUNCH_CHECK();
  if (pxlht::on) pxlht::add(16, pxlht::now() - t0);
  return PXL_OK;
}

template <int BM, int BN, int WM, int WN, int NST>
int launch_dma(const DmaArgs& a, bool gather, int want_split, size_t ws_bytes, hipStream_t stream, int groups = 1) {
  DmaArgs p = a;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const int grid = p.tiles_m * p.tiles_n;
  // split-K: launches that cannot fill the chip and have a long reduction (ASPP: 137 tiles, K = 73728)
  int splitk = 1;
  const bool can_split = p.ws != nullptr && p.stats == nullptr && p.addend == nullptr &&
                         ws_bytes >= (size_t)p.M * p.Cout * sizeof(float);
  if (can_split) {
    if (want_split > 1) splitk = want_split;
    else if (want_split <= 0 && grid < 200 && p.nk >= 64) splitk = min(cdiv(768, grid), p.nk / 16);
    if (splitk > p.nk) splitk = p.nk;
    if (splitk < 1) splitk = 1;
  }
  if (p.raw_slabs) {
    // exactly the slice count that was asked for, each slice a slab (the caller sized the workspace for it)
    if (p.ws == nullptr || p.stats != nullptr || p.addend != nullptr || want_split < 1 || p.nk % want_split != 0 ||
        ws_bytes < (size_t)want_split * p.M * p.Cout * sizeof(float) || p.Cout % 4 != 0 || groups != 1 || p.bin.coef != nullptr)
      return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_slabs: %d slices of %d K steps into %zu workspace bytes", want_split, p.nk, ws_bytes);
    splitk = want_split;
  }
  p.nk_per = p.nk > 0 ? cdiv(p.nk, splitk) : 0;
  splitk = p.nk > 0 ? cdiv(p.nk, p.nk_per) : 1;          // (nk == 0: a read-back-only launch of a tap-less parity class)
  p.kc_per = 0;
  {
    // multi-tap gather launches split K by channel slice (see DmaArgs::kc_per): the number of slices = the divisor of the
    // K steps per tap nearest above what the heuristic asked for (ASPP: 32 steps per tap, 6 asked -> 8 slices of 4 steps).
    // PXL_SPLITK_CHAN=0: contiguous (tap, channel) slices as in rounds 1-4, for A/B runs
    static const bool chan_on = getenv("PXL_SPLITK_CHAN") == nullptr || getenv("PXL_SPLITK_CHAN")[0] != '0';
    const int spt = p.Cin / 64;                           // K steps per tap
    const int ntaps_walked = spt > 0 ? p.nk / spt : 0;
    if (chan_on && gather && splitk > 1 && ntaps_walked > 1 && p.nk == ntaps_walked * spt) {
      int s2 = 0;
      for (int c = splitk; c <= spt && c <= 4 * splitk; ++c) if (spt % c == 0) { s2 = c; break; }
      if (s2 == 0) for (int c = splitk; c >= 2; --c) if (spt % c == 0) { s2 = c; break; }
      if (s2 >= 2) {
        splitk = s2;
        p.kc_per = (unsigned)(spt / s2) * 128u;
        p.nk_per = (spt / s2) * ntaps_walked;             // (informational: K steps per slice)
      }
    }
  }
  // one slab per slice when the workspace holds them (and the tile's fp32 staging fits in the ring), else atomics into one buffer
  constexpr bool can_stage = (size_t)BM * (BN * 4 + 16) <= (size_t)NST * (BM + BN) * 128;
  p.ws_slab = 0;
  if ((splitk > 1 || p.raw_slabs) && can_stage && p.Cout % 4 == 0 && ws_bytes >= (size_t)splitk * p.M * p.Cout * sizeof(float))
    p.ws_slab = (size_t)p.M * p.Cout;
  if (p.raw_slabs && p.ws_slab == 0) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_slabs: this tile cannot stage its fp32 partial sums");
  if (splitk > 1 && p.ws_slab == 0) PXL_CHECK_HIP(hipMemsetAsync(p.ws, 0, (size_t)p.M * p.Cout * sizeof(float), stream));
  if (splitk <= 1 && !p.raw_slabs) p.ws = nullptr;
  const bool bnin = p.bin.coef != nullptr;
  const size_t smem = (size_t)NST * (BM + BN) * 128 + (bnin ? (size_t)p.Cin * 8 : 0);
  if (smem > 156 * 1024) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: tile + coefficient table exceed the LDS");
  if (groups == 2 && (splitk > 1 || p.addend || p.bn_y || p.fin.coef || p.trace))
    return pxl_set_error(PXL_ERR_ARG, "conv_dma: a paired launch is a plain forward convolution (no split-K / addend / finalize)");
  const dim3 g(grid, splitk, groups), b(WM * WN * 64);
  if (p.trace != nullptr) {          // timeline probe (tools/cbench): the forward-with-statistics kernel with cycle stamps
    if (bnin || p.stats == nullptr || p.addend != nullptr || p.bias != nullptr)
      return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: the trace build is the plain forward convolution with statistics");
    return gather ? launch_one<BM, BN, WM, WN, NST, true, false, true, 4>(g, b, smem, stream, p)
                  : launch_one<BM, BN, WM, WN, NST, false, false, true, 4>(g, b, smem, stream, p);
  }
  // operand combination of the read-back passes (epi_passes): a compile-time constant of the kernel that is launched
  const bool has_stats = p.stats != nullptr, has_bnr = has_stats && p.bn_y != nullptr, has_mask = has_bnr && p.bn_mask != nullptr;
  int em = (p.addend ? 1 : 0) | (p.bias ? 2 : 0) | (has_stats ? 4 : 0) | (has_bnr ? 8 : 0) | (has_mask ? 16 : 0) |
           ((has_bnr && !has_mask && p.bn_relu) ? 32 : 0) | ((has_mask && p.mask_bits) ? 64 : 0);
  if (splitk > 1 || p.raw_slabs) em = -2;
  else if (p.sub_mul != 1) em = -1;          // sub-grid row addresses: the generic read-back
  int rc;
#define PXL_EM(E) case E: rc = gather ? launch_one<BM, BN, WM, WN, NST, true, false, false, E>(g, b, smem, stream, p) \
                                       : launch_one<BM, BN, WM, WN, NST, false, false, false, E>(g, b, smem, stream, p); break;
#define PXL_EM_BNIN(E) case E: rc = gather ? launch_one<BM, BN, WM, WN, NST, true, true, false, E>(g, b, smem, stream, p) \
                                            : launch_one<BM, BN, WM, WN, NST, false, true, false, E>(g, b, smem, stream, p); break;
  if (!bnin) {
    switch (em) {
      PXL_EM(-2)                       // split-K partial sums
      PXL_EM(0) PXL_EM(1)              // plain store; + addend (gradient accumulation)
      PXL_EM(4)                        // forward with batch statistics
      PXL_EM(12) PXL_EM(13)            // data gradient + BatchNorm-backward sums (+ addend)
      PXL_EM(44) PXL_EM(45)            // ... through the BN's ReLU
      PXL_EM(28) PXL_EM(29)            // ... of a residual join
      PXL_EM(92) PXL_EM(93)            // ... with the join's ReLU mask as a bit plane
      default: rc = gather ? launch_one<BM, BN, WM, WN, NST, true, false, false, -1>(g, b, smem, stream, p)
                           : launch_one<BM, BN, WM, WN, NST, false, false, false, -1>(g, b, smem, stream, p);
    }
  } else {
    switch (em) {
      PXL_EM_BNIN(0) PXL_EM_BNIN(4)
      default: rc = gather ? launch_one<BM, BN, WM, WN, NST, true, true, false, -1>(g, b, smem, stream, p)
                           : launch_one<BM, BN, WM, WN, NST, false, true, false, -1>(g, b, smem, stream, p);
    }
  }
#undef PXL_EM
#undef PXL_EM_BNIN
  if (rc != PXL_OK) return rc;
  if (p.raw_slabs) return PXL_OK;               // (the caller sums the slabs)
  if (splitk > 1 && p.ws_slab != 0)
    return pxl_splitk_finish_slabs(PXL_BF16, (long)p.M * p.Cout, p.Cout, p.Kreal, splitk, p.ws, p.bias, p.out, stream);
  if (splitk > 1)
    return pxl_splitk_finish(PXL_BF16, (long)p.M * p.Cout, p.Cout, p.Kreal, p.ws, p.bias, p.out, stream);
  return PXL_OK;
}

}  // namespace pxl_dma

// The tile configurations are instantiated in six translation units (conv_dma_a .. f.hip) so that the build compiles them in
// parallel: each defines one dispatcher over its share of the configuration numbers.
int pxl_dma_launch_a(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 8..11  (3-stage 2x2)
int pxl_dma_launch_b(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 16..19 (2-stage 2x2)
int pxl_dma_launch_c(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 20..23 (3-stage tall)
int pxl_dma_launch_d(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 24..27 (2-stage tall)
int pxl_dma_launch_e(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 28..31 (8 waves)
int pxl_dma_launch_f(int cfg, const pxl_dma::DmaArgs& a, bool gather, int sk, size_t ws_bytes, hipStream_t s, int groups);   // 32..35 (8 waves)
// conv_halo.hip: tile configurations 40..43 = the halo-tile kernel (conv_halo_kernel.h) for "same" multi-tap convolutions
int pxl_halo_reach(const pxl_dma::DmaArgs& a);                       // 0 = not eligible, else the padding reach d
int pxl_halo_launch(int cfg, const pxl_dma::DmaArgs& a, hipStream_t s);
