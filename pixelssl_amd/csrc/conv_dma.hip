// LDS-DMA implicit-GEMM convolution for gfx950 (bf16, NHWC, channel pitch % 64 == 0).
//
//   out[m][n] = sum_{t,c} in[pix(m,t)][c] * w[n][t][c]        m = (b,oy,ox), n = out channel
//
// This is the fast path behind pxl_conv_igemm for plain (already activated) bf16 inputs: forward
// 1x1 / 3x3 / atrous / strided convolutions and the data gradients of stride-1 convolutions.
// Differences to conv_igemm.hip (which stays the generic / fp32-parity kernel):
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
#include <cstdlib>
#include "common.h"

namespace {

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
  int B, Hi, Wi, Cin;
  int Ho, Wo, Cout, Kreal;
  int ntaps, so;
  int div_shift;     // data gradient of a stride-2 convolution: source pixel = (oy + dy, ox + dx) / 2 where both are even
  int M, Ktot, nk;
  int tiles_m, tiles_n;
  float* ws;         // split-K: pre-zeroed fp32 [M][Cout] accumulation buffer (blockIdx.y = K slice), else nullptr
  int nk_per;        // K steps per slice
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
  unsigned* trace;   // TRACE kernels (tools/cbench): [workgroup][TRACE_WORDS] cycle stamps of wave 0, else unused
  int taps[64];      // (dy << 16) | (dx & 0xffff)
};

// timeline probe (tools/cbench.cpp; never on the product path): per workgroup, words 0..63 = s_memtime stamps of wave 0
// (0 entry, 1 prologue issued, 2 + k = end of K step k (first 56), then loop drained / tile staged / stores done),
// 64 = number of stamps, 65 = HW_ID, 66 = XCC_ID, 67/68 = s_memrealtime (100 MHz) at entry, 69/70 at exit, 71 = nk
constexpr int TRACE_WORDS = 72;

constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds, 16, (int)voff, (int)soff, 0, 0);
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
  if constexpr (LA == 2)
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
template <int I, int N> struct XformLoad {
  static __device__ __forceinline__ void run(u32x4 (&d)[N], unsigned addr) {
    d[I] = lds_read128<I * 4096>(addr);
    if constexpr (I + 1 < N) XformLoad<I + 1, N>::run(d, addr);
  }
};
template <int I, int N> struct XformStore {
  static __device__ __forceinline__ void run(const u32x4 (&d)[N], unsigned addr) {
    lds_write128<I * 4096>(addr, d[I]);
    if constexpr (I + 1 < N) XformStore<I + 1, N>::run(d, addr);
  }
};

template <int I, int N, int STRIDE, int BASE> struct FragLoad {
  static __device__ __forceinline__ void run(u32x4 (&f)[N], unsigned addr) {
    f[I] = lds_read128<BASE + I * STRIDE>(addr);
    if constexpr (I + 1 < N) FragLoad<I + 1, N, STRIDE, BASE>::run(f, addr);
  }
};

// BM x BN output tile (pixels x channels), 4 waves as WM x WN, NST LDS stages, GATHER = taps / padding logic
// ABL: timing ablations for tools/conv_bench.py (results are garbage): 1 = no DMA in the loop, 2 = no MFMA,
// 4 = no fragment reads, 8 = no barrier.  0 in every product instantiation.
template <int BM, int BN, int WM, int WN, int NST, bool GATHER, int ABL = 0, bool BNIN = false, bool TRACE = false>
__global__ __launch_bounds__(256) void conv_dma_kernel(const DmaArgs p) {
  constexpr int TMI = BM / WM / 32;          // 32-pixel tiles per wave
  constexpr int TNI = BN / WN / 32;          // 32-channel tiles per wave
  constexpr int LA = BM / 32, LB = BN / 32;  // DMA instructions per wave per K step
  constexpr int SB = (BM + BN) * 128;        // bytes per stage
  constexpr int TP = BN * 2 + 16;            // epilogue staging row pitch (bank-conflict-free ds_write_b64)
  constexpr int TPR = BN / 8;                // threads per output row on the read-back pass
  constexpr int RPP = 256 / TPR;             // rows per pass
  constexpr int NPASS = BM / RPP;
  static_assert(WM * WN == 4 && TMI >= 1 && TNI >= 1 && ((TMI <= 2 && TNI <= 2) || (TNI == 1 && TMI <= 6)), "tile");
  static_assert(BM % RPP == 0, "epilogue rows per pass must divide the tile");
  static_assert(NST * SB >= BM * TP + 4 * BN * 8, "epilogue staging must fit in the ring");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

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

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- loader coordinates: DMA instruction g = wave + 4*q covers tile rows 8g .. 8g+7, lane -> (row, 16-byte slot)
  const int lrow = lane >> 3, lslot = lane & 7;
  unsigned voffA[LA], voffB[LB];
  int a_pix[LA], a_iy[LA], a_ix[LA], a_img[LA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int q = 0; q < LA; ++q) {
    const int row = (wave + 4 * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int m = m0 + row;
    if (m < p.M) {
      const int b = m / HoWo;
      const int r = m - b * HoWo;
      const int oy = r / p.Wo;
      const int ox = r - oy * p.Wo;
      a_iy[q] = oy * p.so;
      a_ix[q] = ox * p.so;
      a_img[q] = b * p.Hi * p.Wi * p.Cin * 2 + chunk * 16;
      a_pix[q] = a_img[q] + (a_iy[q] * p.Wi + a_ix[q]) * p.Cin * 2;
      voffA[q] = (unsigned)a_pix[q];
    } else {
      a_iy[q] = -(1 << 20);
      a_ix[q] = 0;
      a_pix[q] = 0;
      a_img[q] = 0;
      voffA[q] = OOB;
    }
  }
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int row = (wave + 4 * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int n = n0 + row;
    voffB[q] = n < p.Kreal ? (unsigned)(n * p.Ktot * 2 + chunk * 16) : OOB;
  }

  // ---- load cursor: tap t, byte offset kcb inside the pixel's channel vector, kwb inside the weight row
  const unsigned cin_bytes = (unsigned)p.Cin * 2;
  const int ks_begin = blockIdx.y * p.nk_per;                   // split-K slice (whole range when gridDim.y == 1)
  const int nk_here = min(p.nk, ks_begin + p.nk_per) - ks_begin;
  int ld_t = (int)(((unsigned)ks_begin * 128u) / cin_bytes);
  unsigned kcb = (unsigned)ks_begin * 128u - (unsigned)ld_t * cin_bytes, kwb = (unsigned)ks_begin * 128u;
  auto set_tap = [&](int t) {
    if constexpr (GATHER) {
      const int tp = p.taps[min(t, p.ntaps - 1)];
      const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
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
    for (int q = 0; q < LA; ++q) dma16(r_in, sa + q * 4096, voffA[q], kcb);
    if constexpr (BNIN) {
      unsigned bits = 0;
#pragma unroll
      for (int q = 0; q < LA; ++q) bits |= (voffA[q] != OOB ? 1u : 0u) << q;
      vm = (vm & ~(0xffu << (8 * stage))) | (bits << (8 * stage));
    }
    unsigned char* sb = smem + stage * SB + BM * 128 + wave * 1024;
#pragma unroll
    for (int q = 0; q < LB; ++q) dma16(r_w, sb + q * 4096, voffB[q], kwb);
    kwb += 128;
    kcb += 128;
    if (kcb == cin_bytes) {      // block-uniform: next tap
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
    const pxl_bn_fin& f = p.bin;
    const int C = p.Cin;
    for (int c = tid; c < C; c += 256) {
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
      XformLoad<0, LA>::run(dd, pa);
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
      XformStore<0, LA>::run(dd, pa);
      if constexpr (!GATHER) {
        // materialise z = relu(bn(y)) for the weight gradient: one workgroup per pixel tile writes what it transformed (the
        // same bytes, the same offsets as the source; zero-filled lanes are out of range for the store as well)
        if (p.bin_z != nullptr && tn == 0) {
          const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(p.bin_z, 0, p.in_bytes, 0x00020000);
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
    if constexpr (TRACE) { if (ks < 56) stamp(); }
  }
  wait_vmcnt<0>();                 // the tail DMAs (tiles past nk) must not land in the staging area
  __builtin_amdgcn_s_barrier();
  stamp();

  if constexpr (ABL & 16) return;
  if (p.ws != nullptr) {
    // split-K: fp32 partial sums of this K slice -> workspace; bias / rounding happen in the finish kernel
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
  bf16_t* __restrict__ gout = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* __restrict__ gadd = reinterpret_cast<const bf16_t*>(p.addend);
  const bool has_bias = p.bias != nullptr;
  const bool has_add = gadd != nullptr;
  const bool has_stats = p.stats != nullptr && !(ABL & 32);
  const bf16_t* __restrict__ gbny = reinterpret_cast<const bf16_t*>(p.bn_y);
  const bool has_bnr = has_stats && gbny != nullptr;
  const bf16_t* __restrict__ gmask = reinterpret_cast<const bf16_t*>(p.bn_mask);
  const bool has_mask = has_bnr && gmask != nullptr;
  float bn_mean[8], bn_rstd[8], bn_sc[8], bn_sh[8];
  if (has_bnr && ncol) {
    load_cvec<8>(p.bn_coef + n, bn_mean);
    load_cvec<8>(p.bn_coef + p.Cout + n, bn_rstd);
    load_cvec<8>(p.bn_coef + 2 * p.Cout + n, bn_sc);
    load_cvec<8>(p.bn_coef + 3 * p.Cout + n, bn_sh);
  }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (has_bias && n + e < p.Kreal) ? p.bias[n + e] : 0.f;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int ml = ps * RPP + er;
    const int m = m0 + ml;
    uint4 v = *reinterpret_cast<const uint4*>(T + ml * TP + ec * 16);
    if (m < p.M && ncol) {
      const size_t o = (size_t)m * p.Cout + n;
      if (has_bias || has_add || has_stats) {
        float f[8];
        Chunk<bf16_t>::unpack(v, f);
        if (has_bias || has_add) {
          if (has_add) {
            float ad[8];
            Chunk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(gadd + o), ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += ad[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += bv[e];
          v = Chunk<bf16_t>::pack(f);
          if (has_stats) Chunk<bf16_t>::unpack(v, f);     // statistics of the stored (rounded) values
        }
        if (has_mask) {
          float fy[8], fm[8];
          Chunk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(gbny + o), fy);
          Chunk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(gmask + o), fm);
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
          Chunk<bf16_t>::unpack(*reinterpret_cast<const uint4*>(gbny + o), fy);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float gd = f[e];
            if (p.bn_relu && !(fy[e] * bn_sc[e] + bn_sh[e] > 0.f)) gd = 0.f;
            s1[e] += gd;
            s2[e] += gd * (fy[e] - bn_mean[e]) * bn_rstd[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[e] += f[e];
            s2[e] += f[e] * f[e];
          }
        }
      }
      *reinterpret_cast<uint4*>(gout + o) = v;
    }
  }
  if (has_stats) {
    // reduce over the threads that share a channel chunk: lanes ec + TPR*k inside the wave, then the 4 waves
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = TPR; o < 64; o <<= 1) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    }
    float* red = reinterpret_cast<float*>(smem + BM * TP);     // [4 waves][2][BN]
    if (lane < TPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 2 + 0) * BN + ec * 8 + e] = s1[e];
        red[(wave * 2 + 1) * BN + ec * 8 + e] = s2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] +
                      red[(2 * 2 + which) * BN + c] + red[(3 * 2 + which) * BN + c];
      if (n0 + c < p.Kreal) {
        float* rep = p.stats + (size_t)(tm % p.stats_rep) * 2 * p.Kreal;
        atomicAdd(rep + which * p.Kreal + n0 + c, v);
      }
    }
  }
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
      for (int c = tid; c < C; c += 256) {
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
      unsigned* t = p.trace + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * TRACE_WORDS;
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
  p.nk_per = p.nk;
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

template <int BM, int BN, int WM, int WN, int NST>
int launch_dma(const DmaArgs& a, bool gather, int want_split, size_t ws_bytes, hipStream_t stream) {
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
  p.nk_per = cdiv(p.nk, splitk);
  splitk = cdiv(p.nk, p.nk_per);
  if (splitk > 1) PXL_CHECK_HIP(hipMemsetAsync(p.ws, 0, (size_t)p.M * p.Cout * sizeof(float), stream));
  else p.ws = nullptr;
  const bool bnin = p.bin.coef != nullptr;
  const size_t smem = (size_t)NST * (BM + BN) * 128 + (bnin ? (size_t)p.Cin * 8 : 0);
  static bool raised[4] = {false, false, false, false};
  const int vi = (gather ? 1 : 0) + (bnin ? 2 : 0);
  const void* fn = vi == 0 ? reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, false>)
                 : vi == 1 ? reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, true>)
                 : vi == 2 ? reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, false, 0, true>)
                           : reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, true, 0, true>);
  if (!raised[vi]) {
    PXL_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    raised[vi] = true;
  }
  if (smem > 156 * 1024) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: tile + coefficient table exceed the LDS");
  if (p.trace != nullptr) {          // timeline probe (tools/cbench): the same kernel with cycle stamps
    if (bnin) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma: no trace build of the BN-on-load kernel");
    const void* tf = gather ? reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, true, 0, false, true>)
                            : reinterpret_cast<const void*>(&conv_dma_kernel<BM, BN, WM, WN, NST, false, 0, false, true>);
    PXL_CHECK_HIP(hipFuncSetAttribute(tf, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    if (gather) hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, true, 0, false, true>), dim3(grid, splitk), dim3(256), smem, stream, p);
    else hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, false, 0, false, true>), dim3(grid, splitk), dim3(256), smem, stream, p);
    PXL_LAUNCH_CHECK();
    return PXL_OK;
  }
  switch (vi) {
    case 0: hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, false>), dim3(grid, splitk), dim3(256), smem, stream, p); break;
    case 1: hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, true>), dim3(grid, splitk), dim3(256), smem, stream, p); break;
    case 2: hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, false, 0, true>), dim3(grid, splitk), dim3(256), smem, stream, p); break;
    default: hipLaunchKernelGGL((conv_dma_kernel<BM, BN, WM, WN, NST, true, 0, true>), dim3(grid, splitk), dim3(256), smem, stream, p); break;
  }
  PXL_LAUNCH_CHECK();
  if (splitk > 1)
    return pxl_splitk_finish(PXL_BF16, (long)p.M * p.Cout, p.Cout, p.Kreal, p.ws, p.bias, p.out, stream);
  return PXL_OK;
}

}  // namespace

// 1 if the descriptor / operand combination can run on the LDS-DMA kernel
extern "C" int pxl_conv_dma_eligible(const pxl_conv_desc* d, const float* in_scale, const void* workspace) {
  (void)workspace;
  if (d->dtype != PXL_BF16 || in_scale != nullptr) return 0;
  if (d->Cin % 64 != 0 || d->Cout % 8 != 0 || (d->div != 1 && d->div != 2)) return 0;
  if (d->div == 2 && d->out_stride != 1) return 0;
  if (d->div == 2) {
    // data gradient of a stride-2 convolution: the kernel walks ALL taps and lets the parity test zero the 3 of 4 that do
    // not apply to a pixel.  Measured a win for the 1x1 / 3x3 strided convolutions of the ResNet (125 -> 48 us) and still
    // slightly ahead of the gathering kernel for the 4x4 stacks of the discriminators (AdvSSL 31.5 vs 32.4 ms, GCT 54.0 vs
    // 55.4 ms in the same call).  PXL_DMA_STRIDED_DGRAD=<max taps> restricts it.
    static const int max_taps = getenv("PXL_DMA_STRIDED_DGRAD") ? atoi(getenv("PXL_DMA_STRIDED_DGRAD")) : 64;
    if (d->ntaps > max_taps) return 0;
  }
  if ((long)d->Kreal * d->ntaps * d->Cin * 2 >= (1L << 31)) return 0;
  return 1;
}

// tile configurations 8..: 8 = 128x128, 9 = 128(pixels)x64, 10 = 64x128, 11 = 64x64 with a 3-stage LDS ring;
// 12..15 the same with 4 stages, 16..19 with 2 stages (more blocks per CU);
// 20..27: tall tiles 96 / 160 / 192 / 256 (pixels) x 128 with ONE wave column (WM = 1, WN = 4): M = 8 * 33 * 33 = 8712 is
// 68.06 tiles of 128 and 136.1 tiles of 64 -- every ResNet-101 stage-3/4 launch ends with a nearly empty last wave of
// workgroups (274 on 256 CUs).  96-row tiles give 91 x N/128 workgroups (182 for N = 256: one wave at 1.17x the tile
// traffic instead of two), 160 / 192 rows fit N = 512 / 2048 the same way.  20..23 = 3 stages, 24..27 = 2 stages
namespace { int conv_dma_launch(const pxl_conv_desc* d, DmaArgs& a, int sk, size_t ws_bytes, void* stream); }

extern "C" int pxl_conv_dma(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                            const void* addend, float* stats, void* workspace, size_t ws_bytes, void* stream) {
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = addend; a.stats = stats;
  a.ws = reinterpret_cast<float*>(workspace);
  a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  const int sk = d->split_k;
  return conv_dma_launch(d, a, sk, ws_bytes, stream);
}

// Timeline probe of pxl_conv_dma (tools/cbench.cpp, not on the product path): the same launch built with cycle stamps;
// trace = [workgroups][72] uint32 (layout: TRACE_WORDS above).  Plain operands only (no split-K, no BN-on-load).
extern "C" int pxl_conv_dma_trace(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                  float* stats, unsigned* trace, void* stream) {
  PXL_REQUIRE(d && in && w && out && trace, "conv_dma_trace: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_trace: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = trace;
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Forward convolution with batch statistics AND the BatchNorm finalize: `stats` ([stats_rep][2*Kreal], caller-zeroed)
// receives the sums as in pxl_conv_igemm, and the workgroup that finishes last computes what pxl_bn_finalize would
// (fin->coef, running statistics) from them -- `counter` is one caller-zeroed uint32 per launch.  fin->stats / nrep are
// ignored (the launch's own stats / stats_rep are used).  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the
// descriptor or the launch would split K.
extern "C" int pxl_conv_dma_finalize(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                     float* stats, const pxl_bn_fin* fin, unsigned* counter, void* stream) {
  PXL_REQUIRE(d && in && w && out && stats && fin && fin->coef && counter && fin->count > 0.f, "conv_dma_finalize: bad argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || (d->tile_cfg >= 0 && d->tile_cfg < 8) || !fin->training)
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_finalize: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin = *fin; a.fin_counter = counter;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Forward convolution whose input is relu?(bn(y)) of the previous convolution's RAW output y, applied to the tiles as they
// land in LDS: stands in for pxl_bn_finalize + pxl_bn_apply_fwd + pxl_conv_igemm (no materialised activation, one launch
// instead of two or three).  `bin`: the input BatchNorm (statistics [nrep][2*Cin] or running statistics, affine
// parameters, coef [4*Cin] written by workgroup 0, running statistics updated there); d->Cin == the BatchNorm's channel
// count <= 512, a multiple of 64.  z (optional, 1x1 / stride-1 convolutions only): the activated tensor, written by the
// workgroups of output-channel tile 0.  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the descriptor.
extern "C" int pxl_conv_dma_bnin(const pxl_conv_desc* d, const void* y, const void* w, void* out, const float* bias,
                                 float* stats, const pxl_bn_fin* bin, int bin_relu, void* z, void* stream) {
  PXL_REQUIRE(d && y && w && out && bin && bin->coef && bin->count > 0.f, "conv_dma_bnin: bad argument");
  PXL_REQUIRE(bin->training ? (bin->stats != nullptr && bin->nrep >= 1) : (bin->running_mean && bin->running_var),
              "conv_dma_bnin: missing statistics");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->div != 1 || d->Cin > 512 || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: descriptor is not eligible for the BN-on-load kernel");
  DmaArgs a;
  a.in = y; a.w = w; a.out = out; a.bias = bias; a.addend = nullptr; a.stats = stats;
  a.ws = nullptr; a.nk_per = 0;
  a.bn_y = nullptr; a.bn_coef = nullptr; a.bn_relu = 0; a.bn_mask = nullptr;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bin = *bin; a.bin_relu = bin_relu; a.bin_z = z; a.trace = nullptr;
  if (z != nullptr) {          // the activated tensor can only be written by a kernel that walks every input pixel exactly once per tile row
    bool plain = d->ntaps == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->out_stride == 1 && d->Ho == d->Hi && d->Wo == d->Wi;
    if (!plain) return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dma_bnin: z output needs a 1x1 / stride-1 convolution");
  }
  return conv_dma_launch(d, a, 1, 0, stream);
}

// Data gradient with the BatchNorm-backward reduction of its output fused into the epilogue: din = dgrad(dy) (+ addend)
// and bn_sums[0..C) += sum_m gd, bn_sums[C..2C) += sum_m gd * xhat over the tensor just written (C = d->Kreal ==
// d->Cout: an unpadded channel count), gd = din * (bn_relu ? bn_coef.scale * bn_y + bn_coef.shift > 0 : 1).  Replaces
// a separate pxl_bn_bwd_reduce pass over (din, bn_y).  PXL_ERR_UNSUPPORTED when the LDS-DMA kernel cannot run the
// descriptor (the caller then uses pxl_conv_igemm + pxl_bn_bwd_reduce).
extern "C" int pxl_conv_dgrad_bnreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                       const void* bn_y, const float* bn_coef, int bn_relu, float* bn_sums, void* stream) {
  PXL_REQUIRE(d && dy && wt && din && bn_y && bn_coef && bn_sums, "conv_dgrad_bnreduce: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->Kreal != d->Cout || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dgrad_bnreduce: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = dy; a.w = wt; a.out = din; a.bias = nullptr; a.addend = addend; a.stats = bn_sums;
  a.ws = nullptr; a.nk_per = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = bn_relu; a.bn_mask = nullptr;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  q.stats_rep = 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}

// Data gradient that completes the gradient of a residual join's output, with the join's backward fused into the
// epilogue: g = dgrad(dy) (+ addend); din = g * (join_out > 0) -- the ReLU after the join -- and bn_sums[0..C) +=
// sum_m din, bn_sums[C..2C) += sum_m din * xhat(bn_y) for the main branch's last BatchNorm (the one without a ReLU of its
// own).  Replaces pxl_residual_bwd_reduce (3 tensor reads + 2 writes) by two extra reads in this epilogue.
extern "C" int pxl_conv_dgrad_joinreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                         const void* join_out, const void* bn_y, const float* bn_coef, float* bn_sums,
                                         void* stream) {
  PXL_REQUIRE(d && dy && wt && din && join_out && bn_y && bn_coef && bn_sums, "conv_dgrad_joinreduce: null argument");
  if (!pxl_conv_dma_eligible(d, nullptr, nullptr) || d->Kreal != d->Cout || (d->tile_cfg >= 0 && d->tile_cfg < 8))
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "conv_dgrad_joinreduce: descriptor is not eligible for the LDS-DMA kernel");
  DmaArgs a;
  a.in = dy; a.w = wt; a.out = din; a.bias = nullptr; a.addend = addend; a.stats = bn_sums;
  a.ws = nullptr; a.nk_per = 0;
  a.fin.coef = nullptr; a.fin_counter = nullptr;
  a.bn_y = bn_y; a.bn_coef = bn_coef; a.bn_relu = 0; a.bn_mask = join_out;
  a.bin.coef = nullptr; a.bin_relu = 0; a.bin_z = nullptr; a.trace = nullptr;
  pxl_conv_desc q = *d;
  q.stats_rep = 1;
  return conv_dma_launch(&q, a, 1, 0, stream);
}

namespace {
int conv_dma_launch(const pxl_conv_desc* d, DmaArgs& a, int sk, size_t ws_bytes, void* stream) {
  a.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.Kreal = d->Kreal;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.div_shift = d->div == 2 ? 1 : 0;
  a.M = d->B * d->Ho * d->Wo;
  a.Ktot = d->ntaps * d->Cin;
  a.nk = a.Ktot / 64;
  a.tiles_m = a.tiles_n = 0;
  a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->Cin * 2);
  a.w_bytes = (unsigned)((size_t)d->Kreal * a.Ktot * 2);
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // taps at offset (0,0) only and every output pixel inside the input: no bounds logic at all
  bool gather = false;
  for (int t = 0; t < d->ntaps; ++t) gather |= d->dy[t] != 0 || d->dx[t] != 0;
  gather |= d->ntaps != 1;
  gather |= d->div != 1;
  gather |= (d->Ho - 1) * d->out_stride >= d->Hi || (d->Wo - 1) * d->out_stride >= d->Wi;
  int cfg = d->tile_cfg;
  if (cfg < 8) {
    const long t128 = (long)cdiv(a.M, 128) * cdiv(a.Cout, 128);
    if (a.Cout <= 64) cfg = 9;
    else if (t128 >= 384) cfg = 8;
    else cfg = 9;
  }
  switch (cfg) {
    case 8: return launch_dma<128, 128, 2, 2, 3>(a, gather, sk, ws_bytes, s);
    case 9: return launch_dma<128, 64, 2, 2, 3>(a, gather, sk, ws_bytes, s);
    case 10: return launch_dma<64, 128, 2, 2, 3>(a, gather, sk, ws_bytes, s);
    case 11: return launch_dma<64, 64, 2, 2, 3>(a, gather, sk, ws_bytes, s);
    case 12: return launch_dma<128, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 13: return launch_dma<128, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 14: return launch_dma<64, 128, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 15: return launch_dma<64, 64, 2, 2, 4>(a, gather, sk, ws_bytes, s);
    case 16: return launch_dma<128, 128, 2, 2, 2>(a, gather, sk, ws_bytes, s);
    case 17: return launch_dma<128, 64, 2, 2, 2>(a, gather, sk, ws_bytes, s);
    case 18: return launch_dma<64, 128, 2, 2, 2>(a, gather, sk, ws_bytes, s);
    case 19: return launch_dma<64, 64, 2, 2, 2>(a, gather, sk, ws_bytes, s);
    case 20: return launch_dma<96, 128, 1, 4, 3>(a, gather, sk, ws_bytes, s);
    case 21: return launch_dma<160, 128, 1, 4, 3>(a, gather, sk, ws_bytes, s);
    case 22: return launch_dma<192, 128, 1, 4, 3>(a, gather, sk, ws_bytes, s);
    case 23: return launch_dma<128, 128, 1, 4, 3>(a, gather, sk, ws_bytes, s);
    case 24: return launch_dma<96, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s);
    case 25: return launch_dma<160, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s);
    case 26: return launch_dma<192, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s);
    case 27: return launch_dma<128, 128, 1, 4, 2>(a, gather, sk, ws_bytes, s);
    // timing ablations (garbage results): 64x128 3-stage gather kernel, 100 + ABL bits
    case 100: return launch_abl<64, 128, 3, 0>(a, s);
    case 101: return launch_abl<64, 128, 3, 1>(a, s);
    case 102: return launch_abl<64, 128, 3, 2>(a, s);
    case 103: return launch_abl<64, 128, 3, 3>(a, s);
    case 104: return launch_abl<64, 128, 3, 4>(a, s);
    case 106: return launch_abl<64, 128, 3, 6>(a, s);
    case 107: return launch_abl<64, 128, 3, 7>(a, s);
    case 115: return launch_abl<64, 128, 3, 15>(a, s);
    case 116: return launch_abl<64, 128, 3, 16>(a, s);
    case 131: return launch_abl<64, 128, 3, 31>(a, s);
    case 132: return launch_abl<64, 128, 3, 32>(a, s);
    case 147: return launch_abl<64, 128, 3, 47>(a, s);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_dma: unknown tile config %d", cfg);
  }
}
}  // namespace
