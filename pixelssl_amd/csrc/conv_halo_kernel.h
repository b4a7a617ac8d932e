// Halo-tile LDS-DMA convolution for gfx950: "same" multi-tap convolutions (3x3, stride 1, any dilation d) whose A operand is
// loaded ONCE per 64-channel slab and walked by all taps from LDS.
//
// conv_dma_kernel.h makes one K step = one tap: the pixel tile crosses the L2 -> LDS path nine times for a 3x3, in a loop
// that tools/cbench --floor showed to be bound by exactly that path (~40 B/clk/CU).  Here the output tile is BM consecutive
// positions of a PADDED, flattened grid:
//
//     position v = (b * Hp + yy) * Wp + xx,   Hp = H + d, Wp = W + d   (d padding rows below / columns right of every image)
//
// In that grid a tap (dy, dx) is the CONSTANT shift dy * Wp + dx, and the zero padding of the convolution is the padding of
// the grid: column -1 of row yy is column Wp - 1 of row yy - 1, row -1 of image b is row Hp - 1 of image b - 1 (the d shared
// padding rows / columns are enough for |dy|, |dx| <= d).  So the BM + 2 * (d * Wp + d) grid positions around a tile -- the
// "halo slab", 128 bytes = 64 channels per position -- are DMA'd once per channel slab (padding positions are out-of-range
// lanes: the buffer descriptor zero-fills them), and tap t reads its MFMA fragments at LDS row r + shift[t].  The XOR swizzle
// of conv_dma_kernel.h is a function of the LDS row, and a 16-lane ds_read_b128 group still covers all 8 swizzle values x
// both row parities for ANY row shift: the shifted reads stay bank-conflict-free.
//
//   DMA bytes per K step (BM = BN = 128): 16 KB weights + 1/9 of a ~25-33 KB slab, instead of 16 KB + 16 KB;
//   the weight tiles go through an NSTW-stage ring exactly as before; the next slab's pieces are issued at most one per wave and
//   step underneath the current slab's nine taps (double-buffered), each BEFORE the weight pieces of its step, and the loop's
//   one `s_waitcnt vmcnt(LB)` per step leaves only the youngest weight tile in flight (a slab piece is at most one step old
//   when it is waited for: with two waves per SIMD the partner computes meanwhile);
//   outputs at padding positions (1 - H * W / (Hp * Wp): 5.8 % at 33 x 33, 3 % at 65 x 65) are computed and dropped.
//
// With BNIN the slab holds the RAW output y of the previous convolution and relu?(bn(y)) is applied in LDS by the lane that
// DMA'd the piece -- once per element (the tap-per-step kernel would transform every piece nine times: measured 59 vs 40 us
// per layer-3 convolution in round 3, which is why only 1x1 consumers were on-load until now).  The workgroups of
// output-channel tile 0 write the activated CENTRE rows of their slab to z for the weight gradient.
//
// Epilogue: the read-back passes of conv_dma_kernel.h (epi_passes) with the padded-grid row mapping (VIRT).
#pragma once
#include "conv_dma_kernel.h"

namespace pxl_halo {
using namespace pxl_dma;

struct HaloArgs {
  const void* in;
  const void* w;
  void* out;
  const float* bias;
  const void* addend;
  float* stats;
  int stats_rep;
  const void* bn_y;          // data-gradient launches: see DmaArgs
  const float* bn_coef;
  int bn_relu;
  const void* bn_mask;
  int B, H, W, Cin, Cout, Kreal;
  int ntaps, Ktot;           // Ktot = (taps of the weight tensor) * Cin
  int Hp, Wp, Mv;            // padded grid, Mv = B * Hp * Wp positions
  int hoff;                  // d * Wp + d: slab rows in front of the tile's first position
  int AG;                    // 1 KiB pieces (8 positions x 128 B) of one slab: ceil((BM + 2 * hoff) / 8)
  int nslab;                 // Cin / 64
  int tiles_m, tiles_n;
  pxl_bn_fin fin;            // forward: the last workgroup finalizes the BatchNorm (fin.coef == nullptr: off)
  unsigned* fin_counter;
  pxl_bn_fin bin;            // BNIN: BatchNorm of the input (DmaArgs::bin)
  int bin_relu;
  void* bin_z;
  unsigned in_bytes, w_bytes, out_bytes;
  int taps[16];              // (weight tap index << 24) | slab row shift (dy + d) * Wp + (dx + d)
};

// BM x BN tile, WM x WN = 4 or 8 waves, NSTW-stage weight ring.  Eight waves (two per SIMD) are the default: the slab +
// ring of a 128 x 128 tile is ~100 KB of LDS = ONE workgroup per CU, and a lone wave per SIMD serialises its DMA issue
// (100-180 cycles per `buffer_load ... lds`), address arithmetic and MFMAs -- measured 33.8 us on l3.3x3 with four waves
// against 20.4 us for the tap-per-step kernel, which hides the same costs behind three co-resident workgroups.
template <int BM, int BN, int WM, int WN, int NSTW, bool BNIN, int EM>
__global__ __launch_bounds__(WM * WN * 64) void conv_halo_kernel(const HaloArgs p) {
  constexpr int TMI = BM / WM / 32;
  constexpr int TNI = BN / WN / 32;
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int LB = BN / (8 * NW);          // weight-tile DMA instructions per wave per K step
  constexpr int WB = BN * 128;               // bytes of one weight stage
  constexpr int TP = BN * 2 + 16;
  constexpr int TPR = BN / 8;
  constexpr int RPP = NT / TPR;
  constexpr int NPASS = BM / RPP;
  static_assert((NW == 4 || NW == 8) && NSTW == 3 && LB >= 1 && TMI >= 1 && TNI >= 1 && TMI <= 2 && TNI <= 2, "tile");
  static_assert(BM % RPP == 0, "epilogue rows per pass must divide the tile");

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  kernarg_touch<sizeof(HaloArgs)>();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // tap words, one per lane (compile-time indices: scalar loads + selects; read back with v_readlane -- no memory counter in
  // the loop, where every lgkmcnt / vmcnt value is spoken for)
  int tapv = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) tapv = lane == t ? p.taps[t] : tapv;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int v0 = tm * BM, n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- LDS map: [slab buffer 0][slab buffer 1 (nslab > 1)][weight ring][BNIN table]
  const int AB = p.AG * 1024;
  const int nbuf = p.nslab > 1 ? 2 : 1;
  const int wring = nbuf * AB;
  const int tabo = wring + NSTW * WB;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);

  const int lrow = lane >> 3, lslot = lane & 7;
  const int HpWp = p.Hp * p.Wp;
  const float inv_hpwp = 1.0f / (float)HpWp, inv_wp = 1.0f / (float)p.Wp;
  unsigned voffB[LB];
#pragma unroll
  for (int q = 0; q < LB; ++q) {
    const int row = (wave + NW * q) * 8 + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int n = n0 + row;
    voffB[q] = n < p.Kreal ? (unsigned)(n * p.Ktot * 2 + chunk * 16) : OOB;
  }

  f32x16 acc[TNI][TMI];
#pragma unroll
  for (int j = 0; j < TNI; ++j)
#pragma unroll
    for (int i = 0; i < TMI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fx = (frow >> 1) & 7;
  unsigned boff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) boff[kk] = frow * 128 + (((2 * kk + fhalf) ^ fx) << 4) + wn * TNI * 4096;
  const int rbase = frow + wm * TMI * 32;     // slab row of this lane's first A fragment at shift 0

  // (loop-invariant arguments as locals: left in the argument block, hipcc re-loads them with s_load + `s_waitcnt lgkmcnt(0)` in
  // every interval -- an EMPTY interval took 1 us)
  const int ntaps = p.ntaps, nslab = p.nslab, AG = p.AG, hoff = p.hoff, Mv = p.Mv, pH = p.H, pW = p.W, pWp = p.Wp, pCin = p.Cin;
  const int nk = nslab * ntaps;
  const unsigned cin2 = (unsigned)pCin * 2u;

  // source offset of this lane's 16 bytes of slab piece j (8 positions): position -> pixel, padding / outside -> OOB
  auto a_voff = [&](int j) -> unsigned {
    const int row = 8 * j + lrow;
    const int chunk = lslot ^ ((row >> 1) & 7);
    const int v = v0 - hoff + row;
    int b, r, yy, xx;
    fast_divmod(v < 0 ? 0 : v, HpWp, inv_hpwp, b, r);
    fast_divmod(r, pWp, inv_wp, yy, xx);
    const bool ok = v >= 0 && v < Mv && yy < pH && xx < pW;
    return ok ? (unsigned)((((b * pH + yy) * pW + xx) * pCin) * 2 + chunk * 16) : OOB;
  };
  // ---- prologue: slab 0 entirely, then the first NSTW - 1 weight tiles
  for (int j = wave; j < AG; j += NW) dma16(r_in, smem + j * 1024, a_voff(j), 0u);
  int l_t = 0, l_s = 0;                       // load cursor of the weight ring
  auto issue_w = [&](int stage) {
    const int tp = __builtin_amdgcn_readlane(tapv, l_t);
    const bool past = l_s >= nslab;
    const unsigned kwb = ((unsigned)tp >> 24) * cin2 + (unsigned)l_s * 128u;
    unsigned char* sb = smem + wring + stage * WB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LB; ++q) dma16(r_w, sb + q * NW * 1024, past ? OOB : voffB[q], kwb);
    if (++l_t == ntaps) { l_t = 0; ++l_s; }
  };
#pragma unroll
  for (int s = 0; s < NSTW - 1; ++s) issue_w(s);

  // ---- BNIN: (scale, shift) of every input channel -> LDS table
  const unsigned tab0 = lds0 + (unsigned)tabo;
  if constexpr (BNIN) {
    float* tab = reinterpret_cast<float*>(smem + tabo);
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
        if (blockIdx.x == 0 && f.running_mean != nullptr) {
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
      if (blockIdx.x == 0) { f.coef[c] = mean; f.coef[C + c] = rstd; f.coef[2 * C + c] = scale; f.coef[3 * C + c] = shift; }
    }
    __syncthreads();
  }
  // relu?(scale * y + shift) on slab piece j of channel slab `slab`, in place, by the lane that DMA'd it (its own bytes are
  // visible to it after its vmcnt wait); zero-filled lanes stay zero; the workgroups of channel tile 0 write the CENTRE rows
  // (each pixel is the centre of exactly one tile) to z
  auto xform = [&](int j, int slab, int buf) {
    if constexpr (BNIN) {
      const unsigned vo = a_voff(j);
      const int row = 8 * j + lrow;
      const int chunk = lslot ^ ((row >> 1) & 7);
      const unsigned pa = lds0 + (unsigned)(buf * AB + j * 1024 + lane * 16);
      const unsigned tb = tab0 + (unsigned)(slab * 64 + chunk * 8) * 4u;
      const unsigned tb2 = tb + (unsigned)p.Cin * 4u;
      u32x4 cf[4], dd[1];
      cf[0] = lds_read128<0>(tb);  cf[1] = lds_read128<16>(tb);
      cf[2] = lds_read128<0>(tb2); cf[3] = lds_read128<16>(tb2);
      dd[0] = lds_read128<0>(pa);
      wait_xform<1>(dd, cf);
      float f[8];
      Chunk<bf16_t>::unpack(make_uint4(dd[0][0], dd[0][1], dd[0][2], dd[0][3]), f);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = f[e] * __uint_as_float(cf[0][e]) + __uint_as_float(cf[2][e]);
        const float b = f[4 + e] * __uint_as_float(cf[1][e]) + __uint_as_float(cf[3][e]);
        f[e] = p.bin_relu ? fmaxf(a, 0.f) : a;
        f[4 + e] = p.bin_relu ? fmaxf(b, 0.f) : b;
      }
      const uint4 r = Chunk<bf16_t>::pack(f);
      if (vo != OOB) {
        const u32x4 rv = u32x4{r.x, r.y, r.z, r.w};
        lds_write128<0>(pa, rv);
        if (p.bin_z != nullptr && tn == 0 && row >= hoff && row < hoff + BM) {
          const __amdgpu_buffer_rsrc_t r_z = __builtin_amdgcn_make_buffer_rsrc(p.bin_z, 0, p.in_bytes, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b128(rv, r_z, (int)vo, slab * 128, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  __builtin_amdgcn_s_waitcnt(0xc07f);          // retire the scalar loads the compiler still counts (see conv_dma_kernel.h)
  // ---- K loop: one barrier per step; the DMAs of the step NSTW - 1 ahead are issued before this step's fragment reads
  int st_c = 0, st_l = NSTW - 1;
  int c_t = 0, c_s = 0;                        // tap / slab being multiplied
  bool had_a = false;                          // this wave issued a slab piece in the previous step (still in flight)
  int pend_j = -1, pend_s = 0;                 // BNIN: that piece (landed and waited for one step later)
  for (int ks = 0; ks < nk; ++ks) {
    // my share of weight tile ks has landed -- and every slab piece I issued before it; the piece of the previous step sits
    // between the two weight tiles in flight and may stay in flight with the younger one
    if (had_a && !BNIN) wait_vmcnt<(NSTW - 2) * LB + 1>(); else wait_vmcnt<(NSTW - 2) * LB>();
    if constexpr (BNIN) {
      if (ks == 0) {
        for (int j = wave; j < AG; j += NW) xform(j, 0, 0);
      }
      if (pend_j >= 0) xform(pend_j, pend_s, pend_s & 1);
      pend_j = -1;
    }
    __builtin_amdgcn_s_barrier();
    // this wave's piece of the NEXT slab (piece c_t * NW + wave, while there are any), then the weight tile NSTW - 1 steps ahead
    {
      const int j = c_t * NW + wave;
      had_a = c_s + 1 < nslab && j < AG && c_t + 1 < ntaps;      // (the last tap's piece must have landed at the slab switch)
      if (c_s + 1 < nslab && j < AG) {
        dma16(r_in, smem + ((c_s + 1) & 1) * AB + j * 1024, a_voff(j), (unsigned)(c_s + 1) * 128u);
        if constexpr (BNIN) { pend_j = j; pend_s = c_s + 1; }
      }
      issue_w(st_l);
    }
    const int tp = __builtin_amdgcn_readlane(tapv, c_t);
    const int row0 = rbase + (tp & 0xffffff);
    const unsigned sw = (unsigned)(row0 >> 1) & 7u;
    const unsigned abase = lds0 + (unsigned)((c_s & 1) * AB) + (unsigned)row0 * 128u;
    const unsigned wbase = lds0 + (unsigned)(wring + st_c * WB);
    u32x4 fa[4][TMI], fw[4][TNI];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      FragLoad<0, TMI, 4096, 0>::run(fa[kk], abase + (((unsigned)(2 * kk + fhalf) ^ sw) << 4));
      FragLoad<0, TNI, 4096, 0>::run(fw[kk], wbase + boff[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      constexpr int PER = TMI + TNI;
      if (kk == 0) wait_chunk<(3 * PER > 15 ? 15 : 3 * PER)>(fa[0], fw[0], acc);
      if (kk == 1) wait_chunk<(2 * PER > 15 ? 15 : 2 * PER)>(fa[1], fw[1], acc);
      if (kk == 2) wait_chunk<(1 * PER > 15 ? 15 : 1 * PER)>(fa[2], fw[2], acc);
      if (kk == 3) wait_chunk<0>(fa[3], fw[3], acc);
#pragma unroll
      for (int j = 0; j < TNI; ++j)
#pragma unroll
        for (int i = 0; i < TMI; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[kk][j]),
                                                              __builtin_bit_cast(bf16x8, fa[kk][i]), acc[j][i], 0, 0, 0);
    }
    st_c = st_c + 1 == NSTW ? 0 : st_c + 1;
    st_l = st_l + 1 == NSTW ? 0 : st_l + 1;
    if (++c_t == ntaps) { c_t = 0; ++c_s; }
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  // ---- epilogue 1: accumulators -> bf16 tile in LDS (conv_dma_kernel.h)
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

  // ---- epilogue 2: read-back passes with the padded-grid row mapping
  const int ec = tid % TPR, er = tid / TPR;
  const int n = n0 + ec * 8;
  constexpr bool has_stats = (EM & 4) != 0, has_bnr = (EM & 8) != 0;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  {
    EpiCtx c;
    c.T = T; c.er = er; c.ec = ec; c.m0 = v0; c.M = p.Mv; c.Cout = p.Cout; c.n = n; c.ncol = n < p.Cout;
    c.sub_mul = 1; c.sub_py = 0; c.sub_px = 0; c.sub_hw = 1; c.sub_w = 1; c.out_H = p.H; c.out_W = p.W;
    c.vH = p.H; c.vW = p.W; c.vHpWp = HpWp; c.vWp = p.Wp; c.vinv_hpwp = inv_hpwp; c.vinv_wp = inv_wp;
    c.r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    c.r_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.addend), 0, p.out_bytes, 0x00020000);
    c.r_bny = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_y), 0, p.out_bytes, 0x00020000);
    c.r_msk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bn_mask), 0, p.out_bytes, 0x00020000);
    c.bias = p.bias; c.bn_coef = p.bn_coef; c.Kreal = p.Kreal;
    auto nostamp = []() {};
    epi_passes<EM, NPASS, RPP, TP, true>(nostamp, c, s1, s2);
  }
  if constexpr (has_stats) {
    float* red = reinterpret_cast<float*>(smem + BM * TP);     // [RPP rows][2][BN] (the host checks that it fits)
    {
      float* mine = red + er * 2 * BN + ec * 8;
      *reinterpret_cast<float4*>(mine) = make_float4(s1[0], s1[1], s1[2], s1[3]);
      *reinterpret_cast<float4*>(mine + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
      *reinterpret_cast<float4*>(mine + BN) = make_float4(s2[0], s2[1], s2[2], s2[3]);
      *reinterpret_cast<float4*>(mine + BN + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
    }
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
  if (has_stats && !has_bnr && p.fin.coef != nullptr) {
    // last-block-done finalize of the BatchNorm (conv_dma_kernel.h)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int mine = 0;
    if (tid == 0) mine = atomicAdd(p.fin_counter, 1u) == (unsigned)gridDim.x - 1u;
    const int is_last = __syncthreads_or(mine);
    if (is_last) {
      const int C = p.Kreal;
      for (int c = tid; c < C; c += NT) {
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < p.stats_rep; ++r) {
          t1 += __hip_atomic_load(p.stats + (size_t)r * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t2 += __hip_atomic_load(p.stats + (size_t)r * 2 * C + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float mean = t1 / p.fin.count;
        float var = t2 / p.fin.count - mean * mean;
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
}

}  // namespace pxl_halo
