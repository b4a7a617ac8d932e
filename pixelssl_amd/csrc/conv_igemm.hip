// Implicit-GEMM convolution for gfx950 (MI355X), NHWC activations, [K][taps][C] weights.
//
//   out[m][n] = sum_{t,c} A(m,t,c) * W[n][t][c]      m = (b,oy,ox), n = out channel
//   A(m,t,c)  = act(in[b][iy][ix][c])                iy = (oy*so + dy[t]) / div  (if divisible & in range)
//
// One kernel covers every dense contraction of the hot path (SURVEY.md 8a'):
//   * forward 1x1 / 3x3 (dilated, strided) / 7x7 stem / 4x4 / multi-rate ASPP (36 taps),
//   * data-gradient (so=1, negated taps, div=stride, transposed weights),
// with the producer's BatchNorm-apply(+ReLU) fused into the A-tile load (prologue) and
// bias / residual-addend / per-channel sum & sum-of-squares (BN statistics) fused into
// the epilogue.
//
// Mapping to CDNA4: 256 threads = 4 waves (2x2 or 4x1), each wave owns 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16 for bf16, v_mfma_f32_32x32x2_f32 for the exact-fp32 parity
// mode).  K is walked in 64- or 128-byte slices per row; global loads are 16 B per lane along
// the channel axis (NHWC => coalesced), register-staged (2-3 tiles in flight) so that the
// prologue can run, written to a double-buffered, XOR-swizzled LDS image (conflict-free
// ds_read_b128 for the 32-row fragment pattern), one barrier per K step.  blockIdx is remapped
// so that tiles sharing an A row-panel run on the same XCD (shared L2).
#include "common.h"

namespace {

struct ConvArgs {
  const void* in;
  const void* w;
  void* out;
  const float* in_scale;
  const float* in_shift;
  const float* bias;
  const void* addend;
  float* stats;
  float* ws;       // split-K fp32 accumulation buffer [M][Cout] (pre-zeroed), nullptr when splitk == 1
  int stats_rep;   // replicas of the [2*Kreal] statistics vector (spreads same-address atomics)
  int splitk, nk_per;
  int B, Hi, Wi, Cin;
  int Ho, Wo, Cout, Kreal;
  int ntaps, so, div_shift, div_mask, relu_in;
  int M;           // B*Ho*Wo
  int Ktot;        // ntaps*Cin
  int nk;          // K steps
  int tiles_m, tiles_n;
  int taps[64];    // (dy << 16) | (dx & 0xffff)
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// XOR swizzle of the 16-byte chunk index so that the ds_read_b128 of a 32-row MFMA fragment is
// conflict-free: 64-byte rows (4 chunks) -> chunk ^ (row>>2)&3 ; 128-byte rows (8 chunks) -> chunk ^ (row>>1)&7
template <int CH> __device__ __forceinline__ int swz(int row, int chunk) {
  if constexpr (CH == 4) return chunk ^ ((row >> 2) & 3);
  else return chunk ^ ((row >> 1) & 7);
}

// BKB   = bytes of K per row slice and K step (64 or 128)
// SIMPLE = 1x1 / stride 1 / no padding: A(m, c) = in[m][c], no tap table, no bounds logic
template <typename T, int BM, int BN, int WM, int WN, int BKB, bool SIMPLE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int CH = BKB / 16;               // 16-byte chunks per row slice
  constexpr int BK = CH * EPC;               // elements per K step
  constexpr int RP = 256 / CH;               // rows covered by one pass of the 256 loader threads
  constexpr int RA = (BM + RP - 1) / RP;     // A chunks per thread per K step
  constexpr int RB = (BN + RP - 1) / RP;
  constexpr int PF = (RA + RB) >= 8 ? 2 : 3; // register prefetch stages (tiles in flight)
  constexpr int TM = BM / (32 * WM);         // 32x32 tiles per wave along M
  constexpr int TN = BN / (32 * WN);
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: [2][BM][BKB] A | [2][BN][BKB] B | taps[64] int | affine [2*Cin] float
  unsigned char* sA = smem;
  unsigned char* sB = smem + 2 * BM * BKB;
  int* sTaps = reinterpret_cast<int*>(smem + 2 * (BM + BN) * BKB);
  float* sAff = reinterpret_cast<float*>(smem + 2 * (BM + BN) * BKB + 256);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const bool has_aff = p.in_scale != nullptr;
  if (!SIMPLE && tid < 64) sTaps[tid] = p.taps[tid];
  if (has_aff) {
    for (int i = tid; i < p.Cin; i += 256) {
      sAff[i] = p.in_scale[i];
      sAff[p.Cin + i] = p.in_shift[i];
    }
  }

  // ---- per-thread load coordinates
  const int chunk = tid % CH;
  const int lrow = tid / CH;                 // 0..RP-1
  int a_iy0[RA], a_ix0[RA], a_base[RA];      // SIMPLE: a_base = pixel index (or -1), others unused
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int row = lrow + RP * i;
    const int m = m0 + row;
    if (SIMPLE) {
      a_base[i] = (row < BM && m < p.M) ? m : -1;
      a_iy0[i] = a_ix0[i] = 0;
    } else if (row < BM && m < p.M) {
      const int b = m / HoWo;
      const int r = m - b * HoWo;
      const int oy = r / p.Wo;
      const int ox = r - oy * p.Wo;
      a_iy0[i] = oy * p.so;
      a_ix0[i] = ox * p.so;
      a_base[i] = b * p.Hi * p.Wi;
    } else {
      a_iy0[i] = -(1 << 28);
      a_ix0[i] = 0;
      a_base[i] = 0;
    }
  }
  // split-K slice of this block and the k cursor of this thread's chunk column: (tap, channel)
  const int ks_begin = blockIdx.y * p.nk_per;
  const int ks_end = min(p.nk, ks_begin + p.nk_per);
  int kt, kc;
  {
    const int kidx0 = ks_begin * BK + chunk * EPC;
    kt = kidx0 / p.Cin;
    kc = kidx0 - kt * p.Cin;
  }

  const T* __restrict__ gin = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.w);

  // PF register stages: tiles k+1 .. k+PF are in flight while tile k is multiplied, so a block keeps
  // PF global-load batches outstanding (the M = 8712 layers run ~2 blocks per CU and are latency-bound
  // with a single stage).  The prologue transform is applied when a stage is written to LDS, never at
  // issue time, so it does not serialise the load.
  uint4 ra[PF][RA], rb[PF][RB];
  int st_ok[PF], st_kc[PF];

  auto issue_loads = [&](int kstep, uint4 (&qa)[RA], uint4 (&qb)[RB], int& okmask, int& kc_saved) {
    const bool kvalid = kt < p.ntaps && kstep < ks_end;
    int dy = 0, dx = 0;
    if (!SIMPLE && kvalid) {
      const int tp = sTaps[kt];
      dy = tp >> 16;
      dx = (int)(short)(tp & 0xffff);
    }
    okmask = 0;
    kc_saved = kc;
    // Loads are UNCONDITIONAL (invalid lanes read element 0 and are zeroed when the stage is written to
    // LDS): a branch around a load makes hipcc wait vmcnt(0) right behind it, which serialises every
    // load of the stage (measured: 39 vmcnt(0) per loop body, 4-6x slower).
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      bool ok;
      size_t off;
      if (SIMPLE) {
        ok = kvalid && a_base[i] >= 0;
        off = (size_t)a_base[i] * p.Cin + kc;
      } else {
        const int ny = a_iy0[i] + dy, nx = a_ix0[i] + dx;
        const int iy = ny >> p.div_shift, ix = nx >> p.div_shift;
        ok = kvalid && ((ny & p.div_mask) == 0) && ((nx & p.div_mask) == 0) &&
             ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi) && (ny >= 0) && (nx >= 0);
        off = ((size_t)(a_base[i] + iy * p.Wi + ix)) * p.Cin + kc;
      }
      qa[i] = *reinterpret_cast<const uint4*>(gin + (ok ? off : (size_t)0));
      okmask |= (ok ? 1 : 0) << i;
    }
    const int kidx = kstep * BK + chunk * EPC;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = lrow + RP * i;
      const int n = n0 + row;
      const bool ok = row < BN && n < p.Kreal && kvalid;
      qb[i] = *reinterpret_cast<const uint4*>(gw + (ok ? (size_t)n * p.Ktot + kidx : (size_t)0));
      okmask |= (ok ? 1 : 0) << (16 + i);
    }
    // advance the k cursor by one K step (no data-dependent loop: hipcc drains vmcnt at loop headers)
    if (p.Cin >= BK) {                       // block-uniform: at most one tap boundary per step
      kc += BK;
      const bool wrap = kc >= p.Cin;
      kc = wrap ? kc - p.Cin : kc;
      kt = wrap ? kt + 1 : kt;
    } else {                                 // tiny Cin (7x7 stem): several taps per K step
      const int kl = (kstep + 1) * BK + chunk * EPC;
      kt = kl / p.Cin;
      kc = kl - kt * p.Cin;
    }
  };

  auto store_stage = [&](int buf, uint4 (&qa)[RA], uint4 (&qb)[RB], int okmask, int kc_saved) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = lrow + RP * i;
      uint4 v = qa[i];
      if (has_aff) {                        // block-uniform branch; the per-lane validity is a select below
        float f[EPC];
        Chunk<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float z = f[e] * sAff[kc_saved + e] + sAff[p.Cin + kc_saved + e];
          f[e] = p.relu_in ? fmaxf(z, 0.f) : z;
        }
        v = Chunk<T>::pack(f);
      }
      if (!((okmask >> i) & 1)) v = zero;   // zero padding / ragged edges stay exactly 0
      if (row < BM)
        *reinterpret_cast<uint4*>(sA + (buf * BM + row) * BKB + swz<CH>(row, chunk) * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = lrow + RP * i;
      const uint4 v = ((okmask >> (16 + i)) & 1) ? qb[i] : zero;
      if (row < BN)
        *reinterpret_cast<uint4*>(sB + (buf * BN + row) * BKB + swz<CH>(row, chunk) * 16) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();   // taps / affine visible
  // prologue: tile 0 -> LDS buffer 0; tiles 1..PF in flight in stages 1..PF-1, 0.  Tiles past ks_end are
  // all-zero (kvalid false), so the steady-state loop has NO conditionals: its trip count is rounded up
  // to a multiple of PF and the padding iterations multiply zero tiles.
  issue_loads(ks_begin, ra[0], rb[0], st_ok[0], st_kc[0]);
  store_stage(0, ra[0], rb[0], st_ok[0], st_kc[0]);
#pragma unroll
  for (int u = 1; u <= PF; ++u) issue_loads(ks_begin + u, ra[u % PF], rb[u % PF], st_ok[u % PF], st_kc[u % PF]);
  __syncthreads();

  const int frow = lane & 31;
  const int fhalf = lane >> 5;
  const int nsteps = (ks_end - ks_begin + PF - 1) / PF * PF;
  for (int j0 = 0; j0 < nsteps; j0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = j0 + u;               // tile index relative to ks_begin; successor lives in stage (u+1)%PF
      const int buf = j & 1;
#pragma unroll
      for (int kk = 0; kk < CH / 2; ++kk) {
        const int ch = fhalf + 2 * kk;
        uint4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 32 + frow;
          fa[i] = *reinterpret_cast<const uint4*>(sA + (buf * BM + row) * BKB + swz<CH>(row, ch) * 16);
        }
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int row = (wn * TN + jn) * 32 + frow;
          fb[jn] = *reinterpret_cast<const uint4*>(sB + (buf * BN + row) * BKB + swz<CH>(row, ch) * 16);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) Mfma<T>::run(fa[i], fb[jn], acc[i][jn]);
      }
      const int nx = (u + 1) % PF;        // static after unrolling (checked: no scratch in the resource report)
      store_stage(buf ^ 1, ra[nx], rb[nx], st_ok[nx], st_kc[nx]);
      issue_loads(ks_begin + j + 1 + PF, ra[nx], rb[nx], st_ok[nx], st_kc[nx]);
      __syncthreads();
    }
  }

  if (p.splitk > 1) {
    // partial sums of this K slice -> fp32 workspace; bias / conversion happen in splitk_finish_kernel
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + frow;
      if (n >= p.Kreal) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          if (m < p.M) atomicAdd(p.ws + (size_t)m * p.Cout + n, acc[i][j][r]);
        }
    }
    return;
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
  T* __restrict__ gout = reinterpret_cast<T*>(p.out);
  const T* __restrict__ gadd = reinterpret_cast<const T*>(p.addend);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + frow;
    const bool ncol = n < p.Cout;
    const float bv = (p.bias != nullptr && n < p.Kreal) ? p.bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        if (m < p.M && ncol) {
          float v = acc[i][j][r] + bv;
          const size_t o = (size_t)m * p.Cout + n;
          if (gadd != nullptr) v += to_f(gadd[o]);
          s1 += v;
          s2 += v * v;
          gout[o] = from_f<T>(v);
        }
      }
    }
    if (p.stats != nullptr) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (fhalf == 0 && n < p.Kreal) {
        float* rep = p.stats + (size_t)(tm % p.stats_rep) * 2 * p.Kreal;
        atomicAdd(rep + n, s1);
        atomicAdd(rep + p.Kreal + n, s2);
      }
    }
  }
}

// out[m][n] = T(ws[m][n] + bias[n]) for n < Kreal, 0 for padded channels
template <typename T>
__global__ void splitk_finish_kernel(long total, int Cout, int Kreal, const float* __restrict__ ws,
                                     const float* __restrict__ bias, T* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % Cout);
    float v = 0.f;
    if (n < Kreal) v = ws[i] + (bias ? bias[n] : 0.f);
    out[i] = from_f<T>(v);
  }
}

template <typename T, int BM, int BN, int WM, int WN, int BKB>
int launch_cfg(const ConvArgs& a, int want_split, size_t ws_bytes, bool simple, hipStream_t stream) {
  ConvArgs p = a;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  constexpr int BK = (BKB / 16) * Elem<T>::EPC;
  p.nk = cdiv(p.Ktot, BK);
  const int grid = p.tiles_m * p.tiles_n;
  // split-K: only for launches that cannot fill the chip and have a long reduction (ASPP: 69 tiles, K = 73728)
  int splitk = 1;
  const bool can_split = p.ws != nullptr && p.stats == nullptr && p.addend == nullptr &&
                         ws_bytes >= (size_t)p.M * p.Cout * sizeof(float);
  if (can_split) {
    if (want_split > 1) splitk = want_split;
    else if (want_split <= 0 && grid < 200 && p.nk >= 64) splitk = min(cdiv(640, grid), p.nk / 16);
    if (splitk > p.nk) splitk = p.nk;
    if (splitk < 1) splitk = 1;
  }
  p.nk_per = cdiv(p.nk, splitk);
  splitk = cdiv(p.nk, p.nk_per);
  p.splitk = splitk;
  if (splitk > 1) PXL_CHECK_HIP(hipMemsetAsync(p.ws, 0, (size_t)p.M * p.Cout * sizeof(float), stream));
  else p.ws = nullptr;
  const size_t smem = 2 * (BM + BN) * BKB + 256 + (p.in_scale ? 2 * (size_t)p.Cin * 4 : 0);
  if (smem > 64 * 1024) {
    static bool raised[2] = {false, false};     // opt in once per instantiation to > 64 KiB of dynamic LDS
    if (!raised[simple ? 1 : 0]) {
      if (simple) PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, WM, WN, BKB, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      else PXL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, BM, BN, WM, WN, BKB, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised[simple ? 1 : 0] = true;
    }
  }
  if (simple)
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, BKB, true>), dim3(grid, splitk), dim3(256), smem, stream, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, BKB, false>), dim3(grid, splitk), dim3(256), smem, stream, p);
  PXL_LAUNCH_CHECK();
  if (splitk > 1) {
    const long total = (long)p.M * p.Cout;
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((int)g), dim3(256), 0, stream, total, p.Cout, p.Kreal, p.ws,
                       p.bias, reinterpret_cast<T*>(p.out));
    PXL_LAUNCH_CHECK();
  }
  return PXL_OK;
}

// tile configurations: 0-3 walk K in 64-byte slices, 4-7 in 128-byte slices (half the barriers and
// address arithmetic per MFMA, twice the LDS)
template <typename T>
int launch_conv(const ConvArgs& a, int force_cfg, int want_split, size_t ws_bytes, bool simple, hipStream_t stream) {
  const int M = a.M, N = a.Cout;
  int cfg = force_cfg;
  if (cfg < 0) {
    if (N <= 32) cfg = 7;
    else if (N <= 64) cfg = (cdiv(M, 128) >= 512) ? 5 : 6;
    else {
      const long t128 = (long)cdiv(M, 128) * cdiv(N, 128);
      const long t12864 = (long)cdiv(M, 128) * cdiv(N, 64);
      if (t128 >= 512) cfg = 4;
      else if (t12864 >= 512) cfg = 5;
      else cfg = 6;
    }
  }
  switch (cfg) {
    case 0: return launch_cfg<T, 128, 128, 2, 2, 64>(a, want_split, ws_bytes, simple, stream);
    case 1: return launch_cfg<T, 128, 64, 2, 2, 64>(a, want_split, ws_bytes, simple, stream);
    case 2: return launch_cfg<T, 64, 64, 2, 2, 64>(a, want_split, ws_bytes, simple, stream);
    case 3: return launch_cfg<T, 128, 32, 4, 1, 64>(a, want_split, ws_bytes, simple, stream);
    case 4: return launch_cfg<T, 128, 128, 2, 2, 128>(a, want_split, ws_bytes, simple, stream);
    case 5: return launch_cfg<T, 128, 64, 2, 2, 128>(a, want_split, ws_bytes, simple, stream);
    case 6: return launch_cfg<T, 64, 64, 2, 2, 128>(a, want_split, ws_bytes, simple, stream);
    case 7: return launch_cfg<T, 128, 32, 4, 1, 128>(a, want_split, ws_bytes, simple, stream);
    default: return pxl_set_error(PXL_ERR_ARG, "conv_igemm: unknown tile config %d", cfg);
  }
}

}  // namespace

extern "C" int pxl_conv_dma_eligible(const pxl_conv_desc* d, const float* in_scale, const void* workspace);
extern "C" int pxl_conv_dma(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                            const void* addend, float* stats, void* workspace, size_t ws_bytes, void* stream);

// split-K epilogue over SLABS: the K slices wrote their partial tiles side by side (ws[slab][M][Cout], plain 16-byte stores, no
// atomics, no pre-zeroed buffer); out = T(sum of the slabs in index order + bias) -- the same bits on every run
template <typename T>
__global__ void splitk_finish_slabs_kernel(long chunks, int Cout, int Kreal, int nslab, const float* __restrict__ ws,
                                           const float* __restrict__ bias, T* __restrict__ out) {
  const long total = chunks * 4;
  for (long c = blockIdx.x * (long)blockDim.x + threadIdx.x; c < chunks; c += (long)gridDim.x * blockDim.x) {
    float4 v = *reinterpret_cast<const float4*>(ws + 4 * c);
    for (int s = 1; s < nslab; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(ws + (size_t)s * total + 4 * c);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int n = (int)((4 * c) % Cout);
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = n + e < Kreal ? f[e] + (bias ? bias[n + e] : 0.f) : 0.f;
      out[4 * c + e] = from_f<T>(f[e]);
    }
  }
}
extern "C" int pxl_splitk_finish_slabs(int dtype, long total, int Cout, int Kreal, int nslab, const float* ws, const float* bias,
                                       void* out, void* stream) {
  PXL_REQUIRE(total % 4 == 0 && Cout % 4 == 0 && nslab >= 1, "splitk_finish_slabs: bad geometry");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long chunks = total / 4;
  long g = (chunks + 255) / 256;
  if (g > 2048) g = 2048;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(splitk_finish_slabs_kernel<float>, dim3((int)g), dim3(256), 0, s, chunks, Cout, Kreal, nslab, ws, bias, (float*)out);
  else
    hipLaunchKernelGGL(splitk_finish_slabs_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, chunks, Cout, Kreal, nslab, ws, bias, (bf16_t*)out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

// split-K epilogue shared with the LDS-DMA kernel: out = T(ws + bias)
extern "C" int pxl_splitk_finish(int dtype, long total, int Cout, int Kreal, const float* ws, const float* bias, void* out,
                                 void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long g = (total + 255) / 256;
  if (g > 2048) g = 2048;
  if (dtype == PXL_F32)
    hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3((int)g), dim3(256), 0, s, total, Cout, Kreal, ws, bias, (float*)out);
  else
    hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, total, Cout, Kreal, ws, bias, (bf16_t*)out);
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}

extern "C" int pxl_conv_igemm(const pxl_conv_desc* d, const void* in, const void* w, void* out,
                              const float* in_scale, const float* in_shift, const float* bias,
                              const void* addend, float* stats, void* workspace, size_t ws_bytes,
                              void* stream) {
  PXL_REQUIRE(d && in && w && out, "conv_igemm: null argument");
  PXL_REQUIRE(d->dtype == PXL_F32 || d->dtype == PXL_BF16, "conv_igemm: bad dtype %d", d->dtype);
  const int epc = d->dtype == PXL_F32 ? 4 : 8;
  PXL_REQUIRE(d->Cin > 0 && d->Cin % epc == 0, "conv_igemm: Cin pitch %d must be a multiple of %d", d->Cin, epc);
  PXL_REQUIRE(d->ntaps >= 1 && d->ntaps <= 64, "conv_igemm: ntaps %d out of range", d->ntaps);
  PXL_REQUIRE(d->div == 1 || d->div == 2, "conv_igemm: div must be 1 or 2 (got %d)", d->div);
  PXL_REQUIRE(d->Kreal >= 1 && d->Kreal <= d->Cout, "conv_igemm: Kreal %d vs Cout pitch %d", d->Kreal, d->Cout);
  PXL_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_igemm: scale/shift must come together");
  PXL_REQUIRE((long)d->B * d->Hi * d->Wi * d->Cin < (1L << 31) && (long)d->B * d->Ho * d->Wo * d->Cout < (1L << 31),
              "conv_igemm: tensor too large for 32-bit indexing");
  // plain bf16 operands take the LDS-DMA kernel (conv_dma.hip); tile_cfg 0..7 forces this generic kernel
  if (d->tile_cfg < 0 || d->tile_cfg >= 8) {
    if (pxl_conv_dma_eligible(d, in_scale, workspace)) return pxl_conv_dma(d, in, w, out, bias, addend, stats, workspace, ws_bytes, stream);
    PXL_REQUIRE(d->tile_cfg < 8, "conv_igemm: tile config %d needs plain bf16 operands with Cin %% 64 == 0", d->tile_cfg);
  }
  ConvArgs a;
  a.in = in; a.w = w; a.out = out;
  a.in_scale = in_scale; a.in_shift = in_shift; a.bias = bias; a.addend = addend; a.stats = stats;
  a.ws = reinterpret_cast<float*>(workspace);
  a.stats_rep = d->stats_rep >= 1 ? d->stats_rep : 1;
  a.splitk = 1; a.nk_per = 0;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.Kreal = d->Kreal;
  a.ntaps = d->ntaps; a.so = d->out_stride;
  a.div_shift = d->div == 2 ? 1 : 0; a.div_mask = d->div - 1;
  a.relu_in = d->relu_in;
  a.M = d->B * d->Ho * d->Wo;
  a.Ktot = d->ntaps * d->Cin;
  for (int t = 0; t < 64; ++t)
    a.taps[t] = t < d->ntaps ? (((int)d->dy[t]) << 16) | (((int)d->dx[t]) & 0xffff) : 0;
  a.nk = 0; a.tiles_m = a.tiles_n = 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // 1x1 / stride 1 / unpadded convolutions (and their data gradients) skip the whole gather logic
  const bool simple = d->ntaps == 1 && d->out_stride == 1 && d->div == 1 && d->dy[0] == 0 && d->dx[0] == 0 &&
                      d->Hi == d->Ho && d->Wi == d->Wo;
  if (d->dtype == PXL_F32) return launch_conv<float>(a, d->tile_cfg, d->split_k, ws_bytes, simple, s);
  return launch_conv<bf16_t>(a, d->tile_cfg, d->split_k, ws_bytes, simple, s);
}
