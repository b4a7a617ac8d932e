// Shared device/host helpers for libpixelhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/pixelhip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef uint16_t bf16_t;   // storage type of bf16 activations / packed weights

// ----------------------------------------------------------------------------
// error plumbing (thread-local last error, negative return codes)
// ----------------------------------------------------------------------------
int pxl_set_error(int code, const char* fmt, ...);
int pxl_tune_get(int key);
#define PXL_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return pxl_set_error(PXL_ERR_HIP, "%s failed: %s (%s:%d)", #expr,            \
                           hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)
#define PXL_LAUNCH_CHECK() PXL_CHECK_HIP(hipGetLastError())
// PXL_HOST_TRACE=1: host nanoseconds per slot, printed when the process ends (csrc/net.cpp): where the executor's per-launch host time goes
namespace pxlht {
extern bool on;
long now();
void add(int slot, long ns);
}
#define PXL_REQUIRE(cond, ...)                                                     \
  do {                                                                             \
    if (!(cond)) return pxl_set_error(PXL_ERR_ARG, __VA_ARGS__);                   \
  } while (0)

// ----------------------------------------------------------------------------
// element traits: T = float (parity mode) or bf16_t (throughput mode)
// ----------------------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;          // elements per 16-byte chunk
  static constexpr int DTYPE = PXL_F32;
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int DTYPE = PXL_BF16;
};

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;   // RNE, lowers to v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// unpack one 16-byte chunk into EPC floats and back
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static __device__ __forceinline__ void unpack(const uint4& c, float* f) {
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
    f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]),
                      __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Chunk<bf16_t> {
  static __device__ __forceinline__ void unpack(const uint4& c, float* f) {
    f[0] = bf_lo(c.x); f[1] = bf_hi(c.x); f[2] = bf_lo(c.y); f[3] = bf_hi(c.y);
    f[4] = bf_lo(c.z); f[5] = bf_hi(c.z); f[6] = bf_lo(c.w); f[7] = bf_hi(c.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]),
                      pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
  }
};

// EPC consecutive per-channel floats (coefficients, sums) as 16-byte loads: p must be 16-byte aligned (channel chunk
// offsets are multiples of EPC >= 4 floats and every coefficient vector starts 16-byte aligned).  The scalar form
// `for e: f[e] = p[e]` compiles to EPC dword loads whose lanes are 32 bytes apart -- 16 cache lines per wave-load.
template <int EPC> __device__ __forceinline__ void load_cvec(const float* __restrict__ p, float* f) {
#pragma unroll
  for (int q = 0; q < EPC / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Kernel-argument prefetch for the large-argument contraction kernels.  hipcc fetches a 400-600 byte argument block in 3-4
// DEPENDENT batches (tile index -> geometry -> pointers -> taps), each a cold scalar-cache miss of 0.4-0.5 us right after
// the launch (tools/cbench --trace: 1.6-2.8 us between kernel entry and the first DMA).  This touches every 64-byte line in
// the same round trip as the compiler's first batch: one s_load per line + the wait in a single statement (the destination
// of an asm load is unprotected until its own wait: cdna_hip_programming.md 5.7 item 1).  Call it first thing in the kernel.
template <size_t BYTES> __device__ __forceinline__ void kernarg_touch() {
  static_assert(BYTES <= 768, "kernarg_touch: at most twelve 64-byte lines");
  auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  // every load lands in VCC (discarded; an SGPR picked by the allocator may still be the target of one of the compiler's
  // own argument loads in flight, which forces a wait BEFORE this statement and the second round trip it is meant to avoid)
#define PXL_KT(OFF) "s_load_dword vcc_lo, %0, " #OFF "\n\t"
  if constexpr (BYTES > 704)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180) PXL_KT(0x1c0) PXL_KT(0x200)
                 PXL_KT(0x240) PXL_KT(0x280) PXL_KT(0x2c0) "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else if constexpr (BYTES > 640)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180) PXL_KT(0x1c0) PXL_KT(0x200)
                 PXL_KT(0x240) PXL_KT(0x280) "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else if constexpr (BYTES > 576)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180) PXL_KT(0x1c0) PXL_KT(0x200)
                 PXL_KT(0x240) "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else if constexpr (BYTES > 512)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180) PXL_KT(0x1c0) PXL_KT(0x200)
                 "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else if constexpr (BYTES > 448)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180) PXL_KT(0x1c0)
                 "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else if constexpr (BYTES > 384)
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140) PXL_KT(0x180)
                 "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
  else
    asm volatile(PXL_KT(0x0) PXL_KT(0x40) PXL_KT(0x80) PXL_KT(0xc0) PXL_KT(0x100) PXL_KT(0x140)
                 "s_waitcnt lgkmcnt(0)" : : "s"(ka) : "vcc", "memory");
#undef PXL_KT
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// branch-free q = m / d, r = m % d for 0 <= m < 2^24 (pixel indices) using a float reciprocal
__device__ __forceinline__ void fast_divmod(int m, int d, float inv_d, int& q, int& r) {
  q = (int)((float)m * inv_d);
  r = m - q * d;
  if (r >= d) { ++q; r -= d; }
  if (r < 0) { --q; r += d; }
}

// Column-group geometry shared by the row-streaming kernels below: a block is CG 16-byte chunk columns
// (<= 16 chunks = 256 contiguous bytes of a row) x RL row lanes; blockIdx.x = column group, blockIdx.y = row
// group.  Each thread keeps its channel chunk (and the per-channel coefficients) in registers and walks rows.
struct ColGeom { int cg, rl, ncg; };
__host__ __device__ inline ColGeom col_geom(int C, int epc, int cgmax = 16) {
  const int cpr = C / epc;
  ColGeom g;
  g.cg = cpr < cgmax ? cpr : cgmax;
  while (256 % g.cg != 0 || cpr % g.cg != 0) --g.cg;   // channel pitches are multiples of 32 -> cg in {4, 8, 16}
  g.rl = 256 / g.cg;
  g.ncg = cpr / g.cg;
  return g;
}


// rows per row group so that the launch has about `target` blocks (each block streams >= 4 row-lane passes)
static inline int rows_per_group(int M, const ColGeom& g, int target) {
  int groups = target / g.ncg;
  if (groups < 1) groups = 1;
  int rpg = (M + groups - 1) / groups;
  const int min_rows = 4 * g.rl;
  if (rpg < min_rows) rpg = min_rows;
  return (rpg + g.rl - 1) / g.rl * g.rl;
}

// XCD-aware bijective remap of a linear block id (guide T1): consecutive logical
// tiles land on the same XCD (= same L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
