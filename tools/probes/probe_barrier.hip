// Probe (round 3, VERDICT "run the persistent experiment, record the A/B even if negative"): what does keeping a
// BatchNorm-style seam INSIDE one launch cost on the MI355X, against cutting the launch there?
//
// The seam: phase 1 produces a tensor and per-channel sums (a convolution with its statistics epilogue), phase 2 needs the
// COMPLETE sums before it can touch any element (BN apply).  Two forms, same arithmetic, same bytes produced:
//   (a) two launches: phase 1 writes y + atomics, the boundary is the barrier, phase 2 re-reads y, writes z;
//   (b) one launch with a grid barrier (monotonic arrival counter, one-lane agent-scope release before arriving,
//       relaxed sc1 poll + s_sleep, one-lane agent-scope acquire after): every workgroup keeps its slice of y in
//       registers across the barrier and writes only z -- the fusion a persistent layer kernel would buy (one write and
//       one read of the tensor less, no second launch).
// Sizes: the layer3 tensors of the step (8712 x 256 and 8712 x 1024 bf16), 256 / 512 workgroups of 256 threads.  Each form is
// timed alone and next to a second stream that keeps the chip busy with convolution-sized streaming kernels (the step
// always has a second network or the weight gradients in flight).  Spins are bounded: a lost barrier sets a flag and the
// kernel exits (results wrong, no hang).
//
//   hipcc --offload-arch=gfx950 -O3 -o probe_barrier probe_barrier.hip && ./probe_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned short bf16_t;
__device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pk(float a, float b) {
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
constexpr int C = 256;          // channels reduced over (sums[C])
constexpr int VPT = 16;         // 16-byte vectors per thread kept in registers by the fused form (64 VGPRs)

// phase 1: y = 0.5 * x + 1 (stand-in for the convolution output), per-channel sums by one atomic per thread-channel group
__global__ __launch_bounds__(256) void phase1(const uint4* __restrict__ x, uint4* __restrict__ y, float* __restrict__ sums, long nvec) {
  float s = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    uint4 v = x[i];
    unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (int k = 0; k < 4; ++k) { float a = 0.5f * lo(w[k]) + 1.f, b = 0.5f * hi(w[k]) + 1.f; s += a + b; w[k] = pk(a, b); }
    y[i] = v;
  }
  atomicAdd(sums + (threadIdx.x % C), s);
}
// phase 2: z = y * g(sums)  (needs the complete sums)
__global__ __launch_bounds__(256) void phase2(const uint4* __restrict__ y, uint4* __restrict__ z, const float* __restrict__ sums, long nvec) {
  const float g = 1.f / (1.f + 1e-9f * sums[threadIdx.x % C]);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
    uint4 v = y[i];
    unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = pk(lo(w[k]) * g, hi(w[k]) * g);
    z[i] = v;
  }
}
// (b) both phases in one launch; a workgroup's slice stays in registers across the barrier (nvec <= grid * 256 * VPT)
__global__ __launch_bounds__(256) void fused(const uint4* __restrict__ x, uint4* __restrict__ z, float* __restrict__ sums,
                                             unsigned* __restrict__ counter, unsigned* __restrict__ lost, long nvec, unsigned target) {
  uint4 keep[VPT];
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    const long i = (long)p * gridDim.x * 256L + blockIdx.x * 256L + threadIdx.x;
    if (i < nvec) {
      uint4 v = x[i];
      unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
      for (int k = 0; k < 4; ++k) { float a = 0.5f * lo(w[k]) + 1.f, b = 0.5f * hi(w[k]) + 1.f; s += a + b; w[k] = pk(a, b); }
      keep[p] = v;
    }
  }
  atomicAdd(sums + (threadIdx.x % C), s);
  // ---- grid barrier (guide: every wave drains, block barrier, ONE lane releases, arrives, polls relaxed, acquires)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) { *lost = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  const float g = 1.f / (1.f + 1e-9f * __hip_atomic_load(sums + (threadIdx.x % C), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
  for (int p = 0; p < VPT; ++p) {
    const long i = (long)p * gridDim.x * 256L + blockIdx.x * 256L + threadIdx.x;
    if (i < nvec) {
      uint4 v = keep[p];
      unsigned* w = reinterpret_cast<unsigned*>(&v);
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = pk(lo(w[k]) * g, hi(w[k]) * g);
      z[i] = v;
    }
  }
}
// the co-running stream: streaming kernels of a convolution's size and duration
__global__ __launch_bounds__(256) void noise(const uint4* __restrict__ a, uint4* __restrict__ b, long nvec, int reps) {
  for (int r = 0; r < reps; ++r)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += gridDim.x * 256L) {
      uint4 v = a[i]; v.x ^= r; b[i] = v;
    }
}

int main() {
  hipStream_t s, s2;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const long M = 8712;
  const size_t maxb = (size_t)M * 1024 * 2;
  uint4 *x, *y, *z, *na, *nb; float* sums; unsigned *counter, *lost;
  CHECK(hipMalloc(&x, maxb)); CHECK(hipMalloc(&y, maxb)); CHECK(hipMalloc(&z, maxb));
  CHECK(hipMalloc(&na, maxb)); CHECK(hipMalloc(&nb, maxb));
  CHECK(hipMalloc(&sums, C * 4)); CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&lost, 4));
  CHECK(hipMemset(x, 0x3c, maxb)); CHECK(hipMemset(na, 1, maxb)); CHECK(hipMemset(lost, 0, 4));
  const int REPS = 200;
  printf("%-14s %-6s %-9s | two launches (us) | one launch + grid barrier (us) | barrier lost\n", "tensor", "grid", "co-run");
  for (int ch : {256, 1024}) {
    const long nvec = M * ch * 2 / 16;
    for (int grid : {256, 512}) {
      if (nvec > (long)grid * 256 * VPT) continue;      // the fused form keeps the whole tensor in registers
      for (int corun = 0; corun < 2; ++corun) {
        float t[2];
        for (int form = 0; form < 2; ++form) {
          CHECK(hipMemsetAsync(counter, 0, 4, s));
          CHECK(hipStreamSynchronize(s));
          if (corun) hipLaunchKernelGGL(noise, dim3(512), dim3(256), 0, s2, na, nb, (long)(M * 1024 * 2 / 16), 4000);
          for (int w = 0; w < 20; ++w) {                 // warm-up (also lets the co-running kernel get going)
            if (form == 0) { hipLaunchKernelGGL(phase1, dim3(grid), dim3(256), 0, s, x, y, sums, nvec); hipLaunchKernelGGL(phase2, dim3(grid), dim3(256), 0, s, y, z, sums, nvec); }
            else hipLaunchKernelGGL(fused, dim3(grid), dim3(256), 0, s, x, z, sums, counter, lost, nvec, (unsigned)(grid * (w + 1)));
          }
          CHECK(hipEventRecord(e0, s));
          for (int r = 0; r < REPS; ++r) {
            if (form == 0) { hipLaunchKernelGGL(phase1, dim3(grid), dim3(256), 0, s, x, y, sums, nvec); hipLaunchKernelGGL(phase2, dim3(grid), dim3(256), 0, s, y, z, sums, nvec); }
            else hipLaunchKernelGGL(fused, dim3(grid), dim3(256), 0, s, x, z, sums, counter, lost, nvec, (unsigned)(grid * (20 + r + 1)));
          }
          CHECK(hipEventRecord(e1, s));
          CHECK(hipEventSynchronize(e1));
          CHECK(hipEventElapsedTime(&t[form], e0, e1));
          CHECK(hipDeviceSynchronize());
        }
        unsigned l = 0;
        CHECK(hipMemcpy(&l, lost, 4, hipMemcpyDeviceToHost));
        printf("8712 x %-6d %-6d %-9s | %17.2f | %30.2f | %u\n", ch, grid, corun ? "yes" : "no", 1e3 * t[0] / REPS, 1e3 * t[1] / REPS, l);
      }
    }
  }
  return 0;
}
