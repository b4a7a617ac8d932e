// Hardware probes (gfx950): ds_read_b64_tr_b16 lane mapping, buffer_load..lds OOB fill and soffset range check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_tr(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  // lane l points at 8 bytes: elements 4l .. 4l+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * threadIdx.x));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (uint16_t)v[e];
}

__global__ void k_lds(const char* g, int nbytes, uint32_t* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* s32 = (uint32_t*)smem;
  for (int i = threadIdx.x; i < 512; i += 64) s32[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  int voff = threadIdx.x * 16;
  int soff = 0;
  if (mode == 1) { if (threadIdx.x & 1) voff = 0x7ffffff0; }              // OOB lanes via voffset
  if (mode == 2) { soff = nbytes - 512; }                                 // voff+soff crosses the end for lanes >= 32
  if (mode == 3) { voff = threadIdx.x * 16 + 1024; soff = -1024; }        // negative soffset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = s32[i];
}

int main() {
  uint16_t* d; hipMalloc(&d, 256 * 2);
  k_tr<<<1, 64>>>(d);
  std::vector<uint16_t> h(256);
  hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  printf("ds_read_tr16_b64: lane -> 4 source element indices (element i lives at lane i/4, slot i%%4)\n");
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  const int nbytes = 4096;
  char* g; hipMalloc(&g, nbytes + 4096);
  std::vector<uint32_t> src((nbytes + 4096) / 4);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (uint32_t)i;
  hipMemcpy(g, src.data(), src.size() * 4, hipMemcpyHostToDevice);
  uint32_t* o; hipMalloc(&o, 1024);
  std::vector<uint32_t> ho(256);
  for (int mode = 0; mode < 4; ++mode) {
    k_lds<<<1, 64, 4096>>>(g, nbytes, o, mode);
    hipMemcpy(ho.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("buffer_load_lds mode %d (first dword of each lane's 16B):", mode);
    for (int l = 0; l < 64; ++l) printf(" %x", ho[4 * l]);
    printf("\n");
  }
  return 0;
}
