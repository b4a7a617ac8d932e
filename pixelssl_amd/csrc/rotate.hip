// S4L batch pre-handling (pixelssl/ssl_algorithm/ssl_s4l.py:296-355): [B][C][N][N] -> fp32 [2B][C][N][N] =
// the batch followed by one copy per sample rotated by a quarter-turn multiple.  The reference fills a zero float
// tensor sample by sample with numpy-style transposes / flips (`_rotate_tensor`); here ONE launch moves 32 x 32 tiles
// through LDS so that both the source rows and the destination rows are accessed with unit stride, whatever the angle.
#include "common.h"

namespace {

struct RotArgs {
  int B, C, N;
  unsigned long long codes[4];          // 2 bits per sample: angle index 0..3 (ssl_s4l.py:347-355), up to 128 samples
};

template <typename S>
__global__ __launch_bounds__(256) void rotate_append_kernel(const RotArgs p, const S* __restrict__ src,
                                                            float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int N = p.N;
  const int plane = blockIdx.z;                       // (sample of dst, channel)
  const int sd = plane / p.C, c = plane % p.C;
  const int ss = sd < p.B ? sd : sd - p.B;
  const int ang = sd < p.B ? 0 : (int)((p.codes[ss >> 5] >> ((ss & 31) * 2)) & 3ull);
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;        // destination tile origin
  // source tile origin (rows, columns) for the destination tile:
  //   angle 1 (clockwise):          out[i][j] = in[N-1-j][i]
  //   angle 2 (half turn):          out[i][j] = in[N-1-i][N-1-j]
  //   angle 3 (counter-clockwise):  out[i][j] = in[j][N-1-i]
  int rb, cb;
  switch (ang) {
    case 1: rb = N - 32 - j0; cb = i0; break;
    case 2: rb = N - 32 - i0; cb = N - 32 - j0; break;
    case 3: rb = j0; cb = N - 32 - i0; break;
    default: rb = i0; cb = j0; break;
  }
  const S* __restrict__ sp = src + ((size_t)ss * p.C + c) * N * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rr = ty + 8 * k;
    const int r = rb + rr, cc = cb + tx;
    float v = 0.f;
    if (r >= 0 && r < N && cc >= 0 && cc < N) v = (float)sp[(size_t)r * N + cc];
    tile[rr][tx] = v;
  }
  __syncthreads();
  float* __restrict__ dp = dst + ((size_t)sd * p.C + c) * N * N;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int a = ty + 8 * k, b = tx;                           // destination (i0 + a, j0 + b)
    if (i0 + a < N && j0 + b < N) {
      float v;
      switch (ang) {
        case 1: v = tile[31 - b][a]; break;
        case 2: v = tile[31 - a][31 - b]; break;
        case 3: v = tile[b][31 - a]; break;
        default: v = tile[a][b]; break;
      }
      dp[(size_t)(i0 + a) * N + j0 + b] = v;
    }
  }
}

}  // namespace

// src_kind: 0 = float32, 1 = int64, 2 = uint8 (what the loaders hand over: images, label maps).  angles: B HOST ints in
// 0..3 (np.random.randint(1, 4) in the reference; 0 = plain copy).  dst: fp32 [2B][C][N][N].
extern "C" int pxl_rotate_append(int src_kind, int B, int C, int N, const void* src, const int* angles, float* dst,
                                 void* stream) {
  PXL_REQUIRE(src && angles && dst && B > 0 && B <= 128 && C > 0 && N > 0, "rotate_append: bad argument (1 <= B <= 128)");
  PXL_REQUIRE(src_kind >= 0 && src_kind <= 2, "rotate_append: src_kind must be 0 (f32), 1 (i64) or 2 (u8)");
  PXL_REQUIRE((long)2 * B * C <= 65535, "rotate_append: 2*B*C exceeds the grid's z extent");
  RotArgs p;
  p.B = B; p.C = C; p.N = N;
  for (int k = 0; k < 4; ++k) p.codes[k] = 0ull;
  for (int i = 0; i < B; ++i) {
    PXL_REQUIRE(angles[i] >= 0 && angles[i] <= 3, "rotate_append: angle index %d of sample %d outside 0..3", angles[i], i);
    p.codes[i >> 5] |= (unsigned long long)angles[i] << ((i & 31) * 2);
  }
  const int t = cdiv(N, 32);
  const dim3 grid(t, t, 2 * B * C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (src_kind) {
    case 0: hipLaunchKernelGGL(rotate_append_kernel<float>, grid, dim3(256), 0, s, p, (const float*)src, dst); break;
    case 1: hipLaunchKernelGGL(rotate_append_kernel<long>, grid, dim3(256), 0, s, p, (const long*)src, dst); break;
    default: hipLaunchKernelGGL(rotate_append_kernel<unsigned char>, grid, dim3(256), 0, s, p, (const unsigned char*)src, dst);
  }
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
