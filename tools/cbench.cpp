// tools/cbench -- torch-free micro-benchmark and timeline probe for the contraction kernels of libpixelhip.
//
// Tuning aid, not part of the product path.  Every ResNet-101 / DeepLab-v2 convolution shape at the BASELINE batch
// (8 x 513 x 513 -> 129 / 65 / 33 feature maps) ALONE on the GPU, per tile configuration, through the C-ABI
// (pxl_conv_igemm); plus
//   --dual            the same launch on two streams at once (what the MT step does with student || teacher)
//   --trace S:CFG[:M] cycle-stamp timeline of one launch (pxl_conv_dma_trace): launch skew, prologue, first tile,
//                     K-step cadence, drain, epilogue; CU census (which CUs ran how many workgroups)
//   --floor           what the chip does with NO arithmetic: an empty launch of the same grid / LDS footprint, and the
//                     tile stream of a 1x1 convolution (same DMA instructions, ring depth 2 .. 6) without MFMAs
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/cbench.cpp -o tools/cbench -Lpixelssl_amd -lpixelhip \
//        -Wl,-rpath,'$ORIGIN/../pixelssl_amd'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pixelhip.h"

extern "C" int pxl_conv_dma_trace(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias,
                                  float* stats, unsigned* trace, void* stream);

#define CK(e)                                                                                         \
  do {                                                                                                \
    hipError_t _e = (e);                                                                              \
    if (_e != hipSuccess) { fprintf(stderr, "%s failed: %s (%s:%d)\n", #e, hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } \
  } while (0)

struct Shape { const char* name; int cin, cout, k, s, d, H, cnt; };
static const Shape SHAPES[] = {
    {"l1.1x1a", 64, 64, 1, 1, 1, 129, 1},     {"l1.3x3", 64, 64, 3, 1, 1, 129, 3},      {"l1.1x1b", 64, 256, 1, 1, 1, 129, 4},
    {"l1.1x1c", 256, 64, 1, 1, 1, 129, 2},    {"l2.1x1a", 256, 128, 1, 1, 1, 129, 1},   {"l2.3x3s2", 128, 128, 3, 2, 1, 129, 1},
    {"l2.1x1b", 128, 512, 1, 1, 1, 65, 4},    {"l2.ds", 256, 512, 1, 2, 1, 129, 1},     {"l2.1x1c", 512, 128, 1, 1, 1, 65, 3},
    {"l2.3x3", 128, 128, 3, 1, 1, 65, 3},     {"l3.1x1a", 512, 256, 1, 1, 1, 65, 1},    {"l3.3x3s2", 256, 256, 3, 2, 1, 65, 1},
    {"l3.1x1b", 256, 1024, 1, 1, 1, 33, 23},  {"l3.ds", 512, 1024, 1, 2, 1, 65, 1},     {"l3.1x1c", 1024, 256, 1, 1, 1, 33, 22},
    {"l3.3x3", 256, 256, 3, 1, 1, 33, 22},    {"l4.1x1a", 1024, 512, 1, 1, 1, 33, 1},   {"l4.3x3d2", 512, 512, 3, 1, 2, 33, 1},
    {"l4.1x1b", 512, 2048, 1, 1, 1, 33, 3},   {"l4.ds", 1024, 2048, 1, 1, 1, 33, 1},    {"l4.1x1c", 2048, 512, 1, 1, 1, 33, 2},
    {"l4.3x3d4", 512, 512, 3, 1, 4, 33, 1},
    // GCT flaw detector (4x4 kernels, pad 1; ssl_gct.py:539-585): only with --only fd
    {"fd.conv2", 64, 128, 4, 2, 1, 256, 0},   {"fd.conv2_1", 128, 128, 4, 1, 1, 128, 0}, {"fd.conv3", 128, 256, 4, 2, 1, 127, 0},
    {"fd.conv3_1", 256, 256, 4, 1, 1, 63, 0}, {"fd.conv4", 256, 512, 4, 2, 1, 62, 0},    {"fd.conv4_1", 512, 512, 4, 1, 1, 31, 0},
};

static int pitch(int c) { return c <= 8 ? 8 : (c + 31) / 32 * 32; }

static uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static bool g_f32 = false;       // --f32: fp32 operands (conv_dma_f32.hip / conv_wgrad_dma_f32.hip), timing modes only
static void* dev_random_bf16(size_t n, float scale, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  if (g_f32) {
    std::vector<float> hf(n);
    for (size_t i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      hf[i] = (((s >> 8) & 0xffff) / 65536.f + ((s >> 24) & 0xff) / 256.f - 1.0f) * scale;
    }
    void* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, hf.data(), n * 4, hipMemcpyHostToDevice));
    return d;
  }
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const float u = ((s >> 8) & 0xffff) / 65536.f + ((s >> 24) & 0xff) / 256.f - 1.0f;      // roughly triangular in (-1, 1)
    h[i] = f2bf(u * scale);
  }
  void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

struct Problem {
  pxl_conv_desc fwd, bwd;
  void *x, *y, *wf, *wt; float* stats; float* dw; int creal;
  size_t nx, ny; double flops; int M;
};

static Problem make_problem(const Shape& sh, int B, unsigned seed) {
  Problem p; memset(&p, 0, sizeof(p));
  const int pad = sh.k > 1 ? sh.d * (sh.k - 1) / 2 : 0;
  const int Ho = (sh.H + 2 * pad - sh.d * (sh.k - 1) - 1) / sh.s + 1;
  const int cip = pitch(sh.cin), cop = pitch(sh.cout);
  pxl_conv_desc f; memset(&f, 0, sizeof(f));
  f.dtype = g_f32 ? PXL_F32 : PXL_BF16; f.B = B; f.Hi = sh.H; f.Wi = sh.H; f.Cin = cip; f.Ho = Ho; f.Wo = Ho; f.Cout = cop; f.Kreal = sh.cout;
  f.ntaps = sh.k * sh.k; f.out_stride = sh.s; f.div = 1; f.relu_in = 0; f.tile_cfg = -1; f.stats_rep = 4; f.split_k = 1;
  for (int r = 0; r < sh.k; ++r)
    for (int c = 0; c < sh.k; ++c) { f.dy[r * sh.k + c] = (int16_t)(r * sh.d - pad); f.dx[r * sh.k + c] = (int16_t)(c * sh.d - pad); }
  p.fwd = f;
  pxl_conv_desc b = f;
  b.Hi = Ho; b.Wi = Ho; b.Cin = cop; b.Ho = sh.H; b.Wo = sh.H; b.Cout = cip; b.Kreal = sh.cin; b.out_stride = 1; b.div = sh.s; b.stats_rep = 1;
  for (int t = 0; t < f.ntaps; ++t) { b.dy[t] = (int16_t)(-f.dy[t]); b.dx[t] = (int16_t)(-f.dx[t]); }
  p.bwd = b;
  p.nx = (size_t)B * sh.H * sh.H * cip; p.ny = (size_t)B * Ho * Ho * cop;
  p.x = dev_random_bf16(p.nx, 1.0f, seed + 1);
  p.y = dev_random_bf16(p.ny, 1.0f, seed + 2);
  p.wf = dev_random_bf16((size_t)sh.cout * f.ntaps * cip, 0.05f, seed + 3);
  p.wt = dev_random_bf16((size_t)sh.cin * f.ntaps * cop, 0.05f, seed + 4);
  CK(hipMalloc((void**)&p.stats, 4 * 2 * (size_t)cop * sizeof(float)));
  CK(hipMemset(p.stats, 0, 4 * 2 * (size_t)cop * sizeof(float)));
  CK(hipMalloc((void**)&p.dw, (size_t)sh.cout * f.ntaps * cip * sizeof(float)));
  CK(hipMemset(p.dw, 0, (size_t)sh.cout * f.ntaps * cip * sizeof(float)));
  p.creal = sh.cin;
  p.flops = 2.0 * B * Ho * Ho * (double)sh.cout * sh.k * sh.k * sh.cin;
  p.M = B * Ho * Ho;
  return p;
}
static void free_problem(Problem& p) { (void)hipFree(p.x); (void)hipFree(p.y); (void)hipFree(p.wf); (void)hipFree(p.wt); (void)hipFree(p.stats); (void)hipFree(p.dw); }

static int launch_wgrad(const Problem& p, int cfg, hipStream_t s) {
  pxl_conv_desc q = p.fwd; q.tile_cfg = cfg;
  return pxl_conv_wgrad(&q, p.x, nullptr, nullptr, p.y, p.dw, p.creal, p.fwd.Cin, s);
}
static double time_wgrad(const Problem& p, int cfg, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  if (launch_wgrad(p, cfg, s) != PXL_OK) return -1.0;
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if (launch_wgrad(p, cfg, s) != PXL_OK) return -1.0;
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms * 1e3 / iters);
  }
  return best;
}
// --splitk N: the forward launch WITHOUT statistics as N K-slices through the library's split-K path (one fp32 partial-sum slab
// per slice + pxl_splitk_finish_slabs; PXL_CBENCH_SPLITK_ATOMICS=1: one shared buffer and fp32 atomics, the form of rounds 1-5)
// -- what a split-K of the sub-one-wave grids costs before any last-arriver epilogue is built;
// N = 1: the same launch unsplit (no statistics either), the like-for-like baseline
static int g_splitk = 0;
static void* g_ws = nullptr; static size_t g_ws_bytes = 0;
static int launch(const Problem& p, bool dgrad, int cfg, hipStream_t s) {
  if (!dgrad && g_splitk > 0) {
    pxl_conv_desc q = p.fwd; q.tile_cfg = cfg; q.split_k = g_splitk;
    const size_t need = (size_t)g_splitk * p.M * q.Cout * sizeof(float);      // one partial-sum slab per slice: plain stores, no atomics
    if (getenv("PXL_CBENCH_SPLITK_ATOMICS")) { const size_t one = (size_t)p.M * q.Cout * sizeof(float); if (g_ws_bytes < need) { if (g_ws) (void)hipFree(g_ws); CK(hipMalloc(&g_ws, need)); g_ws_bytes = need; }
      return pxl_conv_igemm(&q, p.x, p.wf, p.y, nullptr, nullptr, nullptr, nullptr, nullptr, g_splitk > 1 ? g_ws : nullptr, g_splitk > 1 ? one : 0, s); }
    if (g_ws_bytes < need) { if (g_ws) (void)hipFree(g_ws); CK(hipMalloc(&g_ws, need)); g_ws_bytes = need; }
    return pxl_conv_igemm(&q, p.x, p.wf, p.y, nullptr, nullptr, nullptr, nullptr, nullptr, g_splitk > 1 ? g_ws : nullptr, g_splitk > 1 ? g_ws_bytes : 0, s);
  }
  if (!dgrad) { pxl_conv_desc q = p.fwd; q.tile_cfg = cfg;
    return pxl_conv_igemm(&q, p.x, p.wf, p.y, nullptr, nullptr, nullptr, nullptr, p.stats, nullptr, 0, s); }
  pxl_conv_desc q = p.bwd; q.tile_cfg = cfg;
  return pxl_conv_igemm(&q, p.y, p.wt, p.x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s);
}

// best-of-3 average time (us) of `iters` back-to-back launches
static double time_single(const Problem& p, bool dgrad, int cfg, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  if (launch(p, dgrad, cfg, s) != PXL_OK) return -1.0;
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if (launch(p, dgrad, cfg, s) != PXL_OK) return -1.0;
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms * 1e3 / iters);
  }
  return best;
}

// two streams, two operand sets, `iters` launches each: wall time per PAIR of launches (us)
static double time_dual(const Problem& p0, const Problem& p1, bool dgrad, int cfg, int iters, hipStream_t s0, hipStream_t s1) {
  hipEvent_t a, b0, b1; CK(hipEventCreate(&a)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  launch(p0, dgrad, cfg, s0); launch(p1, dgrad, cfg, s1);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s0));
    CK(hipStreamWaitEvent(s1, a, 0));
    for (int i = 0; i < iters; ++i) { launch(p0, dgrad, cfg, s0); launch(p1, dgrad, cfg, s1); }
    CK(hipEventRecord(b0, s0)); CK(hipEventRecord(b1, s1));
    CK(hipEventSynchronize(b0)); CK(hipEventSynchronize(b1));
    float m0, m1; CK(hipEventElapsedTime(&m0, a, b0)); CK(hipEventElapsedTime(&m1, a, b1));
    best = std::min(best, (double)std::max(m0, m1) * 1e3 / iters);
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b0); (void)hipEventDestroy(b1);
  return best;
}

static float bf2f_h(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
// max |a - b| / max |b| of the tensor a launch writes (forward: y, data gradient: x) against the generic register-staged
// kernel (tile_cfg 1: independent code path), and of the statistics (forward)
static void check_cfg(Problem& p, bool dgrad, int cfg, hipStream_t s) {
  void* dst = dgrad ? p.x : p.y;
  const size_t n = dgrad ? p.nx : p.ny;
  const int C2 = 4 * 2 * p.fwd.Cout;
  std::vector<uint16_t> ref(n), got(n);
  std::vector<float> sref(C2), sgot(C2);
  // the data gradient overwrites x (its output) -- and x is not an input of the data gradient, so running twice is fine
  CK(hipMemsetAsync(p.stats, 0, C2 * sizeof(float), s));
  if (launch(p, dgrad, 1, s) != PXL_OK) { printf(" chk:%d REF-ERR(%s)", cfg, pxl_last_error()); return; }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(ref.data(), dst, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(sref.data(), p.stats, C2 * 4, hipMemcpyDeviceToHost));
  CK(hipMemsetAsync(dst, 0xff, n * 2, s)); CK(hipMemsetAsync(p.stats, 0, C2 * sizeof(float), s));
  if (launch(p, dgrad, cfg, s) != PXL_OK) { printf(" chk:%d ERR(%s)", cfg, pxl_last_error()); return; }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(got.data(), dst, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(sgot.data(), p.stats, C2 * 4, hipMemcpyDeviceToHost));
  double mx = 0, md = 0; size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const double a = bf2f_h(got[i]), b = bf2f_h(ref[i]);
    if (!(a == a)) { ++bad; continue; }
    mx = std::max(mx, std::fabs(b)); md = std::max(md, std::fabs(a - b));
  }
  // statistics: fold the replicas
  double smx = 0, smd = 0;
  if (!dgrad) {
    const int C = p.fwd.Kreal;
    for (int w = 0; w < 2; ++w) for (int c = 0; c < C; ++c) {
      double a = 0, b = 0;
      for (int r = 0; r < 4; ++r) { a += sgot[(size_t)r * 2 * C + w * C + c]; b += sref[(size_t)r * 2 * C + w * C + c]; }
      smx = std::max(smx, std::fabs(b)); smd = std::max(smd, std::fabs(a - b));
    }
  }
  printf(" chk:%d %.1e/%.1e%s", cfg, mx > 0 ? md / mx : md, smx > 0 ? smd / smx : smd, bad ? " NAN!" : "");
}

extern "C" void pxl_dma_capture_begin(void** slot);
extern "C" void pxl_dma_capture_end(void);
extern "C" int pxl_dma_launch_captured(void* slot0, void* slot1);
// the forward convolution of TWO operand sets as one paired launch (what pxl_net_forward_pair issues for student || teacher)
static int launch_pair(const Problem& p0, const Problem& p1, int cfg, hipStream_t s) {
  void* sl[2] = {nullptr, nullptr};
  pxl_dma_capture_begin(&sl[0]); int rc = launch(p0, false, cfg, s); pxl_dma_capture_end();
  pxl_dma_capture_begin(&sl[1]); if (rc == PXL_OK) rc = launch(p1, false, cfg, s); pxl_dma_capture_end();
  const int pr = pxl_dma_launch_captured(sl[0], sl[1]);
  return rc != PXL_OK ? rc : (pr < 0 ? pr : (pr == 1 ? PXL_OK : -100));
}
static double time_pair(const Problem& p0, const Problem& p1, int cfg, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  if (launch_pair(p0, p1, cfg, s) != PXL_OK) return -1.0;
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if (launch_pair(p0, p1, cfg, s) != PXL_OK) return -1.0;
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms * 1e3 / iters);
  }
  return best;
}

// ---------------------------------------------------------------------------------------------------------------------
// timeline analysis
// ---------------------------------------------------------------------------------------------------------------------
static double pct(std::vector<double> v, double q) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  const size_t i = std::min(v.size() - 1, (size_t)(q * (v.size() - 1) + 0.5));
  return v[i];
}
static void row(const char* what, const std::vector<double>& v, double ns_per_cyc) {
  printf("    %-34s med %8.0f cyc (%6.2f us)   p90 %8.0f   max %8.0f\n", what, pct(v, 0.5), pct(v, 0.5) * ns_per_cyc * 1e-3, pct(v, 0.9),
         pct(v, 1.0));
}

struct TraceSet { std::vector<unsigned> w; int nwg; };

static void analyze(const TraceSet& t, const char* title, const TraceSet* other) {
  const int W = 72;
  unsigned long long rmin = ~0ull, rmax = 0;
  std::vector<double> pro, first, steps, drain, stage, pass0, pass1, store, statb, statv, ackv, total, skew, mhz;
  std::map<unsigned, int> cu_count;
  for (int g = 0; g < t.nwg; ++g) {
    const unsigned* r = &t.w[(size_t)g * W];
    const int ns = (int)r[64]; const int nk = (int)r[71];
    if (ns < 5) continue;
    const unsigned long long r0 = r[67] | ((unsigned long long)r[68] << 32), r1 = r[69] | ((unsigned long long)r[70] << 32);
    rmin = std::min(rmin, r0); rmax = std::max(rmax, r1);
  }
  for (int g = 0; g < t.nwg; ++g) {
    const unsigned* r = &t.w[(size_t)g * W];
    const int ns = std::min((int)r[64], 64); const int nk = (int)r[71];
    if (ns < 5) continue;
    const int nkst = std::min(nk, 52);                           // stamped K steps
    auto d = [&](int i, int j) { return (double)(unsigned)(r[j] - r[i]); };
    pro.push_back(d(0, 1));
    first.push_back(d(1, 2));
    if (nkst > 1) steps.push_back(d(2, 1 + nkst) / (nkst - 1));
    if (2 + nkst < ns) drain.push_back(d(1 + nkst, 2 + nkst));
    if (3 + nkst < ns) stage.push_back(d(2 + nkst, 3 + nkst));
    if (4 + nkst < ns) pass0.push_back(d(3 + nkst, 4 + nkst));
    if (5 + nkst < ns) pass1.push_back(d(4 + nkst, 5 + nkst));
    if (6 + nkst < ns) store.push_back(d(5 + nkst, 6 + nkst));
    if (7 + nkst < ns) statb.push_back(d(6 + nkst, 7 + nkst));
    if (8 + nkst < ns) statv.push_back(d(7 + nkst, 8 + nkst));
    if (9 + nkst < ns) ackv.push_back(d(8 + nkst, 9 + nkst));
    total.push_back(d(0, ns - 1));
    const unsigned long long r0 = r[67] | ((unsigned long long)r[68] << 32), r1 = r[69] | ((unsigned long long)r[70] << 32);
    skew.push_back((double)(r0 - rmin) * 10.0);                  // ns
    if (r1 > r0 + 50) mhz.push_back(d(0, ns - 1) / ((double)(r1 - r0) * 10.0) * 1e3);
    const unsigned key = (r[66] & 0xf) << 16 | ((r[65] >> 8) & 0xff);      // xcc, (se, sh, cu)
    cu_count[key]++;
  }
  const double clk = pct(mhz, 0.5);                               // MHz (cycles per us)
  const double nspc = clk > 0 ? 1e3 / clk : 0.42;
  printf("  -- %s: %d workgroups, s_memtime clock ~%.0f MHz, first entry -> last exit %.2f us\n", title, t.nwg, clk,
         (double)(rmax - rmin) * 0.01);
  printf("    %-34s med %8.2f us   p90 %8.2f   max %8.2f\n", "entry skew (vs first workgroup)", pct(skew, 0.5) * 1e-3, pct(skew, 0.9) * 1e-3,
         pct(skew, 1.0) * 1e-3);
  row("prologue (args, addresses, issue)", pro, nspc);
  row("first K step (first tiles land)", first, nspc);
  row("K step, steady state (per step)", steps, nspc);
  row("drain + barrier", drain, nspc);
  row("accumulators -> staged tile", stage, nspc);
  row("read-back pass 0 (dispatch + loads)", pass0, nspc);
  row("read-back pass 1", pass1, nspc);
  row("remaining passes (stores issued)", store, nspc);
  row("statistics parked + barrier", statb, nspc);
  row("statistics summed, atomics issued", statv, nspc);
  row("stores / atomics acknowledged", ackv, nspc);
  row("workgroup total", total, nspc);
  // first-round workgroups (entered within 1 us of the launch) against the rest: cold vs warm instruction / scalar caches
  {
    std::vector<double> e_pro, l_pro, e_store, l_store, e_stat, l_stat, e_stage, l_stage;
    size_t k = 0;
    for (int g = 0; g < t.nwg; ++g) {
      const unsigned* r = &t.w[(size_t)g * W];
      const int ns = std::min((int)r[64], 64); const int nk = (int)r[71];
      if (ns < 5) continue;
      const int nkst = std::min(nk, 52);
      auto d = [&](int i, int j) { return (double)(unsigned)(r[j] - r[i]); };
      const bool early = skew[k++] < 1000.0;
      (early ? e_pro : l_pro).push_back(d(0, 1));
      if (3 + nkst < ns) (early ? e_stage : l_stage).push_back(d(2 + nkst, 3 + nkst));
      if (6 + nkst < ns) (early ? e_store : l_store).push_back(d(3 + nkst, 6 + nkst));
      if (8 + nkst < ns) (early ? e_stat : l_stat).push_back(d(6 + nkst, 8 + nkst));
    }
    if (!l_pro.empty())
      printf("    first round (%zu wgs) vs later (%zu): prologue %.0f / %.0f, staging %.0f / %.0f, passes %.0f / %.0f, statistics %.0f / %.0f cyc (medians)\n",
             e_pro.size(), l_pro.size(), pct(e_pro, 0.5), pct(l_pro, 0.5), pct(e_stage, 0.5), pct(l_stage, 0.5), pct(e_store, 0.5), pct(l_store, 0.5),
             pct(e_stat, 0.5), pct(l_stat, 0.5));
  }
  std::map<int, int> hist;
  for (auto& kv : cu_count) hist[kv.second]++;
  printf("    CUs used %zu; workgroups per CU:", cu_count.size());
  for (auto& kv : hist) printf(" %dx:%d", kv.first, kv.second);
  printf("\n");
  if (other) {
    std::map<unsigned, int> oc;
    for (int g = 0; g < other->nwg; ++g) {
      const unsigned* r = &other->w[(size_t)g * W];
      if (r[64] < 5) continue;
      oc[(r[66] & 0xf) << 16 | ((r[65] >> 8) & 0xff)]++;
    }
    int both = 0, only_a = 0, only_b = 0;
    for (auto& kv : cu_count) (oc.count(kv.first) ? both : only_a)++;
    for (auto& kv : oc) if (!cu_count.count(kv.first)) only_b++;
    printf("    co-running launch: CUs shared %d, only this %d, only other %d\n", both, only_a, only_b);
  }
}

static TraceSet run_trace(const Problem& p, int cfg, hipStream_t s, int* grid_out) {
  // grid is not known here: allocate for the smallest tile (64 x 64)
  const int maxwg = ((p.M + 63) / 64) * ((p.fwd.Cout + 63) / 64) + 8;
  unsigned* d; CK(hipMalloc((void**)&d, (size_t)maxwg * 72 * 4)); CK(hipMemset(d, 0, (size_t)maxwg * 72 * 4));
  pxl_conv_desc q = p.fwd; q.tile_cfg = cfg;
  const int rc = pxl_conv_dma_trace(&q, p.x, p.wf, p.y, nullptr, p.stats, d, s);
  TraceSet t; t.nwg = 0;
  if (rc != PXL_OK) { printf("  trace launch failed: %s\n", pxl_last_error()); (void)hipFree(d); return t; }
  CK(hipStreamSynchronize(s));
  t.w.resize((size_t)maxwg * 72);
  CK(hipMemcpy(t.w.data(), d, t.w.size() * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  int n = 0;
  for (int g = 0; g < maxwg; ++g) if (t.w[(size_t)g * 72 + 64] >= 5) n = g + 1;
  t.nwg = n;
  if (grid_out) *grid_out = n;
  return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// floor kernels: no arithmetic
// ---------------------------------------------------------------------------------------------------------------------
struct BigArgs { const void* a; const void* w; unsigned* sink; int K2, tiles_n, nk, pad; int filler[100]; };   // ~ the size of the conv's kernarg

__global__ __launch_bounds__(256) void null_kernel(const BigArgs p) {
  extern __shared__ unsigned char smem[];
  if (p.sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p.sink[0] = smem[0] + p.filler[3];
}

typedef __attribute__((address_space(3))) void lds_void;
// the tile stream of a 1x1 convolution: workgroup (tm, tn) pulls rows [tm*BM, +BM) of A and [tn*BN, +BN) of W, RB bytes
// of every row per step, through an NST-deep LDS ring by `buffer_load ... lds` -- and does nothing with them.
// MODE bits: 1 = XCD-aware tile order (conv_dma's), 2 = every workgroup starts at its own K offset and wraps (no two
// neighbours hammer the same 128-byte column of their rows at the same time), 4 = every workgroup reads tile (0, 0)
// (L2-hot: the DMA issue / L1 -> LDS rate alone)
__device__ __forceinline__ int xcd_remap_h(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
template <int BM, int BN, int NST, int RB, int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const BigArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int RPI = 1024 / RB;                       // rows per DMA instruction
  constexpr int LA = BM * RB / 4096, LB = BN * RB / 4096, SB = (BM + BN) * RB;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = (MODE & 1) ? xcd_remap_h(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
  if (MODE & 4) tm = tn = 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, 0x7fffffff, 0x00020000);
  unsigned va[LA], vb[LB];
#pragma unroll
  for (int q = 0; q < LA; ++q) va[q] = (unsigned)((tm * BM + (wave + 4 * q) * RPI + lane / (RB / 16)) * p.K2 + (lane % (RB / 16)) * 16);
#pragma unroll
  for (int q = 0; q < LB; ++q) vb[q] = (unsigned)((tn * BN + (wave + 4 * q) * RPI + lane / (RB / 16)) * p.K2 + (lane % (RB / 16)) * 16);
  const int nk = p.K2 / RB;
  unsigned kb = (MODE & 2) ? (unsigned)(((tm * 5 + tn * 3) % nk) * RB) : 0u;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 breg[LB];
  // MODE 8: the B (weight) rows go to REGISTERS by plain 16-byte loads, 1 KB contiguous per wave-instruction (what a
  // fragment-ordered weight pack would give), instead of through the LDS-DMA path: is the ~40 B/clk/CU ceiling the DMA's or the L1's?
  unsigned wbase = (unsigned)((tn * (gridDim.x / p.tiles_n > 0 ? 1 : 1)) * BN * p.K2) + (unsigned)(wave * LB) * 1024u + (unsigned)lane * 16u;
  auto issue = [&](int st) {
    unsigned char* sa = smem + st * SB + wave * 1024;
#pragma unroll
    for (int q = 0; q < LA; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sa + q * 4096), 16, (int)va[q], (int)kb, 0, 0);
    if constexpr (MODE & 8) {
#pragma unroll
      for (int q = 0; q < LB; ++q) breg[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(wbase + q * 1024), (int)(kb * (BN / (RB / 16) / 8)), 0);
    } else {
      unsigned char* sb = smem + st * SB + BM * RB + wave * 1024;
#pragma unroll
      for (int q = 0; q < LB; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(sb + q * 4096), 16, (int)vb[q], (int)kb, 0, 0);
    }
    kb += RB;
    if (kb == (unsigned)p.K2) kb = 0;
  };
  for (int s = 0; s < NST - 1; ++s) issue(s);
  int st = NST - 1;
  for (int ks = 0; ks < nk; ++ks) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (LA + LB)) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (MODE & 8) {
#pragma unroll
      for (int q = 0; q < LB; ++q) asm volatile("" ::"v"(breg[q]));     // (consumes the previous step's registers: the compiler waits for them here)
    }
    if (ks + NST - 1 < nk) issue(st);
    st = st + 1 == NST ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (MODE & 8) {
#pragma unroll
    for (int q = 0; q < LB; ++q) asm volatile("" ::"v"(breg[q]));
  }
  if (p.sink && blockIdx.x == 0x7fffffff) p.sink[threadIdx.x] = smem[threadIdx.x];
}

template <typename F> static double time_kernel(F&& fn, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  fn(); CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = std::min(best, (double)ms * 1e3 / iters);
  }
  return best;
}

// store burst: every workgroup writes ROWS x 256-byte row segments (16 lanes x 16 B per row, 16 rows per pass of 256 threads) of a
// [M][pitch] bf16 tensor -- the epilogue's store pattern -- and nothing else.  AUX = cache-policy bits of the buffer store
// (0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1 = write-through, 3 = sc0 nt ...)
template <int ROWS, int AUX>
__global__ __launch_bounds__(256) void store_kernel(unsigned char* out, int pitch_bytes, int tiles_n, unsigned total_bytes, unsigned* stamps) {
  const int tile = xcd_remap_h(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int er = threadIdx.x >> 4, ec = threadIdx.x & 15;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, total_bytes, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const unsigned t0 = (unsigned)__builtin_readcyclecounter();
#pragma unroll
  for (int ps = 0; ps < ROWS / 16; ++ps) {
    const unsigned off = (unsigned)((tm * ROWS + ps * 16 + er) * pitch_bytes + tn * 256 + ec * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, AUX);
  }
  const unsigned t1 = (unsigned)__builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned t2 = (unsigned)__builtin_readcyclecounter();
  if (threadIdx.x == 0 && stamps) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = t2 - t0; }
}
template <int ROWS, int AUX>
static void store_probe(int M, int N, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  const int tiles_m = M / ROWS, tiles_n = N / 128, grid = tiles_m * tiles_n;
  const size_t bytes = (size_t)M * N * 2;
  unsigned char* out; CK(hipMalloc((void**)&out, bytes));
  unsigned* st; CK(hipMalloc((void**)&st, (size_t)grid * 8));
  const double t = time_kernel([&]() { hipLaunchKernelGGL((store_kernel<ROWS, AUX>), dim3(grid), dim3(256), 0, s, out, N * 2, tiles_n, (unsigned)bytes, st); }, iters, s, a, b);
  std::vector<unsigned> h((size_t)grid * 2); CK(hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> iss, ack; for (int g = 0; g < grid; ++g) { iss.push_back(h[2 * g]); ack.push_back(h[2 * g + 1]); }
  const double wr = (double)grid * ROWS * 256;
  printf("  store burst M %6d N %4d tile %3dx128 aux %2d: %4d wgs, %7.2f us per launch = %5.2f TB/s written; per workgroup issue med %5.0f cyc, acknowledged med %5.0f p90 %5.0f\n",
         M, N, ROWS, AUX, grid, t, wr / t * 1e-6, pct(iss, 0.5), pct(ack, 0.5), pct(ack, 0.9));
  (void)hipFree(out); (void)hipFree(st);
}

template <int BM, int BN, int NST, int RB, int MODE>
static void floor_stream(const char* name, const Problem& p, int iters, hipStream_t s, hipEvent_t a, hipEvent_t b) {
  BigArgs g; memset(&g, 0, sizeof(g));
  g.a = p.x; g.w = p.wf; g.K2 = p.fwd.Cin * 2; g.nk = g.K2 / RB;
  if (g.K2 % RB != 0) return;
  const int tiles_m = p.M / BM;       // whole tiles only (no ragged last tile: this is a bandwidth probe)
  g.tiles_n = (p.fwd.Cout + BN - 1) / BN;
  const int grid = tiles_m * g.tiles_n;
  const size_t lds = (size_t)NST * (BM + BN) * RB;
  if (lds > 160 * 1024) return;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<BM, BN, NST, RB, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const double t = time_kernel([&]() { hipLaunchKernelGGL((stream_kernel<BM, BN, NST, RB, MODE>), dim3(grid), dim3(256), lds, s, g); }, iters, s, a, b);
  const double bytes = (double)grid * g.nk * (BM + BN) * RB;
  printf("  %-9s stream %3dx%3d ring %d rowbytes %d mode %d%s%s%s: %4d wgs, %5.0f KB LDS, %7.2f us, %6.2f TB/s into LDS = %4.1f B/clk/CU @2.1GHz\n", name, BM, BN,
         NST, RB, MODE, (MODE & 1) ? " xcd" : "", (MODE & 2) ? " krot" : "", (MODE & 4) ? " hot" : (MODE & 8) ? " Breg" : "", grid, lds / 1024.0, t, bytes / t * 1e-6,
         bytes / t * 1e-6 * 1e12 / 256.0 / 2.1e9);
}

int main(int argc, char** argv) {
  std::string only, cfgs_s = "-1", wcfgs_s = "-1,8,9,10,11,12,13", modes = "fwd,dgrad", trace;
  int iters = 20, B = 8; bool dual = false, do_floor = false, check = false, pair = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&]() { return std::string(i + 1 < argc ? argv[++i] : ""); };
    if (a == "--only") only = val();
    else if (a == "--cfgs") cfgs_s = val();
    else if (a == "--wcfgs") wcfgs_s = val();
    else if (a == "--modes") modes = val();
    else if (a == "--iters") iters = atoi(val().c_str());
    else if (a == "--batch") B = atoi(val().c_str());
    else if (a == "--dual") dual = true;
    else if (a == "--f32") g_f32 = true;
    else if (a == "--floor") do_floor = true;
    else if (a == "--check") check = true;
    else if (a == "--pair") pair = true;
    else if (a == "--splitk") g_splitk = atoi(val().c_str());
    else if (a == "--trace") trace = val();
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 64; }
  }
  std::vector<int> cfgs, wcfgs;
  for (size_t p = 0; p < cfgs_s.size();) { cfgs.push_back(atoi(cfgs_s.c_str() + p)); p = cfgs_s.find(',', p); if (p == std::string::npos) break; ++p; }
  for (size_t p = 0; p < wcfgs_s.size();) { wcfgs.push_back(atoi(wcfgs_s.c_str() + p)); p = wcfgs_s.find(',', p); if (p == std::string::npos) break; ++p; }
  auto wanted = [&](const char* n) {
    if (only.empty()) return true;
    for (size_t p = 0; p < only.size();) {
      size_t e = only.find(',', p); if (e == std::string::npos) e = only.size();
      if (strstr(n, only.substr(p, e - p).c_str())) return true;
      p = e + 1;
    }
    return false;
  };
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("# cbench on %s (%d CUs), batch %d, %d launches per timing, best of 3\n", prop.name, prop.multiProcessorCount, B, iters);

  if (!trace.empty()) {
    // S:CFG[:dual]
    const size_t c1 = trace.find(':');
    const std::string sn = trace.substr(0, c1);
    const size_t c2 = trace.find(':', c1 + 1);
    const int cfg = atoi(trace.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1).c_str());
    const bool tdual = c2 != std::string::npos;
    for (const Shape& sh : SHAPES) {
      if (sn != sh.name) continue;
      Problem p0 = make_problem(sh, B, 1), p1 = make_problem(sh, B, 2);
      for (int w = 0; w < 3; ++w) launch(p0, false, cfg, s0);
      CK(hipDeviceSynchronize());
      printf("trace %s cfg %d (forward, statistics on)\n", sh.name, cfg);
      TraceSet t = run_trace(p0, cfg, s0, nullptr);
      analyze(t, "alone", nullptr);
      TraceSet t2 = run_trace(p0, cfg, s0, nullptr);
      analyze(t2, "alone (repeat)", nullptr);
      if (tdual) {
        // both launches enqueued before either runs: hold the streams behind an event recorded after a long-ish kernel
        const int maxwg = ((p0.M + 63) / 64) * ((p0.fwd.Cout + 63) / 64) + 8;
        unsigned *d0, *d1; CK(hipMalloc((void**)&d0, (size_t)maxwg * 288)); CK(hipMalloc((void**)&d1, (size_t)maxwg * 288));
        CK(hipMemset(d0, 0, (size_t)maxwg * 288)); CK(hipMemset(d1, 0, (size_t)maxwg * 288));
        pxl_conv_desc q = p0.fwd; q.tile_cfg = cfg;
        for (int rep = 0; rep < 4; ++rep) { launch(p0, false, cfg, s0); launch(p1, false, cfg, s1); }     // both queues busy
        pxl_conv_dma_trace(&q, p0.x, p0.wf, p0.y, nullptr, p0.stats, d0, s0);
        pxl_conv_dma_trace(&q, p1.x, p1.wf, p1.y, nullptr, p1.stats, d1, s1);
        CK(hipDeviceSynchronize());
        TraceSet a, b; a.w.resize((size_t)maxwg * 72); b.w.resize((size_t)maxwg * 72);
        CK(hipMemcpy(a.w.data(), d0, a.w.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.w.data(), d1, b.w.size() * 4, hipMemcpyDeviceToHost));
        a.nwg = b.nwg = 0;
        for (int g = 0; g < maxwg; ++g) { if (a.w[(size_t)g * 72 + 64] >= 5) a.nwg = g + 1; if (b.w[(size_t)g * 72 + 64] >= 5) b.nwg = g + 1; }
        unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
        for (int g = 0; g < a.nwg; ++g) { const unsigned* r = &a.w[(size_t)g * 72]; a0 = std::min(a0, r[67] | ((unsigned long long)r[68] << 32)); a1 = std::max(a1, r[69] | ((unsigned long long)r[70] << 32)); }
        for (int g = 0; g < b.nwg; ++g) { const unsigned* r = &b.w[(size_t)g * 72]; b0 = std::min(b0, r[67] | ((unsigned long long)r[68] << 32)); b1 = std::max(b1, r[69] | ((unsigned long long)r[70] << 32)); }
        printf("  two streams: launch A runs %.2f .. %.2f us, launch B %.2f .. %.2f us (common clock, 0 = earlier start)\n", 0.01 * (double)(a0 - std::min(a0, b0)),
               0.01 * (double)(a1 - std::min(a0, b0)), 0.01 * (double)(b0 - std::min(a0, b0)), 0.01 * (double)(b1 - std::min(a0, b0)));
        analyze(a, "stream A (co-running)", &b);
        analyze(b, "stream B (co-running)", &a);
        (void)hipFree(d0); (void)hipFree(d1);
      }
      free_problem(p0); free_problem(p1);
    }
    return 0;
  }

  if (do_floor) {
    BigArgs g; memset(&g, 0, sizeof(g));
    for (int grid : {137, 274, 548, 1096}) {
      for (size_t lds : {(size_t)0, (size_t)48 * 1024, (size_t)72 * 1024, (size_t)144 * 1024}) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&null_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const double t = time_kernel([&]() { hipLaunchKernelGGL(null_kernel, dim3(grid), dim3(256), lds, s0, g); }, 50, s0, ea, eb);
        printf("  empty launch: %4d wgs x 256 threads, %5.0f KB LDS, %zu-byte arguments: %6.2f us back to back\n", grid, lds / 1024.0, sizeof(BigArgs), t);
      }
    }
    if (wanted("store")) {
      store_probe<64, 0>(8704, 256, iters, s0, ea, eb);  store_probe<64, 0>(8704, 1024, iters, s0, ea, eb);  store_probe<64, 0>(8704, 2048, iters, s0, ea, eb);
      store_probe<64, 0>(133120, 256, iters, s0, ea, eb);
      store_probe<128, 0>(8704, 256, iters, s0, ea, eb); store_probe<128, 0>(8704, 1024, iters, s0, ea, eb);
      store_probe<64, 2>(8704, 256, iters, s0, ea, eb);  store_probe<64, 2>(8704, 1024, iters, s0, ea, eb);  store_probe<64, 2>(133120, 256, iters, s0, ea, eb);
      store_probe<64, 17>(8704, 256, iters, s0, ea, eb); store_probe<64, 17>(8704, 1024, iters, s0, ea, eb);
      store_probe<64, 16>(8704, 256, iters, s0, ea, eb); store_probe<64, 16>(8704, 1024, iters, s0, ea, eb);
      store_probe<64, 1>(8704, 256, iters, s0, ea, eb);  store_probe<64, 1>(8704, 1024, iters, s0, ea, eb);
      store_probe<64, 3>(8704, 1024, iters, s0, ea, eb); store_probe<64, 19>(8704, 1024, iters, s0, ea, eb);
    }
    for (const Shape& sh : SHAPES) {
      if (sh.k != 1 || sh.s != 1 || !wanted(sh.name)) continue;
      Problem p = make_problem(sh, B, 1);
      floor_stream<64, 128, 3, 128, 9>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 2, 128, 9>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 2, 128, 9>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 2, 128, 1>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 2, 128, 1>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 128, 0>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 128, 1>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 128, 2>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 128, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 128, 4>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 256, 1>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 3, 256, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<64, 128, 2, 256, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 3, 128, 1>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 3, 128, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 3, 128, 4>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 128, 2, 256, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<128, 64, 3, 128, 3>(sh.name, p, iters, s0, ea, eb);
      floor_stream<256, 128, 3, 128, 3>(sh.name, p, iters, s0, ea, eb);
      free_problem(p);
    }
    return 0;
  }

  std::map<std::string, double> tot; double totf = 0.0;
  printf("%-9s %5s %5s %1s %1s %1s %4s %6s |", "shape", "Cin", "Cout", "k", "s", "d", "H", "M");
  printf(" mode:cfg us (TFLOP/s)%s\n", dual ? " [pair on two streams: us per pair]" : "");
  for (const Shape& sh : SHAPES) {
    if (!wanted(sh.name) || (only.empty() && strncmp(sh.name, "fd.", 3) == 0)) continue;
    Problem p0 = make_problem(sh, B, 1), p1;
    if (dual || pair) p1 = make_problem(sh, B, 2);
    printf("%-9s %5d %5d %1d %1d %1d %4d %6d |", sh.name, sh.cin, sh.cout, sh.k, sh.s, sh.d, sh.H, p0.M);
    for (const char* mode : {"fwd", "dgrad"}) {
      if (modes.find(mode) == std::string::npos) continue;
      const bool dg = mode[0] == 'd';
      double best = 1e30, best_pair = 1e30;
      for (int cfg : cfgs) {
        const double t = time_single(p0, dg, cfg, iters, s0, ea, eb);
        if (t < 0) { printf(" %s:%d ERR", mode, cfg); continue; }
        printf(" %s:%d %6.1f (%5.0f)", mode, cfg, t, p0.flops / t * 1e-6);
        if (check && !g_f32) check_cfg(p0, dg, cfg, s0);
        if (dual) { const double t2 = time_dual(p0, p1, dg, cfg, iters, s0, s1); printf(" [%6.1f]", t2); }
        if (pair && !dg) {
          const double t2 = time_pair(p0, p1, cfg, iters, s0, ea, eb);
          printf(" {pair %6.1f}", t2);
          if (t2 > 0) { best_pair = std::min(best_pair, t2); }
          if (check && t2 > 0) {        // both halves of the pair against their single launches
            std::vector<uint16_t> r0(p0.ny), r1(p1.ny), g0(p0.ny), g1(p1.ny);
            launch(p0, false, cfg, s0); launch(p1, false, cfg, s0); CK(hipStreamSynchronize(s0));
            CK(hipMemcpy(r0.data(), p0.y, p0.ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), p1.y, p1.ny * 2, hipMemcpyDeviceToHost));
            CK(hipMemset(p0.y, 0xff, p0.ny * 2)); CK(hipMemset(p1.y, 0xff, p1.ny * 2));
            launch_pair(p0, p1, cfg, s0); CK(hipStreamSynchronize(s0));
            CK(hipMemcpy(g0.data(), p0.y, p0.ny * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), p1.y, p1.ny * 2, hipMemcpyDeviceToHost));
            printf(" %s", (r0 == g0 && r1 == g1) ? "same" : "PAIR-DIFFERS!");
          }
        }
        best = std::min(best, t);
      }
      if (best < 1e29) tot[mode] += best * sh.cnt;
      if (best_pair < 1e29) tot["pair(2 nets)"] += best_pair * sh.cnt;
    }
    if (modes.find("wgrad") != std::string::npos) {
      double best = 1e30;
      for (int cfg : wcfgs) {
        if ((cfg == 8 || cfg == 9 || cfg == 13) && p0.fwd.Cin % 128 != 0) continue;
        const double t = time_wgrad(p0, cfg, iters, s0, ea, eb);
        if (t < 0) { printf(" wgrad:%d ERR", cfg); continue; }
        printf(" wgrad:%d %6.1f (%5.0f)", cfg, t, p0.flops / t * 1e-6);
        best = std::min(best, t);
      }
      if (best < 1e29) tot["wgrad"] += best * sh.cnt;
    }
    totf += p0.flops * sh.cnt;
    printf("\n"); fflush(stdout);
    free_problem(p0); if (dual || pair) free_problem(p1);
  }
  for (auto& kv : tot) printf("sum over net (%s, best cfg per shape x count): %.3f ms -> %.1f TFLOP/s\n", kv.first.c_str(), kv.second * 1e-3, totf / kv.second * 1e-6);
  return 0;
}
