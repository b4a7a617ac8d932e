// Host cost of a kernel launch on this box, the floor every host-paced step (CCT, GCT) is measured against:
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_floor tools/launch_floor.cpp && tools/launch_floor
// (a) back-to-back launches of an empty kernel on one stream (host time per hipLaunchKernelGGL, queue never full: sync every 256),
// (b) the same with a 256-byte by-value argument struct (the size of the convolution kernels' argument block),
// (c) launches dealt over two streams with an event record + stream wait between them (the fork / join the executor does),
// (d) hipMemsetAsync of 4 KB, (e) hipEventRecord alone.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[64]; };
__global__ void k_empty(int* p) { if (p != nullptr && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p != nullptr && threadIdx.x == 9999) *p = b.v[3]; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t ev[8]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  int* d; hipMalloc(&d, 1 << 20);
  Big b{}; const int N = 4096, CH = 256;
  auto run = [&](const char* name, auto&& body) {
    for (int i = 0; i < 64; ++i) body(i);
    hipDeviceSynchronize();
    double host = 0;
    for (int c = 0; c < N / CH; ++c) {
      const double t0 = now();
      for (int i = 0; i < CH; ++i) body(i);
      host += now() - t0;
      hipDeviceSynchronize();
    }
    printf("%-58s %7.2f us per call (host)\n", name, 1e6 * host / N);
  };
  run("(a) empty kernel, one stream", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s0, d); });
  run("(b) 256-byte argument block, one stream", [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s0, b, d); });
  run("(c) two streams, record + wait per launch", [&](int i) {
    hipStream_t a = (i & 1) ? s1 : s0, o = (i & 1) ? s0 : s1;
    hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, a, b, d); hipEventRecord(ev[i & 7], a); hipStreamWaitEvent(o, ev[i & 7], 0); });
  run("(d) hipMemsetAsync 4 KB", [&](int) { hipMemsetAsync(d, 0, 4096, s0); });
  run("(e) hipEventRecord", [&](int i) { hipEventRecord(ev[i & 7], s0); });
  run("(f) hipEventRecord + hipStreamWaitEvent (other stream)", [&](int i) { hipEventRecord(ev[i & 7], s0); hipStreamWaitEvent(s1, ev[i & 7], 0); });
  // the same launch with the queue kept full (no sync): what the host pays when it runs ahead of the GPU
  { hipDeviceSynchronize(); const double t0 = now();
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s0, d);
    const double t1 = now(); hipDeviceSynchronize(); const double t2 = now();
    printf("(g) 20000 empty launches without sync: host %.2f us per launch, GPU drained %.2f us per launch\n", 1e6 * (t1 - t0) / 20000, 1e6 * (t2 - t0) / 20000); }
  return 0;
}
