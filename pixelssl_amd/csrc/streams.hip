// Hardware-queue-aware stream placement.
// HIP deals a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues; two streams that land on the same queue
// execute their kernels one after the other, whatever the program's dependency graph says.  Which queue a stream gets
// depends on how many streams the process (PyTorch's pools, profilers, ...) created before it, so the overlap this engine
// is built around -- teacher next to student, weight gradients next to data gradients, the discriminator update next to
// the task backward -- used to depend on creation order: measured on MI355X, one extra stream created at start-up moves
// the MT step from 13.2 to 15.8 ms, AdvSSL between 17.5 and 19.5 ms, CCT between 27.0 and 34.7 ms.
// Here the engine's concurrent ROLES get streams that are PROVEN to sit on different queues: candidates are created,
// every pair that matters is probed (a spinning kernel on one, a tiny kernel on the other: did the tiny one finish
// first?), and one representative per queue is kept.
#include <mutex>
#include <vector>

#include "common.h"

namespace {

__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void tiny_kernel(int* p) {
  if (p != nullptr && threadIdx.x == 1024) *p = 0;
}

struct Pool {
  std::mutex mu;
  bool ready = false, enabled = true;
  hipStream_t main = nullptr;
  std::vector<hipStream_t> reps;            // one stream per hardware queue other than main's, in discovery order
  int probes = 0;
};
Pool& pool() { static Pool p; return p; }

// 1: kernels of a and b overlap (different hardware queues), 0: they serialize, < 0: HIP error
int overlaps(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb) {
  int votes = 0;
  for (int rep = 0; rep < 2; ++rep) {
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 40000LL);         // 400 us of the 100 MHz wall clock
    if (hipEventRecord(ea, a) != hipSuccess) return -1;
    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, b, (int*)nullptr);
    if (hipEventRecord(eb, b) != hipSuccess) return -1;
    if (hipEventSynchronize(eb) != hipSuccess) return -1;
    const hipError_t q = hipEventQuery(ea);
    if (q == hipErrorNotReady) ++votes;                                        // b finished while a was still spinning
    else if (q != hipSuccess) return -1;
    (void)hipGetLastError();
    if (hipEventSynchronize(ea) != hipSuccess) return -1;
  }
  ++pool().probes;
  return votes == 2 ? 1 : 0;
}

}  // namespace

// Build the pool around `main_stream` (the stream the training step is enqueued on).  Idempotent; PXL_STREAM_POOL=0 turns
// placement off (every component then creates its own streams, as before).  *nqueues = queues found besides main's.
extern "C" int pxl_stream_pool_init(void* main_stream, int* nqueues) {
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.mu);
  if (!p.ready) {
    const char* e = getenv("PXL_STREAM_POOL");
    p.enabled = !(e != nullptr && e[0] == '0');
    p.main = reinterpret_cast<hipStream_t>(main_stream);
    if (p.enabled) {
      hipEvent_t ea, eb;
      PXL_CHECK_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
      PXL_CHECK_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
      const int want = 3, ncand = 16;
      for (int c = 0; c < ncand && (int)p.reps.size() < want; ++c) {
        hipStream_t s = nullptr;
        PXL_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        int fresh = overlaps(p.main, s, ea, eb);
        for (size_t r = 0; fresh == 1 && r < p.reps.size(); ++r) fresh = overlaps(p.reps[r], s, ea, eb);
        if (fresh < 0) return pxl_set_error(PXL_ERR_HIP, "stream_pool_init: a queue probe failed: %s", hipGetErrorString(hipGetLastError()));
        if (fresh == 1) p.reps.push_back(s);            // a queue nobody in the pool sits on yet
        // (streams that share a queue are left allocated: destroying them would hand their queue slot to the next candidate)
      }
      (void)hipEventDestroy(ea);
      (void)hipEventDestroy(eb);
    }
    p.ready = true;
  }
  if (nqueues) *nqueues = (int)p.reps.size();
  return PXL_OK;
}

// Stream of a role (include/pixelhip.h: PXL_STREAM_*), or NULL when placement is off / the pool has not been built / the
// device offered no second queue.  Roles beyond the queues found wrap around (they then share a queue with another ROLE,
// never with the main stream).
extern "C" void* pxl_stream_role(int role) {
  Pool& p = pool();
  std::lock_guard<std::mutex> lock(p.mu);
  if (!p.ready || !p.enabled || p.reps.empty() || role < 0) return nullptr;
  return p.reps[(size_t)role % p.reps.size()];
}

extern "C" int pxl_stream_pool_probes(void) { return pool().probes; }
