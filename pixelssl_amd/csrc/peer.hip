// One-shot all-reduce of SMALL vectors over peer-mapped buffers (Sync-BN statistics: 2*C floats, <= 16 KB).
//   reference: the master/slave queue exchange of sync_batchnorm/comm.py:59-137 (one round trip per BatchNorm layer and
//   pass); RCCL's ring / tree all-reduce costs 15-25 us per call at these sizes, and an MT step has ~310 of them on its
//   critical path.
// Every rank owns one exchange buffer in device memory that is mapped into every other rank's address space (HIP IPC,
// dmabuf); xGMI is point-to-point, so a rank STORES its vector straight into its slot of every peer's buffer and adds up
// the slots of its own buffer in rank order -- one kernel, one xGMI hop, no intermediate rank, bit-identical sums on all
// ranks.  Every element travels as ONE 8-byte word {value, epoch}: a reader spins on the word until it carries the
// current epoch, so there is no separate flag, no fence and no ordering requirement between stores (the scheme of RCCL's
// low-latency protocol).  Two slot sets alternate by epoch parity: a rank can only be one exchange ahead of the slowest
// rank (it needs that rank's words of the current exchange before it returns), so the set it overwrites next has been
// read by everyone.  A spin that exceeds the time-out (a peer died) raises the context's status word instead of hanging
// the GPU; the word is sticky (see the kernel).
#include <cstring>
#include <new>

#include "common.h"

namespace {

constexpr int MAXW = 16;

struct PeerArgs {
  unsigned long long* slots[MAXW];       // every rank's buffer: [2][world][slot] words {epoch << 32 | float bits}
  unsigned* status;                      // local: != 0 after a time-out
  unsigned* host_status;                 // the same word in mapped host memory: the host polls it without a device sync
  int rank, world, slot;
  long long timeout_ticks;               // of the 100 MHz wall clock
};

// The logical vector is [buf0[0..n_each) | buf1[0..n_each)] (buf1 == nullptr: one vector); this launch exchanges its elements
// base .. base + m.  nrep > 1: the operands are statistics replicas [nrep][n_each] that are folded on the way in (what
// pxl_bn_fold_replicas did in a launch of its own); the all-reduced sum lands in replica 0.
// add_lo / add_hi (optional, single-vector exchanges): the LOCAL value of element g is also accumulated into add_lo[g] for
// g < n_each / 2 and into add_hi[g - n_each / 2] above -- the BatchNorm backward's d(beta) += sum(dz), d(gamma) += sum(dz * xhat),
// which the multi-rank pass takes from the local sums before they are all-reduced (pxl_bn_param_grad's launch).
__global__ __launch_bounds__(256) void peer_allreduce_kernel(const PeerArgs a, float* __restrict__ buf0, float* __restrict__ buf1,
                                                             int n_each, int nrep, int base, int m, unsigned epoch,
                                                             float* __restrict__ add_lo, float* __restrict__ add_hi) {
  const int par = epoch & 1u;
  const size_t set = (size_t)par * a.world * a.slot;
  // sticky abort: once an exchange of this context has given up on a peer, every later one posts its words (the peers may
  // still be alive and waiting for them) but does not wait -- a dead peer costs ONE time-out, not one per exchange (an MT
  // step has ~310 of them).  The sums are invalid from then on and SAY so: every element whose peer word did not arrive is
  // written back as NaN (never the stale word of an older exchange), so the losses and the update of that step are NaN; the host
  // reads the status word from mapped host memory at every optimizer step (pxl_peer_status_nosync, dist.poll_peers) and aborts
  // the run -- or, opt-in, moves the statistics to RCCL / torch.distributed.
  const bool aborted = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    const int g = base + i;
    float* const src = g < n_each ? buf0 + g : buf1 + (g - n_each);
    float v = src[0];
    for (int r = 1; r < nrep; ++r) v += src[(size_t)r * n_each];
    if (g < n_each / 2) { if (add_lo != nullptr) add_lo[g] += v; }
    else if (g < n_each) { if (add_hi != nullptr) add_hi[g - n_each / 2] += v; }
    // my element into my slot of every rank's buffer (my own included: the sum below reads every slot the same way)
    const unsigned long long word = ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v);
    for (int r = 0; r < a.world; ++r)
      __hip_atomic_store(a.slots[r] + set + (size_t)a.rank * a.slot + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // every rank's element from MY buffer, in rank order (identical on every rank)
    const unsigned long long* mine = a.slots[a.rank] + set + i;
    float acc = 0.f;
    long long t0 = 0;
    bool valid = true;
    for (int q = 0; q < a.world; ++q) {
      unsigned long long w = __hip_atomic_load(mine + (size_t)q * a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      while ((unsigned)(w >> 32) != epoch) {
        if (aborted) { valid = false; break; }
        if (t0 == 0) t0 = wall_clock64();
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > a.timeout_ticks) {
          atomicExch(a.status, 1u + (unsigned)q);
          if (a.host_status != nullptr)
            __hip_atomic_store(a.host_status, 1u + (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          valid = false;
          break;
        }
        w = __hip_atomic_load(mine + (size_t)q * a.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (!valid) break;              // (the word that is there belongs to an older exchange: never added)
      acc += __uint_as_float((unsigned)w);
    }
    src[0] = valid ? acc : __uint_as_float(0x7fc00000u);
  }
}

}  // namespace

struct pxl_peer {
  int rank = 0, world = 1, slot = 0;
  size_t bytes = 0, status_off = 0;
  char* local = nullptr;
  char* mapped[MAXW] = {};
  bool opened = false;
  unsigned* host_status = nullptr;       // hipHostMalloc'ed mirror of the status word (NULL: not available)
  unsigned epoch = 0;
  long long timeout_ticks = 0;
  unsigned long exchanges = 0;           // exchanges issued on this context (the first ones wait longer: start-up skew)
};

extern "C" int pxl_peer_create(int rank, int world, int slot_floats, int timeout_ms, pxl_peer** out) {
  PXL_REQUIRE(out && world >= 1 && world <= MAXW && rank >= 0 && rank < world && slot_floats > 0 && timeout_ms > 0,
              "peer_create: bad argument (1 <= world <= %d)", MAXW);
  pxl_peer* p = new (std::nothrow) pxl_peer();
  PXL_REQUIRE(p != nullptr, "peer_create: out of host memory");
  p->rank = rank; p->world = world; p->slot = (slot_floats + 63) / 64 * 64;
  p->status_off = (size_t)2 * world * p->slot * sizeof(unsigned long long);
  p->bytes = p->status_off + 256;
  p->timeout_ticks = (long long)timeout_ms * 100000LL;          // 100 MHz
  void* mem = nullptr;
  // uncached / fine-grained device memory: stores of a peer become visible while a kernel of this rank is running
  hipError_t e = hipExtMallocWithFlags(&mem, p->bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&mem, p->bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    delete p;
    return pxl_set_error(PXL_ERR_UNSUPPORTED, "peer_create: no fine-grained device memory (%s)", hipGetErrorString(e));
  }
  p->local = static_cast<char*>(mem);
  e = hipMemset(p->local, 0, p->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(p->local);
    delete p;
    return pxl_set_error(PXL_ERR_HIP, "peer_create: hipMemset failed: %s", hipGetErrorString(e));
  }
  p->mapped[rank] = p->local;
  // host-visible mirror of the status word: written once, by the exchange that times out; read by the host every step
  void* hs = nullptr;
  if (hipHostMalloc(&hs, 64, hipHostMallocMapped) == hipSuccess) {
    p->host_status = static_cast<unsigned*>(hs);
    *p->host_status = 0u;
  } else {
    (void)hipGetLastError();
  }
  *out = p;
  return PXL_OK;
}

extern "C" int pxl_peer_handle(pxl_peer* p, void* handle64) {
  PXL_REQUIRE(p && handle64, "peer_handle: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  PXL_CHECK_HIP(hipIpcGetMemHandle(&h, p->local));
  std::memcpy(handle64, &h, 64);
  return PXL_OK;
}

// handles: [world][64] bytes, rank order (every rank's own entry is ignored)
extern "C" int pxl_peer_open(pxl_peer* p, const void* handles) {
  PXL_REQUIRE(p && handles && !p->opened, "peer_open: bad argument");
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * 64, 64);
    void* q = nullptr;
    PXL_CHECK_HIP(hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess));
    p->mapped[r] = static_cast<char*>(q);
  }
  p->opened = true;
  return PXL_OK;
}

extern "C" void pxl_peer_destroy(pxl_peer* p) {
  if (!p) return;
  for (int r = 0; r < p->world; ++r)
    if (r != p->rank && p->mapped[r]) (void)hipIpcCloseMemHandle(p->mapped[r]);
  if (p->local) (void)hipFree(p->local);
  if (p->host_status) (void)hipHostFree(p->host_status);
  delete p;
}

namespace {
int peer_exchange(pxl_peer* p, float* buf0, float* buf1, long n_each, int nrep, void* stream, float* add_lo = nullptr,
                  float* add_hi = nullptr) {
  PXL_REQUIRE(p->opened || p->world == 1, "peer_allreduce: peer buffers not opened (pxl_peer_open)");
  PeerArgs a;
  for (int r = 0; r < MAXW; ++r) a.slots[r] = r < p->world ? reinterpret_cast<unsigned long long*>(p->mapped[r]) : nullptr;
  a.status = reinterpret_cast<unsigned*>(p->local + p->status_off);
  a.host_status = p->host_status;
  a.rank = p->rank; a.world = p->world; a.slot = p->slot; a.timeout_ticks = p->timeout_ticks;
  // Start-up skew: the ranks reach their first exchanges seconds apart (per-rank autotune of several networks, first-touch page-ins, a
  // rank-0 validation pass outside the epoch barrier) -- not a dead peer.  The first PXL_PEER_WARM_EXCHANGES (1024: about three
  // Mean-Teacher steps) exchanges of a context wait PXL_PEER_WARM_SCALE (15) times as long before they give up; no hidden barrier.
  static const long warm_n = getenv("PXL_PEER_WARM_EXCHANGES") ? atol(getenv("PXL_PEER_WARM_EXCHANGES")) : 1024;
  static const long warm_scale = getenv("PXL_PEER_WARM_SCALE") ? atol(getenv("PXL_PEER_WARM_SCALE")) : 15;
  if ((long)p->exchanges < warm_n && warm_scale > 1) a.timeout_ticks = p->timeout_ticks * warm_scale;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long total = buf1 != nullptr ? 2 * n_each : n_each;
  for (long off = 0; off < total; off += p->slot) {
    const int m = (int)((total - off) < p->slot ? (total - off) : p->slot);
    p->epoch += 1;
    p->exchanges += 1;
    if (p->epoch == 0) p->epoch = 2;            // 0 is what the zero-filled buffer carries; keep the parity sequence
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(m > 1024 ? 4 : 1), dim3(256), 0, s, a, buf0, buf1, (int)n_each, nrep, (int)off, m,
                       p->epoch, add_lo, add_hi);
  }
  PXL_LAUNCH_CHECK();
  return PXL_OK;
}
}  // namespace

// in-place all-reduce(sum) of n floats at device pointer buf, enqueued on `stream`; vectors longer than the slot go in
// several exchanges.  Every rank must issue the same sequence of calls on a context.
extern "C" int pxl_peer_allreduce_sum(pxl_peer* p, float* buf, long n, void* stream) {
  PXL_REQUIRE(p && buf && n > 0 && n < (1L << 30), "peer_allreduce_sum: bad argument");
  return peer_exchange(p, buf, nullptr, n, 1, stream);
}

// Sync-BN statistics in ONE launch: buf0 (and buf1, optional: the same BatchNorm of a second network -- the MT student ||
// teacher pass of pxl_net_forward_pair) hold nrep replicas [nrep][n] of the local sums; the replicas are folded, the sums of
// both vectors exchanged together, and the all-reduced vector lands in replica 0 of each.  Replaces pxl_bn_fold_replicas +
// pxl_peer_allreduce_sum per network (4 launches and 2 exchanges per BatchNorm of a paired pass -> 1 and 1).
extern "C" int pxl_peer_allreduce_fold(pxl_peer* p, float* buf0, float* buf1, long n, int nrep, void* stream) {
  PXL_REQUIRE(p && buf0 && n > 0 && n < (1L << 29) && nrep >= 1, "peer_allreduce_fold: bad argument");
  return peer_exchange(p, buf0, buf1, n, nrep, stream);
}

// BatchNorm backward, multi-rank: sums = [sum(dz) | sum(dz * xhat)] ([2C], local).  dbeta += sums[0..C), dgamma += sums[C..2C)
// from the LOCAL values (the gradient all-reduce averages the parameter gradients over the ranks), then sums <- all-reduce(sums)
// for the batch-mean terms of the data gradient -- pxl_bn_param_grad + pxl_peer_allreduce_sum in one launch.
extern "C" int pxl_peer_allreduce_bnbwd(pxl_peer* p, float* sums, int C, float* dgamma, float* dbeta, void* stream) {
  PXL_REQUIRE(p && sums && C > 0, "peer_allreduce_bnbwd: bad argument");
  return peer_exchange(p, sums, nullptr, 2L * C, 1, stream, dbeta, dgamma);
}

extern "C" int pxl_peer_allreduce_hook(void* user, float* buf, int n, void* stream) {
  return pxl_peer_allreduce_sum(reinterpret_cast<pxl_peer*>(user), buf, (long)n, stream) == PXL_OK ? 0 : 1;
}

// Exchanges issued on this context so far (every rank issues the same sequence: the counter doubles as a consistency check).
extern "C" long pxl_peer_exchanges(const pxl_peer* p) { return p ? (long)p->epoch : 0; }

// The status word as the host sees it WITHOUT synchronising the device (mapped host memory the timing-out exchange writes):
// 0 = no exchange that has COMPLETED so far timed out; cheap enough for every optimizer step.  -1: no host mirror (use
// pxl_peer_status).
extern "C" int pxl_peer_status_nosync(const pxl_peer* p) {
  if (!p || !p->host_status) return -1;
  return (int)__atomic_load_n(p->host_status, __ATOMIC_RELAXED);
}

// 0 = every exchange so far met its peers; k > 0 = an exchange gave up waiting for rank k-1 (synchronises the device)
extern "C" int pxl_peer_status(pxl_peer* p, int* status) {
  PXL_REQUIRE(p && status, "peer_status: null argument");
  unsigned v = 0;
  PXL_CHECK_HIP(hipMemcpy(&v, p->local + p->status_off, sizeof(v), hipMemcpyDeviceToHost));
  *status = (int)v;
  return PXL_OK;
}
