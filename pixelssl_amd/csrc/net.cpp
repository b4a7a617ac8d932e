// Network executor: runs the whole forward / backward launch sequence of one segmentation network
// (DeepLab-v2: ResNet-101 trunk + ASPP head + upsample/softmax) from C++, so the Python host makes
// one call per network pass instead of ~1000 per-layer calls.
//
// Fusion plan (what the layer program expresses):
//   conv  --epilogue-->  raw output y + per-channel [sum, sumsq]          (HBM: y written once)
//   [SyncBN hook: all-reduce of the 2C statistics]
//   bn_finalize        -> (scale, shift) of that BN, running-stat update
//   the CONSUMER (next conv / max-pool / residual join) applies relu(y*scale+shift) while loading,
//   so normalised activations are never materialised except the block outputs (consumed twice).
// Backward mirrors it: relu-mask / BN reduce / BN apply produce the gradient of each raw conv
// output in place, then wgrad (prologue = the same fused activation of the conv's input) and
// dgrad (implicit GEMM with transposed weights, accumulate via the epilogue addend).
//
// Memory: the caller owns everything.  `arena` holds all forward tensors + BN statistics of one
// forward pass (kept until its backward); `scratch` holds gradient buffers; `packed` the weights
// in kernel layout (fwd + transposed).  288 GB of HBM3E means no recomputation and no buffer
// aliasing games: every tensor gets its own slot.
#include <algorithm>
#include <cmath>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <new>

#include "common.h"

extern "C" {
int pxl_conv_dma_finalize(const pxl_conv_desc* d, const void* in, const void* w, void* out, const float* bias, float* stats,
                          const pxl_bn_fin* fin, unsigned* counter, void* stream);
int pxl_conv_dma_bnin(const pxl_conv_desc* d, const void* y, const void* w, void* out, const float* bias, float* stats,
                      const pxl_bn_fin* bin, int bin_relu, void* z, void* stream);
int pxl_conv_dgrad_bnreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                            const void* bn_y, const float* bn_coef, int bn_relu, float* bn_sums, void* stream);
int pxl_conv_dgrad_joinreduce(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                              const void* join_out, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream);
int pxl_colsum(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream);
int pxl_colsum_ordered(int dtype, int M, int Cp, int Creal, const void* x, float* out, void* stream);
int pxl_residual_bwd_reduce_rep(int dtype, int M, int C, const void* dout, const void* out, const void* y, const float* coef,
                                void* g, void* g2, float* sums, int nrep, void* stream);
int pxl_head_loss_ex(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                     const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                     float mse_weight, const float* mse_weight_dev, int kernel_choice, int ordered, void* dlow, void* workspace,
                     size_t ws_bytes, float* sums, void* stream);
int pxl_conv_dgrad_joinreduce_bits(const pxl_conv_desc* d, const void* dy, const void* wt, void* din, const void* addend,
                                   const void* join_bits, const void* bn_y, const float* bn_coef, float* bn_sums, void* stream);
int pxl_residual_fwd_bits(int dtype, long M, int C, const void* y, const float* ycoef, const void* res, const float* rcoef, void* out,
                          void* bits, void* stream);
int pxl_residual_finalize_fwd_bits(int dtype, long M, int C, const void* y, const pxl_bn_fin* yfin, const void* res,
                                   const pxl_bn_fin* rfin, void* out, void* bits, void* stream);
int pxl_conv_dma_slabs(const pxl_conv_desc* d, const void* in, const void* w, float* ws, size_t ws_bytes, int slices, void* stream);
int pxl_aspp_col2im(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                    const int16_t* dx, const float* P, int nslab, size_t slab_floats, const float* bias, void* out, int Cp, void* stream);
int pxl_aspp_dp_gather(int dtype, int B, int H, int W, int J, int GP, int ngroups, int cout, int tpg, const int16_t* dy,
                       const int16_t* dx, const void* dout, int Cp, void* dP, void* stream);
int pxl_aspp_dw_scatter(const float* tmp, int ngroups, int GP, int cout, int tpg, int Cin, int Cpin, float* grads, const long* w_off, void* stream);
int pxl_aspp_pack(int dtype, const float* params, const long* w_off, int ngroups, int GP, int cout, int tpg, int Cin, int Cp, void* Wp, void* Wd, void* stream);
int pxl_stem_patches(int dtype, const float* x, void* P, int B, int C, int H, int W, int kh, int kw, int stride, int pad,
                     int Ho, int Wo, int Kp, void* stream);
int pxl_nchw_parts_to_nhwc(int dtype, int nparts, const float* const* srcs, const int* chans, void* y, int B, int H, int W,
                           int Cp, void* stream);
int pxl_nhwc_to_nchw_parts(int dtype, const void* x, int nparts, float* const* dsts, const int* chans, int B, int H, int W,
                           int Cp, void* stream);
int pxl_residual_bwd_reduce(int dtype, int M, int C, const void* dout, const void* out, const void* y, const float* coef,
                            void* g, void* g2, float* sums, void* stream);
int pxl_conv_dma_eligible(const pxl_conv_desc* d, const float* in_scale, const void* workspace);
size_t pxl_head_loss_lds_bytes(int w, int C, int W);
int pxl_head_loss(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                  const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                  float mse_weight, void* dlow, void* workspace, size_t ws_bytes, float* sums, void* stream);
int pxl_head_loss_hp(int dtype, int B, int h, int w, int Cp, int C, int H, int W, int align_corners, const void* s_low,
                     const void* t_low, const float* gt, int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight,
                     const float* mse_weight_dev, void* dlow, void* workspace, size_t ws_bytes, float* sums, void* stream);
int pxl_conv_wgrad_dma_eligible(const pxl_conv_desc* d, const float* in_scale);
struct pxl_peer;
int pxl_peer_allreduce_hook(void* user, float* buf, int n, void* stream);
int pxl_peer_allreduce_fold(pxl_peer* p, float* buf0, float* buf1, long n, int nrep, void* stream);
int pxl_peer_allreduce_bnbwd(pxl_peer* p, float* sums, int C, float* dgamma, float* dbeta, void* stream);
}

#include <chrono>
namespace pxlht {
bool on = getenv("PXL_HOST_TRACE") != nullptr && getenv("PXL_HOST_TRACE")[0] == '1';
static long g_ns[64], g_cnt[64];
long now() { return (long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void add(int slot, long ns) { g_ns[slot] += ns; g_cnt[slot] += 1; }
static struct Dump {
  ~Dump() {
    if (!on) return;
    static const char* names[64] = {"fwd INPUT", "fwd CONV", "fwd MAXPOOL", "fwd RESIDUAL", "fwd HEAD", "fwd ACT", "fwd IBN", "fwd AVGPOOL", "fwd CONCAT",
                                    "fwd UPCAT", "fwd PIXSHUF", nullptr, nullptr, nullptr, nullptr, nullptr,
                                    "launch_one: hipLaunchKernelGGL + check", "launch_dma (whole)", "pxl_net_forward (whole)", "pxl_net_backward (whole)",
                                    "bwd: issue_wgrads", "bwd: flush / joins"};
    for (int k = 0; k < 64; ++k)
      if (g_cnt[k]) fprintf(stderr, "PXL_HOST_TRACE %-44s %8ld calls %10.3f ms %8.2f us/call\n", names[k] ? names[k] : "?", g_cnt[k], g_ns[k] / 1e6, g_ns[k] / 1e3 / g_cnt[k]);
  }
} g_dump;
}  // namespace pxlht

namespace {

constexpr size_t ALIGN = 256;
// replicas of every BN statistics vector (atomic-contention spreading); PXL_STATS_REP overrides (tuning experiments)
// (4 since round 2: with the finalize folded into the kernel that applies the BN, every block of that kernel reduces the
// replicas itself -- 32 made that slower than a separate finalize launch, 4 makes it faster: MT 15.17 -> 14.98 ms / step)
static const int STATS_REP = [] { const char* e = getenv("PXL_STATS_REP"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }
inline int pitch_of(int c) { return c <= 8 ? 8 : (c + 31) / 32 * 32; }

struct TensorInfo {
  int H = 0, W = 0, C = 0, Cp = 0;
  size_t off = 0;        // arena offset (bytes)
  size_t goff = 0;       // scratch offset of the gradient buffer
  size_t bytes = 0;
  bool planned = false;
  bool needs_grad = false;
};

struct BnInfo {
  pxl_bn_desc d;
  int relu = 0;            // consumer applies ReLU right after this BN
  int M = 0;               // elements per channel on this device
  size_t stats_off = 0;    // arena: [2C] sums
  size_t coef_off = 0;     // arena: [4C]
  size_t bsum_off = 0;     // scratch: [2C]
  size_t bcoef_off = 0;    // scratch: [2C]
  int y_tensor = -1;
  // bf16 engine: relu(bn(y)) materialised once (z) when a convolution consumes it, so that the LDS-DMA
  // kernels read plain operands (the fp32 parity engine keeps the fused prologue)
  bool has_z = false;
  size_t z_off = 0;
  // the only consumer of relu(bn(y)) is a convolution whose data gradient runs on the LDS-DMA kernel: that launch
  // also produces this BN's backward sums (no separate reduce pass); -1 = no
  int fused_reduce_op = -1;
  int fin_nrep = 1;        // replicas of the forward sums at finalize time (1 after a Sync-BN fold + all-reduce)
  int nrep = 0;            // replicas of the forward statistics vector (STATS_REP; PXL_DETERMINISTIC: one per 64 pixel rows)
  int bnrep = 1;           // replicas of the BACKWARD sums (scratch, bsum_off): 1; PXL_DETERMINISTIC: max(one per 64 rows, 256)
  // forward: the finalize of this BN is folded into the kernel that applies it (z materialisation or the residual
  // join that is its only consumer) instead of a pxl_bn_finalize launch
  bool fin_in_consumer = false;
  size_t cnt_off = 0;      // arena (inside the statistics region, zeroed with it): the last-block-done ticket counter
  bool fin_by_conv = false;   // this pass: the producing convolution's last workgroup finalized the BN
  // BN-apply on load: the single convolution that consumes relu(bn(y)) reads the RAW y and applies the BatchNorm to its
  // tiles in LDS (conv_dma.hip: pxl_conv_dma_bnin), finalizing it in its prologue -- no pxl_bn_apply_fwd launch.  z is
  // then only needed by that convolution's weight gradient: networks that train have the convolution write it on the
  // way (its workgroups of output-channel tile 0), no-grad networks (the MT teacher) never write it.
  bool onload = false;
};

struct OpInfo {
  pxl_op d;
  int ntaps = 0;
  pxl_conv_desc fwd;       // forward geometry (all groups)
  int pair_cfg = -2;       // tile configuration of the PAIRED forward launch (pxl_net_tune_pair); -2 = not tuned: use fwd.tile_cfg
  pxl_conv_desc bwd;       // data-gradient geometry
  pxl_conv_desc grp[4];    // per-group forward geometry (wgrad)
  size_t wf_off = 0, wt_off = 0, bias_off = 0;   // packed buffer offsets
  size_t idx_off = 0;      // arena: maxpool argmax
  // IBNorm op: arena [B][2][C] sums + [2*nb] folded BN part + [B][4][C] coefficients; scratch: the backward twins
  size_t ibn_sums = 0, ibn_bn = 0, ibn_coef = 0, ibn_bsums = 0, ibn_bbn = 0;
  size_t ws_off = 0, ws_bytes = 0;               // arena: split-K fp32 workspace (small-N, long-K convs)
  // multi-rate head as ONE GEMM (csrc/aspp.hip): P = X . Wp^T with a column per (group, class, tap) -- J = ngroups * pg_GP
  // columns -- in the op's workspace, then col2im; backward: dP gathered once (scratch), dX and dWp are plain GEMMs again
  bool pg = false;
  int pg_J = 0, pg_GP = 0;
  pxl_conv_desc pg_fwd, pg_grp, pg_bwd;          // 1x1: Cin -> J (forward; weight gradient), J -> Cin (data gradient)
  size_t pg_wf_off = 0, pg_wt_off = 0;           // packed: Wp [J][Cin] (= the master weights, cast) and its transpose [Cin][J]
  size_t pg_dp_off = 0, pg_dw_off = 0;           // scratch: dP [M][J], the GEMM's weight gradient [J][cin] fp32
  // CONV: this op's data gradient is the last contribution to the gradient of residual join `join_op`'s output and
  // performs that join's backward in its epilogue; RESIDUAL: the convolution that does it (-1 = separate launch)
  int join_op = -1, join_conv = -1;
  // RESIDUAL whose backward is fused into a bf16 LDS-DMA data gradient: the ReLU mask that launch needs, as one byte per 8 channels
  // ([M][C / 8], arena) written by the join's forward kernel -- the backward reads 1/16 of the join output's bytes for it
  size_t bits_off = 0; bool bits = false;
  // stem in patch mode (bf16 engine): the convolution reads the network input, has few input channels and needs no data
  // gradient -> its im2col patches [M][patch_Kp] are written once per forward (arena) and the convolution runs as a 1x1
  // convolution over patch_Kp channels on the LDS-DMA kernels, forward and weight gradient
  bool patch = false;
  size_t patch_off = 0;
  int patch_K = 0, patch_Kp = 0;
};

}  // namespace

struct pxl_net {
  int dtype = PXL_F32;
  int esize = 4;
  int classes = 21;
  std::vector<OpInfo> ops;
  std::vector<BnInfo> bns;
  std::vector<TensorInfo> tensors;
  int B = 0, H = 0, W = 0;
  int Ho = 0, Wo = 0;              // HEAD output size (== H, W unless planned with pxl_net_plan_out)
  bool planned = false;
  size_t packed_bytes = 0, arena_bytes = 0, scratch_bytes = 0;
  size_t stats_region_off = 0, stats_region_bytes = 0;      // arena: all BN forward sums
  size_t bsum_region_off = 0, bsum_region_bytes = 0;        // scratch: all BN backward sums
  size_t ibn_region_off = 0, ibn_region_bytes = 0;          // arena: per-sample sums of every IBNorm layer (forward)
  size_t ibn_bregion_off = 0, ibn_bregion_bytes = 0;        // scratch: the same for the backward pass
  size_t up_ws_off = 0, up_ws_bytes = 0;                    // scratch: upsample backward workspace
  pxl_allreduce_fn sync = nullptr;
  void* sync_user = nullptr;
  int world = 1;                  // ranks sharing the BatchNorm statistics (pxl_net_set_sync); 1 = local statistics
  int grad_world = 1;             // ranks averaging the gradients (pxl_net_set_grad_sync)
  int head_op = -1;
  // optional per-launch timing of the contraction kernels (bench.py roofline leg)
  bool profile = false;
  struct Stamp { hipEvent_t a, b; int kind; double flops; };
  std::vector<Stamp> stamps;
  std::vector<hipEvent_t> pool;
  double prof_bytes[2] = {0.0, 0.0};     // algorithmic operand bytes of the stamped launches, per kind
  // backward runs the weight gradients on a second stream, concurrently with the data gradients (both only read
  // dy): the contraction kernels of this network are ~1 workgroup per CU and latency-bound, two in flight fill
  // each other's bubbles.  Created lazily; PXL_SIDE_STREAM=0 disables it.
  bool stem_patches = getenv("PXL_STEM_PATCHES") == nullptr || getenv("PXL_STEM_PATCHES")[0] != '0';
  // stem patches shared by two passes over ONE input tensor (Mean Teacher without input noise): `patches_made` -- this
  // network's patches for the next forward are already in its arena (pxl_net_make_patches); `patch_src` -- this pass (and its
  // backward) reads another network's patches instead of writing its own (pxl_net_borrow_patches arms `patch_src_next`)
  bool patches_made = false;
  const void* patch_src = nullptr;
  const void* patch_src_next = nullptr;
  bool input_needed = true;              // some op reads the NHWC copy of the input (false: only patch-mode convolutions)
  int in_parts = 0;                      // > 0: the next forward gathers its input from these tensors
  const float* in_src[4] = {nullptr, nullptr, nullptr, nullptr};
  int in_chans[4] = {0, 0, 0, 0};
  int fork_every = getenv("PXL_FORK_EVERY") ? atoi(getenv("PXL_FORK_EVERY")) : 1;   // convolutions per fork event (>= 1)
  hipStream_t side = nullptr;
  bool side_owned = true;       // false: the placement pool's weight-gradient stream (csrc/streams.hip)
  hipStream_t side2 = nullptr;  // PXL_WGRAD_STREAMS=2: every other fork goes to a second placed stream (never owned)
  hipEvent_t join_ev2 = nullptr;
  int fork_count = 0;
  std::vector<hipEvent_t> fork_ev;
  hipEvent_t join_ev = nullptr;
  int use_side = -1;
  // gradient exchange overlapped with the backward pass: contiguous buckets of the flat gradient buffer are all-reduced
  // on a communication stream as soon as every kernel that writes into them has been issued (north_star: "RCCL all-reduce
  // of gradients over xGMI overlapped with backward"; replaces nn.DataParallel's reduction, pixelssl/nn/func.py:54-62)
  pxl_allreduce_fn grad_sync = nullptr;
  void* grad_user = nullptr;
  long grad_bucket = 0;            // floats per bucket
  long grad_total = 0;             // floats in the flat gradient buffer
  hipStream_t comm_stream = nullptr;
  bool comm_owned = true;
  hipEvent_t comm_main_ev = nullptr, comm_side_ev = nullptr, comm_done_ev = nullptr;
  // parameter update pipelined behind the backward pass (pxl_net_set_update_hook): the same buckets, handed to the host's
  // optimizer as soon as their gradients are complete (and, multi-rank, all-reduced) -- SGD / EMA / weight re-packing of a
  // bucket run on the communication stream next to the data gradients of the layers below instead of after the pass
  pxl_update_fn update_fn = nullptr;
  void* update_user = nullptr;
  long update_bucket = 0;          // floats per update bucket (0: one call at the end)
  long update_total = 0;
  long update_tail = 0;            // a bucket boundary is forced where at most this many floats of parameters remain below
  int update_buckets_last = 0;
  std::vector<long> op_lo;         // per op: lowest flat offset (floats) its backward writes a gradient to, or -1
  bool bucket_ok = false;          // parameter offsets grow with the op index: suffixes of the op list = suffixes of the buffer
  int grad_buckets_last = 0;       // buckets issued by the last backward (tests / bench)
  int pairs_last = 0;              // convolutions the last paired forward issued as one launch for both networks
  int pair_syncs_last = 0;         // ... Sync-BN exchanges it issued for both networks at once
  int tune_dual = -1;              // pxl_net_set_tune_dual: forward tiles timed with two copies in flight (-1: PXL_TUNE_DUAL, default off)
  int bn_repeat = 1;               // running statistics updated as if this pass ran bn_repeat times (pxl_net_set_bn_repeat)
  float eff_momentum(float m) const { return bn_repeat <= 1 ? m : 1.f - powf(1.f - m, (float)bn_repeat); }
  bool pair_sync = getenv("PXL_PAIR_SYNC") == nullptr || getenv("PXL_PAIR_SYNC")[0] != '0';
  bool bn_onload = getenv("PXL_BN_ONLOAD") == nullptr || getenv("PXL_BN_ONLOAD")[0] != '0';
  bool wgrad_on = true;
  bool pack_dgrad = true;          // false: pxl_net_pack skips the transposed (data-gradient) weights (no-grad networks)
  int input_tensor = -1;
  bool latent_seeded = false;      // pxl_net_seed_latent_grad ran: the next backward starts from that gradient
  bool fuse_bn_reduce = getenv("PXL_FUSE_BN_REDUCE") == nullptr || getenv("PXL_FUSE_BN_REDUCE")[0] != '0';
  // backward of a residual join (ReLU mask + the main branch BN's sums) inside the data gradient that completes the
  // gradient of the join's output (PXL_FUSE_JOIN=0: separate pxl_residual_bwd_reduce launch)
  bool fuse_join = getenv("PXL_FUSE_JOIN") == nullptr || getenv("PXL_FUSE_JOIN")[0] != '0';
  // fp32 engine on the LDS-DMA kernels (conv_dma_f32.hip, conv_wgrad_dma_f32.hip): like the bf16 engine it then materialises
  // relu(bn(y)) for its convolutions (they read plain operands) and fuses the BatchNorm-backward sums into the data
  // gradients.  PXL_F32_DMA=0: the generic kernels with BN-apply in their prologue (rounds 1-3), for A/B runs
  bool f32_dma = getenv("PXL_F32_DMA") == nullptr || getenv("PXL_F32_DMA")[0] != '0';
  bool plain_operands() const { return dtype == PXL_BF16 || (dtype == PXL_F32 && f32_dma); }
  // folding the forward finalize into its consumer removes 104 launches per pass; every block of the consumer re-reduces
  // the statistics replicas, which was slower with 32 replicas (round 1: 15.7 vs 15.0 ms / step) and is faster with 4
  // (round 2: 14.98 vs 15.17): on by default, PXL_FUSE_BN_FINALIZE=0 restores the separate pxl_bn_finalize launches
  bool fuse_bn_finalize = getenv("PXL_FUSE_BN_FINALIZE") == nullptr || getenv("PXL_FUSE_BN_FINALIZE")[0] != '0';
  // the producing convolution's last workgroup finalizes the BatchNorm (conv_dma.hip: pxl_conv_dma_finalize); one rank only
  // (Sync-BN all-reduces the statistics between the convolution and the finalize).  Measured negative result, kept
  // opt-in (PXL_CONV_FINALIZE=1): the ticket atomic + two extra barriers in EVERY workgroup's epilogue cost more than
  // the consumer-side fold saves -- MT 15.3-15.5 ms / step vs 14.8 (and 26.7 ms with a __threadfence() per block)
  bool conv_finalize = getenv("PXL_CONV_FINALIZE") != nullptr && getenv("PXL_CONV_FINALIZE")[0] == '1';
  // tests: use the reference's multi-device variance formula clamp(var, eps) on a single rank too
  bool force_clamp = getenv("PXL_FORCE_CLAMP_VAR") != nullptr;
  // PXL_DETERMINISTIC=1: a bit-reproducible FORWARD pass.  The forward's only order-dependent arithmetic is fp32 atomics: the
  // BatchNorm statistics that the convolution epilogues add into a few replicas, and split-K partial sums.  Here every
  // statistics vector gets one replica per 64 pixel rows -- a replica then receives at most two adds (tiles are >= 32 rows), and
  // a + b is commutative --, the replicas are folded in index order by ONE kernel (no finalize folded into consumers, no
  // BN-apply on load: those re-reduce the replicas in every workgroup), and no convolution splits K.  Slower (thousands of
  // replicas for the early layers); for parity runs: pre-activations within an ulp of zero no longer take a different ReLU
  // branch from run to run (tools/diag_2rank.py).  Round 5: the BACKWARD pass as well -- the BatchNorm-backward sums get one
  // replica per 64 pixel rows (data-gradient epilogues) / per row group (reduce kernels), folded in index order; weight gradients
  // run with ONE pixel split (every element of dw receives one add); bias column sums with one block per column slab; the
  // training seam on the row-wise kernel with its loss sums folded in row order.  What stays unordered: the IBNorm statistics of
  // the flaw detector, the flaw-map / CCT mask reductions and the loss sums of the stand-alone criterion kernels (values only).
  bool deterministic = getenv("PXL_DETERMINISTIC") != nullptr && getenv("PXL_DETERMINISTIC")[0] == '1';
};

namespace {

int build_conv_descs(pxl_net* n, OpInfo& op, const TensorInfo& tin, const TensorInfo& tout) {
  const pxl_op& d = op.d;
  const int tpg = d.kh * d.kw;
  op.ntaps = tpg * d.ngroups;
  if (op.ntaps > 64) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net: conv with %d taps (max 64)", op.ntaps);
  pxl_conv_desc f;
  std::memset(&f, 0, sizeof(f));
  f.dtype = n->dtype;
  f.B = n->B; f.Hi = tin.H; f.Wi = tin.W; f.Cin = tin.Cp;
  f.Ho = tout.H; f.Wo = tout.W; f.Cout = tout.Cp; f.Kreal = d.cout;
  f.ntaps = op.ntaps; f.out_stride = d.stride; f.div = 1;
  f.relu_in = d.bn_in0 >= 0 ? 1 : 0;
  f.tile_cfg = -1;
  f.stats_rep = STATS_REP;
  f.split_k = 0;
  for (int g = 0; g < d.ngroups; ++g)
    for (int r = 0; r < d.kh; ++r)
      for (int s = 0; s < d.kw; ++s) {
        const int t = g * tpg + r * d.kw + s;
        f.dy[t] = (int16_t)(r * d.dil[g] - d.pads[g]);
        f.dx[t] = (int16_t)(s * d.dil[g] - d.pads[g]);
      }
  op.fwd = f;
  // data gradient: roles of in/out swapped, negated taps, gather-with-divisor for strided convs
  pxl_conv_desc b = f;
  b.Hi = tout.H; b.Wi = tout.W; b.Cin = tout.Cp;
  b.Ho = tin.H; b.Wo = tin.W; b.Cout = tin.Cp; b.Kreal = d.cin;
  b.out_stride = 1; b.div = d.stride; b.relu_in = 0; b.stats_rep = 1;
  for (int t = 0; t < op.ntaps; ++t) { b.dy[t] = (int16_t)(-f.dy[t]); b.dx[t] = (int16_t)(-f.dx[t]); }
  op.bwd = b;
  for (int g = 0; g < d.ngroups; ++g) {
    pxl_conv_desc q = f;
    q.ntaps = tpg;
    for (int t = 0; t < tpg; ++t) { q.dy[t] = f.dy[g * tpg + t]; q.dx[t] = f.dx[g * tpg + t]; }
    op.grp[g] = q;
  }
  if (d.stride != 1 && d.stride != 2) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net: stride %d", d.stride);
  return PXL_OK;
}

struct Timed {
  // records an event pair around one launch when profiling is on
  pxl_net* n; hipStream_t s; int kind; double flops; hipEvent_t a = nullptr, b = nullptr;
  Timed(pxl_net* n_, hipStream_t s_, int kind_, double flops_) : n(n_), s(s_), kind(kind_), flops(flops_) {
    if (!n->profile) return;
    auto get = [&]() { hipEvent_t e = nullptr; if (!n->pool.empty()) { e = n->pool.back(); n->pool.pop_back(); } else if (hipEventCreate(&e) != hipSuccess) e = nullptr; return e; };
    a = get(); b = get();
    if (a) (void)hipEventRecord(a, s);
  }
  ~Timed() {
    if (!n->profile || !a || !b) return;
    (void)hipEventRecord(b, s);
    n->stamps.push_back({a, b, kind, flops});
  }
};

// operand a convolution (forward and weight gradient) reads: the raw tensor, the materialised relu(bn(y)),
// or the raw tensor + the fused (scale, shift) prologue
struct ConvIn { const void* ptr; const float* sc; const float* sh; };
inline ConvIn conv_input(const pxl_net* n, const OpInfo& op, const void* arena) {
  // (weight gradient / tuning view: the materialised activation of an on-load BN, like any other has_z BN)
  const pxl_op& d = op.d;
  const TensorInfo& tin = n->tensors[d.in0];
  const unsigned char* base = reinterpret_cast<const unsigned char*>(arena);
  if (op.patch) return {n->patch_src != nullptr ? reinterpret_cast<const unsigned char*>(n->patch_src) : base + op.patch_off, nullptr, nullptr};
  if (d.bn_in0 < 0) return {base + tin.off, nullptr, nullptr};
  const BnInfo& b = n->bns[d.bn_in0];
  if (b.has_z) return {base + b.z_off, nullptr, nullptr};
  const float* coef = reinterpret_cast<const float*>(base + b.coef_off);
  return {base + tin.off, coef + 2 * b.d.C, coef + 3 * b.d.C};
}

// algorithmic HBM bytes of one contraction launch: every operand once (activations in, weights, output)
inline double conv_bytes(const pxl_net* n, const pxl_op& d, const TensorInfo& tin, const TensorInfo& tout, bool wgrad) {
  const double e = n->esize;
  const double act_in = (double)n->B * tin.H * tin.W * tin.C * e, act_out = (double)n->B * tout.H * tout.W * tout.C * e;
  const double w = (double)d.cout * d.kh * d.kw * d.ngroups * d.cin;
  return wgrad ? act_in + act_out + w * 4.0 : act_in + act_out + w * e;
}

inline double conv_flops(const pxl_net* n, const pxl_op& d, const TensorInfo& tout) {
  return 2.0 * n->B * tout.H * tout.W * (double)d.cout * d.kh * d.kw * d.ngroups * d.cin;
}

inline unsigned char* at(void* base, size_t off) { return reinterpret_cast<unsigned char*>(base) + off; }
inline const unsigned char* at(const void* base, size_t off) { return reinterpret_cast<const unsigned char*>(base) + off; }
inline float* fat(void* base, size_t off) { return reinterpret_cast<float*>(at(base, off)); }
inline const float* fat(const void* base, size_t off) { return reinterpret_cast<const float*>(at(base, off)); }

}  // namespace

extern "C" int pxl_net_create(int dtype, int num_classes, const pxl_op* ops, int nops, const pxl_bn_desc* bns,
                              int nbns, int ntensors, pxl_net** out) {
  PXL_REQUIRE(ops && out && nops > 0 && ntensors > 0, "net_create: bad argument");
  PXL_REQUIRE(dtype == PXL_F32 || dtype == PXL_BF16, "net_create: bad dtype %d", dtype);
  pxl_net* n = new (std::nothrow) pxl_net();
  PXL_REQUIRE(n != nullptr, "net_create: out of host memory");
  n->dtype = dtype;
  n->esize = dtype == PXL_F32 ? 4 : 2;
  n->classes = num_classes;
  if (n->deterministic) { n->bn_onload = false; n->fuse_bn_finalize = false; n->conv_finalize = false; }
  n->ops.resize(nops);
  n->bns.resize(nbns);
  n->tensors.resize(ntensors);
  for (int i = 0; i < nbns; ++i) n->bns[i].d = bns[i];
  for (int i = 0; i < nops; ++i) {
    n->ops[i].d = ops[i];
    const pxl_op& d = ops[i];
    auto bad_t = [&](int t) { return t < -1 || t >= ntensors; };
    auto bad_b = [&](int b) { return b < -1 || b >= nbns; };
    if (bad_t(d.in0) || bad_t(d.in1) || bad_t(d.out) || bad_b(d.bn_in0) || bad_b(d.bn_in1) || bad_b(d.bn_out)) {
      delete n;
      return pxl_set_error(PXL_ERR_ARG, "net_create: op %d references an unknown tensor/BN", i);
    }
    if (d.kind == PXL_OP_CONV && (d.ngroups < 1 || d.ngroups > 4)) {
      delete n;
      return pxl_set_error(PXL_ERR_ARG, "net_create: op %d has %d tap groups", i, d.ngroups);
    }
    // the ReLU after a BN is decided by its consumer kind
    if ((d.kind == PXL_OP_CONV || d.kind == PXL_OP_MAXPOOL || d.kind == PXL_OP_UPCAT) && d.bn_in0 >= 0) n->bns[d.bn_in0].relu = 1;
    if (d.kind == PXL_OP_HEAD && d.bn_in1 >= 0) n->bns[d.bn_in1].relu = 1;
    if (d.kind == PXL_OP_CONV && d.bn_out >= 0) n->bns[d.bn_out].y_tensor = d.out;
    if (d.kind == PXL_OP_HEAD) n->head_op = i;
    if (d.kind == PXL_OP_INPUT) n->input_tensor = d.out;
  }
  if (n->head_op < 0) {
    delete n;
    return pxl_set_error(PXL_ERR_ARG, "net_create: program has no HEAD op");
  }
  *out = n;
  return PXL_OK;
}

extern "C" void pxl_net_destroy(pxl_net* net) {
  if (!net) return;
  for (auto& st : net->stamps) { (void)hipEventDestroy(st.a); (void)hipEventDestroy(st.b); }
  for (auto e : net->pool) (void)hipEventDestroy(e);
  for (auto e : net->fork_ev) if (e) (void)hipEventDestroy(e);
  if (net->join_ev) (void)hipEventDestroy(net->join_ev);
  if (net->join_ev2) (void)hipEventDestroy(net->join_ev2);
  if (net->side && net->side_owned) (void)hipStreamDestroy(net->side);
  for (hipEvent_t e : {net->comm_main_ev, net->comm_side_ev, net->comm_done_ev}) if (e) (void)hipEventDestroy(e);
  if (net->comm_stream && net->comm_owned) (void)hipStreamDestroy(net->comm_stream);
  delete net;
}

extern "C" int pxl_net_set_sync(pxl_net* net, pxl_allreduce_fn fn, void* user, int world_size) {
  PXL_REQUIRE(net && world_size >= 1, "net_set_sync: bad argument");
  net->sync = fn;
  net->sync_user = user;
  net->world = world_size;
  return PXL_OK;
}

extern "C" int pxl_net_set_grad_sync(pxl_net* net, pxl_allreduce_fn fn, void* user, int world_size, long bucket_floats,
                                     long total_floats) {
  PXL_REQUIRE(net && world_size >= 1 && bucket_floats >= 0 && total_floats >= 0, "net_set_grad_sync: bad argument");
  net->grad_sync = fn;
  net->grad_user = user;
  net->grad_world = world_size;      // (not `world`: a network may average gradients over ranks while its BatchNorms stay local)
  net->grad_bucket = bucket_floats;
  net->grad_total = total_floats;
  return PXL_OK;
}

extern "C" int pxl_net_grad_buckets(const pxl_net* net) { return net ? net->grad_buckets_last : 0; }

extern "C" int pxl_net_set_update_hook(pxl_net* net, pxl_update_fn fn, void* user, long bucket_floats, long tail_floats,
                                       long total_floats) {
  PXL_REQUIRE(net && bucket_floats >= 0 && tail_floats >= 0 && total_floats >= 0, "net_set_update_hook: bad argument");
  net->update_fn = fn;
  net->update_user = user;
  net->update_bucket = bucket_floats;
  net->update_tail = tail_floats;
  net->update_total = total_floats;
  return PXL_OK;
}

extern "C" int pxl_net_update_buckets(const pxl_net* net) { return net ? net->update_buckets_last : 0; }

extern "C" int pxl_net_profile(pxl_net* net, int enable) {
  PXL_REQUIRE(net, "net_profile: null net");
  net->profile = enable != 0;
  return PXL_OK;
}

extern "C" int pxl_net_profile_read(pxl_net* net, int kind, double* ms, long* launches, double* flops) {
  PXL_REQUIRE(net && ms && launches && flops, "net_profile_read: null argument");
  double t = 0.0, f = 0.0;
  long c = 0;
  std::vector<pxl_net::Stamp> keep;
  for (auto& st : net->stamps) {
    if (st.kind != kind) { keep.push_back(st); continue; }
    PXL_CHECK_HIP(hipEventSynchronize(st.b));
    float e = 0.f;
    PXL_CHECK_HIP(hipEventElapsedTime(&e, st.a, st.b));
    t += e; f += st.flops; ++c;
    net->pool.push_back(st.a);
    net->pool.push_back(st.b);
  }
  net->stamps.swap(keep);
  *ms = t; *launches = c; *flops = f;
  return PXL_OK;
}

extern "C" int pxl_net_profile_bytes(pxl_net* net, int kind, double* bytes) {
  PXL_REQUIRE(net && bytes && (kind == 0 || kind == 1), "net_profile_bytes: bad argument");
  *bytes = net->prof_bytes[kind];
  net->prof_bytes[kind] = 0.0;
  return PXL_OK;
}

extern "C" int pxl_net_plan(pxl_net* n, int B, int H, int W) { return pxl_net_plan_out(n, B, H, W, H, W); }

extern "C" int pxl_net_plan_out(pxl_net* n, int B, int H, int W, int Hout, int Wout) {
  PXL_REQUIRE(n && B > 0 && H > 0 && W > 0 && Hout > 0 && Wout > 0, "net_plan: bad argument");
  n->B = B; n->H = H; n->W = W; n->Ho = Hout; n->Wo = Wout;
  n->planned = false;
  for (auto& t : n->tensors) t = TensorInfo();
  size_t arena = 0, scratch = 0, packed = 0;
  // (BN statistics: contiguous -> one memset per pass; allocated after the ops below, when every BatchNorm's pixel count is known)
  for (auto& b : n->bns) b.M = 0;
  for (auto& b : n->bns) { b.coef_off = arena; arena += align_up(4 * (size_t)b.d.C * 4); }
  // (the backward sums are planned after the ops, when every BatchNorm's pixel count is known: deterministic mode replicates them)

  auto plan_tensor = [&](int id, int h, int w, int c) {
    TensorInfo& t = n->tensors[id];
    t.H = h; t.W = w; t.C = c; t.Cp = pitch_of(c);
    t.bytes = align_up((size_t)B * h * w * t.Cp * n->esize);
    t.off = arena; arena += t.bytes;
    t.goff = scratch; scratch += t.bytes;
    t.planned = true;
  };

  for (size_t i = 0; i < n->ops.size(); ++i) {
    OpInfo& op = n->ops[i];
    const pxl_op& d = op.d;
    switch (d.kind) {
      case PXL_OP_INPUT:
        plan_tensor(d.out, H, W, d.cout);
        break;
      case PXL_OP_CONV: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& tin = n->tensors[d.in0];
        PXL_REQUIRE(tin.C == d.cin, "net_plan: op %zu expects %d input channels, tensor has %d", i, d.cin, tin.C);
        const int eff_h = d.dil[0] * (d.kh - 1) + 1, eff_w = d.dil[0] * (d.kw - 1) + 1;
        const int ho = (tin.H + 2 * d.pads[0] - eff_h) / d.stride + 1;
        const int wo = (tin.W + 2 * d.pads[0] - eff_w) / d.stride + 1;
        for (int g = 1; g < d.ngroups; ++g) {
          const int eh = d.dil[g] * (d.kh - 1) + 1;
          PXL_REQUIRE((tin.H + 2 * d.pads[g] - eh) / d.stride + 1 == ho, "net_plan: op %zu tap groups disagree on output size", i);
        }
        plan_tensor(d.out, ho, wo, d.cout);
        int rc = build_conv_descs(n, op, n->tensors[d.in0], n->tensors[d.out]);
        if (rc != PXL_OK) return rc;
        const TensorInfo& tout = n->tensors[d.out];
        op.patch = false;
        // (fp32 engine too, round 6: its stem ran on the generic kernels -- 2 x 624 us forward, 1441 us weight gradient at the tail of
        // the backward pass; PXL_STEM_PATCHES_F32=0 restores that)
        static const bool patches_f32 = getenv("PXL_STEM_PATCHES_F32") == nullptr || getenv("PXL_STEM_PATCHES_F32")[0] != '0';
        if (n->stem_patches && (n->dtype == PXL_BF16 || (n->dtype == PXL_F32 && patches_f32)) && d.in0 == n->input_tensor && !d.need_dgrad && d.ngroups == 1 &&
            d.bn_in0 < 0 && d.kh * d.kw > 1 && d.cin * d.kh * d.kw <= 256 && tout.Cp % 8 == 0) {
          op.patch = true;
          op.patch_K = d.cin * d.kh * d.kw;
          op.patch_Kp = (op.patch_K + 63) / 64 * 64;
          op.patch_off = arena; arena += align_up((size_t)B * ho * wo * op.patch_Kp * n->esize);
          pxl_conv_desc f = op.fwd;
          f.Hi = ho; f.Wi = wo; f.Cin = op.patch_Kp; f.ntaps = 1; f.out_stride = 1; f.div = 1; f.relu_in = 0;
          f.dy[0] = f.dx[0] = 0;
          op.fwd = f;
          op.grp[0] = f;
          op.wf_off = packed; packed += align_up((size_t)d.cout * op.patch_Kp * n->esize);
        } else {
          op.wf_off = packed; packed += align_up((size_t)d.cout * op.ntaps * tin.Cp * n->esize);
        }
        if (d.need_dgrad) { op.wt_off = packed; packed += align_up((size_t)tin.Cp * op.ntaps * tout.Cp * n->esize); }
        if (d.b_off[0] >= 0) { op.bias_off = packed; packed += align_up((size_t)d.cout * 4); }
        if (d.bn_out < 0 && tout.Cp <= 32) {     // few output tiles + long reduction: allow split-K
          // (room for one partial-sum slab per K slice -- up to 16 -- so that the slices store instead of adding with atomics)
          op.ws_bytes = align_up((size_t)16 * B * ho * wo * tout.Cp * 4);
          // the head as one GEMM (aspp.hip; PXL_ASPP_GEMM=0: the 36-tap convolution of rounds 1-5): several tap groups of a "same"
          // stride-1 convolution over a plain input with 64-channel granules
          static const bool pg_on = getenv("PXL_ASPP_GEMM") == nullptr || getenv("PXL_ASPP_GEMM")[0] != '0';
          op.pg = false;
          if (pg_on && d.ngroups > 1 && d.stride == 1 && ho == tin.H && wo == tin.W && d.bn_in0 < 0 && tin.Cp % 64 == 0 &&
              tin.Cp == d.cin && (long)d.cout * d.kh * d.kw <= 1024) {
            const int tpg2 = d.kh * d.kw;
            op.pg_GP = (d.cout * tpg2 + 63) / 64 * 64;
            op.pg_J = d.ngroups * op.pg_GP;
            pxl_conv_desc f;
            std::memset(&f, 0, sizeof(f));
            f.dtype = n->dtype; f.B = B; f.Hi = tin.H; f.Wi = tin.W; f.Cin = tin.Cp; f.Ho = ho; f.Wo = wo; f.Cout = op.pg_J; f.Kreal = op.pg_J;
            f.ntaps = 1; f.out_stride = 1; f.div = 1; f.tile_cfg = -1; f.stats_rep = 1; f.split_k = 1;
            op.pg_fwd = f;
            op.pg_grp = f; op.pg_grp.split_k = 0;
            pxl_conv_desc b = f;
            b.Cin = op.pg_J; b.Cout = tin.Cp; b.Kreal = d.cin; b.split_k = 0;
            op.pg_bwd = b;
            if (pxl_conv_dma_eligible(&op.pg_fwd, nullptr, nullptr) && pxl_conv_dma_eligible(&op.pg_bwd, nullptr, nullptr)) {
              op.pg = true;
              op.ws_bytes = std::max(op.ws_bytes, align_up((size_t)B * ho * wo * op.pg_J * 4));          // P: ONE fp32 slab
              op.pg_wf_off = packed; packed += align_up((size_t)op.pg_J * tin.Cp * n->esize);
              if (d.need_dgrad) { op.pg_wt_off = packed; packed += align_up((size_t)tin.Cp * op.pg_J * n->esize); }
              op.pg_dp_off = scratch; scratch += align_up((size_t)B * ho * wo * op.pg_J * n->esize);
              op.pg_dw_off = scratch; scratch += align_up((size_t)op.pg_J * d.cin * 4);
            }
          }
          op.ws_off = arena; arena += op.ws_bytes;
        }
        if (d.bn_out >= 0) {
          PXL_REQUIRE(n->bns[d.bn_out].d.C == d.cout, "net_plan: BN %d has %d channels, conv %zu has %d", d.bn_out,
                      n->bns[d.bn_out].d.C, i, d.cout);
          n->bns[d.bn_out].M = B * ho * wo;
        }
        break;
      }
      case PXL_OP_MAXPOOL: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& tin = n->tensors[d.in0];
        plan_tensor(d.out, (tin.H + 2 - 3) / 2 + 1, (tin.W + 2 - 3) / 2 + 1, tin.C);
        op.idx_off = arena;
        arena += align_up((size_t)B * n->tensors[d.out].H * n->tensors[d.out].W * n->tensors[d.out].Cp);
        break;
      }
      case PXL_OP_RESIDUAL: {
        PXL_REQUIRE(d.in0 >= 0 && d.in1 >= 0 && n->tensors[d.in0].planned && n->tensors[d.in1].planned,
                    "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& a = n->tensors[d.in0];
        const TensorInfo& r = n->tensors[d.in1];
        PXL_REQUIRE(a.H == r.H && a.W == r.W && a.C == r.C, "net_plan: residual op %zu shape mismatch", i);
        PXL_REQUIRE(d.bn_in0 >= 0, "net_plan: residual op %zu needs a BN on its main branch", i);
        plan_tensor(d.out, a.H, a.W, a.C);
        break;
      }
      case PXL_OP_ACT: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& tin = n->tensors[d.in0];
        plan_tensor(d.out, tin.H, tin.W, tin.C);
        op.ws_bytes = 0;
        if (d.bn_in0 >= 0) {          // LeakyReLU(bn(y)): bn(y) is kept (the backward needs its sign)
          PXL_REQUIRE(n->bns[d.bn_in0].d.C == tin.Cp, "net_plan: activation op %zu: BN %d has %d channels, the tensor pitch is %d",
                      i, d.bn_in0, n->bns[d.bn_in0].d.C, tin.Cp);
          op.ws_bytes = align_up(tin.bytes);
          op.ws_off = arena; arena += op.ws_bytes;
        }
        break;
      }
      case PXL_OP_IBN: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        PXL_REQUIRE(d.bn_out >= 0, "net_plan: IBNorm op %zu names no BN half", i);
        const TensorInfo& tin = n->tensors[d.in0];
        PXL_REQUIRE(tin.Cp == tin.C, "net_plan: IBNorm op %zu needs an unpadded channel count (%d)", i, tin.C);
        const int nb = n->bns[d.bn_out].d.C;
        PXL_REQUIRE(nb >= 1 && nb <= tin.C, "net_plan: IBNorm op %zu: BN half %d of %d channels", i, nb, tin.C);
        plan_tensor(d.out, tin.H, tin.W, tin.C);
        // (ibn_sums / ibn_bsums: one contiguous region per pass, planned after this loop -> one memset instead of one per layer)
        op.ibn_bn = arena; arena += align_up(2 * (size_t)nb * 4);
        op.ibn_coef = arena; arena += align_up((size_t)B * 4 * tin.Cp * 4);
        op.ibn_bbn = scratch; scratch += align_up(2 * (size_t)nb * 4);
        break;
      }
      case PXL_OP_AVGPOOL: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned && d.kh >= 1, "net_plan: op %zu consumes an unplanned tensor", i);
        plan_tensor(d.out, d.kh, d.kh, n->tensors[d.in0].C);
        break;
      }
      case PXL_OP_CONCAT: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& tin = n->tensors[d.in0];
        PXL_REQUIRE(d.cout >= tin.C && tin.C == tin.Cp && tin.C % 8 == 0, "net_plan: concat op %zu: %d channels into %d", i, tin.C, d.cout);
        plan_tensor(d.out, tin.H, tin.W, d.cout);
        break;
      }
      case PXL_OP_UPCAT: {
        PXL_REQUIRE(d.in0 >= 0 && d.out >= 0 && n->tensors[d.in0].planned && n->tensors[d.out].planned,
                    "net_plan: op %zu needs its input and the concat tensor planned before it", i);
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        PXL_REQUIRE(d.c_off >= 0 && d.c_off % 8 == 0 && tin.C % 8 == 0 && d.c_off + tin.C <= tout.C,
                    "net_plan: op %zu writes channels [%d, %d) of a %d-channel tensor", i, d.c_off, d.c_off + tin.C, tout.C);
        break;
      }
      case PXL_OP_PIXSHUF: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: op %zu consumes an unplanned tensor", i);
        const TensorInfo& tin = n->tensors[d.in0];
        PXL_REQUIRE(tin.C % 4 == 0, "net_plan: PixelShuffle op %zu on %d channels", i, tin.C);
        plan_tensor(d.out, 2 * tin.H, 2 * tin.W, tin.C / 4);
        break;
      }
      case PXL_OP_HEAD: {
        PXL_REQUIRE(d.in0 >= 0 && n->tensors[d.in0].planned, "net_plan: head consumes an unplanned tensor");
        PXL_REQUIRE(n->tensors[d.in0].C == n->classes, "net_plan: head input has %d channels, expected %d",
                    n->tensors[d.in0].C, n->classes);
        n->up_ws_off = scratch;
        // (+ B * Ho * 3 floats: per-row loss partials of the ordered seam kernel, PXL_DETERMINISTIC)
        n->up_ws_bytes = align_up((size_t)B * n->Ho * n->tensors[d.in0].W * n->classes * 4 + (size_t)B * n->Ho * 3 * 4);
        scratch += n->up_ws_bytes;
        break;
      }
      default:
        return pxl_set_error(PXL_ERR_ARG, "net_plan: unknown op kind %d", d.kind);
    }
  }
  n->bsum_region_off = scratch;
  for (auto& b : n->bns) {
    // PXL_DETERMINISTIC: one replica per 64 pixel rows for the data-gradient epilogues (tile rows are >= 64 rows apart there is at
    // most one add per replica and channel per launch) and at least 256 for the stand-alone reduce kernels (<= 256 row groups)
    b.bnrep = (n->deterministic && b.M > 0) ? std::max((b.M + 63) / 64, 256) : 1;
    b.bsum_off = scratch; scratch += align_up((size_t)b.bnrep * 2 * (size_t)b.d.C * 4);
  }
  n->bsum_region_bytes = scratch - n->bsum_region_off;
  for (auto& b : n->bns) { b.bcoef_off = scratch; scratch += align_up(2 * (size_t)b.d.C * 4); }
  n->stats_region_off = arena;
  for (auto& b : n->bns) {
    b.nrep = (n->deterministic && b.M > 0) ? (b.M + 63) / 64 : STATS_REP;
    b.stats_off = arena; arena += align_up((size_t)b.nrep * 2 * (size_t)b.d.C * 4);
    b.cnt_off = arena; arena += ALIGN;
  }
  n->stats_region_bytes = arena - n->stats_region_off;
  for (auto& op : n->ops) {
    if (op.d.kind != PXL_OP_CONV) continue;
    if (op.d.bn_out >= 0) op.fwd.stats_rep = n->bns[op.d.bn_out].nrep;
    if (n->deterministic) {
      op.fwd.split_k = 1;
      for (int g = 0; g < op.d.ngroups; ++g) op.grp[g].split_k = 1;      // weight gradients: ONE pixel split = one add per element
    }
  }
  for (auto& b : n->bns) b.has_z = false;
  for (auto& op : n->ops) {
    const pxl_op& d = op.d;
    int bn = -1, t = -1;
    if (d.kind == PXL_OP_CONV && d.bn_in0 >= 0 && n->plain_operands()) { bn = d.bn_in0; t = d.in0; }
    if (d.kind == PXL_OP_HEAD && d.bn_in1 >= 0) { bn = d.bn_in1; t = d.in1; }     // activated latent: any dtype
    if (bn < 0) continue;
    BnInfo& b = n->bns[bn];
    const TensorInfo& tin = n->tensors[t];
    if (d.kind == PXL_OP_HEAD) PXL_REQUIRE(b.y_tensor == t && tin.Cp == b.d.C, "net_plan: latent BN %d does not belong to tensor %d", bn, t);
    if (b.y_tensor != t || tin.Cp % 64 != 0 || tin.Cp != b.d.C || b.has_z) continue;
    b.has_z = true;
    b.z_off = arena;
    arena += tin.bytes;
  }
  // BN-apply on load (see BnInfo::onload): materialised BNs with exactly one consumer, a convolution the LDS-DMA kernel
  // can run with its coefficient table in LDS
  for (auto& b : n->bns) b.onload = false;
  // (fp32, round 6: conv_dma_f32.hip applies the BatchNorm on load too -- PXL_BN_ONLOAD_F32=0: materialised as in rounds 1-5)
  static const bool onload_f32 = getenv("PXL_BN_ONLOAD_F32") == nullptr || getenv("PXL_BN_ONLOAD_F32")[0] != '0';
  if ((n->dtype == PXL_BF16 || (n->dtype == PXL_F32 && onload_f32)) && n->bn_onload) {
    std::vector<int> ncons(n->bns.size(), 0), conv_of(n->bns.size(), -1);
    for (size_t i = 0; i < n->ops.size(); ++i) {
      const pxl_op& d = n->ops[i].d;
      if (d.bn_in0 >= 0) { ++ncons[d.bn_in0]; if (d.kind == PXL_OP_CONV) conv_of[d.bn_in0] = (int)i; }
      if (d.bn_in1 >= 0) ncons[d.bn_in1] += 2;           // a second operand (join shortcut, HEAD latent): not on-load
    }
    for (size_t k = 0; k < n->bns.size(); ++k) {
      BnInfo& b = n->bns[k];
      if (!b.has_z || ncons[k] != 1 || conv_of[k] < 0) continue;
      const OpInfo& oc = n->ops[conv_of[k]];
      if (oc.patch || oc.ws_bytes != 0 || oc.fwd.Cin > 512 || oc.fwd.Cin != b.d.C) continue;
      if (!pxl_conv_dma_eligible(&oc.fwd, nullptr, nullptr)) continue;
      // 1x1 / stride-1 consumers only (conv3 of a bottleneck).  Measured on the MI355X (profiles/r03_c_*): for a 3x3
      // consumer every tap re-transforms the tile (9x the work of the materialising kernel) inside a K loop that is
      // already latency-bound -- 59 vs 40 us per layer3 convolution in the step, more than the 12 us launch it removes
      const pxl_conv_desc& f = oc.fwd;
      if (!(f.ntaps == 1 && f.dy[0] == 0 && f.dx[0] == 0 && f.out_stride == 1 && f.Ho == f.Hi && f.Wo == f.Wi)) continue;
      b.onload = true;
    }
  }
  // forward finalize folded into its consumer: BNs that are materialised (z) or whose raw tensor feeds only one
  // residual join
  {
    std::vector<int> nuse(n->tensors.size(), 0), res_use(n->tensors.size(), 0);
    for (auto& op : n->ops) {
      const pxl_op& d = op.d;
      if (d.in0 >= 0) ++nuse[d.in0];
      if (d.in1 >= 0) ++nuse[d.in1];
      if (d.kind == PXL_OP_RESIDUAL) { ++res_use[d.in0]; if (d.in1 >= 0) ++res_use[d.in1]; }
    }
    for (auto& b : n->bns) {
      b.fin_in_consumer = false;
      if (!n->fuse_bn_finalize || b.y_tensor < 0) continue;
      const TensorInfo& ty = n->tensors[b.y_tensor];
      if (ty.Cp != b.d.C) continue;
      if (b.has_z) b.fin_in_consumer = true;
      else if (nuse[b.y_tensor] == 1 && res_use[b.y_tensor] == 1) b.fin_in_consumer = true;
    }
    // a residual join folds either both of its BNs or none
    for (auto& op : n->ops) {
      const pxl_op& d = op.d;
      if (d.kind != PXL_OP_RESIDUAL) continue;
      BnInfo& b0 = n->bns[d.bn_in0];
      const bool both = !b0.has_z && b0.fin_in_consumer && (d.bn_in1 < 0 || (!n->bns[d.bn_in1].has_z && n->bns[d.bn_in1].fin_in_consumer));
      if (!both) {
        if (!b0.has_z) b0.fin_in_consumer = false;
        if (d.bn_in1 >= 0 && !n->bns[d.bn_in1].has_z) n->bns[d.bn_in1].fin_in_consumer = false;
      }
    }
  }
  // BN-backward reduce fused into the data gradient that writes d(relu(bn(y))): y must have exactly one consumer
  // (that convolution) and the launch must be eligible for the LDS-DMA kernel
  for (auto& b : n->bns) b.fused_reduce_op = -1;
  if (n->plain_operands() && n->fuse_bn_reduce) {
    std::vector<int> uses(n->tensors.size(), 0);
    for (auto& op : n->ops) {
      const pxl_op& d = op.d;
      if (d.kind == PXL_OP_HEAD) { if (d.in1 >= 0) uses[d.in1] += 2; continue; }     // the latent may receive a seeded gradient
      if (d.in0 >= 0) ++uses[d.in0];
      if (d.in1 >= 0) ++uses[d.in1];
    }
    for (size_t i = 0; i < n->ops.size(); ++i) {
      OpInfo& op = n->ops[i];
      const pxl_op& d = op.d;
      if (d.kind != PXL_OP_CONV || d.bn_in0 < 0 || !d.need_dgrad) continue;     // (stride-2 data gradients run on the DMA kernel too)
      BnInfo& b = n->bns[d.bn_in0];
      const TensorInfo& tin = n->tensors[d.in0];
      if (b.y_tensor != d.in0 || uses[d.in0] != 1 || tin.Cp != tin.C || tin.C != b.d.C) continue;
      if (!pxl_conv_dma_eligible(&op.bwd, nullptr, nullptr)) continue;
      b.fused_reduce_op = (int)i;
    }
  }
  n->input_needed = false;
  for (auto& op : n->ops) {
    const pxl_op& d = op.d;
    if (d.kind == PXL_OP_INPUT) continue;
    if ((d.in0 == n->input_tensor && !(d.kind == PXL_OP_CONV && op.patch)) || d.in1 == n->input_tensor) n->input_needed = true;
  }
  // residual joins whose backward runs in the epilogue of the data gradient that completes d(join output): that
  // convolution must be the FIRST consumer of the output in program order (= the last contribution in the backward
  // pass), read it as a plain operand and run on the LDS-DMA kernel; every other consumer must be a convolution or the
  // identity input of the next join (they only add their share before it)
  for (auto& op : n->ops) op.join_op = op.join_conv = -1;
  if (n->plain_operands() && n->fuse_bn_reduce && n->fuse_join) {
    for (size_t j = 0; j < n->ops.size(); ++j) {
      const pxl_op& dj = n->ops[j].d;
      if (dj.kind != PXL_OP_RESIDUAL || dj.bn_in0 < 0) continue;
      const BnInfo& b3 = n->bns[dj.bn_in0];
      const TensorInfo& a = n->tensors[dj.in0];
      const TensorInfo& o = n->tensors[dj.out];
      if (b3.y_tensor != dj.in0 || b3.relu || a.Cp != b3.d.C || o.Cp != o.C || o.C != b3.d.C) continue;
      int first = -1;
      bool ok = true;
      for (size_t c = j + 1; c < n->ops.size() && ok; ++c) {
        const pxl_op& dc = n->ops[c].d;
        const bool uses = dc.in0 == dj.out || dc.in1 == dj.out;
        if (!uses) continue;
        if (first < 0) first = (int)c;
        if (dc.kind == PXL_OP_CONV) ok = dc.in0 == dj.out && dc.need_dgrad;
        else if (dc.kind == PXL_OP_RESIDUAL) ok = dc.in1 == dj.out && dc.in0 != dj.out;
        // (the HEAD op names the join output as its LATENT: it sends no gradient of its own there -- a seeded latent gradient,
        // pxl_net_seed_latent_grad, arrives as the addend of the convolution that completes the join's gradient)
        else if (dc.kind == PXL_OP_HEAD) ok = dc.in1 == dj.out && dc.in0 != dj.out && dc.bn_in1 < 0;
        else ok = false;
      }
      if (!ok || first < 0) continue;
      OpInfo& oc = n->ops[first];
      if (oc.d.kind != PXL_OP_CONV || oc.d.bn_in0 >= 0 || (oc.d.ngroups != 1 && !oc.pg) || oc.join_op >= 0) continue;
      const pxl_conv_desc& obw = oc.pg ? oc.pg_bwd : oc.bwd;       // (the multi-rate head: its data gradient is the GEMM J -> Cin)
      if (!pxl_conv_dma_eligible(&obw, nullptr, nullptr) || obw.Kreal != obw.Cout || obw.Cout != o.C) continue;
      oc.join_op = (int)j;
      n->ops[j].join_conv = first;
      // (bf16: that launch takes the join's ReLU mask as a bit plane, PXL_JOIN_BITS=0: the join output itself as in rounds 1-5)
      static const bool bits_on = getenv("PXL_JOIN_BITS") == nullptr || getenv("PXL_JOIN_BITS")[0] != '0';
      n->ops[j].bits = bits_on && n->dtype == PXL_BF16 && o.Cp % 8 == 0;
      if (n->ops[j].bits) { n->ops[j].bits_off = arena; arena += align_up((size_t)n->B * o.H * o.W * (o.Cp / 8)); }
    }
  }
  // data gradients that produce BatchNorm-backward sums in their epilogue: as many replicas as the BatchNorm's sums have
  for (size_t i = 0; i < n->ops.size(); ++i) {
    OpInfo& op = n->ops[i];
    if (op.d.kind != PXL_OP_CONV) continue;
    op.bwd.stats_rep = 1;
    if (op.pg) op.pg_bwd.stats_rep = op.join_op >= 0 ? n->bns[n->ops[op.join_op].d.bn_in0].bnrep : 1;
    if (op.join_op >= 0) op.bwd.stats_rep = n->bns[n->ops[op.join_op].d.bn_in0].bnrep;
    else if (op.d.bn_in0 >= 0 && n->bns[op.d.bn_in0].fused_reduce_op == (int)i) op.bwd.stats_rep = n->bns[op.d.bn_in0].bnrep;
  }
  // lowest gradient offset written by each op's backward, and whether those grow with the op index
  n->op_lo.assign(n->ops.size(), -1);
  n->bucket_ok = true;
  {
    long prev = -1;
    for (size_t i = 0; i < n->ops.size(); ++i) {
      const pxl_op& d = n->ops[i].d;
      long lo = -1;
      auto take = [&](long off) { if (off >= 0 && (lo < 0 || off < lo)) lo = off; };
      if (d.kind == PXL_OP_CONV) {
        for (int g = 0; g < d.ngroups; ++g) { take(d.w_off[g]); take(d.b_off[g]); }
        if (d.bn_out >= 0) { take(n->bns[d.bn_out].d.gamma_off); take(n->bns[d.bn_out].d.beta_off); }
      } else if (d.kind == PXL_OP_IBN && d.bn_out >= 0) {
        take(n->bns[d.bn_out].d.gamma_off); take(n->bns[d.bn_out].d.beta_off);
      }
      n->op_lo[i] = lo;
      if (lo >= 0) { if (lo <= prev) n->bucket_ok = false; prev = lo; }
    }
  }
  // per-sample sums of the IBNorm layers: contiguous, zeroed by ONE memset per pass (GCT's flaw detector has seven such
  // layers and runs four passes per step: 56 memset launches on the critical stream became 8)
  n->ibn_region_off = arena;
  n->ibn_bregion_off = scratch;
  for (auto& op : n->ops) {
    if (op.d.kind != PXL_OP_IBN) continue;
    const TensorInfo& tin = n->tensors[op.d.in0];
    const size_t per = align_up((size_t)B * 2 * tin.Cp * 4);
    op.ibn_sums = arena; arena += per;
    op.ibn_bsums = scratch; scratch += per;
  }
  n->ibn_region_bytes = arena - n->ibn_region_off;
  n->ibn_bregion_bytes = scratch - n->ibn_bregion_off;
  n->arena_bytes = arena;
  n->scratch_bytes = scratch;
  n->packed_bytes = packed;
  n->planned = true;
  return PXL_OK;
}

extern "C" size_t pxl_net_packed_bytes(const pxl_net* n) { return n && n->planned ? n->packed_bytes : 0; }
extern "C" size_t pxl_net_arena_bytes(const pxl_net* n) { return n && n->planned ? n->arena_bytes : 0; }
extern "C" size_t pxl_net_scratch_bytes(const pxl_net* n) { return n && n->planned ? n->scratch_bytes : 0; }

// which: bit 0 = forward operand layout (+ summed biases), bit 1 = transposed data-gradient layout.  The two halves are
// independent: the host packs the forward half on the stream of the forward pass and the data-gradient half, which is
// first read by the backward pass, on a side stream that overlaps the forward.
namespace {
int net_pack_impl(pxl_net* n, const float* params, void* packed, int which, long lo, long hi, void* stream);
// the forward kernel layout of this convolution's weights is its master layout cast to bf16 (what pxl_sgd_ema_pack writes)
inline bool fwd_is_cast(const pxl_net* n, const OpInfo& op) {
  const pxl_op& d = op.d;
  if (d.kind != PXL_OP_CONV || n->dtype != PXL_BF16 || op.patch || d.ngroups != 1) return false;
  const TensorInfo& tin = n->tensors[d.in0];
  const long total = (long)d.cout * d.kh * d.kw * d.cin;
  return tin.Cp == d.cin && (total & 7) == 0 && (d.w_off[0] & 3) == 0 && (op.wf_off & 15) == 0;
}
}

extern "C" int pxl_net_update_segments(pxl_net* n, pxl_net* t, pxl_upd_seg* out, int cap) {
  PXL_REQUIRE(n && n->planned && out && cap > 0, "net_update_segments: bad argument (plan first)");
  PXL_REQUIRE(t == nullptr || (t->planned && t->ops.size() == n->ops.size() && t->dtype == n->dtype),
              "net_update_segments: the second network must run the same program");
  int k = 0;
  for (size_t i = 0; i < n->ops.size(); ++i) {
    const OpInfo& op = n->ops[i];
    if (!fwd_is_cast(n, op)) continue;
    if (t != nullptr && !(fwd_is_cast(t, t->ops[i]) && t->ops[i].d.w_off[0] == op.d.w_off[0])) continue;
    PXL_REQUIRE(k < cap, "net_update_segments: more than %d segments", cap);
    out[k].off = op.d.w_off[0];
    out[k].n = (int64_t)op.d.cout * op.d.kh * op.d.kw * op.d.cin;
    out[k].s_pk = (int64_t)op.wf_off;
    out[k].t_pk = t != nullptr ? (int64_t)t->ops[i].wf_off : -1;
    ++k;
  }
  for (int a = 1; a < k; ++a)                      // sorted by offset (the parameter order of every built-in program already is)
    for (int b = a; b > 0 && out[b].off < out[b - 1].off; --b) { pxl_upd_seg tmp = out[b]; out[b] = out[b - 1]; out[b - 1] = tmp; }
  return k;
}

extern "C" int pxl_net_pack_parts(pxl_net* n, const float* params, void* packed, int which, void* stream) {
  return net_pack_impl(n, params, packed, which, 0, -1, stream);
}

// The same for the convolutions whose master weights lie in params[lo, hi) only (a bucket of the pipelined parameter update,
// pxl_net_set_update_hook: bucket boundaries fall between ops, so a convolution is inside or outside as a whole)
extern "C" int pxl_net_pack_range(pxl_net* n, const float* params, void* packed, int which, long lo, long hi, void* stream) {
  PXL_REQUIRE(lo >= 0 && hi >= lo, "net_pack_range: bad range");
  return net_pack_impl(n, params, packed, which, lo, hi, stream);
}

namespace {
int net_pack_impl(pxl_net* n, const float* params, void* packed, int which, long lo, long hi, void* stream) {
  PXL_REQUIRE(n && n->planned && params && packed && (which & 7) != 0, "net_pack: bad argument (plan first)");
  std::vector<pxl_pack_item> items;
  for (auto& op : n->ops) {
    const pxl_op& d = op.d;
    if (d.kind != PXL_OP_CONV) continue;
    if (hi >= 0 && !(d.w_off[0] >= lo && d.w_off[0] < hi)) continue;
    const TensorInfo& tin = n->tensors[d.in0];
    const TensorInfo& tout = n->tensors[d.out];
    const int tpg = d.kh * d.kw;
    const bool want_t = (which & 2) && d.need_dgrad && n->pack_dgrad;
    // which & 4: forward layouts only of the convolutions the fused update kernel does not write (pxl_net_update_segments)
    const bool want_f = (which & 1) || ((which & 4) && !fwd_is_cast(n, op));
    if (op.pg && (want_f || want_t)) {
      // Wp [J][Cin] in the GEMM's column order (group, tap, class) and its transpose, the data-gradient operand (aspp.hip)
      long woffs[4] = {0, 0, 0, 0};
      for (int g = 0; g < d.ngroups; ++g) woffs[g] = d.w_off[g];
      const int rc = pxl_aspp_pack(n->dtype, params, woffs, d.ngroups, op.pg_GP, d.cout, tpg, d.cin, tin.Cp,
                                   want_f ? at(packed, op.pg_wf_off) : nullptr, want_t ? at(packed, op.pg_wt_off) : nullptr, stream);
      if (rc != PXL_OK) return rc;
    }
    for (int g = 0; g < d.ngroups; ++g) {
      if ((!want_f && !want_t) || op.pg) continue;          // (a head that runs as a GEMM never reads the 36-tap layouts)
      pxl_pack_item it;
      it.src_off = d.w_off[g];
      it.wf_off = want_f ? (int64_t)op.wf_off : -1;
      it.wt_off = want_t ? (int64_t)op.wt_off : -1;
      it.K = d.cout; it.T = tpg; it.C = d.cin;
      it.Cp = tin.Cp; it.T_total = op.ntaps; it.t_off = g * tpg; it.Kp = tout.Cp;
      if (op.patch) { it.T = 1; it.C = op.patch_K; it.Cp = op.patch_Kp; it.T_total = 1; it.t_off = 0; }   // master [Cout][kh*kw*C] as is
      items.push_back(it);
    }
    if ((which & 5) && d.b_off[0] >= 0) {
      const float* b[4] = {nullptr, nullptr, nullptr, nullptr};
      for (int g = 0; g < d.ngroups; ++g) b[g] = d.b_off[g] >= 0 ? params + d.b_off[g] : nullptr;
      int rc = pxl_vec_sum4(d.cout, fat(packed, op.bias_off), b[0], b[1], b[2], b[3], stream);
      if (rc != PXL_OK) return rc;
    }
  }
  if (items.empty()) return PXL_OK;
  return pxl_pack_weights_batched(n->dtype, params, packed, items.data(), (int)items.size(), stream);
}
}  // namespace

extern "C" int pxl_net_pack(pxl_net* n, const float* params, void* packed, void* stream) {
  return pxl_net_pack_parts(n, params, packed, 3, stream);
}

namespace {
// median-free quick timer: one warm-up + `reps` timed launches, returns the best time in ms (< 0 on error)
template <typename F>
float time_launch(F&& fn, hipStream_t s, hipEvent_t a, hipEvent_t b, int reps) {
  if (fn() != PXL_OK) return -1.f;
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    if (hipEventRecord(a, s) != hipSuccess) return -1.f;
    if (fn() != PXL_OK) return -1.f;
    if (hipEventRecord(b, s) != hipSuccess) return -1.f;
    if (hipEventSynchronize(b) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.f;
    if (ms < best) best = ms;
  }
  return best;
}
// the same with TWO copies of the launch in flight, one per stream (what the MT forward does with student || teacher: a tile that
// is best alone is not always best next to a copy of itself -- tools/cbench --dual): returns the time until both are done
template <typename F>
float time_launch_dual(F&& fn, hipStream_t s, hipStream_t s2, hipEvent_t a, hipEvent_t b, hipEvent_t b2, int reps) {
  if (fn(s) != PXL_OK || fn(s2) != PXL_OK) return -1.f;
  if (hipStreamSynchronize(s2) != hipSuccess) return -1.f;
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    if (hipEventRecord(a, s) != hipSuccess || hipStreamWaitEvent(s2, a, 0) != hipSuccess) return -1.f;
    if (fn(s) != PXL_OK || fn(s2) != PXL_OK) return -1.f;
    if (hipEventRecord(b, s) != hipSuccess || hipEventRecord(b2, s2) != hipSuccess) return -1.f;
    if (hipEventSynchronize(b) != hipSuccess || hipEventSynchronize(b2) != hipSuccess) return -1.f;
    float m1 = 0.f, m2 = 0.f;
    if (hipEventElapsedTime(&m1, a, b) != hipSuccess || hipEventElapsedTime(&m2, a, b2) != hipSuccess) return -1.f;
    const float ms = m1 > m2 ? m1 : m2;
    if (ms < best) best = ms;
  }
  return best;
}
// the candidate's own duration on stream s while `load` (three launches) keeps stream s2 busy: data-gradient tiles are
// chosen beside the weight gradient that runs next to them in the backward pass
template <typename F, typename L>
float time_launch_loaded(F&& fn, L&& load, hipStream_t s, hipStream_t s2, hipEvent_t a, hipEvent_t b, hipEvent_t b2, int reps) {
  if (fn() != PXL_OK || load(s2) != PXL_OK) return -1.f;
  if (hipStreamSynchronize(s2) != hipSuccess) return -1.f;
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    if (hipEventRecord(b2, s) != hipSuccess || hipStreamWaitEvent(s2, b2, 0) != hipSuccess) return -1.f;
    for (int k = 0; k < 3; ++k) if (load(s2) != PXL_OK) return -1.f;
    if (hipEventRecord(a, s) != hipSuccess) return -1.f;
    if (fn() != PXL_OK) return -1.f;
    if (hipEventRecord(b, s) != hipSuccess) return -1.f;
    if (hipEventSynchronize(b) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.f;
    if (ms < best) best = ms;
  }
  return best;
}
}  // namespace

namespace {
pxl_bn_fin make_fin(const pxl_net* n, const BnInfo& b, const float* params, float* running, void* arena, int training) {
  pxl_bn_fin f;
  f.stats = fat(arena, b.stats_off);
  f.nrep = b.fin_nrep;
  f.count = (float)b.M * n->world;
  f.gamma = params + b.d.gamma_off;
  f.beta = params + b.d.beta_off;
  f.running_mean = running ? running + b.d.rmean_off : nullptr;
  f.running_var = running ? running + b.d.rvar_off : nullptr;
  f.momentum = n->eff_momentum(b.d.momentum);
  f.eps = b.d.eps;
  f.training = training;
  f.clamp_var = (n->world > 1 || n->force_clamp) ? 1 : 0;
  f.coef = fat(arena, b.coef_off);
  return f;
}
}  // namespace

// "Measure, don't guess": time every tile configuration of every contraction on the planned shapes and
// keep the fastest.  Results do not depend on the choice (same per-element reduction order for
// forward/dgrad; wgrad differs only in fp32 atomic order).  Clobbers arena/scratch/grads contents.
extern "C" int pxl_net_tune(pxl_net* n, const float* params, const void* packed, float* grads, void* arena,
                            size_t arena_bytes, void* scratch, size_t scratch_bytes, void* stream) {
  PXL_REQUIRE(n && n->planned && params && packed && grads && arena && scratch, "net_tune: bad argument (plan first)");
  if (arena_bytes < n->arena_bytes || scratch_bytes < n->scratch_bytes)
    return pxl_set_error(PXL_ERR_WORKSPACE, "net_tune: arena/scratch too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t a, b;
  PXL_CHECK_HIP(hipEventCreate(&a));
  PXL_CHECK_HIP(hipEventCreate(&b));
  const int reps = 3;
  int rc_all = PXL_OK;
  // forward tiles timed with two copies of the launch in flight on two streams, for networks whose forward runs next to a copy of
  // itself (the MT student || teacher on two streams: pxl_net_set_tune_dual; PXL_TUNE_DUAL=0 / 1 overrides).  Measured (DESIGN.md 4,
  // round 4): MT 12.35 -> 12.25 ms; SupOnly (nothing runs beside its forward) 7.27 -> 7.49 ms, which is why it is per network
  hipStream_t dual_s = nullptr;
  hipEvent_t b2 = nullptr;
  const char* td_env = getenv("PXL_TUNE_DUAL");
  // PXL_TUNE_BWD_LOAD=1 (experiment): data-gradient tiles timed beside the convolution's weight gradient on a second stream
  const bool bwd_load = getenv("PXL_TUNE_BWD_LOAD") != nullptr && getenv("PXL_TUNE_BWD_LOAD")[0] == '1';
  if ((td_env != nullptr ? td_env[0] == '1' : n->tune_dual == 1) || bwd_load) {
    // (the placement pool's SIDE stream: probed to sit on another hardware queue than the caller's, csrc/streams.hip)
    dual_s = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_SIDE));
    if (dual_s == s) dual_s = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_WGRAD));
    PXL_CHECK_HIP(hipEventCreate(&b2));
  }
  // PXL_TUNE_CFGS="11,17,18": restrict the LDS-DMA tile candidates (experiments: the tuner times every launch ALONE, while
  // in the step two or three streams share the CUs' LDS -- smaller footprints co-reside better); unset = all of them
  std::vector<int> allow;
  if (const char* e = getenv("PXL_TUNE_CFGS")) {
    for (const char* q = e; *q;) { allow.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
  }
  auto allowed = [&](int cfg) {
    if (allow.empty() || cfg < 8) return true;
    for (int a : allow) if (a == cfg) return true;
    return false;
  };
  for (auto& op : n->ops) {
    const pxl_op& d = op.d;
    if (d.kind != PXL_OP_CONV) continue;
    const TensorInfo& tin = n->tensors[d.in0];
    const TensorInfo& tout = n->tensors[d.out];
    const ConvIn cin = conv_input(n, op, arena);
    const float* sc = cin.sc; const float* sh = cin.sh;
    float* stats = d.bn_out >= 0 ? fat(arena, n->bns[d.bn_out].stats_off) : nullptr;
    const float* bias = d.b_off[0] >= 0 ? fat(packed, op.bias_off) : nullptr;
    if (op.pg) {
      // the multi-rate head as GEMMs (aspp.hip): its three launches, each over the tile configurations that can run it
      const bool f32 = n->dtype == PXL_F32;
      {
        int best_cfg = -1; float best = 1e30f;
        for (int cfg = 8; cfg < 36; ++cfg) {
          if ((cfg >= 12 && cfg < 16) || (f32 && cfg >= 20) || !allowed(cfg)) continue;
          pxl_conv_desc q = op.pg_fwd; q.tile_cfg = cfg;
          int rc1 = PXL_OK;
          const float t = time_launch([&]() {
            rc1 = f32 ? pxl_conv_igemm(&q, cin.ptr, at(packed, op.pg_wf_off), fat(arena, op.ws_off), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream)
                      : pxl_conv_dma_slabs(&q, cin.ptr, at(packed, op.pg_wf_off), fat(arena, op.ws_off), op.ws_bytes, 1, stream);
            return rc1; }, s, a, b, reps);
          if (t < 0) { if (rc1 != PXL_ERR_UNSUPPORTED) rc_all = PXL_ERR_HIP; continue; }      // (a tile without fp32 staging)
          if (t < best) { best = t; best_cfg = cfg; }
        }
        op.pg_fwd.tile_cfg = best_cfg;
      }
      if (n->pack_dgrad) {
        int best_cfg = -1; float best = 1e30f;
        const bool wdma = pxl_conv_wgrad_dma_eligible(&op.pg_grp, nullptr) != 0;
        for (int cfg = 0; cfg < (wdma ? 14 : 3); ++cfg) {
          if (cfg >= 3 && cfg < 8) continue;
          pxl_conv_desc q = op.pg_grp; q.tile_cfg = cfg;
          const float t = time_launch([&]() { return pxl_conv_wgrad(&q, cin.ptr, nullptr, nullptr, at(scratch, op.pg_dp_off),
                                                                    fat(scratch, op.pg_dw_off), d.cin, d.cin, stream); }, s, a, b, reps);
          if (t < 0) { rc_all = PXL_ERR_HIP; continue; }
          if (t < best) { best = t; best_cfg = cfg; }
        }
        op.pg_grp.tile_cfg = best_cfg;
      }
      if (d.need_dgrad && n->pack_dgrad) {
        int best_cfg = -1; float best = 1e30f;
        for (int cfg = 8; cfg < 36; ++cfg) {
          if ((cfg >= 12 && cfg < 16) || (f32 && cfg >= 20) || !allowed(cfg)) continue;
          pxl_conv_desc q = op.pg_bwd; q.tile_cfg = cfg;
          const float t = time_launch([&]() { return pxl_conv_igemm(&q, at(scratch, op.pg_dp_off), at(packed, op.pg_wt_off), at(scratch, tin.goff),
                                                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream); }, s, a, b, reps);
          if (t < 0) { rc_all = PXL_ERR_HIP; continue; }
          if (t < best) { best = t; best_cfg = cfg; }
        }
        op.pg_bwd.tile_cfg = best_cfg;
      }
      continue;
    }
    // forward
    {
      int best_cfg = -1; float best = 1e30f;
      const bool dma = pxl_conv_dma_eligible(&op.fwd, sc, nullptr) != 0;
      const bool onload = dma && d.bn_in0 >= 0 && n->bns[d.bn_in0].onload && getenv("PXL_TUNE_ONLOAD_PLAIN") == nullptr;
      for (int cfg = dma ? 8 : 0; cfg < (dma ? 36 : 8); ++cfg) {
        if (!dma && (cfg & 3) == 3 && tout.Cp > 64) continue;
        if (dma && cfg >= 12 && cfg < 16) continue;          // 4-stage rings never won on the ResNet shapes
        if (dma && n->dtype == PXL_F32 && cfg >= 20) continue;      // fp32 kernel: the 2x2-wave tiles only (conv_dma_f32.hip)
        if (dma && cfg >= 20 && cfg != 29 && tout.Cp < 128) continue;     // tall / 8-wave tiles are 128 channels wide (29: 128 x 64)
        if (dma && cfg == 35 && tout.Cp < 256) continue;
        if (dma && !allowed(cfg)) continue;
        pxl_conv_desc q = op.fwd; q.tile_cfg = cfg;
        float t;
        if (onload) {
          // the launch the forward pass will make: BatchNorm finalize + apply on load (its K loop carries the transform, so the
          // best tile is not the plain kernel's); statistics of the zeroed arena, running statistics left alone
          BnInfo& bi = n->bns[d.bn_in0];
          bi.fin_nrep = bi.nrep;
          const pxl_bn_fin bin = make_fin(n, bi, params, nullptr, arena, 1);
          int rc1 = PXL_OK;
          t = time_launch([&]() { rc1 = pxl_conv_dma_bnin(&q, at(arena, tin.off), at(packed, op.wf_off), at(arena, tout.off), bias,
                                                          stats, &bin, bi.relu, n->pack_dgrad ? at(arena, bi.z_off) : nullptr, stream);
                                  return rc1; }, s, a, b, reps);
          if (t < 0 && rc1 == PXL_ERR_UNSUPPORTED) continue;        // this tile + the coefficient table do not fit
        } else if (dual_s != nullptr && dma && !op.ws_bytes) {
          t = time_launch_dual([&](hipStream_t st) { return pxl_conv_igemm(&q, cin.ptr, at(packed, op.wf_off), at(arena, tout.off), sc, sh, bias,
                                                                           nullptr, stats, nullptr, 0, st); }, s, dual_s, a, b, b2, reps);
        } else {
          t = time_launch([&]() { return pxl_conv_igemm(&q, cin.ptr, at(packed, op.wf_off), at(arena, tout.off),
                                                        sc, sh, bias, nullptr, stats, op.ws_bytes ? at(arena, op.ws_off) : nullptr,
                                                        op.ws_bytes, stream); }, s, a, b, reps);
        }
        if (t < 0) { rc_all = PXL_ERR_HIP; continue; }
        if (t < best) { best = t; best_cfg = cfg; }
      }
      op.fwd.tile_cfg = best_cfg;
    }
    // weight gradient (per tap group)
    for (int g = 0; g < d.ngroups && n->pack_dgrad; ++g) {
      int best_cfg = -1; float best = 1e30f;
      const bool wdma = pxl_conv_wgrad_dma_eligible(&op.grp[g], sc) != 0;
      for (int cfg = 0; cfg < (wdma ? 14 : 3); ++cfg) {
        if (cfg == 2 && d.cout > 64) continue;
        if (cfg >= 3 && cfg < 8) continue;
        pxl_conv_desc q = op.grp[g]; q.tile_cfg = cfg;
        const int creal = op.patch ? op.patch_K : d.cin;
        float t = time_launch([&]() { return pxl_conv_wgrad(&q, cin.ptr, sc, sh, at(scratch, tout.goff),
                                                            grads + d.w_off[g], creal, creal, stream); }, s, a, b, reps);
        if (t < 0) { rc_all = PXL_ERR_HIP; continue; }
        if (t < best) { best = t; best_cfg = cfg; }
      }
      op.grp[g].tile_cfg = best_cfg;
    }
    // data gradient
    if (d.need_dgrad && n->pack_dgrad) {
      int best_cfg = -1; float best = 1e30f;
      const bool dma = pxl_conv_dma_eligible(&op.bwd, nullptr, nullptr) != 0;
      for (int cfg = dma ? 8 : 0; cfg < (dma ? 36 : 8); ++cfg) {
        if (!dma && (cfg & 3) == 3 && tin.Cp > 64) continue;
        if (dma && cfg >= 12 && cfg < 16) continue;
        if (dma && n->dtype == PXL_F32 && cfg >= 20) continue;
        if (dma && cfg >= 20 && cfg != 29 && tin.Cp < 128) continue;
        if (dma && cfg == 35 && tin.Cp < 256) continue;
        if (dma && !allowed(cfg)) continue;
        pxl_conv_desc q = op.bwd; q.tile_cfg = cfg;
        auto cand = [&]() { return pxl_conv_igemm(&q, at(scratch, tout.goff), at(packed, op.wt_off), at(scratch, tin.goff),
                                                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream); };
        float t;
        if (bwd_load && dual_s != nullptr && dma && n->wgrad_on) {
          // beside this convolution's (already tuned) weight gradient, as in the backward pass
          const int creal = op.patch ? op.patch_K : d.cin;
          t = time_launch_loaded(cand, [&](hipStream_t st) { return pxl_conv_wgrad(&op.grp[0], cin.ptr, sc, sh, at(scratch, tout.goff),
                                                                                 grads + d.w_off[0], creal, creal, st); },
                                 s, dual_s, a, b, b2, reps);
        } else {
          t = time_launch(cand, s, a, b, reps);
        }
        if (t < 0) { rc_all = PXL_ERR_HIP; continue; }
        if (t < best) { best = t; best_cfg = cfg; }
      }
      op.bwd.tile_cfg = best_cfg;
    }
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (b2 != nullptr) (void)hipEventDestroy(b2);
  if (rc_all != PXL_OK) return pxl_set_error(rc_all, "net_tune: a candidate launch failed: %s", pxl_last_error());
  return PXL_OK;
}


namespace {
// everything one forward pass hands to its ops
struct FwdCtx {
  const float* params; const void* packed; float* running; const float* x; float* logits; float* prob; void* arena;
  int training; void* stream;
};

// Forward of op i.  phase 0: the whole op.  CONV ops can run in two halves -- 1: everything up to and including the
// convolution launch, 2: what follows it (Sync-BN exchange, finalize, activation) -- so that the paired pass
// (pxl_net_forward_pair) can issue the convolution of two networks as ONE launch between the halves; *fin_flag carries
// "the convolution finalized its BatchNorm itself" from half 1 to half 2.
// sync_done (phase 2 only): the Sync-BN exchange of this op's BatchNorm has already been performed by the caller (the paired
// pass exchanges the statistics of both networks in one launch): replica 0 holds the all-reduced sums.
// Sync-BN statistics of one network: fold the replicas, all-reduce [2C] over the ranks.  With the peer-mapped exchange both
// happen in ONE launch (pxl_peer_allreduce_fold), otherwise fold + the hook (RCCL / torch.distributed).
int sync_stats(pxl_net* n, float* stats, int count, int nrep, void* stream) {
  if (n->sync == &pxl_peer_allreduce_hook)
    return pxl_peer_allreduce_fold(reinterpret_cast<pxl_peer*>(n->sync_user), stats, nullptr, count, nrep, stream);
  int rc = pxl_bn_fold_replicas(count, nrep, stats, stream);
  if (rc != PXL_OK) return rc;
  rc = n->sync(n->sync_user, stats, count, stream);
  if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_forward: SyncBN all-reduce hook failed (%d)", rc);
  return PXL_OK;
}

int forward_op(pxl_net* n, size_t i, const FwdCtx& c, int phase, bool* fin_flag, bool sync_done = false) {
  const float* params = c.params; const void* packed = c.packed; float* running = c.running; const float* x = c.x;
  float* logits = c.logits; float* prob = c.prob; void* arena = c.arena; const int training = c.training; void* stream = c.stream;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int dt = n->dtype;
  OpInfo& op = n->ops[i];
  const pxl_op& d = op.d;
  int rc = PXL_OK;
  if (phase != 0 && d.kind != PXL_OP_CONV) return pxl_set_error(PXL_ERR_ARG, "net_forward: only convolutions run in halves");
    switch (d.kind) {
      case PXL_OP_INPUT: {
        const TensorInfo& t = n->tensors[d.out];
        if (!n->input_needed && n->in_parts == 0) break;      // only patch-mode convolutions read the input: from `x` directly
        if (n->in_parts > 0) {
          int C = 0;
          for (int k = 0; k < n->in_parts; ++k) C += n->in_chans[k];
          if (C != t.C) return pxl_set_error(PXL_ERR_ARG, "net_forward: input parts hold %d channels, the program expects %d", C, t.C);
          rc = pxl_nchw_parts_to_nhwc(dt, n->in_parts, n->in_src, n->in_chans, at(arena, t.off), n->B, t.H, t.W, t.Cp, stream);
          n->in_parts = 0;
        } else {
          rc = pxl_nchw_to_nhwc(dt, x, at(arena, t.off), n->B, t.C, t.H, t.W, t.Cp, stream);
        }
        break;
      }
      case PXL_OP_CONV: {
        if (phase != 2) {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        const ConvIn cin = conv_input(n, op, arena);
        const float* sc = cin.sc; const float* sh = cin.sh;
        float* stats = (d.bn_out >= 0 && training) ? fat(arena, n->bns[d.bn_out].stats_off) : nullptr;
        const float* bias = d.b_off[0] >= 0 ? fat(packed, op.bias_off) : nullptr;
        bool fin_by_conv = false;
        if (op.patch) {
          if (x == nullptr) return pxl_set_error(PXL_ERR_ARG, "net_forward: the patch-mode stem reads ONE NCHW input tensor");
          if (n->patch_src != nullptr) {
            // borrowed: the lender's launch (same input, same geometry) is ordered before this pass by the caller
          } else if (n->patches_made) {
            n->patches_made = false;        // written ahead of this pass by pxl_net_make_patches
          } else {
            rc = pxl_stem_patches(dt, x, at(arena, op.patch_off), n->B, d.cin, tin.H, tin.W, d.kh, d.kw, d.stride, d.pads[0],
                                  tout.H, tout.W, op.patch_Kp, stream);
            if (rc != PXL_OK) return rc;
          }
        }
        // BN-apply on load: this convolution reads the raw output of its producer and finalizes + applies the BatchNorm
        // itself; the activated tensor is written only where a weight gradient will read it, on the side stream
        bool onload_done = false;
        if (d.bn_in0 >= 0 && n->bns[d.bn_in0].onload && !n->bns[d.bn_in0].fin_by_conv) {
          BnInfo& bi = n->bns[d.bn_in0];
          const pxl_bn_fin bin = make_fin(n, bi, params, running, arena, training);
          {
            Timed t(n, s, 0, conv_flops(n, d, tout));
            if (n->profile) n->prof_bytes[0] += conv_bytes(n, d, tin, tout, false);
            // networks that can run a backward pass (pack_dgrad) get z = relu(bn(y)) written on the way, for this op's
            // weight gradient -- whatever wgrad_on says now: it may be switched on between this pass and its backward
            rc = pxl_conv_dma_bnin(&op.fwd, at(arena, tin.off), at(packed, op.wf_off), at(arena, tout.off), bias, stats, &bin,
                                   bi.relu, n->pack_dgrad ? at(arena, bi.z_off) : nullptr, stream);
          }
          if (rc == PXL_OK) {
            onload_done = true;
          } else if (rc == PXL_ERR_UNSUPPORTED) {     // (tile + coefficient table too large ...): materialise, then the plain path
            rc = pxl_bn_finalize_apply_fwd(dt, (long)n->B * tin.H * tin.W, tin.Cp, at(arena, tin.off), &bin, bi.relu,
                                           at(arena, bi.z_off), stream);
            if (rc != PXL_OK) return rc;
          } else {
            return rc;
          }
        }
        if (!onload_done && op.pg) {
          // the multi-rate head as ONE GEMM + col2im (csrc/aspp.hip)
          Timed t(n, s, 0, conv_flops(n, d, tout));
          if (n->profile) n->prof_bytes[0] += conv_bytes(n, d, tin, tout, false);
          float* P = fat(arena, op.ws_off);
          if (n->dtype == PXL_BF16)
            rc = pxl_conv_dma_slabs(&op.pg_fwd, cin.ptr, at(packed, op.pg_wf_off), P, op.ws_bytes, 1, stream);
          else
            rc = pxl_conv_igemm(&op.pg_fwd, cin.ptr, at(packed, op.pg_wf_off), P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
          if (rc != PXL_OK) return rc;
          rc = pxl_aspp_col2im(dt, n->B, tout.H, tout.W, op.pg_J, op.pg_GP, d.ngroups, d.cout, d.kh * d.kw, op.fwd.dy, op.fwd.dx, P, 1,
                               (size_t)n->B * tout.H * tout.W * op.pg_J, bias, at(arena, tout.off), tout.Cp, stream);
          onload_done = true;
        }
        if (!onload_done) {
          Timed t(n, s, 0, conv_flops(n, d, tout));
          if (n->profile) n->prof_bytes[0] += conv_bytes(n, d, tin, tout, false);
          if (stats && n->conv_finalize && !(n->sync && n->world > 1) && n->dtype == PXL_BF16 && !op.ws_bytes &&
              pxl_conv_dma_eligible(&op.fwd, sc, nullptr) && (op.fwd.tile_cfg < 0 || op.fwd.tile_cfg >= 8)) {
            BnInfo& b = n->bns[d.bn_out];
            pxl_bn_fin fin = make_fin(n, b, params, running, arena, training);
            rc = pxl_conv_dma_finalize(&op.fwd, cin.ptr, at(packed, op.wf_off), at(arena, tout.off), bias, stats, &fin,
                                       reinterpret_cast<unsigned*>(at(arena, b.cnt_off)), stream);
            fin_by_conv = rc == PXL_OK;
          }
          if (!fin_by_conv)
            rc = pxl_conv_igemm(&op.fwd, cin.ptr, at(packed, op.wf_off), at(arena, tout.off), sc, sh, bias,
                                nullptr, stats, op.ws_bytes ? at(arena, op.ws_off) : nullptr, op.ws_bytes, stream);
        }
        if (rc != PXL_OK) return rc;
        *fin_flag = fin_by_conv;
        }   // phase != 2
        if (phase == 1) break;
        {
        const bool fin_by_conv = *fin_flag;
        const TensorInfo& tout = n->tensors[d.out];
        if (d.bn_out >= 0) n->bns[d.bn_out].fin_by_conv = fin_by_conv;
        if (d.bn_out >= 0 && fin_by_conv) {
          BnInfo& b = n->bns[d.bn_out];
          if (b.has_z)
            rc = pxl_bn_apply_fwd(dt, (long)n->B * tout.H * tout.W, tout.Cp, at(arena, tout.off), fat(arena, b.coef_off),
                                  b.relu, at(arena, b.z_off), stream);
        } else if (d.bn_out >= 0) {
          BnInfo& b = n->bns[d.bn_out];
          int nrep = b.nrep;
          if (training && n->sync && n->world > 1) {
            if (!sync_done) {
              rc = sync_stats(n, fat(arena, b.stats_off), 2 * b.d.C, b.nrep, stream);
              if (rc != PXL_OK) return rc;
            }
            nrep = 1;
          }
          b.fin_nrep = nrep;
          if (b.onload) {
            // finalized and applied by the convolution that consumes it (BN-apply on load)
          } else if (b.fin_in_consumer) {
            if (b.has_z) {
              const pxl_bn_fin fin = make_fin(n, b, params, running, arena, training);
              rc = pxl_bn_finalize_apply_fwd(dt, (long)n->B * tout.H * tout.W, tout.Cp, at(arena, tout.off), &fin, b.relu,
                                             at(arena, b.z_off), stream);
            }                                   // else: the residual join that consumes y finalizes it
          } else {
            rc = pxl_bn_finalize(b.d.C, fat(arena, b.stats_off), nrep, (float)b.M * n->world, params + b.d.gamma_off,
                                 params + b.d.beta_off, running ? running + b.d.rmean_off : nullptr,
                                 running ? running + b.d.rvar_off : nullptr, n->eff_momentum(b.d.momentum), b.d.eps, training,
                                 (n->world > 1 || n->force_clamp) ? 1 : 0, fat(arena, b.coef_off), stream);
            if (rc != PXL_OK) return rc;
            if (b.has_z)
              rc = pxl_bn_apply_fwd(dt, (long)n->B * tout.H * tout.W, tout.Cp, at(arena, tout.off), fat(arena, b.coef_off),
                                    b.relu, at(arena, b.z_off), stream);
          }
        }
        }   // post
        break;
      }
      case PXL_OP_MAXPOOL: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        const float* coef = d.bn_in0 >= 0 ? fat(arena, n->bns[d.bn_in0].coef_off) : nullptr;
        rc = pxl_maxpool3x3s2_fwd(dt, n->B, tin.H, tin.W, tin.Cp, at(arena, tin.off), coef, at(arena, tout.off),
                                  at(arena, op.idx_off), stream);
        break;
      }
      case PXL_OP_RESIDUAL: {
        const TensorInfo& a = n->tensors[d.in0];
        const TensorInfo& r = n->tensors[d.in1];
        const TensorInfo& o = n->tensors[d.out];
        const float* ac = fat(arena, n->bns[d.bn_in0].coef_off);
        const float* rcoef = d.bn_in1 >= 0 ? fat(arena, n->bns[d.bn_in1].coef_off) : nullptr;
        // (a residual join finalizes either both of its BatchNorms or none: both convolutions took the same path)
        if (n->bns[d.bn_in0].fin_in_consumer && !n->bns[d.bn_in0].has_z && !n->bns[d.bn_in0].fin_by_conv &&
            !(d.bn_in1 >= 0 && n->bns[d.bn_in1].fin_by_conv)) {
          const pxl_bn_fin yfin = make_fin(n, n->bns[d.bn_in0], params, running, arena, training);
          pxl_bn_fin rfin;
          if (d.bn_in1 >= 0) rfin = make_fin(n, n->bns[d.bn_in1], params, running, arena, training);
          rc = pxl_residual_finalize_fwd_bits(dt, (long)n->B * a.H * a.W, a.Cp, at(arena, a.off), &yfin, at(arena, r.off),
                                              d.bn_in1 >= 0 ? &rfin : nullptr, at(arena, o.off),
                                              op.bits && n->pack_dgrad ? at(arena, op.bits_off) : nullptr, stream);
        } else {
          rc = pxl_residual_fwd_bits(dt, (long)n->B * a.H * a.W, a.Cp, at(arena, a.off), ac, at(arena, r.off), rcoef,
                                     at(arena, o.off), op.bits && n->pack_dgrad ? at(arena, op.bits_off) : nullptr, stream);
        }
        break;
      }
      case PXL_OP_ACT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        const void* pre = at(arena, tin.off);
        if (d.bn_in0 >= 0) {          // (the producing convolution finalized this BN: it is neither materialised nor folded)
          rc = pxl_bn_apply_fwd(dt, (long)n->B * tin.H * tin.W, tin.Cp, at(arena, tin.off), fat(arena, n->bns[d.bn_in0].coef_off), 0,
                                at(arena, op.ws_off), stream);
          if (rc != PXL_OK) return rc;
          pre = at(arena, op.ws_off);
        }
        rc = pxl_leaky_fwd(dt, (long)n->B * tin.H * tin.W * tin.Cp, pre, d.slope, at(arena, tout.off), stream);
        break;
      }
      case PXL_OP_IBN: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        const BnInfo& b = n->bns[d.bn_out];
        const int nb = b.d.C, HW = tin.H * tin.W;
        const int train = training ? 1 : 0;
        rc = pxl_ibn_stats_acc(dt, n->B, HW, tin.Cp, at(arena, tin.off), fat(arena, op.ibn_sums), stream);
        if (rc != PXL_OK) return rc;
        // multi-rank: batch sums of the BN half folded over the samples, exchanged, then the coefficients; one rank: the
        // coefficient kernel folds them itself (one launch less per layer)
        const bool exchange = train && n->sync && n->world > 1;
        if (exchange) {
          rc = pxl_ibn_fold(n->B, tin.Cp, nb, fat(arena, op.ibn_sums), fat(arena, op.ibn_bn), nullptr, nullptr, stream);
          if (rc != PXL_OK) return rc;
          rc = n->sync(n->sync_user, fat(arena, op.ibn_bn), 2 * nb, stream);
          if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_forward: SyncBN all-reduce hook failed (%d)", rc);
        }
        rc = pxl_ibn_coef(n->B, tin.Cp, nb, HW, (float)n->B * HW * n->world, fat(arena, op.ibn_sums),
                          exchange ? fat(arena, op.ibn_bn) : nullptr,
                          params + b.d.gamma_off, params + b.d.beta_off, running ? running + b.d.rmean_off : nullptr,
                          running ? running + b.d.rvar_off : nullptr, n->eff_momentum(b.d.momentum), b.d.eps, train,
                          (n->world > 1 || n->force_clamp) ? 1 : 0, fat(arena, op.ibn_coef), stream);
        if (rc != PXL_OK) return rc;
        rc = pxl_ibn_apply_fwd(dt, n->B, HW, tin.Cp, at(arena, tin.off), fat(arena, op.ibn_coef), d.slope,
                               at(arena, tout.off), stream);
        break;
      }
      case PXL_OP_AVGPOOL: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        rc = pxl_adaptive_avgpool_fwd(dt, n->B, tin.H, tin.W, tin.Cp, d.kh, at(arena, tin.off), at(arena, tout.off), stream);
        break;
      }
      case PXL_OP_CONCAT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        rc = pxl_slice_copy(dt, (long)n->B * tin.H * tin.W, tin.C, at(arena, tin.off), tin.Cp, 0, at(arena, tout.off), tout.Cp, 0,
                            0, stream);
        break;
      }
      case PXL_OP_UPCAT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        const float* coef = d.bn_in0 >= 0 ? fat(arena, n->bns[d.bn_in0].coef_off) : nullptr;
        rc = pxl_upsample_slice_fwd(dt, n->B, tin.H, tin.W, tin.Cp, tin.C, at(arena, tin.off), coef, coef ? 1 : 0, tout.H, tout.W,
                                    at(arena, tout.off), tout.Cp, d.c_off, stream);
        break;
      }
      case PXL_OP_PIXSHUF: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        rc = pxl_pixshuf_relu_fwd(dt, n->B, tin.H, tin.W, tin.Cp, tout.C, at(arena, tin.off), at(arena, tout.off), tout.Cp, stream);
        break;
      }
      case PXL_OP_HEAD: {
        if (logits == nullptr) break;
        const TensorInfo& low = n->tensors[d.in0];
        rc = pxl_upsample_softmax_fwd(dt, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off),
                                      logits, prob, stream);
        break;
      }
    }
  return rc;
}
}  // namespace

extern "C" int pxl_net_forward(pxl_net* n, const float* params, const void* packed, float* running,
                               const float* x, float* logits, float* prob, void* arena, size_t arena_bytes,
                               int training, void* stream) {
  // logits == NULL: the HEAD op (up-sampling + soft-max to full resolution) is skipped -- the caller consumes the
  // low-resolution logits in the arena directly (pxl_net_head_loss)
  PXL_REQUIRE(n && n->planned && params && packed && (x || n->in_parts > 0) && arena, "net_forward: bad argument (plan first)");
  if (arena_bytes < n->arena_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_forward: arena too small (%zu < %zu)", arena_bytes, n->arena_bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  n->patch_src = n->patch_src_next;           // (a borrowed stem operand serves exactly this pass and its backward)
  n->patch_src_next = nullptr;
  if (training && n->stats_region_bytes)
    PXL_CHECK_HIP(hipMemsetAsync(at(arena, n->stats_region_off), 0, n->stats_region_bytes, s));
  if (n->ibn_region_bytes)      // (instance statistics are computed in eval mode too)
    PXL_CHECK_HIP(hipMemsetAsync(at(arena, n->ibn_region_off), 0, n->ibn_region_bytes, s));
  const FwdCtx ctx{params, packed, running, x, logits, prob, arena, training, stream};
  const long t_all = pxlht::on ? pxlht::now() : 0;
  for (size_t i = 0; i < n->ops.size(); ++i) {
    bool fin = false;
    const long t0 = pxlht::on ? pxlht::now() : 0;
    const int rc = forward_op(n, i, ctx, 0, &fin);
    if (pxlht::on) pxlht::add(n->ops[i].d.kind & 15, pxlht::now() - t0);
    if (rc != PXL_OK) return rc;
  }
  if (pxlht::on) pxlht::add(18, pxlht::now() - t_all);
  return PXL_OK;
}

// ---- stem patches of ONE input tensor shared by two networks (include/pixelhip.h) ----
namespace {
const OpInfo* patch_op(const pxl_net* n) {
  for (const auto& op : n->ops) if (op.d.kind == PXL_OP_CONV && op.patch) return &op;
  return nullptr;
}
size_t patch_bytes(const pxl_net* n, const OpInfo& op) {
  const TensorInfo& tout = n->tensors[op.d.out];
  return (size_t)n->B * tout.H * tout.W * op.patch_Kp * n->esize;
}
}  // namespace
extern "C" int pxl_net_make_patches(pxl_net* n, const float* x, void* arena, size_t arena_bytes, void** patches, size_t* bytes,
                                    void* stream) {
  PXL_REQUIRE(n && n->planned && x && arena && patches && bytes, "net_make_patches: bad argument (plan first)");
  if (arena_bytes < n->arena_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_make_patches: arena too small");
  const OpInfo* op = patch_op(n);
  if (op == nullptr || n->in_parts > 0) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_make_patches: no patch-mode stem in this plan");
  const pxl_op& d = op->d;
  const TensorInfo& tin = n->tensors[d.in0];
  const TensorInfo& tout = n->tensors[d.out];
  const int rc = pxl_stem_patches(n->dtype, x, at(arena, op->patch_off), n->B, d.cin, tin.H, tin.W, d.kh, d.kw, d.stride, d.pads[0],
                                  tout.H, tout.W, op->patch_Kp, stream);
  if (rc != PXL_OK) return rc;
  n->patches_made = true;
  *patches = at(arena, op->patch_off);
  *bytes = patch_bytes(n, *op);
  return PXL_OK;
}
extern "C" int pxl_net_borrow_patches(pxl_net* n, const void* patches, size_t bytes) {
  PXL_REQUIRE(n && n->planned, "net_borrow_patches: bad argument (plan first)");
  if (patches == nullptr) { n->patch_src_next = nullptr; return PXL_OK; }
  const OpInfo* op = patch_op(n);
  if (op == nullptr || n->in_parts > 0) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_borrow_patches: no patch-mode stem in this plan");
  if (bytes != patch_bytes(n, *op)) return pxl_set_error(PXL_ERR_ARG, "net_borrow_patches: %zu bytes lent, this stem reads %zu", bytes, patch_bytes(n, *op));
  n->patch_src_next = patches;
  return PXL_OK;
}

extern "C" void pxl_dma_capture_begin(void** slot);
extern "C" void pxl_dma_capture_end(void);
extern "C" int pxl_dma_launch_captured(void* slot0, void* slot1);
extern "C" void pxl_elt_pair_begin(void);
extern "C" int pxl_elt_pair_end(void);

// Forward pass of TWO networks with the same program (Mean Teacher's student || teacher, GCT's l || r task models) in
// lockstep on ONE stream: op by op, and every convolution of the pair as ONE launch (conv_dma.hip: gridDim.z = 2) where the
// two launches match.  Per network the arguments and the results are those of pxl_net_forward; nothing is shared between the
// two passes.  Why: at M = 8 x 33 x 33 a ResNet stage-3 convolution has 138 - 274 tiles for 256 CUs -- two networks' tiles in one
// launch fill the chip with tiles of twice the arithmetic intensity, and the pair costs one launch latency instead of two
// (DESIGN.md 4); with Sync-BN both networks' exchanges are issued from this one thread in one order on every rank.
extern "C" int pxl_net_forward_pair(pxl_net* n0, pxl_net* n1, const float* params0, const float* params1, const void* packed0,
                                    const void* packed1, float* running0, float* running1, const float* x0, const float* x1,
                                    float* logits0, float* prob0, float* logits1, float* prob1, void* arena0, void* arena1,
                                    size_t arena_bytes0, size_t arena_bytes1, int training0, int training1, void* stream) {
  PXL_REQUIRE(n0 && n1 && n0 != n1 && n0->planned && n1->planned && params0 && params1 && packed0 && packed1 && x0 && x1 && arena0 && arena1,
              "net_forward_pair: bad argument (plan both networks first)");
  PXL_REQUIRE(n0->ops.size() == n1->ops.size() && n0->B == n1->B && n0->H == n1->H && n0->W == n1->W && n0->dtype == n1->dtype &&
              n0->in_parts == 0 && n1->in_parts == 0, "net_forward_pair: the two networks must run the same program on the same shape");
  for (size_t i = 0; i < n0->ops.size(); ++i)
    PXL_REQUIRE(n0->ops[i].d.kind == n1->ops[i].d.kind, "net_forward_pair: op %zu differs between the networks", i);
  if (arena_bytes0 < n0->arena_bytes || arena_bytes1 < n1->arena_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_forward_pair: arena too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  pxl_net* nets[2] = {n0, n1};
  void* arenas[2] = {arena0, arena1};
  const int trainings[2] = {training0, training1};
  for (int k = 0; k < 2; ++k) {
    if (trainings[k] && nets[k]->stats_region_bytes)
      PXL_CHECK_HIP(hipMemsetAsync(at(arenas[k], nets[k]->stats_region_off), 0, nets[k]->stats_region_bytes, s));
    if (nets[k]->ibn_region_bytes)
      PXL_CHECK_HIP(hipMemsetAsync(at(arenas[k], nets[k]->ibn_region_off), 0, nets[k]->ibn_region_bytes, s));
  }
  const FwdCtx ctx[2] = {{params0, packed0, running0, x0, logits0, prob0, arena0, training0, stream},
                         {params1, packed1, running1, x1, logits1, prob1, arena1, training1, stream}};
  n0->pairs_last = 0;
  n0->pair_syncs_last = 0;
  for (int k = 0; k < 2; ++k) { nets[k]->patch_src = nullptr; nets[k]->patch_src_next = nullptr; }
  for (size_t i = 0; i < n0->ops.size(); ++i) {
    if (n0->ops[i].d.kind != PXL_OP_CONV || n0->profile || n1->profile) {
      // (the finalize-folding element-wise kernels of the two networks -- residual joins, BN + ReLU -- pair up as well:
      // eltwise.hip holds the first network's launch back until the second one's arrives)
      pxl_elt_pair_begin();
      for (int k = 0; k < 2; ++k) {
        bool fin = false;
        const int rc = forward_op(nets[k], i, ctx[k], 0, &fin);
        if (rc != PXL_OK) { (void)pxl_elt_pair_end(); return rc; }
      }
      const int rc = pxl_elt_pair_end();
      if (rc != PXL_OK) return rc;
      continue;
    }
    bool fin[2] = {false, false};
    void* slot[2] = {nullptr, nullptr};
    // both halves on the tile configuration tuned for the PAIR (twice the tiles per launch: larger tiles win), or on the first
    // network's own choice when the pair has not been tuned
    const int pcfg = n0->ops[i].pair_cfg != -2 ? n0->ops[i].pair_cfg : n0->ops[i].fwd.tile_cfg;
    for (int k = 0; k < 2; ++k) {
      const int keep = nets[k]->ops[i].fwd.tile_cfg;
      nets[k]->ops[i].fwd.tile_cfg = pcfg;
      pxl_dma_capture_begin(&slot[k]);
      const int rc = forward_op(nets[k], i, ctx[k], 1, &fin[k]);
      pxl_dma_capture_end();
      nets[k]->ops[i].fwd.tile_cfg = keep;
      if (rc != PXL_OK) { (void)pxl_dma_launch_captured(slot[0], slot[1]); return rc; }
    }
    const int pr = pxl_dma_launch_captured(slot[0], slot[1]);
    if (pr < 0) return pr;
    n0->pairs_last += pr;
    // Sync-BN: the statistics of the SAME BatchNorm of both networks travel in one exchange (one launch folds the replicas
    // of both, posts 2 x 2C words, sums them): both networks on the peer-mapped path, neither finalized by its convolution
    bool sync_done = false;
    {
      const pxl_op& d0 = n0->ops[i].d; const pxl_op& d1 = n1->ops[i].d;
      if (d0.bn_out >= 0 && d1.bn_out >= 0 && training0 && training1 && !fin[0] && !fin[1] && n0->world > 1 && n1->world > 1 &&
          n0->sync == &pxl_peer_allreduce_hook && n1->sync == &pxl_peer_allreduce_hook && n0->pair_sync &&
          n0->bns[d0.bn_out].d.C == n1->bns[d1.bn_out].d.C && n0->bns[d0.bn_out].nrep == n1->bns[d1.bn_out].nrep) {
        const int rc = pxl_peer_allreduce_fold(reinterpret_cast<pxl_peer*>(n0->sync_user), fat(arena0, n0->bns[d0.bn_out].stats_off),
                                               fat(arena1, n1->bns[d1.bn_out].stats_off), 2 * n0->bns[d0.bn_out].d.C, n0->bns[d0.bn_out].nrep, stream);
        if (rc != PXL_OK) return rc;
        sync_done = true;
        ++n0->pair_syncs_last;
      }
    }
    pxl_elt_pair_begin();
    for (int k = 0; k < 2; ++k) {
      const int rc = forward_op(nets[k], i, ctx[k], 2, &fin[k], sync_done);
      if (rc != PXL_OK) { (void)pxl_elt_pair_end(); return rc; }
    }
    const int rce = pxl_elt_pair_end();
    if (rce != PXL_OK) return rce;
  }
  return PXL_OK;
}

// convolutions the last pxl_net_forward_pair(n, ...) issued as paired launches (tests / bench)
extern "C" int pxl_net_pairs(const pxl_net* n) { return n ? n->pairs_last : 0; }
// ... Sync-BN statistics exchanges the last paired pass issued for both networks at once (pxl_peer_allreduce_fold)
extern "C" int pxl_net_pair_syncs(const pxl_net* n) { return n ? n->pair_syncs_last : 0; }

// Tile selection for the paired forward: every convolution of the pair timed as ONE launch per candidate configuration
// (pxl_net_tune times single launches: with twice the tiles per launch the larger tiles win -- tools/cbench --pair: 160 x 128
// instead of 64 x 128 on the stage-3 shapes, the pair at 1.4 - 1.7x a single launch instead of 2x).  Both networks planned,
// packed; arenas are clobbered (call it at warm-up, like pxl_net_tune).  Results do not depend on the choice.
extern "C" int pxl_net_tune_pair(pxl_net* n0, pxl_net* n1, const float* params0, const float* params1, const void* packed0,
                                 const void* packed1, void* arena0, void* arena1, size_t arena_bytes0, size_t arena_bytes1,
                                 void* stream) {
  PXL_REQUIRE(n0 && n1 && n0 != n1 && n0->planned && n1->planned && params0 && params1 && packed0 && packed1 && arena0 && arena1,
              "net_tune_pair: bad argument");
  PXL_REQUIRE(n0->ops.size() == n1->ops.size() && n0->B == n1->B && n0->H == n1->H && n0->W == n1->W && n0->dtype == n1->dtype,
              "net_tune_pair: the two networks must run the same program on the same shape");
  if (arena_bytes0 < n0->arena_bytes || arena_bytes1 < n1->arena_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_tune_pair: arena too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t ea, eb;
  PXL_CHECK_HIP(hipEventCreate(&ea));
  PXL_CHECK_HIP(hipEventCreate(&eb));
  pxl_net* nets[2] = {n0, n1};
  const FwdCtx ctx[2] = {{params0, packed0, nullptr, nullptr, nullptr, nullptr, arena0, 1, stream},
                         {params1, packed1, nullptr, nullptr, nullptr, nullptr, arena1, 1, stream}};
  for (int k = 0; k < 2; ++k)
    if (nets[k]->stats_region_bytes) PXL_CHECK_HIP(hipMemsetAsync(at(k ? arena1 : arena0, nets[k]->stats_region_off), 0, nets[k]->stats_region_bytes, s));
  int rc_all = PXL_OK;
  for (size_t i = 0; i < n0->ops.size(); ++i) {
    OpInfo& op = n0->ops[i];
    if (op.d.kind != PXL_OP_CONV || op.patch || n1->ops[i].d.kind != PXL_OP_CONV) continue;      // (the stem reads x: not available here)
    const TensorInfo& tout = n0->tensors[op.d.out];
    int best_cfg = -2; float best = 1e30f;
    for (int cfg = 8; cfg < 36; ++cfg) {
      if (cfg >= 12 && cfg < 16) continue;
      if (cfg >= 20 && cfg != 29 && tout.Cp < 128) continue;
      if (cfg == 35 && tout.Cp < 256) continue;
      float t_best = 1e30f; bool ok = true;
      for (int rep = 0; rep < 4 && ok; ++rep) {               // one warm-up + three timed launches
        void* slot[2] = {nullptr, nullptr};
        if (rep) ok = hipEventRecord(ea, s) == hipSuccess;
        for (int k = 0; k < 2 && ok; ++k) {
          const int keep = nets[k]->ops[i].fwd.tile_cfg;
          nets[k]->ops[i].fwd.tile_cfg = cfg;
          bool fin = false;
          pxl_dma_capture_begin(&slot[k]);
          ok = forward_op(nets[k], i, ctx[k], 1, &fin) == PXL_OK;
          pxl_dma_capture_end();
          nets[k]->ops[i].fwd.tile_cfg = keep;
        }
        const int pr = pxl_dma_launch_captured(slot[0], slot[1]);
        if (pr != 1) ok = false;                              // not pairable with this configuration (or an error): not a candidate
        if (ok && rep) {
          float ms = 0.f;
          ok = hipEventRecord(eb, s) == hipSuccess && hipEventSynchronize(eb) == hipSuccess && hipEventElapsedTime(&ms, ea, eb) == hipSuccess;
          if (ok && ms < t_best) t_best = ms;
        }
      }
      (void)hipGetLastError();
      if (ok && t_best < best) { best = t_best; best_cfg = cfg; }
    }
    op.pair_cfg = best_cfg;
    n1->ops[i].pair_cfg = best_cfg;
  }
  (void)hipStreamSynchronize(s);
  (void)hipEventDestroy(ea);
  (void)hipEventDestroy(eb);
  return rc_all;
}

extern "C" int pxl_net_set_wgrad(pxl_net* net, int enable) {
  PXL_REQUIRE(net, "net_set_wgrad: null net");
  net->wgrad_on = enable != 0;
  return PXL_OK;
}

// The following forward passes update the BatchNorm running statistics as if each of them had run `times` times on the same
// batch: running <- (1-m)^times running + (1 - (1-m)^times) batch, i.e. momentum 1 - (1-m)^times.  For callers that would
// otherwise run the SAME forward twice with unchanged weights (SSLGCT's step-0 no-grad pass and step-1 pass, ssl_gct.py:196-200
// + 403) and run it once instead.  times = 1 restores the plain update.
extern "C" int pxl_net_set_bn_repeat(pxl_net* net, int times) {
  PXL_REQUIRE(net && times >= 1 && times <= 16, "net_set_bn_repeat: bad argument");
  net->bn_repeat = times;
  return PXL_OK;
}

// enable = 1: pxl_net_tune times every forward tile candidate with TWO copies of the launch in flight on two streams and keeps the
// tile whose pair finishes first -- for a network whose forward pass runs beside a copy of itself (Mean Teacher's student ||
// teacher on two streams).  Call before the first forward pass of a shape (the pass that tunes).
extern "C" int pxl_net_set_tune_dual(pxl_net* net, int enable) {
  PXL_REQUIRE(net, "net_set_tune_dual: null net");
  net->tune_dual = enable != 0 ? 1 : 0;
  return PXL_OK;
}

extern "C" int pxl_net_set_pack_dgrad(pxl_net* net, int enable) {
  PXL_REQUIRE(net, "net_set_pack_dgrad: null net");
  net->pack_dgrad = enable != 0;
  return PXL_OK;
}

// The next forward pass gathers its input from `nparts` NCHW fp32 tensors concatenated along the channels (FlawDetector:
// image + task prediction, ssl_gct.py:578) instead of one tensor; cleared by that pass.
extern "C" int pxl_net_set_input_parts(pxl_net* n, int nparts, const float* const* srcs, const int* chans) {
  PXL_REQUIRE(n && nparts >= 0 && nparts <= 4 && (nparts == 0 || (srcs && chans)), "net_set_input_parts: bad argument");
  n->in_parts = nparts;
  for (int k = 0; k < nparts; ++k) { n->in_src[k] = srcs[k]; n->in_chans[k] = chans[k]; }
  return PXL_OK;
}

// input gradient split into one NCHW tensor per concatenated part (NULL = not needed)
extern "C" int pxl_net_input_grad_parts(pxl_net* n, const void* scratch, int nparts, float* const* dsts, const int* chans,
                                        void* stream) {
  PXL_REQUIRE(n && n->planned && scratch && dsts && chans && n->input_tensor >= 0, "net_input_grad_parts: bad argument");
  const TensorInfo& t = n->tensors[n->input_tensor];
  int C = 0;
  for (int k = 0; k < nparts; ++k) C += chans[k];
  PXL_REQUIRE(C == t.C, "net_input_grad_parts: parts hold %d channels, the input has %d", C, t.C);
  return pxl_nhwc_to_nchw_parts(n->dtype, at(scratch, t.goff), nparts, dsts, chans, n->B, t.H, t.W, t.Cp, stream);
}

extern "C" int pxl_net_input_grad(pxl_net* n, const void* scratch, float* dx, void* stream) {
  PXL_REQUIRE(n && n->planned && scratch && dx && n->input_tensor >= 0, "net_input_grad: bad argument");
  const TensorInfo& t = n->tensors[n->input_tensor];
  return pxl_nhwc_to_nchw(n->dtype, at(scratch, t.goff), dx, n->B, t.C, t.H, t.W, t.Cp, stream);
}

extern "C" int pxl_net_latent_shape(const pxl_net* n, int* C, int* h, int* w) {
  PXL_REQUIRE(n && n->planned && n->head_op >= 0, "net_latent_shape: plan first");
  const int t = n->ops[n->head_op].d.in1;
  PXL_REQUIRE(t >= 0, "net_latent_shape: program names no latent tensor");
  if (C) *C = n->tensors[t].C;
  if (h) *h = n->tensors[t].H;
  if (w) *w = n->tensors[t].W;
  return PXL_OK;
}

extern "C" int pxl_net_latent(pxl_net* n, const void* arena, float* latent, void* stream) {
  PXL_REQUIRE(n && n->planned && arena && latent && n->head_op >= 0, "net_latent: bad argument");
  const int t = n->ops[n->head_op].d.in1;
  PXL_REQUIRE(t >= 0, "net_latent: program names no latent tensor");
  const TensorInfo& ti = n->tensors[t];
  const int bn = n->ops[n->head_op].d.bn_in1;
  const size_t off = bn >= 0 ? n->bns[bn].z_off : ti.off;       // PSPNet: the latent is relu(bn(bottleneck))
  return pxl_nhwc_to_nchw(n->dtype, at(arena, off), latent, n->B, ti.C, ti.H, ti.W, ti.Cp, stream);
}

// Inspection: forward tensor `tensor` of the pass held in `arena` as NCHW fp32; with bn >= 0 the BatchNorm's affine is
// applied first (z = y * scale + shift, the value whose sign the ReLU after that BN decides).  `tmp` = tensor-sized
// device scratch (needed only with bn >= 0).
extern "C" int pxl_net_read_tensor(pxl_net* n, const void* arena, int tensor, int bn, void* tmp, float* out, void* stream) {
  PXL_REQUIRE(n && n->planned && arena && out && tensor >= 0 && tensor < (int)n->tensors.size(), "net_read_tensor: bad argument");
  const TensorInfo& t = n->tensors[tensor];
  PXL_REQUIRE(t.planned, "net_read_tensor: tensor %d is not part of the plan", tensor);
  const void* src = at(arena, t.off);
  if (bn >= 0) {
    PXL_REQUIRE(bn < (int)n->bns.size() && tmp && n->bns[bn].d.C == t.Cp, "net_read_tensor: BN %d does not fit tensor %d", bn, tensor);
    int rc = pxl_bn_apply_fwd(n->dtype, (long)n->B * t.H * t.W, t.Cp, src, fat(arena, n->bns[bn].coef_off), 0, tmp, stream);
    if (rc != PXL_OK) return rc;
    src = tmp;
  }
  return pxl_nhwc_to_nchw(n->dtype, src, out, n->B, t.C, t.H, t.W, t.Cp, stream);
}

extern "C" size_t pxl_net_tensor_bytes(const pxl_net* n, int tensor) {
  return (n && n->planned && tensor >= 0 && tensor < (int)n->tensors.size()) ? n->tensors[tensor].bytes : 0;
}

extern "C" int pxl_net_tensor_shape(const pxl_net* n, int tensor, int* C, int* h, int* w) {
  PXL_REQUIRE(n && n->planned && tensor >= 0 && tensor < (int)n->tensors.size() && n->tensors[tensor].planned, "net_tensor_shape: bad argument");
  if (C) *C = n->tensors[tensor].C;
  if (h) *h = n->tensors[tensor].H;
  if (w) *w = n->tensors[tensor].W;
  return PXL_OK;
}

extern "C" int pxl_net_seed_latent_grad(pxl_net* n, void* scratch, size_t scratch_bytes, const float* dlatent, void* stream) {
  PXL_REQUIRE(n && n->planned && scratch && dlatent && n->head_op >= 0, "net_seed_latent_grad: bad argument");
  if (scratch_bytes < n->scratch_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_seed_latent_grad: scratch too small");
  const int t = n->ops[n->head_op].d.in1;
  PXL_REQUIRE(t >= 0, "net_seed_latent_grad: program names no latent tensor");
  const TensorInfo& ti = n->tensors[t];
  int rc = pxl_nchw_to_nhwc(n->dtype, dlatent, at(scratch, ti.goff), n->B, ti.C, ti.H, ti.W, ti.Cp, stream);
  if (rc == PXL_OK) n->latent_seeded = true;
  return rc;
}

namespace {
int net_backward_impl(pxl_net* n, const float* params, const void* packed, const float* dlogits, const float* dprob,
                      const float* prob, float* grads, void* arena, size_t arena_bytes, void* scratch, size_t scratch_bytes,
                      int training, void* stream, bool from_low);
}

extern "C" int pxl_net_backward(pxl_net* n, const float* params, const void* packed, const float* dlogits,
                                const float* dprob, const float* prob, float* grads, void* arena,
                                size_t arena_bytes, void* scratch, size_t scratch_bytes, int training,
                                void* stream) {
  return net_backward_impl(n, params, packed, dlogits, dprob, prob, grads, arena, arena_bytes, scratch, scratch_bytes, training,
                           stream, false);
}

// Backward pass that starts from d(loss)/d(low-resolution logits), already written into the gradient slot of the HEAD's
// input tensor by pxl_net_head_loss: the HEAD op's own backward is skipped, everything else is pxl_net_backward.
extern "C" int pxl_net_backward_low(pxl_net* n, const float* params, const void* packed, float* grads, void* arena,
                                    size_t arena_bytes, void* scratch, size_t scratch_bytes, int training, void* stream) {
  return net_backward_impl(n, params, packed, nullptr, nullptr, nullptr, grads, arena, arena_bytes, scratch, scratch_bytes,
                           training, stream, true);
}

// Fused training seam on the low-resolution logits of one (student) or two (student + teacher) forward passes held in
// their arenas: per-sample CE of both networks, the MSE consistency term and d(loss)/d(student low-res logits) into the
// student's gradient slot (csrc/head.hip: pxl_head_loss).  `teacher` / `t_arena` NULL: no teacher terms.
namespace {
int net_head_loss_impl(pxl_net* n, const void* arena, const pxl_net* teacher, const void* t_arena, const float* gt,
                       int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight, float mse_weight, const float* mse_weight_dev,
                       void* scratch, size_t scratch_bytes, float* sums, void* stream) {
  PXL_REQUIRE(n && n->planned && arena && scratch && sums && n->head_op >= 0, "net_head_loss: bad argument (plan first)");
  if (scratch_bytes < n->scratch_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_head_loss: scratch too small");
  const pxl_op& d = n->ops[n->head_op].d;
  const TensorInfo& low = n->tensors[d.in0];
  const void* t_low = nullptr;
  if (teacher != nullptr && t_arena != nullptr) {
    PXL_REQUIRE(teacher->planned && teacher->head_op >= 0, "net_head_loss: teacher is not planned");
    const TensorInfo& tl = teacher->tensors[teacher->ops[teacher->head_op].d.in0];
    PXL_REQUIRE(teacher->dtype == n->dtype && teacher->B == n->B && tl.H == low.H && tl.W == low.W && tl.Cp == low.Cp &&
                teacher->Ho == n->Ho && teacher->Wo == n->Wo, "net_head_loss: student and teacher plans differ");
    t_low = at(t_arena, tl.off);
  }
  if (n->deterministic)       // bit-reproducible: row-wise kernel (plain stores), loss sums folded in row order
    return pxl_head_loss_ex(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off), t_low, gt,
                            ignore_index, n_ce, mse_lo, mse_hi, ce_weight, mse_weight, mse_weight_dev, 0, 1, at(scratch, low.goff),
                            at(scratch, n->up_ws_off), n->up_ws_bytes, sums, stream);
  if (mse_weight_dev != nullptr)
    return pxl_head_loss_hp(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off), t_low, gt,
                            ignore_index, n_ce, mse_lo, mse_hi, ce_weight, mse_weight_dev, at(scratch, low.goff),
                            at(scratch, n->up_ws_off), n->up_ws_bytes, sums, stream);
  return pxl_head_loss(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off), t_low, gt,
                       ignore_index, n_ce, mse_lo, mse_hi, ce_weight, mse_weight, at(scratch, low.goff), at(scratch, n->up_ws_off),
                       n->up_ws_bytes, sums, stream);
}
}  // namespace

extern "C" int pxl_net_head_loss(pxl_net* n, const void* arena, const pxl_net* teacher, const void* t_arena, const float* gt,
                                 int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight, float mse_weight,
                                 void* scratch, size_t scratch_bytes, float* sums, void* stream) {
  return net_head_loss_impl(n, arena, teacher, t_arena, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_weight, mse_weight, nullptr, scratch,
                            scratch_bytes, sums, stream);
}

// ... with the consistency weight read from device memory (a captured training step: the weight ramps up from step to step)
extern "C" int pxl_net_head_loss_hp(pxl_net* n, const void* arena, const pxl_net* teacher, const void* t_arena, const float* gt,
                                    int ignore_index, int n_ce, int mse_lo, int mse_hi, float ce_weight, const float* mse_weight_dev,
                                    void* scratch, size_t scratch_bytes, float* sums, void* stream) {
  PXL_REQUIRE(mse_weight_dev != nullptr, "net_head_loss_hp: null weight pointer");
  return net_head_loss_impl(n, arena, teacher, t_arena, gt, ignore_index, n_ce, mse_lo, mse_hi, ce_weight, 0.f, mse_weight_dev, scratch,
                            scratch_bytes, sums, stream);
}

// The HEAD op alone (up-sampling + soft-max of the low-resolution logits held in `arena`): materialises the
// full-resolution planes of a pass that ran with logits == NULL, for a consumer that wants them after all
extern "C" int pxl_net_head_forward(pxl_net* n, const void* arena, float* logits, float* prob, void* stream) {
  PXL_REQUIRE(n && n->planned && arena && logits && n->head_op >= 0, "net_head_forward: bad argument (plan first)");
  const pxl_op& d = n->ops[n->head_op].d;
  const TensorInfo& low = n->tensors[d.in0];
  return pxl_upsample_softmax_fwd(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off),
                                  logits, prob, stream);
}

// Consistency seam of an SSLCCT auxiliary decoder (csrc/head.hip: pxl_cons_head_fwd / _bwd) on the low-resolution logits of a pass
// that ran with logits == NULL: the forward half writes the loss and parks the row-reduced gradient (for a unit incoming gradient)
// in the plan's up-sampling workspace; the backward half turns it into d(loss)/d(low) in the gradient slot pxl_net_backward_low
// starts from, scaled by the incoming gradient gout[0] (device memory).  Nothing else of this plan may run in between.
extern "C" int pxl_net_cons_head_supported(const pxl_net* n) {
  if (!n || !n->planned || n->head_op < 0) return 0;
  const TensorInfo& low = n->tensors[n->ops[n->head_op].d.in0];
  return n->classes <= 32 && low.Cp % 8 == 0 && pxl_cons_head_lds_bytes(low.W, n->classes, n->Wo) <= 64 * 1024 &&
         n->up_ws_bytes >= pxl_cons_head_workspace(n->B, low.W, n->classes, n->Ho) ? 1 : 0;
}
extern "C" int pxl_net_cons_head_fwd(pxl_net* n, const void* arena, const float* target, void* scratch, size_t scratch_bytes, float* loss,
                                     void* stream) {
  PXL_REQUIRE(n && n->planned && arena && target && scratch && loss && n->head_op >= 0, "net_cons_head_fwd: bad argument (plan first)");
  if (scratch_bytes < n->scratch_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_cons_head_fwd: scratch too small");
  const pxl_op& d = n->ops[n->head_op].d;
  const TensorInfo& low = n->tensors[d.in0];
  return pxl_cons_head_fwd(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, at(arena, low.off), target,
                           at(scratch, n->up_ws_off), n->up_ws_bytes, loss, n->deterministic ? 1 : 0, stream);
}
extern "C" int pxl_net_cons_head_bwd(pxl_net* n, void* scratch, size_t scratch_bytes, const float* gout, void* stream) {
  PXL_REQUIRE(n && n->planned && scratch && gout && n->head_op >= 0, "net_cons_head_bwd: bad argument (plan first)");
  if (scratch_bytes < n->scratch_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_cons_head_bwd: scratch too small");
  const pxl_op& d = n->ops[n->head_op].d;
  const TensorInfo& low = n->tensors[d.in0];
  return pxl_cons_head_bwd(n->dtype, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, d.stride, at(scratch, n->up_ws_off), n->up_ws_bytes,
                           gout, at(scratch, low.goff), stream);
}

// 1 when pxl_net_head_loss can run on this plan (the full-resolution row fits the kernel's LDS staging)
extern "C" int pxl_net_head_loss_supported(const pxl_net* n) {
  if (!n || !n->planned || n->head_op < 0) return 0;
  const TensorInfo& low = n->tensors[n->ops[n->head_op].d.in0];
  return n->classes <= 32 && pxl_head_loss_lds_bytes(low.W, n->classes, n->Wo) <= 64 * 1024 ? 1 : 0;
}

namespace {
int net_backward_impl(pxl_net* n, const float* params, const void* packed, const float* dlogits, const float* dprob,
                      const float* prob, float* grads, void* arena, size_t arena_bytes, void* scratch, size_t scratch_bytes,
                      int training, void* stream, bool from_low) {
  PXL_REQUIRE(n && n->planned && params && packed && grads && arena && scratch, "net_backward: bad argument");
  PXL_REQUIRE(from_low || dlogits || dprob, "net_backward: no incoming gradient");
  PXL_REQUIRE(n->pack_dgrad, "net_backward: this network packs no data-gradient weights (pxl_net_set_pack_dgrad(net, 0))");
  if (arena_bytes < n->arena_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_backward: arena too small");
  if (scratch_bytes < n->scratch_bytes) return pxl_set_error(PXL_ERR_WORKSPACE, "net_backward: scratch too small (%zu < %zu)", scratch_bytes, n->scratch_bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int dt = n->dtype;
  if (n->bsum_region_bytes)
    PXL_CHECK_HIP(hipMemsetAsync(at(scratch, n->bsum_region_off), 0, n->bsum_region_bytes, s));
  if (n->ibn_bregion_bytes)
    PXL_CHECK_HIP(hipMemsetAsync(at(scratch, n->ibn_bregion_off), 0, n->ibn_bregion_bytes, s));
  std::vector<char> written(n->tensors.size(), 0);
  std::vector<char> reduced(n->bns.size(), 0);        // BN-backward sums already produced by a fused launch
  // where the gradient of a tensor currently lives: its own buffer, or -- after a fused residual join -- the buffer of
  // the join's output, which both branches then READ (nothing is copied); settle() materialises it for ops that
  // accumulate in place
  std::vector<size_t> gsrc(n->tensors.size());
  for (size_t t = 0; t < n->tensors.size(); ++t) gsrc[t] = n->tensors[t].goff;
  std::vector<char> join_done(n->ops.size(), 0);
  auto settle = [&](int t) -> int {
    if (t < 0 || gsrc[t] == n->tensors[t].goff) return PXL_OK;
    const TensorInfo& ti = n->tensors[t];
    PXL_CHECK_HIP(hipMemcpyAsync(at(scratch, ti.goff), at(scratch, gsrc[t]), (size_t)n->B * ti.H * ti.W * ti.Cp * n->esize,
                                 hipMemcpyDeviceToDevice, s));
    gsrc[t] = ti.goff;
    return PXL_OK;
  };
  if (n->latent_seeded) {
    written[n->ops[n->head_op].d.in1] = 1;
    n->latent_seeded = false;
  }
  if (n->use_side < 0) {
    const char* e = getenv("PXL_SIDE_STREAM");
    n->use_side = (e && e[0] == '0') ? 0 : 1;
    if (n->use_side) {
      // Default priority.  PXL_SIDE_PRIO=1 creates the weight-gradient stream at the LOWEST priority (its results are only
      // needed at the optimizer step): neutral for MT (+-0.1 ms), but where an algorithm keeps further streams busy
      // (AdvSSL's discriminator update, GCT's second task model) the starved stream turned into a serial tail: AdvSSL
      // 18.9 -> 34.4 ms, GCT 45.7 -> 56.2 ms (bisected, gpurun_out/r02_22)
      int lo = 0, hi = 0;
      const char* pe = getenv("PXL_SIDE_PRIO");
      // the process-wide weight-gradient stream of the placement pool (csrc/streams.hip): a hardware queue of its own,
      // shared by every plan -- a stream created here would land on whatever queue the creation order gives it
      hipStream_t placed = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_WGRAD));
      // a network whose passes run on the SIDE stream (GCT's r model, the AdvSSL discriminator) sends its weight gradients to
      // the AUX queue: the two networks' backward passes may overlap, their weight gradients then do too
      if (placed != nullptr && s == reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_SIDE)))
        placed = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_AUX));
      if (placed != nullptr && placed != s) {
        n->side = placed;
        n->side_owned = false;
      } else if (pe != nullptr && pe[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
        PXL_CHECK_HIP(hipStreamCreateWithPriority(&n->side, hipStreamNonBlocking, lo));
      else
        PXL_CHECK_HIP(hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking));
      PXL_CHECK_HIP(hipEventCreateWithFlags(&n->join_ev, hipEventDisableTiming));
      n->fork_ev.assign(n->ops.size(), nullptr);
      // a second weight-gradient stream: the SIDE queue is idle while a network that runs on the main stream goes backward
      // (the MT teacher has finished), and weight gradients of different convolutions are independent
      const char* w2 = getenv("PXL_WGRAD_STREAMS");
      if (w2 != nullptr && w2[0] == '2' && !n->side_owned) {
        hipStream_t cand = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_SIDE));
        if (cand == s) cand = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_WGRAD));
        if (cand != nullptr && cand != s && cand != n->side) {
          n->side2 = cand;
          PXL_CHECK_HIP(hipEventCreateWithFlags(&n->join_ev2, hipEventDisableTiming));
        }
      }
    }
  }
  bool forked = false, forked2 = false;
  // ---- overlapped gradient exchange (multi-rank): flush [lo, hi) of the flat gradient buffer once every kernel writing
  // into it has been issued -- the communication stream waits for the main and the weight-gradient stream at that point
  const bool bucketing = n->grad_sync && n->grad_world > 1 && n->wgrad_on && n->grad_total > 0;
  const bool updating = n->update_fn != nullptr && n->wgrad_on && n->update_total > 0 && n->bucket_ok &&
                        (!bucketing || n->update_total == n->grad_total);
  long grad_hi = bucketing ? n->grad_total : n->update_total;
  n->grad_buckets_last = 0;
  n->update_buckets_last = 0;
  if ((bucketing || updating) && !n->comm_stream) {
    hipStream_t placed = reinterpret_cast<hipStream_t>(pxl_stream_role(PXL_STREAM_AUX));
    if (placed != nullptr && placed != s) {
      n->comm_stream = placed;
      n->comm_owned = false;
    } else {
      PXL_CHECK_HIP(hipStreamCreateWithFlags(&n->comm_stream, hipStreamNonBlocking));
    }
    PXL_CHECK_HIP(hipEventCreateWithFlags(&n->comm_main_ev, hipEventDisableTiming));
    PXL_CHECK_HIP(hipEventCreateWithFlags(&n->comm_side_ev, hipEventDisableTiming));
    PXL_CHECK_HIP(hipEventCreateWithFlags(&n->comm_done_ev, hipEventDisableTiming));
  }
  auto flush = [&](long lo) -> int {
    if (!(bucketing || updating) || lo >= grad_hi) return PXL_OK;
    PXL_CHECK_HIP(hipEventRecord(n->comm_main_ev, s));
    PXL_CHECK_HIP(hipStreamWaitEvent(n->comm_stream, n->comm_main_ev, 0));
    if (forked) {
      PXL_CHECK_HIP(hipEventRecord(n->comm_side_ev, n->side));
      PXL_CHECK_HIP(hipStreamWaitEvent(n->comm_stream, n->comm_side_ev, 0));
    }
    if (forked2) {            // (a second weight-gradient stream: its kernels write into the bucket as well)
      PXL_CHECK_HIP(hipEventRecord(n->join_ev2, n->side2));
      PXL_CHECK_HIP(hipStreamWaitEvent(n->comm_stream, n->join_ev2, 0));
    }
    if (bucketing) {
      // (the hook takes an int count: buckets are far below 2^31 floats)
      for (long o = lo; o < grad_hi; o += (1L << 30)) {
        const long cnt = grad_hi - o < (1L << 30) ? grad_hi - o : (1L << 30);
        const int rc = n->grad_sync(n->grad_user, grads + o, (int)cnt, n->comm_stream);
        if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_backward: gradient all-reduce hook failed (%d)", rc);
      }
      const int rc2 = pxl_scale_inplace(grad_hi - lo, grads + lo, 1.0f / (float)n->grad_world, n->comm_stream);
      if (rc2 != PXL_OK) return rc2;
      ++n->grad_buckets_last;
    }
    if (updating) {           // the optimizer takes the finished (and averaged) bucket from here, on the same stream
      const int rc = n->update_fn(n->update_user, lo, grad_hi, n->comm_stream);
      if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_backward: parameter-update hook failed (%d)", rc);
      ++n->update_buckets_last;
    }
    grad_hi = lo;
    return PXL_OK;
  };

  std::vector<int> pending_w;        // convolutions whose dy is final on the main stream, weight gradient not yet issued
  std::vector<size_t> wsrc(n->ops.size(), 0);     // dy location of a queued convolution without BN (may be an alias)
  auto issue_wgrads = [&](int at_op) -> int {
    if (pending_w.empty()) return PXL_OK;
    hipStream_t ws = s;
    if (n->use_side) {                                     // fork: every queued dy is final on the main stream from here on
      if (!n->fork_ev[at_op]) PXL_CHECK_HIP(hipEventCreateWithFlags(&n->fork_ev[at_op], hipEventDisableTiming));
      PXL_CHECK_HIP(hipEventRecord(n->fork_ev[at_op], s));
      ws = (n->side2 != nullptr && !bucketing && (n->fork_count++ & 1)) ? n->side2 : n->side;
      PXL_CHECK_HIP(hipStreamWaitEvent(ws, n->fork_ev[at_op], 0));
      if (ws == n->side2) forked2 = true; else forked = true;
    }
    for (int k : pending_w) {
      OpInfo& opk = n->ops[k];
      const pxl_op& dk = opk.d;
      const TensorInfo& tik = n->tensors[dk.in0];
      const TensorInfo& tok = n->tensors[dk.out];
      const int Mk = n->B * tok.H * tok.W;
      const void* dyk = at(scratch, dk.bn_out >= 0 ? tok.goff : wsrc[k]);
      const ConvIn cin = conv_input(n, opk, arena);
      if (opk.pg) {
        // ONE GEMM over the gathered dP (written on the main stream before the fork), then the groups' rows added into the master
        // gradient layout
        int rc;
        float* tmp = fat(scratch, opk.pg_dw_off);
        PXL_CHECK_HIP(hipMemsetAsync(tmp, 0, (size_t)opk.pg_J * dk.cin * 4, ws));
        {
          Timed t(n, ws, 1, conv_flops(n, dk, tok));
          if (n->profile) n->prof_bytes[1] += conv_bytes(n, dk, tik, tok, true);
          pxl_conv_desc q = opk.pg_grp;
          if (n->deterministic) q.split_k = 1;
          rc = pxl_conv_wgrad(&q, cin.ptr, nullptr, nullptr, at(scratch, opk.pg_dp_off), tmp, dk.cin, dk.cin, ws);
        }
        if (rc != PXL_OK) return rc;
        long woffs[4] = {0, 0, 0, 0};
        for (int g = 0; g < dk.ngroups; ++g) woffs[g] = dk.w_off[g];
        rc = pxl_aspp_dw_scatter(tmp, dk.ngroups, opk.pg_GP, dk.cout, dk.kh * dk.kw, dk.cin, dk.cin, grads, woffs, ws);
        if (rc != PXL_OK) return rc;
        for (int g = 0; g < dk.ngroups; ++g) {
          if (dk.b_off[g] < 0) continue;
          rc = n->deterministic ? pxl_colsum_ordered(dt, Mk, tok.Cp, dk.cout, dyk, grads + dk.b_off[g], ws)
                                : pxl_colsum(dt, Mk, tok.Cp, dk.cout, dyk, grads + dk.b_off[g], ws);
          if (rc != PXL_OK) return rc;
        }
        continue;
      }
      for (int g = 0; g < dk.ngroups; ++g) {
        int rc;
        {
          Timed t(n, ws, 1, conv_flops(n, dk, tok) / dk.ngroups);
          if (n->profile) n->prof_bytes[1] += conv_bytes(n, dk, tik, tok, true) / dk.ngroups;
          const int creal = opk.patch ? opk.patch_K : dk.cin;       // (patch mode: dw[Cout][1][kh*kw*C] = the master layout)
          rc = pxl_conv_wgrad(&opk.grp[g], cin.ptr, cin.sc, cin.sh, dyk, grads + dk.w_off[g], creal, creal, ws);
        }
        if (rc != PXL_OK) return rc;
        if (dk.b_off[g] >= 0) {
          rc = n->deterministic ? pxl_colsum_ordered(dt, Mk, tok.Cp, dk.cout, dyk, grads + dk.b_off[g], ws)
                                : pxl_colsum(dt, Mk, tok.Cp, dk.cout, dyk, grads + dk.b_off[g], ws);
          if (rc != PXL_OK) return rc;
        }
      }
    }
    pending_w.clear();
    return PXL_OK;
  };
  for (int i = (int)n->ops.size() - 1; i >= 0; --i) {
    OpInfo& op = n->ops[i];
    const pxl_op& d = op.d;
    int rc = PXL_OK;
    if (d.kind != PXL_OP_CONV && d.kind != PXL_OP_RESIDUAL && d.kind != PXL_OP_INPUT) {
      for (int t : {d.in0, d.in1, d.out}) { rc = settle(t); if (rc != PXL_OK) return rc; }
    }
    switch (d.kind) {
      case PXL_OP_HEAD: {
        if (from_low) { written[d.in0] = 1; break; }       // d(low) was written by pxl_net_head_loss
        const TensorInfo& low = n->tensors[d.in0];
        rc = pxl_upsample_softmax_bwd(dt, n->B, low.H, low.W, low.Cp, n->classes, n->Ho, n->Wo, d.stride, dlogits, dprob, prob,
                                      at(scratch, low.goff), at(scratch, n->up_ws_off), n->up_ws_bytes, stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_PIXSHUF: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: PixelShuffle op %d output has no gradient", i);
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: PixelShuffle input consumed twice");
        rc = pxl_pixshuf_relu_bwd(dt, n->B, tin.H, tin.W, tin.Cp, tout.C, at(scratch, tout.goff), tout.Cp, at(arena, tin.off),
                                  at(scratch, tin.goff), stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_UPCAT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: concat tensor of op %d has no gradient", i);
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: pyramid stage output consumed twice");
        rc = pxl_upsample_slice_bwd(dt, n->B, tin.H, tin.W, tin.Cp, tin.C, at(scratch, tout.goff), tout.H, tout.W, tout.Cp, d.c_off,
                                    at(scratch, tin.goff), stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_CONCAT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: concat op %d output has no gradient", i);
        rc = pxl_slice_copy(dt, (long)n->B * tin.H * tin.W, tin.C, at(scratch, tout.goff), tout.Cp, 0, at(scratch, tin.goff), tin.Cp,
                            0, written[d.in0] ? 1 : 0, stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_AVGPOOL: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: pooling op %d output has no gradient", i);
        rc = pxl_adaptive_avgpool_bwd(dt, n->B, tin.H, tin.W, tin.Cp, d.kh, at(scratch, tout.goff), at(scratch, tin.goff),
                                      written[d.in0] ? 1 : 0, stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_ACT: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: activation op %d output has no gradient", i);
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: activation input consumed twice");
        // with a BN in front the gradient written here is d(bn(y)): the producing convolution's backward applies the BN
        rc = pxl_leaky_bwd(dt, (long)n->B * tin.H * tin.W * tin.Cp, at(scratch, tout.goff),
                           d.bn_in0 >= 0 ? at(arena, op.ws_off) : at(arena, tin.off), d.slope, at(scratch, tin.goff), stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_IBN: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: IBNorm op %d output has no gradient", i);
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: IBNorm input consumed twice");
        const BnInfo& b = n->bns[d.bn_out];
        const int nb = b.d.C, HW = tin.H * tin.W;
        rc = pxl_ibn_bwd_reduce_acc(dt, n->B, HW, tin.Cp, at(scratch, tout.goff), at(arena, tin.off), fat(arena, op.ibn_coef),
                                    d.slope, fat(scratch, op.ibn_bsums), stream);
        if (rc != PXL_OK) return rc;
        // BN half: fold over the samples; affine gradients from the LOCAL sums, batch-mean terms from the all-reduced
        rc = pxl_ibn_fold(n->B, tin.Cp, nb, fat(scratch, op.ibn_bsums), fat(scratch, op.ibn_bbn),
                          n->wgrad_on ? grads + b.d.gamma_off : nullptr, n->wgrad_on ? grads + b.d.beta_off : nullptr, stream);
        if (rc != PXL_OK) return rc;
        if (training && n->sync && n->world > 1) {
          rc = n->sync(n->sync_user, fat(scratch, op.ibn_bbn), 2 * nb, stream);
          if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_backward: SyncBN all-reduce hook failed (%d)", rc);
        }
        rc = pxl_ibn_bwd_apply(dt, n->B, HW, tin.Cp, nb, at(scratch, tout.goff), at(arena, tin.off), fat(arena, op.ibn_coef),
                               fat(scratch, op.ibn_bsums), fat(scratch, op.ibn_bbn), (float)n->B * HW * n->world, training,
                               d.slope, at(scratch, tin.goff), stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_RESIDUAL: {
        const TensorInfo& o = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: residual op %d output has no gradient", i);
        const TensorInfo& a = n->tensors[d.in0];
        const TensorInfo& r = n->tensors[d.in1];
        const long nelem = (long)n->B * o.H * o.W * o.Cp;
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: residual main branch consumed twice");
        const BnInfo& b3 = n->bns[d.bn_in0];
        if (join_done[i]) {
          // o's gradient buffer already holds the masked gradient and b3's sums are complete (the consumer's data
          // gradient did it): both branches read it from there
          gsrc[d.in0] = o.goff;
          if (!written[d.in1]) {
            gsrc[d.in1] = o.goff;
            written[d.in1] = 1;
          } else {
            rc = settle(d.in1);
            if (rc != PXL_OK) return rc;
            rc = pxl_add_inplace(dt, nelem, at(scratch, r.goff), at(scratch, o.goff), stream);
          }
          reduced[d.bn_in0] = 1;
          written[d.in0] = 1;
          break;
        }
        rc = settle(d.out);
        if (rc != PXL_OK) return rc;
        rc = settle(d.in1);
        if (rc != PXL_OK) return rc;
        if (!written[d.in1] && n->fuse_bn_reduce && b3.y_tensor == d.in0 && a.Cp == b3.d.C) {
          // relu mask + the main branch BN's backward sums in one pass
          rc = pxl_residual_bwd_reduce_rep(dt, n->B * a.H * a.W, a.Cp, at(scratch, o.goff), at(arena, o.off), at(arena, a.off),
                                           fat(arena, b3.coef_off), at(scratch, a.goff), at(scratch, r.goff),
                                           fat(scratch, b3.bsum_off), b3.bnrep, stream);
          reduced[d.bn_in0] = 1;
          written[d.in1] = 1;
        } else if (!written[d.in1]) {
          rc = pxl_relu_mask(dt, nelem, at(scratch, o.goff), at(arena, o.off), at(scratch, a.goff), at(scratch, r.goff), stream);
          written[d.in1] = 1;
        } else {
          rc = pxl_relu_mask(dt, nelem, at(scratch, o.goff), at(arena, o.off), at(scratch, a.goff), nullptr, stream);
          if (rc != PXL_OK) return rc;
          rc = pxl_add_inplace(dt, nelem, at(scratch, r.goff), at(scratch, a.goff), stream);
        }
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_MAXPOOL: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: maxpool op %d output has no gradient", i);
        if (written[d.in0]) return pxl_set_error(PXL_ERR_UNSUPPORTED, "net_backward: maxpool input consumed twice");
        rc = pxl_maxpool3x3s2_bwd(dt, n->B, tin.H, tin.W, tin.Cp, at(scratch, tout.goff), at(arena, op.idx_off),
                                  at(scratch, tin.goff), stream);
        written[d.in0] = 1;
        break;
      }
      case PXL_OP_CONV: {
        const TensorInfo& tin = n->tensors[d.in0];
        const TensorInfo& tout = n->tensors[d.out];
        if (!written[d.out]) return pxl_set_error(PXL_ERR_ARG, "net_backward: conv op %d output has no gradient", i);
        const int M = n->B * tout.H * tout.W;
        void* dy = at(scratch, tout.goff);
        const void* dy_in = at(scratch, gsrc[d.out]);      // (a fused residual join leaves it in the join output's buffer)
        if (d.bn_out < 0) dy = const_cast<void*>(dy_in);
        wsrc[i] = gsrc[d.out];
        if (d.bn_out >= 0) {
          BnInfo& b = n->bns[d.bn_out];
          const float* coef = fat(arena, b.coef_off);
          // one [2C] vector per BN (the reduce kernel issues one atomic per channel per block, no replicas needed);
          // already filled when the consumer's data gradient ran with the fused reduction
          if (b.fused_reduce_op < 0 && !reduced[d.bn_out]) {
            rc = pxl_bn_bwd_reduce(dt, M, tout.Cp, dy_in, at(arena, tout.off), coef, b.relu, fat(scratch, b.bsum_off), b.bnrep, stream);
            if (rc != PXL_OK) return rc;
          }
          if (b.bnrep > 1) {      // PXL_DETERMINISTIC: the replicas of the sums, folded in index order into replica 0
            rc = pxl_bn_fold_replicas(2 * b.d.C, b.bnrep, fat(scratch, b.bsum_off), stream);
            if (rc != PXL_OK) return rc;
          }
          float* dgam = grads + b.d.gamma_off;
          float* dbet = grads + b.d.beta_off;
          if (training && n->sync && n->world > 1) {
            // affine gradients from the LOCAL sums (the gradient all-reduce averages them over the ranks), then
            // the batch-mean terms of dy from the all-reduced sums
            if (n->sync == &pxl_peer_allreduce_hook) {       // both in one launch on the peer-mapped path
              rc = pxl_peer_allreduce_bnbwd(reinterpret_cast<pxl_peer*>(n->sync_user), fat(scratch, b.bsum_off), b.d.C, dgam, dbet, stream);
              if (rc != PXL_OK) return rc;
            } else {
              rc = pxl_bn_param_grad(b.d.C, fat(scratch, b.bsum_off), dgam, dbet, stream);
              if (rc != PXL_OK) return rc;
              rc = n->sync(n->sync_user, fat(scratch, b.bsum_off), 2 * b.d.C, stream);
              if (rc != 0) return pxl_set_error(PXL_ERR_HIP, "net_backward: SyncBN all-reduce hook failed (%d)", rc);
            }
            dgam = dbet = nullptr;
          }
          rc = pxl_bn_bwd_apply_fused(dt, M, tout.Cp, dy_in, at(arena, tout.off), coef, fat(scratch, b.bsum_off),
                                      (float)b.M * n->world, training, b.relu, dgam, dbet, dy, stream);
          if (rc != PXL_OK) return rc;
        }
        const ConvIn cin = conv_input(n, op, arena);
        const float* sc = cin.sc; const float* sh = cin.sh;
        // weight gradients: queued, and issued on the side stream in groups of `fork_every` convolutions (PXL_FORK_EVERY,
        // default 1).  An event record between two kernels of the main stream costs a 6-7 us bubble there, and dy buffers
        // and activations stay valid until the end of the pass, so the weight gradients could start later in larger
        // groups -- measured (MT 8x513x513, 20 steps): 1 -> 14.04 ms, 3 -> 14.2, 6 -> 14.2: the earlier start of the
        // weight gradients is worth more than the saved bubbles
        (void)sc; (void)sh;
        if (op.pg && (n->wgrad_on || d.need_dgrad)) {
          // dP[(b, y', x')][(g, c, t)] = dOut[b, y' - dy_t, x' - dx_t, c]: gathered ONCE on the main stream (before the weight
          // gradient's fork event), both GEMMs of the head's backward read it
          rc = pxl_aspp_dp_gather(dt, n->B, tout.H, tout.W, op.pg_J, op.pg_GP, d.ngroups, d.cout, d.kh * d.kw, op.fwd.dy, op.fwd.dx, dy,
                                  tout.Cp, at(scratch, op.pg_dp_off), stream);
          if (rc != PXL_OK) return rc;
        }
        if (n->wgrad_on) {
          pending_w.push_back(i);
          const bool now = !n->use_side || !d.need_dgrad || (int)pending_w.size() >= (n->fork_every < 1 ? 1 : n->fork_every);
          if (now) { rc = issue_wgrads(i); if (rc != PXL_OK) return rc; }
        }
        if (d.need_dgrad) {
          void* din = at(scratch, tin.goff);
          Timed t(n, s, 0, conv_flops(n, d, tout));
          if (n->profile) n->prof_bytes[0] += conv_bytes(n, d, tin, tout, false);
          const void* addend = written[d.in0] ? at(scratch, gsrc[d.in0]) : nullptr;
          // (the multi-rate head: data gradient = the GEMM J -> Cin over dP)
          const pxl_conv_desc* bwd = op.pg ? &op.pg_bwd : &op.bwd;
          const void* bdy = op.pg ? at(scratch, op.pg_dp_off) : dy;
          const void* bwt = op.pg ? at(packed, op.pg_wt_off) : at(packed, op.wt_off);
          if (op.join_op >= 0) {
            const pxl_op& dj = n->ops[op.join_op].d;
            const BnInfo& b3 = n->bns[dj.bn_in0];
            const OpInfo& oj = n->ops[op.join_op];
            if (oj.bits)
              rc = pxl_conv_dgrad_joinreduce_bits(bwd, bdy, bwt, din, addend, at(arena, oj.bits_off),
                                                  at(arena, n->tensors[dj.in0].off), fat(arena, b3.coef_off),
                                                  fat(scratch, b3.bsum_off), stream);
            else
              rc = pxl_conv_dgrad_joinreduce(bwd, bdy, bwt, din, addend, at(arena, tin.off),
                                             at(arena, n->tensors[dj.in0].off), fat(arena, b3.coef_off),
                                             fat(scratch, b3.bsum_off), stream);
            join_done[op.join_op] = 1;
          } else if (d.bn_in0 >= 0 && n->bns[d.bn_in0].fused_reduce_op == i) {
            const BnInfo& bi = n->bns[d.bn_in0];
            rc = pxl_conv_dgrad_bnreduce(bwd, bdy, bwt, din, addend,
                                         at(arena, tin.off), fat(arena, bi.coef_off), bi.relu, fat(scratch, bi.bsum_off), stream);
          } else {
            rc = pxl_conv_igemm(bwd, bdy, bwt, din, nullptr, nullptr, nullptr,
                                addend, nullptr, nullptr, 0, stream);
          }
          gsrc[d.in0] = tin.goff;
          written[d.in0] = 1;
        }
        break;
      }
      case PXL_OP_INPUT:
        break;
    }
    if (rc != PXL_OK) return rc;
    // everything that writes gradients at or above op_lo[i] has now been issued
    {
      const long bsz = bucketing ? n->grad_bucket : n->update_bucket;
      bool cut = (bucketing || updating) && n->bucket_ok && bsz > 0 && n->op_lo[i] >= 0 && grad_hi - n->op_lo[i] >= bsz;
      // pipelined update: one more boundary just above the first layers (the stem), so that the LAST bucket -- the one whose
      // update the next forward pass has to wait for -- is a few thousand parameters
      if (!cut && updating && n->update_tail > 0 && n->op_lo[i] >= 0 && n->op_lo[i] <= n->update_tail && grad_hi > n->update_tail)
        cut = true;
      if (cut) {
        rc = issue_wgrads(i);
        if (rc != PXL_OK) return rc;
        rc = flush(n->op_lo[i]);
        if (rc != PXL_OK) return rc;
      }
    }
  }
  {
    int rc = issue_wgrads(0);                    // (only when the program does not end in a convolution without data gradient)
    if (rc != PXL_OK) return rc;
  }
  if (bucketing || updating) {                   // the rest of the buffer (or all of it when bucketing is off)
    int rc = flush(0);
    if (rc != PXL_OK) return rc;
    PXL_CHECK_HIP(hipEventRecord(n->comm_done_ev, n->comm_stream));
    PXL_CHECK_HIP(hipStreamWaitEvent(s, n->comm_done_ev, 0));
  }
  if (forked) {                                  // join: every weight gradient is complete before the caller goes on
    PXL_CHECK_HIP(hipEventRecord(n->join_ev, n->side));
    PXL_CHECK_HIP(hipStreamWaitEvent(s, n->join_ev, 0));
  }
  if (forked2) {
    PXL_CHECK_HIP(hipEventRecord(n->join_ev2, n->side2));
    PXL_CHECK_HIP(hipStreamWaitEvent(s, n->join_ev2, 0));
  }
  return PXL_OK;
}
}  // namespace
