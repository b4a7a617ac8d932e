"""One process per GPU: the replacement of the reference's single-process nn.DataParallel
(pixelssl/nn/func.py:54-62) and thread-based Sync-BN (sync_batchnorm/comm.py).

Three exchanges.  With the nccl backend every rank opens its own RCCL communicators from C (csrc/comm.cpp, one per
engine network in rotation + one for gradients) and the executor issues `ncclAllReduce` itself; with gloo (CPU-launched
tests) the same hooks call torch.distributed:
  * gradients: the flat fp32 gradient buffer of a model is all-reduced (sum, then 1/world) in BUCKETS of contiguous
    memory from INSIDE the backward pass, on a communication stream, as soon as every kernel writing into a bucket has
    been issued (csrc/net.cpp: pxl_net_set_grad_sync; PXL_GRAD_BUCKET_MB, default 32) -- equal per-rank batches make the
    mean of rank means the global mean; PXL_GRAD_OVERLAP=0 falls back to ONE all-reduce after the
    backward; parameters that do not belong to an engine network are averaged by the optimizers (nn/optimizer.py);
  * Sync-BN statistics: all-reduce(sum) of the [sum, sumsq] (forward) / [sum dz, sum dz*xhat] (backward) vectors
    between a convolution's statistics epilogue and the BN finalize, one call per BatchNorm and direction
    (~310 per MT step), with the reference's multi-device variance formula clamp(var, eps)^-1/2;
  * scalars for logging.
No parameter broadcast per forward, no scatter/gather of activations (C1-C3 are gone).
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def local_device():
    """This rank's GPU: LOCAL_RANK, or PXL_FORCE_DEVICE (tests run two gloo ranks on one GPU)."""
    forced = os.environ.get('PXL_FORCE_DEVICE')
    if forced == 'cpu':         # structure-only use (parameter trees, checkpoints): nothing can be executed there
        return torch.device('cpu')
    return torch.device('cuda', int(forced if forced is not None else os.environ.get('LOCAL_RANK', '0')))


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op for WORLD_SIZE=1)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return
    backend = os.environ.get('PXL_DIST_BACKEND', backend)      # tests: two gloo ranks on one GPU
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if torch.cuda.is_available() and local_device().type == 'cuda':
        torch.cuda.set_device(local_device())       # inputs (_to_device) and statistic views follow the current device
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend=backend)


def allreduce_mean_(flat):
    """In-place average of a flat tensor over all ranks (gradient exchange)."""
    if not is_distributed():
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / world_size())
    return flat


def allreduce_sum_(t):
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def shard_batch_sizes(labeled_batch_size, unlabeled_batch_size, n_ranks):
    """Per-rank (labeled, unlabeled) counts: the reference scales the script's per-GPU sizes by #GPUs
    (task_template/proxy.py:258-261); per rank we keep the per-GPU sizes, labeled first."""
    if labeled_batch_size % n_ranks or unlabeled_batch_size % n_ranks:
        raise ValueError('global batch (%d+%d) is not divisible by %d ranks'
                         % (labeled_batch_size, unlabeled_batch_size, n_ranks))
    return labeled_batch_size // n_ranks, unlabeled_batch_size // n_ranks


class _DevView:
    """Wrap a raw device pointer as a tensor via __cuda_array_interface__ (no copy)."""

    def __init__(self, p, n):
        self.__cuda_array_interface__ = {'data': (int(p), False), 'shape': (int(n),), 'typestr': '<f4',
                                         'version': 2, 'strides': None}


_views = {}


def _sync_stats_callback(buf_ptr, n, stream):
    """all-reduce(sum) of `n` floats at device pointer `buf_ptr` (the executor's Sync-BN hook).  Called ~200 times
    per network pass, so the zero-copy tensor views are cached per (pointer, length): the caching allocator hands
    the executor the same arena addresses every iteration."""
    key = (int(buf_ptr), int(n))
    t = _views.get(key)
    if t is None:
        if len(_views) > 8192:
            _views.clear()
        t = torch.as_tensor(_DevView(buf_ptr, n), device=torch.device('cuda', torch.cuda.current_device()))
        _views[key] = t
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return 0


_native = {"tried": False, "comms": [], "hook": None, "next": 0, "grad_comm": None}
# floats per gradient bucket: 8 M floats = 32 MB (DeepLab-v2's 176 MB gradient = 6 buckets, issued in reverse layer order
# while the backward pass is still running).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of
# 32 MB over 8 GPUs moves 2 * 7/8 * 32 MB per link, ~0.4 ms at link speed -- large enough to be bandwidth-bound, small
# enough that the last bucket (the stem's) is the only one the optimizer step waits for.
GRAD_BUCKET_FLOATS = int(float(os.environ.get("PXL_GRAD_BUCKET_MB", "32")) * (1 << 20) / 4)
N_COMMS = max(1, int(os.environ.get("PXL_N_COMMS", "2")))
#               # RCCL executes the collectives of ONE communicator in issue order even across streams: networks that run
#                 concurrently on two streams (MT student / teacher, GCT l / r model) get different communicators


def _open_comm(h, dev):
    import ctypes
    from . import _lib
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank() == 0:
        _lib.check(h.pxl_comm_unique_id(ident.data_ptr()))
    ident_d = ident.to(dev)
    dist.broadcast(ident_d, 0)
    ident = ident_d.cpu()
    comm = ctypes.c_void_p()
    rc = h.pxl_comm_init(ident.data_ptr(), rank(), world_size(), ctypes.byref(comm))
    good = rc == 0
    if good:                                                  # one verified exchange before anything depends on it
        probe = torch.full((64,), float(rank() + 1), device=dev)
        rc = h.pxl_comm_allreduce_sum(comm, probe.data_ptr(), 64, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ws = world_size()
        good = rc == 0 and bool((probe == ws * (ws + 1) / 2).all().item())
    return comm, good


def native_comms():
    """The process's RCCL communicators driven from C (csrc/comm.cpp), or []: only with the nccl backend (one GPU per
    rank), agreed on by ALL ranks, verified against a known sum once, and PXL_NATIVE_RCCL=0 disables them."""
    if _native["tried"]:
        return _native["comms"]
    _native["tried"] = True
    if not is_distributed() or dist.get_backend() != "nccl" or os.environ.get("PXL_NATIVE_RCCL", "1") == "0":
        return []
    import ctypes
    from . import _lib
    h = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    ok = torch.tensor([1.0 if h.pxl_comm_available() else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)              # every rank must be able to load librccl
    if ok.item() < 1:
        return []
    comms, good = [], True
    for _ in range(N_COMMS):
        c, g = _open_comm(h, dev)
        comms.append(c)
        good = good and g
    ok = torch.tensor([1.0 if good else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() < 1:
        for c in comms:
            if c.value:
                h.pxl_comm_destroy(c)
        return []
    # one more communicator for the gradient buckets: RCCL runs the collectives of ONE communicator in issue order even
    # across streams, and a bucket waiting for its weight gradients must not block the Sync-BN exchanges behind it
    gc, g = _open_comm(h, dev)
    ok = torch.tensor([1.0 if g else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    _native["grad_comm"] = gc if ok.item() >= 1 else None
    _native["comms"] = comms
    _native["hook"] = ctypes.cast(h.pxl_comm_allreduce_hook, _lib.ALLREDUCE_FN)
    return comms


_peer = {"ok": None, "ctxs": [], "hook": None}
PEER_SLOT_FLOATS = 8192            # 2 networks x 2 * 2048 channels: the widest BatchNorm of a student || teacher pair in one exchange
PEER_TIMEOUT_MS = int(os.environ.get("PXL_PEER_TIMEOUT_MS", "20000"))      # one exchange is ~6 us; 20 s is a dead (or wedged) peer,
# well above ordinary rank skew (a per-rank autotune, a rank-0 checkpoint save or validation pass).  A time-out is sticky and costs its
# 20 s once; the exchange that times out returns NaN sums (csrc/peer.hip), never stale ones.  The FIRST exchanges of a context
# (PXL_PEER_WARM_EXCHANGES = 1024, about three Mean-Teacher steps) wait PXL_PEER_WARM_SCALE = 15 times as long: start-up skew between
# ranks (uneven autotune of several networks, first-touch page-ins) is not a dead peer and there is no hidden barrier to absorb it.
# What a time-out does: by default the run ABORTS at the next optimizer step (poll_peers reads the status word from mapped host
# memory every step, no device sync) -- at most one update on NaN statistics, and that one is detectably invalid.
# PXL_PEER_FALLBACK=1 (opt-in): every PEER_POLL_STEPS-th step the ranks agree on the status and move the statistics to RCCL /
# torch.distributed for the rest of the run instead; check_peers() at the epoch end still reports that it happened.
PEER_POLL_STEPS = int(os.environ.get("PXL_PEER_POLL_STEPS", "20"))
PEER_FALLBACK = os.environ.get("PXL_PEER_FALLBACK", "0") == "1"


def _all_agree(flag, dev):
    ok = torch.tensor([1.0 if flag else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return ok.item() >= 1


def open_peer_context():
    """One peer-mapped exchange context (csrc/peer.hip) or None.  COLLECTIVE: every rank calls it the same number of
    times in the same order.  The buffers are shared through HIP IPC handles gathered with torch.distributed, and a context
    is only handed out after 64 exchanges with known sums came out right on EVERY rank; PXL_PEER_SYNC=0 turns the path
    off (the statistics then go through RCCL / torch.distributed)."""
    if not is_distributed() or not torch.cuda.is_available() or os.environ.get("PXL_PEER_SYNC", "1") == "0":
        return None
    if _peer["ok"] is False:
        return None
    import ctypes
    from . import _lib
    h = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    ws, rk = world_size(), rank()
    ctx = ctypes.c_void_p()
    good = h.pxl_peer_create(rk, ws, PEER_SLOT_FLOATS, PEER_TIMEOUT_MS, ctypes.byref(ctx)) == 0
    handle = ctypes.create_string_buffer(64)
    good = good and h.pxl_peer_handle(ctx, handle) == 0
    gathered = [None] * ws
    dist.all_gather_object(gathered, handle.raw if good else None)
    good = good and all(g is not None for g in gathered)
    if good:
        good = h.pxl_peer_open(ctx, b"".join(gathered)) == 0
    if _all_agree(good, dev):
        # verified before anything depends on it: vectors of several lengths, every rank's contribution distinct
        stream = torch.cuda.current_stream().cuda_stream
        probe_ok = True
        for k in range(64):
            # (every rank issues ALL 64 exchanges whatever it has seen so far: a rank that stopped at its first mismatch
            # would leave the others spinning in the exchanges it skipped, and the call sequences would diverge)
            n = (64, 256, 1024, PEER_SLOT_FLOATS, PEER_SLOT_FLOATS + 192)[k % 5]
            v = torch.arange(n, device=dev, dtype=torch.float32) * (rk + 1) + k
            want = torch.arange(n, device=dev, dtype=torch.float32) * (ws * (ws + 1) / 2) + k * ws
            issued = h.pxl_peer_allreduce_sum(ctx, v.data_ptr(), n, stream) == 0
            same = bool(torch.equal(v, want))
            probe_ok = probe_ok and issued and same
        st = ctypes.c_int(0)
        probe_ok = probe_ok and h.pxl_peer_status(ctx, ctypes.byref(st)) == 0 and st.value == 0
        good = _all_agree(probe_ok, dev)
    else:
        good = False
    if not good:
        if ctx.value:
            h.pxl_peer_destroy(ctx)
        _peer["ok"] = False
        return None
    _peer["ok"] = True
    _peer["ctxs"].append(ctx)
    if _peer["hook"] is None:
        _peer["hook"] = ctypes.cast(h.pxl_peer_allreduce_hook, _lib.ALLREDUCE_FN)
    return ctx


def align_after_tune():
    """Optional host barrier after a per-rank autotune (engine: pxl_net_tune / pxl_net_tune_pair time ~3000 candidate launches, a few
    seconds whose length differs from rank to rank).  OFF by default since round 5: it is a collective hidden inside a forward pass,
    which deadlocks when only some ranks meet an untuned shape (a rank-0-only pass, an uneven last batch, a rank-dependent branch of
    a plug-in model).  Without it the first Sync-BN exchange of the faster rank simply spins for the difference -- seconds, against
    an exchange time-out of 20 s (PXL_PEER_TIMEOUT_MS).  PXL_TUNE_BARRIER=1 restores it for jobs whose ranks provably plan and tune
    the same shapes in the same order."""
    if is_distributed() and os.environ.get("PXL_TUNE_BARRIER", "0") == "1":
        dist.barrier()


def epoch_barrier():
    """Host barrier at an epoch boundary (ssl_base.train): after it the ranks start their first Sync-BN exchange within
    microseconds of each other, whatever rank 0 did in between (checkpoint, validation, logging)."""
    if is_distributed() and os.environ.get("PXL_EPOCH_BARRIER", "1") != "0":
        dist.barrier()


def peer_contexts():
    """Number of peer-mapped exchange contexts in use (0 = Sync-BN goes through RCCL / torch.distributed)."""
    return len(_peer["ctxs"])


def _peer_error(status):
    from . import _lib
    return _lib.PixelHipError("peer-mapped Sync-BN exchange: rank %d gave up waiting for rank %d after %d ms; the statistics of "
                              "that exchange (and of every later one on the context) are NaN" % (rank(), status - 1, PEER_TIMEOUT_MS))


def check_peers():
    """Raise if any peer-mapped exchange of this process timed out (its sums were invalid) -- including one that a
    PXL_PEER_FALLBACK=1 run has already recovered from by moving to RCCL: the steps between the time-out and the fall-back
    trained on invalid statistics, and the epoch-end guard of ssl_base.train() has to say so.  Synchronises the device: call it at
    a checkpoint / epoch boundary, not per step."""
    import ctypes
    from . import _lib
    if _peer.get("timed_out") is not None:
        raise _peer_error(_peer["timed_out"])
    for ctx in _peer["ctxs"]:
        st = ctypes.c_int(0)
        _lib.check(_lib.lib().pxl_peer_status(ctx, ctypes.byref(st)))
        if st.value:
            _peer["timed_out"] = st.value
            raise _peer_error(st.value)


_poll = {"calls": 0, "fallbacks": 0}


def _local_peer_status(sync=False):
    """worst status word of this process's contexts; without `sync` read from the mapped host mirror (no device sync)"""
    import ctypes
    from . import _lib
    h = _lib.lib()
    worst = 0
    for ctx in _peer["ctxs"]:
        v = -1 if sync else h.pxl_peer_status_nosync(ctx)
        if v < 0:
            st = ctypes.c_int(0)
            _lib.check(h.pxl_peer_status(ctx, ctypes.byref(st)))
            v = st.value
        worst = max(worst, v)
    return worst


def poll_peers(cores=None):
    """Called once per optimizer step.  Every call reads this process's exchange status words from mapped host memory (free: no
    device sync, no collective); a time-out raises PixelHipError -- the run stops within one optimizer step of the first NaN
    statistics instead of training on them (the peers time out on this rank's missing words and stop the same way).
    PXL_PEER_FALLBACK=1 (opt-in) replaces the abort by the round-4 behaviour: every PEER_POLL_STEPS-th call is COLLECTIVE, the ranks
    agree (MAX) on whether any exchange has timed out, and if one has, EVERY rank moves the Sync-BN statistics of every network to
    the RCCL / torch.distributed path for the rest of the run; the time-out stays on record (check_peers raises at the epoch end).
    -> True when a fall-back happened in this call."""
    if not _peer["ctxs"] or not is_distributed():
        return False
    _poll["calls"] += 1
    if not PEER_FALLBACK:
        worst = _local_peer_status()
        if worst:
            _peer["timed_out"] = worst
            raise _peer_error(worst)
        return False
    if _poll["calls"] % PEER_POLL_STEPS:
        return False
    from . import _lib
    worst = _local_peer_status(sync=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    flag = torch.tensor([float(worst)], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if flag.item() == 0:
        return False
    from .utils import logger
    logger.log_warn("peer-mapped Sync-BN exchange timed out (rank %d: local status %d); the statistics move to %s for the rest "
                    "of the run; up to %d optimizer steps ran on NaN statistics\n"
                    % (rank(), worst, "RCCL" if _native["comms"] else "torch.distributed", PEER_POLL_STEPS))
    _peer["timed_out"] = int(flag.item())
    retired, _peer["ctxs"] = _peer["ctxs"], []
    _peer["ok"] = False
    ws = world_size()
    for m in _peer.get("cores", []):
        if getattr(m, "_pxl_peer", None) is None:
            continue
        m._pxl_peer = None
        comm = getattr(m, "_pxl_comm", None)
        if comm is not None:
            m.set_sync_native(_native["hook"], comm, ws)
        else:
            m.set_sync(_sync_stats_callback, ws)
    torch.cuda.synchronize()
    for ctx in retired:
        _lib.lib().pxl_peer_destroy(ctx)
    _poll["fallbacks"] += 1
    return True


def peer_exchanges():
    """Exchanges issued so far on this process's peer-mapped contexts (bench.py: exchanges per step)."""
    from . import _lib
    h = _lib.lib()
    return sum(int(h.pxl_peer_exchanges(ctx)) for ctx in _peer["ctxs"])


def rccl_ranks():
    """Ranks of the C-driven RCCL communicators (0 = the exchanges go through torch.distributed)."""
    return world_size() if _native["comms"] else 0


def _grad_sync_callback(buf_ptr, n, stream):
    """torch.distributed path of the bucketed gradient exchange (gloo in the CPU-launched tests): all-reduce(sum) of the
    bucket on the executor's communication stream."""
    t = torch.as_tensor(_DevView(buf_ptr, n), device=torch.device('cuda', torch.cuda.current_device()))
    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return 0


def _post_backward(core):
    if getattr(core, "_grad_sync", None) is not None:
        return                              # the executor exchanged the buckets from inside pxl_net_backward
    comm = getattr(core, "_pxl_comm", None)
    if comm is not None:
        from . import _lib
        g = core.flat.grads
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().pxl_comm_allreduce_sum(comm, g.data_ptr(), g.numel(), s))
        _lib.check(_lib.lib().pxl_scale_inplace(g.numel(), g.data_ptr(), 1.0 / world_size(), s))
        return
    allreduce_mean_(core.flat.grads)


def attach(model):
    """Wire every engine network inside `model` for multi-rank training (no-op on one rank).  Sync-BN statistics: the
    peer-mapped one-shot exchange (csrc/peer.hip, one context per network) when the ranks can map each other's device
    memory, else the C-driven RCCL communicator (nccl backend), else torch.distributed (gloo in the CPU tests).  Flat-gradient
    all-reduce: RCCL from C when available, else torch.distributed."""
    if not is_distributed():
        return model
    from .engine import SegNetCore
    ws = world_size()
    comms = native_comms()
    # engine networks: registered sub-modules, and the executor front-ends that wrappers keep OUTSIDE their module
    # registry (`core` of FCDiscriminator / FlawDetector / RotationClassifer: their leaves are registered under the
    # reference's names instead)
    cores, seen = [], set()
    for sub in model.modules():
        for m in (sub, getattr(sub, "core", None)):
            if isinstance(m, SegNetCore) and id(m) not in seen:
                seen.add(id(m))
                cores.append(m)
    for m in cores:
        if not getattr(m, "_pxl_attached", False):
            sync_bn = getattr(m, "sync_bn", True)     # False: local BatchNorm statistics (S4L's rotation classifier)
            # Sync-BN statistics: a peer-mapped one-shot exchange per network when the ranks can map each other's memory
            peer = open_peer_context() if sync_bn else None
            if peer is not None:
                m._pxl_peer = peer
                _peer.setdefault("cores", []).append(m)       # (poll_peers re-wires them if an exchange ever times out)
                m.set_sync_native(_peer["hook"], peer, ws)
                sync_bn = False                         # wired; the branches below only handle the gradient path
            if comms:                      # networks take the communicators in creation order (identical on all ranks)
                m._pxl_comm = comms[_native["next"] % len(comms)]
                _native["next"] += 1
                if sync_bn:
                    m.set_sync_native(_native["hook"], m._pxl_comm, ws)
            elif sync_bn:
                m.set_sync(_sync_stats_callback, ws)
            m._post_backward_hook = _post_backward
            if os.environ.get("PXL_GRAD_OVERLAP", "1") != "0":
                if comms and _native["grad_comm"] is not None:
                    m.set_grad_sync(_native["hook"], _native["grad_comm"], ws, GRAD_BUCKET_FLOATS)
                elif not comms:
                    def _cb(user, buf, n, stream):
                        try:
                            return int(_grad_sync_callback(buf, n, stream) or 0)
                        except Exception:   # never unwind through C
                            import traceback
                            traceback.print_exc()
                            return 1
                    from . import _lib
                    m._grad_cb = _lib.ALLREDUCE_FN(_cb)
                    m.set_grad_sync(m._grad_cb, None, ws, GRAD_BUCKET_FLOATS)
            m._pxl_attached = True
    return model
