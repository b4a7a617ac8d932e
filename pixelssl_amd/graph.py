"""Captured training steps: one hipGraph launch per iteration instead of ~700 kernel launches + ~250 event operations from
the host (round 4: the host needed 10.2 of the 12.2 ms of an MT step just to ENQUEUE it; CCT and GCT were paced by it).

The reference's loop (ssl_mt.py:124-224) is host code around per-layer torch calls; here an iteration is already three C
calls (two forward passes, one backward) + a handful of loss / optimizer launches, all asynchronous, on up to four HIP
streams joined by events.  That whole DAG is recorded once with HIP stream capture (torch.cuda.CUDAGraph supplies the
capture-safe allocator pool; every launch of libpixelhip lands in the same capture because it is issued on the captured
streams) and replayed.  Two things make a step replayable:

  * per-step SCALARS (learning rates, EMA coefficient, ramped loss weight) must not sit in kernel arguments: they live in
    a small device block (`HyperBlock`) that is written EAGERLY, by one tiny launch carrying the values in its own
    arguments, right before the graph launch; the kernels of the step read them from there (csrc: pxl_*_hp entry points);
  * per-step HOST bookkeeping (scheduler counters, optimizer step counts, weight-version stamps) is done by the caller
    after the replay -- the captured body only enqueues.

The captured body is exactly the eager body (`fn`), run in "hyper mode": the first `warmup` calls run it eagerly (autotune,
lazy stream / event creation), the next call captures, later calls replay.  Any call whose input shapes differ from the
captured ones runs eagerly (a short last batch).  Opt-in: PXL_GRAPH=1 (see enabled() for the measurement that keeps it off by
default); it is never used multi-rank (the peer-mapped Sync-BN exchanges carry an epoch counter in their arguments)."""
import ctypes
import os

import torch

from . import _lib

_current = [None]      # the HyperBlock of the step being enqueued (None: scalars travel as kernel arguments, as always)


def current_hyper():
    return _current[0]


def enabled():
    """OFF unless PXL_GRAPH=1.  Measured on this image (ROCm 7.2, MI355X; DESIGN.md 4 round 5, profiles/r05_a_graph_*): the captured
    MT step replays CORRECTLY (same losses / weights as the eager step, the 513 x 513 fixture included) but SLOWER -- 13.9 ms
    against 12.5 ms eager: hipGraphLaunch spends as long on the host as the 704 eager launches do (8.9 vs 8.0 ms per step), and
    the runtime executes the four captured branches almost one after the other (1.0 - 1.1 kernels in flight where the eager
    streams keep 1.6 - 2.0)."""
    return os.environ.get("PXL_GRAPH", "0") == "1"


class HyperBlock:
    """<= 32 named fp32 scalars in device memory, refreshed once per step by `upload` (one launch, values in its arguments)."""
    CAP = 32

    def __init__(self, device):
        self.dev = torch.zeros(self.CAP, device=device, dtype=torch.float32)
        self.names = {}
        self._fresh = set()
        self._buf = (ctypes.c_float * self.CAP)()

    def slot(self, name):
        if name not in self.names:
            if len(self.names) >= self.CAP:
                raise _lib.PixelHipError("HyperBlock: more than %d per-step scalars" % self.CAP)
            self.names[name] = len(self.names)
        return self.names[name]

    def upload(self, values):
        """values: {name: float}.  Every scalar the step reads has to be in here -- ptr() refuses names that were not."""
        for k, v in values.items():
            self._buf[self.slot(k)] = float(v)
        self._fresh = set(values.keys())
        _lib.check(_lib.lib().pxl_hyper_set(self.dev.data_ptr(), self._buf, len(self.names), _lib.stream_ptr()))

    def ptr(self, name):
        if name not in self._fresh:
            raise _lib.PixelHipError("HyperBlock: scalar %r was not uploaded for this step (have %s)" % (name, sorted(self._fresh)))
        return self.dev.data_ptr() + 4 * self.names[name]

    def tensor(self, name):
        """0-dim device tensor view of a scalar (for torch arithmetic inside the step)"""
        if name not in self._fresh:
            raise _lib.PixelHipError("HyperBlock: scalar %r was not uploaded for this step" % name)
        return self.dev[self.names[name]]


class HostState:
    """The scalar host-side state (counters, flags, learning rates) of the objects a captured body touches WHILE BEING RECORDED: the
    scheduler's step count, the optimizer's counters and per-group learning rates, the update pipeline's flags.  A capture that fails
    has already advanced them once; StepGraph restores the snapshot before it runs the same iteration eagerly."""

    _SCALAR = (int, float, bool, str, type(None))

    def __init__(self, objs=(), param_groups=()):
        self.objs = [o for o in objs if o is not None]
        self.groups = list(param_groups)
        self._snap = None

    def save(self):
        self._snap = ([{k: v for k, v in vars(o).items() if isinstance(v, self._SCALAR)} for o in self.objs],
                      [{k: v for k, v in g.items() if isinstance(v, self._SCALAR)} for g in self.groups])

    def restore(self):
        if self._snap is None:
            return
        for o, d in zip(self.objs, self._snap[0]):
            for k, v in d.items():
                setattr(o, k, v)
        for g, d in zip(self.groups, self._snap[1]):
            g.update(d)
        self._snap = None


class StepGraph:
    """fn(*tensors) -> flat tuple of 0-dim / small tensors (the step's logged values); scalars_fn(*step_args) -> {name: float};
    after_fn(): the host bookkeeping of one step (called after every replay; the eager calls do theirs inside fn)."""

    def __init__(self, fn, scalars_fn, after_fn, device, warmup=2, host_state=None):
        self.fn, self.scalars_fn, self.after_fn = fn, scalars_fn, after_fn
        self.host_state = host_state          # HostState: rolled back when a capture attempt fails (its body ran the host half once)
        self.hyper = HyperBlock(device)
        self.warmup = int(os.environ.get("PXL_GRAPH_WARMUP", warmup))
        self.calls = 0
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.sig = None
        self.replays = 0
        self.failed = None          # why capture was given up (the step then stays eager)

    @staticmethod
    def _sig(tensors):
        return tuple((tuple(t.shape), t.dtype, t.device) for t in tensors)

    def _eager(self, tensors, step_args):
        self.hyper.upload(self.scalars_fn(*step_args))
        _current[0] = self.hyper
        try:
            return self.fn(*tensors)
        finally:
            _current[0] = None

    def step(self, tensors, step_args):
        """tensors: the device inputs of the iteration; step_args: what scalars_fn needs (step counters).  -> fn's outputs
        (replays: fresh copies, so a caller may keep them across steps)."""
        tensors = tuple(tensors)
        self.calls += 1
        if self.failed is not None or self.calls <= self.warmup:
            return self._eager(tensors, step_args)
        sig = self._sig(tensors)
        if self.graph is None:
            return self._capture(tensors, step_args, sig)
        if sig != self.sig:
            return self._eager(tensors, step_args)
        for dst, src in zip(self.static_in, tensors):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.hyper.upload(self.scalars_fn(*step_args))
        self.graph.replay()
        self.replays += 1
        self.after_fn()
        out = self.static_out
        packed = torch.stack([o.reshape(()) for o in out]) if all(o.numel() == 1 for o in out) else None
        if packed is not None:
            return tuple(packed[i] for i in range(len(out)))
        return tuple(o.clone() for o in out)

    def _capture(self, tensors, step_args, sig):
        torch.cuda.synchronize()
        self.static_in = tuple(t.clone() for t in tensors)
        self.hyper.upload(self.scalars_fn(*step_args))
        g = torch.cuda.CUDAGraph()
        _current[0] = self.hyper
        if self.host_state is not None:
            self.host_state.save()
        try:
            # relaxed: libpixelhip's side streams join the capture through events, and other threads of the process (data
            # loader workers pinning memory) must stay free to call the runtime
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                out = self.fn(*self.static_in)
        except Exception as e:                  # capture is an optimisation: a step that cannot be captured stays eager
            _current[0] = None
            self.failed = "%s: %s" % (type(e).__name__, e)
            torch.cuda.synchronize()
            from .utils import logger
            logger.log_warn("step capture failed (%s); the training step stays on eager launches\n" % self.failed)
            if os.environ.get("PXL_GRAPH_STRICT") == "1":
                raise
            if self.host_state is not None:     # the recorded body stepped the scheduler / counters; the eager run below does it again
                self.host_state.restore()
            return self._eager(tensors, step_args)
        finally:
            _current[0] = None
        self.graph, self.sig, self.static_out = g, sig, tuple(out)
        # the capture recorded the step without running it: the first replay IS this call's iteration
        g.replay()
        self.replays += 1
        # (fn did its own host bookkeeping while being captured)
        return tuple(o.clone() for o in self.static_out)
