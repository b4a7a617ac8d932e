"""Hardware-queue-aware stream placement (csrc/streams.hip): every component that overlaps work with the main stream asks
for the stream of its ROLE instead of creating one.  HIP deals streams onto 4 hardware queues and streams that share a queue
serialize, so with ad-hoc `torch.cuda.Stream()` objects the overlap depended on how many streams the process had created
before (measured: MT 13.2 <-> 15.8 ms, AdvSSL 17.5 <-> 19.5 ms, CCT 27.0 <-> 34.7 ms for 0 ... 3 extra streams at start-up).

    role_stream(SIDE)    a second network next to the main stream (MT teacher, GCT r model, AdvSSL discriminator update, CCT
                         labeled pass)
    role_stream(WGRAD)   weight gradients (the executor takes it itself, csrc/net.cpp)
    role_stream(AUX)     weight packing, a third chain

The pool is built around the stream that is current at the first call (the stream the training step is enqueued on) and
probes which candidate streams really overlap with it.  PXL_STREAM_POOL=0: every caller gets a fresh torch stream, as before."""
import ctypes

import torch

SIDE, WGRAD, AUX = 0, 1, 2
_cache = {}
_state = {"init": False, "queues": 0}


def init(main_stream=None):
    """Build the pool (idempotent) -> number of hardware queues found besides the main stream's."""
    if not _state["init"]:
        from ._lib import lib, check
        main = torch.cuda.current_stream() if main_stream is None else main_stream
        n = ctypes.c_int(0)
        check(lib().pxl_stream_pool_init(ctypes.c_void_p(main.cuda_stream), ctypes.byref(n)))
        _state["init"], _state["queues"] = True, n.value
    return _state["queues"]


def role_stream(role, device=None, index=0):
    """torch stream object of a role.  `index` > 0 asks for a further stream of the same kind (CCT's decoder lanes): roles are
    dealt round-robin over the queues that are not the main stream's."""
    if not torch.cuda.is_available():
        return None
    key = (int(role), int(index))
    s = _cache.get(key)
    if s is None:
        from ._lib import lib
        nq = init()
        # index 0: the role's own queue.  Further lanes of a role are dealt over the OTHER queues (lane 1 of AUX with three
        # queues used to be role 3 -> queue 0 = SIDE's, the stream the labeled pass holds, and (AUX, 1) / (SIDE, 3) shared
        # one cache entry): the role's own queue comes last in the rotation
        if int(index) == 0 or nq <= 1:
            slot = int(role)
        else:
            own = int(role) % nq
            others = [q for q in range(nq) if q != own and (q != SIDE % nq or int(role) == SIDE)] or [q for q in range(nq) if q != own]
            order = others + [own]                      # e.g. AUX lanes with three queues: AUX, WGRAD, AUX, WGRAD ... never SIDE
            slot = order[(int(index) - 1) % len(order)]
        p = lib().pxl_stream_role(slot)
        if p:
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            s = torch.cuda.ExternalStream(int(p), device=dev)
        else:                                   # placement off (PXL_STREAM_POOL=0) or a single hardware queue
            s = torch.cuda.Stream(device=device)
        _cache[key] = s
    return s


def queues():
    return _state["queues"]
