"""Optimizer factories with the reference's calling convention (pixelssl/nn/optimizer.py:57-75):
`sgd(args)` returns `wrapper(param_groups) -> Optimizer`.  The optimizer is a single fused HIP launch per
learning-rate group over the model's flat parameter buffer instead of 320 per-tensor updates."""
import torch
from torch.optim.optimizer import Optimizer

from ..utils import cmd
from .. import _lib
from .. import ops

VALID_OPTIMIZER = ['sgd', 'adam']


def add_parser_arguments(parser):
    """Same flag names / '-1 = use the optimizer default' convention as the reference parser."""
    parser.add_argument('--lr', type=float, default=-1, metavar='', help='optimizer - learning rate')
    parser.add_argument('--dampening', type=float, default=-1, metavar='', help='optimizer - dampening (sgd)')
    parser.add_argument('--nesterov', type=cmd.str2bool, default=False, metavar='', help='optimizer - nesterov (sgd)')
    parser.add_argument('--weight-decay', type=float, default=-1, metavar='', help='optimizer - weight decay')
    parser.add_argument('--momentum', type=float, default=-1, metavar='', help='optimizer - momentum (sgd)')
    parser.add_argument('--alpha', type=float, default=-1, metavar='', help='optimizer - (rmsprop)')
    parser.add_argument('--centered', type=cmd.str2bool, default=False, metavar='', help='optimizer - (rmsprop)')
    parser.add_argument('--eps', type=float, default=-1, metavar='', help='optimizer - eps (adam family)')
    parser.add_argument('--beta1', type=float, default=-1, metavar='', help='optimizer - beta1 (adam family)')
    parser.add_argument('--beta2', type=float, default=-1, metavar='', help='optimizer - beta2 (adam family)')
    parser.add_argument('--amsgrad', type=cmd.str2bool, default=False, metavar='', help='optimizer - (wdadam)')


def _segments(params):
    """Group pixelhip-managed parameters into maximal contiguous (store, offset, length) runs.  Parameters that do not
    live in an engine model's flat store (any other torch TaskModel a plugin brings) are skipped here: the optimizers
    update them through their per-tensor path (`_foreign`)."""
    runs = []
    items = []
    for p in params:
        ref = getattr(p, '_pxl_flat', None)
        if ref is None:
            continue
        items.append(ref)
    items.sort(key=lambda r: (id(r[0]), r[1]))
    for store, off, n in items:
        padded = (n + 3) // 4 * 4
        if runs and runs[-1][0] is store and runs[-1][1] + runs[-1][2] == off:
            runs[-1][2] += padded
        else:
            runs.append([store, off, padded])
    return runs


def _foreign(params):
    return [p for p in params if getattr(p, '_pxl_flat', None) is None]


def _sync_foreign_grads(groups):
    """Multi-rank: mean of the per-rank gradients of every tensor that is NOT owned by an engine model (those are
    exchanged by the executor, overlapped with the backward pass).  One all-reduce over the concatenated gradients."""
    from .. import dist as pdist
    if not pdist.is_distributed():
        return
    pdist.poll_peers()       # (once per optimizer step: every PEER_POLL_STEPS-th call checks the peer-mapped exchanges)
    # EVERY foreign parameter takes part, with zeros where this rank has no gradient: the set of tensors with a gradient
    # may differ from rank to rank (a conditional branch in a plugin model), and all-reduces of different lengths hang
    ps = [p for foreign in groups for p in foreign]
    if not ps:
        return
    # which tensors received a gradient on ANY rank rides in the SAME all-reduce (one flag per tensor appended to the gradients;
    # its mean is > 0 iff some rank had one): those step with the mean on EVERY rank -- a rank that had none contributed zeros to
    # the mean and must apply it too, or the replicas drift apart; tensors without a gradient anywhere keep grad = None (the
    # optimizers skip them, like torch's).  The flags are read back (a device sync) only on a rank that misses a gradient.
    dev = ps[0].device
    flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in ps], device=dev, dtype=ps[0].dtype)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(ps[0].dtype) for p in ps] + [flags])
    pdist.allreduce_mean_(flat)
    had = flat[-len(ps):].tolist() if any(p.grad is None for p in ps) else None
    off = 0
    for k, p in enumerate(ps):
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
        elif had[k] > 0:
            p.grad = flat[off:off + n].view_as(p).clone().to(p.dtype)
        off += n


def _flat_view(store, buf, p):
    """The slice of a flat per-store buffer (momentum, Adam moments) that belongs to parameter `p`, shaped like it."""
    from ..engine import FlatStore
    _, off, n = p._pxl_flat
    return FlatStore._view(buf, tuple(p.shape), n, off, getattr(p, '_pxl_alloc', None))


class _FlatStateMixin:
    """state_dict / load_state_dict in torch's per-parameter format for optimizers whose state lives in flat buffers
    on the FlatStore: checkpoints written here load into torch.optim.SGD / Adam of the reference (and vice versa:
    ssl_mt.py:296-322 saves `optimizer.state_dict()` and restores it on --resume)."""

    _STATE_KEYS = ()          # (state_dict key, store attribute)

    def _param_list(self):
        return [p for g in self.param_groups for p in g['params']]

    def state_dict(self):
        sd = Optimizer.state_dict(self)
        state = {}
        if self._steps_taken > 0:
            for idx, p in enumerate(self._param_list()):
                if getattr(p, '_pxl_flat', None) is None:          # foreign tensor: its state is torch-style already
                    if p in self.state and self.state[p]:
                        state[idx] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.state[p].items()}
                    continue
                store = p._pxl_flat[0]
                entry = {k: _flat_view(store, getattr(store, attr), p).detach().clone().contiguous()
                         for k, attr in self._STATE_KEYS}
                entry.update(self._extra_state())
                state[idx] = entry
        sd['state'] = state
        return sd

    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if len(groups) != len(self.param_groups) or any(len(g['params']) != len(mine['params'])
                                                        for g, mine in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        for g, mine in zip(groups, self.param_groups):
            for k, v in g.items():
                if k != 'params':
                    mine[k] = v
        params = self._param_list()
        state = state_dict.get('state', {})
        for store in self._stores.values():
            for _, attr in self._STATE_KEYS:
                getattr(store, attr).zero_()
        steps = 0
        with torch.no_grad():
            for idx, p in enumerate(params):
                entry = state.get(idx, state.get(str(idx)))
                if not entry:
                    continue
                if getattr(p, '_pxl_flat', None) is None:
                    self.state[p] = {k: (v.to(p.device).clone() if torch.is_tensor(v) else v) for k, v in entry.items()}
                    steps = max(steps, self._steps_of(entry))
                    continue
                store = p._pxl_flat[0]
                for k, attr in self._STATE_KEYS:
                    if entry.get(k) is not None:
                        _flat_view(store, getattr(store, attr), p).copy_(entry[k].to(store.params.device))
                steps = max(steps, self._steps_of(entry))
        self._steps_taken = steps
        self._after_load(steps)

    def _extra_state(self):
        return {}

    def _steps_of(self, entry):
        return 1

    def _after_load(self, steps):
        pass


class FusedSGD(_FlatStateMixin, Optimizer):
    """torch.optim.SGD semantics (momentum, dampening, nesterov, weight decay):
    d = g + wd*p ; buf = m*buf + (1-dampening)*d (buf = d on the first step) ; p -= lr * (nesterov ? d + m*buf : buf).
    Parameters of an engine model are updated by one fused launch per contiguous run of its flat buffer; any other
    tensor (a torch TaskModel a plugin brings) by the same arithmetic through torch ops."""

    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')       # torch.optim.SGD's check
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self._runs, self._foreign = [], []
        for group in self.param_groups:
            group['params'] = list(group['params'])
            self._runs.append(_segments(group['params']))
            self._foreign.append(_foreign(group['params']))
        self._stores = {}
        for runs in self._runs:
            for store, _, _ in runs:
                self._stores[id(store)] = store
        for store in self._stores.values():
            if not hasattr(store, 'momentum'):
                store.momentum = torch.zeros_like(store.params)
        self._steps_taken = 0
        FusedSGD._instances += 1
        self._hp_name = 'sgd%d' % FusedSGD._instances

    _instances = 0
    _STATE_KEYS = (('momentum_buffer', 'momentum'),)

    def hyper_values(self):
        """the per-step scalars of the next step() for a captured training step (graph.HyperBlock.upload): one learning rate
        per parameter group, as the lr scheduler left them"""
        return {'%s.lr%d' % (self._hp_name, gi): float(g['lr']) for gi, g in enumerate(self.param_groups)}

    def after_replayed_step(self):
        """host bookkeeping of one step() that a graph replay performed on the device"""
        self._steps_taken += 1
        for store in self._stores.values():
            store.touch()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        pipe = getattr(self, '_pipeline', None)
        self.last_step_pipelined = False
        if pipe is not None and pipe.pending:
            # the update of this step was applied bucket by bucket from inside the backward pass (PipelinedUpdate)
            _sync_foreign_grads(self._foreign)
            pipe.finish()
            self.last_step_pipelined = True
            return loss
        if pipe is not None and pipe.armed:
            # armed, but the executor never called the hook (a program whose parameter offsets are not monotonic, weight gradients
            # off, multi-rank with a partial gradient range): this IS the ordinary whole-buffer step -- the caller reads
            # `last_step_pipelined` and performs what the buckets would have done besides SGD (Mean Teacher: the EMA update)
            pipe.disarm()
        first = self._steps_taken == 0
        self._steps_taken += 1
        _sync_foreign_grads(self._foreign)
        from .. import graph as pgraph
        hyper = pgraph.current_hyper()          # a captured step: the learning rates are read from device memory
        for gi, (group, runs, foreign) in enumerate(zip(self.param_groups, self._runs, self._foreign)):
            lr, mom, wd = float(group['lr']), float(group['momentum']), float(group['weight_decay'])
            damp, nest = float(group.get('dampening', 0.0)), bool(group.get('nesterov', False))
            if hyper is not None and (damp != 0.0 or nest or foreign):
                raise _lib.PixelHipError('FusedSGD: a captured step supports plain momentum SGD over engine parameters only')
            lr_dev = hyper.ptr('%s.lr%d' % (self._hp_name, gi)) if hyper is not None else None
            for store, off, n in runs:
                if damp == 0.0 and not nest:
                    # (buf starts at zero, so m * 0 + d = d: the plain kernel needs no first-step flag)
                    ops.sgd_step(store.params[off:off + n], store.grads[off:off + n], store.momentum[off:off + n], lr, mom, wd,
                                 lr_dev=lr_dev)
                else:
                    _lib.check(_lib.lib().pxl_sgd_step_general(n, _lib.ptr(store.params[off:off + n]), _lib.ptr(store.grads[off:off + n]),
                                                               _lib.ptr(store.momentum[off:off + n]), lr, mom, damp, wd, int(nest),
                                                               int(first), _lib.stream_ptr()))
            for p in foreign:
                if p.grad is None:
                    continue
                d = p.grad.add(p, alpha=wd) if wd != 0 else p.grad
                if mom != 0:
                    st = self.state[p]
                    if 'momentum_buffer' not in st or st['momentum_buffer'] is None:
                        buf = st['momentum_buffer'] = torch.clone(d).detach()
                    else:
                        buf = st['momentum_buffer']
                        buf.mul_(mom).add_(d, alpha=1 - damp)
                    d = d.add(buf, alpha=mom) if nest else buf
                p.add_(d, alpha=-lr)
        for store in self._stores.values():
            store.touch()
        return loss

    def zero_grad(self, set_to_none=False):
        pipe = getattr(self, '_pipeline', None)
        if pipe is not None and pipe.grads_clean:
            return          # every bucket was zeroed right after its update consumed it
        # gradients are views of one flat buffer: one memset, views stay attached
        for store in self._stores.values():
            store.grads.zero_()
        for foreign in self._foreign:
            for p in foreign:
                if p.grad is not None:
                    p.grad = None if set_to_none else p.grad.detach().zero_()


class PipelinedUpdate:
    """The optimizer step of an engine model -- and, for Mean Teacher, the EMA update of the teacher and the re-packing of both
    networks' kernel-layout weights -- applied BUCKET BY BUCKET from inside the student's backward pass (csrc/net.cpp:
    pxl_net_set_update_hook) instead of after it.  The reference runs backward, optimizer.step(), the EMA loop one after the
    other (ssl_mt.py:198-204); element by element the arithmetic here is the same (the same fused kernels on sub-ranges of the
    flat buffers), only its place in time changes: the update of the head / layer 4 / layer 3 ... runs on the communication
    stream next to the data gradients of the layers below, and what is left between two iterations is the bucket of the first
    layers.  Round 4 trace: 0.7 ms of a 12.2 ms step were SGD + EMA + packing with nothing beside them.

    Protocol (one step): arm(...) before the backward; the executor calls _on_bucket(lo, hi, stream) per bucket; optimizer.step()
    finds `pending` and only does the host bookkeeping (finish).  A backward that was not armed leaves the gradients alone and the
    next optimizer.step() is the ordinary whole-buffer step."""

    def __init__(self, optimizer, s_core, t_core=None, bucket_mb=None, tail_floats=None):
        import os
        if type(optimizer) is not FusedSGD or any(optimizer._foreign):
            raise ValueError('PipelinedUpdate: needs a FusedSGD over engine parameters only')
        stores = list(optimizer._stores.values())
        if len(stores) != 1 or stores[0] is not s_core.flat:
            raise ValueError('PipelinedUpdate: the optimizer must hold exactly the parameters of the student network')
        for g in optimizer.param_groups:
            if float(g.get('dampening', 0.0)) != 0.0 or g.get('nesterov', False):
                raise ValueError('PipelinedUpdate: plain momentum SGD only')
        if t_core is not None and t_core.flat.np != s_core.flat.np:
            raise ValueError('PipelinedUpdate: student and teacher parameter layouts differ')
        covered = sorted((off, off + n) for runs in optimizer._runs for _, off, n in runs)
        pos = 0
        for a, b in covered:
            if a != pos:
                raise ValueError('PipelinedUpdate: the parameter groups do not tile the flat buffer')
            pos = b
        if pos != s_core.flat.np:
            raise ValueError('PipelinedUpdate: the parameter groups do not cover the flat buffer')
        self.optimizer, self.s_core, self.t_core = optimizer, s_core, t_core
        self.pending = False
        self.grads_clean = False
        self.armed = False
        self.covered = 0
        self.buckets = 0
        # PXL_FUSED_UPDATE=1 (default; bf16 engine): SGD + EMA + the bf16 forward copies of both networks + the gradient memset as ONE
        # kernel per bucket (csrc/optim.hip: pxl_sgd_ema_pack), same arithmetic element for element
        self.fused = os.environ.get('PXL_FUSED_UPDATE', '1') == '1' and s_core._code == _lib.PXL_BF16 and \
            (t_core is None or t_core._code == _lib.PXL_BF16)
        if self.fused:
            # the fused kernel takes at most 8 learning-rate runs and ONE momentum / weight decay (checked HERE, not in the C
            # callback during the backward: an optimizer with per-group weight decay must fall back to the ordinary step when the
            # pipeline is built, not fail every iteration with 'parameter-update hook failed')
            nr = sum(len(rr) for rr in optimizer._runs)
            g0 = optimizer.param_groups[0]
            mom, wd = float(g0['momentum']), float(g0['weight_decay'])
            if nr > 8 or any(float(g['momentum']) != mom or float(g['weight_decay']) != wd for g in optimizer.param_groups):
                if os.environ.get('PXL_FUSED_UPDATE_STRICT') == '1':
                    raise ValueError('PipelinedUpdate (fused): needs <= 8 learning-rate runs and one momentum / weight decay '
                                     'for every parameter group')
                self.fused = False           # the per-group kernels of the unfused bucket update serve any plain-SGD groups
        self._segs = None
        # bucket size: by default ONE bucket (the whole buffer at the end of the backward pass; smaller buckets pipeline the update
        # behind the pass, measured no faster); multi-rank runs cut at the gradient exchange's buckets whatever this says
        mb = float(os.environ.get('PXL_UPDATE_BUCKET_MB', '1000000')) if bucket_mb is None else float(bucket_mb)
        tail = int(os.environ.get('PXL_UPDATE_TAIL_FLOATS', '300000')) if tail_floats is None else int(tail_floats)
        s_core.set_update_hook(self._on_bucket, int(mb * (1 << 20) / 4), tail)
        optimizer._pipeline = self

    def detach(self):
        self.s_core.set_update_hook(None)
        self.optimizer._pipeline = None

    def disarm(self):
        """an armed step whose backward never reached the hook: back to the state of an ordinary step (the gradients were not
        zeroed bucket by bucket, nothing is pending)"""
        self.armed, self.pending, self.grads_clean = False, False, False
        self.covered, self.buckets = 0, 0

    def arm(self, s_plan, t_plan=None, ema_alpha=None, hyper=None):
        """before the backward of a step whose update is to be pipelined; ema_alpha: this step's EMA coefficient (ignored when
        `hyper` is given: the captured step reads it -- and the learning rates -- from the device block)"""
        self.s_plan, self.t_plan, self.alpha, self.hyper = s_plan, t_plan, ema_alpha, hyper
        self.armed, self.covered, self.buckets = True, 0, 0
        if self.fused and (self._segs is None or self._segs[0] is not s_plan or self._segs[1] is not t_plan):
            import ctypes
            arr = (_lib.UpdSeg * 512)()
            n = _lib.lib().pxl_net_update_segments(s_plan.net, t_plan.net if t_plan is not None else None, arr, 512)
            if n < 0:
                _lib.check(n)
            host = torch.tensor([[arr[k].off, arr[k].n, arr[k].s_pk, arr[k].t_pk] for k in range(n)], dtype=torch.int64).reshape(-1, 4)
            self._segs = (s_plan, t_plan, host.to(self.s_core.flat.params.device), n)

    def _fused_bucket(self, lo, hi):
        import ctypes
        opt, store, hyper = self.optimizer, self.s_core.flat, self.hyper
        runs = sorted((off, gi) for gi, rr in enumerate(opt._runs) for _, off, _n in rr)
        nr = len(runs)
        if nr > 8:
            raise _lib.PixelHipError('PipelinedUpdate: more than 8 learning-rate runs')
        starts = (ctypes.c_long * nr)(*[r[0] for r in runs])
        lrs = (ctypes.c_float * nr)(*[float(opt.param_groups[r[1]]['lr']) for r in runs])
        devs = (ctypes.c_void_p * nr)(*[(hyper.ptr('%s.lr%d' % (opt._hp_name, r[1])) if hyper is not None else None) for r in runs])
        g0 = opt.param_groups[0]
        mom, wd = float(g0['momentum']), float(g0['weight_decay'])
        if any(float(g['momentum']) != mom or float(g['weight_decay']) != wd for g in opt.param_groups):
            raise _lib.PixelHipError('PipelinedUpdate (fused): momentum / weight decay must be the same in every parameter group')
        t = self.t_core.flat.params if self.t_core is not None else None
        _lib.check(_lib.lib().pxl_sgd_ema_pack(
            lo, hi, _lib.ptr(store.params), _lib.ptr(store.grads), _lib.ptr(store.momentum), _lib.ptr(t), nr, starts, lrs, devs,
            mom, wd, float(self.alpha) if self.alpha is not None else 0.0,
            hyper.ptr('ema_alpha') if (hyper is not None and t is not None) else None,
            _lib.ptr(self._segs[2]), self._segs[3], _lib.ptr(self.s_plan.packed),
            _lib.ptr(self.t_plan.packed) if self.t_plan is not None else None, 1, _lib.stream_ptr()))
        self.s_core.pack_range(self.s_plan, lo, hi, 2 | 4)
        if self.t_core is not None:
            self.t_core.pack_range(self.t_plan, lo, hi, 4)

    @torch.no_grad()
    def _on_bucket(self, lo, hi, stream):
        if not self.armed:
            self.grads_clean = False          # an ordinary backward: the gradients stay for an ordinary step
            return
        opt, store = self.optimizer, self.s_core.flat
        hyper = self.hyper
        if self.fused:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                self._fused_bucket(lo, hi)
            self.covered += hi - lo
            self.buckets += 1
            self.pending = True
            return
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            for gi, (group, runs) in enumerate(zip(opt.param_groups, opt._runs)):
                lr, mom, wd = float(group['lr']), float(group['momentum']), float(group['weight_decay'])
                lr_dev = hyper.ptr('%s.lr%d' % (opt._hp_name, gi)) if hyper is not None else None
                for _, off, n in runs:
                    a, b = max(lo, off), min(hi, off + n)
                    if a < b:
                        ops.sgd_step(store.params[a:b], store.grads[a:b], store.momentum[a:b], lr, mom, wd, lr_dev=lr_dev)
            store.grads[lo:hi].zero_()
            self.s_core.pack_range(self.s_plan, lo, hi, 3)
            if self.t_core is not None:
                ops.ema_update(self.t_core.flat.params[lo:hi], store.params[lo:hi], self.alpha,
                               alpha_dev=hyper.ptr('ema_alpha') if hyper is not None else None)
                self.t_core.pack_range(self.t_plan, lo, hi, 1)
        self.covered += hi - lo
        self.buckets += 1
        self.pending = True

    def finish(self):
        """host bookkeeping of the step the buckets performed (called by optimizer.step())"""
        store = self.s_core.flat
        if self.covered != store.np:
            raise _lib.PixelHipError('PipelinedUpdate: the buckets of this step covered %d of %d parameters' % (self.covered, store.np))
        self.optimizer._steps_taken += 1
        store.touch()
        self.s_plan.packed_version = store.version()
        self.s_plan.wt_ready = None           # (both layouts were packed on the update stream, which the backward's stream has joined)
        if self.t_core is not None:
            self.t_core.flat.touch()
            self.t_plan.packed_version = self.t_core.flat.version()
        self.pending, self.armed, self.grads_clean = False, False, True

    def replayed(self):
        """host bookkeeping of one step that a hipGraph replay performed"""
        self.optimizer._steps_taken += 1
        self.s_core.flat.touch()
        self.s_plan.packed_version = self.s_core.flat.version()
        if self.t_core is not None:
            self.t_core.flat.touch()
            self.t_plan.packed_version = self.t_core.flat.version()
        self.grads_clean = True


class FusedAdam(_FlatStateMixin, Optimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient: g += wd * p; no amsgrad) as one fused launch per
    contiguous run of the flat parameter buffer: the discriminator optimizer of AdvSSL (ssl_adv.py:101-102, betas
    (0.9, 0.99)) and the `adam` factory."""

    _STATE_KEYS = (('exp_avg', 'exp_avg'), ('exp_avg_sq', 'exp_avg_sq'))

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._runs, self._foreign = [], []
        for group in self.param_groups:
            group['params'] = list(group['params'])
            self._runs.append(_segments(group['params']))
            self._foreign.append(_foreign(group['params']))
        self._stores = {}
        for runs in self._runs:
            for store, _, _ in runs:
                self._stores[id(store)] = store
        for store in self._stores.values():
            if not hasattr(store, 'exp_avg'):
                store.exp_avg = torch.zeros_like(store.params)
                store.exp_avg_sq = torch.zeros_like(store.params)
        self._step = 0
        self._steps_taken = 0

    def _extra_state(self):
        return {'step': torch.tensor(float(self._step))}

    def _steps_of(self, entry):
        st = entry.get('step', 0)
        return int(st.item() if torch.is_tensor(st) else st)

    def _after_load(self, steps):
        self._step = steps

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._step += 1
        self._steps_taken = self._step
        _sync_foreign_grads(self._foreign)
        for group, runs, foreign in zip(self.param_groups, self._runs, self._foreign):
            b1, b2 = group['betas']
            wd = float(group.get('weight_decay', 0.0))
            for store, off, n in runs:
                _lib.check(_lib.lib().pxl_adam_step_wd(n, _lib.ptr(store.params[off:off + n]), _lib.ptr(store.grads[off:off + n]),
                                                       _lib.ptr(store.exp_avg[off:off + n]), _lib.ptr(store.exp_avg_sq[off:off + n]),
                                                       float(group['lr']), float(b1), float(b2), float(group['eps']), wd,
                                                       self._step, _lib.stream_ptr()))
            for p in foreign:                  # torch.optim.Adam's single-tensor update
                if p.grad is None:
                    continue
                g = p.grad.add(p, alpha=wd) if wd != 0 else p.grad
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'], st['exp_avg_sq'] = torch.zeros_like(p), torch.zeros_like(p)
                st['step'] = torch.tensor(float(self._step))
                st['exp_avg'].mul_(b1).add_(g, alpha=1 - b1)
                st['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** self._step, 1 - b2 ** self._step
                denom = (st['exp_avg_sq'].sqrt() / (bc2 ** 0.5)).add_(float(group['eps']))
                p.addcdiv_(st['exp_avg'], denom, value=-float(group['lr']) / bc1)
        for store in self._stores.values():
            store.touch()
        return loss

    def zero_grad(self, set_to_none=False):
        for store in self._stores.values():
            store.grads.zero_()
        for foreign in self._foreign:
            for p in foreign:
                if p.grad is not None:
                    p.grad = None if set_to_none else p.grad.detach().zero_()


def sgd(args):
    args.lr = 0.01 if args.lr == -1 else args.lr
    args.weight_decay = 0 if args.weight_decay == -1 else args.weight_decay
    args.momentum = 0 if args.momentum == -1 else args.momentum
    args.dampening = 0 if args.dampening == -1 else args.dampening

    def sgd_wrapper(param_groups):
        return FusedSGD(param_groups, lr=args.lr, momentum=args.momentum, dampening=args.dampening,
                        weight_decay=args.weight_decay, nesterov=args.nesterov)

    return sgd_wrapper


def adam(args):
    """`adam(args)` factory (pixelssl/nn/optimizer.py:103-122) on the fused kernel."""
    args.lr = 0.001 if args.lr == -1 else args.lr
    args.beta1 = 0.9 if args.beta1 == -1 else args.beta1
    args.beta2 = 0.999 if args.beta2 == -1 else args.beta2
    args.eps = 1e-08 if args.eps == -1 else args.eps
    args.weight_decay = 0.0 if args.weight_decay == -1 else args.weight_decay

    def adam_wrapper(param_groups):
        return FusedAdam(param_groups, lr=args.lr, betas=(args.beta1, args.beta2), eps=args.eps,
                         weight_decay=args.weight_decay)

    return adam_wrapper
