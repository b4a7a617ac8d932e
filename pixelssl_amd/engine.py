"""Host side of the network executor: flat parameter storage, the DeepLab-v2 layer program, and the
autograd bridge that makes one C call per network pass.

Design (MI355X-first):
  * all parameters of a model live in ONE fp32 buffer (conv weights stored [K][kh][kw][C], i.e. the
    memory order of a channels_last OIHW tensor, which is exactly what the kernels consume), all
    gradients in a second one, BN running statistics in a third.  `nn.Parameter`s are strided views,
    so state_dict / checkpoints keep the reference's names and OIHW shapes while the optimizer, the
    EMA teacher update and the gradient all-reduce are single launches over contiguous memory.
  * the network itself is a layer program interpreted by libpixelhip (csrc/net.cpp).
"""
import ctypes
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import Op, BnDesc, check, lib, ptr, stream_ptr

ASPP_RATES = (6, 12, 18, 24)          # task/sseg/module/deeplab_v2.py:23
RESNET_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet101-coco": (3, 4, 23, 3)}


class FlatStore:
    """Three flat fp32 device buffers (params / grads / running stats) + named strided views."""

    def __init__(self, device):
        self.device = device
        self._p_entries = []     # (name, shape_oihw_or_1d, numel, offset)
        self._r_entries = []
        self.np = 0
        self.nr = 0
        self.params = None
        self.grads = None
        self.running = None
        self.user_version = 0

    def add_param(self, name, shape, alloc=None):
        """`alloc`: shape of the slot the executor sees when it is larger than the exposed tensor (channels padded to what
        the kernels take: a 21-channel BatchNorm lives in a 32-channel slot whose tail stays zero).  The exposed
        parameter is the leading sub-block of the slot."""
        alloc = tuple(alloc) if alloc is not None else tuple(shape)
        if len(alloc) != len(shape) or any(a < b for a, b in zip(alloc, shape)):
            raise ValueError("add_param %s: slot %s does not hold %s" % (name, alloc, tuple(shape)))
        n = 1
        for s in alloc:
            n *= s
        off = self.np
        self._p_entries.append((name, tuple(shape), n, off, alloc))
        self.np += (n + 3) // 4 * 4        # keep every tensor 16-byte aligned
        return off

    def add_running(self, name, shape, alloc=None):
        alloc = tuple(alloc) if alloc is not None else tuple(shape)
        n = 1
        for s in alloc:
            n *= s
        off = self.nr
        self._r_entries.append((name, tuple(shape), n, off, alloc))
        self.nr += (n + 3) // 4 * 4
        return off

    def allocate(self):
        self.params = torch.zeros(self.np, device=self.device, dtype=torch.float32)
        self.grads = torch.zeros(self.np, device=self.device, dtype=torch.float32)
        self.running = torch.zeros(max(self.nr, 4), device=self.device, dtype=torch.float32)

    @staticmethod
    def _view(flat, shape, n, off, alloc=None):
        alloc = tuple(shape) if alloc is None else tuple(alloc)
        v = flat[off:off + n]
        if len(shape) == 4:            # stored [O][kh][kw][I], exposed as OIHW
            o, i, kh, kw = shape
            oa, ia = alloc[0], alloc[1]
            return v.view(oa, kh, kw, ia)[:o, :, :, :i].permute(0, 3, 1, 2)
        if len(shape) == 2:            # Linear weight [O][I] (a 1x1 convolution over a 1x1 map)
            return v.view(alloc)[:shape[0], :shape[1]]
        return v.view(alloc)[:shape[0]]

    def param_views(self):
        return OrderedDict((name, (self._view(self.params, shape, n, off, alloc), self._view(self.grads, shape, n, off, alloc),
                                   off, n, alloc))
                           for name, shape, n, off, alloc in self._p_entries)

    def running_views(self):
        return OrderedDict((name, self._view(self.running, shape, n, off, alloc))
                           for name, shape, n, off, alloc in self._r_entries)

    def version(self):
        return (self.params._version, self.user_version)

    def touch(self):
        """Call after a raw-pointer kernel modified `params` (bumps the repack trigger)."""
        self.user_version += 1


class _Leaf(nn.Module):
    """Parameter holder with the reference's attribute names (weight / bias / running_*)."""


class SynchronizedBatchNorm2d(_Leaf):
    """Name-compatible stand-in of pixelssl.SynchronizedBatchNorm2d: it only *holds* gamma/beta and the
    running statistics (views into the flat buffers); the normalisation itself is fused into the
    producing / consuming convolution kernels, and the cross-device statistics exchange is one RCCL
    all-reduce of [sum, sumsq] per layer (sync_batchnorm/batchnorm.py:48-78,113-125)."""
    eps = 1e-5
    momentum = 0.1


class ProgramBuilder:
    """Builds the op / BN tables for libpixelhip and the matching module tree."""

    def __init__(self, store):
        self.store = store
        self.ops = []
        self.bns = []
        self.ntensors = 0
        self.modules = OrderedDict()      # dotted name -> (_Leaf, kind)
        self.bn_names = []                # BN index -> dotted name
        self.relu_sites = OrderedDict()   # site name -> (tensor id, BN index or -1): every ReLU decision of the program

    def tensor(self):
        self.ntensors += 1
        return self.ntensors - 1

    def reserve(self, *names):
        """Fix the position of modules in the parameter order before they are built: the reference's optimizers index
        their state by the position of a parameter in `module.parameters()` (conv before its BN, resnet.py:18-27), and an
        optimizer state_dict has to load across the two implementations (ssl_mt.py:296-322)."""
        for n in names:
            self.modules.setdefault(n, None)

    def _op(self, kind, **kw):
        op = Op()
        op.kind = kind
        op.in0 = op.in1 = op.out = op.bn_in0 = op.bn_in1 = op.bn_out = -1
        for g in range(4):
            op.w_off[g] = -1
            op.b_off[g] = -1
            op.dil[g] = 1
            op.pads[g] = 0
        op.ngroups = 1
        op.kh = op.kw = op.stride = 1
        op.need_dgrad = 1
        for k, v in kw.items():
            setattr(op, k, v)
        self.ops.append(op)
        return op

    def input(self, channels):
        t = self.tensor()
        self._op(_lib.OP_INPUT, out=t, cout=channels)
        return t

    def bn(self, name, C, alloc=None):
        """`alloc` > C: the executor normalises `alloc` channels (the tensor's pitch), the module exposes the first C."""
        s = self.store
        A = (alloc,) if alloc else None
        g = s.add_param(name + ".weight", (C,), A)
        b = s.add_param(name + ".bias", (C,), A)
        rm = s.add_running(name + ".running_mean", (C,), A)
        rv = s.add_running(name + ".running_var", (C,), A)
        d = BnDesc()
        d.C, d.gamma_off, d.beta_off, d.rmean_off, d.rvar_off = alloc or C, g, b, rm, rv
        d.eps, d.momentum = SynchronizedBatchNorm2d.eps, SynchronizedBatchNorm2d.momentum
        self.bns.append(d)
        self.bn_names.append(name)
        self.modules[name] = "bn"
        return len(self.bns) - 1

    def conv(self, names, t_in, bn_in, cin, cout, k, stride, dils, pads, bias=False, bn_out=-1, need_dgrad=True,
             cin_alloc=None, cout_alloc=None, linear=False):
        """cin_alloc / cout_alloc: channel counts the executor works with when the module's are not multiples of what the
        BatchNorm kernels take (S4L's 21 / 42-channel rotation classifier): the extra weight rows / columns are zero and
        stay zero (their gradients are sums over zero activations).  linear: expose the 1x1 weight as [cout, cin]."""
        if isinstance(names, str):
            names, dils, pads = [names], [dils], [pads]
        ci, co = cin_alloc or cin, cout_alloc or cout
        t = self.tensor()
        op = self._op(_lib.OP_CONV, in0=t_in, out=t, bn_in0=bn_in, bn_out=bn_out, ngroups=len(names), cin=ci,
                      cout=co, kh=k, kw=k, stride=stride, need_dgrad=int(need_dgrad))
        for g, nm in enumerate(names):
            if linear:
                op.w_off[g] = self.store.add_param(nm + ".weight", (cout, cin), (co, ci))
            else:
                op.w_off[g] = self.store.add_param(nm + ".weight", (cout, cin, k, k), (co, ci, k, k))
            op.b_off[g] = self.store.add_param(nm + ".bias", (cout,), (co,)) if bias else -1
            op.dil[g] = dils[g]
            op.pads[g] = pads[g]
            self.modules[nm] = "conv_bias" if bias else "conv"
        if bn_in >= 0:                     # this conv consumes relu(bn(y)): the ReLU decision of that BN
            self.relu_sites[self.bn_names[bn_in]] = (t_in, bn_in)
        return t

    def act(self, t_in, slope, bn_in=-1):
        """LeakyReLU(t_in), or LeakyReLU(bn_in(t_in)) (BatchNorm without a ReLU of its own)."""
        t = self.tensor()
        self._op(_lib.OP_ACT, in0=t_in, out=t, bn_in0=bn_in, slope=float(slope))
        return t

    def ibn(self, name, t_in, C, slope, split=0.5):
        """IBNorm (+ LeakyReLU): BN half `name.bnorm` with affine parameters and running statistics, IN half
        parameter-free (ssl_gct.py:588-607)."""
        nb = int(C * split + 0.5)
        b = self.bn(name + ".bnorm", nb)
        t = self.tensor()
        self._op(_lib.OP_IBN, in0=t_in, out=t, bn_out=b, cout=C, slope=float(slope))
        return t

    def maxpool(self, t_in, bn_in):
        t = self.tensor()
        self._op(_lib.OP_MAXPOOL, in0=t_in, out=t, bn_in0=bn_in)
        if bn_in >= 0:
            self.relu_sites[self.bn_names[bn_in]] = (t_in, bn_in)
        return t

    def residual(self, t_main, bn_main, t_res, bn_res):
        t = self.tensor()
        self._op(_lib.OP_RESIDUAL, in0=t_main, in1=t_res, out=t, bn_in0=bn_main, bn_in1=bn_res)
        # out = relu(bn3(y) + shortcut): the stored tensor is the ReLU's OUTPUT, its decision is out > 0
        self.relu_sites[self.bn_names[bn_main].rsplit(".", 1)[0] + ".out"] = (t, -1)
        return t

    def avgpool(self, t_in, bins):
        t = self.tensor()
        self._op(_lib.OP_AVGPOOL, in0=t_in, out=t, kh=bins, kw=bins)
        return t

    def concat(self, t_in, cin, ctotal):
        """New tensor of `ctotal` channels whose first `cin` are a copy of t_in (torch.cat slot 0); `upcat` fills the
        remaining slices."""
        t = self.tensor()
        self._op(_lib.OP_CONCAT, in0=t_in, out=t, cin=cin, cout=ctotal)
        return t

    def upcat(self, t_in, bn_in, t_cat, c_off, cin):
        self._op(_lib.OP_UPCAT, in0=t_in, bn_in0=bn_in, out=t_cat, c_off=c_off, cin=cin)

    def pixshuf(self, t_in, cin):
        t = self.tensor()
        self._op(_lib.OP_PIXSHUF, in0=t_in, out=t, cin=cin, cout=cin // 4)
        return t

    def head(self, t_low, t_latent, bn_latent=-1, align_corners=True):
        self._op(_lib.OP_HEAD, in0=t_low, in1=t_latent, bn_in1=bn_latent, stride=int(bool(align_corners)))


def build_resnet_trunk(pb, prefix, layers, output_stride=16):
    """ResNet bottleneck trunk with atrous layer4 / multi-grid (1,2,4), as
    task/sseg/module/backbone/resnet.py:58-119.  Returns (tensor id of the feature map, channels)."""
    if output_stride == 16:
        strides, dilations = (1, 2, 2, 1), (1, 1, 1, 2)
    elif output_stride == 8:
        strides, dilations = (1, 2, 1, 1), (1, 1, 2, 4)
    else:
        raise NotImplementedError("output_stride %r" % output_stride)
    x = pb.input(3)
    pb.reserve(prefix + ".conv1")
    bn1 = pb.bn(prefix + ".bn1", 64)
    y = pb.conv(prefix + ".conv1", x, -1, 3, 64, 7, 2, 1, 3, bn_out=bn1, need_dgrad=False)
    h = pb.maxpool(y, bn1)
    cin = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers)):
        for b in range(nblk if li < 3 else 3):
            p = "%s.layer%d.%d" % (prefix, li + 1, b)
            stride = strides[li] if b == 0 else 1
            dil = dilations[li] * ((1, 2, 4)[b] if li == 3 else 1)
            pb.reserve(p + ".conv1", p + ".bn1", p + ".conv2", p + ".bn2", p + ".conv3", p + ".bn3")
            b1 = pb.bn(p + ".bn1", planes)
            y1 = pb.conv(p + ".conv1", h, -1, cin, planes, 1, 1, 1, 0, bn_out=b1)
            b2 = pb.bn(p + ".bn2", planes)
            y2 = pb.conv(p + ".conv2", y1, b1, planes, planes, 3, stride, dil, dil, bn_out=b2)
            b3 = pb.bn(p + ".bn3", planes * 4)
            y3 = pb.conv(p + ".conv3", y2, b2, planes, planes * 4, 1, 1, 1, 0, bn_out=b3)
            if b == 0 and (stride != 1 or cin != planes * 4):
                pb.reserve(p + ".downsample.0")
                bd = pb.bn(p + ".downsample.1", planes * 4)
                yd = pb.conv(p + ".downsample.0", h, -1, cin, planes * 4, 1, stride, 1, 0, bn_out=bd)
                h = pb.residual(y3, b3, yd, bd)
            else:
                h = pb.residual(y3, b3, h, -1)
            cin = planes * 4
    return h, cin


class SegNetCore(nn.Module):
    """A segmentation network executed by libpixelhip.  Holds the flat parameter store, exposes
    reference-named parameters/buffers, and runs forward/backward as single C calls."""

    def __init__(self, device, engine_dtype, num_classes):
        super().__init__()
        self._device = torch.device(device)
        self._code = _lib.dtype_code(engine_dtype)
        self.num_classes = num_classes
        self._store = FlatStore(self._device)
        self._pb = ProgramBuilder(self._store)
        self._plans = {}                # (B, H, W) -> _Plan: one executor instance + buffers per input shape
        # inference-only plans (validation feeds one image per batch at its own size, task/sseg/data.py:109-123): a small
        # LRU, untuned, no data-gradient weights -- the plan dictionary above would otherwise grow by ~0.5 GB and one
        # autotune run per distinct validation size
        self._eval_plans = OrderedDict()
        self.max_eval_plans = int(os.environ.get("PXL_EVAL_PLANS", "4"))
        self._cur = None
        self._sync_cb = None
        self._sync_world = 1
        self._profile_on = False
        self._anchor = None
        self.freeze_bn = False
        # (PXL_DETERMINISTIC=1: no timing-dependent choices either -- the tile a tuner run picks decides how the statistics of a
        # BatchNorm are grouped into partial sums, so two tuned processes need not agree in the last bits)
        self.autotune = os.environ.get("PXL_AUTOTUNE", "1") != "0" and os.environ.get("PXL_DETERMINISTIC", "0") != "1"
        self.want_prob = True           # HEAD also returns softmax(logits) (segmentation nets; not the discriminators)
        self.has_latent = True
        self.differentiable_latent = False   # SSLCCT: the latent handed out by forward() carries autograd history
        self._wgrad_on = True
        self.keep_arena = False         # inspection (parity tests): keep a handle on the last forward's arena in _last_arena
        self._last_arena = None

    # -- construction -----------------------------------------------------------------------
    def _finalize(self):
        s, pb = self._store, self._pb
        s.allocate()
        pviews, rviews = s.param_views(), s.running_views()
        self._param_list = []
        nbn = sum(1 for k in pb.modules.values() if k == "bn")
        self._nbt = torch.zeros(max(nbn, 1), dtype=torch.long, device=self._device)   # all counters, one buffer
        bn_index = 0
        for dotted, kind in pb.modules.items():
            leaf = SynchronizedBatchNorm2d() if kind == "bn" else _Leaf()
            for attr in ("weight", "bias"):
                key = dotted + "." + attr
                if key in pviews:
                    pv, gv, off, n, alloc = pviews[key]
                    prm = nn.Parameter(pv)
                    prm.grad = gv
                    prm._pxl_grad_view = gv
                    prm._pxl_flat = (s, off, n)
                    prm._pxl_alloc = alloc
                    leaf.register_parameter(attr, prm)
                    self._param_list.append(prm)
            if kind == "bn":
                leaf.register_buffer("running_mean", rviews[dotted + ".running_mean"])
                leaf.register_buffer("running_var", rviews[dotted + ".running_var"])
                leaf.register_buffer("num_batches_tracked", self._nbt[bn_index])
                bn_index += 1
            self._attach(dotted, leaf)
        self._ops_arr = (Op * len(pb.ops))(*pb.ops)
        self._bns_arr = (BnDesc * max(len(pb.bns), 1))(*pb.bns)
        self._anchor = torch.zeros((), device=self._device, requires_grad=True)

    def _attach(self, dotted, leaf):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.add_module(parts[-1], leaf)

    def train(self, mode=True):
        """nn.Module.train, plus the reference's freeze_bn semantics: `freeze_bn=True` only puts the BN modules into
        eval mode inside the constructor (deeplab_v2.py:26-27,35-40), and the first `model.train()` of `_train` switches
        them back -- with the shipped training loops `--freeze-bn True` therefore never freezes anything.  Mirrored
        rather than "fixed": numerics and running statistics have to follow the reference's."""
        self.freeze_bn = False
        return super().train(mode)

    def _all_plans(self):
        return list(self._plans.values()) + list(self._eval_plans.values())

    # the state of the plan in use (kept as attributes for the call sites / tests that read them)
    _net = property(lambda self: self._cur.net)
    _packed = property(lambda self: self._cur.packed)
    _scratch = property(lambda self: self._cur.scratch)
    _arena_bytes = property(lambda self: self._cur.arena_bytes)
    _shape = property(lambda self: self._cur.shape if self._cur is not None else None)

    # -- parameter plumbing -----------------------------------------------------------------
    @property
    def flat(self):
        return self._store

    def reset_parameters(self, generator=None):
        raise NotImplementedError

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)

    def mark_params_changed(self):
        self._store.touch()

    def load_pretrained_backbone(self, src, prefix="backbone"):
        """ResNet._load_pretrained_model (task/sseg/module/backbone/resnet.py:145-156) for the trunk under `prefix`:
        a FILE is loaded as a state dict of the trunk itself (strict, like the reference's `self.load_state_dict(
        torch.load(path))`); anything else is treated as a model-zoo URL (torch.hub cache; `file://` works offline) whose
        entries are FILTERED to the keys the trunk has -- a torchvision ResNet brings `fc.*`, which the dilated trunk
        lacks -- and loaded on top of the current values.  -> the list of keys taken."""
        own = OrderedDict((k[len(prefix) + 1:], v) for k, v in self.state_dict().items() if k.startswith(prefix + "."))
        if os.path.isfile(src):
            loaded = torch.load(src, map_location="cpu")
            missing = [k for k in own if k not in loaded]
            unexpected = [k for k in loaded if k not in own]
            if missing or unexpected:      # torch's strict load_state_dict error, for the trunk
                raise RuntimeError("Error(s) in loading state_dict for the backbone: missing %s, unexpected %s"
                                   % (missing[:4], unexpected[:4]))
            take = loaded
        else:
            loaded = torch.hub.load_state_dict_from_url(src, map_location="cpu")
            take = OrderedDict((k, v) for k, v in loaded.items() if k in own)
        bad = [k for k, v in take.items() if tuple(v.shape) != tuple(own[k].shape)]
        if bad:
            raise RuntimeError("pretrained backbone: shape mismatch for %s" % bad[:4])
        self.load_state_dict(OrderedDict((prefix + "." + k, v) for k, v in take.items()), strict=False)
        self.mark_params_changed()
        return list(take.keys())

    def ensure_grad_views(self):
        """Re-attach .grad views (a foreign optimizer may have set them to None); returns True when
        the flat gradient buffer had to be treated as fresh (and was zeroed)."""
        fresh = False
        for prm in self._param_list:
            if prm.grad is None:
                fresh = True
                break
        if fresh:
            self._store.grads.zero_()
            for prm in self._param_list:
                prm.grad = prm._pxl_grad_view
        return fresh

    # -- planning ---------------------------------------------------------------------------
    def _plan(self, B, H, W, out_size=None, inference=False):
        """Select (or create) the executor instance planned for this input shape.  Networks that see several
        batch sizes per iteration (the AdvSSL discriminator: B fake + lbs real maps) keep one plan each, so a
        backward always runs on the plan its forward used.  `out_size` = (Hout, Wout) of the HEAD when it differs
        from the input size (SSLCCT auxiliary decoders).  inference: a no-grad forward of a shape no training plan
        exists for goes to the bounded LRU of untuned forward-only plans."""
        out_size = tuple(out_size) if out_size is not None else (H, W)
        key = (B, H, W) + out_size
        pl = self._plans.get(key)
        if pl is None and inference:
            pl = self._eval_plans.get(key)
            if pl is not None:
                self._eval_plans.move_to_end(key)
        if pl is None:
            pl = _Plan((B, H, W), out_size)
            pl.inference = bool(inference)
            pb = self._pb
            check(lib().pxl_net_create(self._code, self.num_classes, self._ops_arr, len(pb.ops), self._bns_arr,
                                       len(pb.bns), pb.ntensors, ctypes.byref(pl.net)))
            check(lib().pxl_net_plan_out(pl.net, B, H, W, out_size[0], out_size[1]))
            dev = self._device
            nbytes = lib().pxl_net_packed_bytes(pl.net)
            if inference:
                # the packed-weight layout depends on the layer program, not on the input shape: forward-only plans of all
                # shapes (validation at native image sizes) read ONE copy, re-packed once per weight version
                sh = getattr(self, "_eval_shared", None)
                if sh is None or sh["packed"].numel() != nbytes:
                    sh = dict(packed=torch.empty(nbytes, device=dev, dtype=torch.uint8), version=None)
                    object.__setattr__(self, "_eval_shared", sh)
                pl.shared, pl.packed = sh, sh["packed"]
            else:
                pl.packed = torch.empty(nbytes, device=dev, dtype=torch.uint8)
            pl.scratch = torch.empty(lib().pxl_net_scratch_bytes(pl.net), device=dev, dtype=torch.uint8)
            pl.arena_bytes = lib().pxl_net_arena_bytes(pl.net)
            if self._sync_cb is not None:
                check(lib().pxl_net_set_sync(pl.net, self._sync_cb, getattr(self, "_sync_user", None), self._sync_world))
            if getattr(self, "_bn_repeat", 1) != 1:
                check(lib().pxl_net_set_bn_repeat(pl.net, self._bn_repeat))
            gs = getattr(self, "_grad_sync", None)
            if gs is not None:
                check(lib().pxl_net_set_grad_sync(pl.net, gs[0], gs[1], gs[2], gs[3], self._store.np))
            if not self._wgrad_on:
                check(lib().pxl_net_set_wgrad(pl.net, 0))
            uh = getattr(self, "_update_hook", None)
            if uh is not None and not inference:
                check(lib().pxl_net_set_update_hook(pl.net, uh[0], None, uh[1], uh[2], self._store.np))
            if self._profile_on:
                check(lib().pxl_net_profile(pl.net, 1))
            if inference:
                self._eval_plans[key] = pl
                while len(self._eval_plans) > max(1, self.max_eval_plans):
                    self._eval_plans.popitem(last=False)      # its buffers / executor go when the last handle does
            else:
                self._plans[key] = pl
        self._cur = pl
        return pl

    def _ensure_packed(self):
        pl = self._cur
        v = self._store.version()
        trainable = bool(self._param_list) and self._param_list[0].requires_grad and not pl.inference
        shared = getattr(pl, "shared", None)
        if pl.pack_dgrad != trainable:      # a no-grad network (the MT teacher) needs no transposed weight copies
            check(lib().pxl_net_set_pack_dgrad(pl.net, int(trainable)))
            pl.pack_dgrad = trainable
            pl.packed_version = None
        if shared is not None and shared["version"] == v:
            pl.packed_version = v           # another forward-only plan packed this weight version into the shared copy
        if pl.packed_version != v:
            # forward operands on this stream; the transposed data-gradient copies are first read by the backward pass, so
            # they are packed on a side stream next to the forward (event: pl.wt_ready)
            side = self._pack_stream() if (trainable and torch.is_grad_enabled()) else None
            if side is None:
                check(lib().pxl_net_pack(pl.net, ptr(self._store.params), ptr(pl.packed), stream_ptr()))
                pl.wt_ready = None
            else:
                cur = torch.cuda.current_stream()
                check(lib().pxl_net_pack_parts(pl.net, ptr(self._store.params), ptr(pl.packed), 1, stream_ptr()))
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    check(lib().pxl_net_pack_parts(pl.net, ptr(self._store.params), ptr(pl.packed), 2, stream_ptr()))
                    pl.wt_ready = torch.cuda.Event()
                    pl.wt_ready.record()
                pl.wt_waiters = set()
            pl.packed_version = v
            if shared is not None:
                shared["version"] = v
        if not pl.tuned and self.autotune and not pl.inference:
            if pl.wt_ready is not None:
                torch.cuda.current_stream().wait_event(pl.wt_ready)
            # per-shape tile selection, measured on this GPU (csrc/net.cpp: pxl_net_tune)
            if getattr(self, "tune_dual", False):
                check(lib().pxl_net_set_tune_dual(pl.net, 1))
            arena = torch.zeros(pl.arena_bytes, device=self._device, dtype=torch.uint8)
            pl.scratch.zero_()
            keep = self._store.grads.clone()
            check(lib().pxl_net_tune(pl.net, ptr(self._store.params), ptr(pl.packed), ptr(self._store.grads),
                                     ptr(arena), arena.numel(), ptr(pl.scratch), pl.scratch.numel(),
                                     stream_ptr()))
            self._store.grads.copy_(keep)
            del arena
            from . import dist as _pdist
            _pdist.align_after_tune()
        pl.tuned = True

    def _pack_stream(self):
        if not hasattr(self, "_pk_stream"):
            on = os.environ.get("PXL_PACK_STREAM", "1") != "0" and self._device.type == "cuda"
            from . import streams
            object.__setattr__(self, "_pk_stream", streams.role_stream(streams.AUX, device=self._device) if on else None)
        return self._pk_stream

    def set_sync(self, callback, world_size):
        """callback(buf_ptr:int, n:int, stream:int) -> int ; installs the SyncBN statistics hook."""
        def _cb(user, buf, n, stream):
            try:
                return int(callback(buf, n, stream) or 0)
            except Exception:       # never unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._sync_cb = _lib.ALLREDUCE_FN(_cb)
        self._sync_user = None
        self._sync_world = world_size
        for pl in self._all_plans():
            check(lib().pxl_net_set_sync(pl.net, self._sync_cb, None, world_size))

    def set_grad_sync(self, fn, user, world_size, bucket_floats):
        """Overlapped gradient exchange (csrc/net.cpp: pxl_net_set_grad_sync): fn = a ctypes pxl_allreduce_fn (in-place
        sum on the given stream), user = its context; the executor all-reduces buckets of the flat gradient buffer from
        inside pxl_net_backward.  fn = None turns it off."""
        self._grad_sync = None if fn is None else (fn, user, int(world_size), int(bucket_floats))
        for pl in self._all_plans():
            if fn is None:
                check(lib().pxl_net_set_grad_sync(pl.net, _lib.ALLREDUCE_FN(), None, 1, 0, 0))
            else:
                check(lib().pxl_net_set_grad_sync(pl.net, fn, user, int(world_size), int(bucket_floats), self._store.np))

    def set_update_hook(self, fn, bucket_floats=0, tail_floats=0):
        """Parameter update pipelined behind the backward pass (csrc/net.cpp: pxl_net_set_update_hook): fn(lo, hi, stream) is
        called from inside the backward for every finished bucket grads[lo, hi) of the flat gradient buffer, `stream` (an int) is
        the HIP stream the bucket's update belongs on.  fn = None turns it off."""
        if fn is None:
            self._update_hook = None
            for pl in self._plans.values():
                check(lib().pxl_net_set_update_hook(pl.net, _lib.UPDATE_FN(), None, 0, 0, 0))
            return

        def _cb(user, lo, hi, stream):
            try:
                fn(int(lo), int(hi), int(stream or 0))
                return 0
            except Exception:       # never unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._update_hook = (_lib.UPDATE_FN(_cb), int(bucket_floats), int(tail_floats))
        for pl in self._plans.values():
            check(lib().pxl_net_set_update_hook(pl.net, self._update_hook[0], None, int(bucket_floats), int(tail_floats), self._store.np))

    def update_buckets(self):
        return lib().pxl_net_update_buckets(self._cur.net) if self._cur is not None else 0

    def pack_range(self, plan, lo, hi, which=3):
        """re-pack the kernel-layout weights of the convolutions whose master weights lie in params[lo, hi) (current stream)"""
        check(lib().pxl_net_pack_range(plan.net, ptr(self._store.params), ptr(plan.packed), int(which), int(lo), int(hi), stream_ptr()))

    def grad_buckets(self):
        """buckets the last backward of the current plan exchanged (0 on one rank)"""
        return lib().pxl_net_grad_buckets(self._cur.net) if self._cur is not None else 0

    def twin(self):
        """A second executor front-end over the SAME parameters / gradients / running statistics with its own plans,
        buffers and flags (e.g. weight gradients off): lets a frozen copy of a network run (forward and input-gradient
        backward) on one stream while the trainable one is updated on another (AdvSSL's discriminator).  Not a
        registered sub-module: it owns no parameters."""
        t = object.__new__(type(self))
        nn.Module.__init__(t)
        for k in ("_device", "_code", "num_classes", "_store", "_pb", "_ops_arr", "_bns_arr", "_param_list", "_nbt",
                  "_anchor", "freeze_bn", "autotune", "want_prob", "has_latent", "differentiable_latent", "_profile_on"):
            object.__setattr__(t, k, getattr(self, k))
        object.__setattr__(t, "_plans", {})
        object.__setattr__(t, "_eval_plans", OrderedDict())
        object.__setattr__(t, "max_eval_plans", self.max_eval_plans)
        object.__setattr__(t, "keep_arena", False)
        object.__setattr__(t, "_last_arena", None)
        object.__setattr__(t, "_cur", None)
        object.__setattr__(t, "_sync_cb", getattr(self, "_sync_cb", None))
        object.__setattr__(t, "_sync_user", getattr(self, "_sync_user", None))
        object.__setattr__(t, "_sync_world", getattr(self, "_sync_world", 1))
        object.__setattr__(t, "_grad_sync", None)          # a twin never owns the gradient exchange
        object.__setattr__(t, "_wgrad_on", True)
        t.train(self.training)
        return t

    def set_sync_native(self, fn, user, world_size):
        """fn: a C function with the pxl_allreduce_fn signature (ctypes object), user: its context pointer -- the
        Sync-BN exchange then never leaves C (dist.py wires pxl_comm_allreduce_hook + the RCCL communicator)."""
        self._sync_cb = fn
        self._sync_user = user
        self._sync_world = world_size
        for pl in self._all_plans():
            check(lib().pxl_net_set_sync(pl.net, fn, user, world_size))

    # -- execution --------------------------------------------------------------------------
    def set_bn_repeat(self, times):
        """The following forward passes update the BatchNorm running statistics as if each ran `times` times on its batch
        (csrc/net.cpp: pxl_net_set_bn_repeat) -- for callers that replace two identical passes by one."""
        if int(times) != getattr(self, '_bn_repeat', 1):
            for pl in self._all_plans():
                check(lib().pxl_net_set_bn_repeat(pl.net, int(times)))
            self._bn_repeat = int(times)

    def set_wgrad(self, enable):
        """enable=False: backward only relays dL/dinput (a frozen discriminator inside the task model's step)."""
        if bool(enable) != self._wgrad_on:
            for pl in self._all_plans():
                check(lib().pxl_net_set_wgrad(pl.net, int(bool(enable))))
            self._wgrad_on = bool(enable)

    def _forward_raw(self, x, arena, want_prob=None, parts=None, deferred=False):
        """`parts`: NCHW tensors whose channel concatenation is the input (gathered by the input op itself: no torch.cat
        copy); `x` is then parts[0].  deferred: stop at the low-resolution logits (no full-resolution planes)."""
        if self.keep_arena:
            self._last_arena = arena
        if parts is not None and len(parts) > 1:
            srcs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
            chans = (ctypes.c_int * len(parts))(*[t.shape[1] for t in parts])
            check(lib().pxl_net_set_input_parts(self._net, len(parts), srcs, chans))
        B = x.shape[0]
        H, W = self._cur.out_size
        want_prob = self.want_prob if want_prob is None else want_prob
        logits = None if deferred else torch.empty(B, self.num_classes, H, W, device=x.device, dtype=torch.float32)
        prob = torch.empty_like(logits) if want_prob and not deferred else None
        training = self.training and not self.freeze_bn
        detour = getattr(self, "_running_detour", None) if training else None
        running = detour["tmp"] if detour is not None else self._store.running
        check(lib().pxl_net_forward(self._net, ptr(self._store.params), ptr(self._packed), ptr(running),
                                    ptr(x), ptr(logits), ptr(prob), ptr(arena), arena.numel(), int(training),
                                    stream_ptr()))
        if training:
            # every BN's num_batches_tracked is a view of this buffer; a pass that stands for `_bn_repeat` identical passes
            # (set_bn_repeat: SSLGCT's PXL_GCT_REUSE_FORWARD) counts as that many, like its running statistics
            if detour is not None:
                detour["passes"] += getattr(self, "_bn_repeat", 1)
            else:
                self._nbt += getattr(self, "_bn_repeat", 1)
        return logits, prob

    def detour_running(self):
        """The training passes from here to fold_running() write their BatchNorm running-statistics update into a zeroed side
        buffer instead of the running statistics: a pass may then run CONCURRENTLY with another training pass of this network on
        another stream (two read-modify-write updates of one buffer would race).  Every kernel computes r' = (1 - m) r + m s
        (bn.hip / the fused finalize of the convolution kernels); on r = 0 that leaves m s, and fold_running() applies
        r <- (1 - m) r + (m s) -- the update the pass would have made had it run after everything that touched r in between
        (1 ulp: the sum is rounded twice).  One pass per detour (a second one would need (1 - m) applied to the first's term)."""
        if getattr(self, "_running_detour", None) is not None:
            raise _lib.PixelHipError("detour_running: already active (fold_running() first)")
        tmp = getattr(self, "_running_tmp", None)
        if tmp is None or tmp.numel() != self._store.running.numel():
            tmp = torch.empty_like(self._store.running)
            object.__setattr__(self, "_running_tmp", tmp)
        tmp.zero_()
        object.__setattr__(self, "_running_detour", {"tmp": tmp, "passes": 0})

    def fold_running(self):
        """End of detour_running(): fold the parked update into the running statistics ON THE CURRENT STREAM (the caller has
        ordered it after the detoured pass and after every other pass that updates the running statistics)."""
        d = getattr(self, "_running_detour", None)
        object.__setattr__(self, "_running_detour", None)
        if d is None or d["passes"] == 0:
            return
        if d["passes"] != 1:
            raise _lib.PixelHipError("fold_running: %d passes ran inside one detour (one is supported)" % d["passes"])
        self._store.running.mul_(1.0 - SynchronizedBatchNorm2d.momentum).add_(d["tmp"])
        self._nbt += d["passes"]

    def side_backward_buffers(self, plan=None):
        """-> (gradient buffer, scratch) of their own for a backward pass of this network that runs CONCURRENTLY with another backward
        pass of it (SSLCCT: the labeled pass beside the unlabeled one; both would otherwise accumulate into one flat gradient buffer
        and share the plan's scratch).  The caller zeroes the gradient buffer, brackets the pass's backward with `_alt_backward =
        (grads, scratch)` / None and adds the buffer onto the parameters' gradients once both passes are done."""
        pl = plan if plan is not None else self._cur
        g = getattr(self, "_alt_grads", None)
        if g is None or g.numel() != self._store.grads.numel():
            g = torch.zeros_like(self._store.grads)
            object.__setattr__(self, "_alt_grads", g)
        if getattr(pl, "scratch_alt", None) is None or pl.scratch_alt.numel() != pl.scratch.numel():
            pl.scratch_alt = torch.empty_like(pl.scratch)
        return g, pl.scratch_alt

    def _bn_leaves(self):
        if not hasattr(self, "_bn_cache"):
            self._bn_cache = [m for m in self.modules() if isinstance(m, SynchronizedBatchNorm2d)]
        return self._bn_cache

    def forward(self, x, out_size=None):
        """x: NCHW fp32 on the GPU -> (logits, softmax, latent_fn) with autograd attached."""
        parts = tuple(x) if isinstance(x, (tuple, list)) else (x,)     # several tensors = their concatenation along C
        for t in parts:
            if not t.is_cuda:
                raise _lib.PixelHipError("SegNetCore runs on the GPU only (input is on %s); there is no CPU path" % t.device)
        parts = tuple(t.contiguous().float() for t in parts)
        if len(parts) > 4 or any(t.shape[0] != parts[0].shape[0] or t.shape[2:] != parts[0].shape[2:] for t in parts):
            raise ValueError("SegNetCore: at most 4 input parts of the same batch and spatial size")
        x = parts[0]
        B, _, H, W = x.shape
        need_graph = torch.is_grad_enabled() and (any(p.requires_grad for p in self._param_list[:1]) or
                                                  any(t.requires_grad for t in parts))
        self._plan(B, H, W, out_size, inference=not need_graph and not self.training)
        self._ensure_packed()
        if need_graph and self.differentiable_latent and self.has_latent:
            arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
            logits, prob, latent = _SegNetFn.apply(x, self._anchor, self, arena, True, *parts[1:])
            return logits, prob, (lambda: latent)
        if need_graph:
            arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
            logits, prob, _ = _SegNetFn.apply(x, self._anchor, self, arena, False, *parts[1:])
        else:
            if self._cur.eval_arena is None:
                self._cur.eval_arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
            arena = self._cur.eval_arena
            self._cur.arena_gen = getattr(self._cur, "arena_gen", 0) + 1      # (an earlier no-grad DeferredHead on this arena is stale now)
            logits, prob = self._forward_raw(x, arena, parts=parts)
        return logits, prob, (_LatentHandle(self, arena, self._cur) if self.has_latent else None)

    def seam_supported(self, x):
        """can forward_deferred + functional.head_losses run for this input (shape)?  Plans the shape, runs nothing."""
        if not x.is_cuda or x.dim() != 4:
            return False
        B, _, H, W = x.shape
        need_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list[:1])
        self._plan(B, H, W, None, inference=not need_graph and not self.training)
        return bool(lib().pxl_net_head_loss_supported(self._cur.net))

    def prepare_patches(self, x):
        """Write the stem's im2col patches for `x` NOW, on the current stream, into the arena the next forward_deferred(x,
        prepared=...) of this network will use (csrc/net.cpp: pxl_net_make_patches) -> a token (arena, patches address, bytes)
        that another network with the same stem can borrow (borrow_patches), or None when this plan has no patch-mode stem.
        Mean Teacher without input noise feeds both networks ONE tensor: one 203 MB patch tensor instead of two."""
        if not x.is_cuda:
            return None
        x = x.contiguous().float()
        B, _, H, W = x.shape
        need_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list[:1])
        self._plan(B, H, W, None, inference=not need_graph and not self.training)
        if not lib().pxl_net_head_loss_supported(self._cur.net):
            return None
        if need_graph:
            arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
        else:
            if self._cur.eval_arena is None:
                self._cur.eval_arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
            arena = self._cur.eval_arena
        pp, nb = ctypes.c_void_p(), ctypes.c_size_t()
        rc = lib().pxl_net_make_patches(self._net, ptr(x), ptr(arena), arena.numel(), ctypes.byref(pp), ctypes.byref(nb), stream_ptr())
        if rc != 0:
            return None                     # (PXL_ERR_UNSUPPORTED: a plan without a patch-mode stem -- nothing was written)
        return (arena, pp.value, nb.value, x, self._cur)

    def borrow_patches(self, token, x):
        """The NEXT forward of this network reads its stem operand from another network's patches (prepare_patches token) of the
        same input tensor; False when the geometry does not match (the pass then writes its own)."""
        if token is None or token[3].data_ptr() != x.data_ptr() or token[3].shape != x.shape:
            return False
        B, _, H, W = x.shape
        need_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list[:1])
        self._plan(B, H, W, None, inference=not need_graph and not self.training)
        return lib().pxl_net_borrow_patches(self._net, ctypes.c_void_p(token[1]), token[2]) == 0

    def forward_deferred(self, x, prepared=None, out_size=None, force_graph=False, seam="head"):
        """Forward pass up to the LOW-RESOLUTION logits: the up-sampling / soft-max op is not run and no full-resolution
        plane is written.  -> DeferredHead, which pixelssl_amd.functional.head_losses consumes (criterion + consistency
        term + their backward on the low-resolution maps, csrc/head.hip) and whose .backward() runs the executor's
        backward from the gradient that call left behind.  A consumer that wants the planes after all calls
        .materialize().  None when this plan cannot run the fused seam (the caller then uses forward()).
        out_size: the HEAD's output size when it is not the input size (SSLCCT's auxiliary decoders); force_graph: a pass that
        will be differentiated although grad mode is off here (the caller is an autograd Function's forward); seam: which fused
        seam must be able to run on the plan -- "head" (functional.head_losses) or "cons" (functional.decoder_consistency)."""
        if not x.is_cuda:
            raise _lib.PixelHipError("SegNetCore runs on the GPU only (input is on %s); there is no CPU path" % x.device)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        need_graph = bool(force_graph) or (torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list[:1]))
        self._plan(B, H, W, out_size, inference=not need_graph and not self.training)
        ok = lib().pxl_net_cons_head_supported(self._cur.net) if seam == "cons" else lib().pxl_net_head_loss_supported(self._cur.net)
        if not ok:
            return None
        self._ensure_packed()
        if prepared is not None and prepared[4] is self._cur and prepared[3].data_ptr() == x.data_ptr():
            arena = prepared[0]              # (prepare_patches allocated it and wrote the stem patches into it)
        elif need_graph:
            arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
        else:
            if self._cur.eval_arena is None:
                self._cur.eval_arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
            arena = self._cur.eval_arena
        self._forward_raw(x, arena, deferred=True)
        return DeferredHead(self, arena, self._cur, B, bool(self.training and not self.freeze_bn), need_graph)

    def forward_with_latent(self, x):
        """-> (logits, softmax, latent) where the latent (NCHW fp32) is part of the autograd graph: a gradient that
        reaches it (SSLCCT's auxiliary decoders, which consume 'sslcct_ad_inp' outside this program) is seeded into the
        executor's backward next to dlogits / dprob."""
        if not x.is_cuda:
            raise _lib.PixelHipError("SegNetCore runs on the GPU only (input is on %s); there is no CPU path" % x.device)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        self._plan(B, H, W)
        self._ensure_packed()
        arena = torch.empty(self._arena_bytes, device=x.device, dtype=torch.uint8)
        if torch.is_grad_enabled():
            return _SegNetFn.apply(x, self._anchor, self, arena, True)
        logits, prob = self._forward_raw(x, arena)
        return logits, prob, self.latent_from(arena)

    def profile(self, enable=True):
        """Bracket every contraction launch with HIP events (bench.py roofline leg)."""
        self._profile_on = bool(enable)
        for pl in self._all_plans():
            check(lib().pxl_net_profile(pl.net, int(enable)))

    def profile_read(self, kind):
        """-> (kernel ms, launches, algorithmic flops) of kind 0 = conv igemm (fwd+dgrad), 1 = wgrad."""
        tot = [0.0, 0, 0.0]
        for pl in self._all_plans():
            ms, n, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
            check(lib().pxl_net_profile_read(pl.net, kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
            tot = [tot[0] + ms.value, tot[1] + n.value, tot[2] + fl.value]
        return tuple(tot)

    def profile_bytes(self, kind):
        """algorithmic operand bytes of the stamped launches of `kind` since the last call"""
        tot = 0.0
        for pl in self._all_plans():
            b = ctypes.c_double()
            check(lib().pxl_net_profile_bytes(pl.net, kind, ctypes.byref(b)))
            tot += b.value
        return tot

    def relu_decisions(self, arena, plan=None):
        """name -> bool NCHW tensor of every ReLU decision of the forward pass held in `arena` (inspection for the
        parity tests: '<bn name>' = sign of that BatchNorm's output, '<block>.out' = sign of a bottleneck's output)."""
        pl = plan if plan is not None else self._cur
        out = OrderedDict()
        for name, (tid, bn) in self._pb.relu_sites.items():
            c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib().pxl_net_tensor_shape(pl.net, tid, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
            v = torch.empty(pl.shape[0], c.value, h.value, w.value, device=self._device, dtype=torch.float32)
            tmp = torch.empty(lib().pxl_net_tensor_bytes(pl.net, tid), device=self._device, dtype=torch.uint8) if bn >= 0 else None
            check(lib().pxl_net_read_tensor(pl.net, ptr(arena), tid, bn, ptr(tmp), ptr(v), stream_ptr()))
            out[name] = v > 0
        return out

    def latent_from(self, arena, plan=None):
        pl = plan if plan is not None else self._cur
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().pxl_net_latent_shape(pl.net, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
        out = torch.empty(pl.shape[0], c.value, h.value, w.value, device=self._device, dtype=torch.float32)
        check(lib().pxl_net_latent(pl.net, ptr(arena), ptr(out), stream_ptr()))
        return out


class _Plan:
    """One executor instance (csrc/net.cpp pxl_net) planned for a fixed input shape, with its buffers."""

    def __init__(self, shape, out_size=None):
        self.shape = shape
        self.out_size = tuple(out_size) if out_size is not None else tuple(shape[1:])
        self.net = ctypes.c_void_p()
        self.packed = self.scratch = self.eval_arena = None
        self.packed_version = None
        self.shared = None           # forward-only plans: the record of the packed-weight copy they share
        self.pack_dgrad = True
        self.arena_bytes = 0
        self.tuned = False
        self.inference = False
        self.wt_ready = None         # event: the data-gradient weight copies of `packed` are complete (side-stream pack)

    def __del__(self):
        try:
            if self.net:
                lib().pxl_net_destroy(self.net)
                self.net = ctypes.c_void_p()
        except Exception:
            pass


class DeferredHead:
    """A forward pass that stopped at the low-resolution logits (SegNetCore.forward_deferred)."""

    def __init__(self, core, arena, plan, batch, bn_training, trainable):
        self.core, self.arena, self.plan, self.batch = core, arena, plan, batch
        self.bn_training, self.trainable = bn_training, trainable
        self.has_grad = False           # functional.head_losses wrote d(loss)/d(low-res logits) into the plan's scratch
        # The low-resolution gradient lives in the PLAN's shared scratch and a no-grad pass lives in the plan's reused
        # eval arena: any later pass / head_losses / backward on the same plan overwrites them.  Stamps of the plan's
        # generation counters at the time this pass (and its gradient) were written make a stale use an error instead of
        # a silently wrong gradient / prediction.
        plan.arena_gen = getattr(plan, "arena_gen", 0) + (0 if trainable else 1)
        self._arena_gen = plan.arena_gen if not trainable else None
        self._grad_gen = None

    def _check_arena(self, what):
        if self._arena_gen is not None and self._arena_gen != getattr(self.plan, "arena_gen", 0):
            raise _lib.PixelHipError("DeferredHead.%s: this no-grad pass has been overwritten by a later pass on the same plan "
                                     "(its arena is the plan's reused evaluation arena); materialize() it before the next pass" % what)

    def mark_grad(self):
        """functional.head_losses wrote d(loss)/d(low-res logits) of THIS pass into the plan's scratch"""
        self.plan.grad_gen = getattr(self.plan, "grad_gen", 0) + 1
        self._grad_gen = self.plan.grad_gen
        self.has_grad = True

    def materialize(self, want_prob=True):
        """-> (logits, softmax) NCHW fp32 at full resolution, detached (what forward() would have returned)."""
        core = self.core
        self._check_arena("materialize")
        H, W = self.plan.out_size
        logits = torch.empty(self.batch, core.num_classes, H, W, device=core._device, dtype=torch.float32)
        prob = torch.empty_like(logits) if want_prob else None
        check(lib().pxl_net_head_forward(self.plan.net, ptr(self.arena), ptr(logits), ptr(prob), stream_ptr()))
        return logits, prob

    def backward(self):
        """The executor's backward pass from the low-resolution gradient (parameter gradients accumulate into the flat
        gradient buffer, exactly like the autograd path of forward())."""
        if not (self.trainable and self.has_grad):
            raise _lib.PixelHipError("DeferredHead.backward: no gradient to propagate (no-grad pass, or head_losses not called)")
        if self._grad_gen != getattr(self.plan, "grad_gen", 0):
            raise _lib.PixelHipError("DeferredHead.backward: the low-resolution gradient of this pass has been overwritten (another "
                                     "head_losses / backward ran on the same plan in between)")
        core, pl = self.core, self.plan
        core.ensure_grad_views()
        s = core._store
        if pl.wt_ready is not None:
            torch.cuda.current_stream().wait_event(pl.wt_ready)
        check(lib().pxl_net_backward_low(pl.net, ptr(s.params), ptr(pl.packed), ptr(s.grads), ptr(self.arena), self.arena.numel(),
                                         ptr(pl.scratch), pl.scratch.numel(), int(self.bn_training), stream_ptr()))
        hook = getattr(core, "_post_backward_hook", None)
        if hook is not None and core._wgrad_on:
            hook(core)
        self.has_grad = False
        pl.grad_gen = getattr(pl, "grad_gen", 0) + 1        # (the backward pass used the scratch: nothing in it is a seam gradient now)


def forward_deferred_pair(core_a, xa, core_b, xb):
    """SegNetCore.forward_deferred of TWO networks that run the same program (Mean Teacher's student and teacher, GCT's l and
    r task models) as ONE lockstep executor pass on the current stream: every convolution of the pair is one launch
    (csrc/net.cpp: pxl_net_forward_pair).  -> (DeferredHead a, DeferredHead b), each exactly what its own forward_deferred
    would have returned (bit-identical tensors), or None when the pair cannot run this way (different programs / shapes, a
    plan without the fused seam) -- the caller then runs the two passes separately."""
    if type(core_a) is not type(core_b) or core_a is core_b or not (xa.is_cuda and xb.is_cuda) or tuple(xa.shape) != tuple(xb.shape):
        return None
    if len(core_a._pb.ops) != len(core_b._pb.ops) or core_a._code != core_b._code or core_a._profile_on or core_b._profile_on:
        return None
    heads, plans, arenas, flags, xs = [], [], [], [], []
    for core, x in ((core_a, xa), (core_b, xb)):
        x = x.contiguous().float()
        B, _, H, W = x.shape
        need_graph = torch.is_grad_enabled() and any(p.requires_grad for p in core._param_list[:1])
        core._plan(B, H, W, None, inference=not need_graph and not core.training)
        if not lib().pxl_net_head_loss_supported(core._cur.net):
            return None
        core._ensure_packed()
        if need_graph:
            arena = torch.empty(core._arena_bytes, device=x.device, dtype=torch.uint8)
        else:
            if core._cur.eval_arena is None:
                core._cur.eval_arena = torch.empty(core._arena_bytes, device=x.device, dtype=torch.uint8)
            arena = core._cur.eval_arena
        if core.keep_arena:
            core._last_arena = arena
        plans.append(core._cur); arenas.append(arena); xs.append(x)
        flags.append((bool(core.training and not core.freeze_bn), need_graph))
    pa, pb = plans
    if core_a.autotune and getattr(pa, "pair_tuned_with", None) is not pb.net and not pa.inference and not pb.inference:
        # tile configuration of every paired launch, measured (csrc/net.cpp: pxl_net_tune_pair); scratch arenas: warm-up only
        tmp = [torch.zeros(c._arena_bytes, device=xs[0].device, dtype=torch.uint8) for c in (core_a, core_b)]
        check(lib().pxl_net_tune_pair(pa.net, pb.net, ptr(core_a._store.params), ptr(core_b._store.params), ptr(pa.packed), ptr(pb.packed),
                                      ptr(tmp[0]), ptr(tmp[1]), tmp[0].numel(), tmp[1].numel(), stream_ptr()))
        del tmp
        pa.pair_tuned_with = pb.net
        from . import dist as _pdist
        _pdist.align_after_tune()
    check(lib().pxl_net_forward_pair(pa.net, pb.net, ptr(core_a._store.params), ptr(core_b._store.params), ptr(pa.packed), ptr(pb.packed),
                                     ptr(core_a._store.running), ptr(core_b._store.running), ptr(xs[0]), ptr(xs[1]), None, None, None, None,
                                     ptr(arenas[0]), ptr(arenas[1]), arenas[0].numel(), arenas[1].numel(), int(flags[0][0]), int(flags[1][0]),
                                     stream_ptr()))
    for core, (training, _) in zip((core_a, core_b), flags):
        if training:
            core._nbt += getattr(core, "_bn_repeat", 1)
    for core, pl, arena, x, (training, need_graph) in zip((core_a, core_b), plans, arenas, xs, flags):
        heads.append(DeferredHead(core, arena, pl, x.shape[0], training, need_graph))
    return heads[0], heads[1]


class _LatentHandle:
    def __init__(self, core, arena, plan=None):
        self.core, self.arena, self.plan = core, arena, plan

    def __call__(self):
        return self.core.latent_from(self.arena, self.plan)


class _SegNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, core, arena, with_latent, *more_parts):
        parts = (x,) + tuple(more_parts)
        logits, prob = core._forward_raw(x, arena, parts=parts)
        ctx.core, ctx.arena, ctx.plan = core, arena, core._cur
        ctx.bn_training = bool(core.training and not core.freeze_bn)
        ctx.x_shape = tuple(x.shape)
        ctx.part_shapes = [tuple(t.shape) for t in parts]
        ctx.save_for_backward(prob)
        ctx.set_materialize_grads(False)
        latent = core.latent_from(arena) if with_latent else None
        return logits, prob, latent

    @staticmethod
    def backward(ctx, dlogits, dprob, dlatent):
        core = ctx.core
        (prob,) = ctx.saved_tensors
        nextra = len(ctx.part_shapes) - 1
        if dlogits is None and dprob is None and dlatent is None:
            return (None,) * (5 + nextra)
        if dlogits is None and dprob is None:           # only the latent carries a gradient
            dlogits = torch.zeros((ctx.x_shape[0], core.num_classes) + ctx.plan.out_size, device=core._device)
        if dlogits is not None:
            dlogits = dlogits.contiguous()
        if dprob is not None:
            dprob = dprob.contiguous()
        core.ensure_grad_views()
        s = core._store
        pl = ctx.plan
        if pl.wt_ready is not None:
            torch.cuda.current_stream().wait_event(pl.wt_ready)
        # (side_backward_buffers: a pass that runs beside another backward pass of this network has its own gradient buffer and scratch)
        alt = getattr(core, "_alt_backward", None)
        grads, scratch = (alt if alt is not None else (s.grads, pl.scratch))
        if dlatent is not None:
            dlatent = dlatent.contiguous().float()
            check(lib().pxl_net_seed_latent_grad(pl.net, ptr(scratch), scratch.numel(), ptr(dlatent), stream_ptr()))
        if alt is None:
            pl.grad_gen = getattr(pl, "grad_gen", 0) + 1      # (a seam gradient parked in this plan's scratch is gone after this pass)
        check(lib().pxl_net_backward(pl.net, ptr(s.params), ptr(pl.packed), ptr(dlogits), ptr(dprob), ptr(prob),
                                     ptr(grads), ptr(ctx.arena), ctx.arena.numel(), ptr(scratch),
                                     scratch.numel(), int(ctx.bn_training), stream_ptr()))
        dx, dextra = None, [None] * nextra
        want = [ctx.needs_input_grad[0]] + [ctx.needs_input_grad[5 + k] for k in range(nextra)]
        if nextra and any(want):           # concatenated input: one NCHW gradient per part that asks for it
            outs = [torch.empty(shp, device=core._device, dtype=torch.float32) if w else None
                    for shp, w in zip(ctx.part_shapes, want)]
            dsts = (ctypes.c_void_p * len(outs))(*[t.data_ptr() if t is not None else None for t in outs])
            chans = (ctypes.c_int * len(outs))(*[shp[1] for shp in ctx.part_shapes])
            check(lib().pxl_net_input_grad_parts(pl.net, ptr(scratch), len(outs), dsts, chans, stream_ptr()))
            dx, dextra = outs[0], outs[1:]
        elif ctx.needs_input_grad[0]:      # discriminator / flaw detector: gradient w.r.t. the task model's softmax
            dx = torch.empty(ctx.x_shape, device=core._device, dtype=torch.float32)
            check(lib().pxl_net_input_grad(pl.net, ptr(scratch), ptr(dx), stream_ptr()))
        hook = getattr(core, "_post_backward_hook", None)
        if hook is not None and core._wgrad_on:
            hook(core)
        ctx.arena = None
        return (dx, None, None, None, None) + tuple(dextra)


class DeepLabV2Core(SegNetCore):
    """DeepLab-v2 = ResNet trunk + ASPP (sum of 4 dilated 3x3 convs with bias) + bilinear up-sampling to
    the input size (task/sseg/module/deeplab_v2.py:13-33,71-85).  Parameter names follow the reference
    module tree: backbone.*, classifier.conv2d_list.{0..3}."""

    def __init__(self, backbone="resnet101", output_stride=16, num_classes=21, device="cuda",
                 engine_dtype=torch.float32, freeze_bn=False):
        super().__init__(device, engine_dtype, num_classes)
        if isinstance(backbone, (tuple, list)):
            layers = tuple(backbone)           # blocks per stage (tests use shallow trunks)
        elif backbone in RESNET_LAYERS:
            layers = RESNET_LAYERS[backbone]
        else:
            raise NotImplementedError("backbone %r" % backbone)
        pb = self._pb
        feat, c = build_resnet_trunk(pb, "backbone", layers, output_stride)
        names = ["classifier.conv2d_list.%d" % i for i in range(len(ASPP_RATES))]
        low = pb.conv(names, feat, -1, c, num_classes, 3, 1, list(ASPP_RATES), list(ASPP_RATES), bias=True)
        pb.head(low, feat)
        self._finalize()
        self.freeze_bn = freeze_bn
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        """Reference initialisers: conv ~ N(0, sqrt(2/(k*k*cout))), BN gamma=1/beta=0 (resnet.py:133-143);
        ASPP weights ~ N(0, 0.01) and torch's default Conv2d bias init (deeplab_v2.py:76-79)."""
        import math
        for name, prm in self.named_parameters():
            if name.startswith("classifier"):
                if name.endswith("weight"):
                    prm.copy_(torch.randn(prm.shape, generator=generator) * 0.01)
                else:
                    bound = 1.0 / math.sqrt(2048 * 9)
                    prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)
            elif prm.dim() == 4:
                n = prm.shape[2] * prm.shape[3] * prm.shape[0]
                prm.copy_(torch.randn(prm.shape, generator=generator) * math.sqrt(2.0 / n))
            elif name.endswith("weight"):
                prm.fill_(1.0)
            else:
                prm.zero_()
        for name, buf in self.named_buffers():
            if name.endswith("running_var"):
                buf.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                buf.zero_()

    def get_1x_lr_params(self):
        for name, prm in self.named_parameters():
            if name.startswith("backbone") and prm.requires_grad:
                yield prm

    def get_10x_lr_params(self):
        for name, prm in self.named_parameters():
            if name.startswith("classifier") and prm.requires_grad:
                yield prm


PSP_BINS = (1, 2, 3, 6)               # task/sseg/module/_pspnet.py:117


def build_subpixel_decoder(pb, prefix, t_in, bn_in, cin, num_classes, upscale=8):
    """`upsample(cin, num_classes, upscale)` of task/sseg/module/_pspnet.py:15-24: 1x1 conv without bias, then
    log2(upscale) x [1x1 conv n -> 4n with bias, ReLU, PixelShuffle(2)].  Returns the tensor id of the result."""
    t = pb.conv(prefix + ".0", t_in, bn_in, cin, num_classes, 1, 1, 1, 0)
    steps = int(round(math.log(upscale, 2)))
    for i in range(1, steps + 1):
        y = pb.conv("%s.%d.conv" % (prefix, i), t, -1, num_classes, num_classes * 4, 1, 1, 1, 0, bias=True)
        t = pb.pixshuf(y, num_classes * 4)
    return t


@torch.no_grad()
def init_subpixel_decoder(named_params, prefix, generator=None):
    """kaiming_normal(relu) for the 1x1 conv, ICNR for the PixelShuffle convs (the 4 sub-pixel rows of one output
    channel share a kaiming_normal row), torch's default bias (task/sseg/module/_pspnet.py:18-19, 26-38, 47-48)."""
    for name, prm in named_params:
        if not name.startswith(prefix + "."):
            continue
        if name == prefix + ".0.weight":
            prm.copy_(torch.randn(prm.shape, generator=generator) * math.sqrt(2.0 / prm.shape[1]))
        elif name.endswith("conv.weight"):
            base = torch.randn(prm.shape[0] // 4, prm.shape[1], 1, 1, generator=generator) * math.sqrt(2.0 / prm.shape[1])
            prm.copy_(base.repeat_interleave(4, dim=0))
        elif name.endswith("conv.bias"):
            bound = 1.0 / math.sqrt(prm.shape[0] // 4)
            prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)


class PSPNetCore(SegNetCore):
    """PSPNet = ResNet trunk + pyramid pooling module + sub-pixel decoder + bilinear up-sampling to the input size
    (task/sseg/module/_pspnet.py:58-129).  The pyramid stages are [adaptive avg-pool to 1/2/3/6 bins -> 1x1 conv ->
    BN -> ReLU -> bilinear (align_corners=False)] written straight into their channel slice of the 4096-channel
    concat tensor; the 3x3 4096->512 bottleneck (+BN+ReLU) output is the latent the CCT auxiliary decoders consume.
    Parameter names follow the reference module tree: backbone.*, psp.stages.{i}.{1,2}, psp.bottleneck.{0,1},
    decoder.0, decoder.{1..3}.conv."""

    def __init__(self, backbone="resnet101", output_stride=16, num_classes=21, device="cuda",
                 engine_dtype=torch.float32, freeze_bn=False):
        super().__init__(device, engine_dtype, num_classes)
        if isinstance(backbone, (tuple, list)):
            layers = tuple(backbone)
        elif backbone in RESNET_LAYERS:
            layers = RESNET_LAYERS[backbone]
        else:
            raise NotImplementedError("backbone %r" % backbone)
        pb = self._pb
        feat, c = build_resnet_trunk(pb, "backbone", layers, output_stride)
        oc = c // len(PSP_BINS)
        cat = pb.concat(feat, c, c + oc * len(PSP_BINS))
        for i, bins in enumerate(PSP_BINS):
            pooled = pb.avgpool(feat, bins)
            pb.reserve("psp.stages.%d.1" % i)
            b = pb.bn("psp.stages.%d.2" % i, oc)
            y = pb.conv("psp.stages.%d.1" % i, pooled, -1, c, oc, 1, 1, 1, 0, bn_out=b)
            pb.upcat(y, b, cat, c + i * oc, oc)
        pb.reserve("psp.bottleneck.0")
        bb = pb.bn("psp.bottleneck.1", oc)
        px = pb.conv("psp.bottleneck.0", cat, -1, c + oc * len(PSP_BINS), oc, 3, 1, 1, 1, bn_out=bb)
        low = build_subpixel_decoder(pb, "decoder", px, bb, oc, num_classes, upscale=8)
        pb.head(low, px, bn_latent=bb)
        self._finalize()
        self.freeze_bn = freeze_bn
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        """Backbone: conv ~ N(0, sqrt(2/(k*k*cout))), BN gamma=1/beta=0 (resnet.py:133-143); psp convs
        kaiming_uniform(fan_in, relu) (_pspnet.py:76-80); decoder: see init_subpixel_decoder."""
        for name, prm in self.named_parameters():
            if name.startswith("decoder"):
                continue
            if prm.dim() == 4 and name.startswith("psp"):
                fan_in = prm.shape[1] * prm.shape[2] * prm.shape[3]
                bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)
            elif prm.dim() == 4:
                n = prm.shape[2] * prm.shape[3] * prm.shape[0]
                prm.copy_(torch.randn(prm.shape, generator=generator) * math.sqrt(2.0 / n))
            elif name.endswith("weight"):
                prm.fill_(1.0)
            else:
                prm.zero_()
        init_subpixel_decoder(self.named_parameters(), "decoder", generator)
        for name, buf in self.named_buffers():
            if name.endswith("running_var"):
                buf.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                buf.zero_()

    def get_backbone_params(self):
        return (p for n, p in self.named_parameters() if n.startswith("backbone") and p.requires_grad)

    def get_psp_params(self):
        return (p for n, p in self.named_parameters() if n.startswith("psp") and p.requires_grad)

    def get_decoder_params(self):
        return (p for n, p in self.named_parameters() if n.startswith("decoder") and p.requires_grad)


class AuxDecoderCore(SegNetCore):
    """One SSLCCT auxiliary decoder body: `upsample(in_channels, num_classes, upscale)` (ssl_cct.py:524-532) on the
    encoder latent, followed by the bilinear (align_corners=False) resize + soft-max that WrappedCCTModel applies to
    every auxiliary prediction (ssl_cct.py:483-484) -- the resize target is the `out_size` of the call.  The input is
    the (perturbed) latent, NCHW fp32, and its gradient is returned to autograd.  Parameter names: 0, {1..}.conv (the
    module is attached as `.upsample` of the decoder, so checkpoints read auxiliary_decoders.k.upsample.*)."""

    def __init__(self, upscale, in_channels, num_classes, device="cuda", engine_dtype=torch.float32):
        super().__init__(device, engine_dtype, num_classes)
        self.has_latent = False
        self.upscale = upscale
        pb = self._pb
        x = pb.input(in_channels)
        low = self._build(pb, x, in_channels, num_classes, upscale)
        pb.head(low, -1, align_corners=False)
        self._finalize()
        self.reset_parameters()

    @staticmethod
    def _build(pb, x, in_channels, num_classes, upscale):
        t = pb.conv("0", x, -1, in_channels, num_classes, 1, 1, 1, 0)
        for i in range(1, int(round(math.log(upscale, 2))) + 1):
            y = pb.conv("%d.conv" % i, t, -1, num_classes, num_classes * 4, 1, 1, 1, 0, bias=True)
            t = pb.pixshuf(y, num_classes * 4)
        return t

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        for name, prm in self.named_parameters():
            if name == "0.weight":
                prm.copy_(torch.randn(prm.shape, generator=generator) * math.sqrt(2.0 / prm.shape[1]))
            elif name.endswith("conv.weight"):
                base = torch.randn(prm.shape[0] // 4, prm.shape[1], 1, 1, generator=generator) * math.sqrt(2.0 / prm.shape[1])
                prm.copy_(base.repeat_interleave(4, dim=0))
            elif name.endswith("conv.bias"):
                bound = 1.0 / math.sqrt(prm.shape[0] // 4)
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)


class FCDiscriminatorCore(SegNetCore):
    """FC discriminator of AdvSSL (pixelssl/ssl_algorithm/ssl_adv.py:463-493): 4 x (conv 4x4 / stride 2 / pad 1 + bias
    + LeakyReLU 0.2), a 1-channel classifier conv and bilinear up-sampling (align_corners=True) to the input size --
    the same layer program / executor as the segmentation networks, no BatchNorm.  Parameter names follow the
    reference module: conv1..conv4, classifier.  The input is the task model's softmax (NCHW fp32) and its gradient
    is returned to autograd (the adversarial loss trains the task model through the discriminator)."""
    ndf = 64

    def __init__(self, in_channels, device="cuda", engine_dtype=torch.float32):
        super().__init__(device, engine_dtype, 1)
        self.want_prob = False
        self.has_latent = False
        pb = self._pb
        t = pb.input(in_channels)
        cin = in_channels
        for i, mult in enumerate((1, 2, 4, 8)):
            t = pb.conv("conv%d" % (i + 1), t, -1, cin, self.ndf * mult, 4, 2, 1, 1, bias=True, need_dgrad=True)
            t = pb.act(t, 0.2)
            cin = self.ndf * mult
        low = pb.conv("classifier", t, -1, cin, 1, 4, 2, 1, 1, bias=True)
        pb.head(low, -1)
        self._finalize()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        """torch's default nn.Conv2d initialisation (kaiming_uniform(a=sqrt(5)) weights, uniform bias)."""
        import math
        for name, prm in self.named_parameters():
            if name.endswith("weight"):
                fan_in = prm.shape[1] * prm.shape[2] * prm.shape[3]
                bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)
                self._last_fan_in = fan_in
            else:
                bound = 1.0 / math.sqrt(self._last_fan_in)
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)


class RotationClassifierCore(SegNetCore):
    """Rotation classifier of S4L (pixelssl/ssl_algorithm/ssl_s4l.py:371-393): conv 4x4 / s2 (C -> C) + BatchNorm +
    LeakyReLU 0.2, conv 4x4 / s2 (C -> 2C) + BatchNorm + LeakyReLU 0.2, global average pool, Linear(2C -> 4) -- on the same
    layer-program executor: the Linear is a 1x1 convolution over the pooled 1x1 map, the HEAD resizes 1x1 -> 1x1.  C = 21
    and 2C = 42 are no channel counts the BatchNorm kernels take (multiples of 8, equal to the tensor pitch): the executor
    works on 32 / 64 channels whose extra weight rows, BatchNorm slots and activations are zero and stay zero; the
    module exposes the reference's shapes (conv1.weight [21, 21, 4, 4], bn2.weight [42], classifier.weight [4, 42]).
    Input: the task model's `pred` (NCHW fp32); its gradient is returned to autograd."""

    def __init__(self, in_channels, device="cuda", engine_dtype=torch.float32):
        super().__init__(device, engine_dtype, 4)
        self.want_prob = False
        self.has_latent = False
        # the reference's rotation classifier uses plain nn.BatchNorm2d (ssl_s4l.py:376-383): per-replica batch statistics,
        # never synchronised across GPUs -- dist.attach() leaves the Sync-BN hook of this executor uninstalled
        self.sync_bn = False
        c = in_channels
        pad32 = lambda v: (v + 31) // 32 * 32
        c1, c2 = pad32(c), pad32(2 * c)
        pb = self._pb
        t = pb.input(c)
        pb.reserve("conv1")
        b1 = pb.bn("bn1", c, alloc=c1)
        y1 = pb.conv("conv1", t, -1, c, c, 4, 2, 1, 1, bias=True, bn_out=b1, need_dgrad=True, cout_alloc=c1)
        a1 = pb.act(y1, 0.2, bn_in=b1)
        pb.reserve("conv2")
        b2 = pb.bn("bn2", 2 * c, alloc=c2)
        y2 = pb.conv("conv2", a1, -1, c, 2 * c, 4, 2, 1, 1, bias=True, bn_out=b2, cin_alloc=c1, cout_alloc=c2)
        a2 = pb.act(y2, 0.2, bn_in=b2)
        pooled = pb.avgpool(a2, 1)
        low = pb.conv("classifier", pooled, -1, 2 * c, 4, 1, 1, 1, 0, bias=True, cin_alloc=c2, linear=True)
        pb.head(low, -1, align_corners=False)
        self._finalize()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        """torch's default initialisers of nn.Conv2d / nn.Linear (kaiming_uniform(a=sqrt(5)) weights, uniform bias of the
        same fan-in) and nn.BatchNorm2d (1 / 0, running 0 / 1)."""
        import math
        fan = {}
        for name, prm in self.named_parameters():
            mod, attr = name.rsplit(".", 1)
            if mod.startswith("bn"):
                prm.fill_(1.0) if attr == "weight" else prm.zero_()
            elif attr == "weight":
                fan[mod] = prm[0].numel()
                bound = math.sqrt(6.0 / ((1 + 5.0) * fan[mod]))
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)
            else:
                bound = 1.0 / math.sqrt(fan[mod])
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * bound)
        for name, buf in self.named_buffers():
            if name.endswith("running_var"):
                buf.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                buf.zero_()

    def forward(self, task_pred):
        logits, _, _ = super().forward(task_pred, out_size=(1, 1))
        return logits.reshape(logits.shape[0], 4)


class FlawDetectorCore(SegNetCore):
    """Flaw detector of GCT (pixelssl/ssl_algorithm/ssl_gct.py:539-585): 7 x (conv 4x4 + bias, IBNorm, LeakyReLU 0.2)
    with strides 2,2,1,2,1,2,1, a 1-channel classifier conv (stride 2) and bilinear up-sampling (align_corners=True)
    to the input size.  Input: cat(image, softmax) NCHW fp32; its gradient is returned to autograd.  Parameter names
    follow the reference: conv1, ibn1.bnorm, conv2, ..., conv4_1, ibn4_1.bnorm, classifier."""
    ndf = 64
    LAYERS = (("conv1", 64, 2, "ibn1"), ("conv2", 128, 2, "ibn2"), ("conv2_1", 128, 1, "ibn2_1"), ("conv3", 256, 2, "ibn3"),
              ("conv3_1", 256, 1, "ibn3_1"), ("conv4", 512, 2, "ibn4"), ("conv4_1", 512, 1, "ibn4_1"))

    def __init__(self, in_channels, device="cuda", engine_dtype=torch.float32):
        super().__init__(device, engine_dtype, 1)
        self.want_prob = False
        self.has_latent = False
        pb = self._pb
        t = pb.input(in_channels)
        cin = in_channels
        for name, cout, stride, ibn in self.LAYERS:
            t = pb.conv(name, t, -1, cin, cout, 4, stride, 1, 1, bias=True, need_dgrad=True)
            t = pb.ibn(ibn, t, cout, 0.2)
            cin = cout
        low = pb.conv("classifier", t, -1, cin, 1, 4, 2, 1, 1, bias=True)
        pb.head(low, -1)
        self._finalize()
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self, generator=None):
        import math
        for name, prm in self.named_parameters():
            if ".bnorm." in name:
                prm.fill_(1.0) if name.endswith("weight") else prm.zero_()
            elif name.endswith("weight"):
                fan_in = prm.shape[1] * prm.shape[2] * prm.shape[3]
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) * math.sqrt(1.0 / fan_in))
                self._last_fan_in = fan_in
            else:
                prm.copy_((torch.rand(prm.shape, generator=generator) * 2 - 1) / math.sqrt(self._last_fan_in))
        for name, buf in self.named_buffers():
            if name.endswith("running_var"):
                buf.fill_(1.0)
            elif name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                buf.zero_()
