"""Guided Collaborative Training (reference: pixelssl/ssl_algorithm/ssl_gct.py) on the device.

Two independently initialised task models see the same batch; a flaw detector (conv 4x4 + IBNorm + LeakyReLU stack,
engine.FlawDetectorCore) predicts where each is wrong.  Per iteration (ssl_gct.py:186-269): step 0 no-grad task
forwards + detector forwards whose graphs are kept, flaw-map post-processing and the dynamic-consistency pseudo
ground truth; step 1 trains both task models (CE + flaw-correction + dynamic-consistency, the detector frozen: only
dL/dsoftmax is relayed); step 2 trains the detector against the FDGT maps through the step-0 graphs.  The flaw-map
pipeline -- FlawDetectorCriterion (:610-621), FlawmapHandler (:624-657), DCGTGenerator (:660-689), FDGTGenerator
(:692-728) -- keeps the reference's class names and call signatures on libpixelhip kernels (csrc/flawmap.hip);
tensors are fp32 NCHW on the GPU, exactly what the task model returns.
"""
import math
import os
import time

import numpy as np
import scipy.ndimage
import torch
import torch.nn as nn

from .. import _lib, streams
from .._lib import check, lib, ptr, stream_ptr

MODE_GCT, MODE_DC, MODE_FC = 'gct', 'dc', 'fc'


def _gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.PixelHipError("GCT flaw-map modules run on the GPU only (got %s); there is no CPU path" % t.device)


def _odd_ksize(im_size, div):
    k = int(im_size / div)
    return k + 1 if k % 2 == 0 else k


class GaussianBlurLayer(nn.Module):
    """nn/module/gaussian_blur.py: blur of single-channel maps.  The reference builds a dense k x k depthwise
    kernel with scipy's gaussian filter of a delta; that kernel is rank 1, so only the k 1-D taps are kept."""

    def __init__(self, channels, kernel_size):
        super().__init__()
        assert kernel_size % 2 != 0
        if channels != 1:
            raise NotImplementedError("GaussianBlurLayer: the GCT pipeline only blurs single-channel maps")
        self.channels, self.kernel_size = channels, kernel_size
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
        d = np.zeros(kernel_size)
        d[kernel_size // 2] = 1
        self.register_buffer("taps", torch.from_numpy(scipy.ndimage.gaussian_filter1d(d, sigma)).float())

    def forward(self, x):
        _gpu(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        assert C == 1
        # (the buffer stays where the module was built -- the CPU in the reference's pipeline; a per-call `.to()` is a
        # synchronous pageable copy = a stream sync, six per GCT iteration, which made the step host-bound: 35.4 ms with
        # the GPU idle 40 % of it)
        taps = getattr(self, '_taps_dev', None)
        if taps is None or taps.device != x.device:
            taps = self._taps_dev = self.taps.to(x.device)
        tmp, out = torch.empty_like(x), torch.empty_like(x)
        check(lib().pxl_gauss_sep_reflect(B, H, W, ptr(x), ptr(taps), self.kernel_size, ptr(tmp), ptr(out), stream_ptr()))
        return out


def _minmax_norm(x, clip):
    B = x.shape[0]
    mm = torch.empty(B, 2, device=x.device, dtype=torch.float32)
    out = torch.empty_like(x)
    check(lib().pxl_minmax_norm_persample(B, x.numel() // B, ptr(x), clip, ptr(mm), ptr(out), stream_ptr()))
    return out


class FlawDetectorCriterion(nn.Module):
    """MSE between the predicted flaw map and its ground truth, mean over (C,H,W) per sample (ssl_gct.py:617-621)."""

    def forward(self, pred, gt, is_ssl=False, reduction=True):
        if not reduction:
            return (pred - gt) ** 2
        return _MSEPerSample.apply(pred, gt)


class _MSEPerSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, g):
        _gpu(a, g)
        a, g = a.contiguous(), g.contiguous()
        B = a.shape[0]
        loss = torch.empty(B, device=a.device, dtype=torch.float32)
        check(lib().pxl_mse_persample_fwd(B, a.numel() // B, ptr(a), ptr(g), ptr(loss), stream_ptr()))
        ctx.save_for_backward(a, g)
        return loss

    @staticmethod
    def backward(ctx, gout):
        a, g = ctx.saved_tensors
        B = a.shape[0]
        da = torch.empty_like(a)
        check(lib().pxl_mse_persample_bwd(B, a.numel() // B, ptr(a), ptr(g), ptr(gout.contiguous().float()), ptr(da),
                                          stream_ptr()))
        return da, None


class FlawmapHandler(nn.Module):
    """Post-processing of the predicted flaw map (ssl_gct.py:624-657): clamp to >= 0 IN PLACE on the argument's
    storage (the reference's `flawmap.data.mul_`, which the step-2 FD loss later observes), blur k = im/16, zero the
    sample if its maximum is <= 0.1, min-max normalise."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.clip_threshold = 0.1
        self.blur = GaussianBlurLayer(1, _odd_ksize(args.im_size, 16))

    def forward(self, flawmap):
        flawmap = flawmap.data
        _gpu(flawmap)
        if not flawmap.is_contiguous():
            raise _lib.PixelHipError("FlawmapHandler needs a contiguous flaw map (it is clamped in place)")
        check(lib().pxl_clamp_min0_inplace(flawmap.numel(), ptr(flawmap), stream_ptr()))
        return _minmax_norm(self.blur(flawmap), self.clip_threshold)


class DCGTGenerator(nn.Module):
    """Ground truth of the dynamic consistency constraint (ssl_gct.py:668-689); the handled flaw maps are updated
    in place like in the reference."""

    def __init__(self, args):
        super().__init__()
        self.args = args

    def forward(self, l_pred, r_pred, l_handled_flawmap, r_handled_flawmap):
        _gpu(l_pred, r_pred, l_handled_flawmap, r_handled_flawmap)
        l_pred, r_pred = l_pred.contiguous(), r_pred.contiguous()
        B, C, H, W = l_pred.shape
        l_gt, r_gt = torch.empty_like(l_pred), torch.empty_like(r_pred)
        both_bad = torch.empty_like(l_handled_flawmap)
        check(lib().pxl_dcgt(B, C, H * W, ptr(l_pred), ptr(r_pred), ptr(l_handled_flawmap), ptr(r_handled_flawmap),
                             float(self.args.dc_threshold), ptr(l_gt), ptr(r_gt), ptr(both_bad), stream_ptr()))
        return l_gt, r_gt, both_bad, both_bad


class FDGTGenerator(nn.Module):
    """Ground truth of the flaw detector, pipeline 'C' of the paper (ssl_gct.py:692-728).  `gt` is either the one-hot
    tensor the reference's task hook builds, or (fast path) the float label map [B,1,H,W] itself: the one-hot is then
    formed on the fly inside the |gt - pred| kernel."""

    def __init__(self, args, ignore_index=255):
        super().__init__()
        self.args = args
        self.ignore_index = ignore_index
        self.blur = GaussianBlurLayer(1, _odd_ksize(args.im_size, 8))
        self.reblur = GaussianBlurLayer(1, _odd_ksize(args.im_size, 4))

    def forward(self, pred, gt):
        _gpu(pred, gt)
        pred = pred.detach().contiguous()
        B, C, H, W = pred.shape
        diff = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
        if gt.shape[1] == 1 and C != 1:
            check(lib().pxl_absdiff_chansum(B, C, H * W, ptr(pred), ptr(gt.contiguous().float()), self.ignore_index,
                                            float(self.args.mu), ptr(diff), stream_ptr()))
        else:
            if gt.shape != pred.shape:
                raise ValueError('FDGTGenerator: gt %s vs pred %s' % (tuple(gt.shape), tuple(pred.shape)))
            check(lib().pxl_absdiff_chansum_dense(B, C, H * W, ptr(pred), ptr(gt.detach().contiguous().float()),
                                                  float(self.args.mu), ptr(diff), stream_ptr()))
        diff = self.blur(diff)
        for _ in range(self.args.nu):
            dil = torch.empty_like(diff)
            check(lib().pxl_dilate3_reflect(B, H, W, ptr(diff), ptr(dil), stream_ptr()))
            diff = self.reblur(dil)
        return _minmax_norm(diff, -math.inf)


def onehot_ignore(gt, num_classes, ignore_index=255):
    """sslgct_prepare_task_gt_for_fdgt / ssladv_convert_task_gt_to_fcd_input (task/sseg/func.py:159-168,179-192)."""
    _gpu(gt)
    gt = gt.contiguous().float()
    B, _, H, W = gt.shape
    out = torch.empty(B, num_classes, H, W, device=gt.device, dtype=torch.float32)
    check(lib().pxl_onehot_ignore(B, num_classes, H * W, ptr(gt), ignore_index, ptr(out), stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# flaw detector + trainer
# ---------------------------------------------------------------------------------------------------------------------

def add_parser_arguments(parser):
    from . import ssl_base
    from ..utils import cmd   # noqa: F401
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--ssl-mode', type=str, default=MODE_GCT, choices=[MODE_GCT, MODE_DC, MODE_FC],
                        help='sslgct - select semi-supervised constraints for training (gct = dc + fc)')
    parser.add_argument('--fc-ssl-scale', type=float, default=-1.0, help='sslgct - flaw correction constraint coefficient')
    parser.add_argument('--dc-ssl-scale', type=float, default=-1.0, help='sslgct - dynamic consistency constraint coefficient')
    parser.add_argument('--dc-threshold', type=float, default=-1.0, help='sslgct - threshold of dynamic consistency constraint')
    parser.add_argument('--dc-rampup-epochs', type=int, default=-1, help='sslgct - ramp-up epochs of dynamic consistency constraint')
    parser.add_argument('--fd-lr', type=float, default=1e-4, help='sslgct - the initial learning rate of the flaw detector')
    parser.add_argument('--fd-scale', type=float, default=1.0, help='sslgct - coefficient of the flaw detector constraint')
    parser.add_argument('--mu', type=float, default=-1.0, help="sslgct - channel average coefficient of the FDGT generator")
    parser.add_argument('--nu', type=int, default=-1, help="sslgct - operations repeat coefficient of the FDGT generator")


def ssl_gct(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    """Export function (ssl_gct.py:53-86): one 'model' entry is instantiated twice, or 'lmodel' / 'rmodel'."""
    from ..utils import logger
    if not len(model_dict) == len(optimizer_dict) == len(lrer_dict) == len(criterion_dict):
        logger.log_err('The len(element_dict) of SSL_GCT should be the same\n')
    if len(model_dict) == 1:
        if list(model_dict.keys())[0] != 'model':
            logger.log_err("In SSL_GCT, the key of 1-value element_dict should be 'model',\nbut '{0}' is given\n"
                           .format(model_dict.keys()))
        pick = lambda d: [d['model'], d['model']]
    elif len(model_dict) == 2:
        if 'lmodel' not in model_dict or 'rmodel' not in model_dict:
            logger.log_err("In SSL_GCT, the key of 2-value element_dict should be '(lmodel, rmodel)', "
                           "but '{0}' is given\n".format(model_dict.keys()))
        pick = lambda d: [d['lmodel'], d['rmodel']]
    else:
        logger.log_err('The SSL_GCT algorithm supports element_dict with 1 or 2 elements, '
                       'but given {0} elements\n'.format(len(model_dict)))
    algorithm = SSLGCT(args)
    algorithm.build(pick(model_dict), pick(optimizer_dict), pick(lrer_dict), pick(criterion_dict), task_func)
    return algorithm


class FlawDetector(nn.Module):
    """ssl_gct.py:539-585 on the layer-program executor; forward(task_inp: tuple, task_pred) -> {'flawmap': logits}."""
    ndf = 64

    def __init__(self, in_channels, engine_dtype=torch.float32):
        super().__init__()
        from ..engine import FlawDetectorCore
        from .. import dist as pdist
        core = FlawDetectorCore(in_channels, device=pdist.local_device(), engine_dtype=engine_dtype)
        # the executor front-end stays outside the module registry: its leaves (conv1, ibn1.bnorm, ..., classifier) are
        # registered under the reference's attribute names, so state_dict / load_state_dict / parameters() of this module
        # and of every wrapper around it (`module.conv1.weight`, ssl_gct.py:367-369) are the reference's
        object.__setattr__(self, 'core', core)
        for name, child in core.named_children():
            self.add_module(name, child)

    def train(self, mode=True):
        self.core.train(mode)
        return super().train(mode)

    def forward(self, task_inp, task_pred):
        # concatenation along the channels happens in the executor's input op (no torch.cat copy, ssl_gct.py:578)
        flawmap, _, _ = self.core(tuple(task_inp) + (task_pred,))
        assert flawmap.shape[2:] == task_pred.shape[2:]
        return {'flawmap': flawmap}, {}


class _MaskedSqMean(torch.autograd.Function):
    """mean(mask * x^2): the flaw-correction constraint (MSE of the flaw map against zeros, masked by both_bad)."""

    @staticmethod
    def forward(ctx, x, mask):
        _gpu(x, mask)
        x, mask = x.contiguous(), mask.contiguous()
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib().pxl_masked_sq_mean_fwd(x.numel(), ptr(x), ptr(mask), ptr(out), stream_ptr()))
        ctx.save_for_backward(x, mask)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        x, mask = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(lib().pxl_masked_sq_mean_bwd(x.numel(), ptr(x), ptr(mask), ptr(gout.contiguous().float().view(1)), ptr(dx),
                                           stream_ptr()))
        return dx, None


class SSLGCT:
    pass


def _define_sslgct():
    from . import ssl_base
    from ..utils import REGRESSION, CLASSIFICATION, logger, tool
    from ..nn import func
    from ..nn.lrer import PolynomialLR
    from ..nn.optimizer import FusedAdam
    from ..nn.module import patch_replication_callback
    from ..functional import MSELoss

    class _SSLGCT(ssl_base._SSLBase):
        NAME = 'ssl_gct'
        SUPPORTED_TASK_TYPES = [REGRESSION, CLASSIFICATION]

        def __init__(self, args):
            super().__init__(args)
            self.l_model = self.r_model = self.fd_model = None
            self.l_optimizer = self.r_optimizer = self.fd_optimizer = None
            self.l_lrer = self.r_lrer = self.fd_lrer = None
            self.l_criterion = self.r_criterion = self.fd_criterion = self.dc_criterion = None
            self.flawmap_handler = self.dcgt_generator = self.fdgt_generator = None
            self.args.fd_lr *= getattr(self.args, 'gpus', 1)        # ssl_gct.py:107
            a = self.args
            if a.unlabeled_batch_size > 0:
                if a.ssl_mode in (MODE_GCT, MODE_FC) and a.fc_ssl_scale < 0:
                    logger.log_err('The argument - fc_ssl_scale - is not set (or invalid)\n')
                if a.ssl_mode in (MODE_GCT, MODE_DC):
                    if a.dc_rampup_epochs < 0 or a.dc_ssl_scale < 0 or a.dc_threshold < 0 or a.mu < 0 or a.nu < 0:
                        logger.log_err('The dynamic consistency constraint needs dc_rampup_epochs, dc_ssl_scale, '
                                       'dc_threshold, mu and nu to be set\n')
            if a.ssl_mode not in (MODE_GCT, MODE_DC, MODE_FC):
                logger.log_err('Unknown ssl_mode of SSL_GCT: {0} (gct / dc / fc)\n'.format(a.ssl_mode))

        def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
            a = self.args
            self.task_func = task_func
            self.l_model = patch_replication_callback(func.create_model(model_funcs[0], 'l_model', args=a))
            self.r_model = patch_replication_callback(func.create_model(model_funcs[1], 'r_model', args=a))
            fd_dtype = torch.float32 if getattr(a, 'engine_dtype', 'bf16') in ('fp32', 'f32') else torch.bfloat16
            self.fd_model = patch_replication_callback(func.create_model(
                FlawDetector, 'fd_model', in_channels=self.task_func.sslgct_fd_in_channels(), engine_dtype=fd_dtype))
            self.models = {'l_model': self.l_model, 'r_model': self.r_model, 'fd_model': self.fd_model}
            self.l_optimizer = optimizer_funcs[0](self.l_model.module.param_groups)
            self.r_optimizer = optimizer_funcs[1](self.r_model.module.param_groups)
            self.fd_optimizer = FusedAdam(filter(lambda p: p.requires_grad, self.fd_model.parameters()), lr=a.fd_lr,
                                          betas=(0.9, 0.99))
            self.optimizers = {'l_optimizer': self.l_optimizer, 'r_optimizer': self.r_optimizer,
                               'fd_optimizer': self.fd_optimizer}
            self.l_lrer = lrer_funcs[0](self.l_optimizer)
            self.r_lrer = lrer_funcs[1](self.r_optimizer)
            self.fd_lrer = PolynomialLR(self.fd_optimizer, a.epochs, a.iters_per_epoch, power=0.9, last_epoch=-1)
            self.lrers = {'l_lrer': self.l_lrer, 'r_lrer': self.r_lrer, 'fd_lrer': self.fd_lrer}
            self.l_criterion = criterion_funcs[0](a)
            self.r_criterion = criterion_funcs[1](a)
            self.fd_criterion = FlawDetectorCriterion()
            self.dc_criterion = MSELoss()
            self.criterions = {'l_criterion': self.l_criterion, 'r_criterion': self.r_criterion,
                               'fd_criterion': self.fd_criterion, 'dc_criterion': self.dc_criterion}
            self.flawmap_handler = FlawmapHandler(a)
            self.dcgt_generator = DCGTGenerator(a)
            self.fdgt_generator = FDGTGenerator(a, ignore_index=a.ignore_index)

        # ---- one task-model pass of step 1 (_task_model_iter, ssl_gct.py:401-480)
        def _side_stream(self):
            if not hasattr(self, '_r_stream'):
                on = os.environ.get('PXL_GCT_STREAMS', '1') != '0' and torch.cuda.is_available()
                self._r_stream = streams.role_stream(streams.SIDE) if on else None
            return self._r_stream

        def _reusable_cores(self):
            """The two task models' executors when PXL_GCT_REUSE_FORWARD=1 and both are deterministic DeepLab-v2 programs in
            training mode, else None (see train_step)."""
            if os.environ.get('PXL_GCT_REUSE_FORWARD', '0') != '1':
                return None
            from ..engine import DeepLabV2Core
            cores = []
            for m in (self.l_model, self.r_model):
                core = getattr(getattr(m, 'module', m), 'model', None)
                if not isinstance(core, DeepLabV2Core) or not core.training:
                    return None
                cores.append(core)
            return cores

        def _task_model_iter(self, mid, lbs, inp, gt, dc_gt, fc_mask, dc_rampup_scale, resulter=None):
            a = self.args
            model, criterion = (self.l_model, self.l_criterion) if mid == 'l' else (self.r_model, self.r_criterion)
            if resulter is None:
                resulter, _ = model.forward(inp)
            self._need_pred(resulter, 'SSL_GCT')
            pred = tool.dict_value(resulter, 'pred')
            activated_pred = tool.dict_value(resulter, 'activated_pred')
            flawmap = tool.dict_value(self.fd_model.forward(inp, activated_pred[0])[0], 'flawmap')
            task_loss = torch.mean(criterion.forward(func.split_tensor_tuple(pred, 0, lbs), func.split_tensor_tuple(gt, 0, lbs),
                                                     func.split_tensor_tuple(inp, 0, lbs)))
            # flaw correction: MSE of the flaw map against zeros (ssl_gct.py:429-442), masked by both_bad in 'gct' mode,
            # unmasked in 'fc' mode, off in 'dc' mode
            if a.ssl_mode in (MODE_GCT, MODE_FC):
                mask = fc_mask if a.ssl_mode == MODE_GCT else torch.ones_like(flawmap)
                fc_ssl_loss = a.fc_ssl_scale * _MaskedSqMean.apply(flawmap, mask)
            else:
                fc_ssl_loss = torch.zeros((), device=flawmap.device)
            # dynamic consistency (ssl_gct.py:445-458): off in 'fc' mode
            if a.ssl_mode in (MODE_GCT, MODE_DC):
                if dc_gt is None:
                    logger.log_err('The dynamic consistency constraint is enabled, but no pseudo ground truth is given.\n')
                dc_ssl_loss = dc_rampup_scale * a.dc_ssl_scale * torch.mean(self.dc_criterion.forward(activated_pred[0], dc_gt))
            else:
                dc_ssl_loss = torch.zeros((), device=flawmap.device)
            # (the reference also builds an FDGT map of the full batch here, used only for visualisation: skipped)
            return task_loss + fc_ssl_loss + dc_ssl_loss, dict(task=task_loss.detach(), fc=fc_ssl_loss.detach(),
                                                               dc=dc_ssl_loss.detach())

        def train_step(self, inp, gt, cur_step, total_rampup_steps):
            """One iteration of ssl_gct.py:186-269 on device-resident tuples (both task models see the same batch)."""
            a = self.args
            lbs = a.labeled_batch_size
            fd_core = self.fd_model.module.core
            dc_rampup_scale = func.sigmoid_rampup(cur_step, total_rampup_steps)
            # The two task models never depend on each other inside an iteration (the dynamic-consistency targets come
            # from step 0), so every l / r pair of network passes runs on two HIP streams; the flaw detector -- shared,
            # with BN running statistics that must be updated in the reference's order -- stays on the main stream.
            side = self._side_stream()
            main = torch.cuda.current_stream() if side is not None else None

            def pair(fn_l, fn_r):
                if side is None:
                    return fn_l(), fn_r()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    r = fn_r()
                l = fn_l()
                main.wait_stream(side)
                return l, r

            def keep(res):
                if side is not None:
                    for v in res.values():
                        for t in (v if isinstance(v, (tuple, list)) else (v,)):
                            if torch.is_tensor(t):
                                t.record_stream(main)        # allocated on the side stream, consumed on the main one
                return res

            # ---- step 0: no-grad task forwards, flaw-detector forwards whose graphs are kept for step 2
            # PXL_GCT_REUSE_FORWARD=1 (opt-in, off by default): the reference runs every task model TWICE per iteration on the
            # same batch with the same weights -- no-grad here, with a graph in step 1 (ssl_gct.py:196-200, 403); nothing between
            # the two passes changes a task model, so they produce the same tensors, and the only trace of the repetition is
            # the second running-statistics update.  With the switch on, the step-1 pass runs HERE (with its graph), the
            # running statistics take both updates at once (momentum 1 - (1-m)^2: pxl_net_set_bn_repeat) and step 1 reuses it.
            # Only for networks whose forward is a pure function of (weights, batch): the DeepLab-v2 executor (no dropout).
            reuse_cores = self._reusable_cores()
            l_fwd = r_fwd = None
            if reuse_cores is not None:
                for c in reuse_cores:
                    c.set_bn_repeat(2)
                try:
                    l_fwd, r_fwd = pair(lambda: self.l_model.forward(inp)[0], lambda: keep(self.r_model.forward(inp)[0]))
                finally:
                    for c in reuse_cores:
                        c.set_bn_repeat(1)
                l_prob = tuple(t.detach() for t in tool.dict_value(l_fwd, 'activated_pred'))
                r_prob = tuple(t.detach() for t in tool.dict_value(r_fwd, 'activated_pred'))
            else:
                with torch.no_grad():
                    l_res, r_res = pair(lambda: self.l_model.forward(inp)[0], lambda: keep(self.r_model.forward(inp)[0]))
                    l_prob = tool.dict_value(l_res, 'activated_pred')
                    r_prob = tool.dict_value(r_res, 'activated_pred')
            fd_core.set_wgrad(True)
            l_flawmap = tool.dict_value(self.fd_model.forward(inp, l_prob[0])[0], 'flawmap')
            r_flawmap = tool.dict_value(self.fd_model.forward(inp, r_prob[0])[0], 'flawmap')
            l_dc_gt = r_dc_gt = l_fc_mask = r_fc_mask = None
            if a.ssl_mode in (MODE_GCT, MODE_DC):          # ssl_gct.py:219-224 ('fc' mode: no handled maps, no in-place clamp)
                with torch.no_grad():
                    l_handled = self.flawmap_handler.forward(l_flawmap)         # clamps l_flawmap / r_flawmap IN PLACE
                    r_handled = self.flawmap_handler.forward(r_flawmap)
                    l_dc_gt, r_dc_gt, l_fc_mask, r_fc_mask = self.dcgt_generator.forward(l_prob[0].detach(), r_prob[0].detach(),
                                                                                        l_handled, r_handled)
            # ---- step 1: task models; the flaw detector is frozen (requires_grad False in the reference).  Reference
            # order: [l: forward, losses, backward, step] then [r: ...]; the two are independent, so both forwards run
            # side by side, then the (ordered) detector passes and losses, then ONE backward over both graphs -- the
            # engine runs each network's backward on the stream of its forward -- and both optimizer steps.
            fd_core.set_wgrad(False)
            out = {}
            if l_fwd is None:
                l_fwd, r_fwd = pair(lambda: self.l_model.forward(inp)[0], lambda: keep(self.r_model.forward(inp)[0]))
            losses = []
            for mid, resulter, dc_gt, fc_mask in (('l', l_fwd, l_dc_gt, l_fc_mask), ('r', r_fwd, r_dc_gt, r_fc_mask)):
                loss, parts = self._task_model_iter(mid, lbs, inp, gt, dc_gt, fc_mask, dc_rampup_scale, resulter)
                losses.append(loss)
                for k, v in parts.items():
                    out['{0}_{1}_loss'.format(mid, k)] = v
            self.l_optimizer.zero_grad()
            self.r_optimizer.zero_grad()
            torch.autograd.backward(losses)
            if side is not None:
                main.wait_stream(side)
            self.l_optimizer.step()
            self.r_optimizer.step()
            # ---- step 2: flaw detector, ground truth from the STEP-0 predictions of the labeled samples
            fd_core.set_wgrad(True)
            with torch.no_grad():
                l_fm_gt = self.fdgt_generator.forward(l_prob[0][:lbs, ...].detach(), gt[0][:lbs, ...])
                r_fm_gt = self.fdgt_generator.forward(r_prob[0][:lbs, ...].detach(), gt[0][:lbs, ...])
            l_fd_loss = a.fd_scale * torch.mean(self.fd_criterion.forward(l_flawmap[:lbs, ...], l_fm_gt))
            r_fd_loss = a.fd_scale * torch.mean(self.fd_criterion.forward(r_flawmap[:lbs, ...], r_fm_gt))
            fd_loss = (l_fd_loss + r_fd_loss) / 2
            self.fd_optimizer.zero_grad()
            fd_loss.backward()
            self.fd_optimizer.step()
            self.fd_lrer.step()
            if not a.is_epoch_lrer:
                self.l_lrer.step()
                self.r_lrer.step()
            out['l_fd_loss'], out['r_fd_loss'] = l_fd_loss.detach(), r_fd_loss.detach()
            return out

        def _train(self, data_loader, epoch):
            self.meters.reset()
            for m in (self.l_model, self.r_model, self.fd_model):
                m.train()
            for idx, (inp, gt) in enumerate(data_loader):
                timer = time.time()
                inp, gt = self._to_device(inp), self._to_device(gt)
                cur = len(data_loader) * epoch + idx
                losses = self.train_step(inp, gt, cur, len(data_loader) * self.args.dc_rampup_epochs)
                for k, v in losses.items():
                    self.meters.update(k, v)
                self.meters.update('batch_time', time.time() - timer)
                if idx % self.args.log_freq == 0:
                    m = self.meters
                    logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                    '  l\t=>\ttask {4:.6f}\tdc {5:.6f}\tfc {6:.6f}\n  r\t=>\ttask {7:.6f}\tdc {8:.6f}\tfc {9:.6f}\n'
                                    '  fd\t=>\tl {10:.6f}\tr {11:.6f}\n'
                                    .format(epoch + 1, idx, len(data_loader), m['batch_time'].avg,
                                            float(m['l_task_loss'].avg), float(m['l_dc_loss'].avg), float(m['l_fc_loss'].avg),
                                            float(m['r_task_loss'].avg), float(m['r_dc_loss'].avg), float(m['r_fc_loss'].avg),
                                            float(m['l_fd_loss'].avg), float(m['r_fd_loss'].avg)))
            if self.args.is_epoch_lrer:
                self.l_lrer.step()
                self.r_lrer.step()

        @torch.no_grad()
        def _validate(self, data_loader, epoch):
            """ssl_gct.py:300-361: both task models in eval mode, task loss + metrics of each (the flaw-detector /
            consistency terms the reference also evaluates on the validation set are logging only and are not computed)."""
            self.meters.reset()
            for m in (self.l_model, self.r_model, self.fd_model):
                m.eval()
            for idx, (inp, gt) in enumerate(data_loader):
                timer = time.time()
                inp, gt = self._to_device(inp), self._to_device(gt)
                for mid, model, criterion in (('l', self.l_model, self.l_criterion), ('r', self.r_model, self.r_criterion)):
                    resulter, _ = model.forward(inp)
                    self._need_pred(resulter, 'SSL_GCT')
                    self.meters.update(mid + '_task_loss',
                                       torch.mean(criterion.forward(tool.dict_value(resulter, 'pred'), gt, inp)).detach())
                    self.task_func.metrics(tool.dict_value(resulter, 'activated_pred'), gt, inp, self.meters, id_str=mid)
                self.meters.update('batch_time', time.time() - timer)
                if idx % self.args.log_freq == 0:
                    logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                    '  l-{4}\t=>\tl-task-loss: {5:.6f}\n  r-{4}\t=>\tr-task-loss: {6:.6f}\n'
                                    .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                            float(self.meters['l_task_loss'].avg), float(self.meters['r_task_loss'].avg)))
            self._log_validation_metrics(['l', 'r'])

        def _save_checkpoint(self, epoch):
            state = {'algorithm': self.NAME, 'epoch': epoch}
            for k, v in list(self.models.items()) + list(self.optimizers.items()) + list(self.lrers.items()):
                state[k] = v.state_dict()
            torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

        def _load_checkpoint(self):
            checkpoint = torch.load(self.args.resume, map_location='cpu')
            found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
            if found != self.NAME:
                logger.log_err('Unmatched SSL algorithm format in checkpoint => required: {0} - given: {1}\n'
                               .format(self.NAME, found))
            for k, v in list(self.models.items()) + list(self.optimizers.items()) + list(self.lrers.items()):
                v.load_state_dict(checkpoint[k])                      # ssl_gct.py:389-397: models, optimizers, lrers
            return checkpoint['epoch']

    return _SSLGCT


SSLGCT = _define_sslgct()
SSLGCT.__name__ = SSLGCT.__qualname__ = 'SSLGCT'
NAME = SSLGCT.NAME
