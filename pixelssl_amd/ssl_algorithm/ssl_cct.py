"""Cross-Consistency Training (pixelssl/ssl_algorithm/ssl_cct.py) on the MI355X engine.

One main task model (PSPNet in the shipped script) and K auxiliary decoders that see perturbed versions of the
encoder latent of the UNLABELED pass; consistency = mean over decoders of MSE(softmax(aux prediction), softmax(main
prediction).detach()), scaled by cons_scale * sigmoid ramp-up; the labeled pass is a second, separate forward of the
main model (its own BN batch), exactly as ssl_cct.py:248-266.

Device mapping:
  * every decoder body (`upsample`: 1x1 conv + 3 x [1x1 conv, ReLU, PixelShuffle]) and the resize + soft-max that
    WrappedCCTModel applies to its output are ONE executor program (engine.AuxDecoderCore) -> one C call per pass;
  * the perturbations are single fused kernels over the latent (csrc/cct.hip); their masks are built on the device
    (argmax / nearest resize / channel-mean threshold) except G-Cutout's, whose contour search runs on the host like
    the reference's cv2 call (csrc/contour.cpp);
  * the gradient the decoders send into the latent is seeded into the main model's backward (engine._SegNetFn).
"""
import collections.abc
import math
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from ..utils import CLASSIFICATION, logger, cmd, tool
from ..nn import func
from ..nn.module import patch_replication_callback
from ..functional import MSELoss
from .. import functional as PF
from ..engine import AuxDecoderCore
from .. import _lib, streams
from .. import dist as pdist
from .._lib import check, lib, ptr, stream_ptr
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--cons-scale', type=float, default=-1, help='sslcct - consistency constraint coefficient')
    parser.add_argument('--cons-rampup-epochs', type=int, default=-1, help='sslcct - ramp-up epochs of conistency constraint')
    parser.add_argument('--ad-lr-scale', type=float, default=-1, help='sslcct - learning rate scale for auxiliary decoders')
    parser.add_argument('--vat-dec-num', type=int, default=0, help="sslcct - number of the 'I-VAT' auxiliary decoders")
    parser.add_argument('--vat-dec-xi', type=float, default=1e-6, help="sslcct - the argument 'xi' for 'I-VAT' auxiliary decoders")
    parser.add_argument('--vat-dec-eps', type=float, default=2.0, help="sslcct - the argument 'eps' for 'I-VAT' auxiliary decoders")
    parser.add_argument('--drop-dec-num', type=int, default=0, help="sslcct - number of the 'DropOut' auxiliary decoders")
    parser.add_argument('--drop-dec-rate', type=float, default=0.5, help="sslcct - the argument 'rate' for 'DropOut' auxiliary decoders")
    parser.add_argument('--drop-dec-spatial', type=cmd.str2bool, default=True, help="sslcct - the argument 'spatial' for 'DropOut' auxiliary decoders")
    parser.add_argument('--cut-dec-num', type=int, default=0, help="sslcct - number of the 'G-Cutout' auxiliary decoders")
    parser.add_argument('--cut-dec-erase', type=float, default=0.4, help="sslcct - the argument 'erase' for 'G-Cutout' auxiliary decoders")
    parser.add_argument('--context-dec-num', type=int, default=0, help="sslcct - number of the 'Con-Msk' auxiliary decoders")
    parser.add_argument('--object-dec-num', type=int, default=0, help="sslcct - number of the 'Obj-Msk' auxiliary decoders")
    parser.add_argument('--fn-dec-num', type=int, default=0, help="sslcct - number of the 'F-Noise' auxiliary decoders")
    parser.add_argument('--fn-dec-uniform', type=float, default=0.3, help="sslcct - the argument 'uniform' for 'F-Noise' auxiliary decoders")
    parser.add_argument('--fd-dec-num', type=int, default=0, help="sslcct - number of the 'F-Drop' auxiliary decoders")


def ssl_cct(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_cct', model_dict, optimizer_dict, lrer_dict,
                                                         criterion_dict)
    algorithm = SSLCCT(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


# ----------------------------------------------------------------------------------------------------------------------
# device ops
# ----------------------------------------------------------------------------------------------------------------------

# step timeline (tools/cct_timeline.py): TIMELINE = [] switches marks on; each mark = (name, host perf_counter, an event recorded on
# the current stream).  None (the default): mark() returns at once.
TIMELINE = None


def mark(name):
    if TIMELINE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        TIMELINE.append((name, time.perf_counter(), ev))


def _f32(t):
    return None if t is None else t.contiguous().float()


def latent_perturb(x, mask=None, cscale=None, noise=None, add=None, add_scale=1.0):
    """out = x * mask[b,p] * cscale[b,c] * (1 + noise[c,p]) + add_scale * add   (csrc/cct.hip), no autograd."""
    if not x.is_cuda:
        raise _lib.PixelHipError("latent_perturb runs on the GPU only (got a %s tensor)" % x.device)
    x = _f32(x)
    B, C, h, w = x.shape
    out = torch.empty_like(x)
    mask, cscale, noise, add = _f32(mask), _f32(cscale), _f32(noise), _f32(add)
    check(lib().pxl_latent_perturb(B, C, h * w, ptr(x), ptr(mask), ptr(cscale), ptr(noise), ptr(add), float(add_scale),
                                   ptr(out), stream_ptr()))
    return out


class _PerturbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, cscale, noise, add, add_scale):
        ctx.save_for_backward(mask, cscale, noise)
        return latent_perturb(x, mask, cscale, noise, add, add_scale)

    @staticmethod
    def backward(ctx, dout):
        mask, cscale, noise = ctx.saved_tensors
        return latent_perturb(dout, mask, cscale, noise), None, None, None, None, None


def perturb(x, mask=None, cscale=None, noise=None, add=None, add_scale=1.0):
    return _PerturbFn.apply(x, mask, cscale, noise, add, add_scale)


def fg_mask_nearest(pred, size, invert=False):
    """(argmax_c pred > 0) nearest-resized to `size` (ssl_cct.py:664-669) -> [B, h, w] float mask"""
    pred = _f32(pred)
    B, C, H, W = pred.shape
    out = torch.empty(B, size[0], size[1], device=pred.device, dtype=torch.float32)
    check(lib().pxl_fg_mask_nearest(B, C, H, W, ptr(pred), size[0], size[1], int(invert), ptr(out), stream_ptr()))
    return out


def feature_drop_mask(x, u):
    """mask[b,p] = mean_c x[b,c,p] < max_p(mean_c x[b]) * u   (ssl_cct.py:718-724)"""
    x = _f32(x)
    B, C, h, w = x.shape
    att = torch.empty(B, h, w, device=x.device, dtype=torch.float32)
    check(lib().pxl_chan_mean(B, C, h * w, ptr(x), ptr(att), stream_ptr()))
    mask = torch.empty_like(att)
    check(lib().pxl_fdrop_mask(B, h * w, ptr(att), float(u), ptr(mask), stream_ptr()))
    return mask


def l2_normalize(d, scale=1.0):
    """scale * d / (||d_b||_2 + 1e-8) per sample (ssl_cct.py:577-581)"""
    d = _f32(d)
    out = torch.empty_like(d)
    norm2 = torch.empty(d.shape[0], device=d.device, dtype=torch.float32)
    check(lib().pxl_l2_normalize_persample(d.shape[0], d[0].numel(), ptr(d), float(scale), ptr(norm2), ptr(out), stream_ptr()))
    return out


def sub_scale(a, b, scale):
    a, b = _f32(a), _f32(b)
    out = torch.empty_like(a)
    check(lib().pxl_sub_scale(a.numel(), ptr(a), ptr(b), float(scale), ptr(out), stream_ptr()))
    return out


_CONTOUR_POOL = None


def _contour_pool():
    global _CONTOUR_POOL
    if _CONTOUR_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _CONTOUR_POOL = ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1), thread_name_prefix="pxl-contour")
    return _CONTOUR_POOL


def external_contour_boxes(mask_np, min_vertices=50, max_boxes=1024):
    """Host: bounding boxes (min_x, max_x, min_y, max_y) of the external contours with > min_vertices polygon vertices
    (what cv2.findContours + the `c.shape[0] > 50` filter of ssl_cct.py:627-632 keep), in cv2's list order (newest-found
    first = reverse raster order of the contours' first pixels, csrc/contour.cpp)."""
    import ctypes
    m = np.ascontiguousarray(mask_np, dtype=np.uint8)
    while True:
        boxes = (ctypes.c_int * (4 * max(max_boxes, 1)))()
        n = ctypes.c_int()
        check(lib().pxl_external_contour_boxes_host(m.ctypes.data, m.shape[0], m.shape[1], int(min_vertices), boxes,
                                                    max_boxes, ctypes.byref(n)))
        if n.value <= max_boxes:
            return [tuple(boxes[4 * i:4 * i + 4]) for i in range(n.value)]
        max_boxes = n.value                  # more contours than the buffer: ask again with room for all of them


# ----------------------------------------------------------------------------------------------------------------------
# auxiliary decoders (class names as ssl_cct.py:535-745)
# ----------------------------------------------------------------------------------------------------------------------

class _AuxDecoder(nn.Module):
    """Common part: the `upsample` body + one-shot injection of the random draw (tests replay the reference's)."""

    def __init__(self, upscale, in_channels, num_classes, engine_dtype=torch.bfloat16):
        super().__init__()
        self.upscale = upscale
        self.upsample = AuxDecoderCore(upscale, in_channels, num_classes, device=pdist.local_device(),
                                       engine_dtype=engine_dtype)
        self._draw = None
        self.last_draw = None

    def inject_draw(self, draw):
        self._draw = draw

    def _take_draw(self):
        d, self._draw = self._draw, None
        return d

    def _decode(self, x, out_size):
        logits, prob, _ = self.upsample(x, out_size=out_size)
        return logits, prob

    def perturb(self, x, pred_of_main_decoder):
        raise NotImplementedError

    def forward(self, x, pred_of_main_decoder=None, out_size=None):
        """-> (prediction resized to out_size with bilinear / align_corners=False, its soft-max).  out_size=None keeps
        the decoder's own resolution (upscale x the latent), i.e. the reference decoder's return value."""
        if out_size is None:
            out_size = (x.shape[2] * self.upscale, x.shape[3] * self.upscale)
        return self._decode(self.perturb(x, pred_of_main_decoder), out_size)

    def consistency(self, x, pred_of_main_decoder, target, out_size):
        """The decoder's whole branch of ssl_cct.py:476-484 as one differentiable scalar: perturbation, decoder body, resize to
        `out_size`, soft-max and MSE against `target` (functional.decoder_consistency: no full-resolution plane is written).
        -> (term, DeferredHead) or None when the fused seam cannot run for these shapes (the caller then uses forward())."""
        if not PF.decoder_consistency_supported(self.upsample, x, target, out_size):
            return None
        return PF.decoder_consistency(self.upsample, self.perturb(x, pred_of_main_decoder), target, out_size)


class VATDecoder(_AuxDecoder):
    def __init__(self, upscale, in_channels, num_classes, xi=1e-1, eps=10.0, iterations=1, **kw):
        super().__init__(upscale, in_channels, num_classes, **kw)
        self.xi, self.eps, self.it = xi, eps, iterations

    def get_r_adv(self, x):
        """ssl_cct.py:548-575: one power iteration of virtual adversarial training on the decoder.  The inner passes
        run at the decoder's own resolution and never touch its parameter gradients (the reference zeroes them)."""
        core = self.upsample
        own = (x.shape[2] * self.upscale, x.shape[3] * self.upscale)
        x_det = x.detach()
        with torch.no_grad():
            _, pred = self._decode(x_det, own)
        d = self._take_draw()
        if d is None:
            d = torch.rand(x.shape, device=x.device).sub_(0.5)
        self.last_draw = d
        d = l2_normalize(d.to(x.device))
        was = core._wgrad_on
        core.set_wgrad(False)
        try:
            for _ in range(self.it):
                xin = latent_perturb(x_det, add=d, add_scale=self.xi).requires_grad_(True)
                with torch.enable_grad():
                    logits_hat, prob_hat, _ = core(xin, out_size=own)
                # d/dlogits KL(pred || softmax(logits_hat)) with reduction='batchmean'
                logits_hat.backward(sub_scale(prob_hat.detach(), pred, 1.0 / x.shape[0]))
                d = l2_normalize(xin.grad.mul_(self.xi))
        finally:
            core.set_wgrad(was)
        return d.mul_(self.eps)

    def perturb(self, x, pred_of_main_decoder):
        return perturb(x, add=self.get_r_adv(x))


class DropOutDecoder(_AuxDecoder):
    """ssl_cct.py:585-593: nn.Dropout2d (spatial_dropout, the shipped script: one Bernoulli(1 - p) per (sample, channel)) or
    nn.Dropout (one per element) in front of the decoder body; kept values are scaled by 1 / (1 - p).  The draw is a scale
    tensor -- [B, C] resp. [B, C, h, w] -- so that tests can inject the reference's."""

    def __init__(self, upscale, in_channels, num_classes, drop_rate=0.3, spatial_dropout=True, **kw):
        super().__init__(upscale, in_channels, num_classes, **kw)
        self.drop_rate = drop_rate
        self.spatial_dropout = spatial_dropout

    def perturb(self, x, pred_of_main_decoder):
        scale = self._take_draw()
        B, C, h, w = x.shape
        if scale is None:
            shape = (B, C) if self.spatial_dropout else (B, C, h, w)
            keep = (torch.rand(shape, device=x.device) >= self.drop_rate).float()
            scale = keep / (1.0 - self.drop_rate)
        self.last_draw = scale
        scale = scale.to(x.device)
        if scale.dim() == 2:
            return perturb(x, cscale=scale)
        # element-wise: the kernel's per-(sample, position) factor with every (sample, channel) plane as its own sample
        return perturb(x.reshape(B * C, 1, h, w), mask=scale.reshape(B * C, h, w)).reshape(B, C, h, w)


class CutOutDecoder(_AuxDecoder):
    def __init__(self, upscale, in_channels, num_classes, drop_rate=0.3, spatial_dropout=True, erase=0.4, **kw):
        super().__init__(upscale, in_channels, num_classes, **kw)
        self.erase = erase
        self.min_vertices = 50          # `c.shape[0] > 50` contour filter of ssl_cct.py:632

    def prefetch(self, output):
        """Start the D2H copy of the predicted foreground mask (the input of the host contour search) right after the
        main model's forward: WrappedCCTModel issues it before the other decoders are enqueued and runs the G-Cutout
        decoders last, so that the copy is long finished when the host needs it (the reference stalls on `.cpu()`)."""
        B, _, H, W = output.shape
        fg = fg_mask_nearest(output, (H, W)).to(torch.uint8)
        host = torch.empty(fg.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(fg, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._prefetched = (host, ev, fg)          # fg kept alive until the copy is done

    def guided_cutout(self, output, resize, erase=0.4):
        """ssl_cct.py:615-656 (use_dropout=False): host contour search on the predicted foreground mask, one erased
        window per kept contour, nearest resize to the latent size.  -> [B, h, w] float mask on the device."""
        B, _, H, W = output.shape
        pre = getattr(self, '_prefetched', None)
        if pre is not None:
            self._prefetched = None
            mark('cutout: before the copy event')
            pre[1].synchronize()
            mark('cutout: copy event reached')
            fg = pre[0].numpy()
        else:
            fg = fg_mask_nearest(output, (H, W)).to(torch.uint8).cpu().numpy()    # D2H + sync, like the reference
        # an injected draw is either the list of uniform draws (two per kept contour) or a dict {'u': [...], 'boxes':
        # [[(min_w, max_w, min_h, max_h), ...] per sample]} that also replaces the contour search (the parity fixtures
        # carry the boxes the reference's cv2.findContours call returned: tests/test_multistep.py)
        draws = self._take_draw()
        boxes_in = None
        if isinstance(draws, dict):
            boxes_in, draws = draws.get('boxes'), draws.get('u')
        draws = iter(draws) if draws is not None else None
        used, used_boxes = [], []
        out = np.ones((B, H, W), dtype=np.float32)
        if boxes_in is None:
            # one host thread per sample (the C routine runs without the GIL): the search is ~2 ms per 513 x 513 mask and
            # sat on the step's critical path (the host enqueues nothing meanwhile)
            boxes_in = list(_contour_pool().map(lambda m: external_contour_boxes(m, self.min_vertices), [fg[b] for b in range(B)])) \
                if B > 1 else [external_contour_boxes(fg[0], self.min_vertices)]
        for b in range(B):
            boxes = boxes_in[b]
            used_boxes.append([tuple(int(v) for v in bx) for bx in boxes])
            for (min_w, max_w, min_h, max_h) in boxes:
                bb_w, bb_h = max_w - min_w, max_h - min_h
                nw, nh = int(bb_w * (1 - erase)), int(bb_h * (1 - erase))
                # random.randint(0, n) of ssl_cct.py:637-638, kept as the uniform u with floor(u * (n + 1)) = k
                uw = (random.randint(0, nw) + 0.5) / (nw + 1) if draws is None else next(draws)
                uh = (random.randint(0, nh) + 0.5) / (nh + 1) if draws is None else next(draws)
                used += [uw, uh]
                sw, sh = int(uw * (nw + 1)), int(uh * (nh + 1))
                out[b, min_h + sh:min_h + sh + int(bb_h * erase), min_w + sw:min_w + sw + int(bb_w * erase)] = 0
        mark('cutout: contours done')
        self.last_boxes = used_boxes
        self.last_draw = used
        ys = np.minimum(np.floor(np.arange(resize[0], dtype=np.float32) * np.float32(H / resize[0])).astype(np.int64), H - 1)
        xs = np.minimum(np.floor(np.arange(resize[1], dtype=np.float32) * np.float32(W / resize[1])).astype(np.int64), W - 1)
        small = out[:, ys][:, :, xs]                                             # F.interpolate(mode='nearest')
        return torch.from_numpy(np.ascontiguousarray(small)).to(output.device, non_blocking=True)

    def perturb(self, x, pred_of_main_decoder):
        return perturb(x, mask=self.guided_cutout(pred_of_main_decoder, (x.shape[2], x.shape[3]), self.erase))


class ContextMaskingDecoder(_AuxDecoder):
    def perturb(self, x, pred_of_main_decoder):
        return perturb(x, mask=fg_mask_nearest(pred_of_main_decoder, (x.shape[2], x.shape[3])))


class ObjectMaskingDecoder(_AuxDecoder):
    def perturb(self, x, pred_of_main_decoder):
        return perturb(x, mask=fg_mask_nearest(pred_of_main_decoder, (x.shape[2], x.shape[3]), invert=True))


class FeatureDropDecoder(_AuxDecoder):
    def perturb(self, x, pred_of_main_decoder):
        u = self._take_draw()
        if u is None:
            u = float(np.random.uniform(0.7, 0.9))
        self.last_draw = u
        return perturb(x, mask=feature_drop_mask(x.detach(), u))


class FeatureNoiseDecoder(_AuxDecoder):
    def __init__(self, upscale, in_channels, num_classes, uniform_range=0.3, **kw):
        super().__init__(upscale, in_channels, num_classes, **kw)
        self.uniform_range = uniform_range

    def perturb(self, x, pred_of_main_decoder):
        noise = self._take_draw()
        if noise is None:
            noise = (torch.rand(x.shape[1:], device=x.device) * 2 - 1) * self.uniform_range
        self.last_draw = noise
        return perturb(x, noise=noise.to(x.device))


# ----------------------------------------------------------------------------------------------------------------------
# wrapped model
# ----------------------------------------------------------------------------------------------------------------------

class _LazyPreds(collections.abc.Sequence):
    """resulter['ul_ad_preds'] (ssl_cct.py:478): the auxiliary predictions, resized to the main prediction's size.  The reference's
    training loop fetches the list and never reads it (ssl_cct.py:267); entries of decoders that ran the fused seam are
    materialised from the decoder's low-resolution logits on first access."""

    def __init__(self, items):
        self._items = list(items)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self._items)))]
        it = self._items[i]
        if not torch.is_tensor(it):
            it = self._items[i] = it.materialize(want_prob=False)[0]
        return it


class WrappedCCTModel(nn.Module):
    """ssl_cct.py:425-491: main model + auxiliary decoders + both criterions in one module; param_groups = the main
    model's groups + one group for all decoders at lr * ad_lr_scale."""

    def __init__(self, args, main_model, auxiliary_decoders, task_criterion, cons_criterion, ad_activation_func):
        super().__init__()
        self.args = args
        self.main_model = main_model
        self.auxiliary_decoders = auxiliary_decoders
        self.task_criterion = task_criterion
        self.cons_criterion = cons_criterion
        self.ad_activation_func = ad_activation_func
        core = getattr(main_model, 'model', None)
        if core is not None and hasattr(core, 'differentiable_latent'):
            core.differentiable_latent = True      # the decoders' gradient re-enters the main model through the latent
        self.param_groups = list(self.main_model.param_groups) + \
            [{'params': list(self.auxiliary_decoders.parameters()), 'lr': self.args.lr * self.args.ad_lr_scale}]

    def _lanes(self, device):
        if not hasattr(self, '_lane_streams'):
            # (several ranks: every decoder's gradient all-reduce would be issued on its lane's stream -- collectives of one
            # communicator from streams that are not ordered against each other; one stream then, unless PXL_CCT_STREAMS says otherwise)
            default = '0' if pdist.is_distributed() else '2'
            n = int(os.environ.get('PXL_CCT_STREAMS', default)) if device.type == 'cuda' else 0
            # decoder lanes: dealt over the queues that are not the main stream's (AUX first: the labeled pass holds SIDE)
            self._lane_streams = [streams.role_stream(streams.AUX, device=device, index=i) for i in range(max(n, 0))]
        return self._lane_streams

    def forward(self, inp, gt, is_unlabeled):
        resulter, debugger = {}, {}
        m_resulter, _ = self.main_model.forward(inp)
        if 'pred' not in m_resulter.keys() or 'activated_pred' not in m_resulter.keys():
            ssl_base._SSLBase._need_pred(m_resulter, 'SSL_CCT')
        resulter['pred'] = tool.dict_value(m_resulter, 'pred')
        resulter['activated_pred'] = tool.dict_value(m_resulter, 'activated_pred')
        if not len(resulter['pred']) == len(resulter['activated_pred']) == 1:
            logger.log_err('This implementation of SSL_CCT only support the task model with only one prediction (output). \n'
                           'However, there are {0} predictions.\n'.format(len(resulter['pred'])))
        resulter['task_loss'] = None if is_unlabeled else torch.mean(self.task_criterion.forward(resulter['pred'], gt, inp))

        if is_unlabeled and self.args.unlabeled_batch_size > 0:
            if 'sslcct_ad_inp' not in m_resulter.keys():
                logger.log_err("In SSL_CCT, the 'resulter' dict returned by the task model should contain the key:\n"
                               "    'sslcct_ad_inp'\t=>\tinputs of the auxiliary decoders (a 4-dim tensor)\n")
            ul_ad_inp = tool.dict_value(m_resulter, 'sslcct_ad_inp')
            ul_main_pred = resulter['pred'][0].detach()
            ul_ad_gt = resulter['activated_pred'][0].detach()
            size = (ul_ad_gt.shape[2], ul_ad_gt.shape[3])
            # every decoder returns its prediction already resized (bilinear, align_corners=False) and activated: the
            # executor's HEAD kernel fuses F.interpolate + softmax (ssl_cct.py:483-484)
            # G-Cutout needs a host round trip: its mask copy is started first and its decoders run last (the sum of
            # the consistency terms does not depend on the order)
            order = list(range(len(self.auxiliary_decoders)))
            if os.environ.get('PXL_CCT_PREFETCH', '1') != '0':
                cuts = [i for i in order if isinstance(self.auxiliary_decoders[i], CutOutDecoder)]
                for i in cuts:
                    self.auxiliary_decoders[i].prefetch(ul_main_pred)
                order = [i for i in order if i not in cuts] + cuts
            # (SSLCCT.train_step: the labeled backward is enqueued HERE -- behind the unlabeled forward, ahead of the decoders)
            mark('unlabeled: main forward enqueued')
            hook, self.after_main_forward = getattr(self, 'after_main_forward', None), None
            if hook is not None:
                hook()
            # The decoders are independent between the latent and their loss term and each is a chain of ~30 small
            # kernels: they are dealt round-robin onto the main stream and PXL_CCT_STREAMS side streams (autograd runs
            # each decoder's backward on the stream of its forward)
            lanes = self._lanes(ul_ad_inp.device)
            main = torch.cuda.current_stream() if lanes else None
            for st in lanes:
                st.wait_stream(main)
            ul_ad_preds, terms = [None] * len(order), []
            # PXL_CCT_FUSED_SEAM (default on): a decoder's resize + soft-max + MSE and their backward run as the fused seam on its
            # own-resolution logits (_AuxDecoder.consistency); its resized prediction is produced only if somebody reads it
            fused = os.environ.get('PXL_CCT_FUSED_SEAM', '1') != '0' and isinstance(self.cons_criterion, MSELoss)

            def branch(ad):
                got = ad.consistency(ul_ad_inp, ul_main_pred, ul_ad_gt, size) if fused else None
                if got is not None:
                    return got[1], got[0]
                pred, act = ad.forward(ul_ad_inp, pred_of_main_decoder=ul_main_pred, out_size=size)
                return pred, self.cons_criterion.forward(act, ul_ad_gt)

            for k, i in enumerate(order):
                ad = self.auxiliary_decoders[i]
                lane = lanes[k % (len(lanes) + 1) - 1] if lanes and k % (len(lanes) + 1) else None
                if lane is None:
                    pred, term = branch(ad)
                else:
                    with torch.cuda.stream(lane):
                        pred, term = branch(ad)
                    for t in (pred, term):
                        if torch.is_tensor(t):
                            t.record_stream(main)
                ul_ad_preds[i] = pred
                terms.append(term)
            for st in lanes:
                main.wait_stream(st)
            mark('unlabeled: decoders enqueued')
            cons = terms[0]
            for term in terms[1:]:
                cons = cons + term
            resulter['ul_ad_preds'] = _LazyPreds(ul_ad_preds)
            resulter['cons_loss'] = torch.mean(cons) / len(ul_ad_preds)
        else:
            resulter['ul_ad_preds'] = None
            resulter['cons_loss'] = None
        return resulter, debugger


class SSLCCT(ssl_base._SSLBase):
    NAME = 'ssl_cct'
    SUPPORTED_TASK_TYPES = [CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.main_model = self.auxiliary_decoders = None
        self.model = self.optimizer = self.lrer = self.criterion = self.cons_criterion = None
        if self.args.unlabeled_batch_size > 0:
            if self.args.cons_scale < 0:
                logger.log_err('The argument - cons_scale - is not set (or invalid)\n'
                               'Please set - cons_scale >= 0 - for training\n')
            elif self.args.cons_rampup_epochs < 0:
                logger.log_err('The argument - cons_rampup_epochs - is not set (or invalid)\n'
                               'Please set - cons_rampup_epochs >= 0 - for training\n')
            if self.args.ad_lr_scale < 0:
                logger.log_err('The argument - ad_lr_scale - is not set (or invalid)\n'
                               'Please set - ad_lr_scale >= 0 - for training\n')
        else:
            self.args.ad_lr_scale = 0

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        a = self.args
        self.cons_criterion = MSELoss()
        self.criterion = criterion_funcs[0](a)
        self.criterions = {'criterion': self.criterion, 'cons_criterion': self.cons_criterion}
        self.main_model = func.create_model(model_funcs[0], 'main_model', args=a).module
        up, cin, cout = task_func.sslcct_ad_upsample_scale(), task_func.sslcct_ad_in_channels(), task_func.sslcct_ad_out_channels()
        dt = getattr(a, 'engine_dtype', 'bf16')
        kw = dict(engine_dtype=torch.float32 if dt in ('fp32', 'f32') else torch.bfloat16)
        g = lambda name, default: getattr(a, name, default)
        decoders = \
            [VATDecoder(up, cin, cout, xi=g('vat_dec_xi', 1e-6), eps=g('vat_dec_eps', 2.0), **kw) for _ in range(g('vat_dec_num', 0))] + \
            [DropOutDecoder(up, cin, cout, drop_rate=g('drop_dec_rate', 0.5), spatial_dropout=g('drop_dec_spatial', True), **kw)
             for _ in range(g('drop_dec_num', 0))] + \
            [CutOutDecoder(up, cin, cout, erase=g('cut_dec_erase', 0.4), **kw) for _ in range(g('cut_dec_num', 0))] + \
            [ContextMaskingDecoder(up, cin, cout, **kw) for _ in range(g('context_dec_num', 0))] + \
            [ObjectMaskingDecoder(up, cin, cout, **kw) for _ in range(g('object_dec_num', 0))] + \
            [FeatureDropDecoder(up, cin, cout, **kw) for _ in range(g('fd_dec_num', 0))] + \
            [FeatureNoiseDecoder(up, cin, cout, uniform_range=g('fn_dec_uniform', 0.3), **kw) for _ in range(g('fn_dec_num', 0))]
        self.auxiliary_decoders = nn.ModuleList(decoders)
        wrapped = WrappedCCTModel(a, self.main_model, self.auxiliary_decoders, self.criterion, self.cons_criterion,
                                  task_func.sslcct_activate_ad_preds)
        self.model = patch_replication_callback(func.RankModel(wrapped))
        pdist.attach(self.model)
        self.models = {'model': self.model}
        self.optimizer = optimizer_funcs[0](self.model.module.param_groups)
        self.optimizers = {'optimizer': self.optimizer}
        self.lrer = lrer_funcs[0](self.optimizer)
        self.lrers = {'lrer': self.lrer}

    def _labeled_stream(self):
        if not hasattr(self, '_l_stream'):
            # One rank only: with Sync-BN the labeled and the unlabeled pass of the SAME network would issue their statistics
            # exchanges on ONE exchange context from two streams -- its two slots (epoch parity, csrc/peer.hip) order two exchanges in
            # flight, not the three that two unordered streams can produce.  Several ranks run the reference's order on one stream
            # (PXL_CCT_SPLIT_BACKWARD=force overrides, for runs without Sync-BN).
            mode = os.environ.get('PXL_CCT_SPLIT_BACKWARD', '1')
            on = mode != '0' and torch.cuda.is_available() and (mode == 'force' or not pdist.is_distributed())
            self._l_stream = streams.role_stream(streams.SIDE) if on else None
        return self._l_stream

    def train_step(self, inp, gt, cur_step, total_rampup_steps):
        """One iteration of ssl_cct.py:226-282 on device-resident tuples -> dict(task_loss, cons_loss)."""
        lbs = self.args.labeled_batch_size
        ramp = func.sigmoid_rampup(cur_step, total_rampup_steps)
        self.optimizer.zero_grad()
        l_gt = func.split_tensor_tuple(gt, 0, lbs)
        l_inp = func.split_tensor_tuple(inp, 0, lbs)
        has_ul = self.args.unlabeled_batch_size > 0
        # The labeled and the unlabeled pass are separate forwards of the same model (two BN batches) whose losses are
        # simply added: d(task + cons) = d(task) + d(cons).  The labeled pass -- forward AND backward -- runs on a side
        # stream; the unlabeled forward starts once the labeled FORWARD is done (same running-statistics order as the
        # reference) and overlaps with the labeled backward; the two backward passes accumulate into the same gradient
        # buffers and stay ordered.  PXL_CCT_SPLIT_BACKWARD=0: one backward over the sum, as the reference.
        mark('step start')
        side = self._labeled_stream() if has_ul else None
        main = torch.cuda.current_stream() if side is not None else None
        # PXL_CCT_LATE_LBWD (default on): the labeled BACKWARD is enqueued after the unlabeled forward of the main model (and the
        # start of G-Cutout's mask copy) instead of before it.  The host then reaches the copy's event with the labeled backward
        # and six decoders queued behind it: the contour search on the host (~2.3 ms) no longer leaves the GPU idle, and the
        # labeled backward runs beside the unlabeled forward and the decoders instead of in front of them.
        late = side is not None and os.environ.get('PXL_CCT_LATE_LBWD', '1') != '0'
        # PXL_CCT_CONCURRENT (default on, one rank): the labeled pass -- forward AND backward -- and the unlabeled pass run as two independent
        # chains on two streams from the first launch to the optimizer.  A 4-image pass at 33 x 33 launches grids of 35-70 workgroups
        # on 256 CUs; beside each other two such kernels run at 1.3-1.5 x the serial rate (profiles/r06_h_cct.txt).  What the chains
        # share is separated by hand: the weights are packed before the streams fork; the unlabeled pass parks its BatchNorm
        # running-statistics update (engine.detour_running) and folds it in behind the labeled pass's own (the reference's order:
        # labeled, then unlabeled); the labeled backward accumulates into a gradient buffer and a scratch of its own
        # (engine.side_backward_buffers), added onto the parameters' gradients before the optimizer.  Not with Sync-BN / a gradient
        # exchange: two passes issuing collectives from two streams have no common order across ranks.
        core = getattr(self.model.module.main_model, 'model', None)
        conc = late and os.environ.get('PXL_CCT_CONCURRENT', '1') != '0' and pdist.world_size() == 1 and \
            hasattr(core, 'side_backward_buffers') and core.training and not core.freeze_bn and l_inp[0].is_cuda and \
            getattr(core, '_post_backward_hook', None) is None
        if core is not None and (getattr(core, '_running_detour', None) is not None or getattr(core, '_alt_backward', None) is not None):
            core._running_detour = core._alt_backward = None       # (an iteration that raised half-way: do not inherit its state)
        if conc:
            for t in (l_inp[0], func.split_tensor_tuple(inp, lbs, self.args.batch_size)[0]):
                core._plan(t.shape[0], t.shape[2], t.shape[3], None, inference=False)
                core._ensure_packed()
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                l_res, _ = self.model.forward(l_inp, l_gt, False)
                task_loss = tool.dict_value(l_res, 'task_loss', err=True).mean()
                fwd_done = torch.cuda.Event()
                fwd_done.record()
                mark('labeled: forward enqueued (side)')
                if not late:
                    task_loss.backward()
                    mark('labeled: backward enqueued (side)')
                if conc:
                    l_plan = core._cur
                    alt = core.side_backward_buffers(l_plan)
                    alt[0].zero_()
            if not conc:
                main.wait_event(fwd_done)
            if late:
                def labeled_backward():
                    if conc:
                        main.wait_event(fwd_done)       # (the labeled pass has made its running-statistics update)
                        core.fold_running()
                    with torch.cuda.stream(side):
                        if conc:
                            core._alt_backward = alt
                        try:
                            task_loss.backward()
                        finally:
                            if conc:
                                core._alt_backward = None
                        mark('labeled: backward enqueued (side)')
                self.model.module.after_main_forward = labeled_backward
            if conc:
                core.detour_running()
            for v in l_res.values():
                for t in (v if isinstance(v, (tuple, list)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(main)
            task_loss.record_stream(main)
        else:
            l_res, _ = self.model.forward(l_inp, l_gt, False)
            task_loss = tool.dict_value(l_res, 'task_loss', err=True).mean()
        ul_res = None
        if has_ul:
            ul_gt = func.split_tensor_tuple(gt, lbs, self.args.batch_size)
            ul_inp = func.split_tensor_tuple(inp, lbs, self.args.batch_size)
            ul_res, _ = self.model.forward(ul_inp, ul_gt, True)
            pending, self.model.module.after_main_forward = getattr(self.model.module, 'after_main_forward', None), None
            if pending is not None:         # (the wrapped model did not reach its decoders: nothing ran the labeled backward yet)
                pending()
            cons_loss = ramp * self.args.cons_scale * tool.dict_value(ul_res, 'cons_loss', err=True).mean()
        else:
            cons_loss = torch.zeros((), device=task_loss.device)
        if side is not None:
            if not conc:
                main.wait_stream(side)          # the labeled backward has finished accumulating
            cons_loss.backward()
            if conc:                            # the labeled pass's gradients join the unlabeled pass's
                main.wait_stream(side)
                core.ensure_grad_views()
                core._store.grads.add_(alt[0])
            mark('unlabeled: backward enqueued')
        else:
            (task_loss + cons_loss).backward()
        lanes = getattr(self.model.module, '_lane_streams', [])
        for st in lanes:                    # the decoders' backward ran on their lanes and wrote the flat gradient buffers
            torch.cuda.current_stream().wait_stream(st)
        self.optimizer.step()
        mark('optimizer enqueued')
        if not self.args.is_epoch_lrer:
            self.lrer.step()
        return dict(task_loss=task_loss.detach(), cons_loss=cons_loss.detach()), l_res, ul_res

    def _train(self, data_loader, epoch):
        self.meters.reset()
        self.model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            cur_step = len(data_loader) * epoch + idx
            losses, _, _ = self.train_step(inp, gt, cur_step, len(data_loader) * self.args.cons_rampup_epochs)
            for k, v in losses.items():
                self.meters.update(k, v)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  task-{4}\t=>\ttask-loss: {5:.6f}\tcons-loss: {6:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['task_loss'].avg), float(self.meters['cons_loss'].avg)))
        if self.args.is_epoch_lrer:
            self.lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_cct.py:307-350: the wrapped model's main branch in eval mode, task loss + metrics."""
        self.meters.reset()
        self.model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            resulter, _ = self.model.forward(inp, gt, False)
            self.meters.update('task_loss', tool.dict_value(resulter, 'task_loss', err=True).mean().detach())
            self.task_func.metrics(tool.dict_value(resulter, 'activated_pred'), gt, inp, self.meters, id_str='task')
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n  task-{4}\t=>\ttask-loss: {5:.6f}\t'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['task_loss'].avg)))
        self._log_validation_metrics(['task'])

    def _save_checkpoint(self, epoch):
        state = {'algorithm': self.NAME, 'epoch': epoch, 'model': self.model.state_dict(),
                 'optimizer': self.optimizer.state_dict(), 'lrer': self.lrer.state_dict()}
        torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

    def _load_checkpoint(self):
        checkpoint = torch.load(self.args.resume, map_location='cpu')
        found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
        if found != self.NAME:
            logger.log_err('Unmatched SSL algorithm format in checkpoint => required: {0} - given: {1}\n'
                           .format(self.NAME, found))
        self.model.load_state_dict(checkpoint['model'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])      # ssl_cct.py:374-376
        self.lrer.load_state_dict(checkpoint['lrer'])
        self.main_model = self.model.module.main_model
        self.auxiliary_decoders = self.model.module.auxiliary_decoders
        return checkpoint['epoch']
