"""GCT building blocks on the device (reference: pixelssl/ssl_algorithm/ssl_gct.py).

This module currently holds the flaw-map pipeline of GCT with the reference's class names and call signatures --
FlawDetectorCriterion (:610-621), FlawmapHandler (:624-657), DCGTGenerator (:660-689), FDGTGenerator (:692-728) --
implemented on libpixelhip kernels (csrc/flawmap.hip).  Tensors are fp32 NCHW on the GPU, exactly what the task
model returns.  The SSLGCT trainer itself (three optimizers, FlawDetector with IBNorm) is the next step
(DESIGN.md section 7); `ssl_gct` is therefore not exported yet.
"""
import math

import numpy as np
import scipy.ndimage
import torch
import torch.nn as nn

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr

NAME = 'ssl_gct'


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
        taps = self.taps.to(x.device)
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
        if gt.shape[1] == 1 and C != 1:
            diff = torch.empty(B, 1, H, W, device=pred.device, dtype=torch.float32)
            check(lib().pxl_absdiff_chansum(B, C, H * W, ptr(pred), ptr(gt.contiguous().float()), self.ignore_index,
                                            float(self.args.mu), ptr(diff), stream_ptr()))
        else:
            diff = torch.sum(torch.abs(gt - pred), dim=1, keepdim=True) * self.args.mu
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
