"""CPU oracle for the GCT flaw-map pipeline.

TEST INFRASTRUCTURE ONLY -- the checker, never the product (see torch_oracle.py for the rules).

Plain torch / numpy / scipy restatement of the reference's modules in
pixelssl/ssl_algorithm/ssl_gct.py and pixelssl/nn/module/gaussian_blur.py, each function citing the
lines it follows.  PINNED: oracle/make_golden_gct.py runs the reference's own FDGTGenerator,
FlawmapHandler, DCGTGenerator and FlawDetectorCriterion classes on seeded inputs, asserts this file
reproduces them and stores their outputs in tests/golden/gct_flawmap_65.pt.
"""
import math

import numpy as np
import scipy.ndimage
import torch
import torch.nn.functional as F


def odd_ksize(im_size, div):
    """ssl_gct.py:633-635, 701-707: int(im_size / div), made odd by adding one."""
    k = int(im_size / div)
    return k + 1 if k % 2 == 0 else k


def gaussian_kernel2d(ksize):
    """gaussian_blur.py:56-61: sigma = 0.3*((k-1)/2 - 1) + 0.8, scipy gaussian_filter of a centred delta."""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    n = np.zeros((ksize, ksize))
    n[ksize // 2, ksize // 2] = 1
    return scipy.ndimage.gaussian_filter(n, sigma)


def gaussian_taps1d(ksize):
    """The same kernel is rank 1: outer(a, a) with a = the 1-D filter of a 1-D delta (checked in make_golden_gct)."""
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    d = np.zeros(ksize)
    d[ksize // 2] = 1
    return scipy.ndimage.gaussian_filter1d(d, sigma)


def gaussian_blur(x, ksize):
    """GaussianBlurLayer.forward (gaussian_blur.py:25-29,37-54) for single-channel maps [B,1,H,W]."""
    k = torch.from_numpy(gaussian_kernel2d(ksize)).to(x.dtype).view(1, 1, ksize, ksize)
    p = math.floor(ksize / 2)
    return F.conv2d(F.pad(x, (p, p, p, p), mode="reflect"), k)


def onehot_ignore(gt, num_classes, ignore_index=255):
    """task/sseg/func.py:179-192 (sslgct_prepare_task_gt_for_fdgt): one-hot of the labels, ignored pixels all zero;
    unlabeled samples carry -1 and also give an all-zero one-hot."""
    lab = gt.long().squeeze(1)
    valid = (lab != ignore_index) & (lab >= 0) & (lab < num_classes)
    oh = F.one_hot(lab.clamp(0, num_classes - 1), num_classes).permute(0, 3, 1, 2).float()
    return oh * valid.unsqueeze(1).float()


def fdgt(pred, gt_onehot, im_size, mu=0.5, nu=1):
    """FDGTGenerator.forward (ssl_gct.py:714-728)."""
    diff = torch.abs(gt_onehot - pred.detach())
    diff = torch.sum(diff, dim=1, keepdim=True) * mu
    diff = gaussian_blur(diff, odd_ksize(im_size, 8))
    for _ in range(nu):
        dil = F.max_pool2d(F.pad(diff, (1, 1, 1, 1), mode="reflect"), 3, 1, 0)
        diff = gaussian_blur(dil, odd_ksize(im_size, 4))
    dmax = diff.amax(dim=(1, 2, 3), keepdim=True)
    dmin = diff.amin(dim=(1, 2, 3), keepdim=True)
    return (diff - dmin) / (dmax - dmin + 1e-9)


def flawmap_handle(flawmap, im_size, clip_threshold=0.1):
    """FlawmapHandler.forward (ssl_gct.py:641-657).  Returns (handled map, the clamped input): the reference clamps
    its argument IN PLACE (`flawmap.data.mul_`), which the FD loss of step 2 later sees."""
    clamped = flawmap * (flawmap >= 0).float()
    fm = gaussian_blur(clamped, odd_ksize(im_size, 16))
    fmax = fm.amax(dim=(1, 2, 3), keepdim=True)
    fmin = fm.amin(dim=(1, 2, 3), keepdim=True)
    fm = fm * (fmax > clip_threshold).float()
    return (fm - fmin) / (fmax - fmin + 1e-9), clamped


def dcgt(l_pred, r_pred, l_fm, r_fm, dc_threshold=0.6):
    """DCGTGenerator.forward (ssl_gct.py:668-689).  Returns l_dc_gt, r_dc_gt, both_bad and the updated flaw maps."""
    both_bad = ((l_fm > dc_threshold) & (r_fm > dc_threshold)).float()
    l2 = l_fm * (l_fm <= dc_threshold).float() + (l_fm > dc_threshold).float()
    r2 = r_fm * (r_fm <= dc_threshold).float() + (r_fm > dc_threshold).float()
    l_mask = (r2 >= l2).float()
    r_mask = (l2 >= r2).float()
    return (l_mask * l_pred + (1 - l_mask) * r_pred, r_mask * r_pred + (1 - r_mask) * l_pred, both_bad, l2, r2)


def fd_criterion(pred, gt):
    """FlawDetectorCriterion.forward (ssl_gct.py:617-621), reduction=True."""
    return torch.mean(F.mse_loss(pred, gt, reduction="none"), dim=(1, 2, 3))


def synthetic_case(seed, B=3, C=21, size=65):
    """Seeded inputs shared by make_golden_gct.py and the tests: softmax maps, labels with ignore / unlabeled
    samples, raw flaw-detector outputs (some negative)."""
    g = torch.Generator().manual_seed(seed)
    l_pred = torch.softmax(torch.randn(B, C, size, size, generator=g) * 2, dim=1)
    r_pred = torch.softmax(torch.randn(B, C, size, size, generator=g) * 2, dim=1)
    gt = torch.randint(0, C, (B, 1, size // 8 + 1, size // 8 + 1), generator=g).float()
    gt = F.interpolate(gt, size=(size, size), mode="nearest")
    gt[:, :, ::8, :] = 255.0
    gt[B - 1] = -1.0                                    # an unlabeled sample
    l_fm = torch.randn(B, 1, size, size, generator=g) * 0.4 + 0.3
    r_fm = torch.randn(B, 1, size, size, generator=g) * 0.4 + 0.3
    r_fm[0] = -r_fm[0].abs() * 0.01 + 0.05              # a sample whose handled map stays below the clip threshold
    return l_pred, r_pred, gt, l_fm, r_fm
