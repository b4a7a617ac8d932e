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


# -----------------------------------------------------------------------------
# Flaw detector + the SSLGCT training iteration (rows G1-G3)
# -----------------------------------------------------------------------------
import torch.nn as nn            # noqa: E402
from collections import OrderedDict   # noqa: E402
import torch_oracle as TO        # noqa: E402

FD_LAYERS = (("conv1", None, 64, 2, "ibn1"), ("conv2", 64, 128, 2, "ibn2"), ("conv2_1", 128, 128, 1, "ibn2_1"),
             ("conv3", 128, 256, 2, "ibn3"), ("conv3_1", 256, 256, 1, "ibn3_1"), ("conv4", 256, 512, 2, "ibn4"),
             ("conv4_1", 512, 512, 1, "ibn4_1"))          # ssl_gct.py:548-562, all 4x4 / pad 1 with bias


def init_fd_state(in_channels=24, seed=0):
    """Default nn.Conv2d initialisation in the reference's construction order (ssl_gct.py:548-563; IBNorm's BN half:
    gamma 1, beta 0, running stats 0/1) under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    sd = OrderedDict()
    for name, cin, cout, stride, ibn in FD_LAYERS:
        conv = nn.Conv2d(cin if cin is not None else in_channels, cout, kernel_size=4, stride=stride, padding=1)
        sd[name + ".weight"], sd[name + ".bias"] = conv.weight.detach().clone(), conv.bias.detach().clone()
        nb = int(cout * 0.5 + 0.5)
        sd[ibn + ".bnorm.weight"], sd[ibn + ".bnorm.bias"] = torch.ones(nb), torch.zeros(nb)
        sd[ibn + ".bnorm.running_mean"], sd[ibn + ".bnorm.running_var"] = torch.zeros(nb), torch.ones(nb)
        sd[ibn + ".bnorm.num_batches_tracked"] = torch.tensor(0)
    conv = nn.Conv2d(512, 1, kernel_size=4, stride=2, padding=1)
    sd["classifier.weight"], sd["classifier.bias"] = conv.weight.detach().clone(), conv.bias.detach().clone()
    return sd


def fd_is_buffer(name):
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def ibnorm(sd, prefix, x, train):
    """IBNorm.forward (ssl_gct.py:600-607): first int(C*0.5+0.5) channels SynchronizedBatchNorm2d(affine), the rest
    InstanceNorm2d(affine=False); concatenated."""
    nb = sd[prefix + ".bnorm.weight"].numel()
    xb = F.batch_norm(x[:, :nb], sd[prefix + ".bnorm.running_mean"], sd[prefix + ".bnorm.running_var"],
                      sd[prefix + ".bnorm.weight"], sd[prefix + ".bnorm.bias"], train, 0.1, 1e-5)
    if nb == x.shape[1]:
        return xb
    return torch.cat((xb, F.instance_norm(x[:, nb:], eps=1e-5)), 1)


def fd_forward(sd, task_inp, task_pred, train=True):
    """FlawDetector.forward (ssl_gct.py:567-585).  Running statistics in `sd` are updated in place (train)."""
    x = torch.cat((task_inp, task_pred), dim=1)
    for name, _, _, stride, ibn in FD_LAYERS:
        x = F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride, 1)
        x = F.leaky_relu(ibnorm(sd, ibn, x, train), 0.2)
    x = F.conv2d(x, sd["classifier.weight"], sd["classifier.bias"], 2, 1)
    return F.interpolate(x, size=task_pred.shape[2:], mode="bilinear", align_corners=True)


class GCTOracleTrainer:
    """SSLGCT._train body (ssl_gct.py:186-269 + _task_model_iter :401-480), ssl_mode 'gct', one iteration per call.
    Two independently initialised task models that see the same input, one flaw detector, three optimizers."""

    def __init__(self, l_state, r_state, fd_state, hp):
        self.hp = dict(fc_ssl_scale=1.0, dc_ssl_scale=100.0, dc_threshold=0.6, dc_rampup_iters=0, fd_lr=1e-4,
                       fd_scale=10.0, mu=0.5, nu=1, im_size=65, ignore_index=255, max_iters=100)
        self.hp.update(hp)
        self.l = TO.OracleTrainer(l_state, dict(max_iters=self.hp["max_iters"]))
        self.r = TO.OracleTrainer(r_state, dict(max_iters=self.hp["max_iters"]))
        self.fd_sd = fd_state
        self.fd_leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in fd_state.items()
                                     if not fd_is_buffer(k))
        self.fd_opt = torch.optim.Adam(list(self.fd_leaves.values()), lr=self.hp["fd_lr"], betas=(0.9, 0.99))
        self.it = 0

    def _fd_run(self):
        run = OrderedDict(self.fd_sd)
        run.update(self.fd_leaves)
        return run

    def _fd(self, inp, prob):
        run = self._fd_run()
        out = fd_forward(run, inp, prob, train=True)
        for k in self.fd_sd:                      # running statistics evolve with every forward
            if fd_is_buffer(k):
                self.fd_sd[k] = run[k]
        return out

    def _task_iter(self, tr, x, gt, lbs, dc_gt, fc_mask, ramp):
        """_task_model_iter + the optimizer step that follows it (ssl_gct.py:231-241, 401-480)."""
        hp = self.hp
        leaves = TO._param_leaves(tr.sd)
        run = TO._with_leaves(tr.sd, leaves)
        logits, prob, _, _ = TO.deeplabv2_forward(run, x, train=True)
        for k in tr.sd:
            if TO.is_buffer(k):
                tr.sd[k] = run[k]
        for v in self.fd_leaves.values():          # frozen: requires_grad False during step 1
            v.requires_grad_(False)
        flawmap = self._fd(x, prob)
        for v in self.fd_leaves.values():
            v.requires_grad_(True)
        task = TO.sseg_criterion(logits[:lbs], gt[:lbs], hp["ignore_index"]).mean()
        fc = hp["fc_ssl_scale"] * torch.mean(fc_mask * F.mse_loss(flawmap, torch.zeros_like(flawmap), reduction="none"))
        dc = ramp * hp["dc_ssl_scale"] * F.mse_loss(prob, dc_gt)
        (task + fc + dc).backward()
        grads = OrderedDict((k, v.grad) for k, v in leaves.items())
        with torch.no_grad():
            TO.sgd_step(tr.sd, grads, tr.mom, tr._lrs(), tr.hp["momentum"], tr.hp["weight_decay"])
        tr.it += 1
        return float(task.detach()), float(fc.detach()), float(dc.detach())

    def gct_step(self, x, gt, lbs):
        hp = self.hp
        ramp = TO.sigmoid_rampup(self.it, hp["dc_rampup_iters"])
        C = 21
        # ---- step 0 (ssl_gct.py:203-227)
        with torch.no_grad():
            _, l_prob, _, _ = TO.deeplabv2_forward(self.l.sd, x, train=True)
            _, r_prob, _, _ = TO.deeplabv2_forward(self.r.sd, x, train=True)
        l_flawmap = self._fd(x, l_prob)
        r_flawmap = self._fd(x, r_prob)
        with torch.no_grad():
            l_h, l_clamped = flawmap_handle(l_flawmap.detach(), hp["im_size"])
            r_h, r_clamped = flawmap_handle(r_flawmap.detach(), hp["im_size"])
            l_flawmap.data.copy_(l_clamped)        # FlawmapHandler mutates the flaw maps in place (:643-645)
            r_flawmap.data.copy_(r_clamped)
            l_dc_gt, r_dc_gt, both_bad, _, _ = dcgt(l_prob, r_prob, l_h, r_h, hp["dc_threshold"])
        # ---- step 1 (ssl_gct.py:229-241)
        lt, lfc, ldc = self._task_iter(self.l, x, gt, lbs, l_dc_gt, both_bad, ramp)
        rt, rfc, rdc = self._task_iter(self.r, x, gt, lbs, r_dc_gt, both_bad, ramp)
        # ---- step 2 (ssl_gct.py:246-269): ground truth of the flaw detector on the labeled samples, STEP-0 predictions
        with torch.no_grad():
            oh = onehot_ignore(gt[:lbs], C, hp["ignore_index"])
            l_gt_fm = fdgt(l_prob[:lbs], oh, hp["im_size"], hp["mu"], hp["nu"])
            r_gt_fm = fdgt(r_prob[:lbs], oh, hp["im_size"], hp["mu"], hp["nu"])
        l_fd = hp["fd_scale"] * fd_criterion(l_flawmap[:lbs], l_gt_fm).mean()
        r_fd = hp["fd_scale"] * fd_criterion(r_flawmap[:lbs], r_gt_fm).mean()
        self.fd_opt.zero_grad()
        ((l_fd + r_fd) / 2).backward()
        for gp in self.fd_opt.param_groups:
            gp["lr"] = TO.poly_lr(hp["fd_lr"], self.it + 1, hp["max_iters"], 0.9)
        self.fd_opt.step()
        self.it += 1
        return dict(l_task_loss=lt, l_fc_loss=lfc, l_dc_loss=ldc, r_task_loss=rt, r_fc_loss=rfc, r_dc_loss=rdc,
                    l_fd_loss=float(l_fd.detach()), r_fd_loss=float(r_fd.detach()))

    def fd_state(self):
        out = OrderedDict((k, v.clone()) for k, v in self.fd_sd.items())
        out.update((k, v.detach().clone()) for k, v in self.fd_leaves.items())
        return out
