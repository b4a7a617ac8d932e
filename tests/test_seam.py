"""The fused training seam (csrc/head.hip: pxl_head_loss; engine.DeferredHead; functional.head_losses).

gpu: (1) the kernel against a torch fp32 restatement of what the reference computes between the low-resolution logits
and their gradient -- F.interpolate(bilinear, align_corners) (deeplab_v2.py:32), per-sample CE with ignore_index over
ALL pixels (task/sseg/criterion.py:24-38) for student and teacher, nn.MSELoss between the two predictions
(ssl_mt.py:179-184), autograd's backward -- on ragged sizes, with ignored / out-of-range labels and partial sample
ranges; (2) a whole MT / SupOnly iteration through the fused seam against the same iteration through the generic path
(full-resolution planes + autograd): same losses, same weights; (3) a deferred pass materialises the planes forward()
returns.  The reference fixtures themselves run through the fused seam in tests/test_multistep.py."""
import argparse
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def close(a, b, rtol, atol=1e-8):
    """||a - b|| <= rtol * ||b|| + atol * sqrt(n): the absolute term covers tensors that ARE (numerically) zero -- a BN beta
    after three tiny updates is ~1e-6 and its gradient is a cancelling sum over every pixel, which the two paths (and two runs
    of the same path: fp32 atomics) add up in different orders: measured 2.6e-7 in norm on 256 elements, i.e. 1.6e-8 per
    element; the fp32 floor is 5e-8 per element"""
    d = (a.double() - b.double()).norm().item()
    return d <= rtol * b.double().norm().item() + atol * (b.numel() ** 0.5)


def _torch_seam(s_low, t_low, gt, n_ce, lo, hi, w_ce, w_mse, size, align, ignore=255):
    s_low = s_low.detach().clone().requires_grad_(True)
    zs = F.interpolate(s_low, size=size, mode="bilinear", align_corners=align)
    B, C, H, W = zs.shape
    loss = zs.sum() * 0
    ce_s = ce_t = None
    if n_ce:
        lab = gt.long().view(n_ce, H, W).clone()
        lab[(lab < 0) | (lab >= C)] = ignore
        ce_s = F.cross_entropy(zs[:n_ce], lab, ignore_index=ignore, reduction="none").sum((1, 2)) / (H * W)
        loss = loss + w_ce * ce_s.sum()
    mse = None
    if t_low is not None:
        zt = F.interpolate(t_low, size=size, mode="bilinear", align_corners=align)
        if n_ce:
            ce_t = F.cross_entropy(zt[:n_ce], lab, ignore_index=ignore, reduction="none").sum((1, 2)) / (H * W)
        if hi > lo:
            mse = F.mse_loss(zs[lo:hi], zt[lo:hi])
            loss = loss + w_mse * mse
    loss.backward()
    return ce_s, ce_t, mse, s_low.grad


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["auto", "rows"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [
    dict(B=3, C=21, h=9, w=9, H=129, W=129, align=True, n_ce=2, lo=2, hi=3, teacher=True),
    dict(B=4, C=21, h=5, w=7, H=65, W=97, align=True, n_ce=2, lo=0, hi=4, teacher=True),       # ragged, consistency on all
    dict(B=2, C=21, h=9, w=9, H=129, W=129, align=True, n_ce=2, lo=0, hi=0, teacher=False),    # SupOnly
    dict(B=3, C=7, h=8, w=8, H=50, W=61, align=False, n_ce=1, lo=1, hi=3, teacher=True),       # align_corners=False, odd sizes
    dict(B=2, C=21, h=33, w=33, H=513, W=513, align=True, n_ce=1, lo=1, hi=2, teacher=True),   # the BASELINE geometry
    dict(B=2, C=21, h=9, w=7, H=65, W=67, align=True, n_ce=2, lo=0, hi=2, teacher=True),       # scale 8 x 11: cells 11 wide (padded to 16 lanes)
    dict(B=1, C=21, h=6, w=6, H=70, W=45, align=False, n_ce=1, lo=0, hi=1, teacher=True),      # align_corners=False on the cell kernel
])
def test_head_loss_kernel_vs_torch(dtype, case, kernel, monkeypatch):
    """kernel = auto: the cell-wise kernel where it applies (21 classes, up-sampling factor >= 6), else the row-wise one;
    kernel = rows: the row-wise kernel everywhere (PXL_HEAD_LOSS_CELLS=0)."""
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    if kernel == "rows":
        monkeypatch.setenv("PXL_HEAD_LOSS_CELLS", "0")
    else:
        monkeypatch.delenv("PXL_HEAD_LOSS_CELLS", raising=False)
    c = argparse.Namespace(**case)
    g = torch.Generator().manual_seed(11 + c.h * c.W)
    Cp = 32
    s = torch.randn(c.B, c.C, c.h, c.w, generator=g) * 2
    t = torch.randn(c.B, c.C, c.h, c.w, generator=g) * 2
    if dtype == torch.bfloat16:      # the kernel sees bf16 maps: the torch restatement starts from the same values
        s, t = s.bfloat16().float(), t.bfloat16().float()
    gt = torch.randint(0, c.C, (c.n_ce, 1, c.H, c.W), generator=g).float()
    gt[:, :, ::7, ::5] = 255                    # ignored
    gt[:, :, 3::11, 2::13] = -1                 # "unlabeled" marker of the sseg loaders: outside [0, C) -> ignored too
    w_ce, w_mse = 1.0 / c.n_ce, 0.37

    def nhwc(x):
        out = torch.zeros(c.B, c.h, c.w, Cp)
        out[..., :c.C] = x.permute(0, 2, 3, 1)
        return out.to(dtype).to(DEV).contiguous()
    s_low, t_low = nhwc(s), (nhwc(t) if c.teacher else None)
    dlow = torch.full((c.B, c.h, c.w, Cp), 7.0, device=DEV, dtype=dtype)
    ws = torch.empty(lib().pxl_upsample_bwd_workspace(c.B, c.w, c.C, c.H), device=DEV, dtype=torch.uint8)
    sums = torch.full((2 * c.B + 1,), 9.0, device=DEV)
    check(lib().pxl_head_loss(dtype_code(dtype), c.B, c.h, c.w, Cp, c.C, c.H, c.W, int(c.align), ptr(s_low), ptr(t_low),
                              ptr(gt.to(DEV)), 255, c.n_ce, c.lo, c.hi, w_ce, w_mse, ptr(dlow), ptr(ws), ws.numel(), ptr(sums),
                              stream_ptr()))
    ce_s, ce_t, mse, grad = _torch_seam(s, t if c.teacher else None, gt, c.n_ce, c.lo, c.hi, w_ce, w_mse, (c.H, c.W), c.align)
    sums = sums.cpu()
    assert torch.allclose(sums[:c.n_ce], ce_s, rtol=2e-5, atol=1e-6), (sums[:c.n_ce], ce_s)
    if c.teacher:
        assert torch.allclose(sums[c.B:c.B + c.n_ce], ce_t, rtol=2e-5, atol=1e-6)
        if c.hi > c.lo:
            assert abs(sums[2 * c.B].item() - mse.item()) <= 2e-5 * abs(mse.item())
    assert (sums[c.n_ce:c.B] == 0).all() and (sums[c.B + c.n_ce:2 * c.B] == 0).all()
    got = dlow.float().cpu()
    assert (got[..., c.C:] == 0).all(), "padded channels of the gradient must be zero"
    r = rel(got[..., :c.C].permute(0, 3, 1, 2), grad)
    print("head_loss [%s] %s %s: d(low) rel err %.2e" % (kernel, dtype, case, r))
    assert r < (2e-5 if dtype == torch.float32 else 4e-3)       # bf16: the OUTPUT is rounded to bf16


def _mt_algo(dtype, size, lbs, ubs, cons_for_labeled=False):
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    a = argparse.Namespace(backbone=(1, 1, 1, 1), output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4, momentum=0.9,
                           weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1, epochs=1, iters_per_epoch=8,
                           ignore_index=255, labeled_batch_size=lbs, unlabeled_batch_size=ubs, batch_size=lbs + ubs,
                           ignore_unlabeled=ubs == 0, is_epoch_lrer=False, log_freq=1000, task="sseg", engine_dtype=dtype, gpus=1,
                           im_size=size, gaussian_noise_std=None, cons_for_labeled=cons_for_labeled, cons_scale=1.0,
                           cons_rampup_epochs=3, ema_decay=0.99)
    return a, P, popt, plr


def _condition(core):
    """bottleneck-output BN gammas x 0.1 (torch_oracle.condition_state): without it a freshly initialised trunk amplifies
    a 1-ulp difference of the first iteration into a percent-level difference of the second one"""
    with torch.no_grad():
        for name, prm in core.named_parameters():
            if name.endswith("bn3.weight"):
                prm.mul_(0.1)
    core.mark_params_changed()


class _Shallow:
    """sseg.model.DeepLabV2 on a shallow trunk (1 block per stage): the seam does not care about depth"""

    @staticmethod
    def patch(monkeypatch):
        import pixelssl_amd.sseg.model as M
        import pixelssl_amd.utils.logger as L
        orig = M.DeepLabV2.__init__

        def init(self, args):
            name, args.backbone = args.backbone, "resnet101"
            import pixelssl_amd.engine as E
            real = E.RESNET_LAYERS["resnet101"]
            E.RESNET_LAYERS["resnet101"] = tuple(name)
            try:
                orig(self, args)
            finally:
                E.RESNET_LAYERS["resnet101"] = real
                args.backbone = name
        monkeypatch.setattr(M.DeepLabV2, "__init__", init)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("cons_for_labeled", [False, True])
def test_mt_step_fused_seam_equals_generic_path(dtype, cons_for_labeled, monkeypatch):
    import torch_oracle as TO
    _Shallow.patch(monkeypatch)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PXL_FUSE_SEAM", mode)
        a, P, popt, plr = _mt_algo(dtype, 129, 2, 2, cons_for_labeled)
        torch.manual_seed(5)
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(a, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(a)},
                                            {"model": plr.polynomiallr(a)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
        gen = torch.Generator().manual_seed(3)
        s_core.reset_parameters(gen)
        t_core.reset_parameters(gen)
        _condition(s_core)
        _condition(t_core)
        algo.s_model.train()
        algo.t_model.train()
        losses = []
        for i in range(3):
            x, gt = TO.synthetic_batch(4, 129, 2, seed=40 + i, block=32)
            out, s_res, t_res = algo.train_step((x.to(DEV),), (gt.to(DEV),), i + 2, 6)
            losses.append({k: v.item() for k, v in out.items()})
        results[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in s_core.state_dict().items()},
                         {k: v.detach().float().cpu().clone() for k, v in t_core.state_dict().items()},
                         type(s_res).__name__)
    assert results["1"][3] == "_DeferredResulter" and results["0"][3] != "_DeferredResulter", "the switch selects the path"
    for i, (lf, lg) in enumerate(zip(results["1"][0], results["0"][0])):
        print("mt %s iteration %d fused %s generic %s" % (dtype, i, lf, lg))
        # (bf16: BN statistics are fp32 atomics whose order moves sums by an ulp, which flips bf16 roundings downstream:
        # two runs of the SAME path differ by ~2e-4 in a loss)
        tol = (2e-6 if i == 0 else 2e-4) if dtype == "fp32" else (1e-3 if i == 0 else 5e-3)
        for k in lg:
            assert abs(lf[k] - lg[k]) <= tol * abs(lg[k]) + 1e-8, (i, k, lf, lg)
    for which in (1, 2):
        for k, v in results["0"][which].items():
            if "num_batches" in k:
                continue
            # (bf16: gradients carry bf16 noise, a 1e-6 BN beta moves by 1e-7 between two runs of the SAME path)
            assert close(results["1"][which][k], v, 2e-5 if dtype == "fp32" else 2e-3, 5e-8 if dtype == "fp32" else 5e-7), \
                (which, k, rel(results["1"][which][k], v))


@pytest.mark.gpu
def test_suponly_step_fused_seam_equals_generic_path(monkeypatch):
    import torch_oracle as TO
    _Shallow.patch(monkeypatch)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PXL_FUSE_SEAM", mode)
        a, P, popt, plr = _mt_algo("fp32", 129, 3, 0)
        algo = P.ssl_algorithm.ssl_null.ssl_null(a, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(a)},
                                                {"model": plr.polynomiallr(a)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        core = algo.model.module.model
        core.reset_parameters(torch.Generator().manual_seed(3))
        _condition(core)
        algo.model.train()
        losses = []
        for i in range(2):
            x, gt = TO.synthetic_batch(3, 129, 3, seed=50 + i, block=32)
            loss, res = algo.train_step((x.to(DEV),), (gt.to(DEV),))
            losses.append(loss.item())
        results[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in core.state_dict().items()}, res)
    for i, (lf, lg) in enumerate(zip(results["1"][0], results["0"][0])):
        print("suponly iteration %d fused %.8f generic %.8f" % (i, lf, lg))
        assert abs(lf - lg) <= (2e-6 if i == 0 else 2e-4) * abs(lg)
    for k, v in results["0"][1].items():
        if "num_batches" not in k:
            assert close(results["1"][1][k], v, 2e-5), (k, rel(results["1"][1][k], v))
    # the deferred resulter hands out the planes of the LAST forward pass on demand: same values as the generic path's
    pf, pg = results["1"][2]["pred"][0], results["0"][2]["pred"][0]
    assert pf.shape == pg.shape and rel(pf, pg.detach()) < 2e-4
    af = results["1"][2]["activated_pred"][0]
    assert torch.allclose(af.sum(1), torch.ones_like(af.sum(1)), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mt_step_paired_forward_equals_two_passes(dtype, monkeypatch):
    """The paired student || teacher forward (csrc/net.cpp: pxl_net_forward_pair -- every convolution and finalize-folding
    element-wise kernel of the two networks as ONE launch) against the two separate passes on two streams: same losses,
    same weights after three iterations (the pair only changes which launch computes a tile, never how), and the pass
    really issued paired launches."""
    import torch_oracle as TO
    from pixelssl_amd._lib import lib
    _Shallow.patch(monkeypatch)
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PXL_PAIR_FORWARD", mode)
        a, P, popt, plr = _mt_algo(dtype, 129, 2, 2)
        torch.manual_seed(5)
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(a, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(a)},
                                            {"model": plr.polynomiallr(a)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
        gen = torch.Generator().manual_seed(3)
        s_core.reset_parameters(gen)
        t_core.reset_parameters(gen)
        _condition(s_core)
        _condition(t_core)
        algo.s_model.train()
        algo.t_model.train()
        losses = []
        for i in range(3):
            x, gt = TO.synthetic_batch(4, 129, 2, seed=40 + i, block=32)
            out, s_res, t_res = algo.train_step((x.to(DEV),), (gt.to(DEV),), i + 2, 6)
            losses.append({k: v.item() for k, v in out.items()})
        torch.cuda.synchronize()
        pairs = lib().pxl_net_pairs(s_core._cur.net)
        results[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in s_core.state_dict().items()},
                         {k: v.detach().float().cpu().clone() for k, v in t_core.state_dict().items()}, pairs)
    # (the fp32 engine's convolutions run on the generic register-staged kernel, which has no paired form: its lockstep pass
    # issues them one after the other and pairs only the element-wise kernels)
    assert results["0"][3] == 0 and (results["1"][3] >= 10 or dtype == "fp32"), "paired launches: %s / %s" % (results["1"][3], results["0"][3])
    for i, (lp, ls) in enumerate(zip(results["1"][0], results["0"][0])):
        print("mt %s iteration %d paired %s separate %s" % (dtype, i, lp, ls))
        tol = (2e-6 if i == 0 else 2e-4) if dtype == "fp32" else (1e-3 if i == 0 else 5e-3)
        for k in ls:
            assert abs(lp[k] - ls[k]) <= tol * abs(ls[k]) + 1e-8, (i, k, lp, ls)
    for which in (1, 2):
        for k, v in results["0"][which].items():
            if "num_batches" in k:
                continue
            assert close(results["1"][which][k], v, 2e-5 if dtype == "fp32" else 2e-3, 5e-8 if dtype == "fp32" else 5e-7), \
                (which, k, rel(results["1"][which][k], v))
