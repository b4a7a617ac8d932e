"""Validation + metrics path: confusion-matrix metrics of the sseg task (task/sseg/func.py:36-80)
and the `_validate` loops of the algorithms at arbitrary, non-square batch-1 sizes with eval-mode BN.

CPU: the numpy restatement (oracle/metrics_oracle.py) and the host arithmetic of SSEGFunc against the fixture generated
from the reference's own `SemanticSegmentationFunc.metrics` (oracle/make_golden_metrics.py).
GPU: pxl_confusion_matrix bit-exact against numpy incl. ties / NaN / ignore / unlabeled / out-of-range labels;
SSEGFunc.metrics fills the meters like the reference; SupOnly / MT `_validate` over mixed-size loaders reproduce the
oracle's confusion matrix and mIoU; the inference-plan LRU stays bounded."""
import argparse
import os
import sys

from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
FX = os.path.join(ROOT, "tests", "golden", "metrics_65.pt")
DEV = "cuda"


def test_metrics_oracle_and_host_arithmetic_reproduce_the_reference_fixture():
    import metrics_oracle as MO
    from pixelssl_amd.sseg.func import metrics_from_confusion_matrix, color_map, VOCColorize
    fx = torch.load(FX, weights_only=False)
    cm = np.zeros((21, 21), dtype=np.int64)
    for (b, h, w, seed), ref in zip(fx["shapes"], fx["per_batch"]):
        pred, gt = MO.synthetic_val_batch(b, h, w, seed)
        cm += MO.confusion_matrix(pred.numpy(), gt.numpy(), 21)
        with np.errstate(divide="ignore", invalid="ignore"):
            acc, acc_class, miou, fwiou = metrics_from_confusion_matrix(cm)
        assert (acc, acc_class, miou, fwiou) == (ref["acc"], ref["acc-class"], ref["mIoU"], ref["fwIoU"])
    assert np.array_equal(cm, fx["confusion_matrix"].numpy())
    assert fx["keys"] == ["task_confusion_matrix", "task_metric_acc", "task_metric_acc-class", "task_metric_fwIoU", "task_metric_mIoU"]
    # VOC palette: the published first entries
    assert color_map(22)[:4].tolist() == [[0, 0, 0], [128, 0, 0], [0, 128, 0], [128, 128, 0]]
    col = VOCColorize()(np.array([[0, 1], [255, 2]]))
    assert col[:, 0, 1].tolist() == [128, 0, 0] and col[:, 1, 0].tolist() == [255, 255, 255]


def _args(**kw):
    a = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4, momentum=0.9,
                           weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1, epochs=1,
                           iters_per_epoch=4, ignore_index=255, labeled_batch_size=2, unlabeled_batch_size=0,
                           ignore_unlabeled=True, is_epoch_lrer=False, log_freq=1000, task="sseg", engine_dtype="fp32",
                           cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99,
                           gaussian_noise_std=None, models={"model": "deeplabv2"})
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.gpu
def test_confusion_matrix_kernel_is_bit_exact():
    import metrics_oracle as MO
    from pixelssl_amd import functional as PF
    for b, h, w, seed in [(2, 65, 65, 301), (1, 49, 81, 302), (3, 33, 65, 303), (1, 513, 513, 304), (2, 7, 5, 305)]:
        pred, gt = MO.synthetic_val_batch(b, h, w, seed)
        # hard cases: exact ties (np.argmax keeps the first), NaN (counts as the maximum), unlabeled (-1), out-of-range
        # and fractional labels (astype(int) truncates)
        pred[0, :, 0, 0] = 0.25
        pred[0, 3, 1, 1] = float("nan")
        pred[0, 7, 1, 2] = float("nan"); pred[0, 2, 1, 2] = float("nan")
        gt[0, 0, 2, 2] = -1.0
        gt[0, 0, 2, 3] = 21.0
        gt[0, 0, 2, 4] = 3.7
        gt[0, 0, 1, 1] = 4.0; gt[0, 0, 1, 2] = 4.0; gt[0, 0, 0, 0] = 9.0
        want = MO.confusion_matrix(pred.numpy(), gt.numpy(), 21)
        got = PF.confusion_matrix(pred.to(DEV), gt.to(DEV), 21)
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), (b, h, w)
        assert want[4, 3] >= 1 and want[4, 2] >= 1 and want[9, 0] >= 1 and want[3].sum() >= 1
        am = PF.argmax_u8(pred.to(DEV)).cpu().numpy()
        assert np.array_equal(am, np.argmax(pred.numpy(), axis=1))
        acc = torch.zeros(21, 21, dtype=torch.int64, device=DEV)
        PF.confusion_matrix(pred.to(DEV), gt.to(DEV), 21, out=acc)
        PF.confusion_matrix(pred.to(DEV), gt.to(DEV), 21, out=acc)
        assert np.array_equal(acc.cpu().numpy(), 2 * want)


@pytest.mark.gpu
def test_ssegfunc_metrics_fills_the_meters_like_the_reference():
    import metrics_oracle as MO
    from pixelssl_amd.sseg.func import SSEGFunc
    from pixelssl_amd.utils import logger
    fx = torch.load(FX, weights_only=False)
    f = SSEGFunc(_args())
    meters = logger.AvgMeterSet()
    for (b, h, w, seed), ref in zip(fx["shapes"], fx["per_batch"]):
        pred, gt = MO.synthetic_val_batch(b, h, w, seed)
        f.metrics((pred.to(DEV),), (gt.to(DEV),), None, meters, id_str="task")
        got = {k: float(meters["task_metric_" + k].val) for k in ("acc", "acc-class", "mIoU", "fwIoU")}
        assert got == ref
        assert meters["task_metric_mIoU"].count == 1          # reset + update: the meter holds the running value
    assert np.array_equal(meters["task_confusion_matrix"].sum, fx["confusion_matrix"].numpy())
    assert sorted(meters.keys()) == fx["keys"]
    # softmax hook + the two-tensor ADV hook keep the reference's return contracts
    x = torch.randn(2, 21, 17, 9, device=DEV, requires_grad=True)
    p = f.sslcct_activate_ad_preds([x])[0]
    assert torch.allclose(p, torch.softmax(x.detach(), 1), atol=1e-6)
    (p * torch.arange(21, device=DEV).view(1, 21, 1, 1)).sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    (torch.softmax(xr, 1) * torch.arange(21, device=DEV).view(1, 21, 1, 1)).sum().backward()
    assert torch.allclose(x.grad, xr.grad, atol=1e-5)
    conf = torch.randn(2, 1, 17, 9, device=DEV, requires_grad=True)
    tgt = torch.randint(0, 21, (2, 1, 17, 9), device=DEV).float()
    tgt[0, 0, :3] = 255.0
    pm, gm = f.ssladv_preprocess_fcd_criterion(conf, tgt, True)
    assert torch.is_tensor(pm) and torch.is_tensor(gm) and pm.shape == conf.shape == gm.shape
    m = (tgt != 255).float()
    assert torch.equal(pm.detach(), conf.detach() * m) and torch.equal(gm, m)
    pm.sum().backward()
    assert torch.equal(conf.grad, m)
    pf, gf = f.ssladv_preprocess_fcd_criterion(conf.detach(), None, False)
    assert torch.equal(pf, conf.detach()) and torch.equal(gf, torch.zeros_like(gf))
    # the package's criterion gives the same number on the hook's pair (fused path) and on plain tensors (generic path)
    from pixelssl_amd.ssl_algorithm.ssl_adv import FCDiscriminatorCriterion
    crit = FCDiscriminatorCriterion()
    fused = crit(pm, gm)
    plain = crit(pm.detach().clone(), gm.detach().clone())
    ref = torch.nn.functional.binary_cross_entropy_with_logits(pm.detach(), gm, reduction="none").mean(dim=(1, 2, 3))
    assert torch.allclose(fused, ref, atol=1e-6) and torch.allclose(plain, ref, atol=1e-6)


class _Loader(list):
    pass


def _val_loader(sizes, seed):
    import torch_oracle as TO
    out = _Loader()
    for i, (h, w) in enumerate(sizes):
        g = torch.Generator().manual_seed(seed + i)
        x = torch.randn(1, 3, h, w, generator=g)
        _, gt = TO.synthetic_batch(1, max(h, w), 1, seed=seed + 100 + i, block=16)
        out.append(((x,), (gt[:, :, :h, :w].contiguous(),)))
    return out


@pytest.mark.gpu
def test_validate_suponly_and_mt_at_mixed_sizes_vs_oracle():
    """`_validate` of SSLNULL / SSLMT (eval-mode BN = running statistics, batch 1, non-square images of different sizes):
    task loss, confusion matrix and mIoU against the CPU oracle on the same weights.  Conditioned weights with
    non-trivial running statistics; arg-max decisions may differ from the oracle's only on pixels whose top-2 margin is
    within rounding (< 0.05 % of the pixels), mIoU within 1e-3."""
    import torch_oracle as TO
    import metrics_oracle as MO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from pixelssl_amd.sseg.func import SSEGFunc
    sizes = [(65, 97), (81, 65), (49, 49), (97, 129), (65, 97), (113, 81), (33, 161)]

    def state(seed):
        st = TO.condition_state(TO.init_deeplabv2_state(seed=seed), 0.1)
        g = torch.Generator().manual_seed(seed + 7)
        for k in st:
            if k.endswith("running_mean"):
                st[k] = torch.randn(st[k].shape, generator=g) * 0.05
            elif k.endswith("running_var"):
                st[k] = torch.rand(st[k].shape, generator=g) + 0.5
        return st

    def oracle(st, loader):
        cm = np.zeros((21, 21), dtype=np.int64)
        losses = []
        with torch.no_grad():
            for (x,), (gt,) in loader:
                logits, prob, _, _ = TO.deeplabv2_forward(TO.clone_state(st), x, train=False)
                losses.append(TO.sseg_criterion(logits, gt).mean().item())
                cm += MO.confusion_matrix(prob.numpy(), gt.numpy(), 21)
        return cm, float(np.mean(losses))

    loader = _val_loader(sizes, 900)
    args = _args(labeled_batch_size=1)
    algo = P.ssl_algorithm.ssl_null.ssl_null(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                            SSEGFunc(args))
    st = state(41)
    core = algo.model.module.model
    core.load_state_dict(st)
    before = core.state_dict()["backbone.bn1.running_mean"].clone()
    algo.validate(loader, 0)
    torch.cuda.synchronize()
    want_cm, want_loss = oracle(st, loader)
    got_cm = algo.meters["task_confusion_matrix"].sum
    moved = np.abs(got_cm - want_cm).sum() / 2
    print("suponly validate: %d of %d pixels decided differently, mIoU %.6f vs oracle %.6f, loss %.6f vs %.6f"
          % (moved, want_cm.sum(), algo.meters["task_metric_mIoU"].val, MO.metrics(want_cm)["mIoU"],
             float(algo.meters["task_loss"].avg), want_loss))
    assert got_cm.sum() == want_cm.sum() and moved <= 5e-4 * want_cm.sum()
    assert abs(algo.meters["task_metric_mIoU"].val - MO.metrics(want_cm)["mIoU"]) < 1e-3
    assert abs(float(algo.meters["task_loss"].avg) - want_loss) < 1e-3 * want_loss
    assert torch.equal(core.state_dict()["backbone.bn1.running_mean"], before)           # eval mode: statistics untouched
    assert len(core._eval_plans) <= core.max_eval_plans and len(core._plans) == 0        # bounded, untuned inference plans
    assert all(pl.inference and not pl.pack_dgrad for pl in core._eval_plans.values())
    # the forward-only plans of all image sizes read ONE packed copy of the weights (a new size allocates activations only)
    assert len({pl.packed.data_ptr() for pl in core._eval_plans.values()}) == 1
    # training afterwards still works and uses a training plan
    algo.model.train()
    x, gt = TO.synthetic_batch(2, 65, 2, seed=5, block=16)
    loss, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
    assert torch.isfinite(loss) and len(core._plans) == 1
    # ... and the next validation sees the UPDATED weights through the shared copy (sizes whose plans are still cached included)
    st2 = OrderedDict((k, v.detach().cpu().clone()) for k, v in core.state_dict().items())
    short2 = _Loader(loader[-3:])
    algo.validate(short2, 1)
    torch.cuda.synchronize()
    want2, _ = oracle(st2, short2)
    got2 = algo.meters["task_confusion_matrix"].sum
    assert got2.sum() == want2.sum() and np.abs(got2 - want2).sum() / 2 <= 5e-4 * want2.sum()

    # ---- Mean Teacher: student and teacher metrics + the validation consistency loss
    args = _args(labeled_batch_size=1, unlabeled_batch_size=1, ignore_unlabeled=False)
    mt = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                      {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                      SSEGFunc(args))
    s_st, t_st = state(51), state(52)
    mt.s_model.module.model.load_state_dict(s_st)
    mt.t_model.module.model.load_state_dict(t_st)
    short = _Loader(loader[:4])
    mt.validate(short, 0)
    torch.cuda.synchronize()
    for tag, stt in (("student", s_st), ("teacher", t_st)):
        want_cm, _ = oracle(stt, short)
        got_cm = mt.meters[tag + "_confusion_matrix"].sum
        assert np.abs(got_cm - want_cm).sum() / 2 <= 5e-4 * want_cm.sum() + 2
        assert abs(mt.meters[tag + "_metric_mIoU"].val - MO.metrics(want_cm)["mIoU"]) < 1e-3
    with torch.no_grad():
        cons = [torch.nn.functional.mse_loss(TO.deeplabv2_forward(TO.clone_state(s_st), x, train=False)[0],
                                             TO.deeplabv2_forward(TO.clone_state(t_st), x, train=False)[0]).item()
                for (x,), _ in short]
    assert abs(float(mt.meters["cons_loss"].avg) - float(np.mean(cons))) < 1e-3 * float(np.mean(cons))
