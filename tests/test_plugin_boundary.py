"""Plugins drop in unchanged: options of the reference the mirror used to reject, and task models that
are NOT engine networks.

not gpu: FusedSGD / FusedAdam on foreign (plain torch) parameters = torch.optim.SGD / Adam for every option the
reference's factories pass through (pixelssl/nn/optimizer.py:57-122: momentum, dampening, nesterov, weight decay),
incl. the state_dict round trip with torch's optimizers.
gpu: the same options on the fused kernels (flat buffers of an engine model); GaussianNoiseLayer vs a torch
restatement of pixelssl/nn/module/gaussian_noise.py:18-40; an ssl_mt iteration on a torch TaskModel that is not a
SegNetCore (EMA, optimizer, criterion, consistency loss all through the plugin API) vs a plain-torch restatement."""
import argparse
import copy
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


SGD_CASES = [dict(momentum=0.9, weight_decay=5e-4), dict(momentum=0.9, dampening=0.3, weight_decay=1e-3),
             dict(momentum=0.8, nesterov=True, weight_decay=0.0), dict(momentum=0.0, weight_decay=1e-2)]
ADAM_CASES = [dict(betas=(0.9, 0.99), weight_decay=0.0), dict(betas=(0.9, 0.999), weight_decay=1e-2)]


@pytest.mark.parametrize("kw", SGD_CASES)
def test_fused_sgd_on_foreign_parameters_is_torch_sgd(kw):
    from pixelssl_amd.nn.optimizer import FusedSGD
    g = torch.Generator().manual_seed(1)
    a = [nn.Parameter(torch.randn(7, 5, generator=g)), nn.Parameter(torch.randn(11, generator=g))]
    b = [nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = FusedSGD(a, lr=0.05, **kw), torch.optim.SGD(b, lr=0.05, **kw)
    for it in range(4):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        if it == 1:            # checkpoint round trip through torch's format, both ways
            oa2 = FusedSGD(a, lr=0.05, **kw)
            oa2.load_state_dict(copy.deepcopy(ob.state_dict()))
            ob.load_state_dict(copy.deepcopy(oa.state_dict()))
            oa = oa2
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-7)
    oa.zero_grad()
    assert all(p.grad is None or not p.grad.any() for p in a)


@pytest.mark.parametrize("kw", ADAM_CASES)
def test_fused_adam_on_foreign_parameters_is_torch_adam(kw):
    from pixelssl_amd.nn.optimizer import FusedAdam
    g = torch.Generator().manual_seed(2)
    a = [nn.Parameter(torch.randn(6, 4, generator=g))]
    b = [nn.Parameter(a[0].detach().clone())]
    oa, ob = FusedAdam(a, lr=1e-2, eps=1e-8, **kw), torch.optim.Adam(b, lr=1e-2, eps=1e-8, **kw)
    for _ in range(5):
        gr = torch.randn(a[0].shape, generator=g)
        a[0].grad, b[0].grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-7)


def test_sgd_factory_accepts_the_reference_options():
    from pixelssl_amd.nn import optimizer as popt
    args = argparse.Namespace(lr=0.1, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=True)
    opt = popt.sgd(args)([{"params": [nn.Parameter(torch.zeros(3))], "lr": 0.1}])
    assert opt.param_groups[0]["nesterov"] is True
    with pytest.raises(ValueError):            # torch.optim.SGD's own argument check
        popt.sgd(argparse.Namespace(lr=0.1, momentum=0.0, dampening=0.0, weight_decay=0, nesterov=True))([nn.Parameter(torch.zeros(3))])
    args = argparse.Namespace(lr=1e-3, beta1=-1, beta2=-1, eps=-1, weight_decay=1e-2)
    assert popt.adam(args)([nn.Parameter(torch.zeros(3))]).param_groups[0]["weight_decay"] == 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("kw", SGD_CASES)
def test_fused_sgd_kernels_on_an_engine_model(kw):
    """the flat-buffer kernels (pxl_sgd_step / pxl_sgd_step_general) against torch.optim.SGD on the same parameters"""
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd.nn.optimizer import FusedSGD
    core = DeepLabV2Core(backbone=(1, 1, 1, 1), device=DEV, engine_dtype=torch.float32)
    params = list(core.parameters())
    ref = [nn.Parameter(p.detach().clone()) for p in params]
    oa, ob = FusedSGD(params, lr=0.05, **kw), torch.optim.SGD(ref, lr=0.05, **kw)
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        for p, r in zip(params, ref):
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad.copy_(gr)
            r.grad = gr.clone()
        oa.step()
        ob.step()
    worst = max(rel(p.detach(), r.detach()) for p, r in zip(params, ref))
    assert worst < 2e-6, worst


@pytest.mark.gpu
@pytest.mark.parametrize("kw", ADAM_CASES)
def test_fused_adam_kernel_with_weight_decay(kw):
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd.nn.optimizer import FusedAdam
    core = DeepLabV2Core(backbone=(1, 1, 1, 1), device=DEV, engine_dtype=torch.float32)
    params = list(core.parameters())
    ref = [nn.Parameter(p.detach().clone()) for p in params]
    oa, ob = FusedAdam(params, lr=1e-3, eps=1e-8, **kw), torch.optim.Adam(ref, lr=1e-3, eps=1e-8, **kw)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        for p, r in zip(params, ref):
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad.copy_(gr)
            r.grad = gr.clone()
        oa.step()
        ob.step()
    worst = max(rel(p.detach(), r.detach()) for p, r in zip(params, ref))
    assert worst < 1e-5, worst


def _ref_gaussian_noise(inp, noise):
    """pixelssl/nn/module/gaussian_noise.py:27-40 on a copy, the noise tensor given"""
    inp = inp.clone()
    imax = inp.max(dim=3, keepdim=True)[0].max(dim=2, keepdim=True)[0].max(dim=1, keepdim=True)[0]
    imin = inp.min(dim=3, keepdim=True)[0].min(dim=2, keepdim=True)[0].min(dim=1, keepdim=True)[0]
    inp.sub_(imin).div_(imax - imin + 1e-9)
    inp.add_(noise)
    ub = (inp > 1.0).float()
    lb = (inp < 0.0).float()
    inp.mul_(1 - ub).add_(ub)
    inp.mul_(1 - lb)
    inp.mul_(imax - imin + 1e-9).add_(imin)
    return inp


@pytest.mark.gpu
def test_gaussian_noise_layer_vs_reference_arithmetic():
    import random
    from pixelssl_amd.nn.module import GaussianNoiseLayer
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 65, 47, generator=g) * 40 + 100
    unit = torch.randn(x.shape, generator=g)
    layer = GaussianNoiseLayer(0.15)
    random.seed(9)
    layer.inject_noise(unit)
    xd = x.to(DEV)
    out = layer(xd)
    assert out.data_ptr() == xd.data_ptr(), "in place, like the reference"
    random.seed(9)
    sigma = random.uniform(0, 0.15)
    assert abs(layer.last_sigma - sigma) < 1e-12, "one python draw per call (gaussian_noise.py:25)"
    ref = _ref_gaussian_noise(x, unit * sigma)
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-4), (out.cpu() - ref).abs().max()
    assert (out.cpu() != x).float().mean() > 0.9 and GaussianNoiseLayer(None)(xd) is xd
    # without injection: torch's device generator supplies the deviates; the clip keeps every sample inside its range
    y = torch.rand(2, 3, 33, 33, device=DEV) * 5 - 1
    lo, hi = y.amin(dim=(1, 2, 3)), y.amax(dim=(1, 2, 3))
    z = GaussianNoiseLayer(0.5)(y.clone())
    assert (z.amin(dim=(1, 2, 3)) >= lo - 1e-4).all() and (z.amax(dim=(1, 2, 3)) <= hi + 1e-4).all() and (z != y).any()


class _TorchSegModel(nn.Module):
    """a small plain-torch segmentation TaskModel with the reference's plugin surface (task_template/model.py)"""

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        g = torch.Generator().manual_seed(17)
        self.model = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(),
                                   nn.Conv2d(16, 21, 1))
        with torch.no_grad():
            for p in self.model.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        self.param_groups = [{"params": list(self.model[:2].parameters()), "lr": args.lr},
                             {"params": list(self.model[2:].parameters()), "lr": args.lr * 10}]

    def forward(self, inp):
        pred = self.model(inp[0])
        return {"pred": (pred,), "activated_pred": (F.softmax(pred, dim=1),)}, {}


@pytest.mark.gpu
def test_ssl_mt_trains_a_task_model_that_is_not_an_engine_network():
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    a = argparse.Namespace(lr=0.05, momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1,
                           epochs=1, iters_per_epoch=8, ignore_index=255, labeled_batch_size=2, unlabeled_batch_size=2,
                           batch_size=4, ignore_unlabeled=False, is_epoch_lrer=False, log_freq=1000, task="sseg", gpus=1,
                           gaussian_noise_std=None, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3,
                           ema_decay=0.99, num_classes=21)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(a, {"model": lambda args: _TorchSegModel(args).to(DEV)}, {"model": popt.sgd(a)},
                                        {"model": plr.polynomiallr(a)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.train()
    algo.t_model.train()
    # plain-torch restatement of the same iteration (ssl_mt.py:131-220) on copies
    s_ref, t_ref = _TorchSegModel(a).to(DEV), _TorchSegModel(a).to(DEV)
    for p in t_ref.parameters():
        p.detach_()
    opt = torch.optim.SGD(s_ref.param_groups, lr=a.lr, momentum=0.9, weight_decay=5e-4)
    max_iters = a.epochs * a.iters_per_epoch
    for it in range(3):
        x, gt = TO.synthetic_batch(4, 65, 2, seed=70 + it, block=16)
        x, gt = x.to(DEV), gt.to(DEV)
        out, _, _ = algo.train_step((x,), (gt,), it, 6)
        ramp = P.nn.func.sigmoid_rampup(it, 6)
        opt.zero_grad()
        s_pred = s_ref.model(x)
        with torch.no_grad():
            t_pred = t_ref.model(x)
        lab = gt[:2, 0].long()
        ce = lambda z: (F.cross_entropy(z[:2], lab, ignore_index=255, reduction="none").sum((1, 2)) / (65 * 65)).mean()
        task, cons = ce(s_pred), ramp * F.mse_loss(s_pred[2:], t_pred[2:])
        (task + cons).backward()
        for g, base in zip(opt.param_groups, (a.lr, a.lr * 10)):
            # polynomial decay, power 0.9; the scheduler base class steps once in its constructor, so iteration `it`
            # runs at cur_iter = it + 1 (nn/lrer.py: PolynomialLR)
            g["lr"] = base * (1 - (it + 1) / max_iters) ** 0.9
        opt.step()
        alpha = min(1 - 1 / (it + 1), a.ema_decay)
        with torch.no_grad():
            for tp, sp in zip(t_ref.parameters(), s_ref.parameters()):
                tp.mul_(alpha).add_(sp, alpha=1 - alpha)
        got = {k: v.item() for k, v in out.items()}
        print("foreign-model mt iter %d:" % it, got, task.item(), cons.item())
        assert abs(got["s_task_loss"] - task.item()) < 2e-5 * abs(task.item())
        assert abs(got["cons_loss"] - cons.item()) <= 2e-4 * abs(cons.item()) + 1e-9
    for (k, v), r in zip(algo.s_model.module.state_dict().items(), s_ref.state_dict().values()):
        if v.is_floating_point():
            assert rel(v, r) < 1e-4, k
    for (k, v), r in zip(algo.t_model.module.state_dict().items(), t_ref.state_dict().values()):
        if v.is_floating_point():
            assert rel(v, r) < 1e-4, k
