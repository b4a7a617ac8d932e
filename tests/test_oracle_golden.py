"""CPU: the oracle (oracle/torch_oracle.py) against the golden fixtures that
oracle/make_golden.py produced by running the REAL reference."""
import os
import sys

import pytest
import torch

import torch_oracle as TO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert (a - b).abs().max().item() <= atol + rtol * b.abs().max().item()


def _check_probes(sd, probes, rtol=1e-5):
    for k, ref in probes.items():
        v = sd[k].detach().float().reshape(-1)
        _close(v[:64], ref["head"], rtol)
        assert abs(float(v.double().sum()) - ref["sum"]) <= 1e-6 + rtol * max(abs(ref["abssum"]), 1e-12)


def test_forward_matches_reference():
    fx = _load("deeplabv2_forward_65.pt")
    sd = TO.init_deeplabv2_state(seed=fx["weight_seed"])
    x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=fx["data_seed"],
                               block=fx["block"])
    leaves = TO._param_leaves(sd)
    run = TO._with_leaves(sd, leaves)
    logits, prob, latent, low = TO.deeplabv2_forward(run, x, train=True)
    _close(logits, fx["logits"])
    assert torch.equal(logits.argmax(1).to(torch.uint8), fx["argmax"])   # bit-exact indices
    _close(low, fx["low"])
    _close(latent.detach().reshape(-1)[:256], fx["latent_head"])
    ps = TO.sseg_criterion(logits, gt)
    _close(ps, fx["per_sample"])
    ps.mean().backward()
    _close(leaves["backbone.conv1.weight"].grad, fx["grad_conv1"], rtol=1e-4)
    _close(leaves["classifier.conv2d_list.1.weight"].grad.reshape(-1)[:512],
           fx["grad_aspp1_head"], rtol=1e-4)
    _close(leaves["backbone.layer3.5.conv2.weight"].grad.reshape(-1)[:512],
           fx["grad_l3_head"], rtol=1e-4)
    _check_probes(run, fx["probes"])


def test_softmax_sums_to_one_and_criterion_denominator():
    torch.manual_seed(3)
    logits = torch.randn(2, 21, 9, 9)
    gt = torch.randint(0, 21, (2, 1, 9, 9)).float()
    gt[0, 0, :4] = 255.0
    ps = TO.sseg_criterion(logits, gt)
    # ignored pixels: 0 in the numerator, still counted in the denominator (criterion.py:37-38)
    lse = torch.logsumexp(logits, 1)
    pick = logits.gather(1, gt.long().clamp(max=20)).squeeze(1)
    manual = ((lse - pick) * (gt.squeeze(1) != 255)).sum(dim=(1, 2)) / 81.0
    _close(ps, manual)


def test_suponly_two_iterations_match_reference():
    fx = _load("suponly_65.pt")
    tr = TO.OracleTrainer(TO.init_deeplabv2_state(seed=fx["weight_seed"]),
                          dict(max_iters=fx["max_iters"]))
    losses = []
    for s in fx["data_seeds"]:
        x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=s, block=fx["block"])
        losses.append(tr.suponly_step(x, gt)["task_loss"])
    _close(sum(losses) / len(losses), fx["mean_task_loss"])
    _check_probes(tr.sd, fx["probes"])


def test_mt_two_iterations_match_reference():
    fx = _load("mt_65.pt")
    tr = TO.OracleTrainer(TO.init_deeplabv2_state(seed=fx["weight_seed"]),
                          dict(max_iters=fx["max_iters"], cons_scale=1.0,
                               cons_rampup_iters=fx["rampup_iters"], cons_for_labeled=False,
                               ema_decay=0.99),
                          teacher_state=TO.init_deeplabv2_state(seed=fx["weight_seed"] + 1))
    outs = []
    for s in fx["data_seeds"]:
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s,
                                   block=fx["block"])
        outs.append(tr.mt_step(x, gt, fx["lbs"]))
    for k, v in fx["meters"].items():
        _close(sum(o[k] for o in outs) / len(outs), v)
    _check_probes(tr.sd, fx["student_probes"])
    _check_probes(tr.t_sd, fx["teacher_probes"])


def test_schedules():
    assert TO.sigmoid_rampup(0, 0) == 1.0
    assert abs(TO.sigmoid_rampup(0, 10) - 0.006737946999085467) < 1e-12
    assert TO.sigmoid_rampup(20, 10) == 1.0
    assert abs(TO.poly_lr(1.0, 1, 4, 0.9) - 0.7718895067235705) < 1e-12


def test_oracle_multi_device_syncbn_formula_is_the_references():
    """torch_oracle._bn with SYNC_BN_MULTI_DEVICE restates _SynchronizedBatchNorm's DataParallel path
    (sync_batchnorm/batchnorm.py:56-78,113-125).  Pinned against the reference class itself when the reference tree is
    present (its master / slave message passing driven by hand for two 'replicas' of one batch), and always against the
    defining property: the two formulas differ exactly by clamp(var, eps) vs var + eps."""
    import torch_oracle as TO
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 6, 5, 7, generator=g) * torch.tensor([1e-3, 0.1, 1, 3, 1e-4, 2]).view(1, 6, 1, 1) + 0.2
    sd = {"b.weight": torch.rand(6, generator=g) + 0.5, "b.bias": torch.randn(6, generator=g),
          "b.running_mean": torch.zeros(6), "b.running_var": torch.ones(6)}
    sd2 = {k: v.clone() for k, v in sd.items()}
    TO.SYNC_BN_MULTI_DEVICE = True
    try:
        y_multi = TO._bn(sd, "b", x, True)
    finally:
        TO.SYNC_BN_MULTI_DEVICE = False
    y_single = TO._bn(sd2, "b", x, True)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    want = (x - mean.view(1, 6, 1, 1)) * (var.clamp(1e-5) ** -0.5 * sd["b.weight"]).view(1, 6, 1, 1) + sd["b.bias"].view(1, 6, 1, 1)
    assert torch.allclose(y_multi, want, rtol=1e-4, atol=1e-5)
    assert torch.allclose(sd["b.running_var"], sd2["b.running_var"], rtol=1e-5) and torch.allclose(sd["b.running_mean"], sd2["b.running_mean"], rtol=1e-5, atol=1e-7)
    # low-variance channels (1e-3, 1e-4 scale: var << eps) are where the two formulas part: x 1/sqrt(eps) vs x 1/sqrt(var+eps)
    d = (y_multi - y_single).abs().amax(dim=(0, 2, 3))
    assert d[0] > 1e-3 and d[2] < 1e-4
    import ref_shim
    if not ref_shim.reference_available():
        return
    ref_shim.load_reference()
    from pixelssl.nn.module.third_party.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d, _ChildMessage
    bn = SynchronizedBatchNorm2d(6)
    with torch.no_grad():
        bn.weight.copy_(sd2["b.weight"]); bn.bias.copy_(sd2["b.bias"])
    # the master's reduction of two replicas' messages (each half of the batch), computed the way _data_parallel_master
    # does after ReduceAddCoalesced: sums added, then _compute_mean_std
    halves = [x[:2], x[2:]]
    msgs = [(h.reshape(2, 6, -1).sum(dim=(0, 2)), (h.reshape(2, 6, -1) ** 2).sum(dim=(0, 2)), 2 * 35) for h in halves]
    sum_, ssum, size = msgs[0][0] + msgs[1][0], msgs[0][1] + msgs[1][1], msgs[0][2] + msgs[1][2]
    mean_r, inv_std_r = bn._compute_mean_std(sum_, ssum, size)
    out_r = torch.cat([(h.reshape(2, 6, -1) - mean_r.view(1, 6, 1)) * (inv_std_r * bn.weight).view(1, 6, 1) + bn.bias.view(1, 6, 1)
                       for h in halves]).view(x.shape)
    assert torch.allclose(y_multi, out_r.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd["b.running_var"], bn.running_var, rtol=1e-5) and torch.allclose(sd["b.running_mean"], bn.running_mean, rtol=1e-5, atol=1e-7)
