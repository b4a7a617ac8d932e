"""CPU: the oracle (oracle/torch_oracle.py) against the golden fixtures that
oracle/make_golden.py produced by running the REAL reference."""
import os

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
