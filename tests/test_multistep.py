"""Multi-step parity on CONDITIONED initial weights (oracle/make_golden_conditioned.py, torch_oracle.condition_state).

Six iterations of the reference's own `_train` loops (SupOnly / MT / AdvSSL / CutMix / GCT / CCT, DeepLab-v2 and PSPNet,
full ResNet-101, train-mode BN, shipped hyper-parameters) at 129 x 129 -- and, for every BASELINE.json workload (MT: four
iterations at 4 + 4; AdvSSL / CutMix / GCT / CCT with G-Cutout: two iterations at 2 + 2 resp. 2 + 4) at the BASELINE crop size
513 x 513 -- from weights whose bottleneck-output BN gammas are scaled by 0.1.  On these weights the reference arithmetic reproduces itself (fp32 vs fp64 < 1e-6 in every logged
loss), so the bars below bite:

  * every logged loss of every iteration within LOSS_TOL of the reference's (fp32 engine; the stated band for bf16),
  * every probed weight after the last iteration within WEIGHT_FRAC of the size of its own six-step update, as L2
    norms over a strided 4096-element sample of the tensor -- an engine that skipped an update, used a wrong lr group
    or dropped a loss term scores 1.0 (test_a_noop_optimizer_fails_the_weight_bar).
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda"

LOSS_TOL = {"fp32": 1e-3, "bf16": 1e-2}
# PXL_DETERMINISTIC=1 (set for the whole pytest run: `PXL_DETERMINISTIC=1 python -m pytest tests/test_multistep.py -m gpu`) makes every
# figure of the SupOnly / PSPNet / MT / AdvSSL / CutMix cases the same number in every run (profiles/r05_d_det_*: two consecutive runs,
# identical prints), so the bf16 bars that the default mode keeps outside its run-to-run spread can sit at 1.25 x the value itself:
# DeepLab-v2 weights 0.363 -> 0.45 (default 0.6), PSPNet 0.541 -> 0.65 (default 0.8), CutMix's consistency term 3.1 % -> 10 % (30 %)
DET = os.environ.get("PXL_DETERMINISTIC") == "1"
WEIGHT_FRAC = {"fp32": 0.05, "bf16": 0.45 if DET else 0.6}
PSP_BF16_FRAC = 0.65 if DET else 0.8
EPS32 = 1.1920929e-07
# bf16 engine, CCT (seven decoders back-propagated through ~100 bf16 layers): bars on the direction of the six-step update
# (measured on the MI355X: worst probed tensor 0.67-0.75 -- conv1 and layer4.2.conv2, whose gradient arrives through
# the bf16 latent of the unlabeled pass -- median 0.996-0.999, norm ratios 0.99-1.30; a no-op scores ratio 0)
# The bf16 engine is not bitwise reproducible run to run (fp32 atomics order the BatchNorm statistics), and on these two
# tensors the spread is wide: over ten runs of the G-Cutout fixture (tools/rep_one.sh gcutout) the worst cosine ranged over
# 0.65 ... 0.74 and the largest norm ratio over 1.03 ... 1.35, and one run in ten left the first version of these bars
# (0.55 / 1.45).  They now sit where that spread cannot reach; a no-op (ratio 0) or a wrong-signed update (cosine < 0)
# still fails.
# Round 6: PXL_DETERMINISTIC=1 now covers GCT and CCT as well (ordered IBNorm sums, flaw-map loss sums, I-VAT's norm: two consecutive
# runs print identical figures, profiles/r06_det_run{1,2}_figures.txt) -- under the mode the bars sit at 1.25 x the deterministic
# values (worst cosine 0.694 / 0.698 on conv1, largest norm ratio 1.30, median cosine 0.997) instead of below the default spread
CCT_BF16_MIN_COS = 0.55 if DET else 0.45
CCT_BF16_MEDIAN_COS = 0.98 if DET else 0.95
CCT_BF16_RATIO = (0.8, 1.65) if DET else (0.8, 1.8)


def _fx(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def subsample(v, n=4096):
    v = v.detach().float().cpu().contiguous().reshape(-1)
    return v[::max(1, v.numel() // n)][:n]


def _args(fx, dtype, **kw):
    lbs = fx.get("lbs", fx.get("batch"))
    ubs = fx.get("ubs", 0)
    a = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4,
                           momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1,
                           epochs=1, iters_per_epoch=fx["max_iters"], ignore_index=255, labeled_batch_size=lbs,
                           unlabeled_batch_size=ubs, batch_size=lbs + ubs, ignore_unlabeled=ubs == 0,
                           is_epoch_lrer=False, log_freq=1000, task="sseg", engine_dtype=dtype, gpus=1,
                           im_size=fx["size"], gaussian_noise_std=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _check_losses(tag, i, got, ref, dtype, loose=(), very_loose=()):
    for k, r in ref.items():
        tol = LOSS_TOL[dtype] * (30.0 if any(s in k for s in very_loose) else 10.0 if any(s in k for s in loose) else 1.0)
        assert abs(got[k] - r) <= tol * abs(r) + 1e-7, "%s iteration %d %s: engine %.7g reference %.7g" % (tag, i, k, got[k], r)


def _check_weights(tag, sd, updates, dtype, frac=None, skip=()):
    """||engine - reference|| <= frac * ||update|| + 4 ulp * ||reference|| per probed tensor (L2 over the stored sample).
    The ulp term is the fp32 rounding floor of the stored values themselves: a BN gamma of 1.0 moves by ~1e-6 in six
    steps, 10 ulps -- the reference's `mul_().add_()` and a fused multiply-add round such an EMA differently."""
    frac = WEIGHT_FRAC[dtype] if frac is None else frac
    rows = []
    for k, u in updates.items():
        if u["update_l2"] < 1e-12 or any(s in k for s in skip):
            continue
        got = subsample(sd[k])
        err = (got.double() - u["sample"].double()).norm().item()
        allowed = frac * u["update_l2"] + 4 * EPS32 * u["sample"].double().norm().item()
        rows.append((err / allowed, err / u["update_l2"], k))
    rows.sort(reverse=True)
    print("%s: worst |engine - reference| / |update| = %s, bar %.2f (+ 4 ulp)"
          % (tag, ", ".join("%.3e (%s)" % (r[1], r[2]) for r in rows[:3]), frac))
    assert rows and rows[0][0] <= 1.0, (tag, rows[:3])


def _check_update_direction(tag, sd, init_sd, updates, min_cos, ratio=(0.5, 1.5), median_cos=None, skip=()):
    """A bar that an engine which never updated (or updated the wrong way) fails in EVERY dtype, whatever the size of the
    rounding noise: per probed tensor, the engine's own update (final - initial, same strided sample as the fixture's)
    must point the way the reference's does -- cosine >= min_cos -- and have a comparable size -- norm ratio inside
    `ratio`.  A no-op scores ratio 0; a sign error scores cosine -1; a missing loss term or a wrong lr group moves the
    ratio.  `median_cos`: additional bar on the median cosine over the probed tensors."""
    rows = []
    for k, u in updates.items():
        if u["update_l2"] < 1e-12 or any(s in k for s in skip):
            continue
        init = subsample(init_sd[k]).double()
        du_e = subsample(sd[k]).double() - init
        du_r = u["sample"].double() - init
        # the ulp floor of the stored values: tensors whose reference update is within a few ulps of their magnitude
        # (BN gammas near 1.0) carry no direction information
        if du_r.norm().item() < 16 * EPS32 * u["sample"].double().norm().item():
            continue
        cos = (du_e @ du_r).item() / (du_e.norm().item() * du_r.norm().item() + 1e-300)
        rows.append((cos, du_e.norm().item() / du_r.norm().item(), k))
    assert rows, tag
    rows.sort()
    cosines = sorted(r[0] for r in rows)
    med = cosines[len(cosines) // 2]
    print("%s: update direction vs reference: worst cosine %s; median %.3f; norm ratio range [%.2f, %.2f]"
          % (tag, ", ".join("%.3f (%s)" % (r[0], r[2]) for r in rows[:3]), med, min(r[1] for r in rows), max(r[1] for r in rows)))
    for cos, rat, k in rows:
        assert cos >= min_cos, "%s %s: update cosine %.3f < %.2f" % (tag, k, cos, min_cos)
        assert ratio[0] <= rat <= ratio[1], "%s %s: update norm ratio %.3f outside %s" % (tag, k, rat, ratio)
    if median_cos is not None:
        assert med >= median_cos, "%s: median update cosine %.3f < %.2f" % (tag, med, median_cos)


def _deeplab_state(seed, gamma3):
    import torch_oracle as TO
    return TO.condition_state(TO.init_deeplabv2_state(seed=seed), gamma3)


def test_a_noop_optimizer_fails_the_weight_bar():
    """CPU: the initial weights (= an engine that never updated) sit at exactly 1.0 x the update, 20 x the fp32 bar."""
    for name, key, seed_off in (("suponly_cond_129.pt", "updates", 0), ("mt_cond_129.pt", "student_updates", 0)):
        fx = _fx(name)
        init = _deeplab_state(fx["weight_seed"] + seed_off, fx["gamma3"])
        for k, u in fx[key].items():
            if u["update_l2"] < 1e-12:
                continue
            r = (subsample(init[k]).double() - u["sample"].double()).norm().item() / u["update_l2"]
            assert abs(r - 1.0) < 1e-6 and r > 10 * WEIGHT_FRAC["fp32"], (name, k, r)
        # ... and _check_weights rejects it (including the ulp floor of the BN gammas)
        with pytest.raises(AssertionError):
            _check_weights("noop", init, fx[key], "fp32")
        with pytest.raises(AssertionError):          # the direction criterion (used for the bf16 bars) rejects it too
            _check_update_direction("noop", init, init, fx[key], min_cos=0.0)
    assert len(_fx("suponly_cond_129.pt")["per_iter"]) >= 5


def test_oracle_replays_the_conditioned_fixtures():
    """CPU: the oracle reproduces the reference's six SupOnly iterations bit for bit (fixture pinned on any box)."""
    import torch_oracle as TO
    fx = _fx("suponly_cond_129.pt")
    tr = TO.OracleTrainer(_deeplab_state(fx["weight_seed"], fx["gamma3"]), dict(max_iters=fx["max_iters"]))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    for i, s in enumerate(fx["data_seeds"][:3]):
        x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=s, block=fx["block"])
        loss = tr.suponly_step(x, gt)["task_loss"]
        assert abs(loss - fx["per_iter"][i]["task_loss"]) < 2e-5 * abs(loss), (i, loss)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_suponly_six_iterations(dtype):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx("suponly_cond_129.pt")
    args = _args(fx, dtype)
    algo = P.ssl_algorithm.ssl_null.ssl_null(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    core = algo.model.module.model
    core.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=s, block=fx["block"])
        loss, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
        print("suponly %s iter %d: %.6f (reference %.6f)" % (dtype, i, loss.item(), fx["per_iter"][i]["task_loss"]))
        _check_losses("suponly", i, {"task_loss": loss.item()}, fx["per_iter"][i], dtype)
    _check_weights("suponly " + dtype, core.state_dict(), fx["updates"], dtype)
    _check_update_direction("suponly " + dtype, core.state_dict(), _deeplab_state(fx["weight_seed"], fx["gamma3"]), fx["updates"],
                            min_cos=0.99 if dtype == "fp32" else 0.85, ratio=(0.97, 1.03) if dtype == "fp32" else (0.9, 1.1))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pspnet_suponly_six_iterations(dtype):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx("pspnet_suponly_cond_129.pt")
    args = _args(fx, dtype, models={"model": "pspnet"})
    algo = P.ssl_algorithm.ssl_null.ssl_null(args, {"model": P.sseg.model.pspnet()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    core = algo.model.module.model
    core.load_state_dict(TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"]))
    algo.model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=s, block=fx["block"])
        loss, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
        print("pspnet %s iter %d: %.6f (reference %.6f)" % (dtype, i, loss.item(), fx["per_iter"][i]["task_loss"]))
        _check_losses("pspnet suponly", i, {"task_loss": loss.item()}, fx["per_iter"][i], dtype)
    # bf16: PSPNet's six-step distance is larger than DeepLab's (measured 0.53 of the update on conv1 against 0.33; run-to-run
    # spread of a few hundredths): a 0.80 bar, plus the direction criterion that a no-op fails
    _check_weights("pspnet suponly " + dtype, core.state_dict(), fx["updates"], dtype, frac=None if dtype == "fp32" else PSP_BF16_FRAC)
    _check_update_direction("pspnet suponly " + dtype, core.state_dict(),
                            TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"]), fx["updates"],
                            min_cos=0.99 if dtype == "fp32" else 0.7, ratio=(0.97, 1.03) if dtype == "fp32" else (0.8, 1.25))
    # (bf16, three runs on the MI355X: distance 0.50 ... 0.54, worst cosine 0.85 ... 0.88, norm ratios 0.91 ... 1.06)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mt_six_iterations(dtype):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx("mt_cond_129.pt")
    args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    algo.s_model.train()
    algo.t_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        got = {k: v.item() for k, v in out.items()}
        print("mt %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i])
        ref = dict(fx["ref_per_iter"][i])
        if i == 1:          # EMA at step 0 copies the student (alpha = 0): iteration 1's consistency loss is exactly 0 in
            assert got["cons_loss"] <= (1e-12 if dtype == "fp32" else 1e-5)      # the reference (student and teacher
            ref.pop("cons_loss")                                                # run on different streams / statistics replicas)
        _check_losses("mt", i, got, ref, dtype, loose=("cons",) if dtype == "bf16" else ())
    _check_weights("mt student " + dtype, algo.s_model.module.model.state_dict(), fx["student_updates"], dtype)
    _check_weights("mt teacher " + dtype, algo.t_model.module.model.state_dict(), fx["teacher_updates"], dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mt_at_the_baseline_configuration(dtype):
    """BASELINE.json configs[1] -- the workload bench.py times: MT, DeepLab-v2 / ResNet-101, 4 labeled + 4 unlabeled
    crops at 513 x 513, shipped hyper-parameters.  Fixture = FOUR iterations of the reference's own SSLMT._train
    (ssl_mt.py:124-224) on conditioned weights (oracle/make_golden_conditioned.py mt513).  fp32 engine: every logged loss
    within 1e-3, student and teacher weights within 0.05 x their own four-step update; bf16 engine (the benchmarked
    precision): losses within 1e-2, weights within 0.6 x the update AND the update pointing the reference's way."""
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx("mt_cond_513.pt")
    assert fx["size"] == 513 and fx["lbs"] == 4 and fx["ubs"] == 4 and len(fx["data_seeds"]) >= 4
    args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    s_init = _deeplab_state(fx["weight_seed"], fx["gamma3"])
    t_init = _deeplab_state(fx["weight_seed"] + 1, fx["gamma3"])
    algo.s_model.module.model.load_state_dict(s_init)
    algo.t_model.module.model.load_state_dict(t_init)
    algo.s_model.train()
    algo.t_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        got = {k: v.item() for k, v in out.items()}
        print("mt 513 %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i])
        ref = dict(fx["ref_per_iter"][i])
        if i == 1:          # (see test_mt_six_iterations: the reference's consistency loss is exactly 0 here)
            assert got["cons_loss"] <= (1e-12 if dtype == "fp32" else 1e-5)
            ref.pop("cons_loss")
        _check_losses("mt 513", i, got, ref, dtype, loose=("cons",) if dtype == "bf16" else ())
    s_sd, t_sd = algo.s_model.module.model.state_dict(), algo.t_model.module.model.state_dict()
    _check_weights("mt 513 student " + dtype, s_sd, fx["student_updates"], dtype)
    _check_weights("mt 513 teacher " + dtype, t_sd, fx["teacher_updates"], dtype)
    _check_update_direction("mt 513 student " + dtype, s_sd, s_init, fx["student_updates"],
                            min_cos=0.99 if dtype == "fp32" else 0.85, ratio=(0.97, 1.03) if dtype == "fp32" else (0.9, 1.1))


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["adv_cond_129.pt", "adv_cond_513.pt", "adv_cond_513_b8.pt"], ids=["129", "513", "513b8"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_advssl_six_iterations(dtype, fixture):
    import torch_oracle as TO
    import adv_oracle as AO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx(fixture)
    args = _args(fx, dtype, adv_for_labeled=True, labeled_adv_scale=0.01, unlabeled_adv_scale=0.001, discriminator_lr=1e-4,
                 discriminator_power=0.9, unlabeled_for_discriminator=True, discriminator_scale=1.0)
    algo = P.ssl_algorithm.ssl_adv.ssl_adv(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                          {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                          P.sseg.func.task_func()(args))
    algo.model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.d_model.module.load_state_dict(AO.init_fcd_state(21, seed=fx["d_seed"]))
    algo.model.train()
    algo.d_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
        got = {k: v.item() for k, v in out.items()}
        print("adv %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i])
        _check_losses("adv", i, got, fx["ref_per_iter"][i], dtype)
    _check_weights("adv task model " + dtype, algo.model.module.model.state_dict(), fx["updates"], dtype)
    # Adam's first steps are ~ lr * sign(g): an element whose gradient is rounding noise may step the other way, so
    # the discriminator is held to a looser fraction of its update
    dsd = OrderedDict((k, v) for k, v in algo.d_model.module.state_dict().items())
    _check_weights("adv discriminator " + dtype, dsd, fx["d_updates"], dtype, frac=0.2 if dtype == "fp32" else 0.6)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["cutmix_cond_129.pt", "cutmix_cond_513.pt"], ids=["129", "513"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cutmix_six_iterations(dtype, fixture):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx(fixture)
    args = _args(fx, dtype, cons_type="mse", cons_scale=fx["cons_scale"], cons_rampup_epochs=0,
                 cons_threshold=fx["cons_threshold"], ema_decay=0.99, mask_prop_range=(0.5, 0.5))
    algo = P.ssl_algorithm.ssl_cutmix.ssl_cutmix(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                                {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    algo.mask_generator.rng = np.random.RandomState(fx["np_seed"])
    algo.s_model.train()
    algo.t_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, 0)
        got = {k: v.item() for k, v in out.items() if k in fx["ref_per_iter"][i]}
        print("cutmix %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i])
        # the consistency loss is scaled by a COUNT of teacher maxima above the threshold: a pixel at the threshold flips
        # it by 1 / (B H W); it gets 10 x the loss tolerance.  bf16: by iteration 5 the term is 5.9e-5 (the task loss is 2.8) and
        # the engine's own run-to-run spread on it (fp32 atomics order the BN statistics differently every run) was measured
        # at 0.6 ... 11.2 % below the reference over seven runs on the MI355X (tools/rep_cutmix.sh): 30 x = a 30 % bar there
        if dtype == "bf16":
            if DET:
                _check_losses("cutmix", i, got, fx["ref_per_iter"][i], dtype, loose=("cons",))
            else:
                _check_losses("cutmix", i, got, fx["ref_per_iter"][i], dtype, very_loose=("cons",))
        else:
            _check_losses("cutmix", i, got, fx["ref_per_iter"][i], dtype, loose=("cons",))
    _check_weights("cutmix student " + dtype, algo.s_model.module.model.state_dict(), fx["student_updates"], dtype)
    _check_weights("cutmix teacher " + dtype, algo.t_model.module.model.state_dict(), fx["teacher_updates"], dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["gct_cond_129.pt", "gct_cond_513.pt", "gct_cond_513_b8.pt"], ids=["129", "513", "513b8"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gct_six_iterations(dtype, fixture):
    _gct_six_iterations(dtype, fixture)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gct_six_iterations_one_forward_per_task_model(dtype, monkeypatch):
    """PXL_GCT_REUSE_FORWARD=1 (opt-in): the step-0 no-grad pass and the step-1 pass of each task model are ONE pass whose
    running statistics take both updates (pxl_net_set_bn_repeat) -- same fixture, same bars (losses, weights, running statistics
    are part of the compared state)."""
    monkeypatch.setenv("PXL_GCT_REUSE_FORWARD", "1")
    _gct_six_iterations(dtype, "gct_cond_129.pt", expect_reuse=True)


def _gct_six_iterations(dtype, fixture, expect_reuse=False):
    import torch_oracle as TO
    import gct_oracle as GO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx(fixture)
    args = _args(fx, dtype, ssl_mode="gct", fc_ssl_scale=1.0, dc_ssl_scale=100.0, dc_threshold=0.6, dc_rampup_epochs=3,
                 fd_lr=1e-4, fd_scale=10.0, mu=0.5, nu=1)
    algo = P.ssl_algorithm.ssl_gct.ssl_gct(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                          {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                          P.sseg.func.task_func()(args))
    algo.l_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.r_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    fd = GO.init_fd_state(24, seed=fx["fd_seed"])
    fd["classifier.weight"] = fd["classifier.weight"] * fx["fd_scale_classifier"]
    fd["classifier.bias"] = fd["classifier.bias"] * fx["fd_scale_classifier"]
    algo.fd_model.module.load_state_dict(fd)
    for m in (algo.l_model, algo.r_model, algo.fd_model):
        m.train()
    assert (algo._reusable_cores() is not None) == expect_reuse
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        got = {k: v.item() for k, v in out.items()}
        print("gct %s iter %d:" % (dtype, i), {k: round(v, 6) for k, v in got.items()}, "\n    ref", {k: round(v, 6) for k, v in fx["per_iter"][i].items()})
        # flaw-map losses pass a 0.6 threshold and an Adam-trained detector: the oracle itself is 3e-3 / 1e-2 from the
        # reference on them (make_golden_gct_train.py); they get 10 x the loss tolerance
        # (bf16: the consistency loss counts pixels whose handled flaw map is above the 0.6 threshold -> 30 %)
        _check_losses("gct", i, got, fx["per_iter"][i], dtype, loose=("fc", "dc", "fd"), very_loose=("dc",) if dtype == "bf16" else ())
    # the reference runs every task model twice per iteration in train mode (ssl_gct.py:196-200 + 403): num_batches_tracked
    # advances by 2 per iteration, whether the engine ran the two passes or one pass that stands for both
    for m in (algo.l_model, algo.r_model):
        nbt = m.module.model.state_dict()["backbone.bn1.num_batches_tracked"]
        assert int(nbt) == 2 * len(fx["data_seeds"]), "num_batches_tracked %d after %d iterations" % (int(nbt), len(fx["data_seeds"]))
    _check_weights("gct l " + dtype, algo.l_model.module.model.state_dict(), fx["l_updates"], dtype)
    _check_weights("gct r " + dtype, algo.r_model.module.model.state_dict(), fx["r_updates"], dtype)
    fsd = OrderedDict((k, v) for k, v in algo.fd_model.module.state_dict().items())
    # convolution biases in front of an IBNorm have an exactly-zero true gradient (the normalisation removes them): their
    # updates are Adam steps on rounding noise in the reference too, a random walk nobody reproduces -> not compared
    _check_weights("gct flaw detector " + dtype, fsd, fx["fd_updates"], dtype, frac=0.3 if dtype == "fp32" else 0.8,
                   skip=("conv1.bias", "conv2.bias", "conv2_1.bias", "conv3.bias", "conv3_1.bias", "conv4.bias", "conv4_1.bias"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cct_six_iterations(dtype):
    import torch_oracle as TO
    import cct_oracle as CO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from pixelssl_amd.sseg.func import SSEGFunc
    fx = _fx("cct_cond_129.pt")
    args = _args(fx, dtype, models={"model": "pspnet"}, cons_scale=30.0, cons_rampup_epochs=5, ad_lr_scale=10.0,
                 vat_dec_num=1, vat_dec_xi=1e-6, vat_dec_eps=2.0, drop_dec_num=1, drop_dec_rate=0.5, drop_dec_spatial=True,
                 cut_dec_num=0, cut_dec_erase=0.4, context_dec_num=1, object_dec_num=1, fd_dec_num=1, fn_dec_num=1,
                 fn_dec_uniform=0.3)
    algo = P.ssl_algorithm.ssl_cct.ssl_cct(args, {"model": P.sseg.model.pspnet()}, {"model": popt.sgd(args)},
                                          {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                          SSEGFunc(args))
    wrapped = algo.model.module
    wrapped.main_model.model.load_state_dict(TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"]))
    for m, s in zip(wrapped.auxiliary_decoders, fx["decoder_seeds"]):
        m.load_state_dict(CO.init_decoder_state(s, in_channels=fx["in_channels"]))
    algo.model.train()
    B = fx["lbs"] + fx["ubs"]
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(B, fx["size"], fx["lbs"], seed=s, block=fx["block"])
        for m, d in zip(wrapped.auxiliary_decoders, fx["draws"][i]):
            if d is not None:
                m.inject_draw(d)
        out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        got = {k: v.item() for k, v in out.items()}
        print("cct %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i])
        # I-VAT's adversarial direction is normalised rounding noise in the reference (xi = 1e-6 < fp32 resolution of
        # the latent): the consistency loss averages it with five reproducible decoders -> 2 % band
        _check_losses("cct", i, got, fx["ref_per_iter"][i], dtype, loose=("cons",))
    # psp.stages.0 = the 1-bin pyramid stage: its BN normalises over the 2 samples of a CCT sub-batch, the output is
    # +-1 whatever the conv computes, so its weight gradient is rounding noise in the reference too (bf16: not compared).
    main_sd = wrapped.main_model.model.state_dict()
    if dtype == "fp32":
        _check_weights("cct main fp32", main_sd, fx["main_updates"], dtype, frac=0.1)
    else:
        # bf16: a distance bar >= 1 would pass an engine that never updated (round 2's 1.25 did), so the bf16 statement
        # is made on the UPDATE itself: every probed tensor moved the way the reference moved it (cosine) by a
        # comparable amount (norm ratio) -- a no-op scores ratio 0 and fails.  Why CCT sits further out than the other
        # algorithms (0.68-0.9 x the update vs 0.35): tools/diag_bf16_grad.py, profiles/r03_bf16_grad_diag.txt.
        init = TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"])
        _check_update_direction("cct main bf16", main_sd, init, fx["main_updates"], min_cos=CCT_BF16_MIN_COS,
                                ratio=CCT_BF16_RATIO, median_cos=CCT_BF16_MEDIAN_COS, skip=("psp.stages.0.",))


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["cct_cut_cond_129.pt", "cct_cut_cond_513.pt", "cct_cut_cond_513_b8.pt"], ids=["129", "513", "513b8"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_cct_six_iterations_with_gcutout(dtype, fixture):
    """BASELINE.json config 5: K = 7 auxiliary decoders INCLUDING G-Cutout.  Fixture = six iterations (129 x 129) / two
    iterations at the BASELINE crop size (513 x 513: oracle/make_golden_conditioned.py cct513) of the reference's
    own SSLCCT._train with its CutOutDecoder (ssl_cct.py:597-650) running on a stand-in for cv2.findContours
    (oracle/cct_oracle.py: find_contours_stand_in; OpenCV is not installed); the fixture carries the boxes that call
    returned and the random.randint draws, `inject_draw` feeds them to the engine's decoder.  Pinned by this test:
    the erase-window arithmetic, the mask, its nearest resize, the masked latent, the decoder body and the whole
    training step with the cutout decoder in it.  NOT pinned: the contour search itself (csrc/contour.cpp)."""
    import torch_oracle as TO
    import cct_oracle as CO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from pixelssl_amd.sseg.func import SSEGFunc
    fx = _fx(fixture)
    assert fx["with_cut"] and [k for k, _ in fx["decoders"]] == ["vat", "drop", "cut", "context", "object", "fd", "fn"]
    args = _args(fx, dtype, models={"model": "pspnet"}, cons_scale=30.0, cons_rampup_epochs=5, ad_lr_scale=10.0,
                 vat_dec_num=1, vat_dec_xi=1e-6, vat_dec_eps=2.0, drop_dec_num=1, drop_dec_rate=0.5, drop_dec_spatial=True,
                 cut_dec_num=1, cut_dec_erase=0.4, context_dec_num=1, object_dec_num=1, fd_dec_num=1, fn_dec_num=1,
                 fn_dec_uniform=0.3)
    algo = P.ssl_algorithm.ssl_cct.ssl_cct(args, {"model": P.sseg.model.pspnet()}, {"model": popt.sgd(args)},
                                          {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                          SSEGFunc(args))
    wrapped = algo.model.module
    assert [type(m).__name__ for m in wrapped.auxiliary_decoders][2] == "CutOutDecoder"
    init = TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"])
    init["decoder.3.conv.bias"][0:4] += fx["bias0_shift"]      # background bias: ragged foreground blobs (make_golden_cct.py)
    wrapped.main_model.model.load_state_dict(init)
    for m, s in zip(wrapped.auxiliary_decoders, fx["decoder_seeds"]):
        m.load_state_dict(CO.init_decoder_state(s, in_channels=fx["in_channels"]))
    algo.model.train()
    B = fx["lbs"] + fx["ubs"]
    nboxes = 0
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(B, fx["size"], fx["lbs"], seed=s, block=fx["block"])
        for m, d in zip(wrapped.auxiliary_decoders, fx["draws"][i]):
            if d is not None:
                m.inject_draw(d)
        out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        got = {k: v.item() for k, v in out.items()}
        cut = wrapped.auxiliary_decoders[2]
        assert cut.last_boxes == [[tuple(b) for b in bs] for bs in fx["draws"][i][2]["boxes"]]
        nboxes += sum(len(b) for b in cut.last_boxes)
        print("cct+cut %s iter %d:" % (dtype, i), got, fx["ref_per_iter"][i], "boxes", [len(b) for b in cut.last_boxes])
        _check_losses("cct+cut", i, got, fx["ref_per_iter"][i], dtype, loose=("cons",))
    assert nboxes > 0, "the fixture must exercise the erase windows"
    main_sd = wrapped.main_model.model.state_dict()
    if dtype == "fp32":
        _check_weights("cct+cut main fp32", main_sd, fx["main_updates"], dtype, frac=0.1)
    else:
        _check_update_direction("cct+cut main bf16", main_sd, init, fx["main_updates"], min_cos=CCT_BF16_MIN_COS,
                                ratio=CCT_BF16_RATIO, median_cos=CCT_BF16_MEDIAN_COS, skip=("psp.stages.0.",))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mt_teacher_is_the_ema_of_the_student(dtype):
    """The EMA path pinned on its own (BatchNorm gammas move by ulps per step, so the fixture's update-relative
    bar reaches them only through its 4-ulp floor): after every iteration the teacher's parameters must be
    alpha * teacher + (1 - alpha) * student with alpha = min(1 - 1 / (step + 1), ema_decay) (ssl_mt.py:359-363), evaluated here
    in fp64 from snapshots of the engine's own student parameters -- every element within 2 fp32 ulp of it (two products and a
    sum, each rounded once; the reference's `mul_().add_()` rounds the same three results)."""
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = _fx("mt_cond_129.pt")
    args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
    s_core.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    t_core.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    algo.s_model.train()
    algo.t_model.train()
    worst = 0.0
    for i, s in enumerate(fx["data_seeds"][:4]):
        t_before = t_core.flat.params.detach().double().clone()
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        alpha = min(1 - 1 / (i + 1), 0.99)
        s_now = s_core.flat.params.detach().double()
        want = alpha * t_before + (1 - alpha) * s_now
        got = t_core.flat.params.detach().double()
        # one fp32 ulp of the larger of the two rounded products (the terms may cancel: BatchNorm betas around zero)
        ulp = torch.clamp(torch.maximum((alpha * t_before).abs(), ((1 - alpha) * s_now).abs()), min=1e-30) * EPS32
        err = ((got - want).abs() / ulp).max().item()
        moved = (got - t_before).abs().max().item()
        worst = max(worst, err)
        assert moved > 0 or i == 0, "the teacher did not move in iteration %d" % i
        assert err <= 2.0, "iteration %d: teacher is %.2f ulp from the EMA of the student" % (i, err)
    print("mt %s: teacher vs fp64 EMA of the engine's student, worst element %.2f ulp over 4 iterations" % (dtype, worst))
