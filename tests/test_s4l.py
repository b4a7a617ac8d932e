"""S4L (pixelssl/ssl_algorithm/ssl_s4l.py): oracle vs the fixtures generated from the reference's own
SSLS4L._train (not gpu); rotation classifier forward / input gradient / parameter gradients on the executor, and the
mirrored training step against the reference's logged meters and weight updates (gpu; fp32 engine: 1e-3, bf16 engine:
1e-2 on the losses)."""
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
KEYS = ("unrotated_task_loss", "rotated_task_loss", "rotation_loss", "rotation_acc")


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _standalone_pred(fx):
    g = torch.Generator().manual_seed(fx["standalone"]["seed"])
    return torch.randn(4, 21, fx["size"], fx["size"], generator=g) * 2


def test_oracle_reproduces_reference_fixture():
    """CPU: the restated rotation classifier and TWO iterations of the restated S4L step give the numbers the reference's
    own modules / _train produced (fixture s4l_65.pt), including the numpy draw of the rotation angles."""
    import torch_oracle as TO
    import s4l_oracle as SO
    fx = torch.load(os.path.join(GOLD, "s4l_65.pt"), weights_only=False)
    st = fx["standalone"]
    pred = _standalone_pred(fx).requires_grad_(True)
    leaves = OrderedDict((k, v.clone().requires_grad_(not SO.rc_is_buffer(k))) for k, v in SO.init_rc_state(21, seed=st["seed"] + 3).items())
    out = SO.rc_forward(leaves, pred, train=True)
    torch.nn.functional.cross_entropy(out, torch.tensor([0, 1, 2, 3])).backward()
    assert torch.allclose(out, st["logits"], atol=1e-6)
    assert torch.allclose(pred.grad[:, :, :4, :8], st["dpred_head"], rtol=1e-4, atol=1e-10)
    assert torch.allclose(leaves["bn2.weight"].grad, st["dbn2"], rtol=1e-4, atol=1e-9)
    assert torch.allclose(leaves["bn1.running_var"], st["bn1_rv"], rtol=1e-6)
    # the rotations: 4 x 90 degrees is the identity, 2 x 180 too, index 1 then 3 cancel
    t = torch.arange(2 * 5 * 5, dtype=torch.float32).reshape(2, 5, 5)
    assert torch.equal(SO.rotate(SO.rotate(t, 1), 3), t) and torch.equal(SO.rotate(SO.rotate(t, 2), 2), t)
    assert torch.equal(SO.rotate(SO.rotate(t, 1), 1), SO.rotate(t, 2))
    # the training loop (first iteration only: ~20 s of CPU)
    np.random.seed(fx["np_seed"])
    bs = fx["lbs"] + fx["ubs"]
    angles = SO.draw_angles(bs)
    assert angles.tolist() == fx["angles"][0]
    tr = SO.S4LOracleTrainer(TO.init_deeplabv2_state(seed=fx["weight_seed"]), SO.init_rc_state(21, seed=fx["rc_seed"]),
                             dict(max_iters=fx["max_iters"], rotated_sup_scale=fx["rotated_sup_scale"],
                                  rotation_scale=fx["rotation_scale"]))
    x, gt = TO.synthetic_batch(bs, fx["size"], fx["lbs"], seed=fx["data_seeds"][0], block=fx["block"])
    o = tr.s4l_step(x, gt, fx["lbs"], angles)
    for k in KEYS:
        assert abs(o[k] - fx["ref_per_iter"][0][k]) <= 2e-5 * abs(fx["ref_per_iter"][0][k]) + 1e-9, k
    assert torch.allclose(o["pred_rotation"], fx["pred_rotation0"], atol=1e-5)


def test_state_dict_and_param_groups_follow_the_reference():
    """Names, shapes and ORDER of the wrapped model's parameters (an optimizer state_dict indexes by position)."""
    os.environ.setdefault("PXL_FORCE_DEVICE", "cpu")
    from pixelssl_amd.engine import RotationClassifierCore
    fx = torch.load(os.path.join(GOLD, "s4l_65.pt"), weights_only=False)
    core = RotationClassifierCore(21, device="cpu")
    names = [k for k, _ in core.named_parameters()]
    assert ["rotation_classifier." + k for k in names] == fx["param_names"]
    sd = core.state_dict()
    for k, v in fx["rc_after"].items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    core.load_state_dict(fx["rc_after"])
    for k, v in fx["rc_after"].items():
        assert torch.equal(core.state_dict()[k].cpu(), v), k
    # the executor's channel padding stays zero after a load
    assert float(core.flat.params.abs().sum()) == pytest.approx(sum(float(v.abs().sum()) for k, v in fx["rc_after"].items()
                                                                       if k.endswith("weight") or k.endswith("bias")), rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_rotation_classifier_matches_reference(dtype):
    import s4l_oracle as SO
    from pixelssl_amd.ssl_algorithm.ssl_s4l import RotationClassifer, RotationCrossEntropy
    fx = torch.load(os.path.join(GOLD, "s4l_65.pt"), weights_only=False)
    st = fx["standalone"]
    tol = 1e-3 if dtype == "fp32" else 1e-1          # (bf16 measured: logits 4e-3, d pred 4e-2, d conv1 5e-2)
    rc = RotationClassifer(21, engine_dtype=torch.float32 if dtype == "fp32" else torch.bfloat16)
    rc.load_state_dict(SO.init_rc_state(21, seed=st["seed"] + 3))
    rc.train()
    pred = _standalone_pred(fx).to(DEV).requires_grad_(True)
    out = rc(pred)
    loss = RotationCrossEntropy()(out, torch.tensor([0, 1, 2, 3], device=DEV))
    loss.backward()
    torch.cuda.synchronize()
    print("rc %s: logits %.2e  dpred %.2e  dconv1 %.2e  dbn2 %.2e  dcls_bias %.2e" % (
        dtype, rel(out.detach().cpu(), st["logits"]), rel(pred.grad[:, :, :4, :8].cpu(), st["dpred_head"]),
        rel(rc.conv1.weight.grad.cpu().reshape(-1)[:256], st["dconv1"]), rel(rc.bn2.weight.grad.cpu(), st["dbn2"]),
        rel(rc.classifier.bias.grad.cpu(), st["dcls_bias"])))
    assert rel(out.detach().cpu(), st["logits"]) < tol
    assert rel(pred.grad[:, :, :4, :8].cpu(), st["dpred_head"]) < tol
    assert abs(pred.grad.double().abs().sum().item() - st["dpred_abssum"]) < tol * st["dpred_abssum"]
    assert rel(rc.conv1.weight.grad.cpu().reshape(-1)[:256], st["dconv1"]) < tol
    assert rel(rc.bn2.weight.grad.cpu(), st["dbn2"]) < tol and rel(rc.classifier.bias.grad.cpu(), st["dcls_bias"]) < tol
    assert rel(rc.bn1.running_var.cpu(), st["bn1_rv"]) < tol
    # the padded channel slots of the executor stayed exactly zero (parameters and gradients)
    core = rc.core
    real = sum(float(p.detach().abs().sum()) for p in core.parameters())
    assert abs(float(core.flat.params.abs().sum()) - real) <= 1e-6 * real
    greal = sum(float(p.grad.abs().sum()) for p in core.parameters())
    assert abs(float(core.flat.grads.abs().sum()) - greal) <= 1e-6 * greal + 1e-12


def _algo(fx, dtype):
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    lbs, ubs = fx["lbs"], fx["ubs"]
    args = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4,
                              momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1,
                              epochs=1, iters_per_epoch=fx["max_iters"], ignore_index=255, labeled_batch_size=lbs,
                              unlabeled_batch_size=ubs, batch_size=lbs + ubs, ignore_unlabeled=False, is_epoch_lrer=False,
                              log_freq=1000, task="sseg", engine_dtype=dtype, gpus=1, rotated_sup_scale=fx["rotated_sup_scale"],
                              rotation_scale=fx["rotation_scale"], visualize=False, im_size=fx["size"])
    task_func = P.sseg.func.task_func()(args)
    algo = P.ssl_algorithm.ssl_s4l.ssl_s4l(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                           {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                           task_func)
    assert args.batch_size == 2 * (lbs + ubs) and args.labeled_batch_size == 2 * lbs
    return algo


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("fixture", ["s4l_65.pt", "s4l_cond_65.pt"])
def test_ssls4l_train_steps_vs_reference(fixture, dtype):
    """The mirrored SSLS4L iteration (own _batch_prehandle with the same numpy draw, train_step) against what the
    reference's _train logged.  Reference initialisers: iteration 0 at 1e-3, iteration 1 a sanity band (ill-conditioned
    random-init task net, see test_gpu_net.py).  Conditioned weights: all four iterations at 1e-3 (fp32) / 1e-2 (bf16) and
    the weight updates of the task model AND the rotation classifier within 5 % (fp32) of their own size."""
    import torch_oracle as TO
    import s4l_oracle as SO
    fx = torch.load(os.path.join(GOLD, fixture), weights_only=False)
    cond = fx["gamma3"] is not None
    if dtype == "bf16" and not cond:
        pytest.skip("bf16 on the reference-initialised net is not a parity statement (the logits decorrelate, "
                    "tests/test_parity_513.py); the conditioned fixture carries the bf16 gate")
    algo = _algo(fx, dtype)
    state = TO.init_deeplabv2_state(seed=fx["weight_seed"])
    if cond:
        TO.condition_state(state, fx["gamma3"])
    algo.model.module.task_model.model.load_state_dict(state)
    rc0 = SO.init_rc_state(21, seed=fx["rc_seed"])
    algo.model.module.rotation_classifier.load_state_dict(rc0)
    assert [k for k, _ in algo.model.module.named_parameters()][-10:] == fx["param_names"]
    algo.model.train()
    np.random.seed(fx["np_seed"])
    bs = fx["lbs"] + fx["ubs"]
    tight = 1e-3 if dtype == "fp32" else 1e-2
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(bs, fx["size"], fx["lbs"], seed=s, block=fx["block"])
        inp, gts = algo._batch_prehandle((x,), (gt,), True)
        assert gts[-1][bs:].tolist() == fx["angles"][i] and int(gts[-1][:bs].abs().sum()) == 0
        out, _ = algo.train_step(inp, gts)
        got = {k: float(v) for k, v in out.items()}
        ref = fx["ref_per_iter"][i]
        print("s4l %s %s iter %d:" % (fixture, dtype, i), got, ref)
        for k in KEYS:
            if k == "rotation_acc":
                # one of the 8 rotation decisions may flip in bf16; after the first update of the reference-initialised (chaotic)
                # net in fp32 as well -- there two: that iteration is a sanity band (its losses are held to 15 %), and which side
                # of a tie a 4-way classifier on 8 samples lands on moved with the summation order of the ASPP split-K (round 5)
                exact = dtype == "fp32" and (cond or i == 0)
                band = 12.6 if (cond or i == 0) else 25.1
                assert abs(got[k] - ref[k]) <= (1e-4 if exact else band), (i, k, got[k], ref[k])
                continue
            # second iteration on the reference-initialised net: sanity band (measured run to run: task losses within 6 %,
            # rotation loss within 3 % of the reference's)
            tol = tight if (cond or i == 0) else 0.15
            assert abs(got[k] - ref[k]) <= tol * abs(ref[k]) + 1e-7, (i, k, got[k], ref[k])
    if cond:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_multistep import _check_weights
        _check_weights("s4l task model " + dtype, algo.model.module.task_model.model.state_dict(), fx["updates"], dtype)
        # (Linear + BN affine parameters of a 4-way classifier on 8 samples: bf16 is a sanity band)
        _check_weights("s4l rotation classifier " + dtype, algo.model.module.rotation_classifier.state_dict(), fx["rc_updates"],
                       dtype, frac=0.05 if dtype == "fp32" else 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 32, 65, 513])
@pytest.mark.parametrize("kind", ["f32", "i64", "u8"])
def test_rotate_append_kernel_vs_oracle(kind, n):
    """csrc/rotate.hip against the restated _batch_prehandle / _rotate_tensor (s4l_oracle.batch_prehandle, pinned against
    the reference's method by make_golden_s4l.py): bit-exact, for images (fp32), label maps (int64, with the 255 ignore
    value) and uint8 crops, at sizes below / at / above the 32 x 32 tile and at the BASELINE size 513."""
    import s4l_oracle as SO
    from pixelssl_amd.ssl_algorithm.ssl_s4l import SSLS4L
    g = torch.Generator().manual_seed(n * 7 + len(kind))
    bs, C = (6, 3) if n < 513 else (4, 3)
    if kind == "f32":
        t = torch.randn(bs, C, n, n, generator=g)
    elif kind == "i64":
        t = torch.randint(0, 21, (bs, 1, n, n), generator=g)
        t[:, :, ::7, ::5] = 255
    else:
        t = torch.randint(0, 256, (bs, C, n, n), generator=g, dtype=torch.uint8)
    angles = np.array([1, 2, 3, 0, 3, 1][:bs])
    want, _, _ = SO.batch_prehandle(t.float(), t.float(), angles)
    got = SSLS4L._with_rotated(SSLS4L, t.cuda(), angles)
    assert got.dtype == torch.float32 and tuple(got.shape) == tuple(want.shape)
    assert torch.equal(got.cpu(), want)


def test_rotate_append_rejects_what_it_cannot_do():
    from pixelssl_amd._lib import PixelHipError
    from pixelssl_amd.ssl_algorithm.ssl_s4l import SSLS4L
    with pytest.raises(PixelHipError):
        SSLS4L._with_rotated(SSLS4L, torch.zeros(2, 3, 8, 9), np.array([1, 2]))        # not square
    with pytest.raises(PixelHipError):
        SSLS4L._with_rotated(SSLS4L, torch.zeros(2, 3, 8, 8), np.array([1, 2]))        # no CPU fallback
