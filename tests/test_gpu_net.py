"""GPU parity of the whole hot path against the CPU oracle (oracle/torch_oracle.py, itself pinned
bit-exact to the reference) and the committed golden fixtures generated from the REAL reference.

fp32 engine mode is the parity gate (north_star: argmax bit-exact, logits/loss within 1e-3 rel);
bf16 mode is the throughput mode and is checked with the looser, explicitly stated tolerances below.
The randomly initialised 101-layer net amplifies 1-ulp perturbations into ~1e-5 loss and ~3% stem-
gradient changes after one step (measured on the reference itself), so multi-step
weight comparisons use tolerances relative to the size of the update."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def _args(**kw):
    import argparse
    a = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False,
                           lr=2.5e-4, momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False,
                           power=-1, last_epoch=-1, epochs=1, iters_per_epoch=4, ignore_index=255,
                           labeled_batch_size=2, unlabeled_batch_size=0, ignore_unlabeled=True,
                           is_epoch_lrer=False, log_freq=1000, task="sseg", engine_dtype="fp32",
                           cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99,
                           gaussian_noise_std=None)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def _core(dtype, state, backbone="resnet101"):
    from pixelssl_amd.engine import DeepLabV2Core
    core = DeepLabV2Core(backbone=backbone, device=DEV, engine_dtype=dtype)
    core.load_state_dict(state)
    core.train()
    return core


def test_forward_backward_fp32_vs_reference_fixture():
    """Engine (fp32) on the reference's own fixture: logits, argmax, CE, latent, selected gradients."""
    import torch_oracle as TO
    from pixelssl_amd import functional as PF
    fx = _load("deeplabv2_forward_65.pt")
    state = TO.init_deeplabv2_state(seed=fx["weight_seed"])
    core = _core(torch.float32, state)
    x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=fx["data_seed"], block=fx["block"])
    logits, prob, latent_fn = core(x.to(DEV))
    torch.cuda.synchronize()
    lg = logits.detach().cpu()
    err = rel(lg, fx["logits"])
    print("fp32 logits rel err %.3e" % err)
    assert err < 1e-3
    agree = (lg.argmax(1).to(torch.uint8) == fx["argmax"]).float().mean().item()
    print("argmax agreement %.6f" % agree)
    # indices must match wherever the reference's top-2 margin exceeds the numeric noise
    top2 = fx["logits"].topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-3 * fx["logits"].abs().max()
    assert torch.equal(lg.argmax(1)[decided].to(torch.uint8), fx["argmax"][decided])
    assert agree > 0.999
    assert rel(prob.detach().cpu(), torch.softmax(fx["logits"], 1)) < 1e-3
    lat = latent_fn().cpu()
    assert rel(lat.reshape(-1)[:256], fx["latent_head"]) < 1e-3
    assert abs(lat.double().sum().item() - fx["latent_sum"]) < 1e-3 * lat.double().abs().sum().item()
    ps = PF.cross_entropy_per_sample(logits, gt.to(DEV), 255)
    assert rel(ps.detach().cpu(), fx["per_sample"]) < 1e-3
    ps.mean().backward()
    torch.cuda.synchronize()
    g1 = core.backbone.conv1.weight.grad.cpu()
    ga = getattr(core.classifier.conv2d_list, "1").weight.grad.cpu().reshape(-1)[:512]
    g3 = getattr(core.backbone.layer3, "5").conv2.weight.grad.cpu()
    print("grad rel errs: conv1 %.3e aspp %.3e layer3.5.conv2 %.3e" %
          (rel(g1, fx["grad_conv1"]), rel(ga, fx["grad_aspp1_head"]), rel(g3.reshape(-1)[:512], fx["grad_l3_head"])))
    assert rel(ga, fx["grad_aspp1_head"]) < 2e-3
    # deeper gradients: 104 train-mode-BN layers on 2x5x5-pixel maps amplify fp32 summation-order noise
    # (the reference itself moves its stem gradient by 3% under a 1-ulp weight change);
    # the executor's backward plan is pinned tightly on a shallow trunk in test_shallow_trunk_*.
    assert rel(g3.reshape(-1)[:512], fx["grad_l3_head"]) < 0.15
    assert rel(g1, fx["grad_conv1"]) < 0.15
    # running statistics follow F.batch_norm(momentum 0.1, unbiased variance)
    sd = core.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.layer4.0.downsample.1.running_var"):
        assert rel(sd[k].cpu().reshape(-1)[:64], fx["probes"][k]["head"]) < 1e-3


def test_every_gradient_fp32_vs_oracle():
    """All 320 parameter gradients against the oracle on a fresh seeded batch (65x65, B=2)."""
    import torch_oracle as TO
    from pixelssl_amd import functional as PF
    state = TO.init_deeplabv2_state(seed=5)
    x, gt = TO.synthetic_batch(2, 65, 2, seed=6, block=16)
    leaves = TO._param_leaves(TO.clone_state(state))
    run = TO._with_leaves(TO.clone_state(state), leaves)
    o_logits, o_prob, _, _ = TO.deeplabv2_forward(run, x, train=True)
    # use both heads so the softmax-Jacobian path of the backward is exercised too
    w = torch.randn(o_prob.shape, generator=torch.Generator().manual_seed(1)) * 1e-3
    (TO.sseg_criterion(o_logits, gt).mean() + (o_prob * w).sum()).backward()
    core = _core(torch.float32, state)
    logits, prob, _ = core(x.to(DEV))
    (PF.cross_entropy_per_sample(logits, gt.to(DEV), 255).mean() + (prob * w.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    bad = []
    for name, p in core.named_parameters():
        e = rel(p.grad.cpu(), leaves[name].grad)
        if e > worst[1]:
            worst = (name, e)
        tol = 2e-3 if name.startswith("classifier") else 0.15      # see the amplification note above
        if e > tol:
            bad.append((name, e))
    print("worst gradient rel err: %s %.3e" % worst)
    assert not bad, bad[:10]


def test_suponly_and_mt_steps_fp32_vs_reference_meters():
    """The mirrored SSLNULL / SSLMT train steps reproduce the reference's logged losses (1e-3 rel)."""
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    # ---- SupOnly
    fx = _load("suponly_65.pt")
    args = _args(labeled_batch_size=fx["batch"], iters_per_epoch=fx["max_iters"])
    algo = P.ssl_algorithm.ssl_null.ssl_null(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)},
                                            {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.model.module.model.load_state_dict(TO.init_deeplabv2_state(seed=fx["weight_seed"]))
    algo.model.train()
    losses = []
    for s in fx["data_seeds"]:
        x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=s, block=fx["block"])
        loss, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
        losses.append(loss.item())
    print("suponly losses", losses, "reference (oracle) ", fx["oracle_losses"])
    assert abs(losses[0] - fx["oracle_losses"][0]) < 1e-3 * abs(fx["oracle_losses"][0])
    # after one SGD step the ill-conditioned random-init net (104 train-mode BN layers on 2x5x5-pixel maps) turns
    # fp32 summation-order noise into a visible loss change: repeated runs of the SAME binary give 2.88 .. 2.98
    # here (atomic accumulation order differs run to run) against the oracle's 2.983, so the second iteration is
    # a sanity band, not a parity bar -- parity is pinned by iteration 0 and by the shallow-trunk tests below.
    assert abs(losses[1] - fx["oracle_losses"][1]) < 0.15 * abs(fx["oracle_losses"][1])
    # post-step weights / later iterations: pinned on the conditioned six-iteration fixtures (tests/test_multistep.py:
    # losses 1e-3, weights within 5 % of the update); this ill-conditioned 65 x 65 fixture pins iteration 0 only
    # ---- Mean Teacher
    fx = _load("mt_65.pt")
    args = _args(labeled_batch_size=fx["lbs"], unlabeled_batch_size=fx["ubs"], ignore_unlabeled=False,
                 iters_per_epoch=fx["max_iters"])
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)},
                                        {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.module.model.load_state_dict(TO.init_deeplabv2_state(seed=fx["weight_seed"]))
    algo.t_model.module.model.load_state_dict(TO.init_deeplabv2_state(seed=fx["weight_seed"] + 1))
    algo.s_model.train()
    algo.t_model.train()
    for i, s in enumerate(fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        ref = fx["per_iter"][i]
        got = {k: v.item() for k, v in out.items()}
        print("mt iter", i, got, ref)
        for k in ref:
            assert abs(got[k] - ref[k]) < (1e-3 if i == 0 else 0.15) * abs(ref[k]) + 1e-7, (i, k)
    # post-step weights / later iterations: pinned on the conditioned six-iteration fixtures (tests/test_multistep.py:
    # losses 1e-3, weights within 5 % of the update); this ill-conditioned 65 x 65 fixture pins iteration 0 only


SHALLOW = (2, 2, 2, 3)     # stem + 9 bottlenecks (identity + strided + dilated blocks) + ASPP: every op kind


def _oracle_run(state, x, gt, w, dtype, train, noise=0.0):
    """Oracle forward/backward in `dtype` (fp64 = ground truth); optional multiplicative weight noise."""
    import torch_oracle as TO
    st = TO.clone_state(state)
    g = torch.Generator().manual_seed(99)
    for k in st:
        if st[k].is_floating_point():
            st[k] = st[k].to(dtype)
            if noise and not TO.is_buffer(k):
                st[k] = st[k] * (1 + noise * torch.randn(st[k].shape, generator=g).to(dtype))
    leaves = TO._param_leaves(st)
    run = TO._with_leaves(st, leaves)
    logits, prob, lat, _ = TO.deeplabv2_forward(run, x.to(dtype), train=train, layers=SHALLOW)
    loss = TO.sseg_criterion(logits, gt).mean() + (prob * w.to(dtype)).sum()
    loss.backward()
    return dict(logits=logits.detach(), latent=lat.detach(), loss=loss.item(), run=run,
                grads={k: v.grad for k, v in leaves.items()})


def _shallow_setup(train, size=97, batch=4, seed=12):
    import torch_oracle as TO
    state = TO.init_deeplabv2_state(seed=seed, layers=SHALLOW)
    if not train:       # non-trivial running statistics for the eval-mode (freeze_bn) path
        g = torch.Generator().manual_seed(4)
        for k in state:
            if k.endswith("running_mean"):
                state[k] = torch.randn(state[k].shape, generator=g) * 0.05
            elif k.endswith("running_var"):
                state[k] = torch.rand(state[k].shape, generator=g) + 0.5
    x, gt = TO.synthetic_batch(batch, size, batch, seed=seed + 1, block=16)
    w = torch.randn(batch, 21, size, size, generator=torch.Generator().manual_seed(1)) * 1e-3
    return state, x, gt, w


def _engine_run(state, x, gt, w, dtype, train):
    from pixelssl_amd import functional as PF
    core = _core(dtype, state, backbone=SHALLOW)
    core.train(train)
    logits, prob, latent_fn = core(x.to(DEV))
    loss = PF.cross_entropy_per_sample(logits, gt.to(DEV), 255).mean() + (prob * w.to(DEV)).sum()
    loss.backward()
    torch.cuda.synchronize()
    return dict(core=core, logits=logits.detach().cpu(), latent=latent_fn().cpu(), loss=loss.item(),
                grads={k: p.grad.cpu() for k, p in core.named_parameters()})


def _assert_grads_as_accurate(e, o, t, tag, factor=3.0, slack=1e-2):
    """Every engine gradient is at most `factor` x as far from the fp64 ground truth `t` as the fp32
    reference arithmetic `o` is (+ a slack at the ReLU/max-pool mask-flip noise level: the oracle's own
    fp32-vs-fp64 gradient distance is 1e-3..1e-2 on these train-mode-BN nets).  A wrong or missing gradient path
    gives O(0.1..1) errors and fails this by two orders of magnitude."""
    rows = []
    for k in t["grads"]:
        eo, ee = rel(o["grads"][k], t["grads"][k]), rel(e["grads"][k], t["grads"][k])
        rows.append((ee - factor * eo, k, ee, eo))
    rows.sort(reverse=True)
    print("%s: worst gradient %s: engine %.2e vs fp64, reference fp32 %.2e vs fp64" % (tag, rows[0][1], rows[0][2], rows[0][3]))
    assert rows[0][0] < slack, rows[:5]


def _engine_run_with_decisions(state, x, gt, w, train, wl=None, backbone=SHALLOW):
    """fp32 engine forward + backward, returning its gradients AND every ReLU decision it took (core.relu_decisions)."""
    from pixelssl_amd import functional as PF
    core = _core(torch.float32, state, backbone=backbone)
    core.train(train)
    core.keep_arena = True
    if wl is None:
        logits, prob, latent_fn = core(x.to(DEV))
        loss = PF.cross_entropy_per_sample(logits, gt.to(DEV), 255).mean() + (prob * w.to(DEV)).sum()
    else:
        logits, prob, latent = core.forward_with_latent(x.to(DEV))
        loss = PF.cross_entropy_per_sample(logits, gt.to(DEV), 255).mean() + (latent * wl.to(DEV)).sum()
    dec = {k: v.cpu() for k, v in core.relu_decisions(core._last_arena).items()}
    loss.backward()
    torch.cuda.synchronize()
    return dict(core=core, logits=logits.detach().cpu(), loss=loss.item(), decisions=dec,
                grads={k: p.grad.cpu() for k, p in core.named_parameters()})


def _oracle_run_with_decisions(state, x, gt, w, train, decisions, wl=None, layers=SHALLOW):
    """fp64 oracle whose ReLUs take the given decisions (activation = z * decision): the exact gradient of the network
    CONDITIONED on the engine's discrete choices.  Also returns where the oracle itself would have decided differently."""
    import torch_oracle as TO
    st = TO.clone_state(state)
    for k in st:
        if st[k].is_floating_point():
            st[k] = st[k].double()
    leaves = TO._param_leaves(st)
    run = TO._with_leaves(st, leaves)
    flips = {}

    def act(name, z):
        d = decisions[name]
        own = z.detach() > 0
        diff = own != d
        flips[name] = (int(diff.sum()), float(z.detach()[diff].abs().max()) if diff.any() else 0.0, float(z.detach().abs().mean()), d.numel())
        return z * d.to(z.dtype)
    logits, prob, lat, _ = TO.deeplabv2_forward(run, x.double(), train=train, layers=layers, relu=act)
    loss = TO.sseg_criterion(logits, gt).mean() + ((prob * w.double()).sum() if wl is None else (lat * wl.double()).sum())
    loss.backward()
    return dict(logits=logits.detach(), loss=loss.item(), grads={k: v.grad for k, v in leaves.items()}, flips=flips)


def _assert_decisions_and_gradients(e, c, tag):
    """(1) the engine's ReLU decisions differ from the exact network's only on pre-activations within rounding of
    zero, a measure-zero set (< 1e-5 of the elements, |z| < 1e-5 of the layer's mean |z|); (2) conditioned on its
    decisions, EVERY engine gradient is within 1e-5 of the exact one."""
    nflip = sum(v[0] for v in c["flips"].values())
    total = sum(v[3] for v in c["flips"].values())
    worst_z = max((v[1] / (v[2] + 1e-30) for v in c["flips"].values()), default=0.0)
    print("%s: %d of %d ReLU decisions differ from the exact network's (largest |z| among them: %.1e of the layer mean)"
          % (tag, nflip, total, worst_z))
    assert nflip <= 1e-5 * total + 2 and worst_z < 1e-5
    rows = sorted(((rel(e["grads"][k], c["grads"][k]), k) for k in c["grads"]), reverse=True)
    print("%s: worst gradient given the decisions: %s %.2e (median %.2e)" % (tag, rows[0][1], rows[0][0], rows[len(rows) // 2][0]))
    assert rows[0][0] < 1e-5, rows[:5]


def test_shallow_trunk_eval_bn_every_gradient_tight():
    """Forward AND backward plan of the executor (every op kind, both heads of the output), eval-mode BN (running
    statistics: a well conditioned network), every parameter gradient at 1e-5.

    Round 1 measured 3e-3 here against torch fp32's 1e-6 and could not explain it.  Root cause (tools/diag_fp32_grad*.py):
    ReLU decisions.  A pre-activation within fp32 rounding of zero (about 1 element in 4e5 per layer) can land on either
    side in two correct fp32 implementations; each such flip changes one element's gradient by O(1), i.e. ~1e-3 of a
    white test gradient's norm, and the flips of ~30 layers add up.  It is not an arithmetic error of the backward
    pass: with the engine's own decisions its gradients are exact (e.g. dbeta = sum(dout * (out > 0)) to the last bit).
    So the comparison is made rigorous instead of loose: the fp64 oracle is run WITH the engine's decisions
    (torch_oracle.resnet_forward(relu=...), core.relu_decisions) and the bar is 1e-5 on every gradient, plus a bound on
    how many decisions differ from the exact network's and how close to zero those pre-activations are."""
    state, x, gt, w = _shallow_setup(train=False)
    o = _oracle_run(state, x, gt, w, torch.float32, train=False)
    e = _engine_run_with_decisions(state, x, gt, w, train=False)
    c = _oracle_run_with_decisions(state, x, gt, w, False, e["decisions"])
    assert rel(e["logits"], o["logits"]) < 1e-5
    assert abs(e["loss"] - o["loss"]) < 1e-5 * abs(o["loss"])
    _assert_decisions_and_gradients(e, c, "shallow eval-BN fp32")
    # eval mode leaves the running statistics untouched
    sd = e["core"].state_dict()
    assert torch.equal(sd["backbone.bn1.running_var"].cpu(), state["backbone.bn1.running_var"])


def test_full_depth_resnet101_eval_bn_every_gradient_tight():
    """The same bar on the FULL ResNet-101 + ASPP (all 320 parameter tensors, 104 convolutions): conditioned weights
    (torch_oracle.condition_state), eval-mode BN with non-trivial running statistics, 97 x 97, B = 2."""
    import torch_oracle as TO
    state = TO.condition_state(TO.init_deeplabv2_state(seed=17), 0.1)
    g = torch.Generator().manual_seed(4)
    for k in state:
        if k.endswith("running_mean"):
            state[k] = torch.randn(state[k].shape, generator=g) * 0.05
        elif k.endswith("running_var"):
            state[k] = torch.rand(state[k].shape, generator=g) + 0.5
    x, gt = TO.synthetic_batch(2, 97, 2, seed=18, block=16)
    w = torch.randn(2, 21, 97, 97, generator=torch.Generator().manual_seed(1)) * 1e-3
    e = _engine_run_with_decisions(state, x, gt, w, train=False, backbone="resnet101")
    c = _oracle_run_with_decisions(state, x, gt, w, False, e["decisions"], layers=TO.RESNET101)
    assert rel(e["logits"], c["logits"]) < 1e-5 and abs(e["loss"] - c["loss"]) < 1e-5 * abs(c["loss"])
    assert len(c["grads"]) == 320
    _assert_decisions_and_gradients(e, c, "full-depth eval-BN fp32")


def test_deeplab_latent_gradient_is_seeded_into_the_executor():
    """SSLCCT on DeepLab-v2 (task/sseg/func.py:228: 2048-channel latent): a loss on the latent handed out by
    forward_with_latent() sends its gradient back through the executor (pxl_net_seed_latent_grad) and adds to the
    head's.  Every parameter gradient at 1e-5 against the exact network taking the engine's ReLU decisions (see
    test_shallow_trunk_eval_bn_every_gradient_tight)."""
    from pixelssl_amd import functional as PF
    state, x, gt, w = _shallow_setup(train=False)
    hw = (x.shape[2] + 15) // 16
    wl = torch.randn(x.shape[0], 2048, hw, hw, generator=torch.Generator().manual_seed(3)) * 1e-3
    e = _engine_run_with_decisions(state, x, gt, w, train=False, wl=wl)
    c = _oracle_run_with_decisions(state, x, gt, w, False, e["decisions"], wl=wl)
    _assert_decisions_and_gradients(e, c, "latent-seeded DeepLab fp32")
    # the latent term really contributes: without it the trunk gradient is different
    core = e["core"]
    core.flat.grads.zero_()
    logits, _, _ = core(x.to(DEV))
    PF.cross_entropy_per_sample(logits, gt.to(DEV), 255).mean().backward()
    torch.cuda.synchronize()
    assert rel(core.backbone.conv1.weight.grad.cpu(), e["grads"]["backbone.conv1.weight"]) > 1e-2


def test_shallow_trunk_train_bn_as_accurate_as_fp32_reference():
    """Train-mode BN makes the gradients of this random-init net ill conditioned: the fp32 oracle itself is
    ~1e-2 away from an fp64 run (and a 1e-7 weight perturbation moves gradients by 6e-3).  Gate: the fp32
    engine is at most 3x as far from the fp64 ground truth as the fp32 reference arithmetic is."""
    state, x, gt, w = _shallow_setup(train=True)
    t = _oracle_run(state, x, gt, w, torch.float64, train=True)        # ground truth
    o = _oracle_run(state, x, gt, w, torch.float32, train=True)        # the reference's own arithmetic
    e = _engine_run(state, x, gt, w, torch.float32, train=True)
    assert rel(e["logits"], t["logits"]) < max(3 * rel(o["logits"], t["logits"]), 1e-5)
    # bit-exact indices wherever the top-2 margin is above the fp32 noise of the reference itself
    top2 = t["logits"].topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-4 * t["logits"].abs().max()
    assert torch.equal(e["logits"].argmax(1)[decided], o["logits"].argmax(1)[decided])
    assert (e["logits"].argmax(1) == o["logits"].argmax(1)).float().mean().item() > 0.9999
    assert abs(e["loss"] - t["loss"]) < 1e-5 * abs(t["loss"])
    _assert_grads_as_accurate(e, o, t, "shallow train-BN fp32")
    sd = e["core"].state_dict()
    for k, v in o["run"].items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert rel(sd[k].cpu(), v) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v) == 1


def test_shallow_trunk_bf16_vs_perturbed_reference():
    """Throughput mode.  Eval-mode BN (well conditioned): logits within 3e-2, every gradient at most 4x as far
    from the fp64 truth as an fp64 run with 2^-9 (bf16-epsilon) multiplicative weight noise.
    Train-mode BN: bf16 rounding (2^-9) is amplified like any other perturbation, so the gate is relative
    to the fp64 oracle run with 2^-9 multiplicative weight noise (4x its error)."""
    state, x, gt, w = _shallow_setup(train=False)
    t = _oracle_run(state, x, gt, w, torch.float64, train=False)
    n = _oracle_run(state, x, gt, w, torch.float64, train=False, noise=2.0 ** -9)
    e = _engine_run(state, x, gt, w, torch.bfloat16, train=False)
    print("shallow eval-BN bf16: logits rel %.3e (2^-9-noise reference %.3e)" % (rel(e["logits"], t["logits"]), rel(n["logits"], t["logits"])))
    assert rel(e["logits"], t["logits"]) < 3e-2
    _assert_grads_as_accurate(e, n, t, "shallow eval-BN bf16 vs 2^-9-noise reference", factor=4.0, slack=2e-2)
    state, x, gt, w = _shallow_setup(train=True)
    t = _oracle_run(state, x, gt, w, torch.float64, train=True)
    n = _oracle_run(state, x, gt, w, torch.float64, train=True, noise=2.0 ** -9)
    e = _engine_run(state, x, gt, w, torch.bfloat16, train=True)
    el, nl = rel(e["logits"], t["logits"]), rel(n["logits"], t["logits"])
    print("shallow train-BN bf16: logits rel %.3e (2^-9-noise reference %.3e), loss %.5f vs %.5f" % (el, nl, e["loss"], t["loss"]))
    assert el < 4 * nl and abs(e["loss"] - t["loss"]) < 2e-2 * abs(t["loss"])
    assert all(torch.isfinite(g).all() for g in e["grads"].values())


def test_bf16_mode_tracks_fp32_oracle():
    """Throughput mode: bf16 activations/weights, fp32 accumulate + fp32 BN statistics.
    On the full-depth random-init net bf16 rounding (2^-9) is amplified until the logits decorrelate
    (measured: rel 0.68, argmax agreement 22%), so only the loss (0.15 rel; measured 7.6e-2) and finiteness are gated
    here; the bf16 numerics are gated on the shallow trunk above."""
    import torch_oracle as TO
    from pixelssl_amd import functional as PF
    fx = _load("deeplabv2_forward_65.pt")
    core = _core(torch.bfloat16, TO.init_deeplabv2_state(seed=fx["weight_seed"]))
    x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=fx["data_seed"], block=fx["block"])
    logits, prob, _ = core(x.to(DEV))
    ps = PF.cross_entropy_per_sample(logits, gt.to(DEV), 255)
    ps.mean().backward()
    torch.cuda.synchronize()
    e = rel(logits.detach().cpu(), fx["logits"])
    agree = (logits.detach().cpu().argmax(1).to(torch.uint8) == fx["argmax"]).float().mean().item()
    le = rel(ps.detach().cpu(), fx["per_sample"])
    print("bf16: logits rel %.3e  argmax agreement %.4f  CE rel %.3e" % (e, agree, le))
    assert torch.isfinite(logits).all() and le < 0.15
    assert all(torch.isfinite(p.grad).all() for p in core.parameters())


def test_eval_mode_uses_running_statistics():
    import torch_oracle as TO
    state = TO.init_deeplabv2_state(seed=3)
    g = torch.Generator().manual_seed(4)
    for k in state:
        if k.endswith("running_mean"):
            state[k] = torch.randn(state[k].shape, generator=g) * 0.05
        elif k.endswith("running_var"):
            state[k] = torch.rand(state[k].shape, generator=g) + 0.5
    x, _ = TO.synthetic_batch(1, 49, 1, seed=9, block=16)
    ref, _, _, _ = TO.deeplabv2_forward(TO.clone_state(state), x, train=False)
    core = _core(torch.float32, state)
    core.eval()
    with torch.no_grad():
        logits, _, _ = core(x.to(DEV))
    torch.cuda.synchronize()
    assert rel(logits.cpu(), ref) < 1e-3


def test_full_size_properties_513():
    """BASELINE size (8 x 513 x 513, bf16): size-independent properties instead of an oracle run --
    softmax rows sum to 1, logits finite, the step is deterministic in its loss to 1e-3, a zero learning
    rate leaves the weights untouched, and a repeated forward after a step changes the loss."""
    import pixelssl_amd as P
    import torch_oracle as TO
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    args = _args(labeled_batch_size=4, unlabeled_batch_size=4, ignore_unlabeled=False, engine_dtype="bf16",
                 iters_per_epoch=100)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)},
                                        {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.train()
    algo.t_model.train()
    x, gt = TO.synthetic_batch(8, 513, 4, seed=77)
    x, gt = x.to(DEV), gt.to(DEV)
    with torch.no_grad():
        r, _ = algo.t_model.forward((x,))
    p = r["activated_pred"][0]
    assert torch.isfinite(r["pred"][0]).all()
    assert (p.sum(1) - 1).abs().max().item() < 1e-4
    assert tuple(r["sslcct_ad_inp"].shape) == (8, 2048, 33, 33)
    w0 = algo.s_model.module.model.flat.params.clone()
    out1, _, _ = algo.train_step((x,), (gt,), 0, 300)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in out1.values())
    assert not torch.equal(w0, algo.s_model.module.model.flat.params)
    # EMA with alpha = 0 at step 0 copies the student into the teacher (ssl_mt.py:361)
    assert torch.equal(algo.t_model.module.model.flat.params, algo.s_model.module.model.flat.params)


@pytest.mark.gpu
@pytest.mark.parametrize("layers", [(1, 1, 1, 1), (2, 2, 2, 2)])
def test_bn_apply_on_load_equals_the_materialised_path(layers, monkeypatch):
    """PXL_BN_ONLOAD (default on): conv3 of every bottleneck reads the RAW output of conv2 and applies relu(bn2(.)) to the
    tiles in LDS (bit-identical per launch: tests/test_gpu_kernels.py::test_conv_with_bn_apply_on_load).  Whole network,
    against the same network with materialised activations (PXL_BN_ONLOAD=0): two runs of ONE path already differ by bf16
    roundings (the BN statistics are fp32 atomics, their order moves a coefficient by an ulp), so the bar is the noise
    floor measured here between two runs of the materialised path: logits, every parameter gradient, running statistics,
    in a training pass, a no-grad training-mode pass (the MT teacher) and an eval pass."""
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd import functional as PF
    x, gt = TO.synthetic_batch(3, 97, 3, seed=77, block=16)
    state = None

    def run(mode):
        nonlocal state
        monkeypatch.setenv("PXL_BN_ONLOAD", mode)
        core = DeepLabV2Core(backbone=layers, device="cuda", engine_dtype=torch.bfloat16)
        if state is None:
            core.reset_parameters(torch.Generator().manual_seed(5))
            with torch.no_grad():                      # conditioned trunk: no chaotic amplification of 1-ulp differences
                for name, prm in core.named_parameters():
                    if name.endswith("bn3.weight"):
                        prm.mul_(0.1)
            core.mark_params_changed()
            state = {k: v.detach().clone() for k, v in core.state_dict().items()}
        else:
            core.load_state_dict(state)
        core.train()
        logits, prob, _ = core(x.cuda())
        PF.cross_entropy_per_sample(logits, gt.cuda(), 255).mean().backward()
        grads = torch.cat([p.grad.detach().float().reshape(-1) for p in core.parameters()]).cpu()
        runs = torch.cat([v.detach().float().reshape(-1) for k, v in core.named_buffers() if "running" in k]).cpu()
        with torch.no_grad():
            l2, _, _ = core(x.cuda())                 # no-grad, train-mode BN: the MT teacher's pass
        core.eval()
        with torch.no_grad():
            l3, _, _ = core(x.cuda())
        return [logits.detach().cpu(), grads, runs, l2.cpu(), l3.cpu()]
    base, again, onload = run("0"), run("0"), run("1")
    rel = lambda a, b: ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
    for name, b0, b1, o in zip(("training logits", "gradients", "running statistics", "no-grad logits", "eval logits"), base, again, onload):
        floor, got = rel(b1, b0), rel(o, b0)
        print("%s: on-load vs materialised %.2e, materialised vs itself %.2e" % (name, got, floor))
        assert got <= 3 * floor + 2e-3, (name, got, floor)
