"""Generate (and self-check) the golden fixtures in tests/golden/ from the REAL
reference imported from /root/reference.  Container-only; TEST INFRASTRUCTURE.

    python oracle/make_golden.py            # writes tests/golden/*.pt

For every case it (1) runs the reference's own classes on seeded inputs with
weights produced by `torch_oracle.init_deeplabv2_state(seed)` (deterministic,
reproducible on the GPU box without the reference), (2) asserts that
`oracle/torch_oracle.py` reproduces the reference, and (3) stores the reference
outputs.  tests/test_oracle_golden.py replays (2) against the stored outputs.
"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim            # noqa: E402
import torch_oracle as TO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

BASE_CFG = dict(models={'model': 'deeplabv2'}, optimizers={'model': 'sgd'},
                lrers={'model': 'polynomiallr'}, criterions={'model': 'sseg_criterion'},
                lr=0.00025, momentum=0.9, weight_decay=0.0005, output_stride=16,
                backbone='resnet101', epochs=1, log_freq=1000)

# parameters whose post-step values are stored (full tensors are 176 MB)
PROBES = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight",
          "backbone.layer2.0.downsample.0.weight", "backbone.layer3.11.conv3.weight",
          "backbone.layer3.22.bn2.weight", "backbone.layer4.2.conv2.weight",
          "backbone.layer4.2.bn3.bias", "backbone.bn1.running_mean",
          "backbone.layer4.0.downsample.1.running_var",
          "classifier.conv2d_list.0.weight", "classifier.conv2d_list.3.bias"]


def probe(sd, prefix=""):
    out = OrderedDict()
    for k in PROBES:
        v = sd[prefix + k].detach().float().reshape(-1)
        out[k] = dict(head=v[:64].clone(), sum=float(v.double().sum()),
                      abssum=float(v.double().abs().sum()))
    return out


def subsample(v, n=4096):
    """Strided sample of a tensor (at most n values) -- the multi-step fixtures store these instead of 176 MB of
    weights; the same function samples the engine's tensors in the GPU tests."""
    v = v.detach().reshape(-1)
    return v[::max(1, v.numel() // n)][:n].float().clone()


def probe_update(final_sd, init_sd, keys):
    """For the multi-step fixtures: a strided sample of every probed tensor after training, plus the L2 norm of the
    update of that sample (||final - init||) so a test can hold the engine to a fraction of the update."""
    out = OrderedDict()
    for k in keys:
        f, i = subsample(final_sd[k]), subsample(init_sd[k])
        out[k] = dict(sample=f, update_l2=float((f.double() - i.double()).norm()),
                      update_max=float((f.double() - i.double()).abs().max()))
    return out


def record_meters(algo):
    """Every value the reference logs through `meters.update`, in order (-> per-iteration losses of its own loop)."""
    seen = []
    real_update = algo.meters.update
    algo.meters.update = lambda k, v, *a: (seen.append((k, float(v))), real_update(k, v, *a))[1]
    return seen


def per_iteration(seen, keys, iters):
    out = [{} for _ in range(iters)]
    for k in keys:
        vals = [v for kk, v in seen if kk == k]
        assert len(vals) == iters, (k, len(vals), iters)
        for i, v in enumerate(vals):
            out[i][k] = v
    return out


def with_prefix(sd, prefix):
    return OrderedDict((prefix + k, v.clone()) for k, v in sd.items())


def check(name, a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = err <= atol + rtol * ref
    print("  %-34s max|d|=%.3e  ref=%.3e  %s" % (name, err, ref, "ok" if ok else "MISMATCH"))
    if not ok:
        raise SystemExit("oracle does not reproduce the reference: " + name)


def case_forward(size=65, batch=2, seed=11):
    """Reference DeepLabV2 TaskModel forward (train-mode BN) + criterion."""
    ref = ref_shim.load_reference()
    args = ref_shim.make_args('ssl_null', dict(BASE_CFG, batch_size=batch,
                                               unlabeled_batch_size=0, im_size=size))
    state = TO.init_deeplabv2_state(seed=seed)
    model = ref['model'].DeepLabV2(args)
    model.load_state_dict(with_prefix(state, "model."))
    model.train()
    x, gt = TO.synthetic_batch(batch, size, batch, seed=seed + 1, block=16)
    resulter, _ = model.forward((x,))
    logits = resulter['pred'][0]
    prob = resulter['activated_pred'][0]
    crit = ref['criterion'].CommonSSEGCriterion(args)
    per_sample = crit.forward((logits,), (gt,), (x,))
    loss = per_sample.mean()
    loss.backward()
    ref_sd = OrderedDict((k[len("model."):], v) for k, v in model.state_dict().items())
    g_conv1 = model.model.backbone.conv1.weight.grad
    g_aspp = model.model.classifier.conv2d_list[1].weight.grad
    g_l3 = model.model.backbone.layer3[5].conv2.weight.grad

    # ---- restatement must agree
    o_state = TO.clone_state(state)
    leaves = TO._param_leaves(o_state)
    run = TO._with_leaves(o_state, leaves)
    o_logits, o_prob, o_lat, o_low = TO.deeplabv2_forward(run, x, train=True)
    o_ps = TO.sseg_criterion(o_logits, gt)
    o_ps.mean().backward()
    print("case forward:")
    check("logits", o_logits, logits)
    check("softmax", o_prob, prob)
    check("latent", o_lat, resulter['sslcct_ad_inp'])
    check("per-sample CE", o_ps, per_sample)
    check("grad conv1", leaves["backbone.conv1.weight"].grad, g_conv1, rtol=1e-4)
    check("grad aspp1", leaves["classifier.conv2d_list.1.weight"].grad, g_aspp, rtol=1e-4)
    check("grad layer3.5.conv2", leaves["backbone.layer3.5.conv2.weight"].grad, g_l3, rtol=1e-4)
    check("running_mean bn1", run["backbone.bn1.running_mean"], ref_sd["backbone.bn1.running_mean"])
    check("running_var l4", run["backbone.layer4.0.downsample.1.running_var"],
          ref_sd["backbone.layer4.0.downsample.1.running_var"])

    fx = dict(kind="forward", size=size, batch=batch, weight_seed=seed, data_seed=seed + 1,
              block=16, logits=logits.detach().clone(), low=o_low.detach().clone(),
              argmax=logits.argmax(1).to(torch.uint8), per_sample=per_sample.detach().clone(),
              latent_sum=float(resulter['sslcct_ad_inp'].double().sum()),
              latent_head=resulter['sslcct_ad_inp'].detach().reshape(-1)[:256].clone(),
              grad_conv1=g_conv1.clone(), grad_aspp1_head=g_aspp.reshape(-1)[:512].clone(),
              grad_l3_head=g_l3.reshape(-1)[:512].clone(),
              grad_l3_abssum=float(g_l3.double().abs().sum()),
              probes=probe(ref_sd))
    torch.save(fx, os.path.join(OUT, "deeplabv2_forward_%d.pt" % size))


class _ListLoader(list):
    pass


def _build_algo(name, args):
    ref = ref_shim.load_reference()
    pixelssl = ref['pixelssl']
    from pixelssl.nn import optimizer as ropt, lrer as rlr
    model_dict = {'model': ref['model'].DeepLabV2}
    crit_dict = {'model': ref['criterion'].CommonSSEGCriterion}
    opt_dict = {'model': ropt.sgd(args)}
    lr_dict = {'model': rlr.polynomiallr(args)}
    task_func = ref['func'].task_func()(args)
    export = pixelssl.ssl_algorithm.__dict__[name].__dict__[name]
    return export(args, model_dict, opt_dict, lr_dict, crit_dict, task_func)


def case_suponly(size=65, batch=2, seed=21, iters=2, gamma3=None, out=None, block=16):
    """Reference SSLNULL._train for `iters` iterations (ssl_null.py:78-144).  gamma3: conditioned initial weights
    (torch_oracle.condition_state) for the multi-step fixtures."""
    args = ref_shim.make_args('ssl_null', dict(BASE_CFG, batch_size=batch,
                                               unlabeled_batch_size=0, im_size=size,
                                               ignore_unlabeled=True))
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo('ssl_null', args)
    state = TO.init_deeplabv2_state(seed=seed)
    if gamma3 is not None:
        TO.condition_state(state, gamma3)
    algo.model.module.load_state_dict(with_prefix(state, "model."))
    batches = [TO.synthetic_batch(batch, size, batch, seed=seed + 10 + i, block=block)
               for i in range(iters)]
    # run the reference's own loop over the whole list (the real code path)
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])
    seen = record_meters(algo)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, ('task_loss',), iters)
    ref_sd = OrderedDict((k[len("module.model."):], v)
                         for k, v in algo.model.state_dict().items())
    ref_avg_loss = float(algo.meters['task_loss'].avg)

    tr = TO.OracleTrainer(TO.clone_state(state), dict(max_iters=args.epochs * args.iters_per_epoch))
    o_losses = [tr.suponly_step(x, gt)["task_loss"] for x, gt in batches]
    print("case suponly%s:" % ("" if gamma3 is None else " (conditioned, gamma3 = %g)" % gamma3))
    check("mean task loss", sum(o_losses) / len(o_losses), ref_avg_loss)
    for i in range(iters):
        check("iter %d task loss" % i, o_losses[i], ref_iters[i]['task_loss'], rtol=2e-5 if gamma3 is None else 2e-6)
    for k in PROBES:
        check("post-step " + k, tr.sd[k], ref_sd[k], rtol=2e-5)
    fx = dict(kind="suponly", size=size, batch=batch, weight_seed=seed, gamma3=gamma3,
              data_seeds=[seed + 10 + i for i in range(iters)], block=block,
              max_iters=args.epochs * args.iters_per_epoch,
              mean_task_loss=ref_avg_loss, oracle_losses=o_losses, per_iter=ref_iters, probes=probe(ref_sd),
              updates=probe_update(ref_sd, state, PROBES))
    torch.save(fx, os.path.join(OUT, out or "suponly_%d.pt" % size))


def case_mt(size=65, lbs=2, ubs=2, seed=31, iters=2, gamma3=None, out=None, block=16):
    """Reference SSLMT._train (ssl_mt.py:124-224) with the shipped MT script's
    hyper-parameters (deeplabv2_pascalvoc_1-8_sslmt.py:23-28)."""
    batch = lbs + ubs
    args = ref_shim.make_args('ssl_mt', dict(BASE_CFG, batch_size=batch,
                                             unlabeled_batch_size=ubs, im_size=size,
                                             ignore_unlabeled=False, cons_for_labeled=False,
                                             cons_scale=1.0, cons_rampup_epochs=3,
                                             ema_decay=0.99))
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo('ssl_mt', args)
    s_state = TO.init_deeplabv2_state(seed=seed)
    t_state = TO.init_deeplabv2_state(seed=seed + 1)
    if gamma3 is not None:
        TO.condition_state(s_state, gamma3)
        TO.condition_state(t_state, gamma3)
    algo.s_model.module.load_state_dict(with_prefix(s_state, "model."))
    algo.t_model.module.load_state_dict(with_prefix(t_state, "model."))
    batches = [TO.synthetic_batch(batch, size, lbs, seed=seed + 10 + i, block=block)
               for i in range(iters)]
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])
    seen = record_meters(algo)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, ('s_task_loss', 't_task_loss', 'cons_loss'), iters)
    strip = lambda sd: OrderedDict((k[len("module.model."):], v) for k, v in sd.items())
    ref_s, ref_t = strip(algo.s_model.state_dict()), strip(algo.t_model.state_dict())
    meters = {k: float(algo.meters[k].avg) for k in ('s_task_loss', 't_task_loss', 'cons_loss')}

    tr = TO.OracleTrainer(TO.clone_state(s_state),
                          dict(max_iters=args.epochs * args.iters_per_epoch,
                               cons_scale=1.0, cons_rampup_iters=len(loader) * 3,
                               cons_for_labeled=False, ema_decay=0.99),
                          teacher_state=TO.clone_state(t_state))
    outs = [tr.mt_step(x, gt, lbs) for x, gt in batches]
    print("case mt%s:" % ("" if gamma3 is None else " (conditioned, gamma3 = %g)" % gamma3))
    for k in meters:
        check("mean " + k, sum(o[k] for o in outs) / len(outs), meters[k])
        for i in range(iters):
            check("iter %d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=2e-5 if gamma3 is None else 2e-6, atol=1e-9)
    for k in PROBES:
        check("student " + k, tr.sd[k], ref_s[k], rtol=2e-5)
        check("teacher " + k, tr.t_sd[k], ref_t[k], rtol=2e-5)
    fx = dict(kind="mt", size=size, lbs=lbs, ubs=ubs, weight_seed=seed, gamma3=gamma3,
              data_seeds=[seed + 10 + i for i in range(iters)], block=block,
              max_iters=args.epochs * args.iters_per_epoch, rampup_iters=len(loader) * 3,
              meters=meters, per_iter=[{k: o[k] for k in meters} for o in outs], ref_per_iter=ref_iters,
              student_probes=probe(ref_s), teacher_probes=probe(ref_t),
              student_updates=probe_update(ref_s, s_state, PROBES), teacher_updates=probe_update(ref_t, t_state, PROBES))
    torch.save(fx, os.path.join(OUT, out or "mt_%d.pt" % size))


if __name__ == "__main__":
    if not ref_shim.reference_available():
        raise SystemExit("reference tree not available; fixtures can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    case_forward()
    case_suponly()
    case_mt()
    print("golden fixtures written to", OUT)
