"""Golden fixtures for the PSPNet rows (M4) from the REAL reference imported from /root/reference.
Container-only; TEST INFRASTRUCTURE.

    python oracle/make_golden_psp.py        # writes tests/golden/pspnet_*.pt

(1) runs the reference's own `PSPNet` TaskModel / `SSLNULL._train` on seeded inputs with weights from
`torch_oracle.init_pspnet_state(seed)`, (2) asserts that `torch_oracle.pspnet_forward` & the oracle trainer
reproduce them, (3) stores the reference outputs for tests/test_psp.py.
"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim            # noqa: E402
import torch_oracle as TO  # noqa: E402
from make_golden import check, with_prefix, _ListLoader, record_meters, per_iteration, probe_update  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

BASE_CFG = dict(models={'model': 'pspnet'}, optimizers={'model': 'sgd'},
                lrers={'model': 'polynomiallr'}, criterions={'model': 'sseg_criterion'},
                lr=0.00025, momentum=0.9, weight_decay=0.0005, output_stride=16,
                backbone='resnet101', epochs=1, log_freq=1000)

PROBES = ["backbone.conv1.weight", "backbone.layer3.11.conv3.weight", "backbone.layer4.2.conv2.weight",
          "backbone.layer4.2.bn3.bias", "psp.stages.0.1.weight", "psp.stages.3.1.weight", "psp.stages.1.2.weight",
          "psp.stages.2.2.running_mean", "psp.bottleneck.0.weight", "psp.bottleneck.1.bias",
          "psp.bottleneck.1.running_var", "decoder.0.weight", "decoder.1.conv.weight", "decoder.2.conv.bias",
          "decoder.3.conv.weight", "decoder.3.conv.bias"]


def probe(sd):
    out = OrderedDict()
    for k in PROBES:
        v = sd[k].detach().float().reshape(-1)
        out[k] = dict(head=v[:64].clone(), sum=float(v.double().sum()), abssum=float(v.double().abs().sum()))
    return out


def case_forward(size=65, batch=2, seed=51):
    ref = ref_shim.load_reference()
    args = ref_shim.make_args('ssl_null', dict(BASE_CFG, batch_size=batch, unlabeled_batch_size=0, im_size=size))
    state = TO.init_pspnet_state(seed=seed)
    model = ref['model'].PSPNet(args)
    model.load_state_dict(with_prefix(state, "model."))
    model.train()
    x, gt = TO.synthetic_batch(batch, size, batch, seed=seed + 1, block=16)
    resulter, _ = model.forward((x,))
    logits = resulter['pred'][0]
    prob = resulter['activated_pred'][0]
    latent = resulter['sslcct_ad_inp']
    crit = ref['criterion'].CommonSSEGCriterion(args)
    per_sample = crit.forward((logits,), (gt,), (x,))
    per_sample.mean().backward()
    ref_sd = OrderedDict((k[len("model."):], v) for k, v in model.state_dict().items())
    m = model.model
    ref_grads = {"backbone.conv1.weight": m.backbone.conv1.weight.grad,
                 "backbone.layer4.2.conv3.weight": m.backbone.layer4[2].conv3.weight.grad,
                 "psp.stages.0.1.weight": m.psp.stages[0][1].weight.grad,
                 "psp.stages.3.1.weight": m.psp.stages[3][1].weight.grad,
                 "psp.stages.2.2.weight": m.psp.stages[2][2].weight.grad,
                 "psp.bottleneck.0.weight": m.psp.bottleneck[0].weight.grad,
                 "psp.bottleneck.1.bias": m.psp.bottleneck[1].bias.grad,
                 "decoder.0.weight": m.decoder[0].weight.grad,
                 "decoder.1.conv.weight": m.decoder[1].conv.weight.grad,
                 "decoder.3.conv.bias": m.decoder[3].conv.bias.grad}

    o_state = TO.clone_state(state)
    leaves = TO._param_leaves(o_state)
    run = TO._with_leaves(o_state, leaves)
    o_logits, o_prob, o_lat, o_low = TO.pspnet_forward(run, x, train=True)
    o_ps = TO.sseg_criterion(o_logits, gt)
    o_ps.mean().backward()
    print("case pspnet forward:")
    check("logits", o_logits, logits)
    check("softmax", o_prob, prob)
    check("latent", o_lat, latent)
    check("per-sample CE", o_ps, per_sample)
    for k, g in ref_grads.items():
        check("grad " + k, leaves[k].grad, g, rtol=1e-4)
    for k in ("psp.stages.2.2.running_mean", "psp.bottleneck.1.running_var"):
        check(k, run[k], ref_sd[k])

    fx = dict(kind="pspnet_forward", size=size, batch=batch, weight_seed=seed, data_seed=seed + 1, block=16,
              logits=logits.detach().clone(), low=o_low.detach().clone(),
              per_sample=per_sample.detach().clone(),
              latent_sum=float(latent.double().sum()), latent_abssum=float(latent.double().abs().sum()),
              latent_head=latent.detach().reshape(-1)[:256].clone(),
              grads={k: dict(head=g.reshape(-1)[:512].clone(), abssum=float(g.double().abs().sum()))
                     for k, g in ref_grads.items()},
              probes=probe(ref_sd))
    torch.save(fx, os.path.join(OUT, "pspnet_forward_%d.pt" % size))


def _build_algo(name, args):
    ref = ref_shim.load_reference()
    pixelssl = ref['pixelssl']
    from pixelssl.nn import optimizer as ropt, lrer as rlr
    model_dict = {'model': ref['model'].PSPNet}
    crit_dict = {'model': ref['criterion'].CommonSSEGCriterion}
    opt_dict = {'model': ropt.sgd(args)}
    lr_dict = {'model': rlr.polynomiallr(args)}
    task_func = ref['func'].task_func()(args)
    export = pixelssl.ssl_algorithm.__dict__[name].__dict__[name]
    return export(args, model_dict, opt_dict, lr_dict, crit_dict, task_func)


def case_suponly(size=65, batch=2, seed=61, iters=2, gamma3=None, out=None, block=16):
    """Reference SSLNULL._train on PSPNet (3 parameter groups: backbone lr, psp / decoder lr x10)."""
    args = ref_shim.make_args('ssl_null', dict(BASE_CFG, batch_size=batch, unlabeled_batch_size=0, im_size=size,
                                               ignore_unlabeled=True))
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo('ssl_null', args)
    state = TO.init_pspnet_state(seed=seed)
    if gamma3 is not None:
        TO.condition_state(state, gamma3)
    algo.model.module.load_state_dict(with_prefix(state, "model."))
    batches = [TO.synthetic_batch(batch, size, batch, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])
    seen = record_meters(algo)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, ('task_loss',), iters)
    ref_sd = OrderedDict((k[len("module.model."):], v) for k, v in algo.model.state_dict().items())
    ref_avg_loss = float(algo.meters['task_loss'].avg)

    tr = TO.OracleTrainer(TO.clone_state(state), dict(max_iters=args.epochs * args.iters_per_epoch),
                          forward=TO.pspnet_forward)
    o_losses = [tr.suponly_step(x, gt)["task_loss"] for x, gt in batches]
    print("case pspnet suponly:")
    check("mean task loss", sum(o_losses) / len(o_losses), ref_avg_loss)
    for i in range(iters):
        check("iter %d task loss" % i, o_losses[i], ref_iters[i]['task_loss'], rtol=2e-5 if gamma3 is None else 2e-6)
    for k in PROBES:
        check("post-step " + k, tr.sd[k], ref_sd[k], rtol=2e-5)
    fx = dict(kind="pspnet_suponly", size=size, batch=batch, weight_seed=seed, gamma3=gamma3,
              data_seeds=[seed + 10 + i for i in range(iters)], block=block,
              max_iters=args.epochs * args.iters_per_epoch,
              mean_task_loss=ref_avg_loss, oracle_losses=o_losses, per_iter=ref_iters, probes=probe(ref_sd),
              updates=probe_update(ref_sd, state, PROBES))
    torch.save(fx, os.path.join(OUT, out or "pspnet_suponly_%d.pt" % size))


if __name__ == "__main__":
    if not ref_shim.reference_available():
        raise SystemExit("reference tree not available; fixtures can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    case_forward()
    case_suponly()
    print("golden fixtures written to", OUT)
