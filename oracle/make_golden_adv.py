"""Pin oracle/adv_oracle.py against the REAL reference (container only) and write tests/golden/adv_65.pt.
TEST INFRASTRUCTURE.   python oracle/make_golden_adv.py"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim                 # noqa: E402
import torch_oracle as TO       # noqa: E402
import adv_oracle as AO         # noqa: E402
from make_golden import (BASE_CFG, PROBES, _ListLoader, _build_algo, check, with_prefix, probe, record_meters,   # noqa: E402
                         per_iteration, probe_update)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "adv_65.pt")


def main(size=65, lbs=2, ubs=2, seed=41, iters=2, gamma3=None, out=None, block=16):
    ref = ref_shim.load_reference()
    from pixelssl.ssl_algorithm import ssl_adv as R
    torch.set_num_threads(8)
    # ---- stand-alone modules
    d_state = AO.init_fcd_state(21, seed=seed + 5)
    torch.manual_seed(seed + 5)
    ref_d = R.FCDiscriminator(21)
    for k, v in ref_d.state_dict().items():
        assert torch.equal(v, d_state[k]), k                  # same default init under the same seed
    g = torch.Generator().manual_seed(seed)
    prob = torch.softmax(torch.randn(3, 21, size, size, generator=g), 1).requires_grad_(True)
    _, gt = TO.synthetic_batch(3, size, 3, seed=seed + 1, block=16)
    conf_ref = ref_d(prob)[0]["confidence"]
    task_func = ref["func"].task_func()(ref_shim.make_args("ssl_adv", dict(BASE_CFG, batch_size=4, unlabeled_batch_size=2,
                                                                          im_size=size)))
    p_ref, g_ref = task_func.ssladv_preprocess_fcd_criterion(conf_ref, gt, True)
    loss_ref = R.FCDiscriminatorCriterion()(p_ref, g_ref)
    loss_ref.mean().backward()
    prob2 = prob.detach().clone().requires_grad_(True)
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in d_state.items())
    conf = AO.fcd_forward(leaves, prob2)
    p, gg = AO.preprocess_fcd_criterion(conf, gt, True)
    loss = AO.fcd_criterion(p, gg)
    loss.mean().backward()
    print("stand-alone:")
    check("confidence map", conf, conf_ref)
    check("masked BCE per sample", loss, loss_ref)
    check("d loss / d prob", prob2.grad, prob.grad, rtol=1e-4)
    check("d loss / d conv1.weight", leaves["conv1.weight"].grad, ref_d.conv1.weight.grad, rtol=1e-4)
    check("d loss / d classifier.bias", leaves["classifier.bias"].grad, ref_d.classifier.bias.grad, rtol=1e-4)
    pf_ref, gf_ref = task_func.ssladv_preprocess_fcd_criterion(conf_ref.detach(), None, False)
    pf, gf = AO.preprocess_fcd_criterion(conf.detach(), None, False)
    assert torch.equal(pf, pf_ref) and torch.equal(gf, gf_ref)
    assert torch.equal(AO.convert_task_gt_to_fcd_input(gt), task_func.ssladv_convert_task_gt_to_fcd_input(gt))
    standalone = dict(conf=conf_ref.detach().clone(), loss=loss_ref.detach().clone(), dprob_head=prob.grad[:, :, :4, :8].clone(),
                      dprob_abssum=float(prob.grad.double().abs().sum()),
                      dconv1_head=ref_d.conv1.weight.grad.reshape(-1)[:256].clone(),
                      dcls_bias=ref_d.classifier.bias.grad.clone(), seed=seed)

    # ---- the reference's own training loop
    batch = lbs + ubs
    args = ref_shim.make_args("ssl_adv", dict(BASE_CFG, batch_size=batch, unlabeled_batch_size=ubs, im_size=size,
                                              ignore_unlabeled=False, adv_for_labeled=True, labeled_adv_scale=0.01,
                                              unlabeled_adv_scale=0.001, discriminator_lr=1e-4,
                                              unlabeled_for_discriminator=True))
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo("ssl_adv", args)
    state = TO.init_deeplabv2_state(seed=seed)
    if gamma3 is not None:
        TO.condition_state(state, gamma3)
    algo.model.module.load_state_dict(with_prefix(state, "model."))
    d0 = AO.init_fcd_state(21, seed=seed + 6)
    algo.d_model.module.load_state_dict(d0)
    batches = [TO.synthetic_batch(batch, size, lbs, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt_,)) for x, gt_ in batches])
    keys = ("task_loss", "labeled_adv_loss", "unlabeled_adv_loss", "fake_d_loss", "real_d_loss")
    seen = record_meters(algo)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, keys, iters)
    meters = {k: float(algo.meters[k].avg) for k in keys}
    ref_sd = OrderedDict((k[len("module.model."):], v) for k, v in algo.model.state_dict().items())
    ref_dsd = OrderedDict((k[len("module."):], v.clone()) for k, v in algo.d_model.state_dict().items())

    tr = AO.AdvOracleTrainer(TO.clone_state(state), OrderedDict((k, v.clone()) for k, v in d0.items()),
                             dict(max_iters=args.epochs * args.iters_per_epoch))
    outs = [tr.adv_step(x, gt_, lbs) for x, gt_ in batches]
    print("SSLADV._train:")
    for k in keys:
        check("mean " + k, sum(o[k] for o in outs) / len(outs), meters[k], rtol=2e-5)
        for i in range(iters):
            check("iter %d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=2e-5, atol=1e-9)
    for k in ("backbone.conv1.weight", "backbone.layer3.11.conv3.weight", "classifier.conv2d_list.0.weight"):
        check("task " + k, tr.sd[k], ref_sd[k], rtol=2e-5)
    dsd = tr.d_state()
    for k in ref_dsd:
        check("discriminator " + k, dsd[k], ref_dsd[k], rtol=2e-5, atol=2e-7)
    torch.save(dict(kind="adv", size=size, lbs=lbs, ubs=ubs, weight_seed=seed, d_seed=seed + 6, gamma3=gamma3,
                    data_seeds=[seed + 10 + i for i in range(iters)], block=block,
                    max_iters=args.epochs * args.iters_per_epoch, meters=meters,
                    per_iter=[{k: o[k] for k in keys} for o in outs] if gamma3 is not None else outs, ref_per_iter=ref_iters,
                    probes=probe(ref_sd), updates=probe_update(ref_sd, state, PROBES),
                    d_updates=probe_update(ref_dsd, d0, list(ref_dsd.keys())),
                    d_after={k: dict(head=v.reshape(-1)[:64].clone(), sum=float(v.double().sum())) for k, v in ref_dsd.items()},
                    d_update={k: float((ref_dsd[k] - d0[k]).abs().max()) for k in ref_dsd},
                    standalone=standalone), OUT if out is None else os.path.join(os.path.dirname(OUT), out))
    print("wrote", out or OUT, os.path.getsize(OUT), "bytes; oracle == reference")


if __name__ == "__main__":
    main()
