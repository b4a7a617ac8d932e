"""Pin the FlawDetector / SSLGCT part of oracle/gct_oracle.py against the REAL reference (container only) and write
tests/golden/gct_129.pt.  TEST INFRASTRUCTURE.   python oracle/make_golden_gct_train.py"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim                 # noqa: E402
import torch_oracle as TO       # noqa: E402
import gct_oracle as GO         # noqa: E402
from make_golden import BASE_CFG, PROBES, _ListLoader, check, with_prefix, probe, probe_update   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "gct_129.pt")


def main(size=129, lbs=2, ubs=2, seed=61, iters=2, gamma3=None, out=None, block=16):
    ref = ref_shim.load_reference()
    pixelssl = ref["pixelssl"]
    from pixelssl.ssl_algorithm import ssl_gct as R
    from pixelssl.nn import optimizer as ropt, lrer as rlr
    torch.set_num_threads(8)
    # ---- stand-alone flaw detector: forward + gradients w.r.t. the softmax input and parameters
    fd0 = GO.init_fd_state(24, seed=seed + 5)
    torch.manual_seed(seed + 5)
    ref_fd = R.FlawDetector(24)
    for k, v in ref_fd.state_dict().items():
        assert torch.equal(v, fd0[k]), k
    ref_fd.train()
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(3, 3, size, size, generator=g)
    prob = torch.softmax(torch.randn(3, 21, size, size, generator=g), 1).requires_grad_(True)
    fm_ref = ref_fd((img,), prob)[0]["flawmap"]
    (fm_ref ** 2).mean().backward()
    sd = OrderedDict((k, v.clone()) for k, v in fd0.items())
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in fd0.items() if not GO.fd_is_buffer(k))
    run = OrderedDict(sd)
    run.update(leaves)
    prob2 = prob.detach().clone().requires_grad_(True)
    fm = GO.fd_forward(run, img, prob2, train=True)
    (fm ** 2).mean().backward()
    print("stand-alone flaw detector:")
    check("flawmap", fm, fm_ref)
    # (weight gradients are fp32 sums over 3 * size^2 pixels: the oracle's conv backward and the reference's agree to the
    # summation-order noise of that length -- 1e-4 at 129^2, 1e-3 at 513^2)
    wtol = 1e-4 if size <= 129 else 2e-3
    check("d/d prob", prob2.grad, prob.grad, rtol=1e-4)
    check("d/d conv1.weight", leaves["conv1.weight"].grad, ref_fd.conv1.weight.grad, rtol=wtol)
    check("d/d ibn3.bnorm.weight", leaves["ibn3.bnorm.weight"].grad, ref_fd.ibn3.bnorm.weight.grad, rtol=wtol)
    check("running_var ibn1", run["ibn1.bnorm.running_var"], ref_fd.ibn1.bnorm.running_var)
    standalone = dict(seed=seed, flawmap=fm_ref.detach().clone(), dprob_head=prob.grad[:, :, :4, :8].clone(),
                      dprob_abssum=float(prob.grad.double().abs().sum()),
                      dconv1_head=ref_fd.conv1.weight.grad.reshape(-1)[:256].clone(),
                      dibn3_gamma=ref_fd.ibn3.bnorm.weight.grad.clone(),
                      dcls_bias=ref_fd.classifier.bias.grad.clone(),
                      rvar_ibn1=ref_fd.ibn1.bnorm.running_var.clone())

    # ---- the reference's own training loop (two task models from ONE 'model' entry, ssl_gct.py:58-66)
    batch = lbs + ubs
    args = ref_shim.make_args("ssl_gct", dict(BASE_CFG, batch_size=batch, unlabeled_batch_size=ubs, im_size=size,
                                              ignore_unlabeled=False, ssl_mode="gct", fc_ssl_scale=1.0, dc_ssl_scale=100.0,
                                              dc_threshold=0.6, dc_rampup_epochs=3, fd_lr=1e-4, fd_scale=10.0, mu=0.5, nu=1))
    args.iters_per_epoch = max(4, iters + 2)
    model_dict = {"model": ref["model"].DeepLabV2}
    crit_dict = {"model": ref["criterion"].CommonSSEGCriterion}
    task_func = ref["func"].task_func()(args)
    algo = pixelssl.ssl_algorithm.ssl_gct.ssl_gct(args, model_dict, {"model": ropt.sgd(args)},
                                                  {"model": rlr.polynomiallr(args)}, crit_dict, task_func)
    l_state, r_state = TO.init_deeplabv2_state(seed=seed), TO.init_deeplabv2_state(seed=seed + 1)
    if gamma3 is not None:
        TO.condition_state(l_state, gamma3)
        TO.condition_state(r_state, gamma3)
    fd1 = GO.init_fd_state(24, seed=seed + 6)
    # a freshly initialised detector outputs ~0 everywhere, FlawmapHandler then zeroes both maps and neither the
    # flaw-correction nor the right-hand consistency path is exercised: start from a detector with a lively output
    # (and the fixture runs at 129 x 129: at 65 x 65 the 8-conv stack ends on a 1 x 1 map, i.e. a constant flaw map)
    fd1["classifier.weight"] = fd1["classifier.weight"] * -3.0
    fd1["classifier.bias"] = fd1["classifier.bias"] * -3.0
    algo.l_model.module.load_state_dict(with_prefix(l_state, "model."))
    algo.r_model.module.load_state_dict(with_prefix(r_state, "model."))
    algo.fd_model.module.load_state_dict(fd1)
    batches = [TO.synthetic_batch(batch, size, lbs, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])
    keys = ("l_task_loss", "l_fc_loss", "l_dc_loss", "r_task_loss", "r_fc_loss", "r_dc_loss", "l_fd_loss", "r_fd_loss")
    seen = []                                   # every value the reference logs, in order -> per-iteration losses
    real_update = algo.meters.update
    algo.meters.update = lambda k, v, *a: (seen.append((k, float(v))), real_update(k, v, *a))[1]
    algo._train(loader, 0)
    ref_iters = [{} for _ in range(iters)]
    for k in keys:
        vals = [v for kk, v in seen if kk == k]
        assert len(vals) == iters, (k, len(vals))
        for i, v in enumerate(vals):
            ref_iters[i][k] = v
    meters = {k: float(algo.meters[k].avg) for k in keys}
    strip = lambda sdd: OrderedDict((k[len("module.model."):], v) for k, v in sdd.items())
    ref_l, ref_r = strip(algo.l_model.state_dict()), strip(algo.r_model.state_dict())
    ref_fdsd = OrderedDict((k[len("module."):], v.clone()) for k, v in algo.fd_model.state_dict().items())

    tr = GO.GCTOracleTrainer(TO.clone_state(l_state), TO.clone_state(r_state), OrderedDict((k, v.clone()) for k, v in fd1.items()),
                             dict(im_size=size, dc_rampup_iters=len(loader) * 3, max_iters=args.epochs * args.iters_per_epoch))
    outs = [tr.gct_step(x, gt, lbs) for x, gt in batches]
    print("SSLGCT._train:")
    # Iteration 0 is reproduced to rounding.  Later iterations are not reproducible by ANYONE: the reference run with 3
    # instead of 8 CPU threads already moves r_model's stem weights by 7e-5 (7e-4 relative) after ONE iteration, and
    # Adam turns near-zero flaw-detector gradients into +-lr steps of either sign; both feed the thresholded
    # (flaw map > 0.6) losses of iteration 1, which then differ by 10-20 %.  Hence: iteration 0 strict, the rest a band.
    for k in keys:
        check("iter0 " + k, outs[0][k], ref_iters[0][k], rtol=2e-5)
    for i in range(1, iters):
        for k in keys:
            if gamma3 is None:
                check("iter%d %s (band)" % (i, k), outs[i][k], ref_iters[i][k], rtol=0.35, atol=1e-3)
            else:       # conditioned task models: every iteration is reproducible -- to the 1e-6 difference between
                # this restatement's flaw detector and the reference's (measured above), which Adam's sign-like first
                # steps and the flaw-map threshold amplify to <= 2e-3 (task / fc / fd) and <= 5e-3 (dc) over six iterations
                check("iter%d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=1e-2 if "dc" in k else 3e-3, atol=1e-8)
    for k in ("backbone.layer3.11.conv3.weight", "classifier.conv2d_list.0.weight"):
        check("l " + k, tr.l.sd[k], ref_l[k], rtol=2e-3 if gamma3 is None else 5e-4)
        check("r " + k, tr.r.sd[k], ref_r[k], rtol=2e-3 if gamma3 is None else 5e-4)
    # (the stem weights are not compared: after two iterations the reference itself moves them by >10 % between
    # a 3-thread and an 8-thread run of the same code)
    fdsd = tr.fd_state()
    for k in ("ibn2.bnorm.weight", "ibn4.bnorm.running_mean", "classifier.weight", "classifier.bias"):
        check("fd " + k, fdsd[k], ref_fdsd[k], rtol=2e-2, atol=5e-7)
    torch.save(dict(kind="gct", size=size, lbs=lbs, ubs=ubs, weight_seed=seed, fd_seed=seed + 6, gamma3=gamma3,
                    l_updates=probe_update(ref_l, l_state, PROBES), r_updates=probe_update(ref_r, r_state, PROBES),
                    fd_updates=probe_update(ref_fdsd, fd1, [k for k, v in ref_fdsd.items() if v.is_floating_point()]),
                    data_seeds=[seed + 10 + i for i in range(iters)], block=block,
                    max_iters=args.epochs * args.iters_per_epoch, rampup_iters=len(loader) * 3, meters=meters,
                    per_iter=ref_iters, oracle_per_iter=outs, l_probes=probe(ref_l), r_probes=probe(ref_r),
                    fd_after={k: dict(head=v.reshape(-1)[:64].float().clone(), sum=float(v.double().sum()))
                              for k, v in ref_fdsd.items() if v.is_floating_point() and not
                              (k.endswith(".bias") and k.startswith("conv"))},
                    fd_scale_classifier=-3.0,
                    standalone=standalone), OUT if out is None else os.path.join(os.path.dirname(OUT), out))
    print("wrote", out or OUT, os.path.getsize(OUT), "bytes; oracle == reference")


if __name__ == "__main__":
    main()
