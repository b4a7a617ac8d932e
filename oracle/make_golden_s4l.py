"""Pin oracle/s4l_oracle.py against the REAL reference (container only) and write tests/golden/s4l_65.pt (reference
initialisers, 2 iterations) and tests/golden/s4l_cond_65.pt (conditioned weights, 4 iterations).
TEST INFRASTRUCTURE.   python oracle/make_golden_s4l.py"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim                 # noqa: E402
import torch_oracle as TO       # noqa: E402
import s4l_oracle as SO         # noqa: E402
from make_golden import (BASE_CFG, PROBES, _ListLoader, _build_algo, check, with_prefix, probe, record_meters,   # noqa: E402
                         per_iteration, probe_update)

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")
KEYS = ("unrotated_task_loss", "rotated_task_loss", "rotation_loss", "rotation_acc")


def main(size=65, lbs=2, ubs=2, seed=61, iters=2, gamma3=None, out_name="s4l_65.pt", block=16, np_seed=7):
    ref_shim.load_reference()
    from pixelssl.ssl_algorithm import ssl_s4l as R
    torch.set_num_threads(8)
    # ---- stand-alone rotation classifier: same default init under the same seed, same forward / backward
    rc0 = SO.init_rc_state(21, seed=seed + 3)
    torch.manual_seed(seed + 3)
    ref_rc = R.RotationClassifer(21)
    for k, v in ref_rc.state_dict().items():
        assert torch.equal(v, rc0[k]), k
    g = torch.Generator().manual_seed(seed)
    pred = (torch.randn(4, 21, size, size, generator=g) * 2).requires_grad_(True)
    ref_rc.train()
    out_ref = ref_rc(pred)
    tgt = torch.tensor([0, 1, 2, 3])
    torch.nn.CrossEntropyLoss()(out_ref, tgt).backward()
    pred2 = pred.detach().clone().requires_grad_(True)
    leaves = OrderedDict((k, v.clone().requires_grad_(not SO.rc_is_buffer(k))) for k, v in rc0.items())
    out = SO.rc_forward(leaves, pred2, train=True)
    torch.nn.functional.cross_entropy(out, tgt).backward()
    print("stand-alone rotation classifier:")
    check("rotation logits", out, out_ref)
    check("d loss / d pred", pred2.grad, pred.grad, rtol=1e-4)
    check("d loss / d conv1.weight", leaves["conv1.weight"].grad, ref_rc.conv1.weight.grad, rtol=1e-4)
    check("d loss / d bn2.weight", leaves["bn2.weight"].grad, ref_rc.bn2.weight.grad, rtol=1e-4)
    check("d loss / d classifier.bias", leaves["classifier.bias"].grad, ref_rc.classifier.bias.grad, rtol=1e-4)
    check("bn1.running_var", leaves["bn1.running_var"], ref_rc.bn1.running_var)
    for a in (1, 2, 3):                     # the three rotations against the reference's method
        t = torch.randn(3, 5, 5, generator=g)
        assert torch.equal(SO.rotate(t, a), R.SSLS4L._rotate_tensor(None, t, a))
    standalone = dict(seed=seed, logits=out_ref.detach().clone(), dpred_head=pred.grad[:, :, :4, :8].clone(),
                      dpred_abssum=float(pred.grad.double().abs().sum()),
                      dconv1=ref_rc.conv1.weight.grad.reshape(-1)[:256].clone(), dbn2=ref_rc.bn2.weight.grad.clone(),
                      dcls_bias=ref_rc.classifier.bias.grad.clone(), bn1_rv=ref_rc.bn1.running_var.clone())

    # ---- the reference's own training loop
    bs = lbs + ubs
    args = ref_shim.make_args("ssl_s4l", dict(BASE_CFG, batch_size=bs, unlabeled_batch_size=ubs, im_size=size,
                                              ignore_unlabeled=False, rotated_sup_scale=0.5, rotation_scale=0.1))
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo("ssl_s4l", args)
    assert args.batch_size == 2 * bs and args.labeled_batch_size == 2 * lbs          # doubled by SSLS4L._build
    state = TO.init_deeplabv2_state(seed=seed)
    if gamma3 is not None:
        TO.condition_state(state, gamma3)
    algo.model.module.task_model.load_state_dict(with_prefix(state, "model."))
    rc_init = SO.init_rc_state(21, seed=seed + 4)
    algo.model.module.rotation_classifier.load_state_dict(rc_init)
    names = [k for k, _ in algo.model.module.named_parameters()]
    assert names[0] == "task_model.model.backbone.conv1.weight" and names[-1] == "rotation_classifier.classifier.bias"
    assert [len(g_["params"]) for g_ in algo.optimizer.param_groups][-1] == 10        # rc: 2 convs, 2 BNs, 1 linear (w + b)
    batches = [TO.synthetic_batch(bs, size, lbs, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt_,)) for x, gt_ in batches])
    seen = record_meters(algo)
    np.random.seed(np_seed)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, KEYS, iters)
    meters = {k: float(algo.meters[k].avg) for k in KEYS}
    ref_sd = OrderedDict((k[len("module.task_model.model."):], v) for k, v in algo.model.state_dict().items()
                         if k.startswith("module.task_model.model."))
    ref_rc_sd = OrderedDict((k[len("module.rotation_classifier."):], v.clone()) for k, v in algo.model.state_dict().items()
                            if k.startswith("module.rotation_classifier."))

    tr = SO.S4LOracleTrainer(TO.clone_state(state), OrderedDict((k, v.clone()) for k, v in rc_init.items()),
                             dict(max_iters=args.epochs * args.iters_per_epoch, rotated_sup_scale=0.5, rotation_scale=0.1))
    np.random.seed(np_seed)
    angles = [SO.draw_angles(bs) for _ in range(iters)]
    outs = [tr.s4l_step(x, gt_, lbs, a) for (x, gt_), a in zip(batches, angles)]
    print("SSLS4L._train%s:" % ("" if gamma3 is None else " (conditioned, gamma3 = %g)" % gamma3))
    tol = 2e-5 if gamma3 is None else 1e-4
    for k in KEYS:
        for i in range(iters):
            check("iter %d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=tol, atol=1e-9)
    for k in ("backbone.conv1.weight", "backbone.layer3.11.conv3.weight", "classifier.conv2d_list.0.weight"):
        check("task " + k, tr.sd[k], ref_sd[k], rtol=5 * tol)
    for k in ref_rc_sd:
        if not k.endswith("num_batches_tracked"):
            check("rotation classifier " + k, tr.rc[k], ref_rc_sd[k], rtol=5 * tol, atol=2e-7)
    fp = os.path.join(GOLD, out_name)
    torch.save(dict(kind="s4l", size=size, lbs=lbs, ubs=ubs, weight_seed=seed, rc_seed=seed + 4, gamma3=gamma3, np_seed=np_seed,
                    angles=[a.tolist() for a in angles], data_seeds=[seed + 10 + i for i in range(iters)], block=block,
                    max_iters=args.epochs * args.iters_per_epoch, rotated_sup_scale=0.5, rotation_scale=0.1,
                    meters=meters, ref_per_iter=ref_iters, per_iter=[{k: o[k] for k in KEYS} for o in outs],
                    pred_rotation0=outs[0]["pred_rotation"], probes=probe(ref_sd), updates=probe_update(ref_sd, state, PROBES),
                    rc_updates=probe_update(ref_rc_sd, rc_init, [k for k in ref_rc_sd if not SO.rc_is_buffer(k)]),
                    rc_after={k: v.clone() for k, v in ref_rc_sd.items()}, standalone=standalone,
                    param_names=names[-10:]), fp)
    print("wrote", fp, os.path.getsize(fp), "bytes; oracle == reference")


if __name__ == "__main__":
    main()
    main(iters=4, gamma3=0.1, out_name="s4l_cond_65.pt")
