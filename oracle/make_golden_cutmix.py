"""Pin oracle/cutmix_oracle.py against the REAL reference (container only) -> tests/golden/cutmix_65.pt.
TEST INFRASTRUCTURE.   python oracle/make_golden_cutmix.py"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim                 # noqa: E402
import torch_oracle as TO       # noqa: E402
import cutmix_oracle as CO      # noqa: E402
from make_golden import (BASE_CFG, PROBES, _ListLoader, _build_algo, check, with_prefix, probe, record_meters,   # noqa: E402
                         per_iteration, probe_update)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "cutmix_65.pt")


def main(size=65, lbs=2, ubs=4, seed=51, iters=2, np_seed=7, gamma3=None, out=None, block=16):
    ref_shim.load_reference()
    from pixelssl.ssl_algorithm import ssl_cutmix as R
    torch.set_num_threads(8)
    # mask generator: same RNG stream -> identical boxes
    np.random.seed(3)
    ref_masks = R.BoxMaskGenerator((0.5, 0.5), boxes_num=1, random_aspect_ratio=True, area_prop=True, within_bounds=True,
                                   invert=True).produce(5, (size, 97))
    mine = CO.box_masks(5, (size, 97), (0.5, 0.5), np.random.RandomState(3))
    assert np.array_equal(ref_masks, mine)
    np.random.seed(4)
    ref2 = R.BoxMaskGenerator((0.2, 0.7), 1, True, True, True, True).produce(3, (33, 33))
    assert np.array_equal(ref2, CO.box_masks(3, (33, 33), (0.2, 0.7), np.random.RandomState(4)))

    batch = lbs + ubs
    args = ref_shim.make_args("ssl_cutmix", dict(BASE_CFG, batch_size=batch, unlabeled_batch_size=ubs, im_size=size,
                                                 ignore_unlabeled=False, cons_type="mse", cons_scale=20.0,
                                                 cons_rampup_epochs=0, cons_threshold=0.25, ema_decay=0.99,
                                                 mask_prop_range=(0.5, 0.5)))
    # (cons_threshold: the script uses 0.97; with random-init weights nothing clears it, so the fixture uses a
    # threshold the ~1/21 softmax maxima of an untrained net straddle -> the confidence path is exercised)
    args.iters_per_epoch = max(4, iters + 2)
    algo = _build_algo("ssl_cutmix", args)
    s_state = TO.init_deeplabv2_state(seed=seed)
    t_state = TO.init_deeplabv2_state(seed=seed + 1)
    if gamma3 is not None:
        TO.condition_state(s_state, gamma3)
        TO.condition_state(t_state, gamma3)
    algo.s_model.module.load_state_dict(with_prefix(s_state, "model."))
    algo.t_model.module.load_state_dict(with_prefix(t_state, "model."))
    batches = [TO.synthetic_batch(batch, size, lbs, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])
    np.random.seed(np_seed)
    seen = record_meters(algo)
    algo._train(loader, 0)
    ref_iters = per_iteration(seen, ("task_loss", "cons_loss"), iters)
    meters = {k: float(algo.meters[k].avg) for k in ("task_loss", "cons_loss")}
    strip = lambda sd: OrderedDict((k[len("module.model."):], v) for k, v in sd.items())
    ref_s, ref_t = strip(algo.s_model.state_dict()), strip(algo.t_model.state_dict())

    tr = CO.CutMixOracleTrainer(TO.clone_state(s_state),
                                dict(max_iters=args.epochs * args.iters_per_epoch, cons_scale=20.0, cons_rampup_iters=0,
                                     cons_threshold=0.25, ema_decay=0.99, mask_prop_range=(0.5, 0.5)),
                                teacher_state=TO.clone_state(t_state))
    rng = np.random.RandomState(np_seed)
    outs = [tr.cutmix_step(x, gt, lbs, rng) for x, gt in batches]
    print("SSLCUTMIX._train:")
    for k in meters:
        check("mean " + k, sum(o[k] for o in outs) / len(outs), meters[k], rtol=2e-5)
        for i in range(iters):
            check("iter %d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=2e-5, atol=1e-9)
    for k in ("backbone.conv1.weight", "backbone.layer3.11.conv3.weight", "classifier.conv2d_list.0.weight",
              "backbone.bn1.running_mean"):
        check("student " + k, tr.sd[k], ref_s[k], rtol=2e-5)
        check("teacher " + k, tr.t_sd[k], ref_t[k], rtol=2e-5)
    assert all(0.0 < o["confidence"] < 1.0 for o in outs), [o["confidence"] for o in outs]
    torch.save(dict(kind="cutmix", size=size, lbs=lbs, ubs=ubs, weight_seed=seed, np_seed=np_seed, gamma3=gamma3,
                    ref_per_iter=ref_iters, student_updates=probe_update(ref_s, s_state, PROBES),
                    teacher_updates=probe_update(ref_t, t_state, PROBES),
                    data_seeds=[seed + 10 + i for i in range(iters)], block=block, cons_threshold=0.25, cons_scale=20.0,
                    max_iters=args.epochs * args.iters_per_epoch, meters=meters, per_iter=outs,
                    student_probes=probe(ref_s), teacher_probes=probe(ref_t),
                    masks_seed3=torch.from_numpy(ref_masks).to(torch.uint8)), OUT if out is None else os.path.join(os.path.dirname(OUT), out))
    print("wrote", out or OUT, os.path.getsize(OUT), "bytes; oracle == reference")


if __name__ == "__main__":
    main()
