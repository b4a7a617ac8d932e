"""Golden fixture for SSLCCT (row C1) from the REAL reference imported from /root/reference.
Container-only; TEST INFRASTRUCTURE.

    python oracle/make_golden_cct.py        # writes tests/golden/cct_65.pt

Runs the reference's own `SSLCCT._train` (WrappedCCTModel: PSPNet + I-VAT / DropOut / Con-Msk / Obj-Msk / F-Drop /
F-Noise auxiliary decoders; G-Cutout needs OpenCV, which this image does not have) for two iterations on seeded
inputs, asserts that oracle/cct_oracle.py reproduces the logged losses and the post-step weights, and stores them
together with the random draws the decoders consumed.
"""
import os
import random
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim            # noqa: E402
import torch_oracle as TO  # noqa: E402
import cct_oracle as CO    # noqa: E402
from make_golden import check, _ListLoader, record_meters, per_iteration, probe_update  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

BASE_CFG = dict(models={'model': 'pspnet'}, optimizers={'model': 'sgd'}, lrers={'model': 'polynomiallr'},
                criterions={'model': 'sseg_criterion'}, lr=0.00025, momentum=0.9, weight_decay=0.0005,
                output_stride=16, backbone='resnet101', epochs=1, log_freq=1000)

DECODERS_BASE = [("vat", dict(xi=1e-6, eps=2.0)), ("drop", dict(rate=0.5, spatial=True)), ("context", {}), ("object", {}),
            ("fd", {}), ("fn", dict(uniform=0.3))]
DECODERS = list(DECODERS_BASE)

MAIN_PROBES = ["backbone.conv1.weight", "backbone.layer4.2.conv2.weight", "psp.stages.0.1.weight",
               "psp.stages.3.2.weight", "psp.bottleneck.0.weight", "psp.bottleneck.1.bias",
               "psp.bottleneck.1.running_mean", "decoder.0.weight", "decoder.2.conv.bias"]
AD_PROBES = ["upsample.0.weight", "upsample.1.conv.weight", "upsample.3.conv.bias"]


def head(v):
    v = v.detach().float().reshape(-1)
    return dict(head=v[:64].clone(), sum=float(v.double().sum()), abssum=float(v.double().abs().sum()))


def case_cct(size=65, lbs=2, ubs=2, seed=71, iters=2, rng_seed=1234, arch="pspnet", gamma3=None, out=None, block=16,
             with_cut=False, bias0_shift=0.0):
    """arch = 'pspnet' (the shipped script) or 'deeplabv2' (task/sseg/func.py:228: 2048-channel latent).
    with_cut: K = 7, the G-Cutout decoder included (BASELINE.json config 5).  The reference's own CutOutDecoder runs
    (ssl_cct.py:597-650: erase-window draws, mask, nearest resize, decoder body) with `cv2.findContours` replaced by
    cct_oracle.find_contours_stand_in; the fixture records the boxes that call returned and the random.randint draws, so
    the GPU test pins everything of the decoder except the OpenCV contour search itself."""
    global DECODERS
    ref = ref_shim.load_reference()
    DECODERS = list(DECODERS_BASE)
    cut_boxes, cut_calls = [], []
    if with_cut:
        DECODERS.insert(2, ("cut", dict(erase=0.4)))         # order of WrappedCCTModel (ssl_cct.py:440-470): vat, drop, cut, ...
        cv2 = sys.modules["cv2"]
        cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE = 0, 2

        def find_contours(mask_np, mode, method):
            contours, hier = CO.find_contours_stand_in(mask_np, mode, method)
            cut_calls.append([tuple(int(v) for v in (c[:, 0, 0].min(), c[:, 0, 0].max(), c[:, 0, 1].min(), c[:, 0, 1].max()))
                              for c in contours if c.shape[0] > 50])
            return contours, hier
        cv2.findContours = find_contours
    pixelssl = ref['pixelssl']
    from pixelssl.nn import optimizer as ropt, lrer as rlr
    batch = lbs + ubs
    args = ref_shim.make_args('ssl_cct', dict(BASE_CFG, models={'model': arch}, batch_size=batch, unlabeled_batch_size=ubs, im_size=size,
                                              ignore_unlabeled=False, cons_scale=30.0, cons_rampup_epochs=5,
                                              ad_lr_scale=10.0, vat_dec_num=1, drop_dec_num=1, cut_dec_num=1 if with_cut else 0, cut_dec_erase=0.4,
                                              context_dec_num=1, object_dec_num=1, fd_dec_num=1, fn_dec_num=1))
    args.iters_per_epoch = max(4, iters + 2)
    task_func = ref['func'].task_func()(args)
    export = pixelssl.ssl_algorithm.__dict__['ssl_cct'].__dict__['ssl_cct']
    psp = arch == "pspnet"
    algo = export(args, {'model': ref['model'].PSPNet if psp else ref['model'].DeepLabV2}, {'model': ropt.sgd(args)},
                  {'model': rlr.polynomiallr(args)}, {'model': ref['criterion'].CommonSSEGCriterion}, task_func)
    state = TO.init_pspnet_state(seed=seed) if psp else TO.init_deeplabv2_state(seed=seed)
    if gamma3 is not None:
        TO.condition_state(state, gamma3)
    if bias0_shift:
        # PSPNet's last layer feeds PixelShuffle(2): channels 0..3 are class 0.  Raising the background bias turns the
        # all-foreground prediction of a fresh network into ragged blobs whose contours pass the `> 50 vertices` filter
        assert psp
        state["decoder.3.conv.bias"][0:4] += bias0_shift
    fwd = TO.pspnet_forward if psp else TO.deeplabv2_forward
    cin = 512 if psp else 2048
    main_probes = MAIN_PROBES if psp else ["backbone.conv1.weight", "backbone.layer4.2.conv2.weight", "backbone.layer3.11.bn2.weight",
                                           "backbone.layer4.0.downsample.1.running_var", "classifier.conv2d_list.0.weight",
                                           "classifier.conv2d_list.3.bias"]
    ad_states = [CO.init_decoder_state(seed + 100 + i, in_channels=cin) for i in range(len(DECODERS))]
    wrapped = algo.model.module
    wrapped.main_model.load_state_dict(OrderedDict(("model." + k, v.clone()) for k, v in state.items()))
    kinds = [type(m).__name__ for m in wrapped.auxiliary_decoders]
    print("reference decoders:", kinds)
    for m, sd in zip(wrapped.auxiliary_decoders, ad_states):
        m.load_state_dict(OrderedDict((k, v.clone()) for k, v in sd.items()))
    batches = [TO.synthetic_batch(batch, size, lbs, seed=seed + 10 + i, block=block) for i in range(iters)]
    loader = _ListLoader([((x,), (gt,)) for x, gt in batches])

    torch.manual_seed(rng_seed); np.random.seed(rng_seed); random.seed(rng_seed)
    seen = record_meters(algo)
    algo._train(loader, 0)
    if with_cut:
        # ssl_cct.py:627-630 tries the OpenCV-3 signature first, so every mask is searched twice: keep one of each pair,
        # ubs masks per iteration
        assert len(cut_calls) == 2 * ubs * iters, (len(cut_calls), ubs, iters)
        assert all(cut_calls[2 * j] == cut_calls[2 * j + 1] for j in range(ubs * iters))
        per_mask = cut_calls[::2]
        cut_boxes = [per_mask[i * ubs:(i + 1) * ubs] for i in range(iters)]
        print("G-Cutout boxes per iteration:", [[len(b) for b in it] for it in cut_boxes])
        del cut_calls[:]
    ref_iters = per_iteration(seen, ('task_loss', 'cons_loss'), iters)
    meters = {k: float(algo.meters[k].avg) for k in ('task_loss', 'cons_loss')}
    ref_main = OrderedDict((k[len("model."):], v) for k, v in wrapped.main_model.state_dict().items())
    ref_ads = [m.state_dict() for m in wrapped.auxiliary_decoders]

    # ---- the restatement, same RNG streams
    torch.manual_seed(rng_seed); np.random.seed(rng_seed); random.seed(rng_seed)
    decs = [(k, c, OrderedDict((n, v.clone()) for n, v in sd.items())) for (k, c), sd in zip(DECODERS, ad_states)]
    tr = CO.CCTOracleTrainer(TO.clone_state(state), decs,
                             dict(max_iters=args.epochs * args.iters_per_epoch, cons_scale=30.0,
                                  cons_rampup_iters=len(loader) * 5, ad_lr_scale=10.0), forward=fwd)
    outs = [tr.cct_step(x, gt, lbs) for x, gt in batches]
    print("case cct (%s):" % arch)
    for k in meters:
        check("mean " + k, sum(o[k] for o in outs) / len(outs), meters[k])
        for i in range(iters):
            check("iter %d %s" % (i, k), outs[i][k], ref_iters[i][k], rtol=2e-5, atol=1e-9)
    for k in main_probes:
        check("main " + k, tr.sd[k], ref_main[k], rtol=2e-5)
    for i, (kind, _, sd) in enumerate(tr.decoders):
        for k in AD_PROBES:
            check("decoder %d (%s) %s" % (i, kind, k), sd[k], ref_ads[i][k], rtol=2e-5)

    # draws are replayable: the oracle fed with its own recorded draws gives the same losses
    tr2 = CO.CCTOracleTrainer(TO.clone_state(state),
                              [(k, c, OrderedDict((n, v.clone()) for n, v in sd.items())) for (k, c), sd in zip(DECODERS, ad_states)],
                              dict(max_iters=args.epochs * args.iters_per_epoch, cons_scale=30.0,
                                   cons_rampup_iters=len(loader) * 5, ad_lr_scale=10.0), forward=fwd)
    o0 = tr2.cct_step(batches[0][0], batches[0][1], lbs, draws=outs[0]["draws"])
    check("replayed draws: cons", o0["cons_loss"], outs[0]["cons_loss"])

    draws_fx = [list(o["draws"]) for o in outs]
    if with_cut:
        ci = [k for k, _ in DECODERS].index("cut")
        for i in range(iters):
            assert len(draws_fx[i][ci]) == 2 * sum(len(b) for b in cut_boxes[i]), "two draws per kept contour"
            draws_fx[i][ci] = dict(u=list(draws_fx[i][ci]), boxes=cut_boxes[i])
    g0 = outs[0]
    fx = dict(kind="cct", arch=arch, in_channels=cin, size=size, lbs=lbs, ubs=ubs, weight_seed=seed, decoder_seeds=[seed + 100 + i for i in range(len(DECODERS))],
              gamma3=gamma3, ref_per_iter=ref_iters, main_updates=probe_update(ref_main, state, main_probes),
              ad_updates=[probe_update(sd, sd0, AD_PROBES) for sd, sd0 in zip(ref_ads, ad_states)],
              decoders=DECODERS, data_seeds=[seed + 10 + i for i in range(iters)], block=block,
              max_iters=args.epochs * args.iters_per_epoch, rampup_iters=len(loader) * 5,
              meters=meters, per_iter=[dict(task_loss=o["task_loss"], cons_loss=o["cons_loss"]) for o in outs],
              draws=draws_fx, with_cut=bool(with_cut), bias0_shift=float(bias0_shift),
              grads0={k: head(g0["grads"][k]) for k in main_probes if k in g0["grads"]},
              ad_grads0=[{k: head(g[k]) for k in AD_PROBES} for g in g0["ad_grads"]],
              main_probes={k: head(ref_main[k]) for k in main_probes},
              ad_probes=[{k: head(sd[k]) for k in AD_PROBES} for sd in ref_ads])
    torch.save(fx, os.path.join(OUT, out or ("cct_%d.pt" if psp else "cct_deeplab_%d.pt") % size))


if __name__ == "__main__":
    if not ref_shim.reference_available():
        raise SystemExit("reference tree not available; fixtures can only be generated in the build container")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    case_cct()
    case_cct(arch="deeplabv2", seed=81, rng_seed=4321)
    print("golden fixtures written to", OUT)
