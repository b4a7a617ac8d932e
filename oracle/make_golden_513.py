"""Parity fixtures at the BASELINE size (513 x 513, B = 2, full ResNet-101) from the REAL reference (container only).
TEST INFRASTRUCTURE.   python oracle/make_golden_513.py

Reference `DeepLabV2` / `PSPNet` TaskModel forward (train-mode BN) + `CommonSSEGCriterion` + backward with the
reference's own initialisers; asserts that the oracle reproduces them at this size too and stores what a GPU test needs
without shipping 44 MB of logits: the logits on an 8-pixel grid, the arg-max map (zlib), the top-2 margin on the same
grid, per-sample CE, latent head / sums and gradient heads / norms.
"""
import os
import sys
import zlib
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim            # noqa: E402
import torch_oracle as TO  # noqa: E402
from make_golden import check, with_prefix, BASE_CFG  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

GRADS = {"deeplabv2": ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.layer3.11.conv3.weight",
                       "backbone.layer3.22.bn2.weight", "backbone.layer4.2.conv2.weight", "backbone.layer4.2.bn3.bias",
                       "classifier.conv2d_list.0.weight", "classifier.conv2d_list.3.bias"],
         "pspnet": ["backbone.conv1.weight", "backbone.layer3.11.conv3.weight", "backbone.layer4.2.conv2.weight",
                    "psp.stages.0.1.weight", "psp.stages.3.1.weight", "psp.stages.2.2.weight", "psp.bottleneck.0.weight",
                    "psp.bottleneck.1.bias", "decoder.0.weight", "decoder.1.conv.weight", "decoder.3.conv.bias"]}


def case(arch, size=513, batch=2, seed=211, gamma3=None):
    ref = ref_shim.load_reference()
    cfg = dict(BASE_CFG, models={'model': arch}, batch_size=batch, unlabeled_batch_size=0, im_size=size)
    args = ref_shim.make_args('ssl_null', cfg)
    psp = arch == "pspnet"
    state = TO.init_pspnet_state(seed=seed) if psp else TO.init_deeplabv2_state(seed=seed)
    if gamma3 is not None:          # conditioned weights (torch_oracle.condition_state): gradients reproducible at 1e-3
        TO.condition_state(state, gamma3)
    model = (ref['model'].PSPNet if psp else ref['model'].DeepLabV2)(args)
    model.load_state_dict(with_prefix(state, "model."))
    model.train()
    x, gt = TO.synthetic_batch(batch, size, batch, seed=seed + 1)
    resulter, _ = model.forward((x,))
    logits, prob, latent = resulter['pred'][0], resulter['activated_pred'][0], resulter['sslcct_ad_inp']
    per_sample = ref['criterion'].CommonSSEGCriterion(args).forward((logits,), (gt,), (x,))
    per_sample.mean().backward()
    named = dict(model.model.named_parameters())
    ref_grads = {k: named[k].grad for k in GRADS[arch]}
    ref_sd = OrderedDict((k[len("model."):], v) for k, v in model.state_dict().items())

    o_state = TO.clone_state(state)
    leaves = TO._param_leaves(o_state)
    run = TO._with_leaves(o_state, leaves)
    o_logits, o_prob, o_lat, _ = (TO.pspnet_forward if psp else TO.deeplabv2_forward)(run, x, train=True)
    o_ps = TO.sseg_criterion(o_logits, gt)
    o_ps.mean().backward()
    print("case %s forward+backward at %d:" % (arch, size))
    check("logits", o_logits, logits)
    check("latent", o_lat, latent)
    check("per-sample CE", o_ps, per_sample)
    for k, g in ref_grads.items():
        check("grad " + k, leaves[k].grad, g, rtol=1e-4)

    # how far the reference ARITHMETIC (fp32) is from the exact result on this fixture: the same functional forward /
    # backward in fp64.  The GPU test holds the engine to max(north_star's 1e-3, 3 x this gap) per quantity.
    st64 = OrderedDict((k, v.double() if v.is_floating_point() else v.clone()) for k, v in state.items())
    leaves64 = TO._param_leaves(st64)
    run64 = TO._with_leaves(st64, leaves64)
    d_logits, _, d_lat, _ = (TO.pspnet_forward if psp else TO.deeplabv2_forward)(run64, x.double(), train=True)
    d_ps = TO.sseg_criterion(d_logits, gt)
    d_ps.mean().backward()
    relf = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    gap = dict(logits=relf(logits.detach(), d_logits.detach()), latent=relf(latent.detach(), d_lat.detach()),
               per_sample=relf(per_sample.detach(), d_ps.detach()),
               grads={k: relf(g, leaves64[k].grad) for k, g in ref_grads.items()})
    print("  fp32-vs-fp64 gap of the reference arithmetic: logits %.2e latent %.2e CE %.2e" % (gap["logits"], gap["latent"], gap["per_sample"]))
    for k, v in gap["grads"].items():
        print("     grad %-40s %.2e" % (k, v))

    lg = logits.detach()
    top2 = lg.topk(2, dim=1).values
    am = lg.argmax(1).to(torch.uint8)
    rstat = [k for k in ref_sd if k.endswith("running_mean") or k.endswith("running_var")]
    fx = dict(kind="forward513", arch=arch, size=size, batch=batch, weight_seed=seed, data_seed=seed + 1, block=32, gamma3=gamma3,
              logits_grid=lg[:, :, ::8, ::8].clone(), logits_absmax=float(lg.abs().max()),
              logits_l2=float(lg.double().norm()), logits_sum=float(lg.double().sum()),
              prob_grid=prob.detach()[:, :, ::8, ::8].clone(),
              argmax_zlib=zlib.compress(am.numpy().tobytes(), 9), argmax_shape=tuple(am.shape),
              margin_zlib=zlib.compress((top2[:, 0] - top2[:, 1]).half().numpy().tobytes(), 6),
              per_sample=per_sample.detach().clone(),
              latent_head=latent.detach().reshape(-1)[:512].clone(), latent_grid=latent.detach()[:, ::16, ::4, ::4].clone(),
              latent_l2=float(latent.double().norm()),
              grads={k: dict(sample=g.reshape(-1)[::max(1, g.numel() // 4096)][:4096].clone(),
                             l2=float(g.double().norm())) for k, g in ref_grads.items()},
              running={k: ref_sd[k].reshape(-1)[:64].clone() for k in rstat[:4] + rstat[-4:]}, fp64_gap=gap)
    name = "%s_forward_513.pt" % arch if gamma3 is None else "%s_cond_forward_513.pt" % arch
    if gamma3 is not None:          # the conditioned twins keep the grids only (arg-max / margin maps are 2 MB)
        fx["argmax_grid"] = am[:, ::8, ::8].clone()
        fx["margin_grid"] = (top2[:, 0] - top2[:, 1])[:, ::8, ::8].clone()
        del fx["argmax_zlib"], fx["margin_zlib"]
    torch.save(fx, os.path.join(OUT, name))
    print("  wrote %s (%d bytes)" % (name, os.path.getsize(os.path.join(OUT, name))))


if __name__ == "__main__":
    if not ref_shim.reference_available():
        raise SystemExit("reference tree not available; fixtures can only be generated in the build container")
    torch.manual_seed(0)
    torch.set_num_threads(8)
    case("deeplabv2")
    case("pspnet", seed=221)
    case("deeplabv2", seed=231, gamma3=0.1)
    case("pspnet", seed=241, gamma3=0.1)
