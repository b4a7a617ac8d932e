#!/usr/bin/env python
"""Where does the bf16 engine's distance to the fp32 reference come from?  (GPU; writes a table to stdout.)

For SupOnly on PSPNet and for SSLCCT (fixtures pspnet_suponly_cond_129.pt / cct_cond_129.pt, iteration 0, the reference's
draws injected) the SAME training step runs on the fp32 engine and on the bf16 engine from identical weights; the table
lists, per parameter group of the main model, the cosine between the two updates, their norm ratio and
|update_bf16 - update_fp32| / |update_fp32|.  CCT is then repeated with ONE auxiliary decoder kind at a time: the rows
show which decoders' gradients are the noisy ones (I-VAT's adversarial direction is normalised rounding noise at the
script's xi = 1e-6; the masking decoders back-propagate through a hard mask of the main prediction).

    python tools/diag_bf16_grad.py > profiles/r03_bf16_grad_diag.txt
"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda"
KINDS = ["vat", "drop", "context", "object", "fd", "fn"]


def groups(sd):
    g = OrderedDict()
    for k, v in sd.items():
        if "running" in k or "num_batches" in k or not torch.is_floating_point(v):
            continue
        if k.startswith("backbone.layer"):
            name = k.split(".")[1]
        elif k.startswith("backbone."):
            name = "stem"
        else:
            name = k.split(".")[0]
        g.setdefault(name, []).append(k)
    return g


def run_cct(dtype, fx, kinds, TM):
    import torch_oracle as TO
    import cct_oracle as CO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from pixelssl_amd.sseg.func import SSEGFunc
    on = {k: int(k in kinds) for k in KINDS}
    args = TM._args(fx, dtype, models={"model": "pspnet"}, cons_scale=30.0, cons_rampup_epochs=5, ad_lr_scale=10.0,
                    vat_dec_num=on["vat"], vat_dec_xi=1e-6, vat_dec_eps=2.0, drop_dec_num=on["drop"], drop_dec_rate=0.5,
                    drop_dec_spatial=True, cut_dec_num=0, cut_dec_erase=0.4, context_dec_num=on["context"],
                    object_dec_num=on["object"], fd_dec_num=on["fd"], fn_dec_num=on["fn"], fn_dec_uniform=0.3)
    algo = P.ssl_algorithm.ssl_cct.ssl_cct(args, {"model": P.sseg.model.pspnet()}, {"model": popt.sgd(args)},
                                          {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()},
                                          SSEGFunc(args))
    wrapped = algo.model.module
    init = TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"])
    wrapped.main_model.model.load_state_dict(init)
    idx = [KINDS.index(k) for k in KINDS if k in kinds]
    for m, j in zip(wrapped.auxiliary_decoders, idx):
        m.load_state_dict(CO.init_decoder_state(fx["decoder_seeds"][j], in_channels=fx["in_channels"]))
        if fx["draws"][0][j] is not None:
            m.inject_draw(fx["draws"][0][j])
    algo.model.train()
    x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=fx["data_seeds"][0], block=fx["block"])
    # ramp-up position of the LAST fixture iteration, so that the consistency term carries its trained weight
    out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), len(fx["data_seeds"]) - 1, fx["rampup_iters"])
    sd = {k: v.detach().double().cpu() for k, v in wrapped.main_model.model.state_dict().items()}
    return init, sd, {k: v.item() for k, v in out.items()}


def run_suponly(dtype, fx, TM):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    args = TM._args(fx, dtype, models={"model": "pspnet"})
    algo = P.ssl_algorithm.ssl_null.ssl_null(args, {"model": P.sseg.model.pspnet()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    core = algo.model.module.model
    init = TO.condition_state(TO.init_pspnet_state(seed=fx["weight_seed"]), fx["gamma3"])
    core.load_state_dict(init)
    algo.model.train()
    x, gt = TO.synthetic_batch(fx["batch"], fx["size"], fx["batch"], seed=fx["data_seeds"][0], block=fx["block"])
    loss, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),))
    return init, {k: v.detach().double().cpu() for k, v in core.state_dict().items()}, {"task_loss": loss.item()}


def table(tag, init, sd32, sd16):
    print("\n== %s" % tag)
    print("%-12s %10s %10s %12s" % ("group", "cosine", "|16|/|32|", "err/|32|"))
    for name, keys in groups(sd32).items():
        u32 = torch.cat([(sd32[k] - init[k].double()).reshape(-1) for k in keys])
        u16 = torch.cat([(sd16[k] - init[k].double()).reshape(-1) for k in keys])
        n32, n16 = u32.norm().item(), u16.norm().item()
        if n32 == 0:
            continue
        print("%-12s %10.4f %10.4f %12.4f" % (name, (u32 @ u16).item() / (n32 * n16 + 1e-300), n16 / n32, (u16 - u32).norm().item() / n32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import test_multistep as TM
    torch.manual_seed(0)
    fx = TM._fx("pspnet_suponly_cond_129.pt")
    i, s32, l32 = run_suponly("fp32", fx, TM)
    _, s16, l16 = run_suponly("bf16", fx, TM)
    print("SupOnly / PSPNet, one iteration: losses fp32 %s bf16 %s" % (l32, l16))
    table("SupOnly / PSPNet: update of the bf16 engine vs the fp32 engine (same weights, same batch)", i, s32, s16)
    fx = TM._fx("cct_cond_129.pt")
    cases = [KINDS] + [[k] for k in KINDS]
    for kinds in cases:
        if a.only and ",".join(kinds) != a.only:
            continue
        i, s32, l32 = run_cct("fp32", fx, kinds, TM)
        _, s16, l16 = run_cct("bf16", fx, kinds, TM)
        print("\nCCT decoders %s: losses fp32 %s bf16 %s" % (kinds, l32, l16))
        table("CCT, decoders = %s" % (",".join(kinds) or "none"), i, s32, s16)


if __name__ == "__main__":
    main()
