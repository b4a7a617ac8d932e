"""Plugin-API conformance of the host mirror against the REAL reference: export functions, parser flags
and defaults, NAME / SUPPORTED_TASK_TYPES, the task-template hooks and their signatures, state_dict keys (including the
`module.model.` prefix), optimizer / scheduler state, and two-way checkpoint compatibility.  CPU only.

The comparisons against the reference run where /root/reference exists (the build container) and are skipped on the GPU
box; the alias test (`import pixelssl` == this package, reference-side plugin imports unchanged) runs everywhere."""
import argparse
import inspect
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402

needs_reference = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present on this box")
ALGOS = ["ssl_null", "ssl_mt", "ssl_adv", "ssl_cutmix", "ssl_gct", "ssl_cct", "ssl_s4l"]


def _params(fn):
    return [p for p in inspect.signature(fn).parameters]


def _flags(add_fn):
    parser = argparse.ArgumentParser()
    add_fn(parser)
    out = {}
    for a in parser._actions:
        if a.dest == "help":
            continue
        tname = getattr(a.type, "__name__", str(a.type))
        out[a.dest] = (a.default, tname, tuple(a.choices) if a.choices else None, tuple(sorted(a.option_strings)))
    return out


def _public_methods(cls):
    return {n: f for n, f in inspect.getmembers(cls, predicate=inspect.isfunction) if not n.startswith("_")}


def test_import_alias_and_reference_side_plugin_imports():
    """`import pixelssl` is this package (same module objects); plugin-style imports work; when the reference tree is
    present, its own task/sseg/criterion.py imports UNCHANGED on top of the alias and computes the per-sample CE."""
    code = r'''
import sys, os, argparse, importlib.util
sys.path.insert(0, %r)
import pixelssl, pixelssl_amd
import pixelssl.ssl_algorithm.ssl_mt as m
from pixelssl.utils import logger, cmd, tool
from pixelssl.nn import func
from pixelssl.nn.module import patch_replication_callback, GaussianNoiseLayer
from pixelssl.ssl_algorithm import ssl_base
assert pixelssl is pixelssl_amd and m is pixelssl_amd.ssl_algorithm.ssl_mt and ssl_base._SSLBase is pixelssl_amd.ssl_algorithm.ssl_base._SSLBase
for name in ("criterion_template", "model_template", "func_template", "SynchronizedBatchNorm2d", "log_err", "log_info",
             "log_warn", "str2bool", "str2intlist", "REGRESSION", "CLASSIFICATION", "SSL_ALGORITHMS", "SSL_MT", "SSL_GCT"):
    assert hasattr(pixelssl, name), name
assert callable(pixelssl.ssl_algorithm.__dict__["ssl_gct"].__dict__["ssl_gct"])      # proxy.py:433 lookup
ref = os.path.join(%r, "task", "sseg", "criterion.py")
if os.path.isfile(ref):
    import torch
    spec = importlib.util.spec_from_file_location("ref_criterion", ref)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    crit = mod.sseg_criterion()(argparse.Namespace(ignore_index=255))
    assert isinstance(crit, pixelssl_amd.task_template.TaskCriterion)
    out = crit.forward((torch.randn(2, 21, 9, 9),), (torch.randint(0, 21, (2, 1, 9, 9)).float(),), (None,))
    assert out.shape == (2,)
    print("REFPLUGIN ok")
print("ALIAS ok")
''' % (ROOT, ref_shim.REFERENCE_ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert "ALIAS ok" in out.stdout, out.stdout + out.stderr
    if ref_shim.reference_available():
        assert "REFPLUGIN ok" in out.stdout


@needs_reference
def test_algorithm_modules_match_the_reference():
    import pixelssl_amd as P
    ref = ref_shim.load_reference()["pixelssl"]
    for a in ALGOS:
        rm, om = ref.ssl_algorithm.__dict__[a], P.ssl_algorithm.__dict__[a]
        # export function named like the module, same parameters (ssl_base.py:19-37; proxy.py:433)
        assert _params(rm.__dict__[a]) == _params(om.__dict__[a]) == ["args", "model_dict", "optimizer_dict", "lrer_dict",
                                                                      "criterion_dict", "task_func"]
        # parser: same flags, defaults, types, choices
        rf, of = _flags(rm.add_parser_arguments), _flags(om.add_parser_arguments)
        assert set(rf) == set(of), (a, set(rf) ^ set(of))
        for k in rf:
            assert rf[k] == of[k], (a, k, rf[k], of[k])
        rc = [c for _, c in inspect.getmembers(rm, inspect.isclass) if getattr(c, "NAME", None) == a][0]
        oc = [c for _, c in inspect.getmembers(om, inspect.isclass) if getattr(c, "NAME", None) == a][0]
        assert rc.NAME == oc.NAME == a and rc.SUPPORTED_TASK_TYPES == oc.SUPPORTED_TASK_TYPES
        assert rc.__name__ == oc.__name__
        for meth in ("build", "train", "validate", "save_checkpoint", "load_checkpoint", "_build", "_train", "_validate",
                     "_save_checkpoint", "_load_checkpoint"):
            assert _params(getattr(rc, meth)) == _params(getattr(oc, meth)), (a, meth)
    assert P.SSL_NULL == ref.SSL_NULL and P.SSL_MT == ref.SSL_MT and P.SSL_ADV == ref.SSL_ADV
    assert P.SSL_GCT == ref.SSL_GCT and P.SSL_CCT == ref.SSL_CCT and P.SSL_CUTMIX == ref.SSL_CUTMIX
    assert set(P.SSL_ALGORITHMS) == set(ref.SSL_ALGORITHMS) and P.SSL_S4L == ref.SSL_S4L
    assert P.REGRESSION == ref.REGRESSION and P.CLASSIFICATION == ref.CLASSIFICATION


@needs_reference
def test_task_templates_and_sseg_hooks_match_the_reference():
    import pixelssl_amd as P
    r = ref_shim.load_reference()
    ref = r["pixelssl"]
    # template classes: same public methods, same signatures
    for rcls, ocls in ((ref.func_template.TaskFunc, P.func_template.TaskFunc),
                       (ref.model_template.TaskModel, P.model_template.TaskModel),
                       (ref.criterion_template.TaskCriterion, P.criterion_template.TaskCriterion)):
        rm, om = _public_methods(rcls), _public_methods(ocls)
        assert set(rm) <= set(om), (rcls.__name__, set(rm) - set(om))
        for n in rm:
            assert _params(rm[n]) == _params(om[n]), (rcls.__name__, n)
        assert _params(rcls.__init__) == _params(ocls.__init__), rcls.__name__
    assert ref.func_template.TaskFunc.METRIC_STR == P.func_template.TaskFunc.METRIC_STR
    # template defaults: identity conversions, NotImplementedError on the size hooks
    t, x = P.func_template.TaskFunc(None), torch.zeros(1)
    assert t.ssladv_convert_task_gt_to_fcd_input(x) is x and t.sslgct_prepare_task_gt_for_fdgt(x) is x
    for hook in ("ssladv_fcd_in_channels", "sslgct_fd_in_channels", "ssls4l_rc_in_channels", "sslcct_ad_in_channels",
                 "sslcct_ad_out_channels", "sslcct_ad_upsample_scale"):
        with pytest.raises(NotImplementedError):
            getattr(t, hook)()
    # the sseg task function class: every public hook of the reference, same signature, same constant answers
    rs, os_ = r["func"].SemanticSegmentationFunc, P.sseg.func.SSEGFunc
    rm, om = _public_methods(rs), _public_methods(os_)
    assert set(rm) <= set(om), set(rm) - set(om)
    for n in rm:
        assert _params(rm[n]) == _params(om[n]), n
    assert P.sseg.func.task_func() is os_ and P.sseg.func.SemanticSegmentationFunc is os_
    args = argparse.Namespace(num_classes=21, ignore_index=255, models={"model": "pspnet"}, im_size=65)
    mine = os_(args)
    assert mine.ssladv_fcd_in_channels() == 21 and mine.sslgct_fd_in_channels() == 24 and mine.ssls4l_rc_in_channels() == 21
    assert (mine.sslcct_ad_in_channels(), mine.sslcct_ad_out_channels(), mine.sslcct_ad_upsample_scale()) == (512, 21, 8)
    args.models = {"model": "deeplabv2"}
    assert mine.sslcct_ad_in_channels() == 2048
    # export functions of the task modules (proxy.py:205-216 resolves them by name)
    assert callable(P.sseg.model.deeplabv2) and callable(P.sseg.model.pspnet) and callable(P.sseg.criterion.sseg_criterion)
    assert r["model"].deeplabv2().__name__ == P.sseg.model.deeplabv2().__name__
    assert r["model"].pspnet().__name__ == P.sseg.model.pspnet().__name__
    assert r["criterion"].sseg_criterion().__name__ == P.sseg.criterion.sseg_criterion().__name__
    rf, of = _flags(r["model"].add_parser_arguments), _flags(P.sseg.model.add_parser_arguments)
    assert all(of[k] == v for k, v in rf.items()) and set(of) - set(rf) == {"engine_dtype", "pretrained_backbone"}
    # the two added flags: the engine's arithmetic, and where the backbone weights come from (the reference always downloads)


@needs_reference
def test_nn_and_utils_surface_matches_the_reference():
    import pixelssl_amd as P
    ref = ref_shim.load_reference()["pixelssl"]
    from pixelssl.nn import func as rfunc, optimizer as ropt, lrer as rlr
    from pixelssl.utils import cmd as rcmd, tool as rtool, logger as rlog
    for n in ("sigmoid_rampup", "split_tensor_tuple", "create_model", "model_str", "pytorch_support"):
        assert _params(getattr(rfunc, n)) == _params(getattr(P.nn.func, n)), n
    for cur, total in ((0, 10), (3, 6), (10, 10), (12, 10), (5, 0), (-1, 4)):
        assert rfunc.sigmoid_rampup(cur, total) == P.nn.func.sigmoid_rampup(cur, total)
    t = (torch.arange(24.).view(6, 2, 2), torch.arange(6.))
    for s, e, red in ((0, 4, False), (2, 6, False), (3, 4, True)):
        a, b = rfunc.split_tensor_tuple(t, s, e, red), P.nn.func.split_tensor_tuple(t, s, e, red)
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    # optimizer / scheduler parsers: identical flags, defaults and types; every factory we export exists in the reference
    assert _flags(ropt.add_parser_arguments) == _flags(P.nn.optimizer.add_parser_arguments)
    assert _flags(rlr.add_parser_arguments) == _flags(P.nn.lrer.add_parser_arguments)
    assert set(P.nn.VALID_OPTIMIZER) <= set(ref.nn.VALID_OPTIMIZER) and "sgd" in P.nn.VALID_OPTIMIZER
    assert P.nn.VALID_LRER == ref.nn.VALID_LRER and P.nn.lrer.ITER_LRERS == rlr.ITER_LRERS
    for n in P.nn.VALID_OPTIMIZER:
        assert _params(getattr(ropt, n)) == _params(getattr(P.nn.optimizer, n)) == ["args"]
    for n in P.nn.VALID_LRER:
        assert _params(getattr(rlr, n)) == _params(getattr(P.nn.lrer, n)) == ["args"]
    assert _params(rlr.PolynomialLR.__init__) == _params(P.nn.lrer.PolynomialLR.__init__)
    # factories resolve '-1 = default' like the reference
    def ns():
        return argparse.Namespace(lr=-1, weight_decay=-1, momentum=-1, dampening=-1, nesterov=False, beta1=-1, beta2=-1,
                                  eps=-1, amsgrad=False, alpha=-1, centered=False, power=-1, last_epoch=-1, epochs=7,
                                  iters_per_epoch=5, step_size=-1, milestones=[], gamma=-1, T_max=-1, eta_min=-1)
    for n in ("sgd", "adam"):
        a, b = ns(), ns()
        getattr(ropt, n)(a), getattr(P.nn.optimizer, n)(b)
        assert vars(a) == vars(b), n
    for n in P.nn.VALID_LRER:
        a, b = ns(), ns()
        getattr(rlr, n)(a), getattr(P.nn.lrer, n)(b)
        assert vars(a) == vars(b), n
    # epoch schedulers drive the fused optimizer's groups exactly like torch's SGD groups
    p1, p2 = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    for n in rlr.EPOCH_LRERS:
        o1 = torch.optim.SGD([{"params": [p1], "lr": 0.1}], lr=0.1)
        o2 = torch.optim.SGD([{"params": [p2], "lr": 0.1}], lr=0.1)
        s1, s2 = getattr(rlr, n)(ns())(o1), getattr(P.nn.lrer, n)(ns())(o2)
        for _ in range(6):
            o1.step(), o2.step(), s1.step(), s2.step()
            assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"], n
    # utils
    for n in ("parse_args", "print_args", "str2bool", "str2intlist", "str2floatlist"):
        assert _params(getattr(rcmd, n)) == _params(getattr(P.utils.cmd, n)), n
    assert rcmd.str2intlist("[1, 2,3]") == P.utils.cmd.str2intlist("[1, 2,3]") == [1, 2, 3]
    assert rcmd.str2floatlist("(0.5,1)") == P.utils.cmd.str2floatlist("(0.5,1)")
    assert _params(rtool.dict_value) == _params(P.utils.tool.dict_value)
    assert set(_public_methods(rlog.AvgMeterSet)) == set(_public_methods(P.utils.logger.AvgMeterSet))
    assert set(_public_methods(rlog.AvgMeter)) == set(_public_methods(P.utils.logger.AvgMeter))
    for n in ("log_mode", "log_file", "log_info", "log_warn", "log_err"):
        assert _params(getattr(rlog, n)) == _params(getattr(P.utils.logger, n)), n
    # GaussianNoiseLayer / patch_replication_callback keep their call shapes
    assert _params(ref.nn.module.GaussianNoiseLayer.__init__) == _params(P.nn.module.GaussianNoiseLayer.__init__)


def _base_args(**kw):
    a = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, lr=2.5e-4, momentum=0.9,
                           weight_decay=5e-4, dampening=-1, nesterov=False, power=-1, last_epoch=-1, epochs=1,
                           iters_per_epoch=4, ignore_index=255, labeled_batch_size=2, unlabeled_batch_size=2, batch_size=4,
                           ignore_unlabeled=False, is_epoch_lrer=False, log_freq=1000, task="sseg", engine_dtype="fp32",
                           cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99,
                           gaussian_noise_std=None, gpus=1, im_size=65, models={"model": "deeplabv2"})
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@needs_reference
def test_state_dict_keys_match_the_reference_models(monkeypatch):
    """`create_model(...)`.state_dict(): same keys (with the `module.model.` prefix), shapes and dtypes as the reference's
    DataParallel-wrapped TaskModels, FC discriminator and flaw detector."""
    monkeypatch.setenv("PXL_FORCE_DEVICE", "cpu")
    import pixelssl_amd as P
    from pixelssl_amd.ssl_algorithm import ssl_adv as OA, ssl_gct as OG
    r = ref_shim.load_reference()
    from pixelssl.nn import func as rfunc
    from pixelssl.ssl_algorithm import ssl_adv as RA, ssl_gct as RG
    rargs = ref_shim.make_args("ssl_null", dict(models={'model': 'deeplabv2'}, optimizers={'model': 'sgd'},
                                                lrers={'model': 'polynomiallr'}, criterions={'model': 'sseg_criterion'},
                                                lr=0.00025, output_stride=16, backbone='resnet101', epochs=1,
                                                batch_size=2, unlabeled_batch_size=0, im_size=65))
    pairs = [(rfunc.create_model(r["model"].DeepLabV2, "m", args=rargs), P.nn.func.create_model(P.sseg.model.DeepLabV2, "m", args=_base_args())),
             (rfunc.create_model(r["model"].PSPNet, "m", args=rargs), P.nn.func.create_model(P.sseg.model.PSPNet, "m", args=_base_args())),
             (rfunc.create_model(RA.FCDiscriminator, "d", in_channels=21), P.nn.func.create_model(OA.FCDiscriminator, "d", in_channels=21)),
             (rfunc.create_model(RG.FlawDetector, "fd", in_channels=24), P.nn.func.create_model(OG.FlawDetector, "fd", in_channels=24))]
    for rm, om in pairs:
        rs, os_ = rm.state_dict(), om.state_dict()
        assert list(rs.keys())[0].startswith("module.")
        assert set(rs.keys()) == set(os_.keys()), (type(rm.module).__name__, sorted(set(rs) ^ set(os_))[:6])
        for k, v in rs.items():
            assert tuple(v.shape) == tuple(os_[k].shape) and v.dtype == os_[k].dtype, k
        om.load_state_dict(rs)                                       # a reference state_dict loads ...
        assert all(torch.equal(om.state_dict()[k], v) for k, v in rs.items())
        rm.load_state_dict(om.state_dict())                          # ... and ours loads into the reference
        assert hasattr(om, "module") and hasattr(om.module, "param_groups") or not hasattr(rm.module, "param_groups")
    # lr groups: same parameter counts per group as the reference's (model.py:45-48, 103-107)
    for rm, om in pairs[:2]:
        rg = [sum(p.numel() for p in g["params"]) for g in rm.module.param_groups]
        og = [sum(p.numel() for p in g["params"]) for g in om.module.param_groups]
        assert rg == og and [g["lr"] for g in rm.module.param_groups] == [g["lr"] for g in om.module.param_groups]


@needs_reference
def test_checkpoints_are_compatible_both_ways(monkeypatch, tmp_path):
    """ssl_mt.py:296-322: a checkpoint written by the reference's `_save_checkpoint` after two training iterations
    (models + SGD momentum + poly-LR state) restores into this package's SSLMT -- weights, momentum buffers, lr and
    cur_iter -- and the checkpoint this package writes from that state loads back into the reference unchanged."""
    monkeypatch.setenv("PXL_FORCE_DEVICE", "cpu")
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_golden as MG
    r = ref_shim.load_reference()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    rargs = ref_shim.make_args('ssl_mt', dict(MG.BASE_CFG, batch_size=4, unlabeled_batch_size=2, im_size=33,
                                              ignore_unlabeled=False, cons_for_labeled=False, cons_scale=1.0,
                                              cons_rampup_epochs=3, ema_decay=0.99))
    rargs.iters_per_epoch = 4
    rargs.checkpoint_path = str(tmp_path)
    ralgo = MG._build_algo('ssl_mt', rargs)
    ralgo.s_model.module.load_state_dict(MG.with_prefix(TO.condition_state(TO.init_deeplabv2_state(seed=7), 0.1), "model."))
    ralgo.t_model.module.load_state_dict(MG.with_prefix(TO.condition_state(TO.init_deeplabv2_state(seed=8), 0.1), "model."))
    loader = MG._ListLoader([((x,), (gt,)) for x, gt in (TO.synthetic_batch(4, 33, 2, seed=70 + i, block=16) for i in range(2))])
    ralgo._train(loader, 0)
    ralgo._save_checkpoint(3)
    path = os.path.join(str(tmp_path), "checkpoint_3.ckpt")
    ref_ck = torch.load(path, weights_only=False)

    args = _base_args(resume=path, checkpoint_path=str(tmp_path / "ours"))
    os.makedirs(args.checkpoint_path)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    assert algo.load_checkpoint() == 3
    for tag, model in (("s_model", algo.s_model), ("t_model", algo.t_model)):
        sd = model.state_dict()
        assert set(sd) == set(ref_ck[tag]) and all(torch.equal(sd[k], v) for k, v in ref_ck[tag].items()), tag
    ropt_state = ref_ck["s_optimizer"]
    mine = algo.s_optimizer.state_dict()
    assert [len(g["params"]) for g in mine["param_groups"]] == [len(g["params"]) for g in ropt_state["param_groups"]]
    for g, rg in zip(mine["param_groups"], ropt_state["param_groups"]):
        for k in ("lr", "momentum", "weight_decay", "initial_lr"):
            assert g[k] == rg[k], k
    assert set(mine["state"].keys()) == set(ropt_state["state"].keys()) and len(mine["state"]) == 320
    for idx, st in ropt_state["state"].items():
        assert torch.equal(mine["state"][idx]["momentum_buffer"], st["momentum_buffer"]), idx
    assert algo.s_lrer.cur_iter == ref_ck["s_lrer"]["cur_iter"] == ralgo.s_lrer.cur_iter
    assert algo.s_lrer.state_dict()["base_lrs"] == ref_ck["s_lrer"]["base_lrs"]
    assert [g["lr"] for g in algo.s_optimizer.param_groups] == [g["lr"] for g in ralgo.s_optimizer.param_groups]

    # ... and back: our checkpoint into a fresh reference algorithm
    algo.save_checkpoint(4)
    ralgo2 = MG._build_algo('ssl_mt', rargs)
    ralgo2.args.resume = os.path.join(args.checkpoint_path, "checkpoint_4.ckpt")
    assert ralgo2._load_checkpoint() == 4
    for k, v in ralgo.s_model.state_dict().items():
        assert torch.equal(ralgo2.s_model.state_dict()[k], v), k
    a, b = ralgo.s_optimizer.state_dict()["state"], ralgo2.s_optimizer.state_dict()["state"]
    assert set(a) == set(b) and all(torch.equal(a[i]["momentum_buffer"], b[i]["momentum_buffer"]) for i in a)
    assert ralgo2.s_lrer.cur_iter == ralgo.s_lrer.cur_iter
