"""Container-only loader for the *real* reference (PixelSSL @ /root/reference).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path imports this file.

It exists for two purposes:
  1. pin `oracle/torch_oracle.py` (our CPU restatement) against the reference's
     own classes on identical seeded inputs,
  2. generate the golden fixtures under `tests/golden/` (see make_golden.py).

The reference hard-codes `.cuda()`, imports cv2/torchvision at module top and
would download pretrained weights; three shims make it importable on a CPU-only
box without touching its sources:
  * stub `cv2`, `torchvision`, `torchvision.transforms` modules,
  * identity `.cuda()` on Tensor / Module,
  * no-op `ResNet._load_pretrained_model`.

`/root/reference` does not exist on the GPU box: callers must guard with
`reference_available()`.
"""
import os
import sys
import types
import argparse

REFERENCE_ROOT = os.environ.get("PIXELSSL_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pixelssl"))


_loaded = {}


def load_reference():
    """Import the reference packages; returns a dict of modules."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import torch

    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    sseg_root = os.path.join(REFERENCE_ROOT, "task", "sseg")
    for p in (REFERENCE_ROOT, sseg_root):
        if p not in sys.path:
            sys.path.insert(0, p)

    import pixelssl  # noqa: the reference package
    import importlib

    # the sseg task uses bare top-level module names (model, criterion, func, ...)
    ref_model = importlib.import_module("model")
    ref_criterion = importlib.import_module("criterion")
    ref_func = importlib.import_module("func")
    ref_proxy = importlib.import_module("proxy")
    from module.backbone import resnet as ref_resnet

    ref_resnet.ResNet._load_pretrained_model = lambda self: None

    _loaded.update(
        pixelssl=pixelssl,
        model=ref_model,
        criterion=ref_criterion,
        func=ref_func,
        proxy=ref_proxy,
        resnet=ref_resnet,
    )
    return _loaded


def make_args(algorithm, overrides):
    """Build the argparse.Namespace the reference's TaskProxy would build,
    without running `_preprocess_arguments` (it aborts without GPUs)."""
    ref = load_reference()
    pixelssl = ref["pixelssl"]
    from pixelssl import runner
    from pixelssl.utils import cmd

    parser = runner.create_parser(algorithm)
    ref["proxy"].add_parser_arguments(parser)
    cfg = {"ssl_algorithm": algorithm}
    cfg.update(overrides)
    args = cmd.parse_args(parser, cfg)
    # autoset fields (task_template/proxy.py:63-71, 251-261)
    args.gpus = 1
    args.task = "sseg"
    args.is_epoch_lrer = False
    if args.labeled_batch_size is None:
        args.labeled_batch_size = args.batch_size - args.unlabeled_batch_size
    if args.iters_per_epoch is None:
        args.iters_per_epoch = 100
    return args
