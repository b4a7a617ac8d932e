"""CPU-only checks of the host logic and the C-ABI surface (no kernel is launched here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from pixelssl_amd import _lib
    h = _lib.lib()
    assert h.pxl_version() == 100
    header = open(os.path.join(ROOT, "include", "pixelhip.h")).read()
    declared = set(re.findall(r"\b(pxl_[a-z0-9_]+)\s*\(", header))
    declared -= {"pxl_allreduce_fn"}
    assert len(declared) >= 40
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libpixelhip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes binding misses %s" % name


def test_argument_errors_are_reported_not_crashed():
    from pixelssl_amd import _lib
    h = _lib.lib()
    d = _lib.ConvDesc()
    d.dtype, d.Cin, d.ntaps, d.div, d.Kreal, d.Cout = 7, 64, 1, 1, 8, 8
    assert h.pxl_conv_igemm(d, 1, 1, 1, None, None, None, None, None, None, 0, None) == -1
    assert b"dtype" in h.pxl_last_error()
    with pytest.raises(_lib.PixelHipError):
        _lib.check(h.pxl_mse_fwd(0, None, None, None, None))
    # the later entry points refuse bad arguments before any launch, too (no GPU needed to see that)
    assert h.pxl_adaptive_avgpool_fwd(1, 1, 33, 33, 64, 6, None, None, None) == -1 and b"adaptive_avgpool" in h.pxl_last_error()
    assert h.pxl_upsample_slice_fwd(1, 1, 2, 2, 64, 60, 1, None, 0, 33, 33, 1, 4096, 2048, None) == -1       # C % 8 != 0
    assert b"16-byte aligned" in h.pxl_last_error()
    assert h.pxl_pixshuf_relu_fwd(1, 1, 4, 4, 32, 21, 1, 1, 32, None) == -1                                  # 4*C > Cp_in
    assert h.pxl_latent_perturb(0, 512, 25, None, None, None, None, None, 1.0, None, None) == -1
    assert h.pxl_residual_bwd_reduce(1, 10, 60, 1, 1, 1, 1, 1, None, 1, None) == -1 and b"16-byte" in h.pxl_last_error()
    fin = _lib.BnFin()                                                          # no statistics, no coefficient output
    assert h.pxl_bn_finalize_apply_fwd(1, 10, 64, 1, fin, 1, 1, None) == -1 and b"bn_finalize_apply_fwd" in h.pxl_last_error()
    d2 = _lib.ConvDesc()
    d2.dtype, d2.Cin, d2.ntaps, d2.div, d2.Kreal, d2.Cout = 0, 64, 1, 1, 64, 64                               # fp32: no LDS-DMA
    assert h.pxl_conv_dgrad_bnreduce(d2, 1, 1, 1, None, 1, 1, 1, 1, None) == -3 and b"not eligible" in h.pxl_last_error()
    assert h.pxl_comm_init(None, 0, 1, None) == -1 and h.pxl_comm_allreduce_sum(None, None, 4, None) == -1
    assert h.pxl_external_contour_boxes_host(None, 4, 4, 50, None, 0, None) == -1
    assert h.pxl_tune_set(99, 1) == -1
    # round-2 entry points
    assert h.pxl_stem_patches(1, 1, 1, 2, 3, 33, 33, 7, 7, 2, 3, 17, 17, 144, None) == -1 and b"pitch" in h.pxl_last_error()   # 147 > 144
    assert h.pxl_stem_patches(1, None, 1, 2, 3, 33, 33, 7, 7, 2, 3, 17, 17, 192, None) == -1
    assert h.pxl_ce_mse_bwd(4, 21, 100, 1, 1, 255, 5, 1, 1, 0, 4, 1, 1, None) == -1                        # n_ce > N
    assert h.pxl_ce_mse_bwd(4, 21, 100, 1, 1, 255, 2, 1, 1, 3, 2, 1, 1, None) == -1 and b"consistency range" in h.pxl_last_error()
    assert h.pxl_ce_mse_bwd(4, 21, 100, 1, None, 255, 2, 1, 1, 2, 4, 1, 1, None) == -1 and b"label maps" in h.pxl_last_error()
    assert h.pxl_nchw_parts_to_nhwc(1, 5, 1, 1, 1, 2, 8, 8, 32, None) == -1                                 # > 4 parts
    assert h.pxl_conv_dgrad_joinreduce(d2, 1, 1, 1, None, None, 1, 1, 1, None) == -1                         # no join output
    assert h.pxl_conv_dgrad_joinreduce(d2, 1, 1, 1, None, 1, 1, 1, 1, None) == -3 and b"not eligible" in h.pxl_last_error()
    assert h.pxl_confusion_matrix(0, 21, 100, None, None, None, None) == -1
    assert h.pxl_colsum(1, 100, 20, 20, None, None, None) == -1          # (null pointers are caught before the pitch)
    assert h.pxl_l2_normalize_persample(4, 100, None, 1.0, None, None, None) == -1 and b"scratch" in h.pxl_last_error()
    assert h.pxl_absdiff_chansum_dense(1, 21, 100, None, 1, 1.0, 1, None) == -1


def test_engine_parameter_tree_matches_reference_names():
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    core = DeepLabV2Core(device="cpu", engine_dtype=torch.bfloat16)
    sd = core.state_dict()
    ref = TO.deeplabv2_param_shapes()
    assert set(sd.keys()) == set(ref.keys())
    for k, shape in ref.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in core.parameters()) == 44048532
    # parameters are views of ONE flat buffer, conv weights physically [K][kh][kw][C]
    w = core.backbone.layer1._modules["0"].conv2.weight
    assert w.is_contiguous(memory_format=torch.channels_last)
    st = TO.init_deeplabv2_state(seed=3)
    core.load_state_dict(st)
    off = w._pxl_flat[1]
    assert torch.equal(core.flat.params[off:off + w.numel()].view(64, 3, 3, 64),
                       st["backbone.layer1.0.conv2.weight"].permute(0, 2, 3, 1))
    groups = [list(core.get_1x_lr_params()), list(core.get_10x_lr_params())]
    assert sum(p.numel() for p in groups[0]) == 42500160 and sum(p.numel() for p in groups[1]) == 1548372


def test_pspnet_and_cct_parameter_trees_and_plans():
    """PSPNet / SSLCCT auxiliary decoder cores: reference state_dict names and shapes, lr groups, and the executor plans
    their layer programs (new ops: adaptive pool, concat slices, PixelShuffle, HEAD with its own output size)."""
    import torch_oracle as TO
    import cct_oracle as CO
    from pixelssl_amd.engine import PSPNetCore, AuxDecoderCore
    from pixelssl_amd._lib import lib, check
    core = PSPNetCore(device="cpu", engine_dtype=torch.bfloat16)
    sd = core.state_dict()
    ref = TO.pspnet_param_shapes()
    assert set(sd.keys()) == set(ref.keys())
    for k, shape in ref.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert sum(p.numel() for p in core.parameters()) == 65590248
    core.load_state_dict(TO.init_pspnet_state(seed=1))
    groups = [list(core.get_backbone_params()), list(core.get_psp_params()), list(core.get_decoder_params())]
    assert sum(len(g) for g in groups) == len(list(core.parameters())) and all(len(g) > 0 for g in groups)
    h = lib()
    core._plan(8, 513, 513)
    c, hh, ww = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(h.pxl_net_latent_shape(core._net, ctypes.byref(c), ctypes.byref(hh), ctypes.byref(ww)))
    assert (c.value, hh.value, ww.value) == (512, 33, 33)                   # 'sslcct_ad_inp' = the pyramid module's output
    assert 0 < h.pxl_net_arena_bytes(core._net) < 32 * 2 ** 30
    dec = AuxDecoderCore(8, 512, 21, device="cpu", engine_dtype=torch.bfloat16)
    assert {"upsample." + k for k in dec.state_dict().keys()} == set(CO.decoder_param_shapes().keys())
    p1 = dec._plan(4, 33, 33, (513, 513))
    p2 = dec._plan(4, 33, 33, (264, 264))                                    # I-VAT's own-resolution passes: a second plan
    assert p1 is not p2 and dec._plan(4, 33, 33, (513, 513)) is p1 and p1.out_size == (513, 513)


def test_plan_sizes_and_workspace_guards():
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd._lib import lib, check
    core = DeepLabV2Core(device="cpu", engine_dtype=torch.float32)
    h = lib()
    assert core._shape is None                            # not planned yet: one executor instance per input shape
    core._plan(2, 65, 65)
    a65 = h.pxl_net_arena_bytes(core._net)
    p65 = core._cur
    core._plan(8, 513, 513)
    a513 = h.pxl_net_arena_bytes(core._net)
    assert 0 < a65 < a513 < 16 * 2 ** 30
    assert core._plan(2, 65, 65) is p65 and core._shape == (2, 65, 65)      # plans are cached per shape
    core._plan(8, 513, 513)
    c, hh, ww = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(h.pxl_net_latent_shape(core._net, ctypes.byref(c), ctypes.byref(hh), ctypes.byref(ww)))
    assert (c.value, hh.value, ww.value) == (2048, 33, 33)
    # a too-small arena is refused before anything is launched
    rc = h.pxl_net_forward(core._net, 1, 1, None, 1, 1, None, 1, 16, 1, None)
    assert rc == -4 and b"arena" in h.pxl_last_error()


def test_product_path_refuses_cpu_tensors():
    from pixelssl_amd import functional as PF, _lib
    from pixelssl_amd.engine import DeepLabV2Core
    with pytest.raises(_lib.PixelHipError):
        PF.mse_loss(torch.zeros(4), torch.zeros(4))
    core = DeepLabV2Core(device="cpu", engine_dtype=torch.float32)
    with pytest.raises(_lib.PixelHipError):
        core(torch.zeros(1, 3, 33, 33))


def test_missing_library_fails_loudly(tmp_path):
    code = ("import pixelssl_amd._lib as L; L.LIB_PATH=%r; L._lib=None\n"
            "try:\n  L.lib()\nexcept L.PixelHipError as e:\n  print('LOUD', e)\n" % str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "LOUD" in out.stdout and "no CPU/PyTorch fallback" in out.stdout


def test_polynomial_lr_matches_oracle_schedule():
    import argparse
    import torch_oracle as TO
    from pixelssl_amd.nn import lrer
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([{"params": [p], "lr": 2.5e-4}, {"params": [torch.nn.Parameter(torch.zeros(1))], "lr": 2.5e-3}],
                          lr=2.5e-4)
    args = argparse.Namespace(power=-1, last_epoch=-1, epochs=2, iters_per_epoch=5)
    sched = lrer.polynomiallr(args)(opt)
    tr = TO.OracleTrainer({}, dict(max_iters=10))
    for it in range(6):
        tr.it = it
        assert abs(opt.param_groups[0]["lr"] - tr._lrs()[0]) < 1e-12
        assert abs(opt.param_groups[1]["lr"] - tr._lrs()[1]) < 1e-12
        sched.step()


def test_helpers_and_registry():
    import pixelssl_amd as P
    from pixelssl_amd.nn import func
    assert P.SSL_ALGORITHMS == ["ssl_null", "ssl_mt", "ssl_adv", "ssl_cutmix", "ssl_gct", "ssl_cct", "ssl_s4l"]
    for name in P.SSL_ALGORITHMS:      # lookup convention of task_template/proxy.py:433
        assert callable(P.ssl_algorithm.__dict__[name].__dict__[name])
        assert callable(P.ssl_algorithm.__dict__[name].add_parser_arguments)
    t = (torch.arange(12).view(6, 2),)
    assert func.split_tensor_tuple(t, 0, 4)[0].shape == (4, 2)
    assert func.split_tensor_tuple(t, 2, 3, reduce_dim=True)[0].shape == (2,)
    assert func.sigmoid_rampup(3, 6) == pytest.approx(0.2865047968601901)
    import argparse
    parser = argparse.ArgumentParser()
    P.nn.optimizer.add_parser_arguments(parser)
    P.nn.lrer.add_parser_arguments(parser)
    P.ssl_algorithm.ssl_mt.add_parser_arguments(parser)
    P.sseg.model.add_parser_arguments(parser)
    a = P.utils.cmd.parse_args(parser, {"lr": 0.00025, "cons_for_labeled": False, "ema_decay": 0.99})
    assert a.lr == 0.00025 and a.cons_for_labeled is False and a.backbone == "resnet101"


def test_attach_reaches_executors_kept_outside_the_module_registry(monkeypatch):
    """dist.attach: FCDiscriminator / FlawDetector / RotationClassifer register the LEAVES of their executor under the
    reference's names and keep the executor itself out of `modules()`; multi-rank wiring (Sync-BN hook, gradient
    exchange) must still reach it -- a missed network trains on un-averaged gradients without any error."""
    os.environ.setdefault("PXL_FORCE_DEVICE", "cpu")
    import torch
    from pixelssl_amd import dist as pdist
    from pixelssl_amd.engine import SegNetCore
    from pixelssl_amd.ssl_algorithm.ssl_s4l import RotationClassifer
    from pixelssl_amd.ssl_algorithm.ssl_adv import FCDiscriminator
    monkeypatch.setattr(pdist, "is_distributed", lambda: True)
    monkeypatch.setattr(pdist, "world_size", lambda: 2)
    monkeypatch.setattr(pdist, "native_comms", lambda: [])
    wired = []
    monkeypatch.setattr(SegNetCore, "set_sync", lambda self, cb, ws: wired.append(("sync", id(self), ws)))
    monkeypatch.setattr(SegNetCore, "set_grad_sync", lambda self, fn, user, ws, bucket: wired.append(("grad", id(self), ws)))
    model = torch.nn.ModuleDict(dict(rc=RotationClassifer(21), d=FCDiscriminator(21)))
    assert not any(isinstance(m, SegNetCore) for m in model.modules())          # invisible to a plain modules() walk
    pdist.attach(model)
    for core in (model["rc"].core, model["d"].core):
        assert getattr(core, "_pxl_attached", False) and ("grad", id(core), 2) in wired
        assert core._post_backward_hook is pdist._post_backward
    # Sync-BN statistics: the discriminator has none to exchange but is wired like every network; the rotation classifier's
    # nn.BatchNorm2d layers are LOCAL in the reference (ssl_s4l.py:376-384: plain BatchNorm inside DataParallel replicas)
    assert ("sync", id(model["d"].core), 2) in wired and ("sync", id(model["rc"].core), 2) not in wired
    n = len(wired)
    pdist.attach(model)                                                           # idempotent
    assert len(wired) == n


def test_pretrained_backbone_file_and_model_zoo_url(tmp_path, monkeypatch):
    """ResNet._load_pretrained_model (task/sseg/module/backbone/resnet.py:145-156) on the engine's trunk: a FILE is a strict
    state dict of the trunk; a URL (here file://, through torch.hub's cache like model_zoo.load_url) is filtered to the keys
    the dilated trunk has -- a torchvision ResNet also carries fc.* -- and everything outside the trunk keeps its values."""
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "hub"))
    core = DeepLabV2Core(device="cpu", engine_dtype=torch.bfloat16)
    before = {k: v.clone() for k, v in core.state_dict().items()}
    donor = TO.init_deeplabv2_state(seed=77)
    trunk = {k[len("backbone."):]: v + 0.25 for k, v in donor.items() if k.startswith("backbone.") and v.dtype.is_floating_point}
    trunk.update({k[len("backbone."):]: v for k, v in donor.items() if k.startswith("backbone.") and not v.dtype.is_floating_point})
    # (1) model-zoo style: torchvision names = the trunk's names + the classifier head the trunk does not have
    zoo = dict(trunk)
    zoo["fc.weight"], zoo["fc.bias"] = torch.randn(1000, 2048), torch.randn(1000)
    f = tmp_path / "resnet101-synthetic.pth"
    torch.save(zoo, f)
    taken = core.load_pretrained_backbone("file://" + str(f))
    assert "fc.weight" not in taken and len(taken) == len(trunk)
    sd = core.state_dict()
    for k, v in trunk.items():
        assert torch.equal(sd["backbone." + k], v), k
    for k, v in before.items():
        if not k.startswith("backbone."):
            assert torch.equal(sd[k], v), k                    # the ASPP head is untouched
    # (2) a file path: strict -- the zoo file with fc.* is refused, the trunk-only file loads
    with pytest.raises(RuntimeError, match="unexpected"):
        core.load_pretrained_backbone(str(f))
    g = tmp_path / "trunk.pth"
    torch.save({k: v * 2 if v.dtype.is_floating_point else v for k, v in trunk.items()}, g)
    core.load_pretrained_backbone(str(g))
    assert torch.equal(core.state_dict()["backbone.layer3.5.conv2.weight"], trunk["layer3.5.conv2.weight"] * 2)
    # (3) the task model's switch: --pretrained-backbone <file>
    import argparse
    from pixelssl_amd.sseg import model as SM
    args = argparse.Namespace(backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, engine_dtype="bf16",
                              lr=0.001, pretrained_backbone=str(g))
    tm = SM.DeepLabV2(args)
    assert torch.equal(tm.model.state_dict()["backbone.conv1.weight"], trunk["conv1.weight"] * 2)
    assert SM.PRETRAINED_BACKBONE_URLS["resnet101"].endswith("resnet101-5d3b4d8f.pth")


def test_a_peer_time_out_stays_on_record_until_the_epoch_guard():
    """after the opt-in fall-back retires the peer-mapped contexts, check_peers() -- the epoch-end guard of
    ssl_base.train() -- iterated an EMPTY list and never raised.  The time-out is now sticky: whoever notices it (poll_peers, a
    status read) records it, and check_peers() raises on the record whatever happened to the contexts since."""
    from pixelssl_amd import dist as pdist, _lib
    keep = dict(pdist._peer)
    try:
        pdist._peer.update(ctxs=[], timed_out=None)
        pdist.check_peers()                                   # nothing on record, no contexts: fine
        pdist._peer["timed_out"] = 3                          # rank 2 was given up on; the contexts are long retired
        with pytest.raises(_lib.PixelHipError, match="rank 2"):
            pdist.check_peers()
        assert pdist.poll_peers() is False                    # (single process: nothing to poll, nothing raised)
    finally:
        pdist._peer.clear()
        pdist._peer.update(keep)
    assert pdist.PEER_TIMEOUT_MS >= 20000 or "PXL_PEER_TIMEOUT_MS" in os.environ


def test_bytes_per_step_tool_reproduces_the_committed_figure():
    """profiles/bytes_per_step.json (what bench.py reports as `bytes_per_step`) is tools/bytes_per_step.py applied to the committed
    PMC traffic table and the committed one-step trace: re-derive it."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bytes_per_step as BPS
    out = os.path.join(ROOT, "profiles", "_bps_check.json")
    try:
        BPS.main(os.path.join(ROOT, "profiles", "r06_traffic.json"), os.path.join(ROOT, "profiles", "r06_c_step_breakdown.txt"), None, out)
        got = json.load(open(out))
    finally:
        if os.path.exists(out):
            os.remove(out)
    want = json.load(open(os.path.join(ROOT, "profiles", "bytes_per_step.json")))
    assert got["bytes_per_step"] == want["bytes_per_step"] and 30e9 < got["bytes_per_step"] < 40e9
    fam = got["families"]
    assert fam["conv_dma_kernel"]["launches"] >= 300 and fam["sgd_ema_pack_kernel"]["launches"] == 2
    # the dominant family carries about half of the step's bytes; nothing is counted twice
    assert 0.40 < fam["conv_dma_kernel"]["bytes"] / got["bytes_per_step"] < 0.55
    assert abs(sum(v["bytes"] for v in fam.values()) - got["bytes_per_step"]) <= len(fam)
    # round 6: ONE stem-patch launch serves both networks (Mean Teacher without input noise), the head runs as a GEMM
    assert fam["stem_patches_kernel"]["launches"] == 1 and fam["aspp_col2im_kernel"]["launches"] == 2
    # profiles/step_trace.json (bench.py: `step_trace`) is the same committed trace
    st = json.load(open(os.path.join(ROOT, "profiles", "step_trace.json")))
    assert st["launches_per_step"] == sum(v["launches"] for v in fam.values()) and st["conv_dma_launches"] == 320
    assert abs(st["kernel_sum_ms"] - st["contraction_ms"] - st["noncontraction_ms"]) < 0.01 and st["conv_dma_ms"] < st["contraction_ms"]


def test_per_variant_traffic_of_the_dominant_kernel_is_on_record():
    """profiles/r05_traffic*.json: conv_dma grouped by epilogue variant -- the ASPP forward's re-reads before
    and after the channel-sliced split-K"""
    import json
    before = json.load(open(os.path.join(ROOT, "profiles", "r05_traffic_before_channel_split.json")))["conv_dma_variants"]
    after = json.load(open(os.path.join(ROOT, "profiles", "r05_traffic.json")))["conv_dma_variants"]
    assert before["EM=-2"]["traffic_bytes_per_launch"] > 5 * after["EM=-2"]["traffic_bytes_per_launch"] > 0
    assert sum(v["launches"] for v in after.values()) == 319
    assert after["EM=29"]["traffic_bytes_per_launch"] > 2.5 * after["EM=4"]["traffic_bytes_per_launch"]
    # round 6: the join backward reads its ReLU mask as a bit plane (EM = 93 = 29 | 64): 17.8 MB less per stage-3 launch
    r6 = json.load(open(os.path.join(ROOT, "profiles", "r06_traffic.json")))["conv_dma_variants"]
    assert "EM=93" in r6 and r6["EM=93"]["traffic_bytes_per_launch"] < after["EM=29"]["traffic_bytes_per_launch"] - 10e6


def test_the_steps_kernels_fit_their_register_budget():
    """tools/isa_resources.py reads every kernel's VGPR / scratch figures from the gfx950 code objects of the built library (no GPU):
    the kernels the training step spends its time in must not spill to scratch, and the row-streaming kernels must leave >= 4 waves
    per SIMD (they are latency-hiding bound).  profiles/r05_isa_resources.txt is this table for the committed tree."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_resources as IR
    if not (os.path.exists(os.path.join(IR.LLVM, "llvm-readelf")) and os.path.exists(os.path.join(IR.LLVM, "clang-offload-bundler"))
            and shutil.which("c++filt")):
        pytest.skip("no llvm binutils / c++filt on this box")
    from pixelssl_amd import _lib
    _lib.lib()                                                # (builds the objects when they are missing)
    rows = {}
    with tempfile.TemporaryDirectory() as tmp:
        for fn in ("conv_dma_a.hip.o", "conv_wgrad_dma.hip.o", "eltwise.hip.o", "bn.hip.o", "optim.hip.o", "head.hip.o"):
            ks = IR.kernels_of(os.path.join(IR.OBJ, fn), tmp)
            assert ks, fn
            for k, n in zip(ks, IR.demangle([k["name"] for k in ks])):
                rows[IR.short(n, 400)] = k
    assert len(rows) > 150

    def pick(prefix):
        got = {n: k for n, k in rows.items() if n.startswith(prefix)}
        assert got, prefix
        return got
    # the workhorse tiles of the LDS-DMA convolution (4 waves of 64 lanes, 128 x 128 and 128 x 64 pixels x channels, every epilogue
    # variant in this object): accumulators in AGPRs, no scratch, at least two workgroups' worth of waves per SIMD
    dma = {n: k for n, k in rows.items() if n.startswith("pxl_dma::conv_dma_kernel<128, 128, 2, 2,") or n.startswith("pxl_dma::conv_dma_kernel<128, 64, 2, 2,")}
    assert len(dma) >= 20
    for n, k in dma.items():
        assert k["scratch"] == 0 or k["vspill"] <= 4, (n, k)              # (one 128 x 64 statistics variant spills 4 registers outside its K loop)
        assert k["agpr"] >= 32 and IR.occupancy(k)[0] >= 2, (n, k)
    for prefix, min_waves in (("bn_apply_fwd_kernel", 8), ("residual_fwd_kernel", 4), ("bn_bwd_apply_fused_kernel", 4),
                              ("bn_bwd_reduce_kernel", 5), ("sgd_ema_pack_kernel", 4), ("maxpool_fwd_kernel", 6)):
        for n, k in pick(prefix).items():
            assert k["scratch"] == 0 and k["vspill"] == 0, (n, k)
            assert IR.occupancy(k)[0] >= min_waves, (n, k, IR.occupancy(k))
    for n, k in pick("conv_wgrad_dma_kernel").items():                    # weight gradients: no scratch, >= 3 waves per SIMD
        assert k["scratch"] == 0 and k["vspill"] == 0 and IR.occupancy(k)[0] >= 3, (n, k)
    for n, k in pick("head_loss_cells_kernel").items():                   # the seam kernel: 47 KB of LDS, 3 workgroups per CU
        assert k["scratch"] == 0 and k["lds"] <= 48 * 1024 and IR.occupancy(k)[1] >= 3, (n, k)


def test_call_profile_proxy_counts_calls_and_keeps_results():
    """_lib._CallProfile (bench.py --host-profile: host time per C-ABI symbol): a transparent proxy -- same return values, one
    [calls, seconds] row per symbol, wrappers bound once."""
    from pixelssl_amd import _lib

    class Fake:
        def __init__(self):
            self.n = 0

        def pxl_a(self, x, y):
            self.n += 1
            return x + y

        def pxl_b(self):
            return 7

    h = Fake()
    p = _lib._CallProfile(h)
    assert p.pxl_a(2, 3) == 5 and p.pxl_a(1, 1) == 2 and p.pxl_b() == 7 and h.n == 2
    assert p.stats["pxl_a"][0] == 2 and p.stats["pxl_b"][0] == 1 and p.stats["pxl_a"][1] >= 0.0
    assert p.pxl_a is p.pxl_a                      # (bound on first use: no __getattr__ on the hot path)
    with pytest.raises(AttributeError):
        p.pxl_missing


def test_lazy_auxiliary_predictions_materialise_on_first_access():
    """ssl_cct._LazyPreds (resulter['ul_ad_preds'] when the decoders ran the fused seam): a sequence whose entries are produced from
    the decoder's low-resolution logits when somebody reads them, once."""
    from pixelssl_amd.ssl_algorithm.ssl_cct import _LazyPreds

    class Head:
        calls = 0

        def materialize(self, want_prob=True):
            assert want_prob is False
            Head.calls += 1
            return torch.full((1, 2), 3.0), None

    t = torch.zeros(1, 2)
    lp = _LazyPreds([t, Head(), Head()])
    assert len(lp) == 3 and Head.calls == 0
    assert lp[0] is t and float(lp[1].sum()) == 6.0 and Head.calls == 1
    assert lp[1] is lp[1] and Head.calls == 1
    assert [tuple(v.shape) for v in lp] == [(1, 2)] * 3 and Head.calls == 2
    assert len(lp[1:]) == 2


def test_graph_host_state_rolls_back_scalars_only():
    """graph.HostState (a failed capture attempt has already stepped the scheduler and the optimizer's counters; the
    eager run of the same iteration must not step them a second time)."""
    from pixelssl_amd.graph import HostState

    class Sched:
        def __init__(self):
            self.last_epoch, self.name, self.table = 3, "poly", [1, 2]

    sc = Sched()
    groups = [{"lr": 0.1, "params": [torch.zeros(1)]}, {"lr": 1.0, "params": []}]
    hs = HostState([sc, None], groups)
    hs.save()
    sc.last_epoch, sc.table = 4, [9]
    groups[0]["lr"], groups[1]["lr"] = 0.05, 0.5
    hs.restore()
    assert sc.last_epoch == 3 and sc.table == [9]            # scalars rolled back, containers left alone
    assert groups[0]["lr"] == 0.1 and groups[1]["lr"] == 1.0 and len(groups[0]["params"]) == 1
    sc.last_epoch = 7
    hs.restore()                                             # nothing saved: a no-op
    assert sc.last_epoch == 7
