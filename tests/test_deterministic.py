"""PXL_DETERMINISTIC=1: a bit-reproducible forward pass (csrc/net.cpp: one statistics replica per 64 pixel rows, replicas folded in
index order by one kernel, no split-K).  Two fresh executors on the same weights and input must produce the same bits -- logits and
BatchNorm running statistics -- and agree with the default (atomic-replica) mode to rounding."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _run(dtype, seed_state=3):
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    state = TO.init_deeplabv2_state(seed=seed_state, layers=(1, 1, 1, 1))
    x, _ = TO.synthetic_batch(4, 129, 4, seed=4, block=16)
    core = DeepLabV2Core(backbone=(1, 1, 1, 1), device="cuda:0", engine_dtype=dtype)
    core.autotune = False
    core.load_state_dict(state)
    core.train()
    with torch.no_grad():
        logits, _, _ = core(x.cuda())
    torch.cuda.synchronize()
    return logits.detach().float().cpu().clone(), core.flat.running.detach().cpu().clone()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_forward_is_bit_reproducible(dtype, monkeypatch):
    monkeypatch.setenv("PXL_DETERMINISTIC", "1")
    runs = [_run(dtype) for _ in range(3)]
    for lg, rs in runs[1:]:
        assert torch.equal(lg, runs[0][0]), (lg - runs[0][0]).abs().max().item()
        assert torch.equal(rs, runs[0][1]), (rs - runs[0][1]).abs().max().item()
    monkeypatch.delenv("PXL_DETERMINISTIC")
    lg, rs = _run(dtype)                      # the default mode: the same numbers up to the order of the statistics atomics
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(lg, runs[0][0]) < tol and rel(rs, runs[0][1]) < tol, (rel(lg, runs[0][0]), rel(rs, runs[0][1]))


def _run_step(dtype, seam):
    """one training step (forward, criterion, backward) of a shallow trunk: -> (loss, flat gradient buffer)"""
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd import functional as PF
    state = TO.init_deeplabv2_state(seed=5, layers=(1, 1, 2, 1))
    x, gt = TO.synthetic_batch(4, 129, 4, seed=6, block=16)
    core = DeepLabV2Core(backbone=(1, 1, 2, 1), device="cuda:0", engine_dtype=dtype)
    core.autotune = False
    core.load_state_dict(state)
    core.train()
    if seam:
        head = core.forward_deferred(x.cuda())
        ce, _, _ = PF.head_losses(head, None, gt.cuda(), 4, 0, 0, 0.25, 0.0, 255)
        loss = ce.mean()
        head.backward()
    else:
        logits, _, _ = core(x.cuda())
        loss = PF.cross_entropy_per_sample(logits, gt.cuda(), 255).mean()
        loss.backward()
    torch.cuda.synchronize()
    return float(loss), core.flat.grads.detach().cpu().clone(), core.flat.running.detach().cpu().clone()


@pytest.mark.parametrize("seam", [True, False], ids=["seam", "planes"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_backward_is_bit_reproducible(dtype, seam, monkeypatch):
    """Round 5: BatchNorm-backward sums on per-row-group replicas folded in index order, single-split weight gradients, ordered
    bias sums, the seam's row-wise kernel: three fresh executors give the same BITS in every parameter gradient (and, on the
    seam path, the same loss bits).  The non-seam path computes its loss VALUE with the stand-alone criterion kernel, whose sum
    is unordered -- the gradient does not depend on it."""
    monkeypatch.setenv("PXL_DETERMINISTIC", "1")
    runs = [_run_step(dtype, seam) for _ in range(3)]
    for loss, g, rs in runs[1:]:
        assert torch.equal(g, runs[0][1]), ((g - runs[0][1]).abs().max().item(), (g != runs[0][1]).sum().item())
        assert torch.equal(rs, runs[0][2])
        if seam:
            assert loss == runs[0][0], (loss, runs[0][0])
    assert runs[0][1].abs().sum().item() > 0
    monkeypatch.delenv("PXL_DETERMINISTIC")
    loss, g, rs = _run_step(dtype, seam)       # the default mode: the same numbers up to the order of the atomics
    rel = ((g - runs[0][1]).norm() / runs[0][1].norm()).item()
    # (fp32: a ReLU decision within an ulp of zero may flip; bf16: this randomly initialised trunk turns the different rounding
    # points of the two modes -- no BN-apply on load, no folded finalize -- into 0.28 of the gradient norm)
    # fp32 bar: five default-mode runs against one deterministic run measured 2.0e-3 .. 5.5e-3 (tools/det_probe.py, round 6, with
    # either head implementation: each flipped ReLU of this random trunk is ~1e-3 of the gradient norm) -- 1e-2 is above the spread
    assert rel < (1e-2 if dtype == torch.float32 else 0.5), rel
    assert abs(loss - runs[0][0]) < (1e-5 if dtype == torch.float32 else 2e-2) * abs(loss)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mean_teacher_iterations_are_bit_reproducible(dtype, monkeypatch):
    """the benchmarked algorithm end to end under PXL_DETERMINISTIC=1: three Mean-Teacher iterations (full ResNet-101, 129 x 129
    fixture weights) run twice from scratch -- every logged loss and both networks' parameters agree BIT FOR BIT"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from test_multistep import _fx, _args, _deeplab_state
    monkeypatch.setenv("PXL_DETERMINISTIC", "1")
    monkeypatch.setenv("PXL_AUTOTUNE", "0")       # (tile choice changes the summation order inside a convolution)
    fx = _fx("mt_cond_129.pt")

    def run():
        args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99)
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
        algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
        algo.s_model.train()
        algo.t_model.train()
        vals = []
        for i, s in enumerate(fx["data_seeds"][:3]):
            x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
            out, _, _ = algo.train_step((x.to("cuda"),), (gt.to("cuda"),), i, fx["rampup_iters"])
            vals.append(tuple(float(v) for v in out.values()))
        torch.cuda.synchronize()
        return vals, algo.s_model.module.model.flat.params.detach().cpu().clone(), algo.t_model.module.model.flat.params.detach().cpu().clone()
    a, b = run(), run()
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), ((a[1] - b[1]).abs().max().item(), (a[2] - b[2]).abs().max().item())
