"""Captured training steps (pixelssl_amd/graph.py): the iteration replayed from ONE hipGraph launch must be the eager iteration.

  * the *_hp entry points (per-step scalars read from device memory) are bit-identical to the entry points that carry the
    scalars in their arguments;
  * Mean Teacher on the conditioned 129-pixel fixture: eager run vs captured run (two eager warm-up iterations, one capture,
    replays) -- same losses every iteration and the same weights afterwards, to the engine's own run-to-run spread (its
    reductions are fp32 atomics; under PXL_DETERMINISTIC=1 the two runs agree far below any parity bar), AND both inside the
    reference's bars (the fixture comes from the reference's own SSLMT._train);
  * the learning rate / EMA coefficient / ramp-up weight really change from replay to replay (a replay that baked the
    capture-time scalars into its launches fails the fixture's later iterations: the poly-LR and the ramp move every step).
"""
import ctypes
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda"


@pytest.mark.gpu
def test_hyper_entry_points_are_bit_identical():
    from pixelssl_amd import _lib, ops
    from pixelssl_amd.graph import HyperBlock
    h = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 100003 // 4 * 4
    p0 = torch.randn(n, device=DEV, generator=g)
    gr = torch.randn(n, device=DEV, generator=g)
    b0 = torch.randn(n, device=DEV, generator=g)
    hb = HyperBlock(torch.device(DEV))
    hb.upload({"lr": 0.01234567, "alpha": 0.98765, "w": 0.3141592})
    torch.cuda.synchronize()
    assert hb.dev[:3].tolist() == [float(torch.tensor(v, dtype=torch.float32)) for v in (0.01234567, 0.98765, 0.3141592)]
    with pytest.raises(_lib.PixelHipError):
        hb.ptr("not-uploaded")
    # SGD
    pa, ba, pb, bb = p0.clone(), b0.clone(), p0.clone(), b0.clone()
    ops.sgd_step(pa, gr, ba, 0.01234567, 0.9, 5e-4)
    ops.sgd_step(pb, gr, bb, None, 0.9, 5e-4, lr_dev=hb.ptr("lr"))
    assert torch.equal(pa, pb) and torch.equal(ba, bb)
    # EMA
    ta, tb = p0.clone(), p0.clone()
    ops.ema_update(ta, gr, 0.98765)
    ops.ema_update(tb, gr, None, alpha_dev=hb.ptr("alpha"))
    assert torch.equal(ta, tb)
    # the fused seam, both kernels (cell-wise: DeepLab's x16 resize; row-wise: forced)
    B, hh, ww, Cp, C, H, W = 4, 9, 9, 32, 21, 129, 129
    for dt, code in ((torch.float32, 0), (torch.bfloat16, 1)):
        s_low = torch.randn(B, hh, ww, Cp, device=DEV, generator=g).to(dt)
        t_low = torch.randn(B, hh, ww, Cp, device=DEV, generator=g).to(dt)
        gt = torch.randint(0, 21, (2, H, W), device=DEV, generator=g).float()
        gt[0, :5] = 255
        ws_bytes = max(h.pxl_upsample_bwd_workspace(B, ww, C, H), B * hh * ww * (C + 4) * 4 + 64)
        for force in ("0", "1"):
            os.environ["PXL_HEAD_LOSS_CELLS"] = force
            try:
                outs = []
                for dev_w in (None, hb.ptr("w")):
                    dlow = torch.zeros(B, hh, ww, Cp, device=DEV, dtype=dt)
                    ws = torch.zeros(ws_bytes, device=DEV, dtype=torch.uint8)
                    sums = torch.zeros(2 * B + 1, device=DEV)
                    args = [code, B, hh, ww, Cp, C, H, W, 1, s_low.data_ptr(), t_low.data_ptr(), gt.data_ptr(), 255, 2, 0, B, 0.5]
                    if dev_w is None:
                        _lib.check(h.pxl_head_loss(*args, 0.3141592, dlow.data_ptr(), ws.data_ptr(), ws_bytes, sums.data_ptr(), None))
                    else:
                        _lib.check(h.pxl_head_loss_hp(*args, dev_w, dlow.data_ptr(), ws.data_ptr(), ws_bytes, sums.data_ptr(), None))
                    outs.append((dlow, sums))
                torch.cuda.synchronize()
                # (the gradient's accumulation over the cells is fp32 atomics in the cell-wise kernel: compare to rounding there)
                if force == "0":
                    assert torch.equal(outs[0][0], outs[1][0]), (dt, force)
                else:
                    assert torch.allclose(outs[0][0].float(), outs[1][0].float(), rtol=2e-2 if dt == torch.bfloat16 else 1e-5, atol=1e-7)
                assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-6)
                assert outs[0][0].float().abs().sum().item() > 0
            finally:
                os.environ.pop("PXL_HEAD_LOSS_CELLS", None)


def _run_mt(dtype, graph, fixture, iters=None, deterministic=False):
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from test_multistep import _fx, _args, _deeplab_state
    keep = {k: os.environ.get(k) for k in ("PXL_GRAPH", "PXL_GRAPH_STRICT", "PXL_DETERMINISTIC")}
    os.environ["PXL_GRAPH"] = "1" if graph else "0"
    os.environ["PXL_GRAPH_STRICT"] = "1"
    if deterministic:
        os.environ["PXL_DETERMINISTIC"] = "1"
    try:
        fx = _fx(fixture)
        args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=fx["rampup_iters"] / fx["max_iters"],
                     ema_decay=0.99)
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
        algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
        algo.s_model.train()
        algo.t_model.train()
        losses = []
        seeds = fx["data_seeds"][:iters] if iters else fx["data_seeds"]
        for i, s in enumerate(seeds):
            x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
            out, s_res, t_res = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        sg = getattr(algo, "_sgraph", None)
        info = dict(replays=sg.replays if sg is not None else 0, failed=sg.failed if sg is not None else None,
                    lr=[g["lr"] for g in algo.s_optimizer.param_groups], steps=algo.s_optimizer._steps_taken,
                    cur_iter=algo.s_lrer.cur_iter)
        # the resulters of a replayed step still materialise the prediction of THAT step
        pred = s_res["pred"][0] if hasattr(s_res, "__getitem__") else None
        sd_s = {k: v.detach().float().cpu().clone() for k, v in algo.s_model.module.model.state_dict().items()}
        sd_t = {k: v.detach().float().cpu().clone() for k, v in algo.t_model.module.model.state_dict().items()}
        return fx, losses, sd_s, sd_t, info, (pred.detach().float().cpu() if pred is not None else None)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_mt_replayed_iterations_equal_eager_iterations(dtype):
    from test_multistep import _check_losses, _check_weights, LOSS_TOL
    fx, le, se, te, ie, pe = _run_mt(dtype, graph=False, fixture="mt_cond_129.pt", deterministic=True)
    _, lg, sg_, tg, ig, pg = _run_mt(dtype, graph=True, fixture="mt_cond_129.pt", deterministic=True)
    n = len(le)
    assert ie["replays"] == 0 and ig["failed"] is None and ig["replays"] == n - 2, (ie, ig)
    # host bookkeeping of the replayed steps: scheduler position, learning rates, optimizer step count
    assert ig["cur_iter"] == ie["cur_iter"] and ig["lr"] == ie["lr"] and ig["steps"] == ie["steps"] == n
    worst = 0.0
    for i in range(n):
        for k in le[i]:
            if i == 1 and k == "cons_loss":       # ~1e-14 in both runs (the reference's is exactly 0): nothing to compare
                continue
            d = abs(le[i][k] - lg[i][k]) / max(abs(le[i][k]), 1e-6)
            worst = max(worst, d)
        print("mt graph-vs-eager %s iter %d:" % (dtype, i), lg[i], le[i])
        # ... and both are the reference's iteration (a replay that kept the capture-time learning rate, ramp-up weight or EMA
        # coefficient drifts off the fixture from the first replayed step on)
        ref = dict(fx["ref_per_iter"][i])
        if i == 1:          # (tests/test_multistep.py::test_mt_six_iterations: the reference's consistency loss is exactly 0 here)
            assert lg[i]["cons_loss"] <= (1e-12 if dtype == "fp32" else 1e-5)
            ref.pop("cons_loss")
        _check_losses("mt graph", i, lg[i], ref, dtype, loose=("cons",) if dtype == "bf16" else ())
    print("mt graph-vs-eager %s: worst relative loss difference %.3e" % (dtype, worst))
    assert worst <= (1e-4 if dtype == "fp32" else 6e-2), worst      # (bf16: the consistency term, 1e-3 of the loss, moved 3e-2)
    # graph-run weights against the eager run's, in units of the six-step update (the fixture's probes)
    from test_multistep import subsample
    wd = 0.0
    for sd_e, sd_g, ups in ((se, sg_, fx["student_updates"]), (te, tg, fx["teacher_updates"])):
        for k, u in ups.items():
            if u["update_l2"] > 1e-12:
                wd = max(wd, (subsample(sd_e[k]).double() - subsample(sd_g[k]).double()).norm().item() / u["update_l2"])
    print("mt graph-vs-eager %s: worst |graph - eager| / |update| %.3e" % (dtype, wd))
    assert wd <= (0.02 if dtype == "fp32" else 0.6), wd
    _check_weights("mt graph student " + dtype, sg_, fx["student_updates"], dtype)
    _check_weights("mt graph teacher " + dtype, tg, fx["teacher_updates"], dtype)
    if pe is not None and pg is not None:
        assert torch.isfinite(pg).all() and ((pe - pg).norm() / pe.norm()).item() < (1e-3 if dtype == "fp32" else 5e-2)


@pytest.mark.gpu
def test_a_step_with_another_shape_runs_eagerly_between_replays():
    """a short last batch (other shape) between replays: eager, on the same weights, and the replays go on afterwards"""
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from test_multistep import _fx, _args, _deeplab_state
    os.environ["PXL_GRAPH_STRICT"] = "1"
    os.environ["PXL_GRAPH"] = "1"
    try:
        fx = _fx("mt_cond_129.pt")
        args = _args(fx, "bf16", cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=1, ema_decay=0.99)
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
        algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
        algo.s_model.train()
        algo.t_model.train()
        vals = []
        for i in range(6):
            size = 97 if i == 4 else fx["size"]
            x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], size, fx["lbs"], seed=100 + i, block=fx["block"])
            out, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, 100)
            vals.append(float(out["s_task_loss"]))
        sg = algo._sgraph
        assert sg.failed is None and sg.replays == 3, (sg.failed, sg.replays)      # calls 3, 4 and 6 (call 5 had another shape)
        assert all(v == v and 0.5 < v < 6.0 for v in vals), vals
        assert algo.s_optimizer._steps_taken == 6 and algo.s_lrer.cur_iter == 7
    finally:
        os.environ.pop("PXL_GRAPH_STRICT", None)
        os.environ.pop("PXL_GRAPH", None)
