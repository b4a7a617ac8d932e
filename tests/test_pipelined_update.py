"""Parameter update pipelined behind the backward pass (nn/optimizer.py: PipelinedUpdate, csrc/net.cpp: pxl_net_set_update_hook).

  * bucket arithmetic: SGD + EMA + re-packing applied to arbitrary sub-ranges of the flat buffers equals -- bit for bit -- one
    whole-buffer SGD step, one whole-buffer EMA update and a full re-pack on the same gradients;
  * the executor hands out buckets that tile [0, #parameters) exactly, in descending order, the last one small;
  * Mean Teacher on the conditioned fixture: the pipelined run stays inside the reference's bars and agrees with the
    step-after-backward run to the engine's run-to-run spread.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda"


def _mt(dtype, monkeypatch, env):
    """(the switches are read when the algorithm / its executors are built -- the executors lazily, at the first step: they stay
    set for the whole test, monkeypatch restores them)"""
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from test_multistep import _fx, _args, _deeplab_state
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fx = _fx("mt_cond_129.pt")
    args = _args(fx, dtype, cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99)
    algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
    algo.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    algo.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    algo.s_model.train()
    algo.t_model.train()
    return fx, algo


def _steps(fx, algo, n=None):
    import torch_oracle as TO
    out = []
    for i, s in enumerate(fx["data_seeds"][:n] if n else fx["data_seeds"]):
        x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"])
        losses, _, _ = algo.train_step((x.to(DEV),), (gt.to(DEV),), i, fx["rampup_iters"])
        out.append({k: float(v) for k, v in losses.items()})
    torch.cuda.synchronize()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,fused", [("fp32", "0"), ("bf16", "0"), ("bf16", "1")], ids=["fp32", "bf16", "bf16-fused"])
def test_bucket_updates_equal_the_whole_buffer_step_bit_for_bit(dtype, fused, monkeypatch):
    """fused = PXL_FUSED_UPDATE=1: SGD + EMA + both networks' bf16 forward copies + the gradient memset in ONE kernel per bucket
    (csrc/optim.hip: pxl_sgd_ema_pack), the remaining layouts by pxl_net_pack_range(which = 2 | 4) -- the same bits"""
    monkeypatch.setenv("PXL_FUSED_UPDATE", fused)
    from pixelssl_amd import ops
    from pixelssl_amd.nn.optimizer import PipelinedUpdate
    fx, algo = _mt(dtype, monkeypatch, {"PXL_PIPE_UPDATE": "0", "PXL_GRAPH": "0"})
    _steps(fx, algo, 2)                      # plans, tuned tiles, momentum buffers that are not zero
    s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
    opt = algo.s_optimizer
    st, tt = s_core.flat, t_core.flat
    g = torch.Generator(device=DEV).manual_seed(11)
    grads = torch.randn(st.np, device=DEV, generator=g) * 1e-2
    p0, m0, t0 = st.params.clone(), st.momentum.clone(), tt.params.clone()
    s_plan = next(iter(s_core._plans.values()))
    t_plan = next(iter(t_core._plans.values()))
    alpha = 0.97
    # ---- reference: one whole-buffer step, EMA, full re-pack (into zeroed buffers: regions no pack writes -- the teacher holds
    # no data-gradient layouts -- are uninitialised memory otherwise)
    s_plan.packed.zero_(); t_plan.packed.zero_()
    st.grads.copy_(grads)
    opt.step()
    ops.ema_update(tt.params, st.params, alpha)
    s_core._cur, t_core._cur = s_plan, t_plan
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr
    check(lib().pxl_net_pack(s_plan.net, ptr(st.params), ptr(s_plan.packed), stream_ptr()))
    check(lib().pxl_net_pack(t_plan.net, ptr(tt.params), ptr(t_plan.packed), stream_ptr()))
    torch.cuda.synchronize()
    want = dict(p=st.params.clone(), m=st.momentum.clone(), t=tt.params.clone(), sp=s_plan.packed.clone(), tp=t_plan.packed.clone())
    # ---- the same through buckets the executor would hand out (op boundaries of this network, descending)
    st.params.copy_(p0); st.momentum.copy_(m0); tt.params.copy_(t0); st.grads.copy_(grads)
    s_plan.packed.zero_(); t_plan.packed.zero_()
    torch.cuda.synchronize()
    conv_lo = sorted({int(op.w_off[0]) for op in s_core._pb.ops if op.kind == 1})
    cuts = [conv_lo[k] for k in (len(conv_lo) - 1, len(conv_lo) - 7, len(conv_lo) // 2, len(conv_lo) // 5, 3, 1)]
    cuts = sorted(set(c for c in cuts if 0 < c < st.np), reverse=True)
    pipe = PipelinedUpdate(opt, s_core, t_core)
    assert pipe.fused == (fused == "1")
    try:
        pipe.arm(s_plan, t_plan, alpha, None)
        if pipe.fused:
            assert pipe._segs[3] >= 90, pipe._segs[3]          # ~100 of the 105 convolutions of DeepLab-v2 are plain casts
        # the buckets on a stream of their own, as the executor hands them out (its communication stream) -- never the null
        # stream: inside the full suite a bucket enqueued through ExternalStream(0) was twice seen to run ahead of the zero fill above
        bs = torch.cuda.Stream()
        hi = st.np
        for lo in cuts + [0]:
            pipe._on_bucket(lo, hi, bs.cuda_stream)
            hi = lo
        bs.synchronize()
        opt.step()                             # bookkeeping only
        torch.cuda.synchronize()
        assert pipe.buckets == len(cuts) + 1 and not pipe.pending and pipe.grads_clean
        assert torch.equal(st.params, want["p"]) and torch.equal(st.momentum, want["m"]) and torch.equal(tt.params, want["t"])
        for tag, got_pk, want_pk in (("student", s_plan.packed, want["sp"]), ("teacher", t_plan.packed, want["tp"])):
            bad = (got_pk != want_pk).nonzero().flatten()
            assert bad.numel() == 0, "%s packed weights differ in %d bytes, first at byte %d, last at %d of %d (got %s want %s)" % (
                tag, bad.numel(), int(bad[0]), int(bad[-1]), got_pk.numel(), got_pk[bad[:8]].tolist(), want_pk[bad[:8]].tolist())
        assert float(st.grads.abs().max()) == 0.0
        assert opt._steps_taken == 4
    finally:
        pipe.detach()


@pytest.mark.gpu
def test_the_executor_hands_out_buckets_that_tile_the_gradient_buffer(monkeypatch):
    fx, algo = _mt("bf16", monkeypatch, {"PXL_PIPE_UPDATE": "1", "PXL_GRAPH": "0", "PXL_UPDATE_BUCKET_MB": "8"})
    seen = []
    _steps(fx, algo, 1)
    pipe = algo._pipe
    assert pipe is not None
    orig = pipe._on_bucket

    def spy(lo, hi, stream):
        seen.append((lo, hi))
        return orig(lo, hi, stream)
    pipe.s_core.set_update_hook(spy, 8 * (1 << 20) // 4, 300000)
    _steps(fx, algo, 1)
    n = pipe.s_core.flat.np
    assert seen[0][1] == n and seen[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(seen, seen[1:])), seen
    assert len(seen) >= 5 and seen[-1][1] - seen[-1][0] <= 300000, seen       # the last bucket (stem + first stage) is small
    assert algo.s_model.module.model.update_buckets() == len(seen)
    assert algo.s_optimizer._steps_taken == 2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,fused", [("fp32", "0"), ("bf16", "0"), ("bf16", "1")], ids=["fp32", "bf16", "bf16-fused"])
def test_mt_with_the_pipelined_update_is_the_reference_iteration(dtype, fused, monkeypatch):
    from test_multistep import _check_losses, _check_weights
    env = {"PXL_GRAPH": "0", "PXL_DETERMINISTIC": "1", "PXL_FUSED_UPDATE": fused}
    fx, a0 = _mt(dtype, monkeypatch, dict(env, PXL_PIPE_UPDATE="0"))
    l0 = _steps(fx, a0)
    s0 = {k: v.detach().float().cpu() for k, v in a0.s_model.module.model.state_dict().items()}
    t0 = {k: v.detach().float().cpu() for k, v in a0.t_model.module.model.state_dict().items()}
    del a0
    fx, a1 = _mt(dtype, monkeypatch, dict(env, PXL_PIPE_UPDATE="1"))
    l1 = _steps(fx, a1)
    assert a1._pipe is not None and a1._pipe.buckets >= 2 and a1.s_optimizer._steps_taken == len(l1)
    assert a1._pipe.fused == (fused == "1")
    s1 = {k: v.detach().float().cpu() for k, v in a1.s_model.module.model.state_dict().items()}
    t1 = {k: v.detach().float().cpu() for k, v in a1.t_model.module.model.state_dict().items()}
    from test_multistep import subsample
    worst = 0.0
    for i in range(len(l0)):
        ref = dict(fx["ref_per_iter"][i])
        if i == 1:          # (tests/test_multistep.py::test_mt_six_iterations: the reference's consistency loss is exactly 0 here)
            assert l1[i]["cons_loss"] <= (1e-12 if dtype == "fp32" else 1e-5)
            ref.pop("cons_loss")
        _check_losses("mt pipelined", i, l1[i], ref, dtype, loose=("cons",) if dtype == "bf16" else ())
        worst = max([worst] + [abs(l0[i][k] - l1[i][k]) / max(abs(l0[i][k]), 1e-6) for k in l0[i] if not (i == 1 and k == "cons_loss")])
    print("mt pipelined-vs-sequential %s: worst relative loss difference %.3e" % (dtype, worst))
    assert worst <= (1e-4 if dtype == "fp32" else 2e-2), worst
    wd = 0.0
    for a, b, ups in ((s0, s1, fx["student_updates"]), (t0, t1, fx["teacher_updates"])):
        for k, u in ups.items():
            if u["update_l2"] > 1e-12:
                wd = max(wd, (subsample(a[k]).double() - subsample(b[k]).double()).norm().item() / u["update_l2"])
    print("mt pipelined-vs-sequential %s: worst |pipelined - sequential| / |update| %.3e" % (dtype, wd))
    assert wd <= (0.02 if dtype == "fp32" else 0.6), wd
    _check_weights("mt pipelined student " + dtype, s1, fx["student_updates"], dtype)
    _check_weights("mt pipelined teacher " + dtype, t1, fx["teacher_updates"], dtype)


@pytest.mark.gpu
def test_an_armed_pipeline_whose_hook_never_fires_still_updates_the_teacher(monkeypatch):
    """`if pipe is None` skipped the EMA whenever a PipelinedUpdate object EXISTED.  With the executor's hook
    removed (what a program without monotonic parameter offsets / a run without weight gradients amounts to) the iteration must
    be the ordinary one: whole-buffer SGD step, EMA update of the teacher, pipeline disarmed, no error."""
    fx, algo = _mt("bf16", monkeypatch, {"PXL_PIPE_UPDATE": "1", "PXL_GRAPH": "0"})
    _steps(fx, algo, 1)
    pipe = algo._pipe
    assert pipe is not None and algo.s_optimizer.last_step_pipelined
    s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
    s_core.set_update_hook(None)                       # the executor no longer calls back
    t_before, p_before = t_core.flat.params.clone(), s_core.flat.params.clone()
    _steps(fx, algo, 1)
    assert not algo.s_optimizer.last_step_pipelined
    assert not pipe.armed and not pipe.pending and not pipe.grads_clean
    assert not torch.equal(s_core.flat.params, p_before), "the ordinary SGD step did not run"
    assert not torch.equal(t_core.flat.params, t_before), "the teacher was not EMA-updated"
    # EMA of THIS step: t = a * t + (1 - a) * p with a = min(1 - 1 / (step + 1), ema_decay), step = 0 for _steps(.., 1) -> a = 0
    assert torch.allclose(t_core.flat.params, s_core.flat.params, rtol=0, atol=0)
    assert algo.s_optimizer._steps_taken == 2


@pytest.mark.gpu
def test_per_group_weight_decay_falls_back_when_the_pipeline_is_built(monkeypatch):
    """the fused kernel's preconditions (<= 8 learning-rate runs, one momentum / weight decay) were only
    checked inside the C callback during the backward pass, so a plug-in optimizer without weight decay on some group failed
    EVERY step.  Now PipelinedUpdate.__init__ drops to the per-group bucket kernels (fused = False) and the step works."""
    from pixelssl_amd.nn.optimizer import PipelinedUpdate
    fx, algo = _mt("bf16", monkeypatch, {"PXL_PIPE_UPDATE": "0", "PXL_GRAPH": "0"})
    _steps(fx, algo, 1)
    opt = algo.s_optimizer
    s_core, t_core = algo.s_model.module.model, algo.t_model.module.model
    if len(opt.param_groups) < 2:
        pytest.skip("the factory built one parameter group")
    opt.param_groups[-1]['weight_decay'] = 0.0
    pipe = PipelinedUpdate(opt, s_core, t_core)
    try:
        assert pipe.fused is False
        algo._pipe = pipe
        _steps(fx, algo, 1)                            # must not raise 'parameter-update hook failed'
        assert opt.last_step_pipelined and opt._steps_taken == 2
        monkeypatch.setenv("PXL_FUSED_UPDATE_STRICT", "1")
        with pytest.raises(ValueError):
            PipelinedUpdate(opt, s_core, t_core)
    finally:
        pipe.detach()
        algo._pipe = None
