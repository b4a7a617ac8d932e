#!/usr/bin/env python
"""Throughput of the hot path: one Mean-Teacher `sseg` training step (BASELINE.json configs[1]):
DeepLab-v2 / ResNet-101, 8 x 513 x 513 synthetic crops per GPU (4 labeled + 4 unlabeled), 21
classes, bf16 engine (fp32 accumulate, fp32 BN statistics, fp32 master weights / optimizer).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N [--algo gct]        # no launcher: bench.py starts the N ranks itself (the line above)

Prints ONE JSON line on rank 0 (contract in the task statement).  A "step" is the full reference
iteration (ssl_mt.py:131-220): student fwd + CE, teacher fwd (no-grad, train-mode BN), MSE
consistency, backward, fused SGD step, EMA teacher update, poly-LR step; inputs are resident in HBM
before the timed region.  `roofline` is measured live with HIP events bracketing every launch of
the dominant contraction kernel on its launch stream; `cpu_baseline` times the CPU oracle
(oracle/torch_oracle.py, a port pinned bit-exact to the reference) on a bounded sample; `miou_vs_ref`
scores the engine's and the oracle's predictions of the same weights with the reference's mIoU.
Only those two legs touch oracle/ (the checker); the timed region and its inputs do not.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}    # /opt/skills/guides/MI355X_MICROARCH.md (dense)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # (SURVEY.md 8d protocol: >= 10 warm-up and >= 50 timed iterations when the caller names nothing else: 0.75 s of GPU time)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--algo", default="mt", choices=["mt", "suponly", "adv", "gct", "cct"])
    p.add_argument("--size", type=int, default=513)
    p.add_argument("--lbs", type=int, default=4, help="labeled samples per GPU")
    p.add_argument("--ubs", type=int, default=4, help="unlabeled samples per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-events", action="store_true")
    p.add_argument("--cpu-sample-steps", type=int, default=10, help="timed steps of the SupOnly B=2 CPU baseline (SURVEY.md 8d)")
    p.add_argument("--cpu-warmup", type=int, default=3)
    p.add_argument("--cpu-budget-s", type=float, default=45.0, help="the CPU legs stop adding timed steps past this budget")
    p.add_argument("--no-miou", action="store_true")
    p.add_argument("--host-profile", action="store_true",
                   help="cProfile of the timed steps (top functions by own time -> stderr): where a host-bound step spends its time")
    p.add_argument("--no-seq-leg", action="store_true", help="skip the one-kernel-at-a-time roofline leg (stream overlap off, per-launch events)")
    p.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32_parity_mode leg (the engine mode that meets the 1e-3 parity bar)")
    p.add_argument("--fp32-steps", type=int, default=10)
    p.add_argument("--no-conditioned", action="store_true",
                   help="keep the raw reference initialisers (random-init ResNet-101 + x10 head lr diverges within ~20 steps; "
                        "default: bottleneck-output BN gammas x 0.1, the conditioning of the parity fixtures)")
    p.add_argument("--no-scaling-legs", action="store_true", help="multi-rank runs: skip the GCT / exchange-timing legs after the MT line")
    p.add_argument("--no-fixture-parity", action="store_true", help="skip the parity_vs_fixture leg (mt_cond_513.pt replay, both dtypes)")
    p.add_argument("--decoders", type=int, default=7, choices=[7, 11],
                   help="CCT: 7 = one decoder of each kind (BASELINE.json config 5), 11 = the shipped script's setting "
                        "(task/sseg/script/pspnet_pascalvoc_1-8_sslcct.py:27-33: 1 VAT, 2 DropOut, 2 G-Cutout, 1 context, 1 object, 2 F-drop, 2 F-noise)")
    return p.parse_args()


def make_args(a, world):
    ns = argparse.Namespace(
        backbone="resnet101", output_stride=16, num_classes=21, freeze_bn=False, engine_dtype=a.dtype,
        # proxy.py:258-261: lr and batch sizes are multiplied by #GPUs; per rank we keep the per-GPU batch
        lr=2.5e-4 * world, momentum=0.9, weight_decay=5e-4, dampening=-1, nesterov=False, power=-1,
        last_epoch=-1, epochs=20, iters_per_epoch=1000, ignore_index=255, labeled_batch_size=a.lbs,
        unlabeled_batch_size=a.ubs if a.algo != "suponly" else 0, ignore_unlabeled=a.algo == "suponly",
        batch_size=a.lbs + (a.ubs if a.algo != "suponly" else 0), gpus=1,
        # AdvSSL hyper-parameters of the shipped script (task/sseg/script/deeplabv2_pascalvoc_1-8_ssladv.py:23-28)
        adv_for_labeled=True, labeled_adv_scale=0.01, unlabeled_adv_scale=0.001, discriminator_lr=1e-4 * world,
        discriminator_power=0.9, unlabeled_for_discriminator=True, discriminator_scale=1.0,
        # GCT hyper-parameters of the shipped script (task/sseg/script/deeplabv2_pascalvoc_1-8_sslgct.py:21-33)
        im_size=a.size, ssl_mode="gct", fc_ssl_scale=1.0, dc_ssl_scale=100.0, dc_threshold=0.6, dc_rampup_epochs=3,
        fd_lr=1e-4 * world, fd_scale=10.0, mu=0.5, nu=1,
        is_epoch_lrer=False, log_freq=10 ** 9, task="sseg", cons_for_labeled=False, cons_scale=1.0,
        cons_rampup_epochs=3, ema_decay=0.99, gaussian_noise_std=None)
    if a.algo == "cct":
        # CCT hyper-parameters of the shipped script (task/sseg/script/pspnet_pascalvoc_1-8_sslcct.py:23-33) with the
        # K = 7 decoders BASELINE.json's config names (one of each kind)
        ns.models = {"model": "pspnet"}
        ns.cons_scale, ns.cons_rampup_epochs, ns.ad_lr_scale = 30.0, 5, 10.0
        ns.vat_dec_num = ns.drop_dec_num = ns.cut_dec_num = ns.context_dec_num = ns.object_dec_num = 1
        ns.fd_dec_num = ns.fn_dec_num = 1
        if a.decoders == 11:        # the shipped script's eleven decoders (pspnet_pascalvoc_1-8_sslcct.py:27-33)
            ns.drop_dec_num = ns.cut_dec_num = ns.fd_dec_num = ns.fn_dec_num = 2
        ns.vat_dec_xi, ns.vat_dec_eps, ns.drop_dec_rate, ns.drop_dec_spatial = 1e-6, 2.0, 0.5, True
        ns.cut_dec_erase, ns.fn_dec_uniform = 0.4, 0.3
    return ns


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _timed_cpu_steps(step, warmup, steps, budget_s):
    """warm-up + up to `steps` timed calls; stops early (never below 2 timed steps) when the budget is used up"""
    for _ in range(warmup):
        step()
    done, t0 = 0, time.time()
    while done < steps:
        step()
        done += 1
        if done >= 2 and time.time() - t0 > budget_s:
            break
    return done, time.time() - t0


def cpu_baseline(a):
    """Reference CPU path (kind "port": the oracle, pinned bit-exact to the reference).  Two legs on this box's host cores:
    `cpu_baseline` = the benchmarked workload (MT step, 2 + 2 images at 513 x 513) on a bounded sample, and
    `config1` inside it = BASELINE.json configs[0] / SURVEY.md 8d: SupOnly, B = 2, 3 warm-up + 10 timed iterations."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_oracle as TO
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    hp = dict(max_iters=20000, cons_rampup_iters=3000)
    # ---- config 1: SupOnly, DeepLab-v2, B = 2
    tr = TO.OracleTrainer(TO.init_deeplabv2_state(seed=0), hp)
    x, gt = TO.synthetic_batch(2, a.size, 2, seed=5)
    n1, dt1 = _timed_cpu_steps(lambda: tr.suponly_step(x, gt), a.cpu_warmup, a.cpu_sample_steps, a.cpu_budget_s)
    config1 = {"value": round(2 * n1 / dt1, 4), "unit": "img/s", "workload": "SupOnly sseg, DeepLab-v2/ResNet-101, 2x%dx%d" % (a.size, a.size),
               "sample": "%d warm-up + %d timed iterations" % (a.cpu_warmup, n1), "s_per_iter": round(dt1 / n1, 3)}
    # ---- the benchmarked workload on a bounded batch
    lbs, ubs = (2, 2) if a.algo == "mt" else (2, 0)
    if a.algo == "mt":
        tr = TO.OracleTrainer(TO.init_deeplabv2_state(seed=0), hp, teacher_state=TO.init_deeplabv2_state(seed=1))
        x, gt = TO.synthetic_batch(lbs + ubs, a.size, lbs, seed=5)
        n2, dt2 = _timed_cpu_steps(lambda: tr.mt_step(x, gt, lbs), 1, 3, a.cpu_budget_s / 2)
    else:
        n2, dt2 = n1, dt1
    return {"value": round((lbs + ubs) * n2 / dt2, 4), "unit": "img/s", "cores": threads, "kind": "port",
            "cpu": "%s, %d logical cores (%d torch threads)" % (_cpu_name(), cores, threads),
            "sample": "%d timed %s steps (after %d warm-up) of the CPU oracle at %dx%d, batch %d+%d, fp32, torch %s"
                      % (n2, "MT" if a.algo == "mt" else "SupOnly (DeepLab-v2)", 1 if a.algo == "mt" else a.cpu_warmup,
                         a.size, a.size, lbs, ubs, torch.__version__),
            "config1": config1}


def miou_vs_oracle(core, a):
    """BASELINE.json metric, second half ("mIoU vs ref"): the engine's eval-mode prediction (running BN statistics, the
    benchmarked precision) and the CPU oracle's on the SAME weights and the same synthetic validation batch, scored with
    the reference's confusion-matrix mIoU (task/sseg/func.py:36-80).  Random-init weights: the absolute value means
    nothing, the DIFFERENCE is the parity figure."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_oracle as TO
    import metrics_oracle as MO
    from pixelssl_amd import functional as PF
    from pixelssl_amd.utils.synthetic import synthetic_batch
    x, gt = synthetic_batch(2, a.size, 2, seed=4242)
    state = {k: v.detach().float().cpu().clone() for k, v in core.state_dict().items()}
    was_training = core.training
    core.eval()
    with torch.no_grad():
        logits, prob, _ = core(x.to(core.flat.params.device))
        cm_e = PF.confusion_matrix(prob, gt.to(prob.device), 21).cpu().numpy()
    core.train(was_training)
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        o_logits, o_prob, _, _ = TO.deeplabv2_forward(state, x, train=False)
    # (the 64 OpenMP workers of the oracle pass keep spinning for a while after their last parallel region; the legs that follow
    # are timed -- hand the cores back)
    torch.set_num_threads(threads_before)
    cm_o = MO.confusion_matrix(o_prob.numpy(), gt.numpy(), 21)
    me, mo = MO.metrics(cm_e), MO.metrics(cm_o)
    agree = float((logits.argmax(1).cpu() == o_logits.argmax(1)).float().mean())
    return {"engine": round(me["mIoU"], 6), "oracle": round(mo["mIoU"], 6), "abs_diff": round(abs(me["mIoU"] - mo["mIoU"]), 6),
            "argmax_agreement": round(agree, 5), "pixels": int(cm_o.sum()),
            "note": "eval-mode forward of the trained student (%s engine) vs the fp32 CPU oracle on the same weights, 2x%dx%d "
                    "synthetic validation batch" % (a.dtype, a.size, a.size)}


def build_algo(a, args):
    """-> (algorithm, executor cores) of the workload `a.algo` (plugin API: pixelssl_amd.ssl_algorithm.*)"""
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    factories = ({"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)}, {"model": plr.polynomiallr(args)},
                 {"model": P.sseg.criterion.sseg_criterion()})
    if a.algo == "mt":
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, *factories, None)
        cores = [algo.s_model.module.model, algo.t_model.module.model]
        algo.s_model.train()
        algo.t_model.train()
    elif a.algo == "gct":
        algo = P.ssl_algorithm.ssl_gct.ssl_gct(args, *factories, P.sseg.func.task_func()(args))
        cores = [algo.l_model.module.model, algo.r_model.module.model, algo.fd_model.module.core]
        for m in (algo.l_model, algo.r_model, algo.fd_model):
            m.train()
    elif a.algo == "adv":
        algo = P.ssl_algorithm.ssl_adv.ssl_adv(args, *factories, P.sseg.func.task_func()(args))
        cores = [algo.model.module.model, algo.d_model.module.core]
        algo.model.train()
        algo.d_model.train()
    elif a.algo == "cct":
        factories = ({"model": P.sseg.model.pspnet()},) + factories[1:]
        algo = P.ssl_algorithm.ssl_cct.ssl_cct(args, *factories, P.sseg.func.task_func()(args))
        cores = [algo.main_model.model] + [d.upsample for d in algo.auxiliary_decoders]
        algo.model.train()
    else:
        algo = P.ssl_algorithm.ssl_null.ssl_null(args, *factories, None)
        cores = [algo.model.module.model]
        algo.model.train()

    return algo, cores


def condition(cores):
    """The conditioning of the parity fixtures (oracle/torch_oracle.py: condition_state; restated here on the engine's own
    parameters -- the timed region touches nothing under oracle/): every bottleneck-output BatchNorm gamma x 0.1.  With the
    raw initialisers a random-init ResNet-101 under the shipped x10 head learning rate diverges within ~20 steps (round 3:
    task loss 3.1 -> 33); conditioned, the bench trajectory is the one tests/golden/mt_cond_513.pt pins."""
    import torch
    for c in cores:
        touched = False
        with torch.no_grad():
            for name, prm in c.named_parameters():
                if name.endswith("bn3.weight"):
                    prm.mul_(0.1)
                    touched = True
        if touched and hasattr(c, "mark_params_changed"):
            c.mark_params_changed()


def fixture_parity(a, fence):
    """parity_vs_fixture: the first FOUR iterations of this very workload (MT, 4 + 4 crops at 513 x 513, shipped
    hyper-parameters) replayed from the fixture's initial weights and data seeds on the engine, in both precisions, against
    the losses the REFERENCE's own SSLMT._train logged (tests/golden/mt_cond_513.pt, oracle/make_golden_conditioned.py
    mt513).  A checker leg after the timed region (it uses oracle/ for the initial weights and the synthetic batches, like
    tests/test_multistep.py::test_mt_at_the_baseline_configuration, whose bars are 1e-3 fp32 / 1e-2 bf16)."""
    import copy
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_oracle as TO
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "mt_cond_513.pt"), weights_only=False)
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {"fixture": "tests/golden/mt_cond_513.pt (4 iterations of the reference's SSLMT._train, 4 + 4 crops at 513 x 513)"}
    for dtype in ("fp32", "bf16"):
        b = copy.copy(a)
        b.dtype, b.lbs, b.ubs, b.algo = dtype, fx["lbs"], fx["ubs"], "mt"
        args = make_args(b, 1)
        args.iters_per_epoch, args.epochs = fx["max_iters"], 1
        algo = P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                            {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)
        algo.s_model.module.model.load_state_dict(TO.condition_state(TO.init_deeplabv2_state(seed=fx["weight_seed"]), fx["gamma3"]))
        algo.t_model.module.model.load_state_dict(TO.condition_state(TO.init_deeplabv2_state(seed=fx["weight_seed"] + 1), fx["gamma3"]))
        algo.s_model.train()
        algo.t_model.train()
        worst, worst_cons, per_iter = 0.0, 0.0, []
        for i, sd in enumerate(fx["data_seeds"]):
            x, gt = TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=sd, block=fx["block"])
            got, _, _ = algo.train_step((x.to(dev),), (gt.to(dev),), i, fx["rampup_iters"])
            ref = fx["ref_per_iter"][i]
            errs = {}
            for k, v in ref.items():
                if abs(v) < 1e-9:                 # (iteration 1: the reference's consistency loss is exactly 0)
                    continue
                errs[k] = abs(float(got[k]) - v) / abs(v)
            worst = max([worst] + [v for k, v in errs.items() if "cons" not in k])
            worst_cons = max([worst_cons] + [v for k, v in errs.items() if "cons" in k])
            per_iter.append({k: round(float(got[k]), 6) for k in ref})
        out[dtype] = {"max_rel_task_loss_error": float("%.3g" % worst), "max_rel_cons_loss_error": float("%.3g" % worst_cons), "losses": per_iter}
        # ... and the prediction of the weights those four iterations left behind, engine (this precision) vs fp32 CPU oracle
        mv = miou_vs_oracle(algo.s_model.module.model, b)
        out[dtype]["argmax_agreement"] = mv["argmax_agreement"]
        out[dtype]["miou_abs_diff"] = mv["abs_diff"]
        del algo
        torch.cuda.empty_cache()
    out["reference_losses"] = [{k: round(float(v), 6) for k, v in r.items()} for r in fx["ref_per_iter"]]
    out["bars"] = {"fp32": 1e-3, "bf16": 1e-2, "bf16_cons": 1e-1,      # (the consistency term is 1e-3 .. 1e-2 of the task loss: tests give it 10 x)
                   "argmax": "north_star asks for bit-exact arg-max indices; the tests assert it on the pixels whose reference margin "
                             "exceeds the numeric noise (the reference's own fp32-vs-fp64 arg-max differs on 3.2e-4 of the pixels) and "
                             "> 0.998 overall for fp32 (tests/test_parity_513.py:96-101)"}
    return out


def make_step(a, args, algo, batches):
    def one_step(it):
        inp, gt = batches[it % len(batches)]
        if a.algo == "mt":
            return algo.train_step(inp, gt, it, 3 * args.iters_per_epoch)[0]
        if a.algo == "gct":
            return algo.train_step(inp, gt, it, 3 * args.iters_per_epoch)
        if a.algo == "cct":
            return algo.train_step(inp, gt, it, 5 * args.iters_per_epoch)[0]
        return algo.train_step(inp, gt)[0]
    return one_step


def sequential_leg(a, world, batches, fence):
    """Kernel quality without co-runners: the same workload on a second instance of the algorithm with every stream overlap
    switched off (teacher, weight gradients, packing all on the main stream), per-launch HIP events around every contraction
    launch -> TFLOP/s of the dominant kernel family when each launch has the GPU to itself.  The headline `roofline` figures
    come from the overlapped step (two or three kernels share the CUs, so a launch's event pair also spans its co-runner);
    this leg says what the kernels themselves reach."""
    import torch
    switches = ("PXL_TEACHER_STREAM", "PXL_SIDE_STREAM", "PXL_PACK_STREAM", "PXL_GCT_STREAMS", "PXL_ADV_STREAMS",
                "PXL_CCT_STREAMS", "PXL_CCT_SPLIT_BACKWARD")
    saved = {k: os.environ.get(k) for k in switches}
    for k in switches:
        os.environ[k] = "0"
    try:
        args = make_args(a, world)
        algo, cores = build_algo(a, args)
        if not a.no_conditioned:
            condition(cores)
        step = make_step(a, args, algo, batches)
        for it in range(2):
            step(it)
        for c in cores:
            c.profile(True)
        fence()
        t0 = time.perf_counter()
        n_steps = min(a.steps, 5)
        for it in range(2, 2 + n_steps):
            step(it)
        fence()
        dt = time.perf_counter() - t0
        res = {}
        for kind, name in ((0, "conv_dma/conv_igemm (fwd+dgrad)"), (1, "conv_wgrad_dma/conv_wgrad")):
            ms = n = fl = 0.0
            for c in cores:
                m_, n_, f_ = c.profile_read(kind)
                ms, n, fl = ms + m_, n + n_, fl + f_
            if n:
                res[name] = {"launches": int(n), "avg_us": round(1e3 * ms / n, 3), "total_ms_per_step": round(ms / n_steps, 3),
                             "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                             "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS[a.dtype], 4)}
        for c in cores:
            c.profile(False)
        del algo, cores
        torch.cuda.empty_cache()
        return {"ms_per_step_with_kernel_events": round(1e3 * dt / n_steps, 3), "steps": n_steps, "kernels": res,
                "note": "every stream overlap switched off: one kernel at a time, per-launch HIP events = the kernels' own durations"}
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def fp32_parity_leg(a, world, batches, fence):
    """The SAME workload on the fp32 engine -- the mode that meets north_star's 1e-3 parity bar (tests/test_parity_513.py,
    tests/test_multistep.py::test_mt_at_the_baseline_configuration[fp32]) -- timed like the headline figure (barrier +
    synchronize on both sides, K steps), reported next to it: img/s and the whole-step fraction of the 157.3 TFLOP/s
    fp32 MFMA peak (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation)."""
    import copy
    import torch
    a32 = copy.copy(a)
    a32.dtype = "fp32"
    args = make_args(a32, world)
    algo, cores = build_algo(a32, args)
    if not a.no_conditioned:
        condition(cores)
    step = make_step(a32, args, algo, batches)
    warm = 4
    for it in range(warm):
        step(it)
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.fp32_steps + 1)]
    host = []
    t0 = time.perf_counter()
    evs[0].record()
    for k, it in enumerate(range(warm, warm + a.fp32_steps)):
        step(it)
        evs[k + 1].record()
        host.append(time.perf_counter())
    fence()
    dt = time.perf_counter() - t0
    # the same region step by step: GPU time between the events behind consecutive steps, host time to enqueue each step
    per_step = [round(evs[k].elapsed_time(evs[k + 1]), 3) for k in range(a.fp32_steps)]
    per_step_host = [round(1e3 * (h - (host[k - 1] if k else t0)), 3) for k, h in enumerate(host)]
    per_gpu = a.lbs + (a.ubs if a.algo != "suponly" else 0)
    flop_img = {"mt": 449.9e9, "adv": 435.0e9, "gct": 1291.2e9, "suponly": 337.1e9, "cct": 452.6e9}[a.algo]
    val = per_gpu * world * a.fp32_steps / dt
    del algo, cores
    torch.cuda.empty_cache()
    return {"dtype": "fp32", "value": round(val, 3), "unit": "img/s", "ms_per_step": round(1e3 * dt / a.fp32_steps, 3),
            "steps": a.fp32_steps, "warmup": warm, "per_step_gpu_ms": per_step, "per_step_host_enqueue_ms": per_step_host,
            "peak": MFMA_PEAK_TFLOPS["fp32"],
            "step_mfma_frac": round(val * flop_img / world / (MFMA_PEAK_TFLOPS["fp32"] * 1e12), 4) if a.size == 513 else None,
            "note": "fp32 engine (fp32 operands, products and accumulation on v_mfma_f32_32x32x2_f32): the mode whose logits / "
                    "losses / weights are within 1e-3 of the reference (parity tests); same workload, same timing protocol"}


def scaling_legs(a, world, batches, fence, dev, mt_ms_per_step):
    """Multi-rank runs only (the driver's SCALE command is `python bench.py --gpus N`, which times MT): what a first run across
    real GPUs has to show next to the MT line, each leg bounded to a few seconds and none of them able to take the MT line down
    (every leg is wrapped; a failure is reported as a string).
      scaling_gct            the workload north_star's ">= 6x at 8 GPUs" target is quoted on (GCT, BASELINE.json configs[2]):
                             img/s over all ranks, ms/step, Sync-BN exchanges per step
      sync_bn_exchange_us    one peer-mapped statistics exchange (2048 floats), HIP events around 200 back-to-back launches
      grad_allreduce         the flat 178 MB gradient all-reduce alone (RCCL from C), and the MT step with the gradient exchange
                             switched OFF: ms_per_step minus that = what the overlapped exchange still costs (`exposed_ms`)"""
    import copy
    import ctypes
    import torch
    from pixelssl_amd import dist as pdist, _lib
    out = {}
    per_gpu = a.lbs + a.ubs
    # ---- one Sync-BN exchange over the peer-mapped buffers
    try:
        ctxs = pdist._peer["ctxs"]
        if ctxs:
            h = _lib.lib()
            v = torch.ones(2048, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(20):
                h.pxl_peer_allreduce_sum(ctxs[0], v.data_ptr(), 2048, st)
                v.fill_(1.0)
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                h.pxl_peer_allreduce_sum(ctxs[0], v.data_ptr(), 2048, st)
            e1.record()
            e1.synchronize()
            out["sync_bn_exchange_us"] = round(1e3 * e0.elapsed_time(e1) / 200, 2)
        else:
            out["sync_bn_exchange_us"] = None
    except Exception as e:       # noqa: BLE001
        out["sync_bn_exchange_us"] = "failed: %s" % e
    # ---- the gradient all-reduce alone, and the MT step without it
    try:
        args = make_args(a, world)
        algo, cores = build_algo(a, args)
        if not a.no_conditioned:
            condition(cores)
        step = make_step(a, args, algo, batches)
        for it in range(2):
            step(it)
        g = cores[0].flat.grads
        comm = pdist._native.get("grad_comm")
        if comm is not None:
            st = torch.cuda.current_stream().cuda_stream
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                _lib.lib().pxl_comm_allreduce_sum(comm, g.data_ptr(), g.numel(), st)
            e1.record()
            e1.synchronize()
            full_ms = e0.elapsed_time(e1) / 5
        else:
            full_ms = None
        for c in cores:                      # gradient exchange off: every rank trains on its own gradients (timing only)
            c.set_grad_sync(None, None, 1, 0)
            c._post_backward_hook = None
        for it in range(2, 4):
            step(it)
        fence()
        t0 = time.perf_counter()
        for it in range(4, 9):
            step(it)
        fence()
        no_sync_ms = 1e3 * (time.perf_counter() - t0) / 5
        t = torch.tensor([no_sync_ms], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        no_sync_ms = t.item()
        out["grad_allreduce"] = {"full_buffer_ms": None if full_ms is None else round(full_ms, 3), "bytes": int(g.numel() * 4),
                                 "mt_ms_per_step_without_gradient_exchange": round(no_sync_ms, 3),
                                 "exposed_ms": round(mt_ms_per_step - no_sync_ms, 3)}
        del algo, cores, step
        torch.cuda.empty_cache()
    except Exception as e:       # noqa: BLE001
        out["grad_allreduce"] = "failed: %s" % e
    # ---- GCT
    try:
        b = copy.copy(a)
        b.algo = "gct"
        args = make_args(b, world)
        algo, cores = build_algo(b, args)
        if not a.no_conditioned:
            condition(cores)
        step = make_step(b, args, algo, batches)
        for it in range(2):
            step(it)
        fence()
        x0 = pdist.peer_exchanges()
        t0 = time.perf_counter()
        n = 5
        for it in range(2, 2 + n):
            step(it)
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
        out["scaling_gct"] = {"value": round(per_gpu * world * n / dt, 3), "unit": "img/s", "ms_per_step": round(1e3 * dt / n, 3), "steps": n,
                              "warmup": 2, "n_gpus": world, "rccl_ranks": pdist.rccl_ranks(), "peer_contexts": pdist.peer_contexts(),
                              "sync_bn_exchanges_per_step": round((pdist.peer_exchanges() - x0) / n, 1),
                              "workload": "GCT sseg, dual DeepLab-v2/ResNet-101 + flaw detector, %dx%dx%d per GPU" % (per_gpu, a.size, a.size)}
        pdist.check_peers()
        del algo, cores, step
        torch.cuda.empty_cache()
    except Exception as e:       # noqa: BLE001
        out["scaling_gct"] = "failed: %s" % e
    return out


def respawn_under_launcher(a):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks ourselves (one process per GPU, the launch
    line of the docstring) and pass their output through -- the command cannot silently measure ONE GPU.  The reference's
    single command drives every visible GPU as well (pixelssl/nn/func.py:54-62: DataParallel over all of them)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        respawn_under_launcher(a)
    import torch
    import torch.distributed as dist
    import pixelssl_amd as P
    from pixelssl_amd import dist as pdist
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    from pixelssl_amd.utils.synthetic import synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        # (main() re-launches `--gpus N` under torch.distributed.run when no launcher environment is present, so this is a
        # launcher that started a different number of ranks than the command line asks for: refuse, do not measure the wrong job)
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or run `python bench.py --gpus %d` "
                         "without a launcher: it spawns the ranks itself)" % (a.gpus, world, a.gpus, a.gpus))
    if world > 1 and os.environ.get("PXL_FORCE_DEVICE") is None and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(pdist.local_device())
    pdist.init_from_env("nccl")
    # experiment switch: k streams created (and used once) before anything else shifts which hardware queue every later
    # stream of the process lands on (HIP deals streams onto GPU_MAX_HW_QUEUES = 4 queues; streams that share one serialize)
    _dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("PXL_DUMMY_STREAMS", "0")))]
    for _s in _dummies:
        with torch.cuda.stream(_s):
            torch.zeros(1, device="cuda")
    dev = pdist.local_device()

    args = make_args(a, world)
    algo, cores = build_algo(a, args)
    if not a.no_conditioned:
        condition(cores)

    # synthetic data, resident in HBM before timing; 4 distinct batches cycled
    per_gpu = a.lbs + (a.ubs if a.algo != "suponly" else 0)
    batches = []
    for i in range(4):
        x, gt = synthetic_batch(per_gpu, a.size, a.lbs, seed=1234 + rank * 1000 + i)
        batches.append(((x.to(dev),), (gt.to(dev),)))

    one_step = make_step(a, args, algo, batches)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(a.warmup):
        one_step(it)
    fence()
    # ---- the timed region: exactly K steps, nothing but the training iteration inside
    prof = None
    if a.host_profile:
        import cProfile
        from pixelssl_amd import _lib as plib
        call_stats = plib.enable_call_profile()
        one_step(a.warmup)                     # (binds the timing wrappers outside the timed region)
        fence()
        for v in call_stats.values():
            v[0], v[1] = 0, 0.0
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    last = None
    # host time to ENQUEUE a step, taken over the first <= 8 steps: the hardware queues hold that many, beyond them the
    # runtime blocks the host on the GPU and the figure turns into the step time (one perf_counter read, no synchronisation)
    n_host = min(8, a.steps)
    host_mark = None
    for k, it in enumerate(range(a.warmup, a.warmup + a.steps)):
        last = one_step(it)
        if k + 1 == n_host:
            host_mark = time.perf_counter()
    host_enqueue = (host_mark - t0) * a.steps / n_host
    fence()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        st = pstats.Stats(prof, stream=buf)
        st.sort_stats("tottime").print_stats(45)
        st.sort_stats("cumulative").print_stats(60)
        sys.stderr.write(buf.getvalue())
        tot = sum(v[1] for v in call_stats.values())
        sys.stderr.write("host time inside the C-ABI per step (of %.3f ms/step wall; under cProfile): %.3f ms over %d calls\n"
                         % (1e3 * elapsed / a.steps, 1e3 * tot / a.steps, sum(v[0] for v in call_stats.values()) // a.steps))
        for name, (c, sec) in sorted(call_stats.items(), key=lambda kv: -kv[1][1])[:30]:
            sys.stderr.write("  %-40s %6.1f calls/step %8.3f ms/step %7.2f us/call\n" % (name, c / a.steps, 1e3 * sec / a.steps, 1e6 * sec / max(c, 1)))
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    pdist.check_peers()          # a peer-mapped statistics exchange that timed out would have produced invalid sums: fail loudly

    # ---- roofline leg: the SAME K steps once more with a HIP event pair around every contraction launch (recorded on
    # the launch stream).  Kept out of the timed region: ~850 event records per step cost ~13 % of the step time.
    kern = {}
    elapsed_ev = None
    if not a.no_kernel_events:
        for c in cores:
            c.profile(True)
        fence()
        t1 = time.perf_counter()
        for it in range(a.warmup + a.steps, a.warmup + 2 * a.steps):
            one_step(it)
        fence()
        elapsed_ev = time.perf_counter() - t1
        for kind, name in ((0, "conv_dma/conv_igemm (fwd+dgrad)"), (1, "conv_wgrad_dma/conv_wgrad")):
            ms = n = fl = by = 0.0
            for c in cores:
                m_, n_, f_ = c.profile_read(kind)
                ms, n, fl, by = ms + m_, n + n_, fl + f_, by + c.profile_bytes(kind)
            if n:
                kern[name] = {"launches": int(n), "avg_us": round(1e3 * ms / n, 3), "total_ms": round(ms, 3),
                              "algorithmic_gflop_per_launch": round(fl / n / 1e9, 4),
                              "algorithmic_mbytes_per_launch": round(by / n / 1e6, 3),
                              "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2)}
        for c in cores:
            c.profile(False)

    if rank == 0:
        loss_vals = {k: float(v) for k, v in last.items()} if isinstance(last, dict) else {"task_loss": float(last)}
        gb = per_gpu * world
        out = {"metric": "training images/sec (513x513, 21-cls)", "value": round(gb * a.steps / elapsed, 3),
               "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(1e3 * elapsed / a.steps, 3), "host_enqueue_ms_per_step": round(1e3 * host_enqueue / a.steps, 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "%s sseg, %s/ResNet-101, %dx%dx%d (%d labeled + %d unlabeled) per GPU, "
                                      "21 classes" % ({"mt": "MT (mean-teacher)", "adv": "AdvSSL (+ FC discriminator)",
                                                       "gct": "GCT (dual task model + flaw detector)",
                                                       "cct": "CCT (shared encoder + K=%d perturbed aux decoders)" % a.decoders,
                                                       "suponly": "SupOnly"}[a.algo],
                                                      "PSPNet" if a.algo == "cct" else "DeepLab-v2", per_gpu,
                                                      a.size, a.size, a.lbs, per_gpu - a.lbs),
                          "algorithm": "ssl_" + {"mt": "mt", "adv": "adv", "gct": "gct", "cct": "cct", "suponly": "null"}[a.algo], "global_batch": gb,
                          "im_size": a.size, "parallelism": "dp%d" % world, "sync_bn": world > 1},
               "final_losses": loss_vals,
               "init": "reference initialisers" + ("" if a.no_conditioned else ", bottleneck-output BN gammas x 0.1 (the parity fixtures' conditioning)"),
               # multi-rank: ranks on the C-driven RCCL communicators (0 = torch.distributed carries the exchanges) and the
               # gradient buckets all-reduced from inside the last backward pass (overlapped with it)
               "rccl_ranks": pdist.rccl_ranks(), "grad_buckets": int(cores[0].grad_buckets()),
               # networks whose Sync-BN statistics go through the peer-mapped one-shot exchange (csrc/peer.hip)
               "peer_contexts": pdist.peer_contexts()}
        sg = getattr(algo, "_sgraph", None)
        # the iteration as ONE hipGraph launch (pixelssl_amd/graph.py; PXL_GRAPH=0 turns it off): replays inside this process so far
        out["graph"] = ({"captured": sg.graph is not None, "replays": int(sg.replays), "failed": sg.failed} if sg is not None
                        else {"captured": False, "replays": 0, "failed": None})
        # which parameter update ran in the timed steps: the fused SGD + EMA + bf16-copy kernel from inside the backward pass
        # (nn/optimizer.py: PipelinedUpdate; multi-rank: behind each gradient bucket's all-reduce) or the separate kernels after it
        pipe = getattr(algo, "_pipe", None)
        opt0 = getattr(algo, "s_optimizer", None) or getattr(algo, "optimizer", None)
        out["update_path"] = {"pipelined": bool(pipe is not None and getattr(opt0, "last_step_pipelined", False)),
                              "kernel": ("sgd_ema_pack (fused)" if getattr(pipe, "fused", False) else "sgd / ema / pack per bucket") if pipe is not None
                              else "sgd, ema, pack after the backward pass",
                              "buckets_last_step": int(cores[0].update_buckets()) if pipe is not None else 0,
                              "behind_gradient_allreduce": bool(pipe is not None and world > 1)}
        if a.algo == "mt" and getattr(cores[0], "_cur", None) is not None:
            # paired student || teacher pass (the default with Sync-BN): convolutions issued as ONE launch for both networks and
            # Sync-BN statistics exchanges that carried both networks' sums (csrc/net.cpp: pxl_net_forward_pair)
            from pixelssl_amd._lib import lib as _pl
            out["paired_convs"] = int(_pl().pxl_net_pairs(cores[0]._cur.net))
            out["paired_stat_exchanges"] = int(_pl().pxl_net_pair_syncs(cores[0]._cur.net))
        # what the line claims about the job, checked before it is printed: N ranks, every one of them on the RCCL
        # communicators (unless the run was explicitly put on another backend / onto one shared GPU for a test)
        shared = os.environ.get("PXL_FORCE_DEVICE") is not None or os.environ.get("PXL_DIST_BACKEND") not in (None, "nccl")
        assert out["n_gpus"] == a.gpus, "n_gpus %d != --gpus %d" % (out["n_gpus"], a.gpus)
        if world > 1 and not shared:
            assert out["rccl_ranks"] == world, "only %d of %d ranks are on the RCCL communicators" % (out["rccl_ranks"], world)
        out["checked"] = {"n_gpus_equals_gpus_flag": True, "rccl_ranks_equals_world": bool(world == 1 or shared or out["rccl_ranks"] == world),
                          "peer_exchange_active": bool(out["peer_contexts"] > 0) if world > 1 else None}
        if elapsed_ev is not None:
            out["ms_per_step_with_kernel_events"] = round(1e3 * elapsed_ev / a.steps, 3)
        if kern:
            dom = max(kern, key=lambda k: kern[k]["total_ms"])
            peak = MFMA_PEAK_TFLOPS[a.dtype]
            # HBM bytes per launch of the dominant kernel: NOT measured by this run (PMC counters need their own rocprofv3
            # passes) -- read from profiles/traffic.json, which tools/pmc_traffic.py wrote from two `rocprofv3 --pmc` runs
            # of this same command (FETCH_SIZE doubled per the gfx950 note of the guide); traffic_source says which
            traffic, traffic_source = None, None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                key = "conv_dma_kernel" if dom.startswith("conv_dma") else "conv_wgrad_dma_kernel"
                if a.algo == "mt" and a.dtype == "bf16":
                    traffic = tj["kernels"][key]["traffic_bytes_per_launch"]
                    traffic_source = "profiles/traffic.json (%s)" % tj.get("measured", "separate rocprofv3 --pmc passes")
            except Exception:
                traffic = None
            # Which roof binds the dominant kernel: its arithmetic intensity against the ridge of the chip (dense MFMA peak / HBM
            # peak).  Two intensities: algorithmic (operands once) and measured (flops over the PMC traffic, which also holds
            # what the fused epilogues read and write); below the ridge the kernel is HBM-bound and `achieved` is its bandwidth.
            HBM_PEAK_GBPS = 8000.0                                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
            ridge = peak * 1e12 / (HBM_PEAK_GBPS * 1e9)
            us = kern[dom]["avg_us"]
            gflop = kern[dom]["algorithmic_gflop_per_launch"]
            abytes = kern[dom]["algorithmic_mbytes_per_launch"] * 1e6
            inten_alg = gflop * 1e9 / abytes if abytes else None
            inten_meas = gflop * 1e9 / traffic if traffic else None
            hbm_bound = (inten_meas if inten_meas is not None else inten_alg or ridge) < ridge
            mfma_view = {"achieved": kern[dom]["achieved_tflops"], "peak": peak, "unit": "TFLOP/s",
                         "frac": round(kern[dom]["achieved_tflops"] / peak, 4)}
            hbm_view = {"achieved": round(abytes / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(abytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                        "traffic_gbps": round(traffic / (us * 1e-6) / 1e9, 1) if traffic else None,
                        "traffic_frac": round(traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None}
            view = hbm_view if hbm_bound else mfma_view
            out["roofline"] = {"kernel": dom, "bound": "hbm" if hbm_bound else "mfma", "achieved": view["achieved"],
                               "peak": view["peak"], "unit": view["unit"], "frac": view["frac"],
                               # both fractions side by side, whichever roof `bound` names: the north-star target is quoted on the MFMA one
                               "frac_mfma": mfma_view["frac"], "frac_hbm": hbm_view["frac"],
                               "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": us,
                               "algorithmic_gflop_per_launch": gflop, "algorithmic_bytes_per_launch": int(abytes),
                               "intensity_flop_per_byte": {"algorithmic": round(inten_alg, 1) if inten_alg else None,
                                                           "measured": round(inten_meas, 1) if inten_meas else None,
                                                           "ridge": round(ridge, 1)},
                               "mfma": mfma_view, "hbm": hbm_view,
                               "measured": "second pass of the same K steps with per-launch HIP events (outside the timed region)",
                               "note": "kernels of two HIP streams overlap (teacher || student forward, wgrad || dgrad): "
                                       "per-launch event durations include the co-running kernel; step_mfma_frac is the "
                                       "whole-step figure"}
            # whole step against the HBM roof: bytes of ONE step = PMC traffic per launch x launches per step over every kernel
            # family (tools/bytes_per_step.py on this workload's trace; static like `traffic`: counters need their own passes)
            try:
                bj = json.load(open(os.path.join(ROOT, "profiles", "bytes_per_step.json")))
                if a.algo == "mt" and a.dtype == "bf16" and a.size == 513 and per_gpu == 8:
                    out["bytes_per_step"] = int(bj["bytes_per_step"])
                    out["hbm_frac"] = round(bj["bytes_per_step"] / (elapsed / a.steps) / (HBM_PEAK_GBPS * 1e9), 4)
                    out["bytes_per_step_source"] = "profiles/bytes_per_step.json (tools/bytes_per_step.py: " + " x ".join(bj.get("source", [])) + ")"
                    out["hbm_frac_note"] = "bytes of a COMMITTED trace (static, see bytes_per_step_source) over THIS run's step time -- not counters of this run"
                # one traced step of this workload (tools/prof_summary.py --one-step --json on a rocprofv3 kernel trace of this
                # command; static like the two above): launches, sum of kernel durations, the non-contraction share
                tj2 = json.load(open(os.path.join(ROOT, "profiles", "step_trace.json")))
                if a.algo == "mt" and a.dtype == "bf16" and a.size == 513 and per_gpu == 8:
                    out["step_trace"] = {k: tj2[k] for k in ("launches_per_step", "kernel_sum_ms", "contraction_ms", "noncontraction_ms",
                                                             "conv_dma_ms", "conv_dma_launches", "wall_window_ms", "source") if k in tj2}
            except Exception:
                pass
            out["kernels"] = kern
            # whole-step view: algorithmic conv FLOPs per image over wall time
            # CCT: 3 x F_P (150.74 GFLOP) per image through the PSPNet, decoders ~0.05 GFLOP each
            flop_img = {"mt": 449.9e9, "adv": 435.0e9, "gct": 1291.2e9, "suponly": 337.1e9, "cct": 452.6e9}[a.algo]
            out["step_mfma_frac"] = round(out["value"] * flop_img / world / (peak * 1e12), 4) if a.size == 513 else None
        if world == 1 and not a.no_miou and a.algo in ("mt", "suponly") and a.size == 513:
            out["miou_vs_ref"] = miou_vs_oracle(cores[0], a)
    scale_legs, legs_failed = None, None
    if world > 1 and a.algo == "mt" and not a.no_scaling_legs:
        # The legs below run AFTER the measured MT region and must never cost the MT line: a leg that wedges a rank (a collective
        # whose peer died is a GPU-side wait no exception gets out of) would otherwise keep rank 0 from ever printing.  Watchdog on
        # EVERY rank: past the deadline rank 0 prints the line it has (MT fields, the legs marked as timed out) and all ranks leave
        # through os._exit(0), so the launcher returns.
        import threading
        deadline_s = float(os.environ.get("PXL_SCALING_LEGS_TIMEOUT_S", "300"))
        legs_done = threading.Event()

        def _watchdog():
            if legs_done.wait(deadline_s):
                return
            if rank == 0:
                line = dict(out)
                line["scaling_legs"] = "timed out after %.0f s; the MT fields above were measured before the legs started" % deadline_s
                line["ok"] = False
                print(json.dumps(line), flush=True)
            os._exit(0)
        threading.Thread(target=_watchdog, daemon=True).start()
        ms = 1e3 * elapsed / a.steps
        try:
            scale_legs = scaling_legs(a, world, batches, fence, dev, ms)
        except Exception as e:          # a leg that FAILS on this rank must not cost the MT line either
            legs_failed = "%s: %s" % (type(e).__name__, e)
            scale_legs = {"scaling_legs": "failed on rank %d (%s); the MT fields above were measured before the legs started" % (rank, legs_failed)}
        finally:
            legs_done.set()
        if rank == 0:
            out.update(scale_legs)
    do_fp32 = a.dtype == "bf16" and not a.no_fp32_leg and world == 1      # (scaling runs stay the headline workload only)
    do_seq = world == 1 and not a.no_kernel_events and not a.no_seq_leg
    if do_fp32 or do_seq:
        del algo, one_step
        cores = None
        torch.cuda.empty_cache()
    # Everything below is reported NEXT TO the measured line; a leg that raises is put on record (`leg_errors`) instead of costing the line
    leg_errors = {}

    def _leg(name, fn):
        try:
            return fn()
        except Exception as e:
            leg_errors[name] = "%s: %s" % (type(e).__name__, e)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            return None
    seq = _leg("sequential_leg", lambda: sequential_leg(a, world, batches, fence)) if do_seq else None
    leg = _leg("fp32_parity_leg", lambda: fp32_parity_leg(a, world, batches, fence)) if do_fp32 else None
    if rank == 0:
        if seq is not None and "roofline" in out:
            dom = out["roofline"]["kernel"]
            out["roofline"]["one_kernel_at_a_time"] = dict(seq["kernels"].get(dom, {}), ms_per_step_with_kernel_events=seq["ms_per_step_with_kernel_events"],
                                                           note=seq["note"])
            out["kernels_one_at_a_time"] = seq["kernels"]
        if do_fp32 and leg is not None:
            out["fp32_parity_mode"] = leg
        if world == 1 and not a.no_fixture_parity and a.algo == "mt" and a.size == 513:
            algo = None
            torch.cuda.empty_cache()
            out["parity_vs_fixture"] = _leg("fixture_parity", lambda: fixture_parity(a, fence))
        if world == 1 and not a.no_cpu_baseline:
            algo = None
            torch.cuda.empty_cache()
            out["cpu_baseline"] = _leg("cpu_baseline", lambda: cpu_baseline(a))
        if leg_errors:
            out["leg_errors"] = leg_errors
        # one flag a reader (or a driver) can test: every leg that was asked for produced its figures
        out["ok"] = not leg_errors and legs_failed is None and not (isinstance(scale_legs, dict) and "scaling_legs" in scale_legs)
        print(json.dumps(out), flush=True)
    if legs_failed is not None:
        # the other ranks may still sit in a collective of the leg this rank left: no closing barrier with them (their own watchdogs
        # release them), the line -- if this is rank 0 -- is out
        sys.stderr.write("bench.py: rank %d: a scaling leg failed (%s)\n" % (rank, legs_failed))
        sys.stderr.flush()
        os._exit(0)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
