"""Checkpoint / resume on the device (format compatibility with the reference is covered on the CPU
by tests/test_conformance.py::test_checkpoints_are_compatible_both_ways): `_save_checkpoint` after three MT iterations,
`_load_checkpoint` into a fresh algorithm object, and the fourth iteration of the resumed run equals the fourth iteration
of the uninterrupted one -- loss, weights, SGD momentum, learning rate and schedule position."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = "cuda"


def _mt(args):
    import pixelssl_amd as P
    from pixelssl_amd.nn import optimizer as popt, lrer as plr
    return P.ssl_algorithm.ssl_mt.ssl_mt(args, {"model": P.sseg.model.deeplabv2()}, {"model": popt.sgd(args)},
                                        {"model": plr.polynomiallr(args)}, {"model": P.sseg.criterion.sseg_criterion()}, None)


@pytest.mark.gpu
def test_resume_continues_the_uninterrupted_run(tmp_path):
    import torch_oracle as TO
    from test_multistep import _fx, _args, _deeplab_state
    fx = _fx("mt_cond_129.pt")
    args = _args(fx, "fp32", cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99,
                 checkpoint_path=str(tmp_path))
    a = _mt(args)
    a.s_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"], fx["gamma3"]))
    a.t_model.module.model.load_state_dict(_deeplab_state(fx["weight_seed"] + 1, fx["gamma3"]))
    a.s_model.train(), a.t_model.train()
    batches = [TO.synthetic_batch(fx["lbs"] + fx["ubs"], fx["size"], fx["lbs"], seed=s, block=fx["block"]) for s in fx["data_seeds"][:4]]
    for i in range(3):
        a.train_step((batches[i][0].to(DEV),), (batches[i][1].to(DEV),), i, fx["rampup_iters"])
    a.save_checkpoint(0)
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint_0.ckpt"), weights_only=False)
    assert set(ck) == {"algorithm", "epoch", "s_model", "t_model", "s_optimizer", "s_lrer"} and ck["algorithm"] == "ssl_mt"
    assert len(ck["s_optimizer"]["state"]) == 320 and ck["s_optimizer"]["state"][0]["momentum_buffer"].shape == (64, 3, 7, 7)
    assert list(ck["s_model"])[0].startswith("module.model.")
    want, _, _ = a.train_step((batches[3][0].to(DEV),), (batches[3][1].to(DEV),), 3, fx["rampup_iters"])

    args2 = _args(fx, "fp32", cons_for_labeled=False, cons_scale=1.0, cons_rampup_epochs=3, ema_decay=0.99,
                  resume=os.path.join(str(tmp_path), "checkpoint_0.ckpt"))
    b = _mt(args2)
    assert b.load_checkpoint() == 0
    # bit-exact restore: re-saving the loaded state gives the same optimizer tensors
    re = b.s_optimizer.state_dict()["state"]
    assert all(torch.equal(re[i]["momentum_buffer"].cpu(), ck["s_optimizer"]["state"][i]["momentum_buffer"].cpu()) for i in re)
    assert all(torch.equal(v.cpu(), ck["s_model"][k].cpu()) for k, v in b.s_model.state_dict().items())
    b.s_model.train(), b.t_model.train()
    assert b.s_lrer.cur_iter == 4 and [g["lr"] for g in b.s_optimizer.param_groups] == [g["lr"] for g in ck["s_optimizer"]["param_groups"]]
    got, _, _ = b.train_step((batches[3][0].to(DEV),), (batches[3][1].to(DEV),), 3, fx["rampup_iters"])
    torch.cuda.synchronize()
    for k in want:
        assert abs(got[k].item() - want[k].item()) <= 2e-6 * abs(want[k].item()) + 1e-9, (k, got[k].item(), want[k].item())
    pa, pb = a.s_model.module.model.flat, b.s_model.module.model.flat
    # the fourth step is computed twice: fp32 atomics order and ReLU decisions at rounding level differ run to run, which
    # moves the gradient by 3e-4 .. 3e-3 (L2; the reference arithmetic's own fp32-vs-fp64 gap on trunk gradients is
    # 3e-3, tests/test_parity_513.py) -- the restored state itself is bit-exact (checked above)
    assert (pa.params - pb.params).norm().item() <= 1e-6 * pa.params.norm().item()
    assert (pa.momentum - pb.momentum).norm().item() <= 1e-2 * pa.momentum.norm().item()
    # a resumed run that dropped the optimizer state (the round-1 bug: models only) would restart momentum at zero and
    # the schedule at cur_iter 0 -- both visible here
    assert pb.momentum.abs().max().item() > 0 and b.s_lrer.cur_iter == 5
