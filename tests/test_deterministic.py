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
