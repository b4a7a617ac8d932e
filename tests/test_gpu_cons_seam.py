"""The consistency seam of an SSLCCT auxiliary decoder as one fused pass (csrc/head.hip: pxl_cons_head_fwd / pxl_cons_head_bwd;
ssl_cct.py:482-484: F.interpolate(bilinear, align_corners=False) -> soft-max -> nn.MSELoss against the main decoder's soft-max,
and autograd's backward through the three).

Kernel level: against torch's interpolate / softmax / mse_loss + autograd on the same operands, through the C-ABI.
Decoder level: engine.AuxDecoderCore through functional.decoder_consistency against the same decoder through its unfused path
(forward() + functional.MSELoss): loss, latent gradient and parameter gradients."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_kernels import DEV, _pitch, qround, rel_err, to_nhwc, from_nhwc  # noqa: E402

GEOS = [  # B, C, h, w, H, W
    (2, 21, 40, 40, 37, 37),        # a slight down-scale, like the decoders' 520 -> 513
    (1, 21, 17, 19, 33, 41),        # up-scale
    (3, 5, 24, 24, 24, 24),         # identity resize, few classes (pitch 8)
    (2, 21, 72, 72, 65, 65),
]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("geo", GEOS, ids=["down", "up", "same", "72to65"])
def test_cons_seam_against_autograd(geo, dtype):
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    B, C, h, w, H, W = geo
    g = torch.Generator().manual_seed(h * 100 + W)
    low = qround(torch.randn(B, C, h, w, generator=g) * 2.0, dtype).requires_grad_(True)
    target = torch.softmax(torch.randn(B, C, H, W, generator=g) * 1.5, dim=1)
    gout = torch.tensor(0.37)
    prob = torch.softmax(F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False), dim=1)
    ref = F.mse_loss(prob, target)
    (ref * gout).backward()
    cp = _pitch(C)
    code = dtype_code(dtype)
    lowd = to_nhwc(low.detach(), cp, dtype)
    td = target.to(DEV)
    ws_bytes = lib().pxl_cons_head_workspace(B, w, C, H)
    assert ws_bytes >= B * H * w * C * 4 + B * H * 4
    ws = torch.full((ws_bytes // 4,), float("nan"), device=DEV)
    loss = torch.full((1,), float("nan"), device=DEV)
    check(lib().pxl_cons_head_fwd(code, B, h, w, cp, C, H, W, 0, ptr(lowd), ptr(td), ptr(ws), ws_bytes, ptr(loss), 0, stream_ptr()))
    dlow = torch.full((B, h, w, cp), float("nan"), device=DEV, dtype=dtype)
    gd = gout.to(DEV).reshape(1)
    check(lib().pxl_cons_head_bwd(code, B, h, w, cp, C, H, 0, ptr(ws), ws_bytes, ptr(gd), ptr(dlow), stream_ptr()))
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item()), (loss.item(), ref.item())
    assert torch.isfinite(dlow.float()).all() and (dlow[..., C:] == 0).all()
    err = rel_err(from_nhwc(dlow, C), low.grad)
    assert err < (5e-3 if dtype == torch.bfloat16 else 2e-5), err            # (bf16: the rounding of the stored gradient)
    # ordered: the loss folded in row order -- the same value to fp32 rounding, and bit-identical from run to run
    vals = []
    for _ in range(2):
        l2 = torch.full((1,), float("nan"), device=DEV)
        check(lib().pxl_cons_head_fwd(code, B, h, w, cp, C, H, W, 0, ptr(lowd), ptr(td), ptr(ws), ws_bytes, ptr(l2), 1, stream_ptr()))
        vals.append(l2.item())
    assert vals[0] == vals[1] and abs(vals[0] - ref.item()) <= 2e-5 * abs(ref.item())


@pytest.mark.gpu
def test_cons_seam_argument_checks():
    from pixelssl_amd._lib import lib, ptr, stream_ptr
    t = torch.zeros(64, device=DEV)
    # an output row that does not fit the LDS staging is refused (never a silent fallback), as is a short workspace
    assert lib().pxl_cons_head_lds_bytes(8, 21, 1025) > 64 * 1024
    assert lib().pxl_cons_head_fwd(0, 1, 8, 8, 32, 21, 1025, 1025, 0, ptr(t), ptr(t), ptr(t), 1 << 40, ptr(t), 0, stream_ptr()) != 0
    assert lib().pxl_cons_head_fwd(0, 1, 8, 8, 32, 21, 9, 9, 0, ptr(t), ptr(t), ptr(t), 16, ptr(t), 0, stream_ptr()) != 0
    assert lib().pxl_cons_head_fwd(0, 1, 8, 8, 32, 40, 9, 9, 0, ptr(t), ptr(t), ptr(t), 1 << 20, ptr(t), 0, stream_ptr()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_decoder_consistency_against_the_unfused_decoder(dtype):
    from pixelssl_amd.engine import AuxDecoderCore
    from pixelssl_amd import functional as PF
    torch.manual_seed(3)
    B, Cin, h = 2, 64, 9
    size = (h * 8 - 7, h * 8 - 7)                      # 72 -> 65, the 520 -> 513 of the 513 x 513 workload in small
    dec = AuxDecoderCore(8, Cin, 21, device=DEV, engine_dtype=dtype)
    dec.autotune = False
    dec.train()
    x = torch.randn(B, Cin, h, h, device=DEV)
    target = torch.softmax(torch.randn(B, 21, *size, device=DEV), dim=1)
    scale = 30.0 / 7                                    # ramp * cons_scale / K of the training step

    def grads():
        return torch.cat([p.grad.reshape(-1).clone() for p in dec.parameters()])

    xa = x.clone().requires_grad_(True)
    dec.zero_grad(set_to_none=False)
    _, act, _ = dec(xa, out_size=size)
    la = PF.MSELoss()(act, target)
    (la * scale).backward()
    ga, dxa = grads(), xa.grad.clone()

    xb = x.clone().requires_grad_(True)
    dec.zero_grad(set_to_none=False)
    assert PF.decoder_consistency_supported(dec, xb, target, size)
    lb, head = PF.decoder_consistency(dec, xb, target, size)
    (lb * scale).backward()
    gb, dxb = grads(), xb.grad.clone()
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert abs(la.item() - lb.item()) <= 1e-5 * abs(la.item()), (la.item(), lb.item())
    assert rel_err(dxb, dxa) < tol and rel_err(gb, ga) < tol, (rel_err(dxb, dxa), rel_err(gb, ga))
    # the resized prediction on demand: what forward() returns
    pred = head.materialize(want_prob=False)[0]
    ref_pred, _, _ = dec(x, out_size=size)
    assert torch.equal(pred, ref_pred.detach())
    # a target of another shape is not this seam's business
    assert not PF.decoder_consistency_supported(dec, xb, target[:, :, :-1], size)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_running_statistics_detour_equals_the_in_place_update(dtype):
    """engine.detour_running / fold_running (SSLCCT's concurrent labeled / unlabeled chains): a training pass that parks its
    BatchNorm running-statistics update and folds it in later leaves the statistics of the same two passes run in order."""
    from pixelssl_amd.engine import DeepLabV2Core
    torch.manual_seed(11)
    core = DeepLabV2Core(backbone=(1, 1, 1, 1), num_classes=21, device=DEV, engine_dtype=dtype)
    core.autotune = False
    core.train()
    x1, x2 = torch.randn(2, 3, 65, 65, device=DEV), torch.randn(2, 3, 65, 65, device=DEV) * 2 + 0.5
    r0, n0 = core._store.running.clone(), core._nbt.clone()
    with torch.no_grad():
        core(x1); core(x2)
        r_seq, n_seq = core._store.running.clone(), core._nbt.clone()
        core._store.running.copy_(r0); core._nbt.copy_(n0)
        core.detour_running()
        core(x2)                             # parked
        assert torch.equal(core._store.running, r0) and torch.equal(core._nbt, n0)
        with pytest.raises(Exception):
            core.detour_running()
        parked, core._running_detour = core._running_detour, None
        core(x1)                             # the pass that runs "first" in the reference's order, beside the parked one
        core._running_detour = parked
        core.fold_running()
        assert torch.equal(core._nbt, n_seq)
        err = (core._store.running - r_seq).abs().max().item() / r_seq.abs().max().item()
        assert err < 1e-5, err               # (the batch statistics themselves are atomically summed: 1e-6 run to run)
        assert (r_seq - r0).abs().max().item() > 1e-3            # (the passes did move the statistics)
        core.fold_running()                  # nothing parked: a no-op
        assert torch.equal(core._nbt, n_seq)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_backward_into_side_buffers_adds_up_to_the_accumulated_gradient(dtype):
    """engine.side_backward_buffers: two passes whose backward runs into the shared gradient buffer (the ordinary accumulation)
    against the first pass's backward redirected into its own gradient buffer + scratch and added afterwards."""
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd import functional as PF
    torch.manual_seed(12)
    core = DeepLabV2Core(backbone=(1, 1, 1, 1), num_classes=21, device=DEV, engine_dtype=dtype)
    core.autotune = False
    core.train()
    x1, x2 = torch.randn(2, 3, 65, 65, device=DEV), torch.randn(2, 3, 65, 65, device=DEV)
    gt = torch.randint(0, 21, (2, 65, 65), device=DEV).float()

    def loss_of(x):
        logits, _, _ = core(x)
        return PF.cross_entropy_per_sample(logits, gt, 255).mean()

    r0, n0 = core._store.running.clone(), core._nbt.clone()
    core.zero_grad(set_to_none=False)
    loss_of(x1).backward()
    loss_of(x2).backward()
    ref = core._store.grads.clone()
    core._store.running.copy_(r0); core._nbt.copy_(n0)
    core.zero_grad(set_to_none=False)
    l1 = loss_of(x1)
    l2 = loss_of(x2)                          # both forwards alive, as in the concurrent step
    alt = core.side_backward_buffers()
    alt[0].zero_()
    core._alt_backward = alt
    l1.backward()
    core._alt_backward = None
    only2_before = core._store.grads.abs().sum().item()
    assert only2_before == 0.0                # nothing of pass 1 went into the shared buffer
    l2.backward()
    core._store.grads.add_(alt[0])
    err = ((core._store.grads - ref).norm() / ref.norm()).item()
    # (this shallow random-init network with two samples per BatchNorm batch is not reproducible beyond ~4e-3 from run to run --
    # statistics summed by atomics, ReLU / max-pool decisions within rounding of a tie: measured on the ORDINARY path, two runs of
    # the same two passes differ by 3.7e-3 -- so the bar says "the same gradient", not "the same bits")
    assert err < 2e-2, err
