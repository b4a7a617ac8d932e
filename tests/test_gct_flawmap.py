"""GCT flaw-map pipeline: CPU oracle vs the fixtures generated from the real reference
modules (not gpu), device modules vs oracle + fixtures (gpu) at 65 x 65 AND at the BASELINE crop size 513 x 513 (blur
kernels 65 / 129 / 33: tests/golden/gct_flawmap_513.pt, oracle/make_golden_gct.py 513), plus size-independent properties."""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def _case(size=65):
    import gct_oracle as GO
    fx = torch.load(os.path.join(GOLD, "gct_flawmap_%d.pt" % size))
    return GO, fx, GO.synthetic_case(fx["seed"], C=fx["C"], size=fx["size"])


def test_oracle_reproduces_reference_fixtures():
    GO, fx, (l_pred, r_pred, gt, l_fm, r_fm) = _case()
    size = fx["size"]
    onehot = GO.onehot_ignore(gt, fx["C"])
    assert onehot[-1].abs().sum() == 0                       # unlabeled sample: all-zero one-hot
    assert torch.allclose(GO.fdgt(l_pred, onehot, size, fx["mu"], fx["nu"]), fx["fdgt"], atol=2e-6)
    lh, lc = GO.flawmap_handle(l_fm, size)
    rh, rc = GO.flawmap_handle(r_fm, size)
    assert torch.allclose(lh, fx["l_handled"], atol=2e-6) and torch.allclose(rh, fx["r_handled"], atol=2e-6)
    assert torch.equal(lc, fx["l_clamped"]) and torch.equal(rc, fx["r_clamped"])
    m = GO.dcgt(l_pred, r_pred, fx["l_handled"], fx["r_handled"], fx["dc_threshold"])
    assert torch.equal(m[0][:, :, :4, :8], fx["l_dc_gt_head"]) and torch.equal(m[2].to(torch.uint8), fx["both_bad"])
    assert torch.equal(m[3], fx["l_fm_after"]) and torch.equal(m[4], fx["r_fm_after"])
    assert abs(m[0].double().sum().item() - fx["l_dc_gt_sum"]) < 1e-6 * abs(fx["l_dc_gt_sum"])
    assert torch.allclose(GO.fd_criterion(l_fm, fx["fdgt"]), fx["fd_loss"], rtol=1e-6)
    # the handled map of the sample below the clip threshold is the constant -min/(max-min)
    assert (fx["r_handled"][0].max() - fx["r_handled"][0].min()).abs() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("size", [65, 513])
def test_device_modules_match_oracle_and_fixtures(size):
    """513: the reference's own FDGTGenerator / FlawmapHandler / DCGTGenerator outputs at the BASELINE crop size (dense
    65- / 129- / 33-tap blurs, ssl_gct.py:637-639, 701-707) against the separable device kernels."""
    GO, fx, (l_pred, r_pred, gt, l_fm, r_fm) = _case(size)
    from pixelssl_amd.ssl_algorithm import ssl_gct as G
    size = fx["size"]
    args = argparse.Namespace(im_size=size, mu=fx["mu"], nu=fx["nu"], dc_threshold=fx["dc_threshold"])
    dev = "cuda"
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    # one-hot (bit exact) and FDGT through both input forms
    oh = G.onehot_ignore(gt.to(dev), fx["C"])
    assert torch.equal(oh.cpu(), GO.onehot_ignore(gt, fx["C"]))
    gen = G.FDGTGenerator(args).to(dev)
    # the last sample is unlabeled: its |onehot - softmax| map is the constant mu, whose min-max normalisation is
    # 0/1e-9 up to rounding noise of the blur (the reference itself returns ulp-noise / 1e-9 there) -> not compared
    st = fx.get("stride", 1)                  # 513: the fixture holds every map on a stride-4 grid + its full-tensor sum
    sub = lambda t: t[..., ::st, ::st]
    want = fx["fdgt"].detach()[:-1]
    for form in (gt.to(dev), oh):
        full = gen(l_pred.to(dev), form).cpu()
        out = sub(full)[:-1]
        # separable fp32 evaluation vs the dense fp32 convolution, divided by the (small) per-sample range
        # (65 and 129 taps of fp32 rounding at 513 x 513, then a division by the per-sample range: 3.8e-4 measured)
        tol = 1e-4 if st == 1 else 1e-3
        assert rel(out, want) < tol and (out - want).abs().max() < tol          # bar: 1e-3 rel (BASELINE.json)
        if st > 1:
            assert abs(full[:-1].double().sum().item() - fx["sums"]["fdgt"]) < tol * abs(fx["sums"]["fdgt"])
    # FlawmapHandler: clamps its argument in place, thresholded sample stays constant
    handler = G.FlawmapHandler(args).to(dev)
    l_in, r_in = l_fm.to(dev), r_fm.to(dev)
    lh, rh = handler(l_in), handler(r_in)
    assert torch.equal(sub(l_in.cpu()), fx["l_clamped"]) and torch.equal(sub(r_in.cpu()), fx["r_clamped"])
    htol = 1e-4 if st == 1 else 1e-3
    assert (sub(lh.cpu()) - fx["l_handled"]).abs().max() < htol
    # (r sample 0 sits below the clip threshold: its handled map is the CONSTANT -min / (max - min) of a map whose range is
    # rounding noise of the 33-tap blur -- compared as "constant, and the reference's constant", the others element-wise)
    assert (sub(rh.cpu())[1:] - fx["r_handled"][1:]).abs().max() < htol
    r0 = rh[0].cpu()
    assert (r0.max() - r0.min()).item() < 5 * htol and abs(r0.mean().item() - fx["r_handled"][0].mean().item()) < 5 * htol
    if st > 1:
        assert abs(lh.double().sum().item() - fx["sums"]["l_handled"]) < htol * abs(fx["sums"]["l_handled"])
        assert abs(rh.double().sum().item() - fx["sums"]["r_handled"]) < htol * abs(fx["sums"]["r_handled"])
    # DCGT: bit exact, in-place update of the maps.  65: on the reference's handled maps; 513 (the fixture keeps a grid of
    # them only): device kernel and CPU oracle on the SAME handled maps (the device's, just checked against the reference),
    # the count of pixels both networks get wrong against the reference's within the handled maps' 1e-4
    if st == 1:
        lh_in, rh_in = fx["l_handled"], fx["r_handled"]
    else:
        lh_in, rh_in = lh.cpu().clone(), rh.cpu().clone()
    lh2, rh2 = lh_in.to(dev), rh_in.to(dev)
    l_gt, r_gt, bad, bad2 = G.DCGTGenerator(args)(l_pred.to(dev), r_pred.to(dev), lh2, rh2)
    want = GO.dcgt(l_pred, r_pred, lh_in, rh_in, fx["dc_threshold"])
    assert torch.equal(l_gt.cpu(), want[0]) and torch.equal(r_gt.cpu(), want[1]) and torch.equal(bad.cpu(), want[2])
    if st == 1:
        assert torch.equal(lh2.cpu(), fx["l_fm_after"]) and torch.equal(rh2.cpu(), fx["r_fm_after"]) and bad2 is bad
    else:
        assert torch.equal(lh2.cpu(), want[3]) and torch.equal(rh2.cpu(), want[4]) and bad2 is bad
        assert abs(int(bad.sum().item()) - fx["sums"]["both_bad"]) <= max(4, fx["sums"]["both_bad"] // 1000)
        assert (sub(bad.cpu().to(torch.uint8)) != fx["both_bad"]).float().mean().item() < 1e-3
    # FD criterion forward + backward
    fdgt_full = fx["fdgt"].detach() if st == 1 else full.detach()
    a = l_fm.to(dev).requires_grad_(True)
    loss = G.FlawDetectorCriterion()(a, fdgt_full.to(dev))
    if st == 1:
        assert rel(loss.detach().cpu(), fx["fd_loss"]) < 1e-5
    loss.sum().backward()
    ar = l_fm.clone().requires_grad_(True)
    lo = GO.fd_criterion(ar, fdgt_full)
    lo.sum().backward()
    assert rel(loss.detach().cpu(), lo.detach()) < 1e-5 and rel(a.grad.cpu(), ar.grad) < 1e-5


@pytest.mark.gpu
def test_flawmap_pipeline_properties_at_513():
    """size-independent properties at the BASELINE size (next to the fixture comparison above)"""
    from pixelssl_amd.ssl_algorithm import ssl_gct as G
    args = argparse.Namespace(im_size=513, mu=0.5, nu=1, dc_threshold=0.6)
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    B, C = 4, 21
    pred = torch.softmax(torch.randn(B, C, 513, 513, generator=g), 1).to(dev)
    gt = torch.randint(0, C, (B, 1, 513, 513), generator=g).float().to(dev)
    gen = G.FDGTGenerator(args).to(dev)
    assert gen.blur.kernel_size == 65 and gen.reblur.kernel_size == 129
    out = gen(pred, gt)
    flat = out.view(B, -1)
    assert torch.all(flat.min(1).values == 0) and torch.all((flat.max(1).values - 1).abs() < 1e-6)   # per-sample [0, 1]
    # perfect predictions -> |onehot - pred| == 0 -> the normalised map is 0 everywhere (0 / 1e-9)
    perfect = G.onehot_ignore(gt, C)
    assert gen(perfect, gt).abs().max().item() == 0.0
    # blur: constant maps are fixed points (taps sum to 1, reflect padding), and the blur is linear
    blur = G.GaussianBlurLayer(1, 129).to(dev)
    const = torch.full((1, 1, 513, 513), 0.37, device=dev)
    assert (blur(const) - 0.37).abs().max() < 2e-6
    a, b = torch.randn(2, 1, 513, 513, generator=g).to(dev), torch.randn(2, 1, 513, 513, generator=g).to(dev)
    assert (blur(a + 2 * b) - (blur(a) + 2 * blur(b))).abs().max() < 1e-5
    # handler idempotence of the clamp + range of the output
    fm = torch.randn(B, 1, 513, 513, generator=g).to(dev)
    h = G.FlawmapHandler(args).to(dev)(fm)
    assert fm.min().item() >= 0.0 and h.min().item() >= -1e-6 and h.max().item() <= 1 + 1e-6
