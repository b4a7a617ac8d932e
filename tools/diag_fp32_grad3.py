#!/usr/bin/env python
"""Diagnostic 3 (GPU): which elements are wrong in dbeta of the last BN (sum over pixels of dout * (out > 0))?"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pixelssl_amd.engine import DeepLabV2Core  # noqa: E402
from diag_fp32_grad import setup, SHALLOW  # noqa: E402

state, x, gt, w = setup(False)
hw = (x.shape[2] + 15) // 16
wl = (torch.randn(x.shape[0], 2048, hw, hw, generator=torch.Generator().manual_seed(3)) * 1e-3).cuda()
for trial in range(2):
    core = DeepLabV2Core(backbone=SHALLOW, device="cuda", engine_dtype=torch.float32)
    core.load_state_dict(state)
    core.train(False)
    if trial == 1:
        os.environ["PXL_SIDE_STREAM"] = "0"
    logits, prob, latent = core.forward_with_latent(x.cuda())
    (latent * wl).sum().backward()
    torch.cuda.synchronize()
    lat = latent.detach()
    want = (wl.double() * (lat > 0)).sum((0, 2, 3))
    got = getattr(core.backbone.layer4, "2").bn3.bias.grad.double()
    diff = got - want
    bad = (diff.abs() > 1e-6 * want.abs().max()).nonzero().flatten()
    print("trial %d (side stream %s): %d bad channels of %d" % (trial, os.environ.get("PXL_SIDE_STREAM", "1"), bad.numel(), want.numel()))
    for c in bad[:20].tolist():
        col = wl[:, c].double().flatten()
        lc = lat[:, c].flatten()
        d = diff[c].item()
        # is the difference one (or minus one) of the wl elements?
        j = (col - d).abs().argmin().item()
        j2 = (col + d).abs().argmin().item()
        print("  ch %4d diff % .6e | nearest +wl[%d]=% .6e (lat %.3e)  nearest -wl[%d]=% .6e (lat %.3e)"
              % (c, d, j, col[j].item(), lc[j].item(), j2, -col[j2].item(), lc[j2].item()))
