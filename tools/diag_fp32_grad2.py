#!/usr/bin/env python
"""Diagnostic 2 (GPU): eval-BN backward of the fp32 engine, gradient seeded at the latent only (bypasses head + ASPP).
Per-channel error structure of the first BN gradients met in the backward + ReLU-mask agreement of the latent."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch_oracle as TO  # noqa: E402
from pixelssl_amd.engine import DeepLabV2Core  # noqa: E402
from diag_fp32_grad import setup, rel, SHALLOW  # noqa: E402


def main():
    state, x, gt, w = setup(False)
    hw = (x.shape[2] + 15) // 16
    wl = torch.randn(x.shape[0], 2048, hw, hw, generator=torch.Generator().manual_seed(3)) * 1e-3

    def oracle(dtype):
        st = TO.clone_state(state)
        for k in st:
            if st[k].is_floating_point():
                st[k] = st[k].to(dtype)
        leaves = TO._param_leaves(st)
        run = TO._with_leaves(st, leaves)
        logits, prob, lat, _ = TO.deeplabv2_forward(run, x.to(dtype), train=False, layers=SHALLOW)
        (lat * wl.to(dtype)).sum().backward()
        return {k: v.grad for k, v in leaves.items()}, lat.detach()

    t, tlat = oracle(torch.float64)
    o, olat = oracle(torch.float32)
    core = DeepLabV2Core(backbone=SHALLOW, device="cuda", engine_dtype=torch.float32)
    core.load_state_dict(state)
    core.train(False)
    logits, prob, latent = core.forward_with_latent(x.cuda())
    (latent * wl.cuda()).sum().backward()
    torch.cuda.synchronize()
    e = {k: p.grad.cpu() for k, p in core.named_parameters()}
    elat = latent.detach().cpu()
    print("latent rel: engine %.2e oracle32 %.2e" % (rel(elat, tlat), rel(olat, tlat)))
    print("latent mask mismatches: engine-vs-fp64 %d, oracle32-vs-fp64 %d of %d"
          % (((elat > 0) != (tlat > 0)).sum().item(), ((olat > 0) != (tlat > 0)).sum().item(), elat.numel()))
    for k in ("backbone.layer4.2.bn3.bias", "backbone.layer4.2.bn3.weight", "backbone.layer4.2.conv3.weight",
              "backbone.layer4.2.bn2.bias", "backbone.layer4.2.conv2.weight", "backbone.layer4.2.bn1.bias",
              "backbone.layer4.1.bn3.bias", "backbone.layer4.0.downsample.1.bias", "backbone.conv1.weight"):
        d = (e[k].double() - t[k]).abs().reshape(e[k].shape[0], -1).amax(1)
        sc = t[k].abs().reshape(e[k].shape[0], -1).amax(1)
        r = d / (sc + 1e-30)
        srt = torch.sort(r, descending=True).values
        print("%-40s rel %.2e (oracle32 %.2e) | per-out-channel rel err: max %.2e, 10th %.2e, median %.2e, frac>1e-5 %.3f"
              % (k, rel(e[k], t[k]), rel(o[k], t[k]), srt[0], srt[min(9, len(srt) - 1)], srt[len(srt) // 2],
                 (r > 1e-5).float().mean()))


if __name__ == "__main__":
    main()
