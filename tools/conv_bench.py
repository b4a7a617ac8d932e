#!/usr/bin/env python
"""Per-shape micro-benchmark of the contraction kernels at the BASELINE batch (8 x 513 x 513 feature
map sizes, SURVEY.md 8a').  Prints TFLOP/s (algorithmic) for forward, data-gradient and
weight-gradient, per dtype and tile configuration.  Tuning aid, not part of the product path."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelssl_amd import ops  # noqa: E402

SHAPES = [
    # name, Cin, Cout, k, stride, dil, H (input), count in ResNet-101/ASPP
    ("stem7x7", 3, 64, 7, 2, 1, 513, 1),
    ("l1.1x1a", 64, 64, 1, 1, 1, 129, 1),
    ("l1.3x3", 64, 64, 3, 1, 1, 129, 3),
    ("l1.1x1b", 64, 256, 1, 1, 1, 129, 4),
    ("l1.1x1c", 256, 64, 1, 1, 1, 129, 2),
    ("l2.1x1a", 256, 128, 1, 1, 1, 129, 1),
    ("l2.3x3s2", 128, 128, 3, 2, 1, 129, 1),
    ("l2.1x1b", 128, 512, 1, 1, 1, 65, 4),
    ("l2.ds", 256, 512, 1, 2, 1, 129, 1),
    ("l2.1x1c", 512, 128, 1, 1, 1, 65, 3),
    ("l2.3x3", 128, 128, 3, 1, 1, 65, 3),
    ("l3.1x1a", 512, 256, 1, 1, 1, 65, 1),
    ("l3.3x3s2", 256, 256, 3, 2, 1, 65, 1),
    ("l3.1x1b", 256, 1024, 1, 1, 1, 33, 23),
    ("l3.ds", 512, 1024, 1, 2, 1, 65, 1),
    ("l3.1x1c", 1024, 256, 1, 1, 1, 33, 22),
    ("l3.3x3", 256, 256, 3, 1, 1, 33, 22),
    ("l4.1x1a", 1024, 512, 1, 1, 1, 33, 1),
    ("l4.3x3d2", 512, 512, 3, 1, 2, 33, 1),
    ("l4.1x1b", 512, 2048, 1, 1, 1, 33, 3),
    ("l4.ds", 1024, 2048, 1, 1, 1, 33, 1),
    ("l4.1x1c", 2048, 512, 1, 1, 1, 33, 2),
    ("l4.3x3d4", 512, 512, 3, 1, 4, 33, 1),
    ("aspp", 2048, 21, 3, 1, 6, 33, 1),
]


def pitch(c):
    return 8 if c <= 8 else (c + 31) // 32 * 32


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cfgs", default="-1")
    ap.add_argument("--only", default="")
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cfgs = [int(c) for c in a.cfgs.split(",")]
    B = a.batch
    tot = {m: 0.0 for m in a.modes.split(",")}
    totf = 0.0
    print("%-10s %5s %5s %2s %2s %2s %4s | %s" % ("shape", "Cin", "Cout", "k", "s", "d", "H", "mode:cfg TFLOP/s (us)"))
    for name, cin, cout, k, s, d, H, cnt in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):       # comma list of substrings
            continue
        p = d * (k - 1) // 2 if k > 1 else 0
        if name == "stem7x7":
            p = 3
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        cip, cop = pitch(cin), pitch(cout)
        x = torch.randn(B, H, H, cip, device="cuda").to(dtype)
        y = torch.randn(B, Ho, Ho, cop, device="cuda").to(dtype)
        wf = torch.randn(cout, k * k, cip, device="cuda").to(dtype) * 0.05
        wt = torch.randn(cin, k * k, cop, device="cuda").to(dtype) * 0.05
        dw = torch.zeros(cout, k * k, cin, device="cuda")
        stats = torch.zeros(2 * cout, device="cuda")
        taps = ops.fwd_taps(k, k, d, p)
        flops = 2.0 * B * Ho * Ho * cout * k * k * cin
        line = "%-10s %5d %5d %2d %2d %2d %4d |" % (name, cin, cout, k, s, d, H)
        for mode in a.modes.split(","):
            best = None
            for cfg in cfgs:
                if mode == "fwd":
                    desc = ops.conv_desc(dtype, B, H, H, cip, Ho, Ho, cop, cout, taps, out_stride=s, tile_cfg=cfg)
                    fn = lambda: ops.conv_igemm(desc, x, wf, y, stats=stats)
                elif mode == "dgrad":
                    if name == "stem7x7":
                        continue
                    desc = ops.conv_desc(dtype, B, Ho, Ho, cop, H, H, cip, cin, [(-i, -j) for i, j in taps],
                                         out_stride=1, div=s, tile_cfg=cfg)
                    fn = lambda: ops.conv_igemm(desc, y, wt, x)
                else:
                    if 2 < cfg < 8 or cfg > 13:
                        continue
                    desc = ops.conv_desc(dtype, B, H, H, cip, Ho, Ho, cop, cout, taps, out_stride=s, tile_cfg=cfg)
                    fn = lambda: ops.conv_wgrad(desc, x, y, dw, cin, cin)
                try:
                    t = timeit(fn, a.iters)
                except Exception as e:      # noqa
                    line += " %s:%d ERR" % (mode, cfg)
                    continue
                line += " %s:%d %6.1f (%6.1f)" % (mode, cfg, flops / t / 1e12, t * 1e6)
                best = t if best is None else min(best, t)
            if best is not None:
                tot[mode] += best * cnt
        totf += flops * cnt
        print(line, flush=True)
    for m, t in tot.items():
        if t:
            print("sum over net (%s): %.2f ms  -> %.1f TFLOP/s" % (m, t * 1e3, totf / t / 1e12))


if __name__ == "__main__":
    main()
