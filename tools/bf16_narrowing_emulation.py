#!/usr/bin/env python
"""Which bf16 STORAGE POINTS of the engine make its update differ from the fp32 reference's -- and what would narrowing buy?
CPU emulation (torch fp32 arithmetic, tensors rounded to bf16 exactly where the bf16 engine stores them); no GPU needed.

    python tools/bf16_narrowing_emulation.py [--size 513] [--batch 4] [--threads 8] > profiles/r05_bf16_narrowing_emulation.txt

VERDICT round 4, item 4: "one real attempt at narrowing bf16: fp32 residual stream and/or fp32 ASPP + head ... keep it if it
buys >= 2x in distance for <= 10 % time; a measured 'no' is acceptable".  Building a mixed-precision plan into the executor
costs days; whether it can pay is a question about ROUNDING, which this answers exactly: one SupOnly training step of
DeepLab-v2 / ResNet-101 on conditioned weights (the parity fixtures' initialisation and synthetic batch), gradients of the
fp32 graph against gradients of the same graph with bf16 rounding inserted at

    W   convolution weights (the packed bf16 copies; master weights and weight GRADIENTS stay fp32)
    Y   every convolution output y (forward value; backward: the gradient that reaches it = dy written by the BN backward)
    Z   relu(bn(y)) fed to the next convolution (forward value and the gradient dz written by that convolution's data gradient)
    R   the residual stream: every bottleneck's output relu(bn3 + shortcut) (value and gradient)
    H   the ASPP input / output (low-resolution logits)

Engine emulation = all five.  Variants switch points off: "R off" = fp32 residual stream (join outputs and shortcuts kept in
fp32, convolution OPERANDS still bf16: conv1 of the next block reads a rounded copy), "H off" = fp32 ASPP + head, and the
controls (gradients only / activations only / weights only).  Reported per parameter group: |g - g_fp32| / |g_fp32| and the cosine; for the
forward: relative logits error and arg-max agreement at full resolution."""
import argparse
import os
import sys
import time
from collections import OrderedDict

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch_oracle as TO      # noqa: E402


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


def rq(x, on, fwd=True, bwd=True):
    return _Round.apply(x, fwd, bwd) if on else x


def forward(sd, x, P, fwd=True, bwd=True):
    """TO.deeplabv2_forward with rounding points P (a set of letters), train-mode BN"""
    w = lambda k: rq(sd[k], "W" in P, fwd, False)        # weight gradients accumulate in fp32
    y_ = lambda t: rq(t, "Y" in P, fwd, bwd)
    z_ = lambda t: rq(t, "Z" in P, fwd, bwd)
    prefix = "backbone"
    h = F.conv2d(rq(x, "Z" in P, fwd, False), w(prefix + ".conv1.weight"), None, stride=2, padding=3)
    h = z_(F.relu(TO._bn(sd, prefix + ".bn1", y_(h), True)))
    h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
    for stage in TO.resnet101_os16_table(TO.RESNET101):
        for blk in stage:
            p = prefix + "." + blk["name"]
            # the residual stream h: stored bf16 by the engine (R); with R off it stays fp32 and the convolution that reads it
            # gets a rounded copy (its operand is bf16 either way when Z is on)
            h_store = rq(h, "R" in P, fwd, bwd)
            h_op = h_store if "R" in P else rq(h_store, "Z" in P, fwd, bwd)
            o = F.conv2d(h_op, w(p + ".conv1.weight"))
            o = z_(F.relu(TO._bn(sd, p + ".bn1", y_(o), True)))
            o = F.conv2d(o, w(p + ".conv2.weight"), None, stride=blk["stride"], padding=blk["dil"], dilation=blk["dil"])
            o = z_(F.relu(TO._bn(sd, p + ".bn2", y_(o), True)))
            o = F.conv2d(o, w(p + ".conv3.weight"))
            o = TO._bn(sd, p + ".bn3", y_(o), True)
            if blk["down"]:
                r = F.conv2d(h_op, w(p + ".downsample.0.weight"), None, stride=blk["stride"])
                r = TO._bn(sd, p + ".downsample.1", y_(r), True)
            else:
                r = h_store
            h = F.relu(o + r)
    feat = rq(h, "R" in P, fwd, bwd)
    feat_op = feat if ("R" in P or "H" not in P) else rq(feat, True, fwd, bwd)
    out = None
    for i, rate in enumerate(TO.ASPP_RATES):
        wk = "classifier.conv2d_list.%d.weight" % i
        wi = rq(sd[wk], "W" in P and "H" in P, fwd, False)
        yy = F.conv2d(feat_op, wi, sd["classifier.conv2d_list.%d.bias" % i], padding=rate, dilation=rate)
        out = yy if out is None else out + yy
    low = rq(out, "H" in P, fwd, bwd)
    return F.interpolate(low, size=x.shape[2:], mode="bilinear", align_corners=True)


def groups(names):
    g = OrderedDict()
    for k in names:
        if k.startswith("backbone.layer"):
            name = k.split(".")[1]
        elif k.startswith("backbone."):
            name = "stem"
        else:
            name = "aspp"
        g.setdefault(name, []).append(k)
    return g


def run(state, x, gt, P, fwd=True, bwd=True):
    sd = TO.clone_state(state)
    leaves = TO._param_leaves(sd)
    sdl = TO._with_leaves(sd, leaves)
    logits = forward(sdl, x, P, fwd, bwd)
    loss = TO.sseg_criterion(logits, gt).mean()
    loss.backward()
    return logits.detach(), OrderedDict((k, v.grad.detach().clone()) for k, v in leaves.items()), float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--seed", type=int, default=191)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    state = TO.condition_state(TO.init_deeplabv2_state(seed=a.seed), 0.1)
    x, gt = TO.synthetic_batch(a.batch, a.size, a.batch, seed=a.seed + 1, block=32)
    t0 = time.time()
    ref_logits, ref_g, ref_loss = run(state, x, gt, set())
    print("# DeepLab-v2 / ResNet-101, conditioned weights (seed %d), %d x %d x %d labeled crops, one SupOnly step; fp32 graph: loss %.6f (%.0f s)"
          % (a.seed, a.batch, a.size, a.size, ref_loss, time.time() - t0))
    gr = groups(ref_g.keys())
    variants = [
        ("engine emulation: W Y Z R H", "WYZRH", True, True),
        ("R off  (fp32 residual stream)", "WYZH", True, True),
        ("H off  (fp32 ASPP + head)", "WYZR", True, True),
        ("R + H off", "WYZ", True, True),
        ("R + H + Y off (only operands W, Z bf16)", "WZ", True, True),
        ("control: gradients only (forward fp32)", "YZRH", False, True),
        ("control: activations only (gradients fp32)", "WYZRH", True, False),
        ("control: weights only", "W", True, True),
    ]
    hdr = "%-46s %9s %9s | " % ("variant", "logit err", "arg-max") + " ".join("%13s" % n for n in gr) + " | %13s" % "all"
    print(hdr)
    print("%-46s %9s %9s | " % ("", "", "") + " ".join("%13s" % "dist  cos" for _ in gr) + " | %13s" % "dist  cos")
    for name, P, fwd, bwd in variants:
        t0 = time.time()
        lg, g, loss = run(state, x, gt, set(P), fwd, bwd)
        lerr = ((lg - ref_logits).norm() / ref_logits.norm()).item()
        agree = (lg.argmax(1) == ref_logits.argmax(1)).float().mean().item()
        cells = []
        num = den = dot = na = nb = 0.0
        for gname, keys in gr.items():
            d = sum(((g[k] - ref_g[k]).double() ** 2).sum().item() for k in keys)
            r = sum((ref_g[k].double() ** 2).sum().item() for k in keys)
            e = sum((g[k].double() ** 2).sum().item() for k in keys)
            c = sum((g[k].double() * ref_g[k].double()).sum().item() for k in keys)
            cells.append("%6.3f %6.3f" % ((d / r) ** 0.5, c / ((r * e) ** 0.5 + 1e-300)))
            num += d; den += r; dot += c; na += e
        cells.append("%6.3f %6.3f" % ((num / den) ** 0.5, dot / ((den * na) ** 0.5)))
        print("%-46s %9.2e %9.5f | " % (name, lerr, agree) + " ".join("%13s" % c for c in cells[:-1]) + " | %13s" % cells[-1]
              + "   (%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
