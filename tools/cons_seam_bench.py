"""The decoder consistency seam alone (csrc/head.hip: pxl_cons_head_fwd / pxl_cons_head_bwd) at the CCT workload's shape -- 4 x 21 classes,
264 x 264 -> 513 x 513 -- against the five separate launches it replaces (up-sampling + soft-max, MSE forward / backward, the two adjoint
passes), each timed with events over back-to-back launches.

    python tools/cons_seam_bench.py [--dtype bf16]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--dtype", default="bf16")
    a = p.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    code = dtype_code(dt)
    B, C, cp, h, w, H, W = 4, 21, 32, 264, 264, 513, 513
    dev = "cuda"
    low = torch.randn(B, h, w, cp, device=dev).to(dt)
    target = torch.softmax(torch.randn(B, C, H, W, device=dev), dim=1)
    ws_bytes = lib().pxl_cons_head_workspace(B, w, C, H)
    ws = torch.empty(ws_bytes // 4, device=dev)
    loss = torch.zeros(1, device=dev)
    g = torch.ones(1, device=dev)
    dlow = torch.empty_like(low)
    logits = torch.empty(B, C, H, W, device=dev)
    prob = torch.empty_like(logits)
    dprob = torch.empty_like(logits)

    def fused_fwd():
        check(lib().pxl_cons_head_fwd(code, B, h, w, cp, C, H, W, 0, ptr(low), ptr(target), ptr(ws), ws_bytes, ptr(loss), 0, stream_ptr()))

    def fused_bwd():
        check(lib().pxl_cons_head_bwd(code, B, h, w, cp, C, H, 0, ptr(ws), ws_bytes, ptr(g), ptr(dlow), stream_ptr()))

    def up_fwd():
        check(lib().pxl_upsample_softmax_fwd(code, B, h, w, cp, C, H, W, 0, ptr(low), ptr(logits), ptr(prob), stream_ptr()))

    def mse_f():
        check(lib().pxl_mse_fwd(prob.numel(), ptr(prob), ptr(target), ptr(loss), stream_ptr()))

    def mse_b():
        check(lib().pxl_mse_bwd(prob.numel(), ptr(prob), ptr(target), ptr(g), ptr(dprob), stream_ptr()))

    def up_bwd():
        check(lib().pxl_upsample_softmax_bwd(code, B, h, w, cp, C, H, W, 0, None, ptr(dprob), ptr(prob), ptr(dlow), ptr(ws), ws_bytes, stream_ptr()))

    alg = (low.numel() * low.element_size() + target.numel() * 4 + B * H * w * C * 4) / 1e6
    t = dict(fused_fwd=timeit(fused_fwd), fused_bwd=timeit(fused_bwd), up_fwd=timeit(up_fwd), mse_fwd=timeit(mse_f), mse_bwd=timeit(mse_b),
             up_bwd=timeit(up_bwd))
    print("%s, %d x %d x %d x %d -> %d x %d" % (a.dtype, B, C, h, w, H, W))
    for k, v in t.items():
        print("  %-10s %8.1f us" % (k, v))
    print("  fused seam %.1f us (forward pass moves %.0f MB: %.2f TB/s) against %.1f us for the separate launches"
          % (t["fused_fwd"] + t["fused_bwd"], alg, alg / t["fused_fwd"], t["up_fwd"] + t["mse_fwd"] + t["mse_bwd"] + t["up_bwd"]))


if __name__ == "__main__":
    main()
