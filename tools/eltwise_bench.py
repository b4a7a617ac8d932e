"""Achieved HBM bandwidth of the row-streaming kernels (BN backward reduce / apply, relu(bn(y)) materialisation,
residual join, ReLU mask) on the tensor shapes of the ResNet-101 trunk at 8x513x513.  Run on the GPU box:

    python tools/eltwise_bench.py [--dtype bf16]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelssl_amd import ops  # noqa: E402

SHAPES = [(8 * 257 * 257, 64), (8 * 129 * 129, 64), (8 * 129 * 129, 256), (8 * 65 * 65, 128), (8 * 65 * 65, 512),
          (8 * 33 * 33, 256), (8 * 33 * 33, 1024), (8 * 33 * 33, 512), (8 * 33 * 33, 2048)]


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3      # us


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--sweep", action="store_true", help="sweep the target block count of every kernel (pxl_tune_set)")
    a = p.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if a.dtype == "bf16" else 4
    dev = "cuda"
    print("%-16s %10s %10s %10s %10s %10s   (us | GB/s of algorithmic traffic)" %
          ("M x C", "bwd_reduce", "bwd_apply", "apply_fwd", "residual", "relu_mask"))
    for M, C in SHAPES:
        y = torch.randn(M, C, device=dev).to(dt)
        dz = torch.randn(M, C, device=dev).to(dt)
        res = torch.randn(M, C, device=dev).to(dt)
        coef = torch.randn(4 * C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        sums = torch.zeros(2 * C, device=dev)
        from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
        code = dtype_code(dt)
        dy = torch.empty_like(dz)

        def red():
            check(lib().pxl_bn_bwd_reduce(code, M, C, ptr(dz), ptr(y), ptr(coef), 1, ptr(sums), 1, stream_ptr()))

        def app():
            check(lib().pxl_bn_bwd_apply_fused(code, M, C, ptr(dz), ptr(y), ptr(coef), ptr(sums), float(M), 1, 1, ptr(dg),
                                               ptr(db), ptr(dy), stream_ptr()))
        fns = [red, app, lambda: ops.residual_fwd(y, coef, res, coef), lambda: ops.bn_apply_fwd(y, coef)]
        if a.sweep:
            targets = (256, 512, 768, 1024, 1536, 2048, 4096)
            defaults = (1024, 2048, 2048, 2048)
            for key, name in enumerate(("bwd_reduce", "bwd_apply", "residual", "apply_fwd")):
                row = []
                for tg in targets:
                    check(lib().pxl_tune_set(key, tg))
                    row.append(timeit(fns[key], reps=10))
                check(lib().pxl_tune_set(key, defaults[key]))
                print("  %-14s %-11s " % ("%dx%d" % (M, C), name) + " ".join("%d:%5.1f" % (tg, us) for tg, us in zip(targets, row)))
            for cg in (32, 64, 128):
                check(lib().pxl_tune_set(4, cg))
                for key, name in enumerate(("bwd_reduce", "bwd_apply", "residual", "apply_fwd")):
                    row = []
                    for tg in (256, 512, 1024):
                        check(lib().pxl_tune_set(key, tg))
                        row.append(timeit(fns[key], reps=10))
                    check(lib().pxl_tune_set(key, defaults[key]))
                    print("  %-14s %-11s cg<=%-3d " % ("%dx%d" % (M, C), name, cg) + " ".join("%d:%5.1f" % (tg, us) for tg, us in zip((256, 512, 1024), row)))
            check(lib().pxl_tune_set(4, 16))
            continue
        t = [timeit(red), timeit(app), timeit(lambda: ops.bn_apply_fwd(y, coef)), timeit(lambda: ops.residual_fwd(y, coef, res, coef)),
             timeit(lambda: ops.relu_mask(dz, y, second=True))]
        traffic = [2, 3, 2, 3, 4]                  # tensors read + written per kernel
        cells = ["%6.1f|%5.0f" % (us, traffic[i] * M * C * es / us / 1e3) for i, us in enumerate(t)]
        print("%-16s %s" % ("%dx%d" % (M, C), " ".join("%12s" % c for c in cells)))


if __name__ == "__main__":
    main()
