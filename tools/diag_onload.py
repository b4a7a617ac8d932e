#!/usr/bin/env python
"""Debug aid: first forward tensor that differs between PXL_BN_ONLOAD=1 and =0 (same weights, same input)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run(mode, layers, size, B, autotune):
    os.environ["PXL_BN_ONLOAD"] = mode
    os.environ["PXL_AUTOTUNE"] = autotune
    import torch_oracle as TO
    from pixelssl_amd.engine import DeepLabV2Core
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr
    x, gt = TO.synthetic_batch(B, size, B, seed=77, block=16)
    core = DeepLabV2Core(backbone=layers, device="cuda", engine_dtype=torch.bfloat16)
    core.reset_parameters(torch.Generator().manual_seed(5))
    core.train()
    core.keep_arena = True
    logits, _, _ = core(x.cuda())
    arena, pl = core._last_arena, core._cur
    outs = []
    for i, op in enumerate(core._pb.ops):
        if op.kind != 1:
            continue
        c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().pxl_net_tensor_shape(pl.net, op.out, ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
        v = torch.empty(B, c.value, h.value, w.value, device="cuda")
        check(lib().pxl_net_read_tensor(pl.net, ptr(arena), op.out, -1, None, ptr(v), stream_ptr()))
        outs.append((i, op.bn_in0, op.cin, op.cout, op.kh, op.stride, v.cpu()))
    return outs, logits.detach().cpu()


def main():
    layers = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,1,1,1").split(","))
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 97
    autotune = sys.argv[3] if len(sys.argv) > 3 else "1"
    a, la = run("1", layers, size, 3, autotune)
    b, lb = run("0", layers, size, 3, autotune)
    print("logits equal:", torch.equal(la, lb))
    for (i, bn, cin, cout, k, s, va), (_, _, _, _, _, _, vb) in zip(a, b):
        d = (va - vb).abs().max().item()
        print("op %3d bn_in %3d  %4d -> %4d k%d s%d  out %s  max|diff| %.4g  %s" % (i, bn, cin, cout, k, s, tuple(va.shape), d, "" if d == 0 else "<<<<"))


if __name__ == "__main__":
    main()
