"""The multi-rate head as ONE GEMM (csrc/aspp.hip; deeplab_v2.py:76-85: out = sum_g conv3x3(x, dilation d_g) + bias_g).

Forward: pxl_conv_dma_slabs (bf16: the fp32 partial-sum slab of a 1x1 convolution with a column per (group, class, tap)) /
pxl_conv_igemm (fp32) + pxl_aspp_col2im against the sum of four torch convolutions; backward: pxl_aspp_dp_gather + the two plain
GEMMs + pxl_aspp_dw_scatter against autograd.  All through the C-ABI."""
import ctypes
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_kernels import DEV, TOL, _ops, _pitch, from_nhwc, qround, rel_err, to_nhwc  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("geo", [(2, 128, 21, 17, 19, (1, 2, 3, 4)), (3, 256, 21, 33, 33, (6, 12, 18, 24)), (1, 64, 5, 9, 7, (2, 3))],
                         ids=["small", "aspp_33", "two_groups"])
def test_multirate_head_as_one_gemm(geo, dtype):
    ops = _ops()
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    B, Cin, Cout, H, W, rates = geo
    g = torch.Generator().manual_seed(Cin + H)
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(True)
    ws = [qround(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9 * len(rates)), dtype).requires_grad_(True) for _ in rates]
    bias = torch.randn(Cout, generator=g)
    ref = sum(F.conv2d(x, w, None, 1, r, r) for w, r in zip(ws, rates)) + bias.view(1, -1, 1, 1)
    ng, tpg = len(rates), 9
    GP = (Cout * tpg + 63) // 64 * 64
    J = ng * GP
    cop = _pitch(Cout)
    code = dtype_code(dtype)
    M = B * H * W
    taps = []
    for r in rates:
        taps += ops.fwd_taps(3, 3, r, r)
    dy = (ctypes.c_int16 * 64)(*([t[0] for t in taps] + [0] * (64 - len(taps))))
    dx = (ctypes.c_int16 * 64)(*([t[1] for t in taps] + [0] * (64 - len(taps))))
    # the groups' master weights [Cout][3][3][Cin] at scattered offsets of one flat buffer, as in the engine's parameter store
    nparam = Cout * 9 * Cin
    flat = torch.zeros(ng * (nparam + 5), device=DEV)
    offs = (ctypes.c_long * 4)(*([gi * (nparam + 5) + 2 for gi in range(ng)] + [0] * (4 - ng)))
    for gi, w in enumerate(ws):
        flat[offs[gi]:offs[gi] + nparam] = w.detach().permute(0, 2, 3, 1).reshape(-1).to(DEV)
    Wp = torch.full((J, Cin), float("nan"), device=DEV, dtype=dtype)
    Wd = torch.full((Cin, J), float("nan"), device=DEV, dtype=dtype)
    check(lib().pxl_aspp_pack(code, ptr(flat), offs, ng, GP, Cout, tpg, Cin, Cin, ptr(Wp), ptr(Wd), stream_ptr()))
    assert torch.equal(Wp.t().contiguous(), Wd) and torch.isfinite(Wp.float()).all()
    xd = to_nhwc(x.detach(), Cin, dtype)
    fdesc = ops.conv_desc(dtype, B, H, W, Cin, H, W, J, J, [(0, 0)], out_stride=1)
    P = torch.full((M, J), float("nan"), device=DEV, dtype=torch.float32)
    if dtype == torch.bfloat16:
        check(lib().pxl_conv_dma_slabs(fdesc, ptr(xd), ptr(Wp), ptr(P), P.numel() * 4, 1, stream_ptr()))
    else:
        ops.conv_igemm(fdesc, xd, Wp, P.view(B, H, W, J))
    out = torch.full((B, H, W, cop), 7.0, device=DEV, dtype=dtype)
    check(lib().pxl_aspp_col2im(code, B, H, W, J, GP, ng, Cout, tpg, dy, dx, ptr(P), 1, M * J, ptr(bias.to(DEV)), ptr(out), cop, stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cout), ref.detach()) < TOL[dtype]
    if cop > Cout:
        assert out[..., Cout:].float().abs().max().item() == 0.0
    # ---- backward
    dout = qround(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(dout)
    dod = to_nhwc(dout, cop, dtype)
    dP = torch.full((M, J), 3.0, device=DEV, dtype=dtype)
    check(lib().pxl_aspp_dp_gather(code, B, H, W, J, GP, ng, Cout, tpg, dy, dx, ptr(dod), cop, ptr(dP), stream_ptr()))
    bdesc = ops.conv_desc(dtype, B, H, W, J, H, W, Cin, Cin, [(0, 0)], out_stride=1, div=1)
    dxd = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
    ops.conv_igemm(bdesc, dP.view(B, H, W, J), Wd, dxd)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dxd, Cin), x.grad) < TOL[dtype]
    tmp = torch.zeros(J, Cin, device=DEV)
    ops.conv_wgrad(fdesc, xd, dP.view(B, H, W, J), tmp, Cin, Cin)
    grads = torch.ones(ng * (nparam + 5), device=DEV)                  # (the groups' weights are NOT contiguous in the flat buffer)
    check(lib().pxl_aspp_dw_scatter(ptr(tmp), ng, GP, Cout, tpg, Cin, Cin, ptr(grads), offs, stream_ptr()))
    torch.cuda.synchronize()
    for gi, w in enumerate(ws):
        got = grads[offs[gi]:offs[gi] + nparam].cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2) - 1.0       # (+= into ones)
        assert rel_err(got, w.grad) < (2e-2 if dtype == torch.bfloat16 else 1e-4), (gi, rel_err(got, w.grad))
    assert float(grads[:2].sum()) == 2.0                              # nothing outside the groups' rows was touched
