"""Halo-tile convolution (csrc/conv_halo_kernel.h, tile configurations 40..43): "same" stride-1 3x3 convolutions whose pixel
slab is DMA'd once per 64-channel slab and walked by all nine taps from LDS.

Its K order is slab-major (the tap-per-step kernel's is tap-major), so results equal torch's fp32 convolution within the bf16
bar instead of the other kernels bit for bit; what IS bit-exact is checked bit-exactly: the padding positions of the padded
grid never reach memory or the statistics, the activated tensor a BN-on-load launch writes equals pxl_bn_apply_fwd, the
BatchNorm coefficients equal pxl_bn_finalize, the fused BatchNorm-backward sums are the sums of the tensor that was stored.
(Measured slower than the tap-per-step kernel on every ResNet-101 shape -- profiles/r06_a_halo_cbench.txt -- and therefore
not offered to the tuner; selectable through desc.tile_cfg.)
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_gpu_kernels import DEV, TOL, _ops, _pitch, from_nhwc, pack_w, qround, rel_err, to_nhwc  # noqa: E402

HALO_CASES = [
    # name, B, Cin, Cout, H, W, dilation        (3x3, stride 1, padding = dilation)
    ("one_slab", 2, 64, 64, 17, 17, 1),
    ("two_slabs_ragged", 3, 128, 72, 9, 13, 1),
    ("four_slabs_33", 2, 256, 256, 33, 33, 1),
    ("d2", 2, 128, 128, 19, 17, 2),
    ("tiny", 1, 192, 40, 3, 5, 1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [40, 41, 42, 43])
@pytest.mark.parametrize("case", HALO_CASES, ids=[c[0] for c in HALO_CASES])
def test_halo_forward_statistics_and_data_gradient(case, cfg):
    ops = _ops()
    dtype = torch.bfloat16
    name, B, Cin, Cout, H, W, d = case
    g = torch.Generator().manual_seed(len(name) * 31 + cfg)
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(True)
    w = qround(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype).requires_grad_(True)
    y0 = F.conv2d(x, w, None, 1, d, d)
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(3, 3, d, d)
    xd = to_nhwc(x.detach(), cip, dtype)
    wf, wt = pack_w(w.detach(), dtype, cip, kp=cop)
    desc = ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, taps, out_stride=1, tile_cfg=cfg, stats_rep=3)
    out = torch.full((B, H, W, cop), 7.0, device=DEV, dtype=dtype)
    ops.conv_igemm(desc, xd, wf, out)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cout), y0.detach()) < TOL[dtype], (name, cfg)
    if cop > Cout:
        assert out[..., Cout:].float().abs().max().item() == 0.0
    # statistics of the STORED values: the padded grid's padding positions hold real accumulators and must stay out
    stats = torch.zeros(3, 2 * Cout, device=DEV)
    out2 = torch.empty_like(out)
    ops.conv_igemm(desc, xd, wf, out2, stats=stats)
    torch.cuda.synchronize()
    assert torch.equal(out2[..., :Cout], out[..., :Cout])
    got = from_nhwc(out2, Cout)
    folded = stats.sum(0).cpu()
    assert rel_err(folded[:Cout], got.sum(dim=(0, 2, 3))) < 1e-4
    assert rel_err(folded[Cout:], (got * got).sum(dim=(0, 2, 3))) < 1e-4
    # data gradient = the same kernel over mirrored taps and transposed weights
    if cop % 64 == 0:
        dy = qround(torch.randn(y0.shape, generator=g), dtype)
        y0.backward(dy)
        dx = torch.empty(B, H, W, cip, device=DEV, dtype=dtype)
        bdesc = ops.conv_desc(dtype, B, H, W, cop, H, W, cip, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=1, tile_cfg=cfg)
        ops.conv_igemm(bdesc, to_nhwc(dy, cop, dtype), wt, dx)
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(dx, Cin), x.grad) < TOL[dtype], ("dgrad", name, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("cfg", [40, 41, 43])
@pytest.mark.parametrize("case", HALO_CASES[:4], ids=[c[0] for c in HALO_CASES[:4]])
def test_halo_bn_apply_on_load_once_per_element(case, cfg, training):
    """BN-on-load for a 3x3 consumer: relu(bn(y)) applied to the slab in LDS (once per element, not once per tap), z written
    from the centre rows by the workgroups of output-channel tile 0."""
    ops = _ops()
    dtype = torch.bfloat16
    name, B, Cin, Cout, H, W, d = case
    g = torch.Generator().manual_seed(len(name) * 17 + cfg)
    y = qround(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.4, dtype)
    w = qround(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype)
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.5 + 0.3).to(DEV)
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(3, 3, d, d)
    yd = to_nhwc(y, cip, dtype)
    wf, _ = pack_w(w, dtype, cip)
    nrep, count = 4, float(B * H * W)
    yf = yd.float().reshape(-1, cip)
    st = torch.zeros(nrep, 2 * Cin, device=DEV)
    for r in range(nrep):
        part = yf[r::nrep]
        st[r, :Cin], st[r, Cin:] = part.sum(0), (part * part).sum(0)
    rm_a, rv_a = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_a = ops.bn_finalize(st, count, gamma, beta, rm_a, rv_a, nrep=nrep, training=training)
    z = ops.bn_apply_fwd(yd, coef_a, relu=True)
    ref = F.conv2d(from_nhwc(z, Cin), w, None, 1, d, d)
    desc = ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, taps, out_stride=1, tile_cfg=cfg, stats_rep=4)
    rm_b, rv_b = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_b = torch.full((4 * Cin,), float("nan"), device=DEV)
    fin = ops.bn_fin(st, nrep, count, gamma, beta, rm_b, rv_b, coef_b, training=training)
    out = torch.full((B, H, W, cop), 5.0, device=DEV, dtype=dtype)
    st_b = torch.zeros(4, 2 * Cout, device=DEV)
    z_b = torch.full_like(yd, 9.0)
    ops.conv_dma_bnin(desc, yd, wf, out, fin, relu=True, stats=st_b, z=z_b)
    torch.cuda.synchronize()
    assert torch.equal(z_b, z), "every pixel is the centre of exactly one tile"
    assert torch.equal(coef_a, coef_b) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    got = from_nhwc(out, Cout)
    assert rel_err(got, ref) < TOL[dtype], (name, cfg)          # padding taps stay zero: relu(shift) must not leak into them
    folded = st_b.sum(0).cpu()
    assert rel_err(folded[:Cout], got.sum(dim=(0, 2, 3))) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [40, 41])
def test_halo_data_gradient_with_fused_bn_backward_sums(cfg):
    ops = _ops()
    dtype = torch.bfloat16
    B, Cin, Cout, H, W = 2, 128, 64, 17, 19
    g = torch.Generator().manual_seed(77 + cfg)
    w = qround(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9), dtype)
    taps = ops.fwd_taps(3, 3, 1, 1)
    dy = to_nhwc(qround(torch.randn(B, Cout, H, W, generator=g), dtype), Cout, dtype)
    _, wt = pack_w(w, dtype, Cin, kp=Cout)
    y = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    add = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    coef = torch.cat([torch.randn(Cin, generator=g) * 0.1, torch.rand(Cin, generator=g) + 0.5,
                      torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3]).to(DEV)
    mk = lambda c: ops.conv_desc(dtype, B, H, W, Cout, H, W, Cin, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=1, tile_cfg=c)
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    for relu in (1, 0):
        for addend in (None, add):
            ref = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
            ops.conv_igemm(mk(-1), dy, wt, ref, addend=addend)
            got = torch.empty_like(ref)
            sums = torch.zeros(2 * Cin, device=DEV)
            ops.conv_dgrad_bnreduce(mk(cfg), dy, wt, got, y, coef, relu, sums, addend=addend)
            rs = torch.zeros(2 * Cin, device=DEV)       # the sums of the tensor the halo launch STORED
            check(lib().pxl_bn_bwd_reduce(dtype_code(dtype), B * H * W, Cin, ptr(got), ptr(y), ptr(coef), relu, ptr(rs), 1, stream_ptr()))
            torch.cuda.synchronize()
            assert rel_err(got.float().cpu(), ref.float().cpu()) < TOL[dtype]
            assert rel_err(sums.cpu(), rs.cpu()) < 1e-4, (cfg, relu, addend is not None)
