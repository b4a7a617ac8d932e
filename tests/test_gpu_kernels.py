"""GPU parity of every libpixelhip kernel (called through the C-ABI via ctypes) against plain fp32
PyTorch on the CPU (the same operator the reference uses at each call site).  Seeded inputs, odd
sizes (513-style 16k+1 extents scaled down), ragged tiles, every tile configuration."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from pixelssl_amd import ops
    return ops


def to_nhwc(x, cp, dtype):
    """NCHW cpu fp32 -> NHWC device tensor with channel pitch cp."""
    b, c, h, w = x.shape
    y = torch.zeros(b, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(DEV).to(dtype).contiguous()


def from_nhwc(y, c):
    return y[..., :c].float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_w(w, dtype, cp, kp=None):
    """OIHW cpu -> fwd [K][T][Cp] and dgrad [C][T][Kp] device tensors via pxl_pack_weights."""
    ops = _ops()
    k, c, kh, kw = w.shape
    master = w.permute(0, 2, 3, 1).contiguous().to(DEV)           # [K][kh][kw][C]
    wf = torch.empty(k, kh * kw, cp, device=DEV, dtype=dtype)
    wt = torch.empty(c, kh * kw, kp, device=DEV, dtype=dtype) if kp else None
    ops.pack_weights(dtype, master, k, kh * kw, c, wf, cp, wt=wt, Kp=kp or 0)
    return wf, wt


def rel_err(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def qround(x, dtype):
    return x.to(dtype).float() if dtype == torch.bfloat16 else x


TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3}

CONV_CASES = [
    # name, B, Cin, Cout, H, W, k, stride, dil, pad
    ("1x1", 2, 64, 256, 17, 17, 1, 1, 1, 0),
    ("1x1_ragged", 3, 96, 72, 9, 13, 1, 1, 1, 0),
    ("1x1_s2", 2, 64, 128, 17, 17, 1, 2, 1, 0),
    ("3x3", 2, 64, 64, 17, 17, 3, 1, 1, 1),
    ("3x3_s2", 2, 32, 64, 17, 17, 3, 2, 1, 1),
    ("3x3_d2", 2, 32, 32, 9, 9, 3, 1, 2, 2),
    ("3x3_d4", 1, 64, 160, 9, 9, 3, 1, 4, 4),
    ("7x7_stem", 2, 3, 64, 33, 33, 7, 2, 1, 3),
    ("4x4_s2", 2, 24, 64, 33, 33, 4, 2, 1, 1),
    ("cout21", 2, 128, 21, 9, 9, 3, 1, 1, 1),
]


def _seed(name):
    return sum(ord(ch) for ch in name)


def _pitch(c):
    return 8 if c <= 8 else (c + 31) // 32 * 32


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize("cfg", [-1, 0, 1, 2, 3, 4, 5, 6, 7])
def test_conv_forward(dtype, case, cfg):
    ops = _ops()
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    ref = F.conv2d(qround(x, dtype), qround(w, dtype), None, s, p, d)
    Ho, Wo = ref.shape[2:]
    cip, cop = _pitch(Cin), _pitch(Cout)
    xd = to_nhwc(x, cip, dtype)
    wf, _ = pack_w(w, dtype, cip)
    out = torch.full((B, Ho, Wo, cop), 7.0, device=DEV, dtype=dtype)
    desc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, ops.fwd_taps(k, k, d, p), out_stride=s, tile_cfg=cfg)
    ops.conv_igemm(desc, xd, wf, out)
    torch.cuda.synchronize()
    got = from_nhwc(out, Cout)
    assert rel_err(got, ref) < TOL[dtype], "%s cfg %d" % (name, cfg)
    if cop > Cout:   # padded channels are written as zeros
        assert out[..., Cout:].float().abs().max().item() == 0.0


DMA_CASES = [
    # name, B, Cin, Cout, H, W, k, stride, dil, pad   (Cin % 64 == 0: the LDS-DMA kernel, conv_dma.hip)
    ("dma_1x1", 2, 64, 256, 17, 17, 1, 1, 1, 0),
    ("dma_1x1_k256", 3, 256, 72, 9, 13, 1, 1, 1, 0),
    ("dma_1x1_s2", 2, 128, 160, 17, 17, 1, 2, 1, 0),
    ("dma_3x3", 2, 64, 64, 17, 17, 3, 1, 1, 1),
    ("dma_3x3_s2", 2, 64, 96, 17, 17, 3, 2, 1, 1),
    ("dma_3x3_d2", 2, 128, 128, 9, 9, 3, 1, 2, 2),
    ("dma_3x3_d4_wide", 1, 64, 288, 9, 9, 3, 1, 4, 4),
    ("dma_4x4_s2", 2, 64, 64, 33, 33, 4, 2, 1, 1),
    ("dma_tiny_m", 1, 192, 40, 3, 5, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", DMA_CASES, ids=[c[0] for c in DMA_CASES])
@pytest.mark.parametrize("cfg", [-1, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27,
                                 28, 29, 30, 31, 32, 33, 34, 35])
def test_conv_dma_forward_and_dgrad(case, cfg):
    """LDS-DMA kernel: forward (every tile configuration, 3- and 4-stage rings), bias + addend + BN statistics
    on the coalesced read-back pass, and the data gradient of stride-1 convolutions."""
    _dma_forward_and_dgrad(case, cfg, torch.bfloat16)


@pytest.mark.parametrize("case", DMA_CASES, ids=[c[0] for c in DMA_CASES])
@pytest.mark.parametrize("cfg", [-1, 8, 9, 10, 11, 16, 17, 18, 19])
def test_conv_dma_f32_forward_and_dgrad(case, cfg):
    """fp32 LDS-DMA kernel (csrc/conv_dma_f32.hip, v_mfma_f32_32x32x2_f32): the same checks against torch's fp32 convolution."""
    _dma_forward_and_dgrad(case, cfg, torch.float32)


def _dma_forward_and_dgrad(case, cfg, dtype):
    ops = _ops()
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name) + 7)
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(True)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype).requires_grad_(True)
    bias = torch.randn(Cout, generator=g)
    y0 = F.conv2d(x, w, None, s, p, d)
    Ho, Wo = y0.shape[2:]
    add = qround(torch.randn(B, Cout, Ho, Wo, generator=g), dtype)
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(k, k, d, p)
    xd = to_nhwc(x.detach(), cip, dtype)
    wf, wt = pack_w(w.detach(), dtype, cip, kp=cop)
    # plain forward
    out = torch.full((B, Ho, Wo, cop), 7.0, device=DEV, dtype=dtype)
    desc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, taps, out_stride=s, tile_cfg=cfg, stats_rep=3)
    ops.conv_igemm(desc, xd, wf, out)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cout), y0.detach()) < TOL[dtype], "%s cfg %d" % (name, cfg)
    if cop > Cout:
        assert out[..., Cout:].float().abs().max().item() == 0.0
    # bias + addend + statistics
    ref = qround(qround(y0.detach(), dtype) + add + bias.view(1, -1, 1, 1), dtype)
    stats = torch.zeros(3, 2 * Cout, device=DEV)
    out2 = torch.empty(B, Ho, Wo, cop, device=DEV, dtype=dtype)
    ops.conv_igemm(desc, xd, wf, out2, bias=bias.to(DEV), addend=to_nhwc(add, cop, dtype), stats=stats)
    torch.cuda.synchronize()
    got = from_nhwc(out2, Cout)
    assert rel_err(got, ref) < TOL[dtype], "%s cfg %d epilogue" % (name, cfg)
    folded = stats.sum(0).cpu()
    assert rel_err(folded[:Cout], got.sum(dim=(0, 2, 3))) < 1e-4          # statistics of the STORED values
    assert rel_err(folded[Cout:], (got * got).sum(dim=(0, 2, 3))) < 1e-4
    # data gradient (stride-1 convolutions run on the DMA kernel, strided ones on the generic kernel)
    dy = qround(torch.randn(y0.shape, generator=g), dtype)
    y0.backward(dy)
    dx = torch.empty(B, H, W, cip, device=DEV, dtype=dtype)
    # (stride-2 data gradients run on the DMA kernel too: odd source coordinates are zero-fill lanes)
    bdesc = ops.conv_desc(dtype, B, Ho, Wo, cop, H, W, cip, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=s,
                          tile_cfg=cfg if cop % 64 == 0 else -1)
    ops.conv_igemm(bdesc, to_nhwc(dy, cop, dtype), wt, dx)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx, Cin), x.grad) < TOL[dtype], "dgrad %s cfg %d" % (name, cfg)


@pytest.mark.parametrize("cfg", [-1, 8, 10, 18, 20, 24, 26, 28, 31, 34, 35])
@pytest.mark.parametrize("case", DMA_CASES[:7], ids=[c[0] for c in DMA_CASES[:7]])
def test_conv_with_last_block_bn_finalize(case, cfg):
    """pxl_conv_dma_finalize == pxl_conv_igemm + pxl_bn_finalize: same output, same (mean, rstd, scale, shift), same
    running statistics -- the workgroup that draws the last ticket sees every other workgroup's statistics atomics.
    Repeated launches (fresh counters) give the same coefficients: no stale reads."""
    ops = _ops()
    dtype = torch.bfloat16
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name) + 3)
    x = qround(torch.randn(B, Cin, H, W, generator=g) + 0.3, dtype)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.2).to(DEV)
    Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(k, k, d, p)
    xd = to_nhwc(x, cip, dtype)
    wf, _ = pack_w(w, dtype, cip)
    nrep, count = 4, float(B * Ho * Wo)
    desc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, taps, out_stride=s, tile_cfg=cfg, stats_rep=nrep)
    # reference path: conv + statistics, then the finalize launch
    out_a = torch.empty(B, Ho, Wo, cop, device=DEV, dtype=dtype)
    st_a = torch.zeros(nrep, 2 * Cout, device=DEV)
    rm_a, rv_a = torch.full((Cout,), 0.25, device=DEV), torch.full((Cout,), 1.5, device=DEV)
    ops.conv_igemm(desc, xd, wf, out_a, stats=st_a)
    coef_a = ops.bn_finalize(st_a, count, gamma, beta, rm_a, rv_a, nrep=nrep)
    for trial in range(3):
        out_b = torch.empty_like(out_a)
        st_b = torch.zeros(nrep, 2 * Cout, device=DEV)
        cnt = torch.zeros(4, device=DEV, dtype=torch.int32)
        rm_b, rv_b = torch.full((Cout,), 0.25, device=DEV), torch.full((Cout,), 1.5, device=DEV)
        coef_b = torch.full((4 * Cout,), float("nan"), device=DEV)
        fin = ops.bn_fin(st_b, nrep, count, gamma, beta, rm_b, rv_b, coef_b)
        ops.conv_dma_finalize(desc, xd, wf, out_b, st_b, fin, cnt)
        torch.cuda.synchronize()
        assert torch.equal(out_a, out_b)
        assert torch.isfinite(coef_b).all()
        # the sums differ only in fp32 atomic order
        assert rel_err(coef_b.cpu(), coef_a.cpu()) < 2e-6, (name, cfg, trial)
        assert rel_err(rm_b.cpu(), rm_a.cpu()) < 2e-6 and rel_err(rv_b.cpu(), rv_a.cpu()) < 2e-6
        assert cnt[0].item() > 0


BNIN_CASES = [
    # name, B, Cin, Cout, H, W, k, stride, dil, pad
    ("bnin_1x1", 2, 64, 256, 17, 17, 1, 1, 1, 0),
    ("bnin_1x1_k256", 3, 256, 72, 9, 13, 1, 1, 1, 0),
    ("bnin_1x1_k512_m8712", 8, 512, 128, 33, 33, 1, 1, 1, 0),
    ("bnin_3x3", 2, 64, 64, 17, 17, 3, 1, 1, 1),
    ("bnin_3x3_s2", 2, 128, 96, 17, 17, 3, 2, 1, 1),
    ("bnin_3x3_d2", 2, 128, 128, 9, 9, 3, 1, 2, 2),
    ("bnin_3x3_d4_wide", 1, 64, 288, 9, 9, 3, 1, 4, 4),
    ("bnin_tiny_m", 1, 192, 40, 3, 5, 3, 1, 1, 1),
]


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("cfg", [-1, 8, 9, 10, 11, 17, 18, 20, 21, 25, 26, 28, 29, 30, 33, 34, 35])
@pytest.mark.parametrize("case", BNIN_CASES, ids=[c[0] for c in BNIN_CASES])
def test_conv_with_bn_apply_on_load(case, cfg, training):
    """pxl_conv_dma_bnin(y, BN) == pxl_bn_finalize + pxl_bn_apply_fwd + pxl_conv_igemm BIT FOR BIT (the tile transformed in
    LDS is rounded to bf16 exactly like the materialised activation), padding taps stay zero (relu(shift) must not leak
    into them), coef / running statistics as pxl_bn_finalize writes them, output statistics as pxl_conv_igemm's."""
    ops = _ops()
    dtype = torch.bfloat16
    name, B, Cin, Cout, H, W, k, s, d, p = case
    if cfg >= 20 and cfg != 29 and _pitch(Cout) < 128:
        pytest.skip("tall / 8-wave tiles are 128 channels wide")
    g = torch.Generator().manual_seed(_seed(name) + 11)
    y = qround(torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.4, dtype)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.5 + 0.3).to(DEV)
    Ho, Wo = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(k, k, d, p)
    yd = to_nhwc(y, cip, dtype)
    wf, _ = pack_w(w, dtype, cip)
    nrep, count = 4, float(B * H * W)
    # statistics of y spread over the replicas (what the producing convolution's epilogue leaves behind)
    yf = yd.float().reshape(-1, cip)
    st = torch.zeros(nrep, 2 * Cin, device=DEV)
    for r in range(nrep):
        part = yf[r::nrep]
        st[r, :Cin], st[r, Cin:] = part.sum(0), (part * part).sum(0)
    desc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, taps, out_stride=s, tile_cfg=cfg, stats_rep=4)
    rm_a, rv_a = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_a = ops.bn_finalize(st, count, gamma, beta, rm_a, rv_a, nrep=nrep, training=training)
    z = ops.bn_apply_fwd(yd, coef_a, relu=True)
    out_a = torch.full((B, Ho, Wo, cop), 3.0, device=DEV, dtype=dtype)
    st_a = torch.zeros(4, 2 * Cout, device=DEV)
    ops.conv_igemm(desc, z, wf, out_a, stats=st_a)
    rm_b, rv_b = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_b = torch.full((4 * Cin,), float("nan"), device=DEV)
    fin = ops.bn_fin(st, nrep, count, gamma, beta, rm_b, rv_b, coef_b, training=training)
    out_b = torch.full((B, Ho, Wo, cop), 5.0, device=DEV, dtype=dtype)
    st_b = torch.zeros(4, 2 * Cout, device=DEV)
    plain = k == 1 and s == 1                  # 1x1 / stride 1: the kernel can also write the activated tensor on the way
    z_b = torch.full_like(yd, 9.0) if plain else None
    ops.conv_dma_bnin(desc, yd, wf, out_b, fin, relu=True, stats=st_b, z=z_b)
    torch.cuda.synchronize()
    if plain:
        assert torch.equal(z_b, z), "activated tensor written by the workgroups of output-channel tile 0"
    else:
        with pytest.raises(Exception):         # a gathering convolution cannot materialise z (each pixel is visited per tap)
            ops.conv_dma_bnin(desc, yd, wf, out_b, fin, relu=True, z=torch.empty_like(yd))
    assert torch.equal(coef_a, coef_b), "coef written by workgroup 0 == pxl_bn_finalize"
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert torch.equal(out_a[..., :Cout], out_b[..., :Cout]), (name, cfg, rel_err(out_b.float().cpu(), out_a.float().cpu()))
    assert rel_err(st_b.sum(0).cpu(), st_a.sum(0).cpu()) < 2e-6
    # a non-ReLU BatchNorm in front (relu = 0) on one configuration
    if cfg == -1:
        z0 = ops.bn_apply_fwd(yd, coef_a, relu=False)
        ops.conv_igemm(desc, z0, wf, out_a)
        ops.conv_dma_bnin(desc, yd, wf, out_b, fin, relu=False)
        torch.cuda.synchronize()
        assert torch.equal(out_a[..., :Cout], out_b[..., :Cout])


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("cfg", [-1, 8, 9, 10, 11, 16, 17, 18, 19])
@pytest.mark.parametrize("case", BNIN_CASES[:3], ids=[c[0] for c in BNIN_CASES[:3]])
def test_conv_f32_with_bn_apply_on_load(case, cfg, training):
    """fp32 engine, round 6 (csrc/conv_dma_f32.hip BNIN): pxl_conv_dma_bnin(y, BN) against pxl_bn_finalize + pxl_bn_apply_fwd +
    pxl_conv_igemm.  fp32 keeps every bit of the transformed tile, so the two paths may differ by the contraction of
    scale * y + shift (fma or not) in the last place: 1e-6 instead of bit equality; coefficients and running statistics equal."""
    ops = _ops()
    dtype = torch.float32
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name) + 13)
    y = torch.randn(B, Cin, H, W, generator=g) * 1.5 + 0.4
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.5 + 0.3).to(DEV)
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(k, k, d, p)
    yd = to_nhwc(y, cip, dtype)
    wf, _ = pack_w(w, dtype, cip)
    nrep, count = 4, float(B * H * W)
    yf = yd.reshape(-1, cip)
    st = torch.zeros(nrep, 2 * Cin, device=DEV)
    for r in range(nrep):
        part = yf[r::nrep]
        st[r, :Cin], st[r, Cin:] = part.sum(0), (part * part).sum(0)
    desc = ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, taps, out_stride=1, tile_cfg=cfg, stats_rep=4)
    rm_a, rv_a = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_a = ops.bn_finalize(st, count, gamma, beta, rm_a, rv_a, nrep=nrep, training=training)
    z = ops.bn_apply_fwd(yd, coef_a, relu=True)
    out_a = torch.full((B, H, W, cop), 3.0, device=DEV, dtype=dtype)
    st_a = torch.zeros(4, 2 * Cout, device=DEV)
    ops.conv_igemm(desc, z, wf, out_a, stats=st_a)
    rm_b, rv_b = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
    coef_b = torch.full((4 * Cin,), float("nan"), device=DEV)
    fin = ops.bn_fin(st, nrep, count, gamma, beta, rm_b, rv_b, coef_b, training=training)
    out_b = torch.full((B, H, W, cop), 5.0, device=DEV, dtype=dtype)
    st_b = torch.zeros(4, 2 * Cout, device=DEV)
    z_b = torch.full_like(yd, 9.0)
    ops.conv_dma_bnin(desc, yd, wf, out_b, fin, relu=True, stats=st_b, z=z_b)
    torch.cuda.synchronize()
    assert torch.equal(coef_a, coef_b) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert rel_err(z_b.cpu(), z.cpu()) < 1e-6 and (z_b == 0).float().mean().item() > 0.1
    assert rel_err(out_b[..., :Cout].cpu(), out_a[..., :Cout].cpu()) < 2e-6, (name, cfg)
    assert rel_err(st_b.sum(0).cpu(), st_a.sum(0).cpu()) < 1e-5


_FUSED_CFGS = [(c, torch.bfloat16) for c in (-1, 8, 10, 11, 18, 20, 21, 26, 28, 30, 34)] + \
              [(c, torch.float32) for c in (-1, 8, 11, 17, 18)]


@pytest.mark.parametrize("cfg,dtype", _FUSED_CFGS, ids=["%s-%d" % ("f32" if t == torch.float32 else "bf16", c) for c, t in _FUSED_CFGS])
@pytest.mark.parametrize("shape", [(2, 128, 64, 17, 19, 3, 1, 1), (3, 64, 256, 9, 13, 1, 1, 0), (2, 192, 128, 12, 12, 3, 2, 2)])
def test_conv_dgrad_with_fused_bn_backward_reduce(shape, cfg, dtype):
    """pxl_conv_dgrad_bnreduce == pxl_conv_igemm (data gradient) followed by pxl_bn_bwd_reduce over the tensor just
    written: identical din, and the [sum gd, sum gd*xhat] vectors of the stored (rounded) values, with and without an
    addend and the ReLU mask.  bf16 and fp32 LDS-DMA kernels."""
    ops = _ops()
    B, Cin, Cout, H, W, k, d, p = shape
    g = torch.Generator().manual_seed(B * 100 + Cin + k)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    taps = ops.fwd_taps(k, k, d, p)
    Ho, Wo = H + 2 * p - d * (k - 1), W + 2 * p - d * (k - 1)
    dy = to_nhwc(qround(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), Cout, dtype)
    _, wt = pack_w(w, dtype, Cin, kp=Cout)
    y = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)           # BN input of the tensor
    add = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    coef = torch.cat([torch.randn(Cin, generator=g) * 0.1, torch.rand(Cin, generator=g) + 0.5,
                      torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3]).to(DEV)
    bdesc = ops.conv_desc(dtype, B, Ho, Wo, Cout, H, W, Cin, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=1, tile_cfg=cfg)
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    for relu in (1, 0):
        for addend in (None, add):
            ref = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
            ops.conv_igemm(bdesc, dy, wt, ref, addend=addend)
            rs = torch.zeros(2 * Cin, device=DEV)
            check(lib().pxl_bn_bwd_reduce(dtype_code(dtype), B * H * W, Cin, ptr(ref), ptr(y), ptr(coef), relu, ptr(rs), 1,
                                          stream_ptr()))
            got = torch.empty_like(ref)
            sums = torch.zeros(2 * Cin, device=DEV)
            ops.conv_dgrad_bnreduce(bdesc, dy, wt, got, y, coef, relu, sums, addend=addend)
            torch.cuda.synchronize()
            assert torch.equal(got, ref)
            assert rel_err(sums.cpu(), rs.cpu()) < 1e-4, (shape, cfg, relu, addend is not None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("cfg", [-1, 10, 17, 18])
@pytest.mark.parametrize("shape", [(2, 128, 64, 17, 19, 3, 1), (3, 64, 128, 12, 13, 4, 1), (2, 128, 192, 9, 9, 1, 0)],
                         ids=["3x3s2", "4x4s2", "1x1s2"])
def test_stride2_dgrad_per_parity_class_with_fused_epilogue(shape, cfg, dtype, monkeypatch):
    """Data gradient of a stride-2 convolution as one launch per output-parity class (csrc/conv_dma.hip: output sub-grid + tap
    subset; tap-less classes of the 1x1 = zero-step launches), with the addend and the fused BatchNorm-backward sums reading
    through the same sub-grid row addresses: equal to the single launch that walks every tap (PXL_S2_CLASSES=0 is read once per
    process, so the single-launch reference is the GENERIC kernel, tile_cfg 1: another summation order) + a separate reduction;
    odd sizes."""
    ops = _ops()
    B, Cin, Cout, H, W, k, p = shape
    g = torch.Generator().manual_seed(B * 100 + Cin + k + 11)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    taps = ops.fwd_taps(k, k, 1, p)
    Ho, Wo = (H + 2 * p - k) // 2 + 1, (W + 2 * p - k) // 2 + 1
    dy = to_nhwc(qround(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), Cout, dtype)
    _, wt = pack_w(w, dtype, Cin, kp=Cout)
    y = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    add = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    coef = torch.cat([torch.randn(Cin, generator=g) * 0.1, torch.rand(Cin, generator=g) + 0.5,
                      torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3]).to(DEV)
    mk = lambda c: ops.conv_desc(dtype, B, Ho, Wo, Cout, H, W, Cin, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=2, tile_cfg=c)
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    for relu in (1, 0):
        for addend in (None, add):
            ref = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
            ops.conv_igemm(mk(1), dy, wt, ref, addend=addend)                      # generic kernel: every tap, parity test
            got = torch.full((B, H, W, Cin), 7.0, device=DEV, dtype=dtype)
            sums = torch.zeros(2 * Cin, device=DEV)
            ops.conv_dgrad_bnreduce(mk(cfg), dy, wt, got, y, coef, relu, sums, addend=addend)
            rs = torch.zeros(2 * Cin, device=DEV)            # the separate reduction over the tensor the class launches STORED
            check(lib().pxl_bn_bwd_reduce(dtype_code(dtype), B * H * W, Cin, ptr(got), ptr(y), ptr(coef), relu, ptr(rs), 1,
                                          stream_ptr()))
            torch.cuda.synchronize()
            # (another summation order: fp32 1e-6; bf16 one rounding step of the stored value)
            assert rel_err(got.float().cpu(), ref.float().cpu()) < (1e-6 if dtype == torch.float32 else 4e-3), (shape, cfg, relu)
            assert rel_err(sums.cpu(), rs.cpu()) < 1e-4, (shape, cfg, relu, addend is not None)
    # against torch: the transposed convolution
    xg = torch.zeros(B, Cin, H, W, requires_grad=True)
    F.conv2d(xg, w.float(), None, 2, p).backward(from_nhwc(dy, Cout))
    plain = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
    ops.conv_igemm(mk(cfg), dy, wt, plain)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(plain, Cin), xg.grad) < TOL[dtype], (shape, cfg)


_JOIN_CFGS = [(c, torch.bfloat16) for c in (-1, 9, 10, 18, 20, 26, 28, 31, 35)] + [(c, torch.float32) for c in (-1, 9, 10, 16, 19)]


@pytest.mark.parametrize("cfg,dtype", _JOIN_CFGS, ids=["%s-%d" % ("f32" if t == torch.float32 else "bf16", c) for c, t in _JOIN_CFGS])
@pytest.mark.parametrize("shape", [(2, 128, 64, 17, 19, 1, 1, 0), (3, 64, 256, 9, 13, 1, 1, 0), (2, 256, 128, 12, 12, 3, 1, 1)])
def test_conv_dgrad_with_fused_residual_join_backward(shape, cfg, dtype):
    """pxl_conv_dgrad_joinreduce == pxl_conv_igemm (data gradient + addend) followed by pxl_residual_bwd_reduce: the stored
    tensor is the ReLU-masked gradient (bit-identical) and the sums are bn3's [sum g, sum g * xhat].  bf16 and fp32."""
    ops = _ops()
    B, Cin, Cout, H, W, k, d, p = shape
    g = torch.Generator().manual_seed(B * 100 + Cin + k + 7)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    taps = ops.fwd_taps(k, k, d, p)
    Ho, Wo = H + 2 * p - d * (k - 1), W + 2 * p - d * (k - 1)
    dy = to_nhwc(qround(torch.randn(B, Cout, Ho, Wo, generator=g), dtype), Cout, dtype)
    _, wt = pack_w(w, dtype, Cin, kp=Cout)
    y = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)            # bn3's input
    out = to_nhwc(qround(torch.relu(torch.randn(B, Cin, H, W, generator=g)), dtype), Cin, dtype)    # the join's output
    add = to_nhwc(qround(torch.randn(B, Cin, H, W, generator=g), dtype), Cin, dtype)
    coef = torch.cat([torch.randn(Cin, generator=g) * 0.1, torch.rand(Cin, generator=g) + 0.5,
                      torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3]).to(DEV)
    bdesc = ops.conv_desc(dtype, B, Ho, Wo, Cout, H, W, Cin, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=1, tile_cfg=cfg)
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    for addend in (None, add):
        full = torch.empty(B, H, W, Cin, device=DEV, dtype=dtype)
        ops.conv_igemm(bdesc, dy, wt, full, addend=addend)
        ref = torch.empty_like(full)
        rs = torch.zeros(2 * Cin, device=DEV)
        check(lib().pxl_residual_bwd_reduce(dtype_code(dtype), B * H * W, Cin, ptr(full), ptr(out), ptr(y), ptr(coef), ptr(ref),
                                            None, ptr(rs), stream_ptr()))
        got = torch.empty_like(full)
        sums = torch.zeros(2 * Cin, device=DEV)
        ops.conv_dgrad_joinreduce(bdesc, dy, wt, got, out, y, coef, sums, addend=addend)
        torch.cuda.synchronize()
        assert torch.equal(got, ref) and (got == 0).float().mean().item() > 0.3
        assert rel_err(sums.cpu(), rs.cpu()) < 1e-4, (shape, cfg, addend is not None)
        if dtype == torch.bfloat16:
            # round 6: the join's ReLU mask as ONE BIT per element, written by the join's forward kernel (pxl_residual_fwd_bits):
            # the same masked gradient bit for bit, the same sums
            res0 = to_nhwc(torch.zeros(B, Cin, H, W), Cin, dtype)
            ident = torch.cat([torch.zeros(2 * Cin), torch.ones(Cin), torch.zeros(Cin)]).to(DEV)      # bn(y) = y
            bits = torch.full((B * H * W, Cin // 8), 0xAA, device=DEV, dtype=torch.uint8)
            out2 = torch.empty_like(out)
            check(lib().pxl_residual_fwd_bits(dtype_code(dtype), B * H * W, Cin, ptr(out), ptr(ident), ptr(res0), None, ptr(out2),
                                              ptr(bits), stream_ptr()))          # relu(out + 0) = out, and its mask
            want_bits = (out.reshape(B * H * W, Cin // 8, 8).float() > 0).to(torch.int32)
            want_bits = (want_bits * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8)
            got_b = torch.empty_like(full)
            sums_b = torch.zeros(2 * Cin, device=DEV)
            check(lib().pxl_conv_dgrad_joinreduce_bits(bdesc, ptr(dy), ptr(wt), ptr(got_b), ptr(addend), ptr(bits), ptr(y), ptr(coef),
                                                       ptr(sums_b), stream_ptr()))
            torch.cuda.synchronize()
            assert torch.equal(out2, out) and torch.equal(bits, want_bits)
            assert torch.equal(got_b, ref)
            assert rel_err(sums_b.cpu(), rs.cpu()) < 1e-4, (shape, cfg, "bits")


WDMA_CASES = [
    # name, B, Cin, Cout, H, W, k, stride, dil, pad   (Cin % 128 == 0: conv_wgrad_dma.hip)
    ("wdma_1x1", 3, 256, 72, 9, 13, 1, 1, 1, 0),
    ("wdma_1x1_s2", 2, 128, 160, 17, 17, 1, 2, 1, 0),
    ("wdma_3x3", 2, 128, 128, 33, 33, 3, 1, 1, 1),
    ("wdma_3x3_s2", 2, 128, 96, 17, 19, 3, 2, 1, 1),
    ("wdma_3x3_d2", 2, 128, 128, 9, 9, 3, 1, 2, 2),
    ("wdma_3x3_tiny", 5, 256, 64, 3, 5, 3, 1, 1, 1),      # Ho*Wo < 64: a reduction step spans several images
    ("wdma_4x4_s2", 1, 128, 256, 33, 33, 4, 2, 1, 1),
    ("wdma_3x3_c64", 2, 64, 96, 17, 17, 3, 1, 1, 1),        # Cin = 64: 64-channel column tiles only
    ("wdma_1x1_c192", 2, 192, 40, 9, 9, 1, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
@pytest.mark.parametrize("case", WDMA_CASES, ids=[c[0] for c in WDMA_CASES])
@pytest.mark.parametrize("cfg", [-1, 8, 9, 10, 11, 12, 13])
def test_conv_wgrad_dma(case, cfg, dtype):
    """LDS-DMA weight gradient (bf16: ds_read_b64_tr_b16 fragments, csrc/conv_wgrad_dma.hip; fp32: ds_read_b32 +
    v_mfma_f32_32x32x2_f32, csrc/conv_wgrad_dma_f32.hip): accumulates on top of a non-zero buffer."""
    ops = _ops()
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name) + 3)
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p, d)
    Ho, Wo = y.shape[2:]
    dy = qround(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    cip, cop = _pitch(Cin), _pitch(Cout)
    fdesc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, ops.fwd_taps(k, k, d, p), out_stride=s, tile_cfg=cfg)
    dw = torch.ones(Cout, k * k, Cin, device=DEV)
    ops.conv_wgrad(fdesc, to_nhwc(x, cip, dtype), to_nhwc(dy, cop, dtype), dw, Cin, Cin)
    torch.cuda.synchronize()
    got = (dw.cpu() - 1.0).view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, w.grad) < 2 * TOL[dtype], "wgrad %s cfg %d" % (name, cfg)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_prologue_epilogue(dtype):
    """relu(bn(x)) fused into the load (zero padding stays zero), bias, addend, BN statistics."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 2, 64, 96, 13, 13
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    scale = torch.rand(Cin, generator=g) + 0.5
    shift = torch.randn(Cin, generator=g) * 0.3
    bias = torch.randn(Cout, generator=g)
    add = torch.randn(B, Cout, H, W, generator=g)
    xq = qround(x, dtype)
    a = qround(F.relu(xq * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), dtype)
    ref = F.conv2d(a, qround(w, dtype), bias, 1, 1, 1) + qround(add, dtype)
    cip, cop = _pitch(Cin), _pitch(Cout)
    out = torch.empty(B, H, W, cop, device=DEV, dtype=dtype)
    nrep = 3
    stats = torch.zeros(nrep, 2 * Cout, device=DEV)
    wf, _ = pack_w(w, dtype, cip)
    desc = ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, ops.fwd_taps(3, 3, 1, 1), relu_in=True, stats_rep=nrep)
    ops.conv_igemm(desc, to_nhwc(x, cip, dtype), wf, out, in_scale=scale.to(DEV), in_shift=shift.to(DEV),
                   bias=bias.to(DEV), addend=to_nhwc(add, cop, dtype), stats=stats)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cout), ref) < TOL[dtype]
    s1 = ref.sum(dim=(0, 2, 3))
    s2 = (ref * ref).sum(dim=(0, 2, 3))
    folded = stats.sum(0).cpu()
    assert rel_err(folded[:Cout], s1) < 5 * TOL[dtype]
    assert rel_err(folded[Cout:], s2) < 5 * TOL[dtype]
    assert (stats.abs().sum(1) > 0).all()          # every replica received some tiles


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES[:-1], ids=[c[0] for c in CONV_CASES[:-1]])
def test_conv_dgrad_and_wgrad(dtype, case):
    ops = _ops()
    name, B, Cin, Cout, H, W, k, s, d, p = case
    g = torch.Generator().manual_seed(_seed(name) + 1)
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype).requires_grad_(True)
    w = qround(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p, d)
    Ho, Wo = y.shape[2:]
    dy = qround(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    cip, cop = _pitch(Cin), _pitch(Cout)
    taps = ops.fwd_taps(k, k, d, p)
    # ---- data gradient: implicit GEMM with transposed weights, negated taps, divisor = stride
    _, wt = pack_w(w.detach(), dtype, cip, kp=cop)
    dx = torch.empty(B, H, W, cip, device=DEV, dtype=dtype)
    bdesc = ops.conv_desc(dtype, B, Ho, Wo, cop, H, W, cip, Cin, [(-a, -b) for a, b in taps], out_stride=1, div=s)
    ops.conv_igemm(bdesc, to_nhwc(dy, cop, dtype), wt, dx)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dx, Cin), x.grad) < TOL[dtype], "dgrad " + name
    # ---- weight gradient (fp32 master layout [K][T][C]), accumulated on top of a non-zero buffer
    fdesc = ops.conv_desc(dtype, B, H, W, cip, Ho, Wo, cop, Cout, taps, out_stride=s)
    dw = torch.ones(Cout, k * k, Cin, device=DEV)
    ops.conv_wgrad(fdesc, to_nhwc(x.detach(), cip, dtype), to_nhwc(dy, cop, dtype), dw, Cin, Cin)
    torch.cuda.synchronize()
    got = (dw.cpu() - 1.0).view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, w.grad) < 2 * TOL[dtype], "wgrad " + name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_wgrad_with_prologue_and_tile_cfgs(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, H, W = 2, 64, 21, 11, 11
    x = qround(torch.randn(B, Cin, H, W, generator=g), dtype)
    scale = torch.rand(Cin, generator=g) + 0.5
    shift = torch.randn(Cin, generator=g) * 0.3
    a = qround(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), dtype)
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    y = F.conv2d(a, w, None, 1, 2, 2)
    dy = qround(torch.randn(y.shape, generator=g), dtype)
    y.backward(dy)
    cip, cop = _pitch(Cin), _pitch(Cout)
    for cfg in (-1, 0, 1, 2):
        desc = ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, ops.fwd_taps(3, 3, 2, 2), relu_in=True, tile_cfg=cfg)
        dw = torch.zeros(Cout, 9, Cin, device=DEV)
        ops.conv_wgrad(desc, to_nhwc(x, cip, dtype), to_nhwc(dy, cop, dtype), dw, Cin, Cin,
                       in_scale=scale.to(DEV), in_shift=shift.to(DEV))
        torch.cuda.synchronize()
        got = dw.cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        assert rel_err(got, w.grad) < 2 * TOL[dtype], "cfg %d" % cfg


@pytest.mark.parametrize("Cin", [64, 256, 512])
def test_aspp_multirate_as_one_launch(Cin):
    """36 taps (4 dilation groups) in one implicit GEMM == sum of four dilated convs (deeplab_v2.py:81-85).  Cin = 256 / 512: several
    K steps per tap, so the bf16 LDS-DMA kernel splits K by CHANNEL slice (round 5: every slice walks all 36 taps over its
    share of the input channels; 2, 4 or 8 slices depending on the request)."""
    ops = _ops()
    dtype = torch.float32
    g = torch.Generator().manual_seed(11)
    B, Cout, H, W = 2, 21, 9, 9
    rates = (1, 2, 3, 4)
    x = torch.randn(B, Cin, H, W, generator=g)
    ws = [torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05 for _ in rates]
    ref = sum(F.conv2d(x, w, None, 1, r, r) for w, r in zip(ws, rates))
    cip, cop = _pitch(Cin), _pitch(Cout)
    wf = torch.empty(Cout, 36, cip, device=DEV)
    taps = []
    for gi, (w, r) in enumerate(zip(ws, rates)):
        ops.pack_weights(dtype, w.permute(0, 2, 3, 1).contiguous().to(DEV), Cout, 9, Cin, wf, cip, T_total=36, t_off=9 * gi)
        taps += ops.fwd_taps(3, 3, r, r)
    out = torch.empty(B, H, W, cop, device=DEV)
    ops.conv_igemm(ops.conv_desc(dtype, B, H, W, cip, H, W, cop, Cout, taps), to_nhwc(x, cip, dtype), wf, out)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, Cout), ref) < 2e-5
    # split-K over the 36 taps (few output tiles, long reduction) with bias, bf16 and fp32
    bias = torch.randn(Cout, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        wf2 = torch.empty(Cout, 36, cip, device=DEV, dtype=dt)
        for gi, w in enumerate(ws):
            ops.pack_weights(dt, w.permute(0, 2, 3, 1).contiguous().to(DEV), Cout, 9, Cin, wf2, cip, T_total=36, t_off=9 * gi)
        ref2 = sum(F.conv2d(qround(x, dt), qround(w, dt), None, 1, r, r) for w, r in zip(ws, rates)) + bias.view(1, -1, 1, 1)
        for sk in (0, 1, 2, 5, 8):
            o2 = torch.full((B, H, W, cop), 3.0, device=DEV, dtype=dt)
            wsb = torch.empty(B * H * W * cop, device=DEV, dtype=torch.float32)
            ops.conv_igemm(ops.conv_desc(dt, B, H, W, cip, H, W, cop, Cout, taps, split_k=sk), to_nhwc(x, cip, dt), wf2, o2,
                           bias=bias.to(DEV), workspace=wsb)
            torch.cuda.synchronize()
            assert rel_err(from_nhwc(o2, Cout), ref2) < TOL[dt], (dt, sk)
            assert o2[..., Cout:].float().abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,B,H,W", [(64, 2, 9, 7), (256, 2, 9, 7), (2048, 2, 9, 7), (96, 3, 33, 17), (1024, 2, 33, 33)])
def test_batchnorm_finalize_and_backward(dtype, C, B, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    y = qround(torch.randn(B, C, H, W, generator=g) * 2 + 0.5, dtype).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    z = F.relu(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5))
    dz = qround(torch.randn(z.shape, generator=g), dtype)
    z.backward(dz)
    M = B * H * W
    yd = to_nhwc(y.detach(), C, dtype)
    stats = torch.stack([y.detach().sum(dim=(0, 2, 3)), (y.detach() ** 2).sum(dim=(0, 2, 3))]).reshape(-1).to(DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    coef = ops.bn_finalize(stats, M, gamma.detach().to(DEV), beta.detach().to(DEV), rmd, rvd)
    torch.cuda.synchronize()
    assert rel_err(rmd.cpu(), rm) < 1e-5 and rel_err(rvd.cpu(), rv) < 1e-5
    mean = y.detach().mean(dim=(0, 2, 3))
    var = y.detach().var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(coef[:C].cpu(), mean) < 1e-4
    assert rel_err(coef[C:2 * C].cpu(), (var + 1e-5).rsqrt()) < 1e-4
    dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy = ops.bn_backward(to_nhwc(dz, C, dtype).view(M, C), yd.view(M, C), coef, M, True, dgamma, dbeta)
    torch.cuda.synchronize()
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_err(from_nhwc(dy.view(B, H, W, C), C), y.grad) < tol
    assert rel_err(dgamma.cpu(), gamma.grad) < tol
    assert rel_err(dbeta.cpu(), beta.grad) < tol
    # the executor's two-launch form (finalize fused into apply), accumulating on top of existing gradients
    dgamma2, dbeta2 = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    dy2 = ops.bn_backward_fused(to_nhwc(dz, C, dtype).view(M, C), yd.view(M, C), coef, M, True, dgamma2, dbeta2)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(dy2.view(B, H, W, C), C), y.grad) < tol
    assert rel_err(dgamma2.cpu() - 1, gamma.grad) < tol and rel_err(dbeta2.cpu() - 1, beta.grad) < tol
    # materialised activation relu(bn(y))
    zz = ops.bn_apply_fwd(yd, coef, relu=True)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(zz, C), z.detach()) < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training,clamp", [(True, False), (True, True), (False, False)])
def test_finalize_folded_into_apply_and_residual(dtype, training, clamp):
    """pxl_bn_finalize_apply_fwd / pxl_residual_finalize_fwd == pxl_bn_finalize followed by pxl_bn_apply_fwd /
    pxl_residual_fwd: same outputs, same coefficient vectors (what backward reads), same running-statistics update,
    for replicated statistics, both variance formulas and eval mode."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    M, C, nrep = 1234, 192, 5
    y = torch.randn(M, C, generator=g).to(DEV).to(dtype)
    res = torch.randn(M, C, generator=g).to(DEV).to(dtype)

    def fresh():
        gg = torch.Generator().manual_seed(7)
        yf = y.float()
        parts = torch.stack([torch.cat([yf[i::nrep].sum(0), (yf[i::nrep] ** 2).sum(0)]) for i in range(nrep)])   # [nrep][2C]
        return dict(stats=parts.contiguous(), gamma=(torch.rand(C, generator=gg) + 0.5).to(DEV),
                    beta=(torch.randn(C, generator=gg) * 0.2).to(DEV), rm=(torch.randn(C, generator=gg) * 0.1).to(DEV),
                    rv=(torch.rand(C, generator=gg) + 0.5).to(DEV))
    a, b = fresh(), fresh()
    coef_ref = ops.bn_finalize(a["stats"], M, a["gamma"], a["beta"], a["rm"], a["rv"], training=training, clamp_var=clamp, nrep=nrep)
    z_ref = ops.bn_apply_fwd(y, coef_ref, relu=True)
    coef = torch.zeros(4 * C, device=DEV)
    fin = ops.bn_fin(b["stats"], nrep, M, b["gamma"], b["beta"], b["rm"], b["rv"], coef, training=training, clamp_var=clamp)
    z = ops.bn_finalize_apply_fwd(y, fin, relu=True)
    torch.cuda.synchronize()
    assert rel_err(coef.cpu(), coef_ref.cpu()) < 1e-6
    assert rel_err(z.float().cpu(), z_ref.float().cpu()) < (1e-6 if dtype == torch.float32 else 4e-3)
    assert rel_err(b["rm"].cpu(), a["rm"].cpu()) < 1e-6 and rel_err(b["rv"].cpu(), a["rv"].cpu()) < 1e-6
    # residual join with both BNs folded in (downsample block) and with the identity shortcut
    for with_r in (True, False):
        a, b, a2, b2 = fresh(), fresh(), fresh(), fresh()
        c1 = ops.bn_finalize(a["stats"], M, a["gamma"], a["beta"], a["rm"], a["rv"], training=training, clamp_var=clamp, nrep=nrep)
        c2 = ops.bn_finalize(a2["stats"], M, a2["gamma"], a2["beta"], a2["rm"], a2["rv"], training=training, clamp_var=clamp,
                             nrep=nrep) if with_r else None
        out_ref = ops.residual_fwd(y, c1, res, c2)
        k1, k2 = torch.zeros(4 * C, device=DEV), torch.zeros(4 * C, device=DEV)
        f1 = ops.bn_fin(b["stats"], nrep, M, b["gamma"], b["beta"], b["rm"], b["rv"], k1, training=training, clamp_var=clamp)
        f2 = ops.bn_fin(b2["stats"], nrep, M, b2["gamma"], b2["beta"], b2["rm"], b2["rv"], k2, training=training,
                        clamp_var=clamp) if with_r else None
        out = ops.residual_finalize_fwd(y, f1, res, f2)
        torch.cuda.synchronize()
        assert rel_err(out.float().cpu(), out_ref.float().cpu()) < (1e-6 if dtype == torch.float32 else 4e-3)
        assert rel_err(k1.cpu(), c1.cpu()) < 1e-6 and (not with_r or rel_err(k2.cpu(), c2.cpu()) < 1e-6)
        assert rel_err(b["rv"].cpu(), a["rv"].cpu()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_residual_join_and_relu_mask(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    B, C, H, W = 2, 256, 9, 9
    y = qround(torch.randn(B, C, H, W, generator=g), dtype)
    r = qround(torch.randn(B, C, H, W, generator=g), dtype)
    yc, rc = torch.randn(4 * C, generator=g), torch.randn(4 * C, generator=g)
    aff = lambda t, c: t * c[2 * C:3 * C].view(1, -1, 1, 1) + c[3 * C:].view(1, -1, 1, 1)
    for rcoef in (None, rc):
        ref = F.relu(aff(y, yc) + (aff(r, rcoef) if rcoef is not None else r))
        out = ops.residual_fwd(to_nhwc(y, C, dtype), yc.to(DEV), to_nhwc(r, C, dtype),
                               rcoef.to(DEV) if rcoef is not None else None)
        torch.cuda.synchronize()
        assert rel_err(from_nhwc(out, C), ref) < TOL[dtype]
    dout = qround(torch.randn(B, C, H, W, generator=g), dtype)
    g1, g2 = ops.relu_mask(to_nhwc(dout, C, dtype), to_nhwc(ref, C, dtype), second=True)
    torch.cuda.synchronize()
    want = dout * (qround(ref, dtype) > 0)
    assert torch.equal(from_nhwc(g1, C), want) and torch.equal(from_nhwc(g2, C), want)
    # the same mask fused with the main-branch BN's backward reduction (pxl_residual_bwd_reduce)
    coef = torch.cat([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5, yc[2 * C:3 * C], yc[3 * C:]])
    for second in (True, False):
        f1, f2, sums = ops.residual_bwd_reduce(to_nhwc(dout, C, dtype), to_nhwc(ref, C, dtype), to_nhwc(y, C, dtype),
                                               coef.to(DEV), second=second)
        torch.cuda.synchronize()
        assert torch.equal(f1, g1) and (f2 is None or torch.equal(f2, g1))
        xhat = (y - coef[:C].view(1, -1, 1, 1)) * coef[C:2 * C].view(1, -1, 1, 1)
        assert rel_err(sums[:C].cpu(), want.sum(dim=(0, 2, 3))) < 1e-4
        assert rel_err(sums[C:].cpu(), (want * xhat).sum(dim=(0, 2, 3))) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_fused_bn_relu(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    B, C, H, W = 2, 64, 17, 17
    y = qround(torch.randn(B, C, H, W, generator=g), dtype)
    coef = torch.randn(4 * C, generator=g)
    z = F.relu(y * coef[2 * C:3 * C].view(1, -1, 1, 1) + coef[3 * C:].view(1, -1, 1, 1)).requires_grad_(True)
    ref = F.max_pool2d(z, 3, 2, 1)
    out, idx = ops.maxpool_fwd(to_nhwc(y, C, dtype), coef.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(out, C), qround(ref.detach(), dtype)) < TOL[dtype]
    dp = qround(torch.randn(ref.shape, generator=g), dtype)
    ref.backward(dp)
    dz = ops.maxpool_bwd(to_nhwc(dp, C, dtype), idx, H, W)
    torch.cuda.synchronize()
    # gradients routed to zero-valued (relu-clamped) positions are killed by the following relu mask
    mask = (z.detach() > 0).float()
    assert rel_err(from_nhwc(dz, C) * mask, z.grad * mask) < (1e-6 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("size", [(5, 65), (3, 33), (40, 65), (24, 24)])
def test_upsample_softmax_forward_backward(dtype, size, align):
    """align=True: the segmentation heads; align=False: F.interpolate's default, used on SSLCCT's auxiliary
    predictions (ssl_cct.py:483); (24, 24) = the identity resize of the I-VAT inner passes."""
    ops = _ops()
    h, H = size
    g = torch.Generator().manual_seed(h)
    B, C, Cp = 2, 21, 32
    low = qround(torch.randn(B, C, h, h, generator=g) * 2, dtype).requires_grad_(True)
    logits = F.interpolate(low, size=(H, H), mode="bilinear", align_corners=align)
    prob = F.softmax(logits, dim=1)
    gl, gp = torch.randn(logits.shape, generator=g), torch.randn(logits.shape, generator=g)
    (logits * gl).sum().backward(retain_graph=True)
    g_only_logits = low.grad.clone()
    low.grad = None
    ((logits * gl).sum() + (prob * gp).sum()).backward()
    lowd = to_nhwc(low.detach(), Cp, dtype)
    lg, pr = ops.upsample_softmax_fwd(lowd, C, H, H, align_corners=align)
    torch.cuda.synchronize()
    assert rel_err(lg.cpu(), logits.detach()) < 1e-5
    assert rel_err(pr.cpu(), prob.detach()) < 1e-5
    tol = 1e-4 if dtype == torch.float32 else 8e-3
    d1 = ops.upsample_softmax_bwd(dtype, gl.to(DEV), None, None, h, h, Cp, align_corners=align)
    d2 = ops.upsample_softmax_bwd(dtype, gl.to(DEV), gp.to(DEV), pr, h, h, Cp, align_corners=align)
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(d1, C), g_only_logits) < tol
    assert rel_err(from_nhwc(d2, C), low.grad) < tol
    assert d1[..., C:].float().abs().max().item() == 0.0


def test_cross_entropy_and_mse_losses():
    import torch_oracle as TO
    from pixelssl_amd import functional as PF
    g = torch.Generator().manual_seed(8)
    N, C, H = 3, 21, 33
    logits = (torch.randn(N, C, H, H, generator=g) * 3).requires_grad_(True)
    gt = torch.randint(0, C, (N, 1, H, H), generator=g).float()
    gt[0, 0, :5] = 255.0
    gt[1, 0, :, ::3] = 255.0
    ref = TO.sseg_criterion(logits, gt)
    wgt = torch.tensor([0.2, 0.5, 0.3])
    (ref * wgt).sum().backward()
    ld = logits.detach().to(DEV).requires_grad_(True)
    got = PF.cross_entropy_per_sample(ld, gt.to(DEV), 255)
    (got * wgt.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(got.detach().cpu(), ref.detach()) < 1e-5
    assert rel_err(ld.grad.cpu(), logits.grad) < 1e-5
    # MSE on an odd-sized (unaligned) batch slice, gradient to the first operand only
    a = torch.randn(4, C, H, H, generator=g).requires_grad_(True)
    b = torch.randn(4, C, H, H, generator=g)
    r = TO.mse_loss(a[1:], b[1:]) * 0.37
    r.backward()
    ad = a.detach().to(DEV).requires_grad_(True)
    m = PF.mse_loss(ad[1:], b.to(DEV)[1:]) * 0.37
    m.backward()
    torch.cuda.synchronize()
    assert abs(m.item() - r.item()) < 1e-5 * abs(r.item())
    assert rel_err(ad.grad.cpu(), a.grad) < 1e-5


@pytest.mark.parametrize("rng", [(2, 5), (0, 5), (3, 3)], ids=["unlabeled-only", "all-samples", "no-consistency"])
def test_fused_task_and_consistency_gradient_is_bit_identical(rng):
    """PF.task_consistency (one backward launch) == CE on pred[:lbs] + MSE on pred[lo:hi] + autograd's slice padding and
    sum: same loss values; d(pred) bit-identical wherever ONE term contributes and within 1 ulp of the larger term where
    both do (measured: 1 ulp on ~10 % of those elements -- each term alone is bit-identical), odd sizes, ignore labels."""
    from pixelssl_amd import functional as PF
    g = torch.Generator().manual_seed(12)
    N, C, H, W, lbs = 5, 21, 33, 29, 2
    lo, hi = rng
    pred = (torch.randn(N, C, H, W, generator=g) * 3).to(DEV)
    target = (torch.randn(N, C, H, W, generator=g) * 3).to(DEV)
    gt = torch.randint(0, C, (lbs, 1, H, W), generator=g).float()
    gt[0, 0, :7] = 255.0
    gt = gt.to(DEV)
    scale = 0.37
    a = pred.clone().requires_grad_(True)
    task = PF.cross_entropy_per_sample(a[:lbs], gt, 255).mean()
    cons = scale * PF.mse_loss(a[lo:hi], target[lo:hi]) if hi > lo else torch.zeros((), device=DEV)
    (task + cons).backward()
    b = pred.clone().requires_grad_(True)
    with torch.no_grad():
        ce_values = PF.cross_entropy_per_sample(b[:lbs], gt, 255)
    ce, mse = PF.task_consistency(b, gt, ce_values, target, lo, hi, 255)
    (ce.mean() + scale * mse).backward()
    torch.cuda.synchronize()
    assert torch.equal(ce.detach(), ce_values) and abs(ce.mean().item() - task.item()) == 0
    assert abs(mse.item() * scale - cons.item()) <= 2e-6 * abs(cons.item())        # (atomic order of the forward sum)
    diff = (a.grad - b.grad).abs()
    per_sample = [(int((diff[n] > 0).sum()), float(diff[n].max()), float(a.grad[n].abs().max())) for n in range(N)]
    print("fused vs separate, per sample (mismatching elements, max |diff|, max |grad|):", per_sample)
    both = [n for n in range(N) if n < lbs and lo <= n < hi]
    for n in range(N):
        if n in both:
            assert float(diff[n].max()) <= 2.0 ** -23 * float(a.grad[n].abs().max()), per_sample
        else:
            assert torch.equal(a.grad[n], b.grad[n]), per_sample
    # only one of the two losses is differentiated
    c = pred.clone().requires_grad_(True)
    ce, mse = PF.task_consistency(c, gt, ce_values, target, lo, hi, 255)
    ce.mean().backward()
    d = pred.clone().requires_grad_(True)
    PF.cross_entropy_per_sample(d[:lbs], gt, 255).mean().backward()
    assert torch.equal(c.grad, d.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("geo", [(2, 3, 33, 29, 7, 2, 3, 192), (1, 3, 65, 65, 7, 2, 3, 192), (2, 4, 17, 19, 3, 1, 1, 64)])
def test_stem_patches_equal_unfold(geo, dtype):
    """pxl_stem_patches == F.unfold re-ordered to (ky, kx, c), zero padded to the row pitch; exact (a copy + one rounding)."""
    from pixelssl_amd._lib import lib, check, ptr, stream_ptr, dtype_code
    B, C, H, W, k, stride, pad, Kp = geo
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    cols = torch.nn.functional.unfold(x, k, padding=pad, stride=stride)                  # [B, C*k*k, L], rows (c, ky, kx)
    ref = cols.reshape(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * C)
    ref = torch.cat([ref, torch.zeros(ref.shape[0], Kp - ref.shape[1])], 1).to(dtype)
    got = torch.full((B * Ho * Wo, Kp), float("nan"), device=DEV, dtype=dtype)
    check(lib().pxl_stem_patches(dtype_code(dtype), ptr(x.to(DEV)), ptr(got), B, C, H, W, k, k, stride, pad, Ho, Wo, Kp, stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref)


def test_sgd_and_ema_flat_updates():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    n = 100003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    buf = torch.zeros(n)
    pd, gd, bd = p.to(DEV), gr.to(DEV), buf.to(DEV)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.SGD([ref_p], lr=0.01, momentum=0.9, weight_decay=5e-4)
    for _ in range(3):
        ref_p.grad = gr.clone()
        opt.step()
        ops.sgd_step(pd, gd, bd, 0.01, 0.9, 5e-4)
    torch.cuda.synchronize()
    assert rel_err(pd.cpu(), ref_p.detach()) < 1e-6
    t, s = torch.randn(n, generator=g), torch.randn(n, generator=g)
    td = t.to(DEV)
    ops.ema_update(td, s.to(DEV), 0.99)
    torch.cuda.synchronize()
    assert rel_err(td.cpu(), t * 0.99 + 0.01 * s) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_pack_equals_per_tensor_pack(dtype):
    """pxl_pack_weights_batched (one launch for many tensors) == pxl_pack_weights per tensor, bit for bit."""
    import ctypes
    from pixelssl_amd import _lib
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    shapes = [(64, 49, 3, 8, 49, 0, 0), (21, 9, 40, 64, 36, 9, 32), (21, 9, 40, 64, 36, 27, 32), (96, 1, 72, 96, 1, 0, 96),
              (33, 16, 24, 32, 16, 0, 64)]                      # K, T, C, Cp, T_total, t_off, Kp (0 = no dgrad operand)
    flat = torch.randn(sum(k * t * c for k, t, c, *_ in shapes) + 64, generator=g).to(DEV)
    es = 4 if dtype == torch.float32 else 2
    items, off, boff, singles = [], 0, 0, []
    for k, t, c, cp, tt, to, kp in shapes:
        it = _lib.PackItem()
        it.src_off, it.K, it.T, it.C, it.Cp, it.T_total, it.t_off, it.Kp = off, k, t, c, cp, tt, to, kp
        it.wf_off = boff
        boff += (k * tt * cp * es + 255) // 256 * 256
        it.wt_off = boff if kp else -1
        boff += (c * tt * kp * es + 255) // 256 * 256 if kp else 0
        items.append(it)
        off += k * t * c
    packed = torch.zeros(boff, device=DEV, dtype=torch.uint8)
    ref = torch.zeros_like(packed)
    arr = (_lib.PackItem * len(items))(*items)
    _lib.check(_lib.lib().pxl_pack_weights_batched(_lib.dtype_code(dtype), flat.data_ptr(), packed.data_ptr(), arr, len(items),
                                                   _lib.stream_ptr()))
    for it in items:
        wf = ref[it.wf_off:].view(dtype)
        wt = ref[it.wt_off:].view(dtype) if it.wt_off >= 0 else None
        ops.pack_weights(dtype, flat[it.src_off:], it.K, it.T, it.C, wf, it.Cp, T_total=it.T_total, t_off=it.t_off, wt=wt,
                         Kp=it.Kp)
    torch.cuda.synchronize()
    assert torch.equal(packed, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, (3,), 8, 37, 41), (2, (3, 21), 32, 65, 65), (4, (512,), 512, 33, 33), (1, (21,), 32, 129, 129),
                                   (2, (5, 1, 7, 2), 32, 9, 11)])
def test_layout_kernels_chunk_per_thread(shape, dtype):
    """NCHW (one tensor, or up to four concatenated along C on load) <-> NHWC with a channel pitch, against torch's permute: the
    image (3 -> 8), GCT's flaw-detector input (3 + 21 -> 32), CCT's 512-channel latent, a head (21 -> 32), four ragged parts.
    Bit-exact in both directions (bf16: round-to-nearest-even like torch's cast; padded channels are zero)."""
    import ctypes
    from pixelssl_amd import _lib
    h = _lib.lib()
    B, chans, Cp, H, W = shape
    g = torch.Generator().manual_seed(sum(chans) + H)
    parts = [torch.randn(B, c, H, W, generator=g).to(DEV) for c in chans]
    C = sum(chans)
    y = torch.full((B, H, W, Cp), 7.0, device=DEV, dtype=dtype)
    srcs = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    cs = (ctypes.c_int * len(parts))(*chans)
    code = _lib.dtype_code(dtype)
    _lib.check(h.pxl_nchw_parts_to_nhwc(code, len(parts), srcs, cs, y.data_ptr(), B, H, W, Cp, _lib.stream_ptr()))
    want = torch.zeros(B, H, W, Cp, device=DEV, dtype=dtype)
    want[..., :C] = torch.cat(parts, 1).permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(y, want)
    if len(parts) == 1:
        assert torch.equal(_ops().nchw_to_nhwc(dtype, parts[0], Cp), want)
    # and back: one NCHW fp32 tensor per part (the second part, when there is one, is not wanted -> untouched)
    outs = [torch.full((B, c, H, W), -3.0, device=DEV) for c in chans]
    dsts = (ctypes.c_void_p * len(parts))(*[None if (k == 1 and len(parts) > 1) else o.data_ptr() for k, o in enumerate(outs)])
    _lib.check(h.pxl_nhwc_to_nchw_parts(code, y.data_ptr(), len(parts), dsts, cs, B, H, W, Cp, _lib.stream_ptr()))
    torch.cuda.synchronize()
    for k, (o, p) in enumerate(zip(outs, parts)):
        if k == 1 and len(parts) > 1:
            assert bool((o == -3.0).all())
        else:
            assert torch.equal(o, p.to(dtype).float())
    if len(parts) == 1:
        assert torch.equal(_ops().nhwc_to_nchw(y, C), parts[0].to(dtype).float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,Cp", [(8 * 256 * 256, 64, 64), (4 * 264 * 264, 84, 96), (8 * 33 * 33, 21, 32), (5, 512, 512),
                                    (2 * 17 * 17, 1, 8), (301, 2048, 2048)])
def test_colsum_chunked(M, C, Cp, dtype):
    """Bias gradients: out[c] += sum_m x[m][c] over 16-byte chunks, row lanes folded through LDS, one atomic per channel and block
    -- the flaw detector's 64-channel rows, a decoder's 84 of 96, a head's 21 of 32, a pooled map of 5 rows, one channel, 2048."""
    import ctypes
    from pixelssl_amd import _lib
    g = torch.Generator().manual_seed(M % 997 + C)
    x = torch.randn(M, Cp, generator=g).to(DEV).to(dtype)
    out = torch.full((Cp,), 0.5, device=DEV)
    _lib.check(_lib.lib().pxl_colsum(_lib.dtype_code(dtype), M, Cp, C, x.data_ptr(), out.data_ptr(), _lib.stream_ptr()))
    want = x.double().sum(0).cpu()
    got = out.double().cpu()
    scale = x.double().abs().sum(0).cpu().clamp_min(1.0)
    assert float(((got[:C] - 0.5 - want[:C]).abs() / scale[:C]).max()) < 2e-6       # fp32 accumulation of M terms, accumulated onto `out`
    assert bool((got[C:] == 0.5).all())                                             # padded channels are not touched
