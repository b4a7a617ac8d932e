"""Pin oracle/gct_oracle.py against the REAL reference modules (container only) and write
tests/golden/gct_flawmap_<size>.pt.  TEST INFRASTRUCTURE.

    python oracle/make_golden_gct.py [size ...]        (default 65; 513 = the BASELINE crop size: blur kernels 65 / 129 / 33,
                                                         dense CPU convolutions -- minutes, once)
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim              # noqa: E402
import gct_oracle as GO      # noqa: E402

GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main(size=65):
    OUT = os.path.join(GOLD, "gct_flawmap_%d.pt" % size)
    ref_shim.load_reference()
    from pixelssl.ssl_algorithm import ssl_gct as R
    from pixelssl.nn.module import GaussianBlurLayer
    C, seed = 21, 11
    args = argparse.Namespace(im_size=size, mu=0.5, nu=1, dc_threshold=0.6)
    l_pred, r_pred, gt, l_fm, r_fm = GO.synthetic_case(seed, C=C, size=size)
    # kernel: rank-1 check (the device path evaluates the blur separably)
    for div in (16, 8, 4):
        k = GO.odd_ksize(size, div)
        K, a = GO.gaussian_kernel2d(k), GO.gaussian_taps1d(k)
        assert np.abs(K - np.outer(a, a)).max() < 1e-15, k
        ref_layer = GaussianBlurLayer(1, k)
        assert torch.equal(ref_layer.op[1].weight.detach().view(k, k), torch.from_numpy(K).float())
    # FDGT
    onehot = GO.onehot_ignore(gt, C)
    ref_fdgt = R.FDGTGenerator(args)(l_pred.clone(), onehot.clone())
    mine = GO.fdgt(l_pred, onehot, size, args.mu, args.nu)
    # (the last sample is unlabeled: its map is the constant mu, whose min-max normalisation is rounding noise / 1e-9 in the
    # reference itself -- compared on the labeled samples)
    assert torch.allclose(mine[:-1], ref_fdgt[:-1], rtol=0, atol=2e-6 if size <= 65 else 2e-5), (mine[:-1] - ref_fdgt[:-1]).abs().max()
    # FlawmapHandler (mutates its argument)
    l_in, r_in = l_fm.clone(), r_fm.clone()
    handler = R.FlawmapHandler(args)
    ref_lh, ref_rh = handler(l_in), handler(r_in)
    my_lh, my_lc = GO.flawmap_handle(l_fm, size)
    my_rh, my_rc = GO.flawmap_handle(r_fm, size)
    assert torch.allclose(my_lh, ref_lh, atol=2e-6 if size <= 65 else 2e-5) and torch.allclose(my_rh, ref_rh, atol=2e-6 if size <= 65 else 2e-5)
    assert torch.equal(my_lc, l_in) and torch.equal(my_rc, r_in)       # the in-place clamp of the argument
    # DCGT (mutates the handled maps)
    lh2, rh2 = ref_lh.clone(), ref_rh.clone()
    ref_lgt, ref_rgt, ref_bad, _ = R.DCGTGenerator(args)(l_pred, r_pred, lh2, rh2)
    m = GO.dcgt(l_pred, r_pred, ref_lh, ref_rh, args.dc_threshold)
    assert torch.equal(m[0], ref_lgt) and torch.equal(m[1], ref_rgt) and torch.equal(m[2], ref_bad)
    assert torch.equal(m[3], lh2) and torch.equal(m[4], rh2)
    # FD criterion
    ref_loss = R.FlawDetectorCriterion()(l_fm, ref_fdgt)
    assert torch.allclose(GO.fd_criterion(l_fm, ref_fdgt), ref_loss, rtol=1e-6)
    # 513: every map is stored on a stride-4 grid (1/16 of the pixels, 1.5 MB instead of 23) next to its full-tensor sum
    st = 1 if size <= 65 else 4
    sub = lambda t: t[..., ::st, ::st].clone()
    torch.save(dict(seed=seed, size=size, C=C, mu=args.mu, nu=args.nu, dc_threshold=args.dc_threshold, stride=st,
                    fdgt=sub(ref_fdgt), l_handled=sub(ref_lh), r_handled=sub(ref_rh), l_clamped=sub(l_in), r_clamped=sub(r_in),
                    l_dc_gt_sum=ref_lgt.double().sum().item(), r_dc_gt_sum=ref_rgt.double().sum().item(),
                    l_dc_gt_head=ref_lgt[:, :, :4, :8].clone(), both_bad=sub(ref_bad.to(torch.uint8)),
                    l_fm_after=sub(lh2), r_fm_after=sub(rh2), fd_loss=ref_loss,
                    sums=dict(fdgt=ref_fdgt[:-1].double().sum().item(), l_handled=ref_lh.double().sum().item(),
                              r_handled=ref_rh.double().sum().item(), both_bad=int(ref_bad.sum().item()))), OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; oracle == reference")


if __name__ == "__main__":
    for sz in ([int(a) for a in sys.argv[1:]] or [65]):
        main(sz)
