"""Tensor-level wrappers over the C-ABI (device pointers + current HIP stream).

Used by the fused-loss autograd functions, the optimizer/EMA, and the per-kernel parity tests.
Activations here are NHWC tensors [B,H,W,Cp] in the engine dtype (fp32 or bf16).
"""
import torch

from . import _lib
from ._lib import ConvDesc, check, lib, ptr, stream_ptr, dtype_code


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PixelHipError("libpixelhip operands must live on the GPU (got a %s tensor)" % t.device)


def conv_desc(dtype, B, Hi, Wi, Cin_p, Ho, Wo, Cout_p, Kreal, taps, out_stride=1, div=1, relu_in=False,
              tile_cfg=-1, stats_rep=1, split_k=0):
    d = ConvDesc()
    d.dtype = dtype_code(dtype)
    d.B, d.Hi, d.Wi, d.Cin = B, Hi, Wi, Cin_p
    d.Ho, d.Wo, d.Cout, d.Kreal = Ho, Wo, Cout_p, Kreal
    d.ntaps = len(taps)
    d.out_stride, d.div, d.relu_in, d.tile_cfg = out_stride, div, int(relu_in), tile_cfg
    d.stats_rep, d.split_k = stats_rep, split_k
    for i, (dy, dx) in enumerate(taps):
        d.dy[i], d.dx[i] = dy, dx
    return d


def fwd_taps(kh, kw, dil, pad):
    return [(r * dil - pad, s * dil - pad) for r in range(kh) for s in range(kw)]


def conv_igemm(desc, x, w, out, in_scale=None, in_shift=None, bias=None, addend=None, stats=None, workspace=None):
    _require_cuda(x, w, out)
    check(lib().pxl_conv_igemm(desc, ptr(x), ptr(w), ptr(out), ptr(in_scale), ptr(in_shift), ptr(bias),
                               ptr(addend), ptr(stats), ptr(workspace),
                               workspace.numel() * workspace.element_size() if workspace is not None else 0,
                               stream_ptr()))
    return out


def conv_dma_finalize(desc, x, w, out, stats, fin, counter, bias=None):
    """forward conv (LDS-DMA kernel) whose last workgroup finalizes the BN of its output; raises when not eligible"""
    _require_cuda(x, w, out, stats, counter)
    check(lib().pxl_conv_dma_finalize(desc, ptr(x), ptr(w), ptr(out), ptr(bias), ptr(stats), fin, ptr(counter), stream_ptr()))
    return out


def conv_dma_bnin(desc, y, w, out, bin_fin, relu=True, stats=None, bias=None, z=None):
    """forward conv reading the RAW previous output y with relu?(bn(y)) applied on load (z: also written out, 1x1 convs);
    raises when not eligible"""
    _require_cuda(y, w, out)
    check(lib().pxl_conv_dma_bnin(desc, ptr(y), ptr(w), ptr(out), ptr(bias), ptr(stats), bin_fin, int(relu), ptr(z),
                                  stream_ptr()))
    return out


def conv_dgrad_bnreduce(desc, dy, wt, din, bn_y, bn_coef, bn_relu, bn_sums, addend=None):
    """data gradient + fused BN-backward reduction of its output (LDS-DMA kernel); raises when not eligible"""
    _require_cuda(dy, wt, din, bn_y, bn_coef, bn_sums)
    check(lib().pxl_conv_dgrad_bnreduce(desc, ptr(dy), ptr(wt), ptr(din), ptr(addend), ptr(bn_y), ptr(bn_coef), int(bn_relu),
                                        ptr(bn_sums), stream_ptr()))
    return din


def conv_dgrad_joinreduce(desc, dy, wt, din, join_out, bn_y, bn_coef, bn_sums, addend=None):
    """data gradient completing d(residual join output) + the join's backward (ReLU mask, bn3 sums) in its epilogue"""
    _require_cuda(dy, wt, din, join_out, bn_y, bn_coef, bn_sums)
    check(lib().pxl_conv_dgrad_joinreduce(desc, ptr(dy), ptr(wt), ptr(din), ptr(addend), ptr(join_out), ptr(bn_y),
                                          ptr(bn_coef), ptr(bn_sums), stream_ptr()))
    return din


def conv_wgrad(desc, x, dy, dw, creal, dw_cpitch, in_scale=None, in_shift=None):
    _require_cuda(x, dy, dw)
    check(lib().pxl_conv_wgrad(desc, ptr(x), ptr(in_scale), ptr(in_shift), ptr(dy), ptr(dw), creal, dw_cpitch,
                               stream_ptr()))
    return dw


def pack_weights(dtype, w, K, T, C, wf, Cp, T_total=None, t_off=0, wt=None, Kp=0):
    _require_cuda(w, wf)
    check(lib().pxl_pack_weights(dtype_code(dtype), ptr(w), K, T, C, ptr(wf), Cp, T_total or T, t_off, ptr(wt), Kp,
                                 stream_ptr()))


def nchw_to_nhwc(dtype, x, Cp):
    _require_cuda(x)
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, Cp, device=x.device, dtype=_lib.torch_dtype(dtype_code(dtype)))
    check(lib().pxl_nchw_to_nhwc(dtype_code(dtype), ptr(x), ptr(y), B, Cc, H, W, Cp, stream_ptr()))
    return y


def nhwc_to_nchw(x, C):
    _require_cuda(x)
    B, H, W, Cp = x.shape
    y = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    check(lib().pxl_nhwc_to_nchw(dtype_code(x.dtype), ptr(x), ptr(y), B, C, H, W, Cp, stream_ptr()))
    return y


def bn_finalize(stats, count, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5, training=True, clamp_var=False,
                nrep=1):
    Cc = gamma.numel()
    coef = torch.empty(4 * Cc, device=gamma.device, dtype=torch.float32)
    check(lib().pxl_bn_finalize(Cc, ptr(stats), nrep, float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
                                momentum, eps, int(training), int(clamp_var), ptr(coef), stream_ptr()))
    return coef


def bn_backward(dz, y, coef, count, relu, dgamma, dbeta, nrep=4):
    """dz,y: [M,C] engine dtype.  Returns dy (new tensor); accumulates dgamma/dbeta."""
    M, Cc = dz.shape[0], dz.shape[1]
    code = dtype_code(dz.dtype)
    sums = torch.zeros(nrep * 2 * Cc, device=dz.device, dtype=torch.float32)
    bcoef = torch.empty(2 * Cc, device=dz.device, dtype=torch.float32)
    dy = torch.empty_like(dz)
    check(lib().pxl_bn_bwd_reduce(code, M, Cc, ptr(dz), ptr(y), ptr(coef), int(relu), ptr(sums), nrep, stream_ptr()))
    check(lib().pxl_bn_bwd_finalize(Cc, ptr(sums), nrep, float(count), ptr(dgamma), ptr(dbeta), ptr(bcoef), 1,
                                    stream_ptr()))
    check(lib().pxl_bn_bwd_apply(code, M, Cc, ptr(dz), ptr(y), ptr(coef), ptr(bcoef), int(relu), ptr(dy),
                                 stream_ptr()))
    return dy


def bn_backward_fused(dz, y, coef, count, relu, dgamma, dbeta, training=True):
    """Reduce + (finalize fused into) apply: the two-launch backward the executor uses."""
    M, Cc = dz.shape[0], dz.shape[1]
    code = dtype_code(dz.dtype)
    sums = torch.zeros(2 * Cc, device=dz.device, dtype=torch.float32)
    dy = torch.empty_like(dz)
    check(lib().pxl_bn_bwd_reduce(code, M, Cc, ptr(dz), ptr(y), ptr(coef), int(relu), ptr(sums), 1, stream_ptr()))
    check(lib().pxl_bn_bwd_apply_fused(code, M, Cc, ptr(dz), ptr(y), ptr(coef), ptr(sums), float(count), int(training),
                                       int(relu), ptr(dgamma), ptr(dbeta), ptr(dy), stream_ptr()))
    return dy


def bn_apply_fwd(y, coef, relu=True):
    z = torch.empty_like(y)
    M = y.numel() // y.shape[-1]
    check(lib().pxl_bn_apply_fwd(dtype_code(y.dtype), M, y.shape[-1], ptr(y), ptr(coef), int(relu), ptr(z), stream_ptr()))
    return z


def bn_fin(stats, nrep, count, gamma, beta, rmean, rvar, coef, momentum=0.1, eps=1e-5, training=True, clamp_var=False):
    """pxl_bn_fin descriptor over device tensors (kept alive by the caller)"""
    f = _lib.BnFin()
    f.stats, f.nrep, f.count = ptr(stats), nrep, float(count)
    f.gamma, f.beta, f.running_mean, f.running_var = ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar)
    f.momentum, f.eps, f.training, f.clamp_var, f.coef = momentum, eps, int(training), int(clamp_var), ptr(coef)
    return f


def bn_finalize_apply_fwd(y, fin, relu=True):
    z = torch.empty_like(y)
    M = y.numel() // y.shape[-1]
    check(lib().pxl_bn_finalize_apply_fwd(dtype_code(y.dtype), M, y.shape[-1], ptr(y), fin, int(relu), ptr(z), stream_ptr()))
    return z


def residual_finalize_fwd(y, yfin, res, rfin=None):
    out = torch.empty_like(y)
    M = y.numel() // y.shape[-1]
    check(lib().pxl_residual_finalize_fwd(dtype_code(y.dtype), M, y.shape[-1], ptr(y), yfin, ptr(res), rfin, ptr(out),
                                          stream_ptr()))
    return out


def residual_fwd(y, ycoef, res, rcoef=None):
    out = torch.empty_like(y)
    M = y.numel() // y.shape[-1]
    check(lib().pxl_residual_fwd(dtype_code(y.dtype), M, y.shape[-1], ptr(y), ptr(ycoef), ptr(res), ptr(rcoef),
                                 ptr(out), stream_ptr()))
    return out


def residual_bwd_reduce(dout, out, y, coef, second=True):
    """relu mask of the join + bn3's backward sums in one pass -> (g, g2 | None, sums[2C])"""
    M, Cc = dout.numel() // dout.shape[-1], dout.shape[-1]
    g = torch.empty_like(dout)
    g2 = torch.empty_like(dout) if second else None
    sums = torch.zeros(2 * Cc, device=dout.device, dtype=torch.float32)
    check(lib().pxl_residual_bwd_reduce(dtype_code(dout.dtype), M, Cc, ptr(dout), ptr(out), ptr(y), ptr(coef), ptr(g), ptr(g2),
                                        ptr(sums), stream_ptr()))
    return g, g2, sums


def relu_mask(dout, out, second=False):
    g = torch.empty_like(dout)
    g2 = torch.empty_like(dout) if second else None
    check(lib().pxl_relu_mask(dtype_code(dout.dtype), dout.numel(), ptr(dout), ptr(out), ptr(g), ptr(g2), stream_ptr()))
    return (g, g2) if second else g


def maxpool_fwd(y, coef=None):
    B, Hi, Wi, Cc = y.shape
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    out = torch.empty(B, Ho, Wo, Cc, device=y.device, dtype=y.dtype)
    idx = torch.empty(B, Ho, Wo, Cc, device=y.device, dtype=torch.uint8)
    check(lib().pxl_maxpool3x3s2_fwd(dtype_code(y.dtype), B, Hi, Wi, Cc, ptr(y), ptr(coef), ptr(out), ptr(idx),
                                     stream_ptr()))
    return out, idx


def maxpool_bwd(dp, idx, Hi, Wi):
    B, _, _, Cc = dp.shape
    dz = torch.empty(B, Hi, Wi, Cc, device=dp.device, dtype=dp.dtype)
    check(lib().pxl_maxpool3x3s2_bwd(dtype_code(dp.dtype), B, Hi, Wi, Cc, ptr(dp), ptr(idx), ptr(dz), stream_ptr()))
    return dz


def adaptive_avgpool_fwd(x, bins):
    """x NHWC [B,H,W,Cp] -> [B,bins,bins,Cp] (nn.AdaptiveAvgPool2d windows)"""
    _require_cuda(x)
    B, H, W, Cp = x.shape
    out = torch.empty(B, bins, bins, Cp, device=x.device, dtype=x.dtype)
    check(lib().pxl_adaptive_avgpool_fwd(dtype_code(x.dtype), B, H, W, Cp, bins, ptr(x), ptr(out), stream_ptr()))
    return out


def adaptive_avgpool_bwd(dout, H, W, din=None):
    """gradient of adaptive_avgpool_fwd; `din` given = accumulate into it"""
    _require_cuda(dout, din)
    B, bins, _, Cp = dout.shape
    acc = din is not None
    if din is None:
        din = torch.empty(B, H, W, Cp, device=dout.device, dtype=dout.dtype)
    check(lib().pxl_adaptive_avgpool_bwd(dtype_code(dout.dtype), B, H, W, Cp, bins, ptr(dout), ptr(din), int(acc),
                                         stream_ptr()))
    return din


def upsample_slice_fwd(x, C, out, c_off, coef=None, relu=False):
    """bilinear (align_corners=False) of relu?(x*scale+shift) [B,h,w,Cp] into channels [c_off, c_off+C) of `out`"""
    _require_cuda(x, out, coef)
    B, h, w, Cpi = x.shape
    _, H, W, Cpo = out.shape
    check(lib().pxl_upsample_slice_fwd(dtype_code(x.dtype), B, h, w, Cpi, C, ptr(x), ptr(coef), int(relu), H, W, ptr(out),
                                       Cpo, c_off, stream_ptr()))
    return out


def upsample_slice_bwd(dout, c_off, C, h, w, Cp_in):
    _require_cuda(dout)
    B, H, W, Cpo = dout.shape
    din = torch.empty(B, h, w, Cp_in, device=dout.device, dtype=dout.dtype)
    check(lib().pxl_upsample_slice_bwd(dtype_code(dout.dtype), B, h, w, Cp_in, C, ptr(dout), H, W, Cpo, c_off, ptr(din),
                                       stream_ptr()))
    return din


def slice_copy(src, s_off, C, dst, d_off, accumulate=False):
    _require_cuda(src, dst)
    M = src.numel() // src.shape[-1]
    check(lib().pxl_slice_copy(dtype_code(src.dtype), M, C, ptr(src), src.shape[-1], s_off, ptr(dst), dst.shape[-1], d_off,
                               int(accumulate), stream_ptr()))
    return dst


def pixshuf_relu_fwd(x, C, Cp_out):
    """x NHWC [B,h,w,Cp_in] holding 4*C channels -> PixelShuffle(2)(relu(x)) [B,2h,2w,Cp_out]"""
    _require_cuda(x)
    B, h, w, Cpi = x.shape
    out = torch.empty(B, 2 * h, 2 * w, Cp_out, device=x.device, dtype=x.dtype)
    check(lib().pxl_pixshuf_relu_fwd(dtype_code(x.dtype), B, h, w, Cpi, C, ptr(x), ptr(out), Cp_out, stream_ptr()))
    return out


def pixshuf_relu_bwd(dout, x, C):
    _require_cuda(dout, x)
    B, h, w, Cpi = x.shape
    din = torch.empty_like(x)
    check(lib().pxl_pixshuf_relu_bwd(dtype_code(x.dtype), B, h, w, Cpi, C, ptr(dout), dout.shape[-1], ptr(x), ptr(din),
                                     stream_ptr()))
    return din


def upsample_softmax_fwd(low, C, H, W, want_prob=True, align_corners=True):
    B, h, w, Cp = low.shape
    logits = torch.empty(B, C, H, W, device=low.device, dtype=torch.float32)
    prob = torch.empty_like(logits) if want_prob else None
    check(lib().pxl_upsample_softmax_fwd(dtype_code(low.dtype), B, h, w, Cp, C, H, W, int(align_corners), ptr(low), ptr(logits),
                                         ptr(prob), stream_ptr()))
    return logits, prob


def upsample_softmax_bwd(dtype, dlogits, dprob, prob, h, w, Cp, align_corners=True):
    ref = dlogits if dlogits is not None else dprob
    B, Cc, H, W = ref.shape
    dlow = torch.empty(B, h, w, Cp, device=ref.device, dtype=_lib.torch_dtype(dtype_code(dtype)))
    nbytes = lib().pxl_upsample_bwd_workspace(B, w, Cc, H)
    ws = torch.empty(nbytes, device=ref.device, dtype=torch.uint8)
    check(lib().pxl_upsample_softmax_bwd(dtype_code(dtype), B, h, w, Cp, Cc, H, W, int(align_corners), ptr(dlogits), ptr(dprob), ptr(prob),
                                         ptr(dlow), ptr(ws), nbytes, stream_ptr()))
    return dlow


def sgd_step(p, g, buf, lr, momentum, weight_decay, lr_dev=None):
    """lr_dev: device address of the learning rate (graph.HyperBlock.ptr) -- a captured step reads it from there"""
    if lr_dev is not None:
        check(lib().pxl_sgd_step_hp(p.numel(), ptr(p), ptr(g), ptr(buf), lr_dev, momentum, weight_decay, 0, stream_ptr()))
    else:
        check(lib().pxl_sgd_step(p.numel(), ptr(p), ptr(g), ptr(buf), lr, momentum, weight_decay, 0, stream_ptr()))


def ema_update(teacher, student, alpha, alpha_dev=None):
    if alpha_dev is not None:
        check(lib().pxl_ema_update_hp(teacher.numel(), ptr(teacher), ptr(student), alpha_dev, stream_ptr()))
    else:
        check(lib().pxl_ema_update(teacher.numel(), ptr(teacher), ptr(student), alpha, stream_ptr()))
