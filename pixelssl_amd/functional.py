"""Fused per-pixel losses (autograd Functions over libpixelhip kernels, NCHW fp32 tensors)."""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.PixelHipError("fused losses run on the GPU only (got %s); there is no CPU path" % t.device)


class _CrossEntropyPerSample(torch.autograd.Function):
    """CommonSSEGCriterion arithmetic (task/sseg/criterion.py:24-38): CE with ignore_index summed over a
    sample and divided by ALL H*W pixels."""

    @staticmethod
    def forward(ctx, logits, gt, ignore_index):
        _gpu(logits, gt)
        logits = logits.contiguous()
        gt = gt.contiguous().float()
        N, C, H, W = logits.shape
        loss = torch.empty(N, device=logits.device, dtype=torch.float32)
        check(lib().pxl_ce_fwd(N, C, H * W, ptr(logits), ptr(gt), int(ignore_index), ptr(loss), stream_ptr()))
        ctx.save_for_backward(logits, gt)
        ctx.ignore_index = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, gout):
        logits, gt = ctx.saved_tensors
        N, C, H, W = logits.shape
        gout = gout.contiguous().float()
        dlogits = torch.empty_like(logits)
        check(lib().pxl_ce_bwd(N, C, H * W, ptr(logits), ptr(gt), ctx.ignore_index, ptr(gout), ptr(dlogits), stream_ptr()))
        return dlogits, None, None


def cross_entropy_per_sample(logits, gt, ignore_index=255):
    """logits [N,C,H,W] fp32, gt [N,1,H,W] or [N,H,W] float class ids -> loss [N]."""
    return _CrossEntropyPerSample.apply(logits, gt, ignore_index)


class _MSE(torch.autograd.Function):
    """nn.MSELoss() (mean reduction); the gradient flows to the first operand only (the second is the
    detached teacher / pseudo-label, ssl_mt.py:179-184)."""

    @staticmethod
    def forward(ctx, a, b):
        _gpu(a, b)
        a = a.contiguous()
        b = b.contiguous()
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        check(lib().pxl_mse_fwd(a.numel(), ptr(a), ptr(b), ptr(out), stream_ptr()))
        ctx.save_for_backward(a, b)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        gout = gout.contiguous().float().view(1)
        da = torch.empty_like(a)
        check(lib().pxl_mse_bwd(a.numel(), ptr(a), ptr(b), ptr(gout), ptr(da), stream_ptr()))
        return da, None


def mse_loss(a, b):
    if a.shape != b.shape:
        raise ValueError("mse_loss: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
    return _MSE.apply(a, b.detach())


class _TaskConsistency(torch.autograd.Function):
    """Task loss (per-sample CE on the first `n_ce` samples) + consistency loss (MSE of samples [lo, hi) against a
    detached target) of ONE logits tensor, with ONE backward launch writing d(logits) once: what autograd builds from
    `pred[:lbs]` -> CE, `pred[lo:]` -> MSE, `task + cons` is two zero-filled slice gradients, two copies and a sum over the
    full [N, C, H, W] tensor (ssl_mt.py:166-196 at 8 x 21 x 513 x 513: 0.33 ms of the step).  `ce_values` is the CE forward
    already computed (no-grad) on the same operands -- the caller may have launched it before the target existed."""

    @staticmethod
    def forward(ctx, logits, gt, ce_values, target, lo, hi, ignore_index):
        _gpu(logits, gt, target)
        logits = logits.contiguous()
        target = target.contiguous()
        gt = gt.contiguous().float()
        if target.shape != logits.shape:
            raise ValueError("task_consistency: target %s vs logits %s" % (tuple(target.shape), tuple(logits.shape)))
        N = logits.shape[0]
        n_ce = ce_values.shape[0]
        if not (0 <= lo <= hi <= N and n_ce <= N and gt.shape[0] == n_ce):
            raise ValueError("task_consistency: bad ranges n_ce=%d [%d, %d) of %d" % (n_ce, lo, hi, N))
        out = torch.zeros(1, device=logits.device, dtype=torch.float32)
        if hi > lo:
            a, b = logits[lo:hi], target[lo:hi]
            check(lib().pxl_mse_fwd(a.numel(), ptr(a), ptr(b), ptr(out), stream_ptr()))
        ctx.save_for_backward(logits, gt, target)
        ctx.rng = (int(lo), int(hi), int(n_ce), int(ignore_index))
        ctx.set_materialize_grads(False)
        return ce_values.clone(), out.view(())

    @staticmethod
    def backward(ctx, g_ce, g_mse):
        logits, gt, target = ctx.saved_tensors
        lo, hi, n_ce, ignore = ctx.rng
        if g_ce is None and g_mse is None:
            return (None,) * 7
        N, C, H, W = logits.shape
        g_ce = g_ce.contiguous().float() if g_ce is not None else None
        g_mse = g_mse.contiguous().float().view(1) if g_mse is not None else None
        dlogits = torch.empty_like(logits)
        check(lib().pxl_ce_mse_bwd(N, C, H * W, ptr(logits), ptr(gt), ignore, n_ce, ptr(g_ce), ptr(target), lo, hi, ptr(g_mse),
                                   ptr(dlogits), stream_ptr()))
        return dlogits, None, None, None, None, None, None


def task_consistency(logits, gt, ce_values, target, lo, hi, ignore_index=255):
    """-> (per-sample CE [n_ce] (= ce_values, now differentiable), MSE(logits[lo:hi], target[lo:hi]) scalar)."""
    return _TaskConsistency.apply(logits, gt, ce_values.detach(), target.detach(), int(lo), int(hi), int(ignore_index))


def head_losses(s_head, t_head, gt, n_ce, mse_lo, mse_hi, ce_weight, mse_weight, ignore_index=255, mse_weight_dev=None):
    """The training seam on the low-resolution logits of deferred forward passes (engine.DeferredHead; csrc/head.hip:
    pxl_head_loss): per-sample cross-entropy of the student and of the teacher on the first `n_ce` samples
    (task/sseg/criterion.py:24-38), the MSE between their predictions over samples [mse_lo, mse_hi) (ssl_mt.py:179-184)
    and d(loss)/d(student low-res logits) for loss = ce_weight * sum(CE_student) + mse_weight * MSE, left in the
    student's executor for s_head.backward().  t_head None: student terms only.
    mse_weight_dev: device address of mse_weight (graph.HyperBlock.ptr) -- a captured step reads the ramped weight from there.
    -> (student CE [n_ce], teacher CE [n_ce] or None, MSE mean scalar), detached fp32 device tensors."""
    _gpu(gt)
    gt = gt.contiguous().float()
    B = s_head.batch
    if t_head is not None and (t_head.batch != B or t_head.plan.out_size != s_head.plan.out_size):
        raise ValueError("head_losses: student and teacher passes differ in shape")
    H, W = s_head.plan.out_size
    if n_ce and gt.numel() != n_ce * H * W:
        raise ValueError("head_losses: gt %s does not hold %d maps of %d x %d" % (tuple(gt.shape), n_ce, H, W))
    sums = torch.empty(2 * B + 1, device=gt.device, dtype=torch.float32)
    pl = s_head.plan
    s_head._check_arena("head_losses")
    if t_head is not None:
        t_head._check_arena("head_losses")
    if mse_weight_dev is not None:
        check(lib().pxl_net_head_loss_hp(pl.net, ptr(s_head.arena), t_head.plan.net if t_head is not None else None,
                                         ptr(t_head.arena) if t_head is not None else None, ptr(gt), int(ignore_index), int(n_ce),
                                         int(mse_lo), int(mse_hi), float(ce_weight), mse_weight_dev, ptr(pl.scratch),
                                         pl.scratch.numel(), ptr(sums), stream_ptr()))
    else:
        check(lib().pxl_net_head_loss(pl.net, ptr(s_head.arena), t_head.plan.net if t_head is not None else None,
                                      ptr(t_head.arena) if t_head is not None else None, ptr(gt), int(ignore_index), int(n_ce),
                                      int(mse_lo), int(mse_hi), float(ce_weight), float(mse_weight), ptr(pl.scratch),
                                      pl.scratch.numel(), ptr(sums), stream_ptr()))
    s_head.mark_grad()
    return sums[:n_ce], (sums[B:B + n_ce] if t_head is not None else None), sums[2 * B]


class _DecoderConsistency(torch.autograd.Function):
    """One SSLCCT auxiliary decoder from its (perturbed) latent to its consistency term, without the full-resolution planes:
    the decoder's executor stops at its own-resolution logits, csrc/head.hip: pxl_cons_head_fwd evaluates resize + soft-max + MSE
    and parks the gradient for a unit incoming gradient; the backward scales it by the incoming gradient (a device scalar), runs
    the executor's backward from there and returns the latent's gradient (ssl_cct.py:476-484 + autograd)."""

    @staticmethod
    def forward(ctx, x, anchor, core, target, out_size, holder):
        head = core.forward_deferred(x, out_size=out_size, force_graph=True, seam="cons")
        if head is None:
            raise _lib.PixelHipError("decoder_consistency: the fused seam cannot run on this plan (check decoder_consistency_supported first)")
        pl = head.plan
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib().pxl_net_cons_head_fwd(pl.net, ptr(head.arena), ptr(target), ptr(pl.scratch), pl.scratch.numel(), ptr(loss), stream_ptr()))
        pl.grad_gen = getattr(pl, "grad_gen", 0) + 1         # the parked gradient is this plan's seam state from here on
        head._parked_gen = pl.grad_gen
        ctx.head, ctx.x_shape = head, tuple(x.shape)
        holder.append(head)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        head = ctx.head
        pl, core = head.plan, head.core
        if getattr(head, "_parked_gen", None) != getattr(pl, "grad_gen", 0):
            raise _lib.PixelHipError("decoder_consistency: the gradient parked by the forward has been overwritten (another pass / seam ran "
                                     "on the same plan before this backward)")
        g = gout.detach().reshape(1).contiguous().float()
        check(lib().pxl_net_cons_head_bwd(pl.net, ptr(pl.scratch), pl.scratch.numel(), ptr(g), stream_ptr()))
        head.mark_grad()
        head.backward()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.x_shape, device=g.device, dtype=torch.float32)
            check(lib().pxl_net_input_grad(pl.net, ptr(pl.scratch), ptr(dx), stream_ptr()))
        ctx.head = None
        return dx, None, None, None, None, None


def decoder_consistency_supported(core, x, target, out_size):
    """can decoder_consistency run for this decoder / latent / target?  (plans the shape, runs nothing)"""
    if not (x.is_cuda and target.is_cuda and x.dim() == 4 and target.dim() == 4):
        return False
    B, _, H, W = x.shape
    if tuple(target.shape) != (B, core.num_classes) + tuple(out_size) or target.dtype != torch.float32:
        return False
    core._plan(B, H, W, out_size, inference=False)
    return bool(lib().pxl_net_cons_head_supported(core._cur.net))


def decoder_consistency(core, x, target, out_size):
    """MSE(softmax(resize(decoder(x), out_size)), target) of an engine.AuxDecoderCore as ONE differentiable scalar (gradients: the
    decoder's parameters and x).  -> (term, DeferredHead); head.materialize() yields the resized prediction when somebody wants it."""
    _gpu(x, target)
    holder = []
    term = _DecoderConsistency.apply(x.contiguous().float(), core._anchor, core, target.detach().contiguous(), tuple(out_size), holder)
    return term, holder[0]


class MSELoss(torch.nn.Module):
    """Drop-in for the `nn.MSELoss()` the SSL algorithms instantiate."""

    def forward(self, a, b):
        return mse_loss(a, b)


def confusion_matrix(pred, gt, num_classes, out=None):
    """Device confusion matrix of the sseg validation metrics (task/sseg/func.py:41-48): pred [N,C,H,W] fp32 (activated
    or not: only its channel arg-max counts), gt [N,1,H,W] / [N,H,W] float class ids -> int64 [C,C] with
    cm[g][a] = #pixels labeled g (0 <= g < C) predicted a.  Accumulates into `out` when given.  Bit-exact integer counts."""
    _gpu(pred, gt)
    pred = pred.detach().contiguous().float()
    gt = gt.detach().contiguous().float()
    N, C, H, W = pred.shape
    if C != num_classes:
        raise ValueError("confusion_matrix: prediction has %d channels, num_classes = %d" % (C, num_classes))
    if gt.numel() != N * H * W:
        raise ValueError("confusion_matrix: gt %s does not match prediction %s" % (tuple(gt.shape), tuple(pred.shape)))
    cm = out if out is not None else torch.zeros(C, C, device=pred.device, dtype=torch.int64)
    check(lib().pxl_confusion_matrix(N, C, H * W, ptr(pred), ptr(gt), ptr(cm), stream_ptr()))
    return cm


def argmax_u8(pred):
    """[N,C,H,W] fp32 -> uint8 [N,H,W] channel arg-max (np.argmax tie / NaN rules)."""
    _gpu(pred)
    pred = pred.detach().contiguous().float()
    N, C, H, W = pred.shape
    out = torch.empty(N, H, W, device=pred.device, dtype=torch.uint8)
    check(lib().pxl_argmax_u8(N, C, H * W, ptr(pred), ptr(out), stream_ptr()))
    return out


class _SoftmaxChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _gpu(x)
        x = x.contiguous().float()
        N, C = x.shape[0], x.shape[1]
        p = torch.empty_like(x)
        check(lib().pxl_softmax_nchw_fwd(N, C, x.numel() // (N * C), ptr(x), ptr(p), stream_ptr()))
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        (p,) = ctx.saved_tensors
        N, C = p.shape[0], p.shape[1]
        dx = torch.empty_like(p)
        check(lib().pxl_softmax_nchw_bwd(N, C, p.numel() // (N * C), ptr(p), ptr(dp.contiguous().float()), ptr(dx), stream_ptr()))
        return dx


def softmax_channels(x):
    """F.softmax(x, dim=1) of an NCHW fp32 prediction (the sseg activation), differentiable."""
    return _SoftmaxChannels.apply(x)
