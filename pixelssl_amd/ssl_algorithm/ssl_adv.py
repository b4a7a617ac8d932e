"""Adversarial-learning-based semi-supervised learning (pixelssl/ssl_algorithm/ssl_adv.py): a task model and a
fully-convolutional discriminator on the task model's softmax.  Step 1 trains the task model with CE on the labeled
slice + the adversarial constraint (the discriminator only relays dL/dsoftmax); step 2 trains the discriminator on
detached predictions (fake) and one-hot ground truth (real) with Adam(0.9, 0.99) and a polynomial LR."""
import os
import time

import torch
import torch.nn as nn

from ..utils import REGRESSION, CLASSIFICATION, logger, cmd, tool
from ..nn import func
from ..nn.lrer import PolynomialLR
from ..nn.optimizer import FusedAdam
from ..nn.module import patch_replication_callback
from .. import _lib, dist as pdist, streams
from .._lib import check, lib, ptr, stream_ptr
from ..engine import FCDiscriminatorCore
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--adv-for-labeled', type=cmd.str2bool, default=False,
                        help='ssladv - calculate the adversarial constraint on the labeled data if True')
    parser.add_argument('--labeled-adv-scale', type=float, default=-1, help='ssladv - adversarial constraint coefficient of labeled data')
    parser.add_argument('--unlabeled-adv-scale', type=float, default=-1, help='ssladv - adversarial constraint coefficient of unlabeled data')
    parser.add_argument('--discriminator-lr', type=float, default=1e-4, help='ssladv - the initial learning rate of the FC discriminator')
    parser.add_argument('--discriminator-power', type=float, default=0.9, help='ssladv - power of the PolynomialLR of the FC discriminator')
    parser.add_argument('--unlabeled-for-discriminator', type=cmd.str2bool, default=False,
                        help='ssladv - train FC discriminator with unlabeled data if True')
    parser.add_argument('--discriminator-scale', type=float, default=1.0, help='ssladv - coefficient of the FC discriminator constraint')


def ssl_adv(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_adv', model_dict, optimizer_dict, lrer_dict, criterion_dict)
    algorithm = SSLADV(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


class FCDiscriminator(nn.Module):
    """ssl_adv.py:463-493 on the layer-program executor (engine.FCDiscriminatorCore); parameters conv1..conv4,
    classifier keep the reference's names so its checkpoints load."""
    ndf = 64

    def __init__(self, in_channels, engine_dtype=torch.float32):
        super().__init__()
        core = FCDiscriminatorCore(in_channels, device=pdist.local_device(), engine_dtype=engine_dtype)
        # the executor front-end stays outside the module registry; its leaves carry the reference's attribute names, so
        # the state_dict of this module and of every wrapper around it (`module.conv1.weight`) is the reference's
        object.__setattr__(self, 'core', core)
        for name in ("conv1", "conv2", "conv3", "conv4", "classifier"):
            self.add_module(name, getattr(core, name))

    def train(self, mode=True):
        self.core.train(mode)
        return super().train(mode)

    def forward(self, task_pred):
        conf, _, _ = self.core(task_pred)
        assert conf.shape[2:] == task_pred.shape[2:]
        # not activated here: FCDiscriminatorCriterion applies the sigmoid inside BCE-with-logits
        return {'confidence': conf}, {}


class _MaskedBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, task_gt, ignore_index, target):
        pred = pred.contiguous()
        B = pred.shape[0]
        loss = torch.empty(B, device=pred.device, dtype=torch.float32)
        gt = None if task_gt is None else task_gt.contiguous().float()
        check(lib().pxl_bce_logits_masked_fwd(B, pred.numel() // B, ptr(pred), ptr(gt), ignore_index, float(target),
                                              ptr(loss), stream_ptr()))
        ctx.save_for_backward(pred, gt)
        ctx.meta = (ignore_index, float(target))
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, gt = ctx.saved_tensors
        B = pred.shape[0]
        dx = torch.empty_like(pred)
        check(lib().pxl_bce_logits_masked_bwd(B, pred.numel() // B, ptr(pred), ptr(gt), ctx.meta[0], ctx.meta[1],
                                              ptr(gout.contiguous().float()), ptr(dx), stream_ptr()))
        return dx, None, None, None


class _PlainBCE(torch.autograd.Function):
    """BCE-with-logits, mean over (1,2,3) per sample, on plain (pred, target) tensors."""

    @staticmethod
    def forward(ctx, pred, target):
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        B = pred.shape[0]
        loss = torch.empty(B, device=pred.device, dtype=torch.float32)
        check(lib().pxl_bce_logits_fwd(B, pred.numel() // B, ptr(pred), ptr(target), ptr(loss), stream_ptr()))
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target = ctx.saved_tensors
        B = pred.shape[0]
        dx = torch.empty_like(pred)
        check(lib().pxl_bce_logits_bwd(B, pred.numel() // B, ptr(pred), ptr(target), ptr(gout.contiguous().float()), ptr(dx),
                                       stream_ptr()))
        return dx, None


class FCDiscriminatorCriterion(nn.Module):
    """ssl_adv.py:496-503: `F.binary_cross_entropy_with_logits(pred, gt, reduction='none')` averaged over (1,2,3) ->
    one value per sample, for any (pred, gt) pair of tensors.  When the pair comes from the sseg hook
    `ssladv_preprocess_fcd_criterion` (it carries the un-masked prediction and the task labels, sseg/func.py) the mask,
    the target and the loss run as ONE kernel on the original prediction instead of three passes."""

    def forward(self, pred, gt):
        if not pred.is_cuda:
            raise _lib.PixelHipError("FCDiscriminatorCriterion runs on the GPU only; there is no CPU path")
        src = getattr(pred, '_pxl_fcd_source', None)
        if (src is not None and getattr(gt, '_pxl_fcd_source', None) is src
                and pred._version == pred._pxl_fcd_version and gt._version == gt._pxl_fcd_version):
            raw_pred, task_gt, ignore_index, is_real = src
            return _MaskedBCE.apply(raw_pred, task_gt, ignore_index, 1.0 if is_real else 0.0)
        return _PlainBCE.apply(pred, gt)


class SSLADV(ssl_base._SSLBase):
    NAME = 'ssl_adv'
    SUPPORTED_TASK_TYPES = [REGRESSION, CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.model = self.d_model = None
        self.optimizer = self.d_optimizer = None
        self.lrer = self.d_lrer = None
        self.criterion = self.d_criterion = None
        self.args.discriminator_lr *= getattr(self.args, 'gpus', 1)       # ssl_adv.py:72
        if self.args.adv_for_labeled and self.args.labeled_adv_scale < 0:
            logger.log_err('The argument - labeled_adv_scale - is not set (or invalid)\n'
                           'Please set - labeled_adv_scale >= 0 - for the adversarial loss on the labeled data\n')
        if self.args.unlabeled_batch_size > 0 and self.args.unlabeled_adv_scale < 0:
            logger.log_err('The argument - unlabeled_adv_scale - is not set (or invalid)\n'
                           'Please set - unlabeled_adv_scale >= 0 - for the adversarial loss on the unlabeled data\n')

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        self.model = patch_replication_callback(func.create_model(model_funcs[0], 'model', args=self.args))
        d_dtype = torch.float32 if getattr(self.args, 'engine_dtype', 'bf16') in ('fp32', 'f32') else torch.bfloat16
        self.d_model = patch_replication_callback(func.create_model(
            FCDiscriminator, 'd_model', in_channels=self.task_func.ssladv_fcd_in_channels(), engine_dtype=d_dtype))
        self.models = {'model': self.model, 'd_model': self.d_model}
        self.optimizer = optimizer_funcs[0](self.model.module.param_groups)
        self.d_optimizer = FusedAdam(filter(lambda p: p.requires_grad, self.d_model.parameters()),
                                     lr=self.args.discriminator_lr, betas=(0.9, 0.99))
        self.optimizers = {'optimizer': self.optimizer, 'd_optimizer': self.d_optimizer}
        self.lrer = lrer_funcs[0](self.optimizer)
        self.d_lrer = PolynomialLR(self.d_optimizer, self.args.epochs, self.args.iters_per_epoch,
                                   power=self.args.discriminator_power, last_epoch=-1)
        self.lrers = {'lrer': self.lrer, 'd_lrer': self.d_lrer}
        self.criterion = criterion_funcs[0](self.args)
        self.d_criterion = FCDiscriminatorCriterion()
        self.criterions = {'criterion': self.criterion, 'd_criterion': self.d_criterion}

    def _side_stream(self):
        if not hasattr(self, '_d_stream'):
            on = os.environ.get('PXL_ADV_STREAMS', '1') != '0' and torch.cuda.is_available()
            self._d_stream = streams.role_stream(streams.SIDE) if on else None
        return self._d_stream

    def train_step(self, inp, gt):
        """One iteration of ssl_adv.py:126-246 on device-resident tuples -> dict of detached loss scalars."""
        a = self.args
        lbs = a.labeled_batch_size
        d_core = self.d_model.module.core
        # Step 1 (task model, the discriminator a fixed function) and step 2 (discriminator on the DETACHED step-1
        # predictions) only share the task forward: step 1 differentiates through a frozen twin of the discriminator
        # (same weights, own plans, no weight gradients) on the main stream while step 2 runs -- forward, backward, Adam
        # -- on a side stream.  PXL_ADV_STREAMS=0: strictly sequential on one stream.
        side = self._side_stream()
        main = torch.cuda.current_stream() if side is not None else None
        if not hasattr(self, '_d_frozen'):
            self._d_frozen = d_core.twin()
            self._d_frozen.set_wgrad(False)
        self._d_frozen.train(d_core.training)
        # ---- step 1: task model; the discriminator's own gradients are discarded by the reference's
        # d_optimizer.zero_grad() below, so they are not computed at all
        self.optimizer.zero_grad()
        resulter, _ = self.model.forward(inp)
        self._need_pred(resulter, 'SSL_ADV')
        pred = tool.dict_value(resulter, 'pred')
        activated_pred = tool.dict_value(resulter, 'activated_pred')
        l_pred = func.split_tensor_tuple(pred, 0, lbs)
        l_gt = func.split_tensor_tuple(gt, 0, lbs)
        l_inp = func.split_tensor_tuple(inp, 0, lbs)

        def step2():
            # ---- step 2: discriminator on detached predictions (fake) and one-hot ground truth (real)
            self.d_optimizer.zero_grad()
            fake_pred = activated_pred[0].detach() if a.unlabeled_for_discriminator else activated_pred[0][:lbs, ...].detach()
            fake_map = tool.dict_value(self.d_model.forward(fake_pred)[0], 'confidence')
            fp, fg = self.task_func.ssladv_preprocess_fcd_criterion(fake_map[:lbs, ...], l_gt[0], False)
            fake_losses = [self.d_criterion.forward(fp, fg)]
            if a.unlabeled_for_discriminator and a.unlabeled_batch_size != 0:
                up, ug = self.task_func.ssladv_preprocess_fcd_criterion(fake_map[lbs:a.batch_size, ...], None, False)
                fake_losses.append(self.d_criterion.forward(up, ug))
            fake_d = a.discriminator_scale * torch.mean(torch.cat(fake_losses, dim=0))
            real_gt = self.task_func.ssladv_convert_task_gt_to_fcd_input(l_gt[0])
            real_map = tool.dict_value(self.d_model.forward(real_gt)[0], 'confidence')
            rp, rg = self.task_func.ssladv_preprocess_fcd_criterion(real_map, l_gt[0], True)
            real_d = a.discriminator_scale * torch.mean(self.d_criterion(rp, rg))
            ((fake_d + real_d) / 2).backward()
            self.d_optimizer.step()
            return fake_d, real_d

        # the frozen twin packs its bf16 weights from the shared fp32 parameters inside this call: it is enqueued BEFORE
        # step 2 is released, so the discriminator's Adam update of this iteration cannot overtake it
        confidence_map = self._d_frozen(activated_pred[0])[0]
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fake_d_loss, real_d_loss = step2()
        task_loss = torch.mean(self.criterion.forward(l_pred, l_gt, l_inp))
        zero = torch.zeros((), device=task_loss.device)
        labeled_adv_loss = unlabeled_adv_loss = zero
        if a.adv_for_labeled:
            p, g = self.task_func.ssladv_preprocess_fcd_criterion(confidence_map[:lbs, ...], l_gt[0], True)
            labeled_adv_loss = a.labeled_adv_scale * torch.mean(self.d_criterion(p, g))
        if a.unlabeled_batch_size > 0:
            p, g = self.task_func.ssladv_preprocess_fcd_criterion(confidence_map[lbs:a.batch_size, ...], None, True)
            unlabeled_adv_loss = a.unlabeled_adv_scale * torch.mean(self.d_criterion(p, g))
        loss = task_loss + labeled_adv_loss + unlabeled_adv_loss
        loss.backward()
        self.optimizer.step()
        if side is not None:
            main.wait_stream(side)
            for t in (fake_d_loss, real_d_loss):
                t.record_stream(main)
        else:
            fake_d_loss, real_d_loss = step2()
        self.d_lrer.step()
        if not a.is_epoch_lrer:
            self.lrer.step()
        return dict(task_loss=task_loss.detach(), labeled_adv_loss=labeled_adv_loss.detach(),
                    unlabeled_adv_loss=unlabeled_adv_loss.detach(), fake_d_loss=fake_d_loss.detach(),
                    real_d_loss=real_d_loss.detach()), resulter

    def _train(self, data_loader, epoch):
        self.meters.reset()
        self.model.train()
        self.d_model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            losses, _ = self.train_step(inp, gt)
            for k, v in losses.items():
                self.meters.update(k, v)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                m = self.meters
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  task-{4}\t=>\ttask-loss: {5:.6f}\tlabeled-adv-loss: {6:.6f}\tunlabeled-adv-loss: {7:.6f}\n'
                                '  fc-discriminator\t=>\tfake-d-loss: {8:.6f}\treal-d-loss: {9:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), m['batch_time'].avg, self.args.task,
                                        float(m['task_loss'].avg), float(m['labeled_adv_loss'].avg),
                                        float(m['unlabeled_adv_loss'].avg), float(m['fake_d_loss'].avg),
                                        float(m['real_d_loss'].avg)))
        if self.args.is_epoch_lrer:
            self.lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_adv.py:285-340: task model in eval mode, task loss + metrics (the discriminator's confidence map is only
        visualised there)."""
        self.meters.reset()
        self.model.eval()
        self.d_model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            resulter, _ = self.model.forward(inp)
            self._need_pred(resulter, 'SSL_ADV')
            pred = tool.dict_value(resulter, 'pred')
            self.meters.update('task_loss', torch.mean(self.criterion.forward(pred, gt, inp)).detach())
            self.task_func.metrics(tool.dict_value(resulter, 'activated_pred'), gt, inp, self.meters, id_str='task')
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n  task-{4}\t=>\ttask-loss: {5:.6f}\t'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['task_loss'].avg)))
        self._log_validation_metrics(['task'])

    def _save_checkpoint(self, epoch):
        state = {'algorithm': self.NAME, 'epoch': epoch, 'model': self.model.state_dict(),
                 'd_model': self.d_model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                 'd_optimizer': self.d_optimizer.state_dict(), 'lrer': self.lrer.state_dict(),
                 'd_lrer': self.d_lrer.state_dict()}
        torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

    def _load_checkpoint(self):
        checkpoint = torch.load(self.args.resume, map_location='cpu')
        found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
        if found != self.NAME:
            logger.log_err('Unmatched SSL algorithm format in checkpoint => required: {0} - given: {1}\n'
                           .format(self.NAME, found))
        self.model.load_state_dict(checkpoint['model'])
        self.d_model.load_state_dict(checkpoint['d_model'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])      # ssl_adv.py:380-385
        self.d_optimizer.load_state_dict(checkpoint['d_optimizer'])
        self.lrer.load_state_dict(checkpoint['lrer'])
        self.d_lrer.load_state_dict(checkpoint['d_lrer'])
        return checkpoint['epoch']
