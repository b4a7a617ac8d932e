"""CutMix consistency training (pixelssl/ssl_algorithm/ssl_cutmix.py): student CE on the labeled samples; the
teacher (EMA, no grad) predicts the unlabeled originals; both the unlabeled images and the teacher's softmax maps are
mixed half-with-half through a random box mask; the student's softmax of the mixed images is pulled (MSE x one
confidence scalar) towards the mixed teacher maps."""
import os
import time

import numpy as np
import torch

from ..utils import CLASSIFICATION, logger, cmd, tool
from ..nn import func
from ..nn.module import patch_replication_callback
from ..functional import MSELoss
from .. import ops, _lib
from .._lib import check, lib, ptr, stream_ptr
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--cons-type', type=str, default='mse', choices=['mse'], help='sslcutmix - consistency constraint type')
    parser.add_argument('--cons-scale', type=float, default=-1, help='sslcutmix - consistency constraint coefficient')
    parser.add_argument('--cons-rampup-epochs', type=int, default=-1, help='sslcutmix - ramp-up epochs of the constraint')
    parser.add_argument('--cons-threshold', type=float, default=-1, help='sslcutmix - confidence threshold of the constraint')
    parser.add_argument('--ema-decay', type=float, default=0.99, help='sslcutmix - EMA coefficient of the teacher')
    parser.add_argument('--mask-prop-range', type=cmd.str2floatlist, default='(0.5, 0.5)', help='sslcutmix - mixing ratio range')


def ssl_cutmix(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_cutmix', model_dict, optimizer_dict, lrer_dict, criterion_dict)
    algorithm = SSLCUTMIX(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


class BoxMaskGenerator:
    """One random box per mask (ssl_cutmix.py:481-547, as instantiated at :127-128: area proportion drawn from
    `prop_range`, random aspect ratio, box inside the image, inverted = 1 inside the box).  Generated on the host with
    numpy's global RNG in the reference's draw order (area, aspect, position), so a seeded run reproduces the
    reference's boxes; a mask is B x H x W floats, the upload is negligible."""

    def __init__(self, prop_range, boxes_num=1, random_aspect_ratio=True, area_prop=True, within_bounds=True,
                 invert=False, rng=None):
        if boxes_num != 1 or not (random_aspect_ratio and area_prop and within_bounds and invert):
            raise NotImplementedError('BoxMaskGenerator: only the configuration SSL_CUTMIX uses is implemented')
        self.prop_range = tuple(prop_range)
        self.rng = rng if rng is not None else np.random

    def produce(self, mask_num, mask_shape):
        r = self.rng
        area = r.uniform(self.prop_range[0], self.prop_range[1], size=(mask_num, 1))
        split = r.uniform(low=0.0, high=1.0, size=(mask_num, 1))
        nonzero = area != 0.0
        hfrac = np.where(nonzero, np.exp(split * np.log(np.where(nonzero, area, 1.0))), 0.0)
        wfrac = np.where(nonzero, area / np.where(nonzero, hfrac, 1.0), 0.0)
        extent = np.array(mask_shape, dtype=np.float64)
        size = np.round(np.stack([hfrac, wfrac], axis=2) * extent[None, None, :])
        origin = np.round((extent - size) * r.uniform(low=0.0, high=1.0, size=size.shape))
        out = np.zeros((mask_num, 1) + tuple(mask_shape), dtype=np.float32)
        for n in range(mask_num):
            (top, left), (h, w) = origin[n, 0], size[n, 0]
            out[n, 0, int(top):int(top + h), int(left):int(left + w)] = 1.0
        return out


def cutmix_mix(mask, a, b, threshold=None):
    """mask*a + (1-mask)*b on the device (csrc/loss.hip); with `threshold` also returns the confidence scalar
    mean(max_c(mixed) > threshold) of ssl_cutmix.py:200."""
    if not a.is_cuda:
        raise _lib.PixelHipError("cutmix_mix runs on the GPU only; there is no CPU path")
    a, b, mask = a.contiguous(), b.contiguous(), mask.contiguous()
    B, C, H, W = a.shape
    out = torch.empty_like(a)
    cnt = torch.empty(1, device=a.device, dtype=torch.float32) if threshold is not None else None
    check(lib().pxl_cutmix_mix(B, C, H * W, ptr(mask), ptr(a), ptr(b), ptr(out),
                               float(threshold) if threshold is not None else 0.0, ptr(cnt), stream_ptr()))
    if threshold is None:
        return out
    return out, (cnt / float(B * H * W)).view(())


class SSLCUTMIX(ssl_base._SSLBase):
    NAME = 'ssl_cutmix'
    SUPPORTED_TASK_TYPES = [CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.s_model = self.t_model = None
        self.s_optimizer = self.s_lrer = self.s_criterion = self.cons_criterion = None
        self.mask_generator = None
        if self.args.unlabeled_batch_size > 0:
            if not self.args.unlabeled_batch_size > 2 or not self.args.unlabeled_batch_size % 2 == 0:
                logger.log_err('This implementation of SSL_CUTMIX requires the unlabeled batch size: \n'
                               '    1. larger than 2 \n    2. is divisible by 2 \n')
            if self.args.cons_scale < 0:
                logger.log_err('The argument - cons_scale - is not set (or invalid)\n')
            if self.args.cons_rampup_epochs < 0:
                logger.log_err('The argument - cons_rampup_epochs - is not set (or invalid)\n')
            if self.args.cons_threshold < 0 or self.args.cons_threshold > 1:
                logger.log_err('The argument - cons_threshold - is not set (or invalid)\n')

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        self.s_model = patch_replication_callback(func.create_model(model_funcs[0], 's_model', args=self.args))
        self.t_model = patch_replication_callback(func.create_model(model_funcs[0], 't_model', args=self.args))
        for param in self.t_model.parameters():
            param.detach_()
        self.models = {'s_model': self.s_model, 't_model': self.t_model}
        self.s_optimizer = optimizer_funcs[0](self.s_model.module.param_groups)
        self.optimizers = {'s_optimizer': self.s_optimizer}
        self.s_lrer = lrer_funcs[0](self.s_optimizer)
        self.lrers = {'s_lrer': self.s_lrer}
        self.s_criterion = criterion_funcs[0](self.args)
        self.cons_criterion = MSELoss()
        self.criterions = {'s_criterion': self.s_criterion, 'cons_criterion': self.cons_criterion}
        self.mask_generator = BoxMaskGenerator(prop_range=self.args.mask_prop_range, boxes_num=1, random_aspect_ratio=True,
                                               area_prop=True, within_bounds=True, invert=True)

    def train_step(self, inp, gt, cur_step, total_rampup_steps):
        """One iteration of ssl_cutmix.py:140-227 on device-resident tuples."""
        a = self.args
        lbs, ubs = a.labeled_batch_size, a.unlabeled_batch_size
        half = int(ubs / 2)
        ramp = func.sigmoid_rampup(cur_step, total_rampup_steps)
        self.s_optimizer.zero_grad()
        l_inp = func.split_tensor_tuple(inp, 0, lbs)
        l_gt = func.split_tensor_tuple(gt, 0, lbs)
        zero = torch.zeros((), device=inp[0].device)
        cons_loss = zero
        teacher = None
        if ubs > 0:
            # box masks on the host (numpy, like the reference), mixing on the device
            shape = (inp[0].shape[2], inp[0].shape[3])
            mask = torch.from_numpy(self.mask_generator.produce(half, shape)).to(inp[0].device, non_blocking=True)
            mix_u_inp = tuple(cutmix_mix(mask, i[lbs:lbs + half], i[lbs + half:lbs + ubs]) for i in inp)
            u_inp = func.split_tensor_tuple(inp, lbs, lbs + ubs)
            with torch.no_grad():
                u_t_resulter, _ = self.t_model.forward(u_inp)
            self._need_pred(u_t_resulter, 'SSL_CUTMIX')
            teacher = tool.dict_value(u_t_resulter, 'activated_pred')
        l_s_resulter, _ = self.s_model.forward(l_inp)
        self._need_pred(l_s_resulter, 'SSL_CUTMIX')
        task_loss = torch.mean(self.s_criterion.forward(tool.dict_value(l_s_resulter, 'pred'), l_gt, l_inp))
        if ubs > 0:
            u_s_resulter, _ = self.s_model.forward(mix_u_inp)
            self._need_pred(u_s_resulter, 'SSL_CUTMIX')
            student = tool.dict_value(u_s_resulter, 'activated_pred')
            total = zero
            for sap, tap in zip(student, teacher):
                mixed_t, confidence = cutmix_mix(mask, tap[:half], tap[half:], a.cons_threshold)
                total = total + torch.mean(self.cons_criterion(sap, mixed_t.detach())) * confidence.detach()
            cons_loss = ramp * a.cons_scale * torch.mean(total)
        loss = task_loss + cons_loss
        loss.backward()
        self.s_optimizer.step()
        self._update_ema_variables(self.s_model, self.t_model, a.ema_decay, cur_step)
        if not a.is_epoch_lrer:
            self.s_lrer.step()
        return dict(task_loss=task_loss.detach(), cons_loss=cons_loss.detach())

    def _train(self, data_loader, epoch):
        self.meters.reset()
        self.s_model.train()
        self.t_model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            cur_step = len(data_loader) * epoch + idx
            losses = self.train_step(inp, gt, cur_step, len(data_loader) * self.args.cons_rampup_epochs)
            for k, v in losses.items():
                self.meters.update(k, v)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  student-{4}\t=>\ts-task-loss: {5:.6f}\ts-cons-loss: {6:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['task_loss'].avg), float(self.meters['cons_loss'].avg)))
        if self.args.is_epoch_lrer:
            self.s_lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_cutmix.py:257-316: student and teacher in eval mode, task losses and metrics of both."""
        self.meters.reset()
        self.s_model.eval()
        self.t_model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            for tag, model in (('student', self.s_model), ('teacher', self.t_model)):
                resulter, _ = model.forward(inp)
                self._need_pred(resulter, 'SSL_CUTMIX')
                pred = tool.dict_value(resulter, 'pred')
                self.meters.update(tag[0] + '_task_loss', torch.mean(self.s_criterion.forward(pred, gt, inp)).detach())
                self.task_func.metrics(tool.dict_value(resulter, 'activated_pred'), gt, inp, self.meters, id_str=tag)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  student-{4}\t=>\ts-task-loss: {5:.6f}\n  teacher-{4}\t=>\tt-task-loss: {6:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['s_task_loss'].avg), float(self.meters['t_task_loss'].avg)))
        self._log_validation_metrics(['student', 'teacher'])

    def _save_checkpoint(self, epoch):
        state = {'algorithm': self.NAME, 'epoch': epoch, 's_model': self.s_model.state_dict(),
                 't_model': self.t_model.state_dict(), 's_optimizer': self.s_optimizer.state_dict(),
                 's_lrer': self.s_lrer.state_dict()}
        torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

    def _load_checkpoint(self):
        checkpoint = torch.load(self.args.resume, map_location='cpu')
        found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
        if found != self.NAME:
            logger.log_err('Unmatched SSL algorithm format in checkpoint => required: {0} - given: {1}\n'
                           .format(self.NAME, found))
        self.s_model.load_state_dict(checkpoint['s_model'])
        self.t_model.load_state_dict(checkpoint['t_model'])
        self.s_optimizer.load_state_dict(checkpoint['s_optimizer'])  # ssl_cutmix.py:355-358
        self.s_lrer.load_state_dict(checkpoint['s_lrer'])
        return checkpoint['epoch']

    def _update_ema_variables(self, s_model, t_model, ema_decay, cur_step):
        """alpha = min(1 - 1/(step+1), decay); parameters only, BN buffers evolve by the teacher's own
        forward (ssl_cutmix.py:432-436).  Engine task models: one fused launch over the flat parameter buffers; any other
        TaskModel (a plugin's torch model): the reference's per-parameter walk."""
        alpha = min(1 - 1 / (cur_step + 1), ema_decay)
        s_core, t_core = getattr(s_model.module, 'model', None), getattr(t_model.module, 'model', None)
        if hasattr(s_core, 'flat') and hasattr(t_core, 'flat') and s_core.flat.np == t_core.flat.np and \
                len(list(s_model.parameters())) == len(s_core._param_list):
            ops.ema_update(t_core.flat.params, s_core.flat.params, alpha)
            t_core.mark_params_changed()
            return
        with torch.no_grad():
            for t_param, s_param in zip(t_model.parameters(), s_model.parameters()):
                t_param.mul_(alpha).add_(s_param.detach(), alpha=1 - alpha)
        for core in (m for m in t_model.modules() if hasattr(m, 'mark_params_changed')):
            core.mark_params_changed()
