"""Pixel-wise S4L (pixelssl/ssl_algorithm/ssl_s4l.py), rotation pretext task, on the MI355X engine.

Every sample of a batch gets one copy rotated by 90 / 180 / 270 degrees (drawn with np.random.randint like the
reference); the task model predicts all 2 * bs samples, a small rotation classifier on top of its `pred`
(`ssls4l_rc_inp`, task/sseg/model.py:63) predicts the rotation class; loss = CE(un-rotated labeled) + rotated_sup_scale
* CE(rotated labeled) + rotation_scale * CE(rotation classes); ONE SGD over task_model.param_groups + [rotation
classifier at the base learning rate].

Device mapping: the rotation classifier (2 x [conv 4x4 / s2 + BatchNorm + LeakyReLU], global pool, Linear) is one
executor program (engine.RotationClassifierCore); its input gradient returns to autograd and reaches the task model's
executor as part of d(pred).  The rotated copies of a batch are written by one tiled-transpose launch (csrc/rotate.hip)."""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from ..utils import REGRESSION, CLASSIFICATION, logger, tool
from ..nn import func
from ..nn.module import patch_replication_callback
from .. import functional as PF
from .. import dist as pdist
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--rotated-sup-scale', type=float, default=-1, help='ssls4l - task-supervised coefficient for rotated labeled data')
    parser.add_argument('--rotation-scale', type=float, default=-1, help='ssls4l - rotation-based self-supervised coefficient')


def ssl_s4l(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_s4l', model_dict, optimizer_dict, lrer_dict,
                                                         criterion_dict)
    algorithm = SSLS4L(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


class RotationCrossEntropy(nn.Module):
    """nn.CrossEntropyLoss() on [N, 4] rotation logits and int64 classes (ssl_s4l.py:98,161): mean over the batch of the
    fused per-sample cross-entropy kernel (one 'pixel' per sample)."""

    def forward(self, pred, gt):
        n, c = pred.shape
        per_sample = PF.cross_entropy_per_sample(pred.reshape(n, c, 1, 1), gt.reshape(n, 1, 1, 1).float(), -100)
        return per_sample.mean()


class RotationClassifer(nn.Module):
    """ssl_s4l.py:371-393 (the reference's spelling); forward(task_pred) -> [N, 4] rotation logits."""

    def __init__(self, in_channels, engine_dtype=torch.float32):
        super().__init__()
        from ..engine import RotationClassifierCore
        core = RotationClassifierCore(in_channels, device=pdist.local_device(), engine_dtype=engine_dtype)
        # the executor front-end stays outside the module registry; its leaves carry the reference's names
        # (conv1, bn1, conv2, bn2, classifier), so state_dict / parameters() are the reference's
        object.__setattr__(self, 'core', core)
        for name, child in core.named_children():
            self.add_module(name, child)

    def train(self, mode=True):
        self.core.train(mode)
        return super().train(mode)

    def forward(self, task_pred):
        return self.core(task_pred)


class WrappedS4LModel(nn.Module):
    """ssl_s4l.py:396-436."""

    def __init__(self, args, task_model, rotation_classifier):
        super().__init__()
        self.args = args
        self.task_model = task_model
        self.rotation_classifier = rotation_classifier
        self.param_groups = self.task_model.param_groups + \
            [{'params': self.rotation_classifier.parameters(), 'lr': self.args.lr}]

    def forward(self, inp):
        resulter, debugger = {}, {}
        t_resulter, _ = self.task_model.forward(inp)
        if 'pred' not in t_resulter.keys() or 'activated_pred' not in t_resulter.keys():
            logger.log_err('SSL_S4L needs both \'pred\' (un-activated) and \'activated_pred\' in the task model\'s resulter:\n'
                           'the task criterion activates on its own (CrossEntropyLoss contains the SoftMax), metrics do not\n')
        if 'ssls4l_rc_inp' not in t_resulter.keys():
            logger.log_err('SSL_S4L needs \'ssls4l_rc_inp\' in the task model\'s resulter: the 4-dim tensor the rotation\n'
                           'classifier reads (a feature map of the task model or its prediction)\n')
        rc_inp = tool.dict_value(t_resulter, 'ssls4l_rc_inp')
        resulter['pred'] = tool.dict_value(t_resulter, 'pred')
        resulter['activated_pred'] = tool.dict_value(t_resulter, 'activated_pred')
        resulter['rotation'] = self.rotation_classifier.forward(rc_inp)
        return resulter, debugger


class SSLS4L(ssl_base._SSLBase):
    NAME = 'ssl_s4l'
    SUPPORTED_TASK_TYPES = [REGRESSION, CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.task_model = self.rotation_classifier = None
        self.model = self.optimizer = self.lrer = self.criterion = self.rotation_criterion = None
        if self.args.rotation_scale < 0:
            logger.log_err('SSL_S4L: --rotation-scale must be set to a value >= 0\n')
        if self.args.rotated_sup_scale < 0:
            logger.log_err('SSL_S4L: --rotated-sup-scale must be set to a value >= 0\n')

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        self.task_model = func.create_model(model_funcs[0], 'task_model', args=self.args).module
        dt = getattr(self.args, 'engine_dtype', 'bf16')
        self.rotation_classifier = RotationClassifer(self.task_func.ssls4l_rc_in_channels(),
                                                     engine_dtype=torch.float32 if dt in ('fp32', 'f32') else torch.bfloat16)
        wrapped = WrappedS4LModel(self.args, self.task_model, self.rotation_classifier)
        self.model = patch_replication_callback(func.RankModel(wrapped))
        pdist.attach(self.model)
        self.models = {'model': self.model}
        self.optimizer = optimizer_funcs[0](self.model.module.param_groups)
        self.optimizers = {'optimizer': self.optimizer}
        self.lrer = lrer_funcs[0](self.optimizer)
        self.lrers = {'lrer': self.lrer}
        self.criterion = criterion_funcs[0](self.args)
        self.rotation_criterion = RotationCrossEntropy()
        self.criterions = {'criterion': self.criterion, 'rotation_criterion': self.rotation_criterion}
        # the batch size is doubled in S4L: every sample gets an extra rotated copy (ssl_s4l.py:101-109)
        self.args.batch_size *= 2
        self.args.labeled_batch_size *= 2
        self.args.unlabeled_batch_size *= 2
        logger.log_info('SSL_S4L doubles the batch (one rotated copy per sample): labeled {0}, unlabeled {1}\n'
                        .format(self.args.labeled_batch_size, self.args.unlabeled_batch_size))
        self._algorithm_warn()

    # -- one iteration -------------------------------------------------------------------------------------------
    def train_step(self, inp, gt):
        """One iteration of ssl_s4l.py:122-207 on the tuples _batch_prehandle returns (rotated copies appended, the rotation
        classes as the last gt) -> (dict of detached meters, resulter)."""
        original_lbs = int(self.args.labeled_batch_size / 2)
        original_bs = int(self.args.batch_size / 2)
        self.optimizer.zero_grad()
        resulter, _ = self.model.forward(inp)
        pred = tool.dict_value(resulter, 'pred')
        pred_rotation = tool.dict_value(resulter, 'rotation')

        l_pred = func.split_tensor_tuple(pred, 0, original_lbs)
        l_gt = func.split_tensor_tuple(gt, 0, original_lbs)
        l_inp = func.split_tensor_tuple(inp, 0, original_lbs)
        unrotated_task_loss = torch.mean(self.criterion.forward(l_pred, l_gt[:-1], l_inp))

        r_pred = func.split_tensor_tuple(pred, original_bs, original_bs + original_lbs)
        r_gt = func.split_tensor_tuple(gt, original_bs, original_bs + original_lbs)
        r_inp = func.split_tensor_tuple(inp, original_bs, original_bs + original_lbs)
        rotated_task_loss = self.args.rotated_sup_scale * torch.mean(self.criterion.forward(r_pred, r_gt[:-1], r_inp))

        rotation_loss = self.args.rotation_scale * torch.mean(self.rotation_criterion.forward(pred_rotation, gt[-1]))

        loss = unrotated_task_loss + rotated_task_loss + rotation_loss
        loss.backward()
        self.optimizer.step()

        _, angle_idx = pred_rotation.detach().topk(1, 1, True, True)
        angle_idx = angle_idx.t()
        rotation_acc = angle_idx.eq(gt[-1].view(1, -1).expand_as(angle_idx))
        rotation_acc = rotation_acc.view(-1).float().sum(0, keepdim=True).mul_(100.0 / self.args.batch_size)
        if not self.args.is_epoch_lrer:
            self.lrer.step()
        return dict(unrotated_task_loss=unrotated_task_loss.detach(), rotated_task_loss=rotated_task_loss.detach(),
                    rotation_loss=rotation_loss.detach(), rotation_acc=rotation_acc[0]), resulter

    def _train(self, data_loader, epoch):
        self.meters.reset()
        self.model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._batch_prehandle(inp, gt, True)
            if len(gt) - 1 > 1 and idx == 0:
                self._inp_warn()
            meters, resulter = self.train_step(inp, gt)
            for k, v in meters.items():
                self.meters.update(k, v)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  task-{4}\t=>\tunrotated-task-loss: {5:.6f}\trotated-task-loss: {6:.6f}\n'
                                '  rotation-{4}\t=>\trotation-loss: {7:.6f}\trotation-acc: {8:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['unrotated_task_loss'].avg), float(self.meters['rotated_task_loss'].avg),
                                        float(self.meters['rotation_loss'].avg), float(self.meters['rotation_acc'].avg)))
            if self.args.visualize and idx % self.args.visual_freq == 0:
                self._visualize(epoch, idx, True, func.split_tensor_tuple(inp, 0, 1, reduce_dim=True),
                                func.split_tensor_tuple(tool.dict_value(resulter, 'activated_pred'), 0, 1, reduce_dim=True),
                                func.split_tensor_tuple(gt[:-1], 0, 1, reduce_dim=True))
        if self.args.is_epoch_lrer:
            self.lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_s4l.py:211-260: eval mode, no rotated copies (every rotation class is 0)."""
        self.meters.reset()
        self.model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._batch_prehandle(inp, gt, False)
            if len(gt) - 1 > 1 and idx == 0:
                self._inp_warn()
            resulter, _ = self.model.forward(inp)
            pred = tool.dict_value(resulter, 'pred')
            activated_pred = tool.dict_value(resulter, 'activated_pred')
            pred_rotation = tool.dict_value(resulter, 'rotation')
            self.meters.update('task_loss', torch.mean(self.criterion.forward(pred, gt[:-1], inp)).detach())
            self.meters.update('rotation_loss', (self.args.rotation_scale *
                                                 torch.mean(self.rotation_criterion.forward(pred_rotation, gt[-1]))).detach())
            self.task_func.metrics(activated_pred, gt[:-1], inp, self.meters, id_str='task')
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  task-{4}\t=>\ttask-loss: {5:.6f}\n  rotation-{4}\t=>\trotation-loss: {6:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['task_loss'].avg), float(self.meters['rotation_loss'].avg)))
            if self.args.visualize and idx % self.args.visual_freq == 0:
                self._visualize(epoch, idx, False, func.split_tensor_tuple(inp, 0, 1, reduce_dim=True),
                                func.split_tensor_tuple(activated_pred, 0, 1, reduce_dim=True),
                                func.split_tensor_tuple(gt[:-1], 0, 1, reduce_dim=True))
        self._log_validation_metrics(['task'])

    def _save_checkpoint(self, epoch):
        state = {'algorithm': self.NAME, 'epoch': epoch, 'model': self.model.state_dict(),
                 'optimizer': self.optimizer.state_dict(), 'lrer': self.lrer.state_dict()}
        torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

    def _load_checkpoint(self):
        checkpoint = torch.load(self.args.resume, map_location='cpu')
        found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
        if found != self.NAME:
            logger.log_err('Unmatched ssl algorithm format in checkpoint => required: {0} - given: {1}\n'
                           .format(self.NAME, found))
        self.model.load_state_dict(checkpoint['model'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])
        self.lrer.load_state_dict(checkpoint['lrer'])
        self.task_model = self.model.module.task_model
        self.rotation_classifier = self.model.module.rotation_classifier
        return checkpoint['epoch']

    # -- tools ---------------------------------------------------------------------------------------------------
    def _visualize(self, epoch, idx, is_train, inp, pred, gt):
        visualize_path = self.args.visual_train_path if is_train else self.args.visual_val_path
        out_path = os.path.join(visualize_path, '{0}_{1}'.format(epoch, idx))
        self.task_func.visualize(out_path, id_str='task', inp=inp, pred=pred, gt=gt)

    def _batch_prehandle(self, inp, gt, is_train):
        """ssl_s4l.py:296-345: rotated copies behind the batch, the rotation classes as the last ground truth.  Same draw
        from numpy's global stream; the copies are built by one kernel launch per tensor."""
        bs = inp[0].shape[0]
        rotation_angles = np.random.randint(low=1, high=4, size=bs)
        inp, gt = self._to_device(inp), self._to_device(gt)
        if is_train:
            inp = tuple(self._with_rotated(i, rotation_angles) for i in inp)
            gt = tuple(self._with_rotated(g, rotation_angles) for g in gt)
        n = inp[0].shape[0]
        rotation_gt = torch.zeros(n, dtype=torch.long, device=inp[0].device)
        if is_train:
            rotation_gt[bs:] = torch.from_numpy(rotation_angles.astype(np.int64)).to(rotation_gt.device)
        return inp, gt + (rotation_gt,)

    # angle index of _rotate_tensor (ssl_s4l.py:347-355) -> quarter turns of torch.rot90 over (H, W): 1 = clockwise
    _QUARTER_TURNS = {0: 0, 1: -1, 2: 2, 3: 1}

    _SRC_KIND = {torch.float32: 0, torch.int64: 1, torch.uint8: 2}

    def _with_rotated(self, t, angles):
        """[bs, C, N, N] -> fp32 [2 * bs, C, N, N]: the batch followed by one rotated copy per sample -- ONE launch of
        csrc/rotate.hip (pxl_rotate_append: 32 x 32 tiles through LDS, unit-stride reads and writes for every angle)
        instead of the reference's per-sample transposes / flips into a zero tensor (ssl_s4l.py:296-355)."""
        import ctypes
        from .._lib import lib, check, ptr, stream_ptr, PixelHipError
        bs = t.shape[0]
        if not (t.dim() == 4 and bs == len(angles) and t.shape[-1] == t.shape[-2]):
            raise PixelHipError('SSL_S4L rotates [bs, C, N, N] tensors by quarter turns (square maps), got %s for %d angles'
                                % (tuple(t.shape), len(angles)))
        if not t.is_cuda:
            raise PixelHipError('SSL_S4L builds the rotated copies on the GPU (csrc/rotate.hip); got a %s tensor' % t.device)
        if t.dtype not in self._SRC_KIND:
            t = t.float()
        t = t.contiguous()
        out = torch.empty((2 * bs,) + tuple(t.shape[1:]), device=t.device, dtype=torch.float32)
        host_angles = (ctypes.c_int * bs)(*[int(v) for v in angles])
        check(lib().pxl_rotate_append(self._SRC_KIND[t.dtype], bs, t.shape[1], t.shape[2], ptr(t), host_angles, ptr(out),
                                      stream_ptr()))
        return out

    def _rotate_tensor(self, tensor, angle_idx):
        """One [C, H, W] tensor, same convention (kept for callers of the reference's helper)."""
        return torch.rot90(tensor, self._QUARTER_TURNS[int(angle_idx)], (-2, -1))

    def _inp_warn(self):
        logger.log_warn('SSL_S4L received several task ground truths: predictions and ground truths are paired by index,\n'
                        'and every ground truth must be a 4-dim tensor (it is rotated together with its input);\n'
                        'other ground-truth formats need their own SSL algorithm\n')

    def _algorithm_warn(self):
        logger.log_warn('SSL_S4L follows \'S4L: Self-Supervised Semi-Supervised Learning\' adapted to pixel-wise tasks;\n'
                        'only the 4-angle rotation pretext task (0 / 90 / 180 / 270 degrees) is implemented\n')
