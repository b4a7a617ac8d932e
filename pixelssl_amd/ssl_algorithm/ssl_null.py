"""SupOnly baseline (pixelssl/ssl_algorithm/ssl_null.py): zero_grad -> forward -> per-sample CE on the
labeled slice -> backward -> SGD step -> per-iteration LR step."""
import os
import time

import torch

from ..utils import REGRESSION, CLASSIFICATION, logger, tool
from ..nn import func
from ..nn.module import patch_replication_callback
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)


def ssl_null(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_null', model_dict, optimizer_dict, lrer_dict,
                                                         criterion_dict)
    algorithm = SSLNULL(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


class SSLNULL(ssl_base._SSLBase):
    NAME = 'ssl_null'
    SUPPORTED_TASK_TYPES = [REGRESSION, CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.model = self.optimizer = self.lrer = self.criterion = None

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        self.model = patch_replication_callback(func.create_model(model_funcs[0], 'model', args=self.args))
        self.models = {'model': self.model}
        self.optimizer = optimizer_funcs[0](self.model.module.param_groups)
        self.optimizers = {'optimizer': self.optimizer}
        self.lrer = lrer_funcs[0](self.optimizer)
        self.lrers = {'lrer': self.lrer}
        self.criterion = criterion_funcs[0](self.args)
        self.criterions = {'criterion': self.criterion}

    def train_step(self, inp, gt):
        """One iteration of ssl_null.py:97-136 on already-device-resident tuples; returns the loss tensor."""
        lbs = self.args.labeled_batch_size
        self.optimizer.zero_grad()
        if lbs > 0 and self._seam_fusable([self.model], self.criterion, inp, gt):
            # fused seam: forward to the low-resolution logits, criterion + its backward on them (no 21 x 513 x 513 planes)
            from .. import functional as PF
            from ..sseg.model import _DeferredResulter
            head = self.model.module.forward_deferred(inp)
            ce, _, _ = PF.head_losses(head, None, gt[0][:lbs], lbs, 0, 0, 1.0 / lbs, 0.0, self.args.ignore_index)
            task_loss = torch.mean(ce)
            pipe = self._update_pipeline(head)
            if pipe is not None:        # SGD + bf16 weight copies + gradient memset as one fused kernel from inside the backward pass
                pipe.arm(head.plan)
            head.backward()
            self.optimizer.step()
            if not self.args.is_epoch_lrer:
                self.lrer.step()
            return task_loss.detach(), _DeferredResulter(head)
        resulter, _ = self.model.forward(inp)
        self._need_pred(resulter, 'SSL_NULL')
        pred = tool.dict_value(resulter, 'pred')
        task_loss = torch.mean(self.criterion.forward(func.split_tensor_tuple(pred, 0, lbs),
                                                      func.split_tensor_tuple(gt, 0, lbs),
                                                      func.split_tensor_tuple(inp, 0, lbs)))
        task_loss.backward()
        self.optimizer.step()
        if not self.args.is_epoch_lrer:
            self.lrer.step()
        return task_loss.detach(), resulter

    def _update_pipeline(self, head):
        """nn.optimizer.PipelinedUpdate for the model (fused parameter update, see SSLMT._update_pipeline), or None: PXL_PIPE_UPDATE=0,
        an fp32 engine, an optimizer / model it does not cover (several ranks: behind the bucketed gradient all-reduce)"""
        if not hasattr(self, '_pipe'):
            import os
            self._pipe = None
            mode = os.environ.get('PXL_PIPE_UPDATE', 'auto')
            if mode == '1' or (mode == 'auto' and os.environ.get('PXL_FUSED_UPDATE', '1') == '1' and
                               getattr(head.core, '_code', None) == 1):
                from ..nn.optimizer import PipelinedUpdate
                try:
                    if len(list(self.model.parameters())) != len(head.core._param_list):
                        raise ValueError('the model holds parameters outside its engine network')
                    self._pipe = PipelinedUpdate(self.optimizer, head.core, None)
                except ValueError as e:
                    logger.log_info('fused parameter update off: %s\n' % e)
        return self._pipe

    def _train(self, data_loader, epoch):
        if not (self.args.ignore_unlabeled and self.args.unlabeled_batch_size == 0):
            logger.log_err('SSL_NULL is a supervised-only algorithm\n'
                           'Please set ignore_unlabeled = True and unlabeled_batch_size = 0\n')
        self.meters.reset()
        self.model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            task_loss, _ = self.train_step(inp, gt)
            self.meters.update('task_loss', task_loss)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n  task-{4}\t=>\ttask-loss: {5:.6f}\t'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg,
                                        self.args.task, float(self.meters['task_loss'].avg)))
        if self.args.is_epoch_lrer:
            self.lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_null.py:146-187: eval-mode forward (running BN statistics) at whatever size the loader yields."""
        self.meters.reset()
        self.model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            resulter, _ = self.model.forward(inp)
            self._need_pred(resulter, 'SSL_NULL')
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
                 'optimizer': self.optimizer.state_dict(), 'lrer': self.lrer.state_dict()}
        torch.save(state, os.path.join(self.args.checkpoint_path, 'checkpoint_{0}.ckpt'.format(epoch)))

    def _load_checkpoint(self):
        checkpoint = torch.load(self.args.resume, map_location='cpu')
        found = tool.dict_value(checkpoint, 'algorithm', default='unknown')
        if found != self.NAME:
            logger.log_err('Unmatched SSL algorithm format in checkpoint => required: {0} - given: {1}\n'
                           .format(self.NAME, found))
        self.model.load_state_dict(checkpoint['model'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])      # momentum buffers + lr groups (ssl_null.py:214-216)
        self.lrer.load_state_dict(checkpoint['lrer'])                # cur_iter of the polynomial schedule
        return checkpoint['epoch']
