"""Pixel-wise Mean Teacher (pixelssl/ssl_algorithm/ssl_mt.py): student + EMA teacher of the same task
model; CE on the labeled slice, MSE consistency between student and (detached) teacher LOGITS,
scaled by cons_scale * sigmoid ramp-up; teacher runs under no_grad with train-mode BN."""
import os
import time

import torch

from ..utils import REGRESSION, CLASSIFICATION, logger, cmd, tool
from ..nn import func
from ..nn.module import patch_replication_callback, GaussianNoiseLayer
from ..functional import MSELoss
from .. import ops, streams
from .. import graph as pgraph
from . import ssl_base


def add_parser_arguments(parser):
    ssl_base.add_parser_arguments(parser)
    parser.add_argument('--cons-for-labeled', type=cmd.str2bool, default=True,
                        help='sslmt - consistency constraint on the labeled data too if True')
    parser.add_argument('--cons-scale', type=float, default=-1, help='sslmt - consistency constraint coefficient')
    parser.add_argument('--cons-rampup-epochs', type=int, default=-1, help='sslmt - ramp-up epochs of the constraint')
    parser.add_argument('--ema-decay', type=float, default=0.999, help='sslmt - EMA coefficient of the teacher')
    parser.add_argument('--gaussian-noise-std', type=float, default=None, help='sslmt - std of the input noise')


def ssl_mt(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    mf, of, lf, cf = ssl_base._SSLBase._single_component('ssl_mt', model_dict, optimizer_dict, lrer_dict,
                                                         criterion_dict)
    algorithm = SSLMT(args)
    algorithm.build([mf], [of], [lf], [cf], task_func)
    return algorithm


class SSLMT(ssl_base._SSLBase):
    NAME = 'ssl_mt'
    SUPPORTED_TASK_TYPES = [REGRESSION, CLASSIFICATION]

    def __init__(self, args):
        super().__init__(args)
        self.s_model = self.t_model = None
        self.s_optimizer = self.s_lrer = self.s_criterion = self.cons_criterion = None
        self.gaussian_noiser = None
        # execution modes, fixed when the algorithm is built (pixelssl_amd/graph.py, nn/optimizer.py: PipelinedUpdate)
        self._want_graph = pgraph.enabled()
        # 'auto' (default): the fused parameter update where its kernel applies (bf16 engine); '1' / '0' force the hook on / off
        self._want_pipe = os.environ.get('PXL_PIPE_UPDATE', 'auto')
        if self.args.cons_for_labeled or self.args.unlabeled_batch_size > 0:
            if self.args.cons_scale < 0:
                logger.log_err('The argument - cons_scale - is not set (or invalid)\n'
                               'Please set - cons_scale >= 0 - for training\n')
            if self.args.cons_rampup_epochs < 0:
                logger.log_err('The argument - cons_rampup_epochs - is not set (or invalid)\n'
                               'Please set - cons_rampup_epochs >= 0 - for training\n')

    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self.task_func = task_func
        self.s_model = patch_replication_callback(func.create_model(model_funcs[0], 's_model', args=self.args))
        self.t_model = patch_replication_callback(func.create_model(model_funcs[0], 't_model', args=self.args))
        for param in self.t_model.parameters():      # the teacher is never trained by back-propagation
            param.detach_()
        self.models = {'s_model': self.s_model, 't_model': self.t_model}
        self.s_optimizer = optimizer_funcs[0](self.s_model.module.param_groups)
        self.optimizers = {'s_optimizer': self.s_optimizer}
        self.s_lrer = lrer_funcs[0](self.s_optimizer)
        self.lrers = {'s_lrer': self.s_lrer}
        self.cons_criterion = MSELoss()
        self.s_criterion = criterion_funcs[0](self.args)
        self.criterions = {'s_criterion': self.s_criterion, 'cons_criterion': self.cons_criterion}
        self.gaussian_noiser = GaussianNoiseLayer(self.args.gaussian_noise_std)

    def _teacher_stream(self):
        if not hasattr(self, '_t_stream'):
            on = os.environ.get('PXL_TEACHER_STREAM', '1') != '0' and torch.cuda.is_available()
            # PXL_TEACHER_PRIO=-1: high priority (the student waits for the teacher's logits before the consistency loss;
            # in the traces the teacher pass ends 0.5 ms after the student's when both have the same priority)
            prio = int(os.environ.get('PXL_TEACHER_PRIO', '0'))
            if prio != 0:
                self._t_stream = torch.cuda.Stream(priority=prio) if on else None
            else:           # the placement pool's stream for a second network: a hardware queue that is not the main stream's
                self._t_stream = streams.role_stream(streams.SIDE) if on else None
        return self._t_stream

    def train_step(self, inp, gt, cur_step, total_rampup_steps):
        """One iteration of ssl_mt.py:131-220 on device-resident tuples.
        Returns dict(s_task_loss, t_task_loss, cons_loss) of detached device scalars."""
        lbs = self.args.labeled_batch_size
        if self.gaussian_noiser.enable:
            # ssl_mt.py:340-348: the batch is uploaded twice and each copy of the first input gets its own noise draw
            # (student first); the layer works in place, so it is handed copies
            s_inp = tuple(self.gaussian_noiser(i.clone()) if k == 0 else i for k, i in enumerate(inp))
            t_inp = tuple(self.gaussian_noiser(i.clone()) if k == 0 else i for k, i in enumerate(inp))
        else:
            s_inp = t_inp = tuple(inp)     # noise disabled => one tensor serves both passes
        ramp = func.sigmoid_rampup(cur_step, total_rampup_steps)

        seam = lbs > 0 and type(self.cons_criterion) is MSELoss and \
            self._seam_fusable([self.s_model, self.t_model], self.s_criterion, s_inp, gt)
        if seam and s_inp is t_inp and not getattr(getattr(self.s_model.module, 'model', None), '_profile_on', False):
            sg = self._step_graph(lbs)
            if sg is not None:          # the iteration as ONE hipGraph launch (pixelssl_amd/graph.py)
                outs = sg.step((s_inp[0], gt[0]), (cur_step, total_rampup_steps))
                return (dict(s_task_loss=outs[0], t_task_loss=outs[1], cons_loss=outs[2]),) + tuple(self._graph_resulters)
        self.s_optimizer.zero_grad()
        l_gt = func.split_tensor_tuple(gt, 0, lbs)
        if seam:
            return self._train_step_fused_seam(s_inp, t_inp, l_gt, lbs, ramp, cur_step)

        def teacher_pass():
            with torch.no_grad():
                t_res, _ = self.t_model.forward(t_inp)
                if 'pred' not in t_res.keys():
                    self._need_pred(t_res, 'SSL_MT')
                t_prd = tool.dict_value(t_res, 'pred')
                t_loss = torch.mean(self.s_criterion.forward(func.split_tensor_tuple(t_prd, 0, lbs), l_gt,
                                                             func.split_tensor_tuple(t_inp, 0, lbs)))
            return t_res, t_prd, t_loss

        # The teacher's no-grad forward does not depend on the student's: it is enqueued FIRST, on a second HIP
        # stream, and runs concurrently with the student forward (each network's kernels are ~1 workgroup per CU
        # and latency-bound, two in flight fill each other's bubbles).  PXL_TEACHER_STREAM=0 keeps one stream.
        side = self._teacher_stream()
        fut = None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)          # inputs and the EMA-updated teacher weights are ready
            dev = torch.cuda.current_device()

            def on_side():
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    return teacher_pass()
            pool = self._enqueue_worker()
            if pool is not None:            # a helper thread enqueues the teacher while this one enqueues the student
                fut = pool.submit(on_side)
            else:
                t_resulter, t_pred, t_task_loss = on_side()
        s_resulter, _ = self.s_model.forward(s_inp)
        self._need_pred(s_resulter, 'SSL_MT')
        s_pred = tool.dict_value(s_resulter, 'pred')
        # fused task + consistency gradient (one launch writes d(pred) once) when the criterion pair allows it:
        # CommonSSEGCriterion.with_consistency + the engine's MSELoss, a single `pred` tensor.  PXL_FUSE_MT_LOSS=0 disables
        cons_range = (0, s_pred[0].shape[0]) if self.args.cons_for_labeled else \
            ((lbs, s_pred[0].shape[0]) if self.args.unlabeled_batch_size > 0 else None)
        fuse = (cons_range is not None and len(s_pred) == 1 and hasattr(self.s_criterion, 'with_consistency')
                and type(self.cons_criterion) is MSELoss and os.environ.get('PXL_FUSE_MT_LOSS', '1') != '0')
        if fuse:
            with torch.no_grad():       # launched now: overlaps the tail of the teacher pass
                ce_values = self.s_criterion.forward(func.split_tensor_tuple(s_pred, 0, lbs), l_gt,
                                                     func.split_tensor_tuple(s_inp, 0, lbs))
        else:
            s_task_loss = torch.mean(self.s_criterion.forward(func.split_tensor_tuple(s_pred, 0, lbs), l_gt,
                                                              func.split_tensor_tuple(s_inp, 0, lbs)))
        if side is not None:
            if fut is not None:
                t_resulter, t_pred, t_task_loss = fut.result()
            main.wait_stream(side)
            outs = [t_task_loss]
            for v in t_resulter.values():
                outs += [t for t in (v if isinstance(v, (tuple, list)) else (v,)) if torch.is_tensor(t)]
            for t in outs:
                t.record_stream(main)       # allocated on the side stream, consumed on the main one
        else:
            t_resulter, t_pred, t_task_loss = teacher_pass()
        t_pseudo_gt = t_pred[0].detach()
        if fuse:
            ce, cons_loss = self.s_criterion.with_consistency(s_pred, l_gt, ce_values, t_pseudo_gt, *cons_range)
            s_task_loss = torch.mean(ce)
        elif self.args.cons_for_labeled:
            cons_loss = self.cons_criterion(s_pred[0], t_pseudo_gt)
        elif self.args.unlabeled_batch_size > 0:
            cons_loss = self.cons_criterion(s_pred[0][lbs:, ...], t_pseudo_gt[lbs:, ...])
        else:
            cons_loss = torch.zeros((), device=s_pred[0].device)
        cons_loss = ramp * self.args.cons_scale * torch.mean(cons_loss)

        loss = s_task_loss + cons_loss
        loss.backward()
        self.s_optimizer.step()
        self._update_ema_variables(self.s_model, self.t_model, self.args.ema_decay, cur_step)
        if not self.args.is_epoch_lrer:
            self.s_lrer.step()
        return dict(s_task_loss=s_task_loss.detach(), t_task_loss=t_task_loss.detach(),
                    cons_loss=cons_loss.detach()), s_resulter, t_resulter

    def _update_pipeline(self, s_head, t_head):
        """PipelinedUpdate for this student / teacher pair, or None: switched off (PXL_PIPE_UPDATE=0), an optimizer it does not cover
        (anything but plain momentum SGD over the student's flat parameter store), task models that are not engine networks, or
        -- in the default 'auto' setting -- an engine dtype the fused update kernel does not serve (fp32).
        Measured on one MI355X (MT 8 x 513 x 513, pairs of runs inside one call):
          * separate SGD / EMA / pack kernels per bucket (PXL_FUSED_UPDATE=0): 12.37 / 12.35 ms against 12.37 / 12.43 ms without --
            neutral: the 0.7 ms of update work move under the backward pass and the backward pass gets 0.5 ms longer (those kernels
            stream 2.2 GB at > 5 TB/s next to bandwidth-bound data gradients);
          * ONE fused kernel (SGD + EMA + both bf16 forward copies + gradient memset, 32 instead of 48 bytes per parameter):
            11.94 / 11.99 ms in 16 MB buckets, 11.93 / 11.96 ms as one bucket at the end of the backward pass, against
            12.12 / 12.13 ms -- fewer bytes and four launches less is what pays, not the overlap.  Hence the default: fused, one
            bucket (PXL_UPDATE_BUCKET_MB), multi-rank runs bucket with the gradient exchange."""
        if not hasattr(self, '_pipe'):
            self._pipe = None
            # (multi-rank: the hook runs on the communication stream right behind each bucket's all-reduce + 1/world scaling, csrc/
            # net.cpp `flush` -- the fused kernel consumes the AVERAGED bucket; pinned bit-identical to the separate kernels on two
            # ranks in tests/test_gpu_dist.py.  A backward whose hook does not fire takes the ordinary step, see FusedSGD.step)
            want = self._want_pipe == '1' or (self._want_pipe == 'auto' and os.environ.get('PXL_FUSED_UPDATE', '1') == '1' and
                                              getattr(s_head.core, '_code', None) == 1)
            if want:
                from ..nn.optimizer import PipelinedUpdate
                s_core, t_core = s_head.core, t_head.core
                try:
                    if len(list(self.s_model.parameters())) != len(s_core._param_list):
                        raise ValueError('the student holds parameters outside its engine network')
                    self._pipe = PipelinedUpdate(self.s_optimizer, s_core, t_core)
                except ValueError as e:
                    logger.log_info('pipelined parameter update off: %s\n' % e)
        return self._pipe

    def _step_graph(self, lbs):
        """The StepGraph of the fused-seam iteration, or None (PXL_GRAPH=0, several ranks, an optimizer / scheduler the captured
        step does not cover).  Captured body = _train_step_fused_seam in hyper mode: the ramped consistency weight, the learning
        rates and the EMA coefficient are read from device memory (graph.HyperBlock), everything else is the eager step."""
        if hasattr(self, '_sgraph'):
            return self._sgraph
        from .. import dist as pdist
        from ..nn.optimizer import FusedSGD
        self._sgraph = None
        ok = self._want_graph and not pdist.is_distributed() and type(self.s_optimizer) is FusedSGD and \
            os.environ.get('PXL_PAIR_FORWARD') != '1' and not any(self.s_optimizer._foreign) and \
            all(float(g.get('dampening', 0.0)) == 0.0 and not g.get('nesterov', False) for g in self.s_optimizer.param_groups)
        if not ok:
            return None

        def body(x, g0):
            self.s_optimizer.zero_grad()
            losses, s_res, t_res = self._train_step_fused_seam((x,), (x,), (g0[:lbs],), lbs, None, None)
            self._graph_resulters = (s_res, t_res)
            return losses['s_task_loss'], losses['t_task_loss'], losses['cons_loss']

        def scalars(cur_step, total_rampup_steps):
            vals = dict(self.s_optimizer.hyper_values())
            vals['ema_alpha'] = min(1 - 1 / (cur_step + 1), self.args.ema_decay)
            vals['w_cons'] = func.sigmoid_rampup(cur_step, total_rampup_steps) * self.args.cons_scale
            return vals

        def after_replay():          # the host half of optimizer.step / EMA / scheduler that the body did while being recorded
            if getattr(self, '_pipe', None) is not None:
                self._pipe.replayed()
            else:
                self.s_optimizer.after_replayed_step()
                for m_ in (self.t_model,):
                    core_ = getattr(m_.module, 'model', None)
                    if core_ is not None and hasattr(core_, 'mark_params_changed'):
                        core_.mark_params_changed()
            if not self.args.is_epoch_lrer:
                self.s_lrer.step()

        host = pgraph.HostState([self.s_lrer, self.s_optimizer, getattr(self, '_pipe', None)], self.s_optimizer.param_groups)
        self._sgraph = pgraph.StepGraph(body, scalars, after_replay, torch.device('cuda', torch.cuda.current_device()), host_state=host)
        return self._sgraph

    def _train_step_fused_seam(self, s_inp, t_inp, l_gt, lbs, ramp, cur_step):
        """The same iteration with the seam between the two forward passes and the backward pass fused
        (functional.head_losses): both networks stop at their low-resolution logits, one kernel evaluates the student's
        and the teacher's task loss, the consistency term and d(loss)/d(student logits) per full-resolution row, and the
        executor's backward starts from that gradient.  No 8 x 21 x 513 x 513 plane (logits, soft-max, their gradients:
        177 MB each) is written; the loss values and the weight update are those of the generic path (same per-pixel
        expressions, tests/test_seam.py)."""
        from .. import functional as PF
        from ..sseg.model import _DeferredResulter
        # Student and teacher run the SAME program on the same shape: one lockstep executor pass can issue every convolution
        # (and finalize-folding element-wise kernel) of the pair as ONE launch (engine.forward_deferred_pair; twice the tiles
        # per launch, half the launches, one enqueue thread and one stream).  Measured on one MI355X (profiles/r04_pair_ab.txt):
        # the paired convolutions take 0.79x the time of two single launches, but the lockstep pass has nothing left to overlap
        # the memory-bound joins with -- 12.8 ms / step against 12.4 ms for the two passes on two streams.  So: two streams on
        # one rank; the paired pass with Sync-BN (multi-rank), where it puts every rank's statistic exchanges into ONE
        # program order on ONE stream (no cross-queue spin hazard between the two networks' exchange kernels, no idle teacher
        # stream while the host enqueues the student).  PXL_PAIR_FORWARD=1 / 0 forces it on / off.
        from .. import dist as pdist
        pair_mode = os.environ.get('PXL_PAIR_FORWARD')
        if pair_mode == '1' or (pair_mode is None and pdist.is_distributed()):
            from ..engine import forward_deferred_pair
            s_core, t_core = getattr(self.s_model.module, 'model', None), getattr(self.t_model.module, 'model', None)
            pair = None
            if s_core is not None and t_core is not None and len(s_inp) == 1 and len(t_inp) == 1:
                # (the teacher's parameters are detached -- ssl_mt.py:99-101 -- so its half of the pair keeps no graph)
                teacher_needs_grad = any(p_.requires_grad for p_ in getattr(t_core, '_param_list', [None])[:1] if p_ is not None)
                if hasattr(s_core, '_pb') and hasattr(t_core, '_pb') and not teacher_needs_grad:
                    pair = forward_deferred_pair(s_core, s_inp[0], t_core, t_inp[0])
            if pair is not None:
                s_head, t_head = pair
                return self._seam_losses_and_update(s_head, t_head, s_inp, l_gt, lbs, ramp, cur_step)
        side = self._teacher_stream()
        if side is not None:
            # the two forward passes run the same launches side by side: their tiles are tuned that way (engine: tune_dual)
            for m_ in (self.s_model, self.t_model):
                core_ = getattr(m_.module, 'model', None)
                if core_ is not None and hasattr(core_, '_pb') and not getattr(core_, 'tune_dual', False):
                    core_.tune_dual = True

        # ONE input tensor for both networks (no input noise): the student writes the stem's im2col patches once, ahead of both
        # passes on the main stream, and the teacher's stem reads them (203 MB less to write, one 115 us kernel instead of two
        # that share the chip for 220 us at the start of every iteration).  PXL_SHARE_PATCHES=0: every network its own.
        shared = None
        if s_inp is t_inp and len(s_inp) == 1 and os.environ.get('PXL_SHARE_PATCHES', '1') != '0' and \
                hasattr(self.s_model.module, 'forward_deferred_shared') and hasattr(getattr(self.s_model.module, 'model', None), 'prepare_patches'):
            shared = self.s_model.module.model.prepare_patches(s_inp[0])
            if shared is not None and side is not None:
                shared[0].record_stream(side)          # (the teacher's stem reads the student's arena on its own stream)

        def teacher_pass():
            with torch.no_grad():
                if shared is not None:
                    return self.t_model.module.forward_deferred_shared(t_inp, borrow=shared)
                return self.t_model.module.forward_deferred(t_inp)
        fut = None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            dev = torch.cuda.current_device()

            def on_side():
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    return teacher_pass()
            pool = self._enqueue_worker()
            if pool is not None:
                fut = pool.submit(on_side)
            else:
                t_head = on_side()
        s_head = self.s_model.module.forward_deferred_shared(s_inp, prepared=shared) if shared is not None else \
            self.s_model.module.forward_deferred(s_inp)
        if side is not None:
            if fut is not None:
                t_head = fut.result()
            main.wait_stream(side)
            t_head.arena.record_stream(main)
        else:
            t_head = teacher_pass()
        return self._seam_losses_and_update(s_head, t_head, s_inp, l_gt, lbs, ramp, cur_step)

    def _seam_losses_and_update(self, s_head, t_head, s_inp, l_gt, lbs, ramp, cur_step):
        from .. import functional as PF
        from ..sseg.model import _DeferredResulter
        B = s_inp[0].shape[0]
        lo, hi = (0, B) if self.args.cons_for_labeled else ((lbs, B) if self.args.unlabeled_batch_size > 0 else (0, 0))
        hyper = pgraph.current_hyper()
        if hyper is not None:       # captured step: the ramped weight is a device scalar (uploaded by StepGraph before the launch)
            ce_s, ce_t, mse = PF.head_losses(s_head, t_head, l_gt[0], lbs, lo, hi, 1.0 / lbs, None, self.args.ignore_index,
                                             mse_weight_dev=hyper.ptr('w_cons'))
            w_cons = hyper.tensor('w_cons')
        else:
            w_cons = ramp * self.args.cons_scale
            ce_s, ce_t, mse = PF.head_losses(s_head, t_head, l_gt[0], lbs, lo, hi, 1.0 / lbs, w_cons, self.args.ignore_index)
        s_task_loss, t_task_loss = torch.mean(ce_s), torch.mean(ce_t)
        cons_loss = w_cons * mse
        pipe = self._update_pipeline(s_head, t_head)
        if pipe is not None:
            # SGD + EMA + weight re-packing bucket by bucket from inside the backward pass (nn/optimizer.py: PipelinedUpdate)
            pipe.arm(s_head.plan, t_head.plan, None if hyper is not None else min(1 - 1 / (cur_step + 1), self.args.ema_decay), hyper)
        s_head.backward()
        self.s_optimizer.step()
        # the EMA update belongs to the buckets only when they RAN (FusedSGD.step reports it): an armed pipeline whose hook never
        # fired -- weight gradients off, a program without monotonic parameter offsets, a partial multi-rank range -- took the
        # ordinary step, and the teacher must not be skipped for it
        if pipe is None or not getattr(self.s_optimizer, 'last_step_pipelined', False):
            self._update_ema_variables(self.s_model, self.t_model, self.args.ema_decay, cur_step)
        if not self.args.is_epoch_lrer:
            self.s_lrer.step()
        return dict(s_task_loss=s_task_loss.detach(), t_task_loss=t_task_loss.detach(),
                    cons_loss=cons_loss.detach()), _DeferredResulter(s_head), _DeferredResulter(t_head)

    def _train(self, data_loader, epoch):
        self.meters.reset()
        self.s_model.train()
        self.t_model.train()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            cur_step = len(data_loader) * epoch + idx
            losses, _, _ = self.train_step(inp, gt, cur_step, len(data_loader) * self.args.cons_rampup_epochs)
            for k, v in losses.items():
                self.meters.update(k, v)
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  student-{4}\t=>\ts-task-loss: {5:.6f}\ts-cons-loss: {6:.6f}\n'
                                '  teacher-{4}\t=>\tt-task-loss: {7:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['s_task_loss'].avg), float(self.meters['cons_loss'].avg),
                                        float(self.meters['t_task_loss'].avg)))
        if self.args.is_epoch_lrer:
            self.s_lrer.step()

    @torch.no_grad()
    def _validate(self, data_loader, epoch):
        """ssl_mt.py:226-294: student and teacher in eval mode, task losses, the consistency loss and the metrics of both."""
        self.meters.reset()
        self.s_model.eval()
        self.t_model.eval()
        for idx, (inp, gt) in enumerate(data_loader):
            timer = time.time()
            inp, gt = self._to_device(inp), self._to_device(gt)
            preds = {}
            for tag, model in (('student', self.s_model), ('teacher', self.t_model)):
                resulter, _ = model.forward(inp)
                self._need_pred(resulter, 'SSL_MT')
                pred = tool.dict_value(resulter, 'pred')
                preds[tag] = (pred, tool.dict_value(resulter, 'activated_pred'))
                self.meters.update(tag[0] + '_task_loss', torch.mean(self.s_criterion.forward(pred, gt, inp)).detach())
            cons_loss = self.args.cons_scale * torch.mean(self.cons_criterion(preds['student'][0][0], preds['teacher'][0][0].detach()))
            self.meters.update('cons_loss', cons_loss.detach())
            self.task_func.metrics(preds['student'][1], gt, inp, self.meters, id_str='student')
            self.task_func.metrics(preds['teacher'][1], gt, inp, self.meters, id_str='teacher')
            self.meters.update('batch_time', time.time() - timer)
            if idx % self.args.log_freq == 0:
                logger.log_info('step: [{0}][{1}/{2}]\tbatch-time: {3:.3f}\n'
                                '  student-{4}\t=>\ts-task-loss: {5:.6f}\ts-cons-loss: {6:.6f}\n'
                                '  teacher-{4}\t=>\tt-task-loss: {7:.6f}\n'
                                .format(epoch + 1, idx, len(data_loader), self.meters['batch_time'].avg, self.args.task,
                                        float(self.meters['s_task_loss'].avg), float(self.meters['cons_loss'].avg),
                                        float(self.meters['t_task_loss'].avg)))
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
        self.s_optimizer.load_state_dict(checkpoint['s_optimizer'])  # ssl_mt.py:318-320
        self.s_lrer.load_state_dict(checkpoint['s_lrer'])
        return checkpoint['epoch']

    def _update_ema_variables(self, s_model, t_model, ema_decay, cur_step):
        """alpha = min(1 - 1/(step+1), decay); parameters only, BN buffers evolve by the teacher's own
        forward (ssl_mt.py:359-363).  Engine task models: one fused launch over the flat parameter buffers; any other
        TaskModel (a plugin's torch model): the reference's per-parameter walk."""
        hyper = pgraph.current_hyper()
        alpha = min(1 - 1 / (cur_step + 1), ema_decay) if hyper is None else None
        s_core, t_core = getattr(s_model.module, 'model', None), getattr(t_model.module, 'model', None)
        if hasattr(s_core, 'flat') and hasattr(t_core, 'flat') and s_core.flat.np == t_core.flat.np and \
                len(list(s_model.parameters())) == len(s_core._param_list):
            ops.ema_update(t_core.flat.params, s_core.flat.params, alpha,
                           alpha_dev=hyper.ptr('ema_alpha') if hyper is not None else None)
            t_core.mark_params_changed()
            return
        if hyper is not None:
            raise RuntimeError('a captured step needs engine task models (flat parameter stores) for the EMA update')
        with torch.no_grad():
            for t_param, s_param in zip(t_model.parameters(), s_model.parameters()):
                t_param.mul_(alpha).add_(s_param.detach(), alpha=1 - alpha)
        for core in (m for m in t_model.modules() if hasattr(m, 'mark_params_changed')):
            core.mark_params_changed()
