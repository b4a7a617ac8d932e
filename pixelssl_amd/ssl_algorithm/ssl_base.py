"""Plugin ABI of the SSL algorithms, kept as in pixelssl/ssl_algorithm/ssl_base.py:19-159:
an export function named like the module, and a class with NAME / SUPPORTED_TASK_TYPES,
build / train / validate / save_checkpoint / load_checkpoint and the public dicts
models / optimizers / lrers / criterions / meters."""
import torch

from ..utils import logger


def add_parser_arguments(parser):
    pass


def ssl_base(args, model_dict, optimizer_dict, lrer_dict, criterion_dict, task_func):
    raise NotImplementedError


class _SSLBase:
    NAME = 'ssl_base'
    SUPPORTED_TASK_TYPES = []

    def __init__(self, args):
        self.args = args
        self.task_func = None
        self.meters = logger.AvgMeterSet()
        self.models, self.optimizers, self.lrers, self.criterions = {}, {}, {}, {}

    # interface used by the task proxy
    def build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        self._build(model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func)

    def train(self, data_loader, epoch):
        # multi-rank: the ranks enter an epoch together -- a rank-0 checkpoint save or an uneven validation pass before it must not
        # eat into the time-out of the first Sync-BN exchange (an explicit collective at a boundary every rank reaches; no-op on one)
        from .. import dist as pdist
        pdist.epoch_barrier()
        self._train(data_loader, epoch)
        # multi-rank: an epoch whose peer-mapped Sync-BN exchanges timed out trained on invalid statistics -- fail here, loudly
        # (one device synchronisation per epoch; no-op on one rank)
        pdist.check_peers()

    def validate(self, data_loader, epoch):
        self._validate(data_loader, epoch)

    def save_checkpoint(self, epoch):
        self._save_checkpoint(epoch)

    def load_checkpoint(self):
        return self._load_checkpoint()

    # to be provided by each algorithm
    def _build(self, model_funcs, optimizer_funcs, lrer_funcs, criterion_funcs, task_func):
        raise NotImplementedError

    def _train(self, data_loader, epoch):
        raise NotImplementedError

    def _validate(self, data_loader, epoch):
        raise NotImplementedError

    def _save_checkpoint(self, epoch):
        raise NotImplementedError

    def _load_checkpoint(self):
        raise NotImplementedError

    # shared helpers -------------------------------------------------------------------------
    def _enqueue_worker(self):
        """One helper thread that ENQUEUES an independent network pass on its own HIP stream while the caller's thread
        enqueues the pass on the critical path.  An executor pass is one C call issuing ~220 launches (~1 ms of host
        time); issued back to back from one thread, the second network's stream sits idle for that long at every step
        (measured: a 1 - 3.7 ms hole at the head of each MT step).  ctypes drops the GIL for the call, the HIP runtime
        is thread-safe, torch's current stream is thread-local.  PXL_ENQUEUE_THREAD=0 turns it off.
        Single-rank only: with Sync-BN the two passes would issue collectives from two threads, whose interleaving differs
        from rank to rank (a torch.distributed group is not thread-safe; two RCCL communicators can dead-lock on it)."""
        import os
        from .. import dist as pdist
        from .. import graph as pgraph
        if pgraph.current_hyper() is not None:     # a step that is (about to be) captured is enqueued by ONE thread
            return None
        # the first iterations autotune (tile timings of one network must not be measured under the other network's
        # kernels, and the tuner is entered from one thread only): the helper thread starts with the third call
        self._enq_calls = getattr(self, '_enq_calls', 0) + 1
        if self._enq_calls <= 2:
            return None
        if not hasattr(self, '_enq_pool'):
            on = os.environ.get('PXL_ENQUEUE_THREAD', '1') != '0' and not pdist.is_distributed()
            if on:
                from concurrent.futures import ThreadPoolExecutor
                self._enq_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='pxl-enqueue')
            else:
                self._enq_pool = None
        return self._enq_pool

    @staticmethod
    def _seam_fusable(task_models, criterion, inp, gt):
        """True when the fused training seam (functional.head_losses on deferred forward passes) computes exactly what
        the generic path would: sseg task models of this engine, the sseg criterion of this engine, one input / one
        ground-truth tensor, gradients enabled, and a plan whose rows fit the kernel.  PXL_FUSE_SEAM=0 turns it off."""
        import os
        from ..sseg.criterion import CommonSSEGCriterion
        if os.environ.get('PXL_FUSE_SEAM', '1') == '0' or not torch.is_grad_enabled():
            return False
        if type(criterion) is not CommonSSEGCriterion or len(inp) != 1 or len(gt) != 1 or not inp[0].is_cuda:
            return False
        for m in task_models:
            tm = getattr(m, 'module', None)
            core = getattr(tm, 'model', None)
            if not hasattr(tm, 'forward_deferred') or not hasattr(core, 'seam_supported') or not core.seam_supported(inp[0]):
                return False
        return True

    @staticmethod
    def _single_component(name, *dicts):
        if not all(len(d) == 1 for d in dicts):
            logger.log_err('The len(element_dict) of {0} should be 1\n'.format(name.upper()))
        if list(dicts[0].keys())[0] != 'model':
            logger.log_err("In {0}, the key of element_dict should be 'model',\nbut '{1}' is given\n"
                           .format(name.upper(), list(dicts[0].keys())))
        return [d['model'] for d in dicts]

    @staticmethod
    def _to_device(tensors):
        """`Variable(i).cuda()` of every _batch_prehandle: non-blocking H2D onto this rank's GPU."""
        import torch
        dev = torch.device('cuda', torch.cuda.current_device())
        return tuple(t.to(dev, non_blocking=True) for t in tensors)

    def _log_validation_metrics(self, id_strs):
        """The 'Validation metrics' banner every reference _validate ends with (e.g. ssl_mt.py:285-294): all meters
        whose key contains TaskFunc.METRIC_STR, grouped by the id_str prefix they were recorded under."""
        info = {k: '' for k in id_strs}
        for key in sorted(list(self.meters.keys())):
            if self.task_func.METRIC_STR in key:
                for id_str in info:
                    if key.startswith(id_str):
                        info[id_str] += '{0}: {1:.6}\t'.format(key, self.meters[key])
        logger.log_info('Validation metrics:\n' + ''.join(
            '  {0}-metrics\t=>\t{1}\n'.format(k, v.replace('_', '-')) for k, v in info.items()))
        return info

    @staticmethod
    def _need_pred(resulter, name):
        if 'pred' not in resulter.keys() or 'activated_pred' not in resulter.keys():
            logger.log_err("In {0}, the 'resulter' dict returned by the task model should contain:\n"
                           "   (1) 'pred'\t=>\tunactivated task predictions\n"
                           "   (2) 'activated_pred'\t=>\tactivated task predictions\n".format(name))
