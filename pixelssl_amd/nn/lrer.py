"""LR scheduler factories (pixelssl/nn/lrer.py:14-179).  The iteration-based polynomial decay is the one on the hot
path: lr_t = base * (1 - it/max_it)^power, stepped once per iteration (host scalar math); the epoch schedulers are
torch's, handed the fused optimizer (they only touch `param_groups[i]['lr']`)."""
import math

from torch.optim import lr_scheduler
from torch.optim.lr_scheduler import LRScheduler

from ..utils import logger, cmd

EPOCH_LRERS = ['steplr', 'multisteplr', 'exponentiallr', 'cosineannealinglr']
ITER_LRERS = ['polynomiallr']
VALID_LRER = EPOCH_LRERS + ITER_LRERS


def add_parser_arguments(parser):
    """Same flags and the '-1 = scheduler default' convention as the reference parser (lrer.py:19-44)."""
    parser.add_argument('--last-epoch', type=int, default=-1, metavar='', help='lr scheduler - index of last epoch [all]')
    parser.add_argument('--step-size', type=int, default=-1, metavar='', help='lr scheduler - decay period in epochs [steplr]')
    parser.add_argument('--milestones', type=cmd.str2intlist, default=[], metavar='', help='lr scheduler - epoch indices [multisteplr]')
    parser.add_argument('--gamma', type=float, default=-1, metavar='', help='lr scheduler - decay factor [steplr, multisteplr, exponentiallr]')
    parser.add_argument('--T-max', type=int, default=-1, metavar='', help='lr scheduler - maximum number of epochs [cosineannealinglr]')
    parser.add_argument('--eta-min', type=float, default=-1, metavar='', help='lr scheduler - minimum learning rate [cosineannealinglr]')
    parser.add_argument('--power', type=float, default=-1, metavar='', help='lr scheduler - power [polynomiallr]')


def steplr(args):
    args.step_size = args.epochs if args.step_size == -1 else args.step_size
    args.gamma = 0.1 if args.gamma == -1 else args.gamma

    def steplr_wrapper(optimizer):
        return lr_scheduler.StepLR(optimizer, step_size=args.step_size, gamma=args.gamma, last_epoch=args.last_epoch)

    return steplr_wrapper


def multisteplr(args):
    args.milestones = list(range(1, args.epochs)) if args.milestones == [] else args.milestones
    args.gamma = 0.1 if args.gamma == -1 else args.gamma

    def multisteplr_wrapper(optimizer):
        return lr_scheduler.MultiStepLR(optimizer, milestones=args.milestones, gamma=args.gamma, last_epoch=args.last_epoch)

    return multisteplr_wrapper


def exponentiallr(args):
    args.gamma = 0.1 if args.gamma == -1 else args.gamma

    def exponentiallr_wrapper(optimizer):
        return lr_scheduler.ExponentialLR(optimizer, gamma=args.gamma, last_epoch=args.last_epoch)

    return exponentiallr_wrapper


def cosineannealinglr(args):
    args.T_max = args.epochs if args.T_max == -1 else args.T_max
    args.eta_min = 0 if args.eta_min == -1 else args.eta_min

    def cosineannealinglr_wrapper(optimizer):
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=args.T_max, eta_min=args.eta_min, last_epoch=args.last_epoch)

    return cosineannealinglr_wrapper


class PolynomialLR(LRScheduler):
    """Subclasses torch's scheduler base exactly like the reference (lrer.py:143-179), so the
    base-class constructor's initial `step()` is inherited: with the torch build of this image the
    first training iteration already runs at cur_iter = 1 (see oracle/torch_oracle.py:_lrs)."""

    def __init__(self, optimizer, epochs, iters_per_epoch, power=0.9, last_epoch=-1):
        self.epochs, self.iters_per_epoch, self.power = epochs, iters_per_epoch, power
        self.max_iters = epochs * iters_per_epoch
        self.cur_iter = 0
        self._warned = False
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        decay = (1 - float(self.cur_iter) / self.max_iters) ** self.power
        return [base * decay for base in self.base_lrs]

    def step(self, epoch=None):
        if epoch is None:
            self.cur_iter += 1
            self.last_epoch = math.floor(self.cur_iter / self.iters_per_epoch)
        elif epoch != 0:
            if not self._warned:
                logger.log_warn('PolynomialLR is designed to be stepped once per iteration; '
                                'stepping it per epoch now.\n')
                self._warned = True
            assert epoch <= self.epochs
            self.last_epoch = epoch
            self.cur_iter = epoch * self.iters_per_epoch
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr


def polynomiallr(args):
    args.power = 0.9 if args.power == -1 else args.power

    def polynomiallr_wrapper(optimizer):
        return PolynomialLR(optimizer, epochs=args.epochs, iters_per_epoch=args.iters_per_epoch,
                            power=args.power, last_epoch=args.last_epoch)

    return polynomiallr_wrapper
