"""LR scheduler factories (pixelssl/nn/lrer.py).  Only the iteration-based polynomial decay is on the
hot path: lr_t = base * (1 - it/max_it)^power, stepped once per iteration (host scalar math)."""
import math

from torch.optim.lr_scheduler import LRScheduler

from ..utils import logger

EPOCH_LRERS = []
ITER_LRERS = ['polynomiallr']
VALID_LRER = EPOCH_LRERS + ITER_LRERS


def add_parser_arguments(parser):
    parser.add_argument('--last-epoch', type=int, default=-1, metavar='', help='lr scheduler - index of last epoch')
    parser.add_argument('--power', type=float, default=-1, metavar='', help='lr scheduler - power (polynomiallr)')


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
