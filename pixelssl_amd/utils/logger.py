"""Logging + running-average meters with the reference's surface (pixelssl/utils/logger.py:14-131).
Only rank 0 prints in multi-process runs; log_err stays fatal-by-exit on every rank."""
import logging
import os
import sys

_fmt = logging.Formatter('%(message)s')
logging.basicConfig(level=logging.INFO, format='%(message)s')
logger = logging.getLogger('PixelSSL-AMD')


def _is_rank0():
    return int(os.environ.get('RANK', '0')) == 0


def log_mode(debug=False):
    logger.setLevel(logging.DEBUG if debug else logging.INFO)


def log_file(fpath, debug=False):
    if not _is_rank0():
        return
    fh = logging.FileHandler(fpath)
    fh.setLevel(logging.DEBUG if debug else logging.INFO)
    fh.setFormatter(_fmt)
    logger.addHandler(fh)


def _text(message):
    return ''.join(message) if isinstance(message, list) else message


def log_info(message):
    if _is_rank0():
        logger.info(_text(message))


def _banner(tag, message):
    bar = '=' * ((78 - len(tag) - 2) // 2)
    return '\n{0} {1} {0}\n{2}{3}\n'.format(bar, tag, _text(message), '=' * 78)


def log_warn(message):
    if _is_rank0():
        logger.warning(_banner('WARN', message))


def log_err(message):
    logger.error(_banner('ERROR', message))
    sys.exit(1)


class AvgMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __format__(self, spec):
        return '{0:{2}} ({1:{2}})'.format(self.val, self.avg, spec)


class AvgMeterSet:
    def __init__(self):
        self.meters = {}

    def __getitem__(self, key):
        return self.meters[key]

    def keys(self):
        return self.meters.keys()

    def has_key(self, key):
        return key in self.meters

    def update(self, name, value, n=1):
        self.meters.setdefault(name, AvgMeter()).update(value, n)

    def reset(self, name=None):
        if name is None:
            for m in self.meters.values():
                m.reset()
        elif name in self.meters:
            self.meters[name].reset()
        else:
            log_err('Unknown key value for AvgMeterSet: {0}\n'.format(name))

    def _collect(self, attr, postfix):
        return {k + postfix: getattr(m, attr) for k, m in self.meters.items()}

    def values(self, postfix=''):
        return self._collect('val', postfix)

    def averages(self, postfix='/avg'):
        return self._collect('avg', postfix)

    def sums(self, postfix='/sum'):
        return self._collect('sum', postfix)

    def counts(self, postfix='/count'):
        return self._collect('count', postfix)
