"""dict -> argv -> argparse plumbing with the reference's conventions (pixelssl/utils/cmd.py:10-59)."""
import re

from . import logger

cmdline_strs = None


def _flag(key):
    return ('-' if len(key) == 1 else '--') + re.sub(r'_', '-', key)


def parse_args(parser, args_dict):
    global cmdline_strs
    pairs = [(_flag(k), str(v)) for k, v in args_dict.items()]
    cmdline_strs = ['{0} = {1}'.format(k, v) for k, v in pairs]
    argv = [tok for pair in pairs for tok in pair]
    return parser.parse_args(argv)


def print_args():
    logger.log_info('Experiment args: \n  {0}\n'.format('\n  '.join(cmdline_strs or [])))


def str2bool(v):
    s = str(v).lower()
    if s in ('yes', 'true', 't', 'y', '1'):
        return True
    if s in ('no', 'false', 'f', 'n', '0'):
        return False
    logger.log_err('str2bool requires a boolean value, but got {0}\n'.format(v))


def _split(v):
    return [t.strip() for t in re.sub(r'[\[\]\(\)]', '', v).split(',') if t.strip() != '']


def str2intlist(v):
    return [int(t) for t in _split(v)]


def str2floatlist(v):
    return [float(t) for t in _split(v)]
