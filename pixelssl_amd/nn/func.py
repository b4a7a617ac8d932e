"""pixelssl.nn.func surface: create_model / split_tensor_tuple / sigmoid_rampup / model_str."""
import math

import torch
import torch.nn as nn

from ..utils import logger
from .. import dist as pdist


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t/T)^2) ramp (pixelssl/nn/func.py:12-20)."""
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length))
    phase = 1.0 - t / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def split_tensor_tuple(ttuple, start, end, reduce_dim=False):
    """Slice every tensor of a tuple along dim 0 (labeled-first batch layout, nn/data.py:148-159)."""
    if reduce_dim and end - start != 1:
        raise AssertionError('reduce_dim requires end - start == 1')
    if reduce_dim:
        return tuple(t[start, ...] for t in ttuple)
    return tuple(t[start:end, ...] for t in ttuple)


class RankModel(nn.Module):
    """What create_model returns: the per-rank replica behind the `.module` attribute the SSL
    algorithms reach through (ssl_null.py:65) -- so state_dict keys keep the reference's `module.`
    prefix -- plus the distributed wiring that replaces nn.DataParallel."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def create_model(mclass, mname, **kwargs):
    model = RankModel(mclass(**kwargs))
    pdist.attach(model)
    logger.log_info('  {0}: {1:,d} parameters'.format(mname, sum(p.numel() for p in model.parameters())))
    return model


def model_str(module):
    total = 0
    lines = ['  ' + '-' * 76]
    for name, p in module.named_parameters():
        total += p.numel()
        lines.append('  {0:<40} {1:>20} = {2:>12,d}'.format(name, ' * '.join(str(s) for s in p.size()), p.numel()))
    lines += ['  ' + '-' * 76, '  {0:<40} {1:>20} = {2:>12,d}'.format('all parameters', 'sum of above', total),
              '  ' + '=' * 76, '']
    return '\n'.join(lines)


def pytorch_support(required_version='1.0.0', info_str=''):
    def key(v):
        return tuple(int(x) for x in v.split('+')[0].split('.')[:3] if x.isdigit())
    if key(torch.__version__) < key(required_version):
        logger.log_err('{0} required PyTorch >= {1}\nHowever, current PyTorch == {2}\n'
                       .format(info_str, required_version, torch.__version__))
    return True
