"""Module names the reference exports from pixelssl.nn.module, re-expressed for one-process-per-GPU."""
import random

import torch
import torch.nn as nn

from ..engine import SynchronizedBatchNorm2d  # noqa: F401  (parameter holder; math is fused in the kernels)


def patch_replication_callback(model):
    """No-op.  The reference needs this to wire its thread-based Sync-BN into nn.DataParallel
    (sync_batchnorm/replicate.py:65-88); here every rank owns a persistent replica and the BN
    statistics are exchanged by an RCCL all-reduce inside the executor."""
    return model


class GaussianNoiseLayer(nn.Module):
    """Input noise of SSL_MT (pixelssl/nn/module/gaussian_noise.py:7-40): per sample, min-max normalise to [0, 1], add
    N(0, sigma) noise with sigma = random.uniform(0, std) (python's RNG, one draw per call), clip to [0, 1], de-normalise
    -- IN PLACE on the given tensor, like the reference.  The normal deviates come from torch's device generator (what
    `self.noise.data.normal_` draws in the reference); everything else is one min-max reduction + one fused kernel
    (csrc/flawmap.hip: pxl_gaussian_noise_apply).  std=None (every shipped script): the input is returned unchanged.
    `inject_noise(t)`: the next call uses `t` as its N(0, 1) deviates * sigma instead of drawing (parity tests)."""

    def __init__(self, std=None):
        super().__init__()
        self.std = std
        self.enable = std is not None
        self.noise = torch.zeros(0)
        self._injected = None
        self.last_sigma = None

    def inject_noise(self, unit_noise, sigma=None):
        self._injected = (unit_noise, sigma)

    def forward(self, inp):
        if not self.enable:
            return inp
        from .._lib import lib, check, ptr, stream_ptr, PixelHipError
        if not inp.is_cuda or inp.dtype != torch.float32 or not inp.is_contiguous() or inp.dim() != 4:
            raise PixelHipError('GaussianNoiseLayer runs in place on a contiguous 4-D fp32 GPU tensor (got %s %s on %s)'
                                % (tuple(inp.shape), inp.dtype, inp.device))
        sigma = random.uniform(0, self.std)                      # gaussian_noise.py:25: one python draw per call
        if self._injected is not None:
            unit, s_inj = self._injected
            self._injected = None
            sigma = sigma if s_inj is None else s_inj
            self.noise = unit.to(inp.device, torch.float32).contiguous() * sigma
        else:
            if self.noise.shape != inp.shape or self.noise.device != inp.device:
                self.noise = torch.zeros(inp.shape, device=inp.device)
            self.noise.normal_(0, std=sigma)
        self.last_sigma = sigma
        B = inp.shape[0]
        mm = torch.empty(B, 2, device=inp.device, dtype=torch.float32)
        check(lib().pxl_gaussian_noise_apply(B, inp.numel() // B, ptr(inp), ptr(self.noise), ptr(mm), stream_ptr()))
        return inp
