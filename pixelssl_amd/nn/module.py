"""Module names the reference exports from pixelssl.nn.module, re-expressed for one-process-per-GPU."""
import torch.nn as nn

from ..engine import SynchronizedBatchNorm2d  # noqa: F401  (parameter holder; math is fused in the kernels)


def patch_replication_callback(model):
    """No-op.  The reference needs this to wire its thread-based Sync-BN into nn.DataParallel
    (sync_batchnorm/replicate.py:65-88); here every rank owns a persistent replica and the BN
    statistics are exchanged by an RCCL all-reduce inside the executor."""
    return model


class GaussianNoiseLayer(nn.Module):
    """Input noise of SSL_MT (pixelssl/nn/module/gaussian_noise.py:7-40).  Disabled (std=None) in every
    shipped script, in which case the input is returned unchanged; a non-None std is rejected until a
    device noise kernel exists (no silent torch fallback)."""

    def __init__(self, std=None):
        super().__init__()
        if std is not None:
            raise NotImplementedError('gaussian_noise_std is not supported by the MI355X engine yet')
        self.std = std

    def forward(self, inp):
        return inp
