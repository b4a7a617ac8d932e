"""Synthetic sseg batches with the statistics of the real input pipeline: what bench.py, smoke() and the
examples feed the engine when no dataset is mounted.  Images ~ N(0, 1) (ImageNet-normalised photos have about unit
variance, task/sseg/data.py:99); labels are float32 [B,1,H,W] class ids per `block` x `block` cell with a 1-pixel ring of
255 (ignore_index) on the cell edges, about 6 % of the pixels like VOC's object boundaries; unlabeled samples carry -1
everywhere (task/sseg/data.py:104-105).  Labeled samples come first (nn/data.py:148-159)."""
import torch
import torch.nn.functional as F


def synthetic_batch(batch, size, lbs, seed, num_classes=21, block=32):
    """-> (x [B,3,size,size] fp32, gt [B,1,size,size] fp32); the first `lbs` samples are labeled."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, size, size, generator=g)
    cells = (size + block - 1) // block
    ids = torch.randint(0, num_classes, (batch, 1, cells, cells), generator=g).float()
    gt = F.interpolate(ids, scale_factor=block, mode="nearest")[:, :, :size, :size].contiguous()
    edge = torch.arange(size) % block == 0
    gt[:, :, edge[:, None] | edge[None, :]] = 255.0
    if lbs < batch:
        gt[lbs:] = -1.0
    return x, gt
