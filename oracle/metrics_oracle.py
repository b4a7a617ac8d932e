"""CPU restatement of the sseg validation metrics (task/sseg/func.py:36-80).  TEST INFRASTRUCTURE ONLY (the checker; the
product computes the confusion matrix on the GPU, pixelssl_amd/csrc/metrics.hip).  Pinned against the reference's own
`SemanticSegmentationFunc.metrics` by oracle/make_golden_metrics.py -> tests/golden/metrics_65.pt."""
import numpy as np


def confusion_matrix(pred, gt, num_classes):
    """func.py:39-48: arg-max over channels, mask (gt >= 0) & (gt < C), bincount of C * gt + pred."""
    pred = np.argmax(np.asarray(pred), axis=1)
    pred = np.expand_dims(pred, axis=1)
    gt = np.asarray(gt)
    mask = (gt >= 0) & (gt < num_classes)
    label = num_classes * gt[mask].astype('int') + pred[mask]
    return np.bincount(label, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def metrics(cm):
    """func.py:63-80 on the accumulated confusion matrix -> dict(acc, acc_class, mIoU, fwIoU)."""
    cm = np.asarray(cm)
    with np.errstate(divide='ignore', invalid='ignore'):
        acc = np.diag(cm).sum() / cm.sum()
        acc_class = np.nanmean(np.diag(cm) / cm.sum(axis=1))
        IoU = np.diag(cm) / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))
        mIoU = np.nanmean(IoU)
        freq = np.sum(cm, axis=1) / np.sum(cm)
        fwIoU = (freq[freq > 0] * IoU[freq > 0]).sum()
    return dict(acc=float(acc), acc_class=float(acc_class), mIoU=float(mIoU), fwIoU=float(fwIoU))


def synthetic_val_batch(batch, h, w, seed, num_classes=21, missing=(5, 17)):
    """Seeded (pred, gt): blobby predictions, labels with ignore (255) rings and two classes that never occur (their IoU
    is nan and must be skipped by nanmean)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    pred = F.interpolate(torch.randn(batch, num_classes, max(2, h // 8), max(2, w // 8), generator=g), size=(h, w), mode="bilinear")
    pred = torch.softmax(pred * 3 + 0.2 * torch.randn(batch, num_classes, h, w, generator=g), 1)
    ids = torch.randint(0, num_classes, (batch, 1, (h + 15) // 16, (w + 15) // 16), generator=g).float()
    for m in missing:
        ids[ids == m] = (m + 1) % num_classes
    gt = F.interpolate(ids, scale_factor=16, mode="nearest")[:, :, :h, :w].contiguous()
    gt[:, :, ::16, :] = 255.0
    gt[:, :, :, ::16] = 255.0
    # a third of the pixels follow the label, so that the diagonal is populated
    agree = torch.rand(batch, 1, h, w, generator=g) < 0.35
    onehot = torch.zeros_like(pred).scatter_(1, gt.clamp(0, num_classes - 1).long(), 1.0)
    pred = torch.where(agree & (gt < num_classes), 0.5 * pred + 0.5 * onehot, pred)
    return pred.contiguous(), gt
