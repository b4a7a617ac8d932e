"""tests/golden/metrics_65.pt from the REAL reference's `SemanticSegmentationFunc.metrics` (container only).
TEST INFRASTRUCTURE.   python oracle/make_golden_metrics.py"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim            # noqa: E402
import metrics_oracle as MO  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "metrics_65.pt")


def main():
    ref = ref_shim.load_reference()
    from pixelssl.utils import logger as rlog
    F = ref["func"].SemanticSegmentationFunc
    me = argparse.Namespace(args=argparse.Namespace(num_classes=21), METRIC_STR=F.METRIC_STR)   # metrics() reads only these
    meters = rlog.AvgMeterSet()
    shapes = [(2, 65, 65, 301), (1, 49, 81, 302), (3, 33, 65, 303)]
    mine = np.zeros((21, 21), dtype=np.int64)
    per_batch = []
    for b, h, w, seed in shapes:
        pred, gt = MO.synthetic_val_batch(b, h, w, seed)
        F.metrics(me, (pred,), (gt,), None, meters, id_str='task')
        cm = MO.confusion_matrix(pred.numpy(), gt.numpy(), 21)
        mine += cm
        assert np.array_equal(meters['task_confusion_matrix'].val, cm)
        per_batch.append({k: float(meters['task_metric_' + k].val) for k in ('acc', 'acc-class', 'mIoU', 'fwIoU')})
        m = MO.metrics(mine)
        for k, mk in (('acc', 'acc'), ('acc-class', 'acc_class'), ('mIoU', 'mIoU'), ('fwIoU', 'fwIoU')):
            assert per_batch[-1][k] == m[mk], (k, per_batch[-1][k], m[mk])
    assert np.array_equal(meters['task_confusion_matrix'].sum, mine)
    assert mine[5].sum() == 0 and mine[17].sum() == 0          # the never-occurring classes: nan IoU, skipped
    print("reference metrics after 3 batches:", per_batch[-1])
    torch.save(dict(shapes=shapes, confusion_matrix=torch.from_numpy(mine), per_batch=per_batch,
                    keys=sorted(meters.keys())), OUT)
    print("wrote", OUT, "; oracle == reference (bit-exact)")


if __name__ == "__main__":
    main()
