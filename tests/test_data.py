"""Input pipeline: SSL dataset wrappers, the two-stream batch sampler (reference RNG stream at one
rank, a partition of the reference's global batch at N ranks), the Pascal-VOC dataset + PIL transforms on a synthetic
VOC-shaped tree, and the device finish (uint8 crops -> normalised fp32 NCHW, bit-exact numpy rounding) + prefetcher.

CPU; the comparisons against the reference's own classes run where /root/reference exists."""
import argparse
import os
import random
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402

needs_reference = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present on this box")
DEV = "cuda"


def _voc_tree(root, n=7, seed=0):
    """A tiny VOC-shaped tree: JPEGImages/*.jpg, SegmentationClassAug/*.png (the last sample has NO label file),
    ImageSets/Segmentation/{train_aug,val}.txt."""
    rng = np.random.RandomState(seed)
    for d in ("JPEGImages", "SegmentationClassAug", "ImageSets/Segmentation"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    names = []
    for i in range(n):
        name = "2007_%06d" % i if i % 2 == 0 else "2008_%06d" % i
        w, h = int(rng.randint(60, 140)), int(rng.randint(60, 140))
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, "JPEGImages", name + ".jpg"), quality=95)
        if i != n - 1:
            lab = rng.randint(0, 21, (h // 8 + 1, w // 8 + 1)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:h, :w].copy()
            lab[::8] = 255
            Image.fromarray(lab, mode="L").save(os.path.join(root, "SegmentationClassAug", name + ".png"))
        names.append(name)
    with open(os.path.join(root, "ImageSets/Segmentation/train_aug.txt"), "w") as f:
        f.write("\n".join(names))
    with open(os.path.join(root, "ImageSets/Segmentation/val.txt"), "w") as f:
        f.write("\n".join(names[:-1]))
    return names


def _dargs(root, **kw):
    a = argparse.Namespace(trainset={"pascal_voc_aug": root}, valset={"pascal_voc_aug": root}, im_size=65, train_base_size=80,
                           val_rescaling=True, num_workers=0, labeled_batch_size=2, unlabeled_batch_size=2, batch_size=4,
                           device_normalize=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_sampler_partitions_the_global_batch_across_ranks():
    from pixelssl_amd.nn.data import TwoStreamBatchSampler
    lab, unl = list(range(10)), list(range(10, 40))
    one = list(TwoStreamBatchSampler(lab, unl, 4, 6, rng=np.random.RandomState(5)))             # the global batch stream
    parts = [list(TwoStreamBatchSampler(lab, unl, 2, 3, rank=r, world_size=2, rng=np.random.RandomState(5))) for r in (0, 1)]
    assert len(one) == len(parts[0]) == len(parts[1]) == 5
    for g, a, b in zip(one, *parts):
        assert len(a) == len(b) == 5 and all(i < 10 for i in a[:2] + b[:2]) and all(i >= 10 for i in a[2:] + b[2:])
        assert tuple(a[:2]) + tuple(b[:2]) == tuple(g[:4]) and tuple(a[2:]) + tuple(b[2:]) == tuple(g[4:])   # labeled first, per rank
    # the longer stream is visited once per epoch, the shorter one re-shuffled as needed
    seen = [i for g in one for i in g[4:]]
    assert len(set(seen)) == len(seen) == 30
    with pytest.raises(AssertionError):
        TwoStreamBatchSampler(lab, unl, 6, 6, rank=0, world_size=2, rng=np.random.RandomState(1))  # 12 labeled per global batch > 10
    with pytest.raises(ValueError):          # multi-rank draws need a generator every rank seeds identically
        TwoStreamBatchSampler(lab, unl, 2, 3, rank=0, world_size=2)


def test_labeled_only_loader_is_sharded_by_rank():
    """the labeled-only path of make_train_loader ignored rank / world_size.  Now: one global permutation per
    epoch, disjoint per-rank slices whose union is the global batch; world_size > 1 without a shared rng is an error."""
    import argparse
    import torch
    from pixelssl_amd.nn.data import ShardedBatchSampler
    from pixelssl_amd.sseg.data import make_train_loader
    one = list(ShardedBatchSampler(23, 4, rng=np.random.RandomState(7)))
    parts = [list(ShardedBatchSampler(23, 2, rank=r, world_size=2, rng=np.random.RandomState(7))) for r in (0, 1)]
    assert len(one) == len(parts[0]) == len(parts[1]) == 5
    for g, a, b in zip(one, *parts):
        assert a + b == g and len(set(g)) == 4
    flat = [i for g in one for i in g]
    assert len(set(flat)) == len(flat) == 20          # drop_last: 23 // 4 batches, no sample twice in an epoch
    with pytest.raises(ValueError):
        ShardedBatchSampler(23, 2, rank=0, world_size=2)

    class DS(torch.utils.data.Dataset):
        unlabeled_idxs, labeled_idxs = [], list(range(23))
        def __len__(self): return 23
        def __getitem__(self, i): return torch.tensor([i])
    args = argparse.Namespace(batch_size=2, unlabeled_batch_size=0, labeled_batch_size=2, num_workers=0)
    seen = [sorted(int(v) for batch in make_train_loader(DS(), args, rank=r, world_size=2, rng=np.random.RandomState(3)) for v in batch.view(-1))
            for r in (0, 1)]
    assert not set(seen[0]) & set(seen[1]) and len(seen[0]) == len(seen[1]) == 10


@needs_reference
def test_sampler_and_wrappers_reproduce_the_reference():
    from pixelssl_amd.nn import data as O
    ref_shim.load_reference()
    from pixelssl.nn import data as R
    for nl, nu, lbs, ubs in ((10, 40, 2, 6), (50, 12, 5, 3), (9, 9, 3, 3)):
        lab, unl = list(range(nl)), list(range(nl, nl + nu))
        np.random.seed(11)
        want = [tuple(int(i) for i in b) for b in R.TwoStreamBatchSampler(lab, unl, lbs, ubs)]
        np.random.seed(11)
        got = [tuple(int(i) for i in b) for b in O.TwoStreamBatchSampler(lab, unl, lbs, ubs)]
        assert got == want and len(O.TwoStreamBatchSampler(lab, unl, lbs, ubs)) == len(R.TwoStreamBatchSampler(lab, unl, lbs, ubs))

    class Fake(torch.utils.data.Dataset):
        def __init__(self, names):
            self.sample_list, self.idxs = list(names), list(range(len(names)))

        def __len__(self):
            return len(self.sample_list)

        def __getitem__(self, i):
            return self.sample_list[i]
    names = ["a1", "b1", "a2", "c1", "b2", "a3", "d1"]
    for ignore in (False, True):
        r, o = R.SplitUnlabeledWrapper(Fake(names), ["a", "c"], ignore), O.SplitUnlabeledWrapper(Fake(names), ["a", "c"], ignore)
        assert r.dataset.sample_list == o.dataset.sample_list and list(r.labeled_idxs) == list(o.labeled_idxs)
        assert list(r.unlabeled_idxs) == list(o.unlabeled_idxs) and len(r) == len(o) and [r[i] for i in range(len(r))] == [o[i] for i in range(len(o))]
        rj = R.JointDatasetsWrapper([Fake(names[:3]), Fake(names[3:5])], [Fake(names[5:])], ignore)
        oj = O.JointDatasetsWrapper([Fake(names[:3]), Fake(names[3:5])], [Fake(names[5:])], ignore)
        assert len(rj) == len(oj) and list(rj.labeled_idxs) == list(oj.labeled_idxs) and list(rj.unlabeled_idxs) == list(oj.unlabeled_idxs)
        assert [rj[i] for i in range(len(rj))] == [oj[i] for i in range(len(oj))]


@needs_reference
def test_pascal_voc_dataset_and_transforms_reproduce_the_reference(tmp_path):
    """Same files, same `random` seed -> identical tensors from the reference's PascalVocAugDataset and this package's, in
    training mode (random scale / crop / flip / normalise; the label-less sample gets the -1 plane) and validation mode
    (short-edge resize + zero pad, where the reference needs OpenCV for the padding)."""
    import types
    import importlib
    from pixelssl_amd.sseg import data as O
    _voc_tree(str(tmp_path))
    ref_shim.load_reference()
    tv = sys.modules["torchvision.transforms"]
    tv.Compose = O.Compose                       # the stubbed torchvision of the shim gets the one function that is used
    cv2 = sys.modules["cv2"]
    cv2.BORDER_CONSTANT = 0
    cv2.copyMakeBorder = lambda img, t, b, l, r, kind, value=None: np.pad(
        img, ((t, b), (l, r)) + (((0, 0),) if img.ndim == 3 else ()), mode="constant")
    R = importlib.import_module("data")          # task/sseg/data.py of the reference
    assert O.pascal_voc_aug().__name__ == R.pascal_voc_aug().__name__ and O.pascal_voc_ori().__name__ == R.pascal_voc_ori().__name__
    for is_train in (True, False):
        rd, od = R.PascalVocAugDataset(_dargs(str(tmp_path)), is_train), O.PascalVocAugDataset(_dargs(str(tmp_path)), is_train)
        assert rd.sample_list == od.sample_list and rd.idxs == od.idxs and len(rd) == len(od)
        for i in range(len(rd)):
            random.seed(100 + i)
            (ri,), (rl,) = rd[i]
            random.seed(100 + i)
            (oi,), (ol,) = od[i]
            assert ri.dtype == oi.dtype == torch.float32 and torch.equal(ri, oi), (is_train, i)
            assert rl.shape == ol.shape and torch.equal(rl, ol), (is_train, i)
        if is_train:
            (img,), (lab,) = od[len(od) - 1]
            assert img.shape == (3, 65, 65) and lab.shape == (65, 65) and bool((lab == -1).all())      # label-less sample
    # parser flags
    p1, p2 = argparse.ArgumentParser(), argparse.ArgumentParser()
    R.add_parser_arguments(p1), O.add_parser_arguments(p2)
    assert {a.dest: a.default for a in p1._actions} == {a.dest: a.default for a in p2._actions}


@pytest.mark.gpu
def test_device_normalize_and_prefetcher_are_bit_exact(tmp_path):
    """The uint8 path (workers ship crops, the GPU normalises) gives exactly the tensors of the reference arrangement
    (workers normalise), batch by batch through the two-stream sampler and the prefetcher."""
    from pixelssl_amd.sseg import data as O
    from pixelssl_amd.nn import data as ND
    names = _voc_tree(str(tmp_path), n=9)
    os.remove(os.path.join(str(tmp_path), "ImageSets/Segmentation/train_aug.txt"))
    with open(os.path.join(str(tmp_path), "ImageSets/Segmentation/train_aug.txt"), "w") as f:
        f.write("\n".join(names[:-1]))                    # every sample of this run has a label file (one collate shape)

    def loader(device_normalize):
        ds = O.PascalVocAugDataset(_dargs(str(tmp_path), device_normalize=device_normalize), True)
        wrapped = ND.SplitUnlabeledWrapper(ds, ["2007"], ignore_unlabeled=False)
        return O.make_train_loader(wrapped, ds.args, rng=np.random.RandomState(3))
    random.seed(21)
    ref_batches = [b for b in loader(False)]
    random.seed(21)
    pre = ND.DevicePrefetcher(loader(True), device=DEV, finish=O.DeviceNormalize())
    got_batches = [b for b in pre]
    torch.cuda.synchronize()
    assert len(ref_batches) == len(got_batches) >= 2
    for ((ri,), (rl,)), ((gi,), (gl,)) in zip(ref_batches, got_batches):
        assert gi.is_cuda and gi.dtype == torch.float32 and gi.shape == ri.shape == (4, 3, 65, 65)
        assert torch.equal(gi.cpu(), ri) and torch.equal(gl.cpu(), rl)
    # the unlabeled stand-in
    lab = torch.full((1, 1, 4, 4), O.UNLABELED_U8, dtype=torch.uint8, device=DEV)
    lab[0, 0, 0, 0] = 255
    _, (lf,) = O.DeviceNormalize()((torch.zeros(1, 4, 4, 3, dtype=torch.uint8, device=DEV),), (lab,))
    assert lf[0, 0, 0, 0].item() == 255.0 and lf[0, 0, 1, 1].item() == -1.0
