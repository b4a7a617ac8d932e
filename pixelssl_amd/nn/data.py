"""SSL dataset wrappers and the labeled/unlabeled batch sampler (pixelssl/nn/data.py:13-177), plus what one process per
GPU needs on the step's input side: a rank-sharded sampler and a device prefetcher.

  * `SplitUnlabeledWrapper` / `JointDatasetsWrapper`: same index contract as the reference (labeled indices first).
  * `TwoStreamBatchSampler`: with rank = 0, world_size = 1 and no `rng` it consumes numpy's global RNG exactly like the
    reference (same batches under the same seed).  With world_size = W every rank draws the SAME global stream of
    W * lbs labeled + W * ubs unlabeled indices and keeps its own slice, labeled first -- the batch layout contract of
    `split_tensor_tuple` (nn/func.py:24-51) holds per rank, and the union over ranks is the reference's global batch
    (the reference scales the script's per-GPU batch sizes by #GPUs, task_template/proxy.py:258-261).
  * `DevicePrefetcher`: pinned-memory batches are copied host-to-device on a side stream one iteration ahead; uint8 image
    batches (sseg.data with device_normalize) are normalised and laid out NCHW by one kernel on arrival, so the workers
    ship 1 byte per value instead of 4 and the reference's duplicate upload of MT (ssl_mt.py:344-348) disappears.
"""
import itertools

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler


class _SSLDatasetWrapper(Dataset):
    def __init__(self):
        super().__init__()
        self.labeled_idxs = []
        self.unlabeled_idxs = []


class SplitUnlabeledWrapper(_SSLDatasetWrapper):
    """A fully labeled dataset whose samples count as labeled only if their name starts with one of
    `sublabeled_prefix`; the others are served as unlabeled (or dropped with ignore_unlabeled).  Re-orders
    `dataset.sample_list` to labeled-first like the reference (nn/data.py:54-77)."""

    def __init__(self, dataset, sublabeled_prefix, ignore_unlabeled=False):
        super().__init__()
        self.dataset = dataset
        self.sublabeled_prefix = sublabeled_prefix
        self.ignore_unlabeled = ignore_unlabeled
        self._split_labeled()

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset[idx]

    def _split_labeled(self):
        prefixes = tuple(self.sublabeled_prefix)
        names = self.dataset.sample_list
        flags = [any(n.startswith(p) for p in prefixes) for n in names]
        labeled = [n for n, f in zip(names, flags) if f]
        unlabeled = [n for n, f in zip(names, flags) if not f]
        if self.ignore_unlabeled:
            unlabeled = []
        self.dataset.sample_list = labeled + unlabeled
        self.dataset.idxs = list(range(len(self.dataset.sample_list)))
        self.labeled_idxs = list(range(len(labeled))) if not self.ignore_unlabeled else self.dataset.idxs
        self.unlabeled_idxs = [len(labeled) + i for i in range(len(unlabeled))]


class JointDatasetsWrapper(_SSLDatasetWrapper):
    """Several labeled and unlabeled datasets behind one index space: labeled samples first (nn/data.py:80-123)."""

    def __init__(self, labeled_datasets, unlabeled_datasets, ignore_unlabeled=False):
        super().__init__()
        self.labeled_datasets = labeled_datasets
        self.unlabeled_datasets = unlabeled_datasets
        self.ignore_unlabeled = ignore_unlabeled
        self.labeled_datasets_size = [len(d) for d in labeled_datasets]
        self.unlabeled_datasets_size = [len(d) for d in unlabeled_datasets]
        self.labeled_size = int(sum(self.labeled_datasets_size))
        self.labeled_idxs = list(range(self.labeled_size))
        self.unlabeled_size = 0
        if not ignore_unlabeled:
            self.unlabeled_size = int(sum(self.unlabeled_datasets_size))
            self.unlabeled_idxs = [self.labeled_size + i for i in range(self.unlabeled_size)]

    def __len__(self):
        return int(self.labeled_size + self.unlabeled_size)

    def __getitem__(self, idx):
        assert 0 <= idx < len(self)
        if idx >= self.labeled_size:
            idx, datasets = idx - self.labeled_size, self.unlabeled_datasets
        else:
            datasets = self.labeled_datasets
        for d in datasets:
            if idx < len(d):
                return d[idx]
            idx -= len(d)
        raise IndexError(idx)


class ShardedBatchSampler(Sampler):
    """The labeled-only training loader of task_template/proxy.py:365-375 (DataLoader(shuffle=True, drop_last=True) over
    batch_size = per-GPU batch x #GPUs) for one rank: ONE global permutation per epoch, cut into global batches of
    batch_size x world_size, of which this rank takes its slice -- the union over the ranks is what the reference's
    single process feeds nn.DataParallel.  All ranks must draw the same permutation: `rng` is a numpy RandomState seeded
    identically everywhere (required when world_size > 1)."""

    def __init__(self, num_samples, batch_size, rank=0, world_size=1, rng=None):
        self.n, self.batch_size = int(num_samples), int(batch_size)
        self.rank, self.world_size, self.rng = int(rank), int(world_size), rng
        assert 0 <= self.rank < self.world_size
        if self.world_size > 1 and rng is None:
            raise ValueError('ShardedBatchSampler: world_size > 1 needs an explicit rng seeded identically on every rank '
                             '(the global numpy state is not synchronised across processes)')
        self._g = self.batch_size * self.world_size
        assert self.n >= self._g > 0

    def __len__(self):
        return self.n // self._g

    def __iter__(self):
        perm = (self.rng or np.random).permutation(self.n)
        b, r = self.batch_size, self.rank
        for k in range(len(self)):
            yield [int(i) for i in perm[k * self._g + r * b:k * self._g + (r + 1) * b]]


class TwoStreamBatchSampler(Sampler):
    """Labeled-first mini-batches from two index streams; an epoch runs through the longer stream once and re-shuffles
    the shorter one as often as needed (nn/data.py:126-177).  `labeled_batch_size` / `unlabeled_batch_size` are PER
    RANK; see the module docstring for the multi-rank draw."""

    def __init__(self, labeled_idxs, unlabeled_idxs, labeled_batch_size, unlabeled_batch_size, rank=0, world_size=1,
                 rng=None):
        self.labeled_idxs = labeled_idxs
        self.unlabeled_idxs = unlabeled_idxs
        self.labeled_batch_size = labeled_batch_size
        self.unlabeled_batch_size = unlabeled_batch_size
        self.rank, self.world_size, self.rng = int(rank), int(world_size), rng
        assert 0 <= self.rank < self.world_size
        if self.world_size > 1 and rng is None:
            raise ValueError('TwoStreamBatchSampler: world_size > 1 needs an explicit rng seeded identically on every rank '
                             '(the global numpy state is not synchronised across processes)')
        self._gl, self._gu = labeled_batch_size * self.world_size, unlabeled_batch_size * self.world_size
        assert len(self.labeled_idxs) >= self._gl > 0
        assert len(self.unlabeled_idxs) >= self._gu > 0
        self.unlabeled_batchs = len(self.unlabeled_idxs) // self._gu
        self.labeled_batchs = len(self.labeled_idxs) // self._gl

    def _permutation(self, idxs):
        return (self.rng or np.random).permutation(idxs)

    def iterate_once(self, iterable):
        return self._permutation(iterable)

    def iterate_eternally(self, indices):
        def shuffles():
            while True:
                yield self._permutation(indices)
        return itertools.chain.from_iterable(shuffles())

    @staticmethod
    def grouper(iterable, n):
        return zip(*([iter(iterable)] * n))

    def __iter__(self):
        if self.unlabeled_batchs >= self.labeled_batchs:
            unlabeled_iter = self.iterate_once(self.unlabeled_idxs)
            labeled_iter = self.iterate_eternally(self.labeled_idxs)
        else:
            unlabeled_iter = self.iterate_eternally(self.unlabeled_idxs)
            labeled_iter = self.iterate_once(self.labeled_idxs)
        lb, ub, r = self.labeled_batch_size, self.unlabeled_batch_size, self.rank
        return (lab[r * lb:(r + 1) * lb] + unl[r * ub:(r + 1) * ub]
                for lab, unl in zip(self.grouper(labeled_iter, self._gl), self.grouper(unlabeled_iter, self._gu)))

    def __len__(self):
        return max(self.unlabeled_batchs, self.labeled_batchs)


class DevicePrefetcher:
    """Iterates a DataLoader whose batches are (inputs: tuple, labels: tuple) of pinned host tensors and hands them out
    already on the device: the copy of batch i+1 is enqueued on a side stream while step i runs.  `finish` (optional) runs
    on the side stream after the copy, e.g. sseg.data.DeviceNormalize for uint8 image batches."""

    def __init__(self, loader, device=None, finish=None):
        self.loader = loader
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.finish = finish
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        inp, gt = batch
        with torch.cuda.stream(self.stream):
            inp = tuple(t.to(self.device, non_blocking=True) for t in inp)
            gt = tuple(t.to(self.device, non_blocking=True) for t in gt)
            if self.finish is not None:
                inp, gt = self.finish(inp, gt)
        return inp, gt

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            batch = nxt
            for t in batch[0] + batch[1]:
                t.record_stream(cur)
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            yield batch
