"""Pascal-VOC input pipeline of the `sseg` task (task/sseg/data.py:18-294): same dataset classes, export functions, parser
flags and PIL transforms (same draws from Python's `random`, so the same seed gives the same crops), and the same sample
format: `(image [3,H,W] fp32 normalised,), (label [1,H,W] fp32,)`, unlabeled samples labeled -1.

MI355X-first split of the work.  The geometric part (decode, bilinear / nearest resize, pad, crop, flip) stays on the CPU
workers with PIL -- its arithmetic is what the reference's results depend on.  The arithmetic part (`Normalize` +
`ToTensor`) can move to the GPU: with `args.device_normalize` the workers return the uint8 crop and `DeviceNormalize`
finishes the batch on arrival (pxl_normalize_u8: bit-exact numpy rounding, NHWC -> NCHW), so 1 byte per value crosses PCIe
instead of 4 and the 64-core host is left with decode + resize.  nn.data.DevicePrefetcher drives it one batch ahead.
"""
import os
import random

import numpy as np
import torch
from PIL import Image, ImageOps, ImageFilter

from ..task_template import data as data_template
from ..utils import logger, cmd
from .._lib import check, lib, ptr, stream_ptr

VOC_MEAN = (0.485, 0.456, 0.406)
VOC_STD = (0.229, 0.224, 0.225)


def add_parser_arguments(parser):
    data_template.add_parser_arguments(parser)
    parser.add_argument('--val-rescaling', type=cmd.str2bool, default=False,
                        help='sseg - if true, the short edge of the outputs is scaled to the size of the inputs, and the '
                             'long edge is scaled by using the same ratio')
    parser.add_argument('--train-base-size', type=int, default=400,
                        help='sseg - base size of random image cropping during training')


def pascal_voc_aug():
    return PascalVocAugDataset


def pascal_voc_ori():
    return PascalVocOriDataset


class Compose:
    """torchvision.transforms.Compose: the only thing the reference pipeline takes from torchvision."""

    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


class PascalVocDataset(data_template.TaskDataset):
    IMAGE = 'image'
    LABEL = 'label'
    PREFIX = 'prefix'

    def __init__(self, args, is_train, train_prefix_path, val_prefix_path):
        super().__init__(args, is_train)
        self.im_size = self.args.im_size
        self.transform = None
        self.device_normalize = bool(getattr(args, 'device_normalize', False))
        self.fliplr = bool(self.is_train)
        self.prefix_path = os.path.join(self.root_dir, train_prefix_path if self.is_train else val_prefix_path)
        self.image_dir = os.path.join(self.root_dir, 'JPEGImages')
        self.label_dir = os.path.join(self.root_dir, 'SegmentationClassAug')
        with open(self.prefix_path, 'r') as f:
            lines = f.read().splitlines()
        for line in lines:
            if not os.path.isfile(os.path.join(self.image_dir, line + '.jpg')):
                logger.log_err('Cannot find image: {0} in Pascal VOC Dataset\n'.format(os.path.join(self.image_dir, line + '.jpg')))
            self.sample_list.append(line)
        self.idxs = list(range(len(self.sample_list)))

    def __getitem__(self, idx):
        name = self.sample_list[idx]
        image_path = os.path.join(self.image_dir, name + '.jpg')
        label_path = os.path.join(self.label_dir, name + '.png')
        has_label = os.path.exists(label_path)
        if not self.is_train and not has_label:
            logger.log_err('The val sample of Pascal VOC dataset should has label\n'
                           'However, cannot find label: {0}\n'.format(label_path))
        image = self.im_loader.load(image_path).convert('RGB')
        label = self.im_loader.load(label_path) if has_label else None
        image, label = self._train_prehandle(image, label) if self.is_train else self._val_prehandle(image, label)
        label = label[None, :, :] if has_label else label      # (the reference leaves an unlabeled sample's -1 plane 2-D)
        return (image,), (label,)

    def _finish(self):
        """Normalize + ToTensor on the worker (the reference's arrangement), or nothing: the device does it."""
        return [ToUint8Tensor()] if self.device_normalize else [Normalize(mean=VOC_MEAN, std=VOC_STD), ToTensor()]

    def _train_prehandle(self, image, label):
        sample = {self.IMAGE: image, self.LABEL: image if label is None else label}
        out = Compose([RandomScaleCrop(base_size=self.args.train_base_size, crop_size=self.args.im_size),
                       RandomHorizontalFlip()] + self._finish())(sample)
        if label is None:
            if self.device_normalize:           # uint8 carrier of "unlabeled": 254 everywhere -> DeviceNormalize writes -1
                return out[self.IMAGE], torch.full(out[self.IMAGE].shape[:2], UNLABELED_U8, dtype=torch.uint8)
            return out[self.IMAGE], out[self.IMAGE][0, ...] * 0.0 - 1.0
        return out[self.IMAGE], out[self.LABEL]

    def _val_prehandle(self, image, label):
        sample = {self.IMAGE: image, self.LABEL: label}
        ts = ([FixedScaleResize(size=self.args.im_size)] if self.args.val_rescaling else []) + self._finish()
        out = Compose(ts)(sample)
        return out[self.IMAGE], out[self.LABEL]


class PascalVocAugDataset(PascalVocDataset):
    def __init__(self, args, is_train):
        super().__init__(args, is_train, 'ImageSets/Segmentation/train_aug.txt', 'ImageSets/Segmentation/val.txt')


class PascalVocOriDataset(PascalVocDataset):
    def __init__(self, args, is_train):
        super().__init__(args, is_train, 'ImageSets/Segmentation/train.txt', 'ImageSets/Segmentation/val.txt')


# ---------------------------------------------------------------------------------------------------------------------
# transforms on {'image': PIL, 'label': PIL} samples (task/sseg/data.py:141-294)
# ---------------------------------------------------------------------------------------------------------------------

class Normalize(object):
    def __init__(self, mean=(0., 0., 0.), std=(1., 1., 1.)):
        self.mean, self.std = mean, std

    def __call__(self, sample):
        img = np.array(sample['image']).astype(np.float32)
        mask = np.array(sample['label']).astype(np.float32)
        img /= 255.0
        img -= self.mean
        img /= self.std
        return {'image': img, 'label': mask}


class ToTensor(object):
    def __call__(self, sample):
        img = np.array(sample['image']).astype(np.float32).transpose((2, 0, 1))
        mask = np.array(sample['label']).astype(np.float32)
        return {'image': torch.from_numpy(img).float(), 'label': torch.from_numpy(mask).float()}


UNLABELED_U8 = 254         # no VOC class id (0..20, 255 = ignore): the uint8 stand-in of the reference's -1 label plane


class ToUint8Tensor(object):
    """device_normalize: hand the crop over as it is (uint8 HWC image, uint8 HW label); DeviceNormalize does the rest."""

    def __call__(self, sample):
        img = np.ascontiguousarray(np.array(sample['image'], dtype=np.uint8))
        mask = np.ascontiguousarray(np.array(sample['label'], dtype=np.uint8))
        if mask.ndim == 3:                # an unlabeled sample carries its image in the label slot
            mask = mask[..., 0]
        return {'image': torch.from_numpy(img), 'label': torch.from_numpy(mask)}


class RandomHorizontalFlip(object):
    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        if random.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
            mask = mask.transpose(Image.FLIP_LEFT_RIGHT)
        return {'image': img, 'label': mask}


class RandomRotate(object):
    def __init__(self, degree):
        self.degree = degree

    def __call__(self, sample):
        deg = random.uniform(-1 * self.degree, self.degree)
        return {'image': sample['image'].rotate(deg, Image.BILINEAR), 'label': sample['label'].rotate(deg, Image.NEAREST)}


class RandomGaussianBlur(object):
    def __call__(self, sample):
        img = sample['image']
        if random.random() < 0.5:
            img = img.filter(ImageFilter.GaussianBlur(radius=random.random()))
        return {'image': img, 'label': sample['label']}


class RandomScaleCrop(object):
    """Random short-edge scale in [0.5, 2.0] x base_size, zero pad to the crop size, random crop (data.py:223-254)."""

    def __init__(self, base_size, crop_size, fill=0):
        self.base_size, self.crop_size, self.fill = base_size, crop_size, fill

    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        short = random.randint(int(self.base_size * 0.5), int(self.base_size * 2.0))
        w, h = img.size
        if h > w:
            ow, oh = short, int(1.0 * h * short / w)
        else:
            oh, ow = short, int(1.0 * w * short / h)
        img = img.resize((ow, oh), Image.BILINEAR)
        mask = mask.resize((ow, oh), Image.NEAREST)
        if short < self.crop_size:
            padh = self.crop_size - oh if oh < self.crop_size else 0
            padw = self.crop_size - ow if ow < self.crop_size else 0
            img = ImageOps.expand(img, border=(0, 0, padw, padh), fill=0)
            mask = ImageOps.expand(mask, border=(0, 0, padw, padh), fill=self.fill)
        w, h = img.size
        x1 = random.randint(0, w - self.crop_size)
        y1 = random.randint(0, h - self.crop_size)
        box = (x1, y1, x1 + self.crop_size, y1 + self.crop_size)
        return {'image': img.crop(box), 'label': mask.crop(box)}


class FixedScaleResize(object):
    """Short edge -> size (bilinear / nearest), zero pad at the right / bottom up to size (data.py:257-294; the
    reference pads with cv2.copyMakeBorder(BORDER_CONSTANT, 0), here numpy)."""

    def __init__(self, size):
        self.size = size

    def __call__(self, sample):
        img, mask = sample['image'], sample['label']
        w, h = img.size
        if w <= h:
            ow, oh = self.size, h * self.size / w
        else:
            oh, ow = self.size, w * self.size / h
        oh, ow = int(oh), int(ow)
        img = img.resize((ow, oh), Image.BILINEAR)
        mask = mask.resize((ow, oh), Image.NEAREST)
        pad_w, pad_h = max(self.size - ow, 0), max(self.size - oh, 0)
        if pad_w > 0 or pad_h > 0:
            a = np.pad(np.array(img).astype(np.float32), ((0, pad_h), (0, pad_w), (0, 0)), mode='constant')
            m = np.pad(np.array(mask).astype(np.float32), ((0, pad_h), (0, pad_w)), mode='constant')
            img, mask = Image.fromarray(a.astype(np.uint8)), Image.fromarray(m.astype(np.uint8))
        return {'image': img, 'label': mask}


# ---------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------

class DeviceNormalize:
    """`finish` callback of nn.data.DevicePrefetcher for datasets built with args.device_normalize: uint8 image batch
    [B,H,W,3] -> `Normalize(mean, std)` + `ToTensor` as fp32 [B,3,H,W] with numpy's rounding; uint8 label batch [B,1,H,W]
    -> fp32, the unlabeled stand-in (254) becoming -1."""

    def __init__(self, mean=VOC_MEAN, std=VOC_STD):
        self.mean, self.std = tuple(mean), tuple(std)
        self._coef = {}

    def __call__(self, inp, gt):
        (img,), (lab,) = inp, gt
        if img.dtype != torch.uint8:
            return inp, gt
        B, H, W, C = img.shape
        key = img.device
        if key not in self._coef:
            self._coef[key] = (torch.tensor(self.mean, dtype=torch.float64, device=img.device),
                               torch.tensor(self.std, dtype=torch.float64, device=img.device))
        mean, std = self._coef[key]
        img = img.contiguous()
        out = torch.empty(B, C, H, W, device=img.device, dtype=torch.float32)
        check(lib().pxl_normalize_u8(B, C, H * W, ptr(img), ptr(mean), ptr(std), ptr(out), stream_ptr()))
        lab = lab.contiguous()
        labf = torch.empty(lab.shape, device=lab.device, dtype=torch.float32)
        check(lib().pxl_u8_to_f32(lab.numel(), ptr(lab), ptr(labf), UNLABELED_U8, -1.0, stream_ptr()))
        return (out,), (labf,)


def make_train_loader(dataset, args, rank=0, world_size=1, rng=None):
    """The training DataLoader of task_template/proxy.py:365-375 for one rank: two-stream batches when the dataset has
    unlabeled samples, plain shuffled batches otherwise; pinned memory; `num_workers` as given (per rank).  With
    world_size > 1 both kinds are sharded by rank and `rng` (a numpy RandomState seeded identically on every rank) is
    REQUIRED: the ranks must draw the same permutations."""
    from ..nn import data as nndata
    unl = getattr(dataset, 'unlabeled_idxs', [])
    if len(unl) > 0 and args.unlabeled_batch_size > 0:
        sampler = nndata.TwoStreamBatchSampler(dataset.labeled_idxs, unl, args.labeled_batch_size, args.unlabeled_batch_size,
                                               rank=rank, world_size=world_size, rng=rng)
        return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, num_workers=args.num_workers, pin_memory=True)
    if world_size > 1:
        # one global permutation per epoch, this rank's slice of every global batch (the plain shuffled DataLoader below
        # would give every rank the same samples with equal seeds and overlapping ones otherwise)
        sampler = nndata.ShardedBatchSampler(len(dataset), args.batch_size, rank=rank, world_size=world_size, rng=rng)
        return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, num_workers=args.num_workers, pin_memory=True)
    return torch.utils.data.DataLoader(dataset, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers,
                                       pin_memory=True, drop_last=True)
