"""task/sseg/func.py on the device: metrics, visualisation and the algorithm-specific task hooks of the `sseg` task
(task/sseg/func.py:20-253), with the reference's signatures and return contracts.  The reference builds its masks /
one-hot tensors / confusion matrices with numpy on the host (D2H + H2D + sync per call); here the arithmetic stays on
the GPU and only the C x C counters of the metrics leave it."""
import numpy as np
import torch

from ..task_template import func as func_template
from ..ssl_algorithm import ssl_gct as gct_modules
from ..utils import logger
from .. import functional as PF
from .._lib import check, lib, ptr, stream_ptr


def add_parser_arguments(parser):
    func_template.add_parser_arguments(parser)


def task_func():
    return SSEGFunc


class _FCDPrepare(torch.autograd.Function):
    """(fcd_pred * mask, target * mask) of ssladv_preprocess_fcd_criterion as ONE kernel (task/sseg/func.py:137-157)."""

    @staticmethod
    def forward(ctx, fcd_pred, task_gt, ignore_index, target):
        x = fcd_pred.contiguous().float()
        gt = None if task_gt is None else task_gt.contiguous().float()
        if gt is not None and gt.numel() != x.numel():
            raise ValueError("ssladv_preprocess_fcd_criterion: task_gt %s does not match fcd_pred %s"
                             % (tuple(task_gt.shape), tuple(fcd_pred.shape)))
        p, g = torch.empty_like(x), torch.empty_like(x)
        check(lib().pxl_fcd_prepare(x.numel(), ptr(x), ptr(gt), int(ignore_index), float(target), ptr(p), ptr(g), stream_ptr()))
        ctx.gt, ctx.ignore_index = gt, int(ignore_index)
        ctx.mark_non_differentiable(g)
        return p, g

    @staticmethod
    def backward(ctx, dp, dg):
        dp = dp.contiguous().float()
        dx = torch.empty_like(dp)
        check(lib().pxl_fcd_prepare(dp.numel(), ptr(dp), ptr(ctx.gt), ctx.ignore_index, 0.0, ptr(dx), None, stream_ptr()))
        return dx, None, None, None


class SSEGFunc(func_template.TaskFunc):
    """`SemanticSegmentationFunc` of the reference (task/sseg/func.py:22-253)."""

    def __init__(self, args=None):
        super().__init__(args)
        self._colorize = VOCColorize()

    # ---- all tasks ------------------------------------------------------------------------------------------------
    def metrics(self, pred, gt, inp, meters, id_str=''):
        """task/sseg/func.py:36-80: confusion matrix of arg-max(pred) vs gt over the pixels with 0 <= gt < num_classes,
        accumulated in `meters['<id>_confusion_matrix']` (numpy int64 [C,C], as the reference stores it); accuracy, class
        accuracy, mIoU and fwIoU are recomputed from the running sum after every batch.  Arg-max + bincount run on the
        device (pxl_confusion_matrix, bit-exact integer counts); 8 * C * C bytes cross PCIe per call instead of the
        whole prediction."""
        assert len(pred) == len(gt) == 1
        nc = self.args.num_classes
        confusion_matrix = PF.confusion_matrix(pred[0], gt[0], nc).cpu().numpy()
        meters.update('{0}_confusion_matrix'.format(id_str), confusion_matrix)

        acc_str = '{0}_{1}_acc'.format(id_str, self.METRIC_STR)
        acc_class_str = '{0}_{1}_acc-class'.format(id_str, self.METRIC_STR)
        mIoU_str = '{0}_{1}_mIoU'.format(id_str, self.METRIC_STR)
        fwIoU_str = '{0}_{1}_fwIoU'.format(id_str, self.METRIC_STR)
        for key in (acc_str, acc_class_str, mIoU_str, fwIoU_str):
            if meters.has_key(key):
                meters.reset(key)

        cmat_sum = meters['{0}_confusion_matrix'.format(id_str)].sum
        with np.errstate(divide='ignore', invalid='ignore'):          # empty classes give nan, nanmean skips them
            acc, acc_class, mIoU, fwIoU = metrics_from_confusion_matrix(cmat_sum)
        meters.update(acc_str, acc)
        meters.update(acc_class_str, acc_class)
        meters.update(mIoU_str, mIoU)
        meters.update(fwIoU_str, fwIoU)

    def visualize(self, out_path, id_str='', inp=None, pred=None, gt=None):
        """task/sseg/func.py:82-126: '<out_path>_<id>1-inp.png' (de-normalised image), '..2-pred.png' and '..3-gt.png'
        (VOC colour map); per-sample tensors as `split_tensor_tuple(..., reduce_dim=True)` hands them over."""
        from PIL import Image
        split = out_path.split('/')[-2]
        if split == 'train':
            dataset = list(self.args.trainset.keys())[0]
        elif split == 'val':
            dataset = list(self.args.valset)[0]
        else:
            dataset = None
            logger.log_err('The arguments \'visual_train_path\' and \'visual_val_path\' auto-set by the file: \n'
                           '\'pixelssl/task_template/proxy.py\' are changed.\n'
                           'The specific names of them are required in semantic segmentation.\n'
                           'Please check the \'visualize\' function in \'task/sseg/func.py\' for details\n')
        if dataset.startswith('pascal_voc'):
            mean = np.array([[[0.485]], [[0.456]], [[0.406]]])
            std = np.array([[[0.229]], [[0.224]], [[0.225]]])
        else:
            mean, std = np.zeros((3, 1, 1)), np.ones((3, 1, 1))
        if inp is not None:
            assert len(inp) == 1
            im = np.clip(inp[0].detach().cpu().numpy() * std + mean, 0, 1)
            Image.fromarray((np.transpose(im, (1, 2, 0)) * 255).astype('uint8')).save(out_path + '_{0}1-inp.png'.format(id_str))
        if pred is not None:
            assert len(pred) == 1
            p = pred[0].detach()
            am = PF.argmax_u8(p[None])[0].cpu().numpy() if p.is_cuda else np.argmax(p.numpy(), axis=0)
            col = np.transpose(self._colorize(am), (1, 2, 0))
            Image.fromarray((col * 255).astype('uint8')).save(out_path + '_{0}2-pred.png'.format(id_str))
        if gt is not None:
            assert len(gt) == 1
            col = np.transpose(self._colorize(gt[0].detach().cpu().numpy()[0]), (1, 2, 0))
            Image.fromarray((col * 255).astype('uint8')).save(out_path + '_{0}3-gt.png'.format(id_str))

    # ---- SSL_ADV (task/sseg/func.py:134-168) -------------------------------------------------------------------------
    def ssladv_fcd_in_channels(self):
        return self.args.num_classes

    def ssladv_preprocess_fcd_criterion(self, fcd_pred, task_gt, is_real):
        """-> (fcd_pred * mask, fcd_gt * mask): two tensors as in the reference (mask = task_gt != ignore_index, all ones
        when task_gt is None; fcd_gt = 1 real / 0 fake; masked pixels stay in the mean's denominator).  The pair also
        remembers what it was made from, which lets this package's FCDiscriminatorCriterion fuse mask, target and loss
        into one kernel; any other criterion just sees the two tensors."""
        p, g = _FCDPrepare.apply(fcd_pred, task_gt, self.args.ignore_index, 1.0 if is_real else 0.0)
        source = (fcd_pred, task_gt, int(self.args.ignore_index), bool(is_real))
        p._pxl_fcd_source = source
        g._pxl_fcd_source = source
        p._pxl_fcd_version, g._pxl_fcd_version = p._version, g._version        # an in-place edit of the pair voids the shortcut
        return p, g

    def ssladv_convert_task_gt_to_fcd_input(self, task_gt):
        # `task_gt == i` for i in range(num_classes): ignored (255) and unlabeled (-1) pixels match no class
        return gct_modules.onehot_ignore(task_gt, self.args.num_classes, ignore_index=self.args.ignore_index)

    # ---- SSL_GCT (task/sseg/func.py:175-199) -------------------------------------------------------------------------
    def sslgct_fd_in_channels(self):
        return self.args.num_classes + 3

    def sslgct_prepare_task_gt_for_fdgt(self, task_gt):
        return gct_modules.onehot_ignore(task_gt, self.args.num_classes, ignore_index=self.args.ignore_index)

    def visualize_pseudo_gt(self, pseudo_gt, out_path, id_str):
        from PIL import Image
        p = pseudo_gt[0].detach()
        am = PF.argmax_u8(p[None])[0].cpu().numpy() if p.is_cuda else np.argmax(p.numpy(), axis=0)
        col = np.transpose(self._colorize(am), (1, 2, 0))
        Image.fromarray((col * 255).astype('uint8')).save(out_path + '_{0}-pseudo-gt.png'.format(id_str))

    # ---- SSL_S4L (task/sseg/func.py:207-208) -------------------------------------------------------------------------
    def ssls4l_rc_in_channels(self):
        return self.args.num_classes

    # ---- SSL_CCT (task/sseg/func.py:216-253) -------------------------------------------------------------------------
    def sslcct_activate_ad_preds(self, ad_preds):
        return [PF.softmax_channels(p) for p in ad_preds]

    def sslcct_ad_in_channels(self):
        arch = self.args.models['model']
        if arch == 'pspnet':
            return 512
        if arch == 'deeplabv2':
            return 2048
        logger.log_err('In the SSL_CCT algorithm, you try to use \'{0}\' as the task model of sseg task.\n'
                       'However, the function \'sslcct_ad_in_channels\' does not support this model.\n'
                       'Please add this model architecture to the above function in the file \'task/sseg/func.py\'\n'.format(arch))
        return -1

    def sslcct_ad_out_channels(self):
        return self.args.num_classes

    def sslcct_ad_upsample_scale(self):
        arch = self.args.models['model']
        if arch in ['pspnet', 'deeplabv2']:
            return 8
        logger.log_err('In the SSL_CCT algorithm, you try to use \'{0}\' as the task model of sseg task.\n'
                       'However, the function \'sslcct_ad_upsample_scale\' does not support this model.\n'
                       'Please add this model architecture to the above function in the file \'task/sseg/func.py\'\n'.format(arch))
        return -1


# the reference exports the class under this name (task/sseg/func.py:22)
SemanticSegmentationFunc = SSEGFunc


def metrics_from_confusion_matrix(cm):
    """acc, class-mean acc, mIoU, fwIoU of a [C,C] confusion matrix (rows = ground truth), task/sseg/func.py:63-80."""
    cm = np.asarray(cm)
    diag = np.diag(cm)
    acc = diag.sum() / cm.sum()
    acc_class = np.nanmean(diag / cm.sum(axis=1))
    IoU = diag / (np.sum(cm, axis=1) + np.sum(cm, axis=0) - diag)
    mIoU = np.nanmean(IoU)
    freq = np.sum(cm, axis=1) / np.sum(cm)
    fwIoU = (freq[freq > 0] * IoU[freq > 0]).sum()
    return acc, acc_class, mIoU, fwIoU


class VOCColorize(object):
    """Gray label map [H,W] -> uint8 colour image [3,H,W] with the Pascal-VOC palette (task/sseg/func.py:301-322);
    255 (void) is drawn white."""

    def __init__(self, n=22):
        self.cmap = color_map(22)[:n]

    def __call__(self, gray_image):
        gray = np.asarray(gray_image)
        color = np.zeros((3,) + gray.shape, dtype=np.uint8)
        for label in range(len(self.cmap)):
            mask = gray == label
            for ch in range(3):
                color[ch][mask] = self.cmap[label][ch]
        color[:, gray == 255] = 255
        return color


def color_map(N=256, normalized=False):
    """The VOC bit-interleaved colour map (task/sseg/func.py:324-343)."""
    cmap = np.zeros((N, 3), dtype='float32' if normalized else 'uint8')
    for i in range(N):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        cmap[i] = (r, g, b)
    return cmap / 255 if normalized else cmap
