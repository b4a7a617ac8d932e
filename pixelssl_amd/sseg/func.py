"""task/sseg/func.py on the device: the algorithm-specific task hooks of the `sseg` task (task/sseg/func.py:130-192).
The reference builds its masks / one-hot tensors with numpy on the host (D2H + H2D + sync per call); here the
hooks stay on the GPU and hand the criterion a *descriptor* it fuses into its kernel."""
import torch

from ..task_template import func as func_template
from ..ssl_algorithm import ssl_gct as gct_modules


def add_parser_arguments(parser):
    func_template.add_parser_arguments(parser)


def task_func():
    return SSEGFunc


class FCDTarget:
    """Ground truth of the FC-discriminator criterion as ssladv_preprocess_fcd_criterion defines it
    (task/sseg/func.py:137-157): target = 1 (real) / 0 (fake) wherever the task label is not ignore_index; ignored
    pixels are masked to 0 in BOTH prediction and target and still count in the mean.  FCDiscriminatorCriterion
    consumes it directly (one fused kernel); `materialize()` gives the reference's (fcd_pred, fcd_gt) tensors."""

    def __init__(self, task_gt, is_real, ignore_index):
        self.task_gt, self.is_real, self.ignore_index = task_gt, bool(is_real), int(ignore_index)

    def mask(self, like):
        if self.task_gt is None:
            return torch.ones_like(like)
        return (self.task_gt != self.ignore_index).float()

    def materialize(self, fcd_pred):
        m = self.mask(fcd_pred)
        return fcd_pred * m, torch.full_like(fcd_pred, 1.0 if self.is_real else 0.0) * m


class SSEGFunc(func_template.TaskFunc):
    # ---- SSL_ADV (task/sseg/func.py:134-168)
    def ssladv_fcd_in_channels(self):
        return self.args.num_classes

    def ssladv_preprocess_fcd_criterion(self, fcd_pred, task_gt, is_real):
        """-> (fcd_pred, FCDTarget): the masking is applied inside the criterion kernel."""
        return fcd_pred, FCDTarget(task_gt, is_real, self.args.ignore_index)

    def ssladv_convert_task_gt_to_fcd_input(self, task_gt):
        # `task_gt == i` for i in range(num_classes): ignored (255) and unlabeled (-1) pixels match no class
        return gct_modules.onehot_ignore(task_gt, self.args.num_classes, ignore_index=self.args.ignore_index)

    # ---- SSL_GCT (task/sseg/func.py:175-192)
    def sslgct_fd_in_channels(self):
        return self.args.num_classes + 3

    def sslgct_prepare_task_gt_for_fdgt(self, task_gt):
        return gct_modules.onehot_ignore(task_gt, self.args.num_classes, ignore_index=self.args.ignore_index)

    # ---- SSL_CCT (task/sseg/func.py:216-253)
    def sslcct_activate_ad_preds(self, ad_preds):
        return [torch.softmax(p, dim=1) for p in ad_preds]

    def sslcct_ad_in_channels(self):
        arch = self.args.models['model'] if hasattr(self.args, 'models') else 'pspnet'
        if arch == 'pspnet':
            return 512
        if arch == 'deeplabv2':
            return 2048
        raise NotImplementedError("sslcct_ad_in_channels: model '%s'" % arch)

    def sslcct_ad_out_channels(self):
        return self.args.num_classes

    def sslcct_ad_upsample_scale(self):
        return 8
