"""task/sseg/model.py on the MI355X engine: `DeepLabV2(args)` is a TaskModel whose `.model` is the
libpixelhip-executed network with the reference's parameter names (checkpoint compatible)."""
import torch

from ..task_template import model as model_template
from ..utils import logger, cmd
from ..engine import DeepLabV2Core, PSPNetCore
from .. import dist as pdist


def add_parser_arguments(parser):
    model_template.add_parser_arguments(parser)
    parser.add_argument('--output-stride', type=int, default=16, help='sseg - output stride of the ResNet backbone')
    parser.add_argument('--backbone', type=str, default='resnet101', help='sseg - architecture of the backbone network')
    parser.add_argument('--freeze-bn', type=cmd.str2bool, default=False,
                        help='sseg - if true, the statistics in BatchNorm will not be updated')
    parser.add_argument('--engine-dtype', type=str, default='bf16',
                        help='sseg/amd - arithmetic of the HIP engine: bf16 (throughput) or fp32 (exact parity)')
    parser.add_argument('--pretrained-backbone', type=str, default=None,
                        help="sseg/amd - backbone weights: a state-dict file of the trunk, a model-zoo URL, or 'default' for "
                             "the reference's URL of the chosen backbone (task/sseg/model.py:70-76); None = random init "
                             "(the reference always downloads: this engine has to run without a network)")


def deeplabv2():
    return DeepLabV2


def pspnet():
    return PSPNet


class _Resulter(dict):
    """resulter dict whose backbone latent ('sslcct_ad_inp') is converted NHWC->NCHW only when read."""

    def __init__(self, latent_fn):
        super().__init__()
        self._latent_fn = latent_fn

    def __missing__(self, key):
        if key == 'sslcct_ad_inp':
            self[key] = self._latent_fn()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key == 'sslcct_ad_inp' or dict.__contains__(self, key)

    def keys(self):
        return list(dict.keys(self)) + ([] if dict.__contains__(self, 'sslcct_ad_inp') else ['sslcct_ad_inp'])


# task/sseg/model.py:70-76,88-93: the weights the reference downloads for each backbone name
PRETRAINED_BACKBONE_URLS = {'resnet50': 'https://download.pytorch.org/models/resnet50-19c8e357.pth',
                            'resnet101': 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth',
                            'resnet101-coco': 'http://vllab1.ucmerced.edu/~whung/adv-semi-seg/resnet101COCO-41f33a49.pth'}


def _maybe_load_pretrained(task_model, args):
    src = getattr(args, 'pretrained_backbone', None)
    if not src:
        return
    if src == 'default':
        src = PRETRAINED_BACKBONE_URLS[args.backbone]
    taken = task_model.model.load_pretrained_backbone(src)
    logger.log_info('  pretrained backbone: {0} tensors from {1}\n'.format(len(taken), src))


class _DeferredResulter(dict):
    """resulter of a deferred forward pass: 'pred' / 'activated_pred' are materialised (detached) when first read."""

    def __init__(self, head):
        super().__init__()
        self.head = head

    def __missing__(self, key):
        if key in ('pred', 'activated_pred'):
            logits, prob = self.head.materialize()
            self['pred'], self['activated_pred'] = (logits,), (prob,)
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key in ('pred', 'activated_pred') or dict.__contains__(self, key)

    def keys(self):
        return list(dict.keys(self)) + [k for k in ('pred', 'activated_pred') if not dict.__contains__(self, k)]


class _DeferredForward:
    """Engine extension of the sseg task models (not in the reference): forward_deferred(inp) stops at the
    low-resolution logits -> engine.DeferredHead (None when unsupported); SSL algorithms whose losses the fused seam
    covers (SupOnly, MT) use it when nothing reads the full-resolution predictions."""

    def forward_deferred(self, inp):
        if not len(inp) == 1:
            logger.log_err('Semantic segmentation models require only one input\n'
                           'However, {0} inputs are given\n'.format(len(inp)))
        return self.model.forward_deferred(inp[0])

    def forward_deferred_shared(self, inp, prepared=None, borrow=None):
        """forward_deferred over an input tensor that a second network reads as well: `prepared` = this network's own
        engine.prepare_patches token (its stem patches are already in the arena), `borrow` = the other network's token (this
        pass reads those patches instead of writing its own)"""
        if borrow is not None:
            self.model.borrow_patches(borrow, inp[0])
        return self.model.forward_deferred(inp[0], prepared=prepared)


class DeepLabV2(model_template.TaskModel, _DeferredForward):
    def __init__(self, args):
        super().__init__(args)
        if args.backbone not in ('resnet50', 'resnet101', 'resnet101-coco'):
            logger.log_err('DeepLabV2 does not support the backbone: {0}\n'.format(args.backbone))
        dtype = getattr(args, 'engine_dtype', 'bf16')
        self.model = DeepLabV2Core(backbone=args.backbone, output_stride=args.output_stride,
                                   num_classes=args.num_classes, device=pdist.local_device(),
                                   engine_dtype=torch.float32 if dtype in ('fp32', 'f32') else torch.bfloat16,
                                   freeze_bn=args.freeze_bn)
        _maybe_load_pretrained(self, args)
        self.param_groups = [{'params': self.model.get_1x_lr_params(), 'lr': self.args.lr},
                             {'params': self.model.get_10x_lr_params(), 'lr': self.args.lr * 10}]

    def forward(self, inp):
        if not len(inp) == 1:
            logger.log_err('Semantic segmentation model DeepLab requires only one input\n'
                           'However, {0} inputs are given\n'.format(len(inp)))
        pred, prob, latent_fn = self.model(inp[0])
        resulter = _Resulter(latent_fn)
        resulter['pred'] = (pred,)
        resulter['activated_pred'] = (prob,)
        resulter['ssls4l_rc_inp'] = pred
        return resulter, {}


class PSPNet(model_template.TaskModel, _DeferredForward):
    """task/sseg/model.py:83-125: PSPNet TaskModel with three parameter groups (backbone lr, psp / decoder lr x10);
    'sslcct_ad_inp' is the pyramid module's 512-channel output."""

    def __init__(self, args):
        super().__init__(args)
        if args.backbone not in ('resnet50', 'resnet101', 'resnet101-coco'):
            logger.log_err('PSPNet does not support the backbone: {0}\n'.format(args.backbone))
        dtype = getattr(args, 'engine_dtype', 'bf16')
        self.model = PSPNetCore(backbone=args.backbone, output_stride=args.output_stride,
                                num_classes=args.num_classes, device=pdist.local_device(),
                                engine_dtype=torch.float32 if dtype in ('fp32', 'f32') else torch.bfloat16,
                                freeze_bn=args.freeze_bn)
        _maybe_load_pretrained(self, args)
        self.param_groups = [{'params': self.model.get_backbone_params(), 'lr': self.args.lr},
                             {'params': self.model.get_psp_params(), 'lr': self.args.lr * 10},
                             {'params': self.model.get_decoder_params(), 'lr': self.args.lr * 10}]

    def forward(self, inp):
        if not len(inp) == 1:
            logger.log_err('Semantic segmentation model PSPNet requires only one input\n'
                           'However, {0} inputs are given\n'.format(len(inp)))
        pred, prob, latent_fn = self.model(inp[0])
        resulter = _Resulter(latent_fn)
        resulter['pred'] = (pred,)
        resulter['activated_pred'] = (prob,)
        resulter['ssls4l_rc_inp'] = pred
        return resulter, {}
