"""task/sseg/criterion.py on the fused HIP cross-entropy kernel."""
from ..task_template import criterion as criterion_template
from ..utils import logger
from .. import functional as PF


def add_parser_arguments(parser):
    criterion_template.add_parser_arguments(parser)


def sseg_criterion():
    return CommonSSEGCriterion


class CommonSSEGCriterion(criterion_template.TaskCriterion):
    """Per-sample CE with ignore_index; mean over ALL H*W pixels (task/sseg/criterion.py:24-38).
    NOTE: `pred` is not activated (logits)."""

    def forward(self, pred, gt, inp):
        if len(pred) != 1 or len(gt) != 1 or len(inp) != 1:
            logger.log_err('DeepLab criterion for semantic segmentation requires\t=>\t'
                           'len(pred) == 1 \t len(gt) == 1 \t len(inp) == 1\n')
        return PF.cross_entropy_per_sample(pred[0], gt[0], self.args.ignore_index)

    def with_consistency(self, pred, gt, ce_values, target, lo, hi):
        """Engine extension (not in the reference): this criterion on the first len(ce_values) samples of `pred` AND
        nn.MSELoss()(pred[lo:hi], target[lo:hi]) with one fused backward; `ce_values` = self.forward(...) computed
        under no_grad on the same samples.  SSL algorithms use it when the criterion offers it and fall back to the
        two separate losses otherwise."""
        return PF.task_consistency(pred[0], gt[0], ce_values, target, lo, hi, self.args.ignore_index)
