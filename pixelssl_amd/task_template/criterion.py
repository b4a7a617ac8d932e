"""Task-criterion plugin ABI (pixelssl/task_template/criterion.py:32-78):
`TaskCriterion(args).forward(pred, gt, inp) -> Tensor[batch]` (sample-level losses)."""
import torch.nn as nn


def add_parser_arguments(parser):
    pass


def task_criterion():
    return TaskCriterion


class TaskCriterion(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args

    def forward(self, pred, gt, inp):
        raise NotImplementedError
