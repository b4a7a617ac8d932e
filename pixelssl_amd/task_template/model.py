"""Task-model plugin ABI (pixelssl/task_template/model.py:31-85): `TaskModel(args)` with `.model`,
`.param_groups` and `forward(inp: tuple) -> (resulter: dict, debugger: dict)` where
resulter['pred'] / resulter['activated_pred'] are tuples of tensors."""
import torch.nn as nn


def add_parser_arguments(parser):
    pass


def task_model():
    return TaskModel


class TaskModel(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.model = None
        self.param_groups = []

    def forward(self, inp):
        raise NotImplementedError
