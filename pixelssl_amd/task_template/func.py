"""Task-function plugin ABI (pixelssl/task_template/func.py:20-261): hooks the SSL algorithms call
back into the task for metrics / visualisation / algorithm-specific conversions."""


def add_parser_arguments(parser):
    pass


def task_func():
    return TaskFunc


class TaskFunc:
    METRIC_STR = 'metric'

    def __init__(self, args):
        self.args = args

    def metrics(self, pred, gt, inp, meters, id_str=''):
        raise NotImplementedError

    def visualize(self, out_path, id_str='', inp=None, pred=None, gt=None):
        raise NotImplementedError
