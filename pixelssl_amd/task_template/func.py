"""Task-function plugin ABI (pixelssl/task_template/func.py:20-261): hooks the SSL algorithms call back into the task for
metrics / visualisation / algorithm-specific conversions.  Same names, argument meaning, defaults and error behaviour as
the reference template: `metrics` / `visualize` warn when a task does not implement them, the SSL_ADV / SSL_GCT
conversion hooks default to the identity, the size hooks raise NotImplementedError."""
from ..utils import logger


def add_parser_arguments(parser):
    pass


def task_func():
    return TaskFunc


class TaskFunc:
    METRIC_STR = 'metric'

    def __init__(self, args=None):
        self.args = args

    # ---- all tasks (func.py:42-74)
    def metrics(self, pred, gt, inp, meters, id_str=''):
        logger.log_warn('No implementation of the \'metrics\' function for current task.\n'
                        'Please implement it in \'task/xxx/func.py\'.\n')

    def visualize(self, out_path, id_str='', inp=None, pred=None, gt=None):
        logger.log_warn('No implementation of the \'visulize\' function for current task.\n'
                        'Please implement it in \'task/xxx/func.py\'.\n')

    # ---- SSL_ADV (func.py:80-138)
    def ssladv_fcd_in_channels(self):
        raise NotImplementedError

    def ssladv_preprocess_fcd_criterion(self, fcd_pred, task_gt, is_real):
        raise NotImplementedError

    def ssladv_convert_task_gt_to_fcd_input(self, task_gt):
        return task_gt

    # ---- SSL_GCT (func.py:146-176)
    def sslgct_fd_in_channels(self):
        raise NotImplementedError

    def sslgct_prepare_task_gt_for_fdgt(self, task_gt):
        return task_gt

    # ---- SSL_S4L (func.py:184-196)
    def ssls4l_rc_in_channels(self):
        raise NotImplementedError

    # ---- SSL_CCT (func.py:204-259)
    def sslcct_activate_ad_preds(self, ad_preds):
        raise NotImplementedError

    def sslcct_ad_in_channels(self):
        raise NotImplementedError

    def sslcct_ad_out_channels(self):
        raise NotImplementedError

    def sslcct_ad_upsample_scale(self):
        raise NotImplementedError
