from . import func, optimizer, lrer, module, data
from .lrer import VALID_LRER
from .optimizer import VALID_OPTIMIZER
from .module import SynchronizedBatchNorm2d, patch_replication_callback, GaussianNoiseLayer
