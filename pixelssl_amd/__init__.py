"""pixelssl_amd -- MI355X-native engine behind the PixelSSL `ssl_algorithm` / `task_template` plugin API.

Mirrors the reference package's public names for the hot path (`pixelssl/__init__.py`): `SSL_*`
constants, `ssl_algorithm`, `nn`, `model_template` / `criterion_template` / `func_template`,
`SynchronizedBatchNorm2d`, `log_*`, `str2bool`.  Everything heavy runs in libpixelhip.so
(hand-written HIP for gfx950); importing the package does not need a GPU, running a model does.
"""
from .utils import REGRESSION, CLASSIFICATION, log_info, log_warn, log_err, str2bool, str2intlist
from . import utils
from . import nn
from .nn import SynchronizedBatchNorm2d, patch_replication_callback, GaussianNoiseLayer
from . import ssl_algorithm
from .ssl_algorithm import SSL_NULL, SSL_MT, SSL_ADV, SSL_CUTMIX, SSL_GCT, SSL_CCT, SSL_S4L, SSL_ALGORITHMS
from . import task_template
from .task_template import model_template, criterion_template, func_template, data_template
from . import functional
from . import sseg

__version__ = '0.1.0'
