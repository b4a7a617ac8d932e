from .constant import REGRESSION, CLASSIFICATION
from . import logger, cmd, tool
from .logger import log_info, log_warn, log_err
from .cmd import str2bool, str2intlist
