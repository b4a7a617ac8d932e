"""The `sseg` task plugins (task/sseg/{model,criterion}.py) on the MI355X engine."""
from . import model, criterion, func
