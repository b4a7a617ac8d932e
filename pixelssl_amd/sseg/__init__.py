"""The `sseg` task plugins (task/sseg/{model,criterion,func,data}.py) on the MI355X engine."""
from . import model, criterion, func, data
