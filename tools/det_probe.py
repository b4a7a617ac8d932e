"""Run-to-run spread of the default (atomics in arbitrary order) mode against one PXL_DETERMINISTIC run of the same shallow-trunk
step (tests/test_deterministic.py): the bar of that comparison has to sit above it.  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_deterministic import _run_step
os.environ["PXL_DETERMINISTIC"] = "1"
ref = _run_step(torch.float32, True)
del os.environ["PXL_DETERMINISTIC"]
for k in range(5):
    loss, g, rs = _run_step(torch.float32, True)
    print("default run %d vs deterministic: rel %.3e  loss diff %.2e" % (k, ((g - ref[1]).norm() / ref[1].norm()).item(), abs(loss - ref[0]) / abs(loss)))
