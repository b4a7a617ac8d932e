"""Hardware-queue-aware stream placement (csrc/streams.hip, pixelssl_amd/streams.py)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import sys, torch
sys.path.insert(0, %r)
from pixelssl_amd import streams
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]          # shift HIP's stream -> queue dealing
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(1, device="cuda")
n = streams.init()
roles = [streams.role_stream(r) for r in (streams.SIDE, streams.WGRAD, streams.AUX)]
main = torch.cuda.current_stream()
bufs = {}
for st in [main] + roles:                       # one buffer per stream, touched once: the probe itself must not allocate
    with torch.cuda.stream(st):
        bufs[st.cuda_stream] = torch.zeros(8, device="cuda")
        bufs[st.cuda_stream].add_(1)
torch.cuda.synchronize()
def overlaps(a, b):
    # a spinning kernel on a, a tiny one on b: did the tiny one finish while a was still busy?  (best of two tries)
    for _ in range(2):
        torch.cuda.synchronize()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(a):
            torch.cuda._sleep(int(2.0e9 * 5e-3))          # ~5 ms
            ea.record()
        with torch.cuda.stream(b):
            bufs[b.cuda_stream].add_(1)
            eb.record()
        eb.synchronize()
        free = not ea.query()
        torch.cuda.synchronize()
        if free:
            return True
    return False
pairs = [(main, r) for r in roles] + [(roles[i], roles[j]) for i in range(3) for j in range(i + 1, 3)]
print("RESULT", n, sum(int(overlaps(a, b)) for a, b in pairs), len(pairs), len({r.cuda_stream for r in roles}))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [0, 1, 3])
def test_role_streams_overlap_whatever_was_created_before(extra):
    """The streams of the three roles and the main stream sit on four different hardware queues -- every pair overlaps -- however
    many streams the process created before the pool was built (without placement, one extra stream at start-up cost the MT
    step 2.6 ms).  Fresh process per case: the pool is built once per process."""
    out = subprocess.run([sys.executable, "-c", _PROBE % ROOT, str(extra)], capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert out.returncode == 0 and line, out.stdout[-1500:] + out.stderr[-1500:]
    n, ok, pairs, distinct = (int(v) for v in line[0].split()[1:])
    assert n == 3 and distinct == 3, line[0]            # three queues besides the main stream's
    assert ok == pairs, line[0]                          # all six pairs ran concurrently


def test_placement_can_be_switched_off_and_needs_a_gpu(monkeypatch):
    from pixelssl_amd import streams
    if not torch.cuda.is_available():
        assert streams.role_stream(streams.SIDE) is None          # nothing to place on a host without a GPU
