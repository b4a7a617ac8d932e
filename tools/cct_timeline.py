"""Host and GPU timeline of ONE steady-state CCT training step (bench.py's workload), without a profiler attached: every mark
(pixelssl_amd/ssl_algorithm/ssl_cct.py: mark()) is a host time stamp plus an event recorded on the stream that is current at that
point -- `host` = when the host got there, `gpu` = when that stream's queue got there, both in ms from the step's first mark.

    python tools/cct_timeline.py [bench.py arguments, e.g. --decoders 11]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--algo", "cct"] + sys.argv[1:]
import bench  # noqa: E402


def main():
    import torch
    from pixelssl_amd import dist as pdist
    from pixelssl_amd.ssl_algorithm import ssl_cct
    from pixelssl_amd.utils.synthetic import synthetic_batch
    a = bench.parse()
    torch.cuda.set_device(pdist.local_device())
    pdist.init_from_env("nccl")
    dev = pdist.local_device()
    args = bench.make_args(a, 1)
    algo, cores = bench.build_algo(a, args)
    bench.condition(cores)
    batches = []
    for i in range(4):
        x, gt = synthetic_batch(a.lbs + a.ubs, a.size, a.lbs, seed=1234 + i)
        batches.append(((x.to(dev),), (gt.to(dev),)))
    step = bench.make_step(a, args, algo, batches)
    for it in range(12):
        step(it)
    # steady state: no synchronisation before the traced step (the host is wherever it is relative to the GPU)
    for rep in range(2):
        ssl_cct.TIMELINE = []
        step(12 + 2 * rep)
        tl, ssl_cct.TIMELINE = ssl_cct.TIMELINE, None
        step(13 + 2 * rep)
        torch.cuda.synchronize()
        h0, e0 = tl[0][1], tl[0][2]
        print("traced step %d: %-44s %9s %9s" % (rep, "mark", "host ms", "gpu ms"))
        for name, h, ev in tl:
            print("               %-44s %9.3f %9.3f" % (name, 1e3 * (h - h0), e0.elapsed_time(ev)))


if __name__ == "__main__":
    main()
