#!/usr/bin/env python
"""HBM traffic per launch of the contraction kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
runs with --kernel-trace only, MI355X_MICROARCH.md "HBM"): FETCH_SIZE is reported in KB and, on gfx950, counts 64 B
per 128-B request for wide coalesced reads -> doubled; WRITE_SIZE (KB) is taken as is.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/traffic.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


TAIL = {"conv_dma_kernel": 250, "conv_wgrad_dma_kernel": 90, "conv_igemm_kernel": 6}     # launches of ONE step at least


def per_kernel(d, tail=True):
    """kernel name -> counter values in dispatch order; with `tail`, only the last launches of the contraction kernels are
    kept (the final training step: the run's earlier launches are the autotune candidates of every tile configuration)"""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    acc = defaultdict(list)
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::|void |pxl_dma::", "", r["Kernel_Name"]).split("<")[0].split("(")[0]
        acc[n].append(float(r["Counter_Value"]))
    if tail:
        for k, keep in TAIL.items():
            if k in acc:
                acc[k] = acc[k][-keep:]
    return acc


def main(fetch_dir, write_dir, out, how="tools/gpu_call.sh <tag> traffic"):
    fe, wr = per_kernel(fetch_dir), per_kernel(write_dir)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-kernel-events --no-miou, MT 8x513x513 bf16, autotuned tiles; contraction kernels: "
                     "the last launches of the run only (the final training step)",
           "measured": "two separate rocprofv3 --pmc passes of this command, " + how,
           "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts half of wide coalesced reads)",
           "kernels": {}}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * sum(fe.get(k, [0])) + sum(wr.get(k, [0])))):
        nf, nw = len(fe.get(k, [])), len(wr.get(k, []))
        if not nf or not nw:
            continue
        f_avg, w_avg = sum(fe[k]) / nf, sum(wr[k]) / nw
        res["kernels"][k] = {"launches": nf, "fetch_kb_avg_raw": round(f_avg, 1), "write_kb_avg": round(w_avg, 1),
                             "traffic_bytes_per_launch": int((2 * f_avg + w_avg) * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in list(res["kernels"].items())[:10]:
        print("%-34s n=%5d  traffic/launch %8.2f MB" % (k, v["launches"], v["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
