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


def conv_dma_variants(d, keep=319):
    """the last `keep` conv_dma_kernel dispatches grouped by epilogue variant: template arguments <BM, BN, WM, WN, stages, gather,
    ablation, BN-on-load, trace, EM>, EM bits: 1 addend, 2 bias, 4 statistics, 8 BatchNorm-backward sums, 16 join mask, 32 ReLU mask
    of that BatchNorm (conv_dma_kernel.h) -> {(EM, BN-on-load): [counter values]}"""
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    sel = [r for r in rows if "conv_dma_kernel<" in r["Kernel_Name"]][-keep:]
    acc = defaultdict(list)
    for r in sel:
        m = re.search(r"conv_dma_kernel<([^>]*)>", r["Kernel_Name"])
        a = [x.strip() for x in m.group(1).split(",")] if m else []
        key = (a[9] if len(a) > 9 else "?", a[7] if len(a) > 7 else "?")
        acc[key].append(float(r["Counter_Value"]))
    return acc


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
    # conv_dma per epilogue variant (which variants carry the bytes beyond the operands)
    try:
        vf, vw = conv_dma_variants(fetch_dir), conv_dma_variants(write_dir)
        names = {"0": "plain", "1": "addend", "2": "bias", "4": "statistics", "5": "statistics + addend", "-2": "run-time flags (split-K ASPP)",
                 "29": "join backward: addend + sums + BN-backward + join mask", "44": "BN-backward sums + ReLU mask", "12": "BN-backward sums",
                 "45": "BN-backward sums + ReLU mask + addend", "13": "BN-backward sums + addend"}
        res["conv_dma_variants"] = {}
        for key in sorted(set(vf) & set(vw), key=lambda k: -(2 * sum(vf[k]) + sum(vw[k]))):
            nf, nw = len(vf[key]), len(vw[key])
            res["conv_dma_variants"]["EM=%s%s" % (key[0], ", BN-apply on load" if key[1] == "true" else "")] = {
                "what": names.get(key[0], "?"), "launches": nf, "fetch_kb_avg_raw": round(sum(vf[key]) / nf, 1),
                "write_kb_avg": round(sum(vw[key]) / nw, 1),
                "traffic_bytes_per_launch": int((2 * sum(vf[key]) / nf + sum(vw[key]) / nw) * 1024),
                "traffic_gb_total": round((2 * sum(vf[key]) + sum(vw[key]) * nf / nw) * 1024 / 1e9, 3)}
    except Exception as e:      # noqa: BLE001
        res["conv_dma_variants"] = "failed: %s" % e
    json.dump(res, open(out, "w"), indent=1)
    for k, v in list(res["kernels"].items())[:10]:
        print("%-34s n=%5d  traffic/launch %8.2f MB" % (k, v["launches"], v["traffic_bytes_per_launch"] / 1e6))
    if isinstance(res.get("conv_dma_variants"), dict):
        for k, v in res["conv_dma_variants"].items():
            print("   conv_dma %-28s n=%4d  %8.2f MB/launch  %6.3f GB  (%s)" % (k, v["launches"], v["traffic_bytes_per_launch"] / 1e6,
                                                                               v["traffic_gb_total"], v["what"]))


if __name__ == "__main__":
    main(*sys.argv[1:5])
