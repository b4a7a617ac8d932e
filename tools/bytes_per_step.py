#!/usr/bin/env python
"""HBM bytes of ONE training step = sum over kernels of (PMC traffic per launch) x (launches per step).

    python tools/bytes_per_step.py profiles/traffic.json profiles/rNN_step_breakdown.txt [out.txt] [--json out.json]

traffic.json: tools/pmc_traffic.py (two rocprofv3 --pmc passes, (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch, kernel names
without template arguments); step_breakdown.txt: tools/prof_summary.py --one-step (rocprofv3 --kernel-trace, one steady-state
step: launches and summed duration per kernel instantiation).  The per-launch traffic of a kernel family is an average over all
of its launches in the PMC run, so the product is exact when that run and the traced step launch the same mix (same command,
same tuned tiles) and an estimate otherwise.  Prints a table (bytes, share, the family's time in the step and its effective
bandwidth) and the step totals against 8 TB/s (MI355X_MICROARCH.md) and the 6.29 TB/s that guide measured."""
import json
import re
import sys


def family(name):
    n = re.sub(r"\(anonymous namespace\)::|void |pxl_dma::", "", name.strip().strip('"'))
    return n.split("<")[0].split("(")[0]


def main(traffic_json, breakdown_txt, out=None, json_out=None):
    tj = json.load(open(traffic_json))["kernels"]
    wall_ms = None
    fam = {}
    for line in open(breakdown_txt):
        if line.startswith("#"):
            m = re.search(r"wall window ([0-9.]+) ms", line)
            if m:
                wall_ms = float(m.group(1))
            continue
        m = re.match(r'"(.+)",(\d+),(\d+),', line)
        if not m:
            continue
        f = family(m.group(1))
        d = fam.setdefault(f, [0, 0])
        d[0] += int(m.group(2))
        d[1] += int(m.group(3))
    rows, total, covered_ns, all_ns = [], 0.0, 0, 0
    for f, (n, ns) in fam.items():
        all_ns += ns
        t = tj.get(f)
        if t is None:
            continue
        b = t["traffic_bytes_per_launch"] * n
        rows.append((b, f, n, ns, t["traffic_bytes_per_launch"]))
        total += b
        covered_ns += ns
    rows.sort(reverse=True)
    lines = ["# HBM bytes per step: %s x launches per step of %s" % (traffic_json, breakdown_txt),
             "%-34s %8s %12s %10s %7s %10s %10s" % ("kernel family", "launches", "MB/launch", "GB/step", "share", "ms in step", "TB/s")]
    for b, f, n, ns, per in rows:
        lines.append("%-34s %8d %12.2f %10.3f %6.1f%% %10.3f %10.2f" % (f, n, per / 1e6, b / 1e9, 100 * b / total, ns / 1e6, b / max(ns, 1) / 1e3))
    lines.append("# total %.2f GB per step over kernels holding %.1f %% of the step's kernel time" % (total / 1e9, 100.0 * covered_ns / max(all_ns, 1)))
    if wall_ms:
        tbs = total / 1e12 / (wall_ms * 1e-3)
        lines.append("# step wall window %.3f ms -> %.2f TB/s = %.2f of 8 TB/s (%.2f of the 6.29 TB/s measured in MI355X_MICROARCH.md); "
                     "at this traffic the step cannot be shorter than %.2f ms (8 TB/s)" % (wall_ms, tbs, tbs / 8.0, tbs / 6.29, total / 8e12 * 1e3))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)
    if json_out:
        json.dump({"bytes_per_step": int(total), "wall_ms": wall_ms, "source": [traffic_json, breakdown_txt],
                   "families": {f: {"launches": n, "bytes": int(b)} for b, f, n, ns, per in rows}}, open(json_out, "w"), indent=1)


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if x != "--json"]
    jo = None
    if "--json" in sys.argv:
        jo = sys.argv[sys.argv.index("--json") + 1]
        a.remove(jo)
    main(a[0], a[1], a[2] if len(a) > 2 else None, jo)
