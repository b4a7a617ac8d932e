#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV passes: per (kernel, grid) mean of every counter, one line each."""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(anonymous namespace\)::", "", row.get("Kernel_Name", ""))
            if "conv_" not in name:
                continue
            key = (name.split("(")[0][-60:], row.get("Grid_Size", ""), row.get("LDS_Block_Size", row.get("LDS_Block_Size_v", "")))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, cs in acc.items():
    print("== %s grid=%s lds=%s" % key)
    line = []
    for c, v in sorted(cs.items()):
        line.append("%s=%.4g" % (c, sum(v) / len(v)))
    print("   " + "  ".join(line))
