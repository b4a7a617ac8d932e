#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel CSV + print the top rows."""
import re
import sqlite3
import sys


def main(db_path, out_csv=None, header="", top=30):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    clean = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)
    if out_csv:
        with open(out_csv, "w") as f:
            if header:
                f.write("# " + header + "\n")
            f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
            for r in rows:
                f.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (clean(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    print("total kernel time %.3f ms" % (tot / 1e6))
    for r in rows[:top]:
        print("%-86s n=%6d total=%9.3f ms avg=%8.1f us %5.1f%%" % (clean(r[0])[:86], r[1], r[2] / 1e6, r[3] / 1e3, 100 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
