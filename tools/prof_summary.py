#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel CSV + print the top rows;
`--one-step <db> [out.txt] [header]`: breakdown of one steady-state training step (window between optimizer launches)."""
import os
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


def one_step(db_path, out_txt=None, marker="sgd_kernel", per_step=2, header=""):
    """Kernel breakdown of ONE steady-state training step: the window between the last launches of `marker` of two
    consecutive steps (`per_step` marker launches per step: one per optimizer lr group), plus the union of all kernel
    intervals (GPU-busy time) and the per-stream busy time / gaps."""
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if not marks and marker == "sgd_kernel":      # round 5: the fused update kernel replaces the per-group SGD launches
        marker = "sgd_ema_pack_kernel"
        marks = [i for i, r in enumerate(rows) if marker in r[0]]
    # marker launches per step: MT / SupOnly have two (one per lr group); algorithms with several optimizers (AdvSSL, GCT, CCT)
    # have more -- derive it from the step count of the traced command line (--steps A --warmup B in the header)
    m = re.search(r"--steps (\d+) --warmup (\d+)", header or "")
    if m:
        nsteps = int(m.group(1)) + int(m.group(2))
        if nsteps > 0 and len(marks) % nsteps == 0 and len(marks) >= nsteps:
            per_step = len(marks) // nsteps
    ends = marks[per_step - 1::per_step]
    a, b = ends[-3], ends[-2]
    win = rows[a + 1:b + 1]
    wall = (win[-1][2] - win[0][1]) / 1e6
    acc = {}
    for n, s, e, _, _ in win:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n).split("(")[0]
        d = acc.setdefault(n, [0, 0])
        d[0] += 1
        d[1] += e - s
    tot = sum(v[1] for v in acc.values())
    iv = sorted((r[1], r[2]) for r in win)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            busy += ce - cs
            cs, ce = s, e
    busy += ce - cs
    lines = ["# " + header] if header else []
    lines += ["# ONE steady-state step (between two '%s' launches): wall window %.3f ms, SUM of kernel durations %.3f ms "
              "(streams overlap), union of kernel intervals %.3f ms, %d launches" % (marker, wall, tot / 1e6, busy / 1e6, len(win))]
    per = {}
    for r in win:
        per.setdefault((r[3], r[4]), []).append(r)
    for k, v in per.items():
        gaps = [v[i + 1][1] - v[i][2] for i in range(len(v) - 1)]
        small = [g for g in gaps if 0 <= g < 20000]
        lines.append("# stream %s: %d kernels, busy %.3f ms, gaps < 20 us: %d (sum %.3f ms)"
                     % (k, len(v), sum(r[2] - r[1] for r in v) / 1e6, len(small), sum(small) / 1e6))
    # phase timeline of the step: per stream first start / last end (ms from the window start), and on every stream the
    # start of a few marker kernels (the seam between forward and backward, the optimizer)
    t0 = win[0][1]
    for k, v in per.items():
        lines.append("# stream %s: runs %.3f .. %.3f ms" % (k, (v[0][1] - t0) / 1e6, (v[-1][2] - t0) / 1e6))
    for mk in ("head_loss", "upsample_softmax_fwd", "ce_mse_bwd", "upsample_bwd", "maxpool_bwd", "sgd_kernel", "ema_kernel", "pack_fwd", "stem_patches"):
        hits = [r for r in win if mk in r[0]]
        if hits:
            lines.append("# marker %-22s: %s" % (mk, ", ".join("%.3f-%.3f ms (stream %s)" % ((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, r[3]) for r in hits[:4])))
    # utilisation over time: number of kernels in flight, sampled per 0.25 ms
    step = 250000
    nb = int((win[-1][2] - t0) // step) + 1
    occ = [0.0] * nb
    for r in win:
        a, b = r[1] - t0, r[2] - t0
        for i in range(int(a // step), min(nb - 1, int(b // step)) + 1):
            lo, hi = max(a, i * step), min(b, (i + 1) * step)
            if hi > lo:
                occ[i] += (hi - lo) / step
    lines.append("# kernels in flight per 0.25 ms bucket: " + " ".join("%.1f" % o for o in occ))
    lines.append("Name,Calls,TotalDurationNs,AverageNs,PercentageOfSum")
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%.2f' % (n, c, t, t / c, 100 * t / tot))
    if out_txt:
        open(out_txt, "w").write("\n".join(lines) + "\n")
        # the same step as figures a program can read (bench.py reports them as `step_trace`): launches, sum of kernel durations,
        # the contraction families and everything else
        import json
        contr = lambda n: any(k in n for k in ("conv_dma_kernel", "conv_halo_kernel", "conv_igemm_kernel", "conv_wgrad"))
        c_ns = sum(t for n, (c, t) in acc.items() if contr(n))
        dma = [(c, t) for n, (c, t) in acc.items() if "conv_dma_kernel" in n or "conv_igemm_kernel" in n or "conv_halo_kernel" in n]
        json.dump({"source": header, "wall_window_ms": round(wall, 3), "launches_per_step": len(win), "kernel_sum_ms": round(tot / 1e6, 3),
                   "kernel_union_ms": round(busy / 1e6, 3), "contraction_ms": round(c_ns / 1e6, 3),
                   "noncontraction_ms": round((tot - c_ns) / 1e6, 3), "conv_dma_ms": round(sum(t for _, t in dma) / 1e6, 3),
                   "conv_dma_launches": sum(c for c, _ in dma)},
                  open(os.path.splitext(out_txt)[0] + ".json", "w"), indent=1)
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one-step":
        one_step(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, header=sys.argv[4] if len(sys.argv) > 4 else "")
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
