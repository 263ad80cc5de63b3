#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (the default output of
`rocprofv3 --kernel-trace --stats`) as CSV: per-kernel calls / total / average (µs) / %.

    python scripts/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "grid_x", "workgroup_x", "vgpr", "sgpr", "lds_bytes"])
meta = {}
for name, gx, wx, vg, sg, lds in db.execute(
        "select name, max(grid_x), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name"):
    meta[name] = (gx, wx, vg, sg, lds)
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    m = meta.get(name, ("", "", "", "", ""))
    w.writerow([name, calls, round(total, 3), round(avg, 3), round(pct, 3), *m])
