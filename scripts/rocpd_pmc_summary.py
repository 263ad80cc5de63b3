#!/usr/bin/env python3
"""Per-kernel average of each PMC counter in a rocprofv3 rocpd database (--pmc run).
    python scripts/rocpd_pmc_summary.py x_results.db > profiles/rNN_pmc.csv"""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in db.execute("pragma table_info(counters_collection)")]
w = csv.writer(sys.stdout)
name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
cnt_col = "counter_name" if "counter_name" in cols else None
val_col = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
if cnt_col is None or val_col is None:
    w.writerow(cols)
    for r in db.execute("select * from counters_collection limit 50"):
        w.writerow(r)
    sys.exit(0)
w.writerow(["kernel", "counter", "dispatches", "avg_value", "total_value"])
q = f"select {name_col}, {cnt_col}, count(*), avg({val_col}), sum({val_col}) from counters_collection group by {name_col}, {cnt_col} order by sum({val_col}) desc"
for r in db.execute(q):
    w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 3)])
