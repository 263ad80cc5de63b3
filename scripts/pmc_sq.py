#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters of one `rocprofv3 --pmc … --kernel-trace` database.
    python scripts/pmc_sq.py run_results.db [name-filter]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    if flt in k:
        acc[k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]][c].append(v)
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
