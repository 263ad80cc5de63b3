#!/usr/bin/env python3
"""Combine the FETCH_SIZE and WRITE_SIZE rocprofv3 passes (scripts/gpu_profile.sh) into HBM bytes per
launch per kernel.  Counters are reported in KiB; FETCH_SIZE is doubled per the gfx950 correction in
MI355X_MICROARCH.md (the counter ticks once per 64 B request but is scaled as if 32 B).
    python scripts/pmc_traffic.py fetch.db write.db ROWS > profiles/rNN_pmc_traffic.json"""
import json, sqlite3, sys


def avg(dbpath, counter):
    db = sqlite3.connect(dbpath)
    q = ("select kernel_name, avg(value) from counters_collection where counter_name=? "
         "group by kernel_name order by sum(value) desc")
    return {k: v for k, v in db.execute(q, (counter,))}


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {"rows": int(sys.argv[3]),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/gpu_profile.sh); "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction",
       "kernels": {}}
for k, f in fetch.items():
    w = write.get(k, 0.0)
    out["kernels"][k] = {"fetch_kb_raw": round(f, 3), "write_kb_raw": round(w, 3),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, sys.stdout, indent=1)
print()
