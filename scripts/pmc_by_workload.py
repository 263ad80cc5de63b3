#!/usr/bin/env python3
"""Cut the rocprofv3 databases of scripts/prof_workloads.py into one row per (workload, kernel): dispatches, average
duration (kernel-trace pass), FETCH_SIZE and WRITE_SIZE per launch (their own --pmc passes, raw KiB as the tool reports them).
Segments are the dispatches between a PAIR of marker launches (kleene_kernel); set-up and warm-up lie outside.
    python scripts/pmc_by_workload.py workloads.json kt.db fetch.db write.db > profiles/rNN_pmc_by_workload.json
`hbm_bytes` applies the gfx950 rule of MI355X_MICROARCH.md (FETCH_SIZE tallies a 128-byte request as 64) only where the
calibration rows justify it — see the "calibration" block this script emits."""
import json, sqlite3, sys

wl = json.load(open(sys.argv[1]))
names = [w["name"] for w in wl["workloads"]]
reps = wl["reps"]


def segments(dbpath, value_sql):
    """→ list over workloads of {kernel: [values...]} for the timed (post-marker) dispatches"""
    db = sqlite3.connect(dbpath)
    rows = list(db.execute(value_sql))
    segs, cur = [], None
    for kname, val in rows:
        if "kleene_kernel" in kname:   # markers come in pairs around the timed launches of one workload
            if cur is None:
                cur = {}
            else:
                segs.append(cur)
                cur = None
            continue
        if cur is not None:
            cur.setdefault(kname, []).append(val)
    return segs


def short(k):
    return k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


kt = segments(sys.argv[2], "select name, duration from kernels order by start")
fe = segments(sys.argv[3], "select kernel_name, value from counters_collection where counter_name='FETCH_SIZE' order by start")
wr = segments(sys.argv[4], "select kernel_name, value from counters_collection where counter_name='WRITE_SIZE' order by start")
assert len(kt) == len(fe) == len(wr) == len(names), (len(kt), len(fe), len(wr), len(names))
out = {"rows": wl["rows"], "hash_rows": wl["hash_rows"], "reps_per_workload": reps,
       "note": "one rocprofv3 pass per counter (--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE); per (workload, kernel): "
               "launches per call, avg µs per launch, raw counter KiB per launch.  fetch_kib_raw counts 64 B per L2→fabric read request "
               "(128-byte requests included: see calibration).",
       "workloads": {}}
for i, name in enumerate(names):
    ks = {}
    for k, durs in kt[i].items():
        if "rocclr" in k:
            continue
        f, w = fe[i].get(k, []), wr[i].get(k, [])
        ks[short(k)] = {"launches_per_call": round(len(durs) / reps, 2), "avg_us": round(sum(durs) / len(durs) / 1e3, 2),
                        "fetch_kib_raw": round(sum(f) / max(len(f), 1), 1), "write_kib": round(sum(w) / max(len(w), 1), 1)}
    tot_us = sum(v["avg_us"] * v["launches_per_call"] for v in ks.values())
    tot_f = sum(v["fetch_kib_raw"] * v["launches_per_call"] for v in ks.values())
    tot_w = sum(v["write_kib"] * v["launches_per_call"] for v in ks.values())
    n = wl["hash_rows"] if (name.startswith("dict") or name.startswith("hash")) else wl["rows"]
    out["workloads"][name] = {"kernels": ks, "per_call": {"kernel_us": round(tot_us, 1), "fetch_raw_B_per_row": round(tot_f * 1024 / n, 2),
                                                          "write_B_per_row": round(tot_w * 1024 / n, 2)}}
cal = {}
for nm in ("add_int64", "take_stride8_calib", "take_stride16_calib", "take_random_nulls10_direct"):
    if nm in out["workloads"]:
        cal[nm] = out["workloads"][nm]["per_call"]
out["calibration"] = {"rows": cal, "reading": "add_int64 reads 16 B/row with 16-byte lane loads: raw ≈ 8 → streaming reads are tallied at half (×2 to get bytes). "
                      "take_stride16 gathers one 8-byte value from every 128-byte line, take_stride8 from every 64-byte half line: equal raw B/row for both "
                      "means the fabric request is 64 bytes (raw = true bytes for gathers); stride8 at half of stride16 means 128-byte requests tallied at 64."}
json.dump(out, sys.stdout, indent=1)
print()
