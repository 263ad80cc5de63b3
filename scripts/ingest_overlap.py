#!/usr/bin/env python3
"""Copy / compute overlap of the chunked ingest from a rocprofv3 --kernel-trace --memory-copy-trace database
(scripts/bench_ingest.py under rocprofv3):  python scripts/ingest_overlap.py run_results.db > profiles/rNN_ingest_overlap.json
Chunked transfers are the copies of ≤ 64 MiB; whole-column copies (the plain h2d / d2h / unpipelined legs) are listed apart.
For every direction: busy time (union of intervals), and how much of the H2D busy time had a kernel / a D2H copy running at
the same moment.  `timeline_excerpt` = 24 consecutive events from the middle of the pipelined Add leg, times in µs from its start."""
import json, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
copies = list(db.execute("select start, end, name, size, stream_name from memory_copies order by start"))
kernels = list(db.execute("select start, end, name, stream from kernels order by start"))


def union(iv):
    iv = sorted(iv); out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def inter(x, y):
    i = j = 0; t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            t += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return t


chunk = [c for c in copies if c[3] <= (64 << 20) and c[3] >= (1 << 20)]
whole = [c for c in copies if c[3] > (64 << 20)]
h2d = union([(c[0], c[1]) for c in chunk if "HOST_TO_DEVICE" in c[2]])
d2h = union([(c[0], c[1]) for c in chunk if "DEVICE_TO_HOST" in c[2]])
short = lambda n: n.replace("void (anonymous namespace)::", "").split("(")[0][:60]
# kernels inside the span of the chunked copies only (the legs that pipeline)
span = (min(c[0] for c in chunk), max(c[1] for c in chunk)) if chunk else (0, 0)
kin = [k for k in kernels if k[1] >= span[0] and k[0] <= span[1]]
ker = union([(k[0], k[1]) for k in kin])
out = {
    "source": "rocprofv3 --kernel-trace --memory-copy-trace --stats -- python scripts/bench_ingest.py 27 (MI355X, 2^27-row columns, 32 MiB chunks, 3 slots)",
    "chunked_copies": {"h2d": sum(1 for c in chunk if "HOST_TO_DEVICE" in c[2]), "d2h": sum(1 for c in chunk if "DEVICE_TO_HOST" in c[2]),
                       "h2d_GiB": round(sum(c[3] for c in chunk if "HOST_TO_DEVICE" in c[2]) / 2**30, 2),
                       "d2h_GiB": round(sum(c[3] for c in chunk if "DEVICE_TO_HOST" in c[2]) / 2**30, 2),
                       "streams": sorted({c[4] for c in chunk})},
    "busy_ms": {"h2d": round(total(h2d) / 1e6, 3), "d2h": round(total(d2h) / 1e6, 3), "kernels": round(total(ker) / 1e6, 3)},
    "h2d_rate_while_busy_GB/s": round(sum(c[3] for c in chunk if "HOST_TO_DEVICE" in c[2]) / max(total(h2d), 1), 2),
    "overlap_ms": {"kernels_inside_h2d": round(inter(ker, h2d) / 1e6, 3), "d2h_inside_h2d": round(inter(d2h, h2d) / 1e6, 3)},
    "fraction_of_kernel_time_hidden_under_uploads": round(inter(ker, h2d) / max(total(ker), 1), 3),
    "fraction_of_download_time_hidden_under_uploads": round(inter(d2h, h2d) / max(total(d2h), 1), 3),
    "kernel_streams": sorted({k[3] for k in kin}),
    "whole_column_copies": [{"dir": "h2d" if "HOST_TO_DEVICE" in c[2] else "d2h", "MiB": c[3] >> 20, "ms": round((c[1] - c[0]) / 1e6, 3),
                             "GB/s": round(c[3] / (c[1] - c[0]), 2)} for c in whole[:12]],
}
# timeline excerpt: middle of the run of 32 MiB D2H copies (only the Add leg downloads full chunks)
dl = [c for c in chunk if "DEVICE_TO_HOST" in c[2] and c[3] == (32 << 20)]
if dl:
    mid = dl[len(dl) // 2][0]
    ev = [(c[0], c[1], ("H2D " if "HOST_TO_DEVICE" in c[2] else "D2H ") + f"{c[3] >> 20} MiB", c[4]) for c in chunk] + [(k[0], k[1], "K   " + short(k[2]), k[3]) for k in kin]
    ev.sort()
    i0 = max(0, next(i for i, e in enumerate(ev) if e[0] >= mid) - 8)
    t0 = ev[i0][0]
    out["timeline_excerpt"] = [{"start_us": round((e[0] - t0) / 1e3, 1), "end_us": round((e[1] - t0) / 1e3, 1), "what": e[2], "stream": e[3]} for e in ev[i0:i0 + 24]]
print(json.dumps(out, indent=1))
