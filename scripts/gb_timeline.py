"""kernel timeline of ONE call per configuration out of a rocprofv3 kernel trace of scripts/bench_gb_mid.py
   python scripts/gb_timeline.py gpurun_out/prof_<tag>/run_results.db [call index within the reserve-1 f64 block]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
def short(n):
    m = re.search(r'(g[a-z]_\w+|\w+_kernel|__amd\w+)(<[^>]*>)?', n)
    return (m.group(0) if m else n)[:48]
ql = [i for i, r in enumerate(rows) if 'quicklook' in r[0]]
per_cfg = len(ql) // 6
want = [int(a) for a in sys.argv[2:]] or list(range(6))
for cfg in want:
    i, j = ql[cfg * per_cfg + per_cfg // 2 + 5], ql[cfg * per_cfg + per_cfg // 2 + 6]
    t0 = rows[i][1]
    print('--- cfg', cfg, 'call span us', (rows[j][1] - t0) / 1000, 'kernel time', sum(e - s for _, s, e in rows[i:j]) / 1000)
    for n, s, e in rows[i:j]:
        print(f"  {short(n):50s} start={(s - t0) / 1000:8.1f} dur={(e - s) / 1000:7.1f}")
