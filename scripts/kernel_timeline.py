"""Print the last N kernel dispatches of a rocprofv3 --kernel-trace database: name, start (relative), duration, gap to the previous
kernel's end (µs).  python scripts/kernel_timeline.py <results.db> [first_from_end] [last_from_end]"""
import sqlite3, sys
db = sys.argv[1]
a = int(sys.argv[2]) if len(sys.argv) > 2 else 120
b = int(sys.argv[3]) if len(sys.argv) > 3 else 0
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[len(rows) - a:len(rows) - b if b else None]
t0 = rows[0][1]; prev = None
for n, s, e in rows:
    gap = (s - prev) / 1e3 if prev else 0
    short = n.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{short[-60:]:60s} start={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f}")
    prev = e
