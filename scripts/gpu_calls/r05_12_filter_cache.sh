# round 5, call 12: does the count's prefix cache help or hurt the two-phase Filter?  events + wall, then a kernel trace of s = 0.5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python scripts/bench_filter_cache.py > gpurun_out/r05_12_filter_cache.json 2> gpurun_out/r05_12_filter_cache.err; echo "rc=$?"; tail -2 gpurun_out/r05_12_filter_cache.err
cat gpurun_out/r05_12_filter_cache.json
cd /tmp; rm -rf /tmp/prof_f
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_f -o run -- python $R/scripts/bench_filter_cache.py 0.5 > /tmp/prof_f.out 2> /tmp/prof_f.err
python $R/scripts/gb_timeline.py /tmp/prof_f/run_results.db --help > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/prof_f/*results.db')[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the last 400 dispatches: print name (short), start relative, duration, gap to previous end
rows = rows[-260:-180]
t0 = rows[0][1]; prev = None
for n, s, e in rows:
    gap = (s - prev) / 1e3 if prev else 0
    print(f"{n.split('(')[0][-48:]:48s} start={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f}")
    prev = e
PY
