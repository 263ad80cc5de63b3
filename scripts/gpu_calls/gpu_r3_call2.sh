# round 3, GPU call 2: ingest parity + legs + a copy/kernel trace; the changed kernels again (vec take v2, scatter registers)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ingest.py tests/test_gpu_parity.py -m gpu -x -q -k "ingest or filter or take or hash_sum" > gpurun_out/r3c2_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c2_pytest.log
tail -12 gpurun_out/r3c2_pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r3c2_bench.json 2> gpurun_out/r3c2_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r3c2_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c2_bench.json'))
print(d['value'], d['roofline']['frac'])
k=d.get('kernels',{})
for n,v in k.items():
    if isinstance(v,dict) and any(s in n for s in ('filter_int64_nulls10_sel0.50','take','hash_sum','dictionary')):
        print(n, v.get('ms'), v.get('GB/s'), v.get('fill_only_ms',''))
print(json.dumps(d.get('host_ingest'),indent=0))
PY
cd /tmp; rm -rf /tmp/prof_ing
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof_ing -o run -- python $R/scripts/bench_ingest.py 27 > $R/gpurun_out/r3c2_ingest_prof.out 2> $R/gpurun_out/r3c2_ingest_prof.err; echo "prof rc=$?"
ls -la /tmp/prof_ing | head
python - <<'PY'
import sqlite3, glob
for f in glob.glob('/tmp/prof_ing/*.db'):
    db=sqlite3.connect(f)
    names=[r for r in db.execute("select type,name from sqlite_master where name like '%copy%' or name like '%copies%' or name like 'kernels' or name like '%memory%'")]
    print(f, names)
    for t,n in names:
        try:
            cols=[c[1] for c in db.execute(f"pragma table_info('{n}')")]
            print(n, cols)
            for row in db.execute(f"select * from '{n}' limit 3"): print('   ', row)
        except Exception as e: print(n, e)
PY
cp /tmp/prof_ing/*.db $R/gpurun_out/r3c2_ingest.db 2>/dev/null; ls -la $R/gpurun_out/r3c2_ingest.db
