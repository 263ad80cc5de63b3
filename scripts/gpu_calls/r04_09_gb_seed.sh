cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r04_09_gb_small.json
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group" > gpurun_out/r04_09_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_09_pytest.log
tail -25 gpurun_out/r04_09_pytest.log | cut -c1-250
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "hash_sum or c5" > gpurun_out/r04_09_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_09_pytest_full.log
tail -3 gpurun_out/r04_09_pytest_full.log | cut -c1-250
bash scripts/gpu_prof_cmd.sh r04_gb_small scripts/bench_gb_small.py > /dev/null
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r04_gb_small/run_results.db')
rows=list(db.execute("select name, start, end from kernels order by start"))
seq=[(n, (e-s)/1000) for n,s,e in rows]
def pick(tag): return [d for n,d in seq if tag in n]
agg, fin, ql, red, prep = pick('gb_aggregate'), pick('gd_finish'), pick('quicklook'), pick('gd_reduce'), pick('gd_prep')
for i in range(0,len(agg),11):
    print('agg', [round(x,1) for x in agg[i+1:i+4]], 'reduce', [round(x,1) for x in red[i+1:i+3]], 'finish', [round(x,1) for x in fin[i+1:i+3]], 'ql', [round(x,1) for x in ql[i+1:i+3]], 'prep', [round(x,1) for x in prep[i+1:i+2]])
PY
