cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash scripts/gpu_prof_cmd.sh r04_gb_small scripts/bench_gb_small.py > /dev/null
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r04_gb_small/run_results.db')
rows=list(db.execute("select name, start, end from kernels order by start"))
seq=[(n, (e-s)/1000, s/1000.0, e/1000.0) for n,s,e in rows]
def pick(tag): return [d for n,d,_,_ in seq if tag in n]
agg, fin, ql, red, prep = pick('gb_aggregate'), pick('gd_finish'), pick('quicklook'), pick('gd_reduce'), pick('gd_prep')
for i in range(0,len(agg),11):
    print('agg', [round(x,1) for x in agg[i+1:i+4]], 'reduce', [round(x,1) for x in red[i+1:i+3]], 'finish', [round(x,1) for x in fin[i+1:i+3]], 'ql', [round(x,1) for x in ql[i+1:i+3]], 'prep', [round(x,1) for x in prep[i+1:i+2]])
# gaps: time between quick look end and prep start (host round trip), per call
names=[n for n,_,_,_ in seq]
gaps=[]
for i in range(len(seq)-1):
    if 'quicklook' in seq[i][0] and 'gd_prep' in seq[i+1][0]: gaps.append(seq[i+1][2]-seq[i][3])
print('ql->prep gap us: median', sorted(gaps)[len(gaps)//2] if gaps else None)
PY
