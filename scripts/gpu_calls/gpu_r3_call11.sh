cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/bench_encode_part.py 19 20 21 22 > gpurun_out/r3c11_encode_part.json 2> gpurun_out/r3c11_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c11_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c11_encode_part.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'auto' in a or 'global_table_encode' in a})
PY
