cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_encode_resolve.py > gpurun_out/r3c30_resolve.json 2> gpurun_out/r3c30_resolve.err; echo "rc=$?"; tail -3 gpurun_out/r3c30_resolve.err
python -c "
import json;d=json.load(open('gpurun_out/r3c30_resolve.json'))
for k,v in d.items(): print(k,{a:b for a,b in v.items() if 'round1' in a})"
