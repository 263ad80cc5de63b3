cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python scripts/bench_encode_unperm.py 19 20 21 22 23 24 25 > gpurun_out/r3c21_encode.json 2> gpurun_out/r3c21_encode.err; echo "rc=$?"; tail -3 gpurun_out/r3c21_encode.err
python -c "
import json;d=json.load(open('gpurun_out/r3c21_encode.json'))
for k,v in d['results'].items(): print(k,v)"
