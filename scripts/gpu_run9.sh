cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "hash or unique or dictionary or smoke" 2>&1 | tail -3
timeout 600 python scripts/bench_extra.py 2>gpurun_out/bench_extra.err | python -c "
import json,sys; d=json.load(sys.stdin); [print(k, v) for k,v in d.items() if 'hash' in k or 'dict' in k]"
