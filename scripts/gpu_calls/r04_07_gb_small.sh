cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r04_07_gb_small.json
bash scripts/gpu_prof_cmd.sh r04_gb_small scripts/bench_gb_small.py > /dev/null
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_r04_gb_small_kernels.csv')):
    print(r['kernel'][:60].ljust(60), r['calls'], r['avg_us'], r['vgpr'], r['lds_bytes'])
PY
