cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof_cmd.sh hash scripts/bench_hash.py 10 16 20 24 > /dev/null
bash scripts/gpu_prof_cmd.sh sort scripts/bench_sort_one.py > /dev/null
bash scripts/gpu_prof_cmd.sh varlen scripts/bench_varlen.py > /dev/null
bash scripts/gpu_prof_cmd.sh ext scripts/bench_ext.py > /dev/null
bash scripts/gpu_prof_cmd.sh isin scripts/bench_isin.py > /dev/null
bash scripts/gpu_prof_cmd.sh scan scripts/bench_scan.py > /dev/null
bash scripts/gpu_prof_cmd.sh cast scripts/bench_cast.py > /dev/null
bash scripts/gpu_prof_cmd.sh hashbin scripts/bench_hash_binary.py > /dev/null
ls -la gpurun_out/*_kernels.csv
