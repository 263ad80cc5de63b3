cd $GRAFT_REPO_ROOT/oracle && ./_ref/bench_ref _ref > ../gpurun_out/bench_ref_c1.json 2>&1; cat ../gpurun_out/bench_ref_c1.json
