cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
ARROWHIP_LIB=$GRAFT_REPO_ROOT/scripts/micro/libarrowhip_before.so timeout 600 python scripts/bench_slices.py 2> /dev/null | sed 's/^/before /'
timeout 600 python scripts/bench_slices.py 2> /dev/null | sed 's/^/after  /'
done | tee gpurun_out/r3c32_slices_before_after.txt
