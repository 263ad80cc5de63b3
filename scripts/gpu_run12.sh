cd $GRAFT_REPO_ROOT
python scripts/bench_filter_only.py
ARROWHIP_LIB=$PWD/scripts/micro/libarrowhip_tile8192.so python scripts/bench_filter_only.py
ARROWHIP_LIB=$PWD/scripts/micro/libarrowhip_tile32768.so python scripts/bench_filter_only.py
