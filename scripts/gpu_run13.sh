cd $GRAFT_REPO_ROOT
python scripts/bench_take_only.py
ARROWHIP_LIB=$PWD/scripts/micro/libarrowhip_take8.so python scripts/bench_take_only.py
ARROWHIP_LIB=$PWD/scripts/micro/libarrowhip_take16.so python scripts/bench_take_only.py
