cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "checked or add or overflow or arith" 2>&1 | tail -3
timeout 600 python scripts/bench_types.py 2>&1 | grep -E "int|uint" | sed 's/.*add_checked/add_checked/' | head -10
