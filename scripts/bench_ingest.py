#!/usr/bin/env python3
"""The host-ingest legs of bench.py on their own (for a rocprofv3 --kernel-trace --memory-copy-trace run):
    python scripts/bench_ingest.py [log2 rows, default 27]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
rows = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
with ah.Context(0) as ctx:
    print(json.dumps(bench.host_ingest_legs(ctx, rows)))
