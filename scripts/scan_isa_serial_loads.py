#!/usr/bin/env python3
"""Heuristic scan of gfx950 assembly for serialised memory loads: per kernel, how often a vector load is followed by
`s_waitcnt vmcnt(0)` and then, within a few instructions, by the next load — one memory round trip per element where the
source meant them to overlap (a data-dependent loop or a per-row guard between two loads does that).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only arrow_go_amd/csrc/X.hip -o /tmp/X.s
    python scripts/scan_isa_serial_loads.py /tmp/X.s [...]"""
import re, sys, collections
for path in sys.argv[1:]:
    kern=None; res=collections.OrderedDict()
    state=0; dist=0
    for line in open(path, errors='ignore'):
        l=line.strip()
        m=re.match(r'^(_Z\w+):\s', line)
        if m: kern=m.group(1); res[kern]=[0,0,0]; state=0; continue
        if kern is None or not l or l.startswith(';') or l.startswith('.'): continue
        op=l.split()[0]
        dist+=1
        if op.startswith('global_load') or op.startswith('buffer_load'):
            res[kern][1]+=1
            if state==2 and dist<16: res[kern][0]+=1
            state=1
        elif op=='s_waitcnt' and 'vmcnt(0)' in l:
            res[kern][2]+=1
            if state==1: state=2; dist=0
    for k,(ser,loads,waits) in res.items():
        if ser>=4: print(f"{ser:4d} serialised  {loads:4d} loads {waits:4d} waits0  {path.split('/')[-1]}  {k[:110]}")
