#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do cp scratch/lib_$v.so pgdrive_amd/libpgdrive_hip.so; timeout 100 python bench.py --steps 1500 --warmup 150 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), round(d['ms_per_step']*1000,2))"; done
