#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export PGD_LIB=$PWD/scratch/lib_$v.so
  python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c3      ', round(d['value']/1e6,1), round(d['roofline']['k_step_ms']*1e3,2))"
  python bench.py --no-cpu-baseline --workload c5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5      ', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"
  python bench.py --no-cpu-baseline --workload c5 --lasers 240 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5-240  ', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"
  python bench.py --no-cpu-baseline --envs 32768 --exact --steps 500 --warmup 1500 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c3-32768', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"
  python bench.py --no-cpu-baseline --topdown --exact --steps 300 --warmup 200 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v topdown ', round(d['value']/1e6,1), round(d['ms_per_step']*1e3,2))"
  python bench.py --no-cpu-baseline --traffic-mode respawn 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v dense   ', round(d['value']/1e6,1), round(d['roofline']['k_step_ms']*1e3,2))"
done
