#!/bin/bash
# A/B of prebuilt libraries scratch/lib_<name>.so with the same steady-state bench command (scratch/ is git-ignored; it
# still travels with gpurun).  usage: ab.sh [-r REPS] name1 name2 ...     build: tools/mk.sh name [-DFLAG ...]
cd $GRAFT_REPO_ROOT
REPS=1; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
for rep in $(seq $REPS); do for v in "$@"; do
  PGD_LIB=$PWD/scratch/lib_$v.so timeout 100 python bench.py --exact --steps 3000 --warmup 1500 --no-cpu-baseline $AB_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1000,2), 'us  k_step', round(d['roofline']['k_step_ms']*1000,2))"
done; done
