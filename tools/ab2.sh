#!/bin/bash
# A/B of prebuilt libraries scratch/lib_<name>.so on the metric's row and the expert row (steady state, bench.py's own kernel time)
# usage: ab2.sh [-r REPS] name1 name2 ...
cd $GRAFT_REPO_ROOT
REPS=1; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
for rep in $(seq $REPS); do for v in "$@"; do for act in uniform expert; do
  PGD_LIB=$PWD/scratch/lib_$v.so timeout 200 python bench.py --no-rows --no-cpu-baseline --actions $act $AB_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-8s' % ('$v','$act'), round(d['value']/1e6,2), 'M/s', round(d['ms_per_step']*1000,2), 'us  k_step', round(d['roofline']['k_step_ms']*1000,2))"
done; done; done
