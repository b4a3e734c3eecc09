#!/bin/bash
# A/B of prebuilt libraries on three rows: the metric (uniform / trigger), expert / trigger, uniform / respawn (dense)
cd $GRAFT_REPO_ROOT
REPS=1; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
VARS="$@"; for rep in $(seq $REPS); do for v in $VARS; do for row in "uniform trigger" "expert trigger" "uniform respawn"; do set -- $row
  PGD_LIB=$PWD/scratch/lib_$v.so timeout 200 python bench.py --no-rows --no-cpu-baseline --actions $1 --traffic-mode $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-8s %-8s' % ('$v','$1','$2'), round(d['value']/1e6,2), 'M/s  k_step', round(d['roofline']['k_step_ms']*1000,2))"
done; done; done
