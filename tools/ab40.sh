#!/bin/bash
# A/B of prebuilt libraries (tools/mk.sh) on the multi-agent rows: 40 slots x 72 beams (2048-step window) and 8 agents x 72 beams
cd $GRAFT_REPO_ROOT
REPS=1; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
VARS="$@"; for rep in $(seq $REPS); do for v in $VARS; do for row in "40 1000 2048" "8 1000 1536"; do set -- $row
  L=$PWD/scratch/lib_$v.so; [ "$v" = head ] && L=$PWD/pgdrive_amd/libpgdrive_hip.so
  PGD_LIB=$L timeout 300 python bench.py --no-rows --no-cpu-baseline --exact --workload c5 --envs 4096 --agents $1 --lasers 72 --warmup $2 --steps $3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s agents %-3s' % ('$v','$1'), round(d['value']/1e6,2), 'M/s  k_step', round(r['k_step_ms']*1000,2), ' k_observe', round((r.get('k_observe_ms') or 0)*1000,2))"
done; done; done
