#!/bin/bash
# k_step time vs the number of maps the envs are spread over (cache footprint of the map tables)
cd $GRAFT_REPO_ROOT
for m in "$@"; do
  python bench.py --exact --steps 3000 --warmup 1500 --no-cpu-baseline --maps $m $MS_ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('maps', $m, round(d['value']/1e6,2), 'M/s  k_step us', round(d['roofline']['k_step_ms']*1000,2))"
done
