#!/bin/bash
# C5 (multi-agent roundabout) figures for the docs: bench lines at 8 agents / 72 beams, 8 agents / 240 beams, 40 agents
cd $GRAFT_REPO_ROOT
for a in "--agents 8" "--agents 8 --lasers 240" "--agents 40" "--agents 8 --groups 2"; do
  python bench.py --workload c5 $a --exact --steps 2000 --warmup 1500 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c5 $a', round(d['value']/1e6,2), 'M env-steps/s  k_step us', round(r['k_step_ms']*1000,1), 'k_observe us', round((r.get('k_observe_ms') or 0)*1000,1))"
done
