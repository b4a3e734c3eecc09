#!/bin/bash
# k_step time vs number of envs (= waves): separates the latency chain of one wave from the contention between waves
cd $GRAFT_REPO_ROOT
for n in "$@"; do timeout 200 python bench.py --envs $n --exact --steps 2000 --warmup 1500 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('envs', $n, 'M/s', round(d['value']/1e6,1), 'us/step', round(d['ms_per_step']*1000,2), 'k_step us', round(d['roofline']['k_step_ms']*1000,2))"; done
