#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err; python -c "
import json; d=json.loads(open('gpurun_out/sweep/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M/s  ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'k_step', round(d['roofline']['k_step_ms'],4), 'k_obs', round(d['roofline']['k_observe_ms'],4))"; }
run c3_4096 --envs 4096
run c3_32768 --envs 32768 --steps 500 --warmup 50
run c3_262144 --envs 262144 --steps 100 --warmup 10
run c3_straight --envs 4096 --actions straight
run c2_1024 --envs 1024 --traffic 0 --lasers 0
run c2_65536 --envs 65536 --traffic 0 --lasers 0 --steps 500 --warmup 50
run c5_4096x8 --workload c5 --envs 4096 --agents 8 --steps 500 --warmup 100
run c5_4096x40 --workload c5 --envs 4096 --agents 40 --steps 200 --warmup 50
