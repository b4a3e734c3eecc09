#!/bin/bash
# workload sweep of profiles/rNN_sweep.md (one GPU): C3 at 4096 / 32768 / 262144 envs, drive-straight, C2, C5, async groups, top-down
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err; python -c "
import json; d=json.loads(open('gpurun_out/sweep/$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('| $name | %s | %.1f M | %.4f | %.4f | %.4f | %.4f |' % (d['config']['workload'][:70], d['value']/1e6, d['ms_per_step'], r['k_step_ms'], r['k_observe_ms'], r['frac']))"; }
echo "| run | workload | env-steps/s | ms/step | k_step ms | k_observe ms | roofline frac |"; echo "|---|---|---|---|---|---|---|"
run c3_4096 --envs 4096
run c3_32768 --envs 32768 --steps 500 --warmup 1500 --exact
run c3_262144 --envs 262144 --steps 100 --warmup 300 --exact
run c3_straight --envs 4096 --actions straight
run c3_groups2_4096 --envs 4096 --groups 2
run c3_groups2_8192 --envs 8192 --groups 2
run c2_1024 --envs 1024 --traffic 0 --lasers 0
run c2_65536 --envs 65536 --traffic 0 --lasers 0 --steps 500 --warmup 500 --exact
run c5_4096x8 --workload c5 --envs 4096 --agents 8 --steps 500 --warmup 500 --exact
run c5_4096x8_240beams --workload c5 --envs 4096 --agents 8 --lasers 240 --steps 300 --warmup 300 --exact
run c5_4096x40 --workload c5 --envs 4096 --agents 40 --steps 200 --warmup 200 --exact
run c3_topdown --envs 4096 --topdown --steps 400 --warmup 300 --exact
