#!/bin/bash
# workload sweep of profiles/rNN_sweep.md (one GPU, final kernels): the C3 workload rows (idle / straight / expert / dense), C3 at
# 16384 / 32768 / 262144 envs (default kernel and throughput mode), C2, C5, async groups, open loop, top-down
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/sweep
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err; python -c "
import json; d=json.loads(open('gpurun_out/sweep/$name.json').read().strip().splitlines()[-1]); r=d['roofline'] or {}; c=d['config']
print('| $name | %s | %.1f M | %.2f | %.2f | %.2f | %.3f | %s | %s |' % (c['workload'][:86], d['value']/1e6, d['ms_per_step']*1e3, r.get('k_step_ms',0)*1e3, r.get('k_observe_ms',0)*1e3, r.get('frac',0),
  ('%.2f' % c['driving_traffic_mean']) if 'driving_traffic_mean' in c else '-', ('%.2f' % c['envs_with_traffic_frac']) if 'envs_with_traffic_frac' in c else '-'))" || tail -2 gpurun_out/sweep/$name.err; }
echo "| run | workload | env-steps/s | us/step | k_step us | k_observe us | roofline frac (algorithmic bytes) | driving IDM vehicles / env | envs with driving traffic |"; echo "|---|---|---|---|---|---|---|---|---|"
run c3_4096 --envs 4096
run c3_straight --envs 4096 --actions straight
run c3_expert --envs 4096 --actions expert
run c3_expert_respawn --envs 4096 --actions expert --traffic-mode respawn
run c3_uniform_respawn --envs 4096 --traffic-mode respawn
PGD_PACK=0 run c3_16384_default --envs 16384 --steps 1000 --warmup 1500 --exact
PGD_PACK=1 run c3_16384_throughput --envs 16384 --steps 1000 --warmup 1500 --exact
PGD_PACK=0 run c3_32768_default --envs 32768 --steps 500 --warmup 1500 --exact
run c3_32768 --envs 32768 --steps 500 --warmup 1500 --exact
PGD_PACK=0 run c3_262144_default --envs 262144 --steps 100 --warmup 300 --exact
run c3_262144 --envs 262144 --steps 100 --warmup 300 --exact
run c3_groups2_4096 --envs 4096 --groups 2
run c3_groups2_8192 --envs 8192 --groups 2
run c3_step_n16 --envs 4096 --step-n 16
run c2_1024 --envs 1024 --traffic 0 --lasers 0
run c2_65536 --envs 65536 --traffic 0 --lasers 0 --steps 500 --warmup 500 --exact
run c5_4096x8 --workload c5 --envs 4096 --agents 8 --steps 1500 --warmup 1000 --exact
run c5_4096x8_240beams --workload c5 --envs 4096 --agents 8 --lasers 240 --steps 1500 --warmup 1000 --exact
run c5_4096x40 --workload c5 --envs 4096 --agents 40 --steps 200 --warmup 200 --exact
run c3_topdown --envs 4096 --topdown --steps 400 --warmup 300 --exact
