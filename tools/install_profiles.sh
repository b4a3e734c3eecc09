#!/bin/bash
# copy the summaries of a tools/r06_final.sh pass (gpurun_out/TAG) into profiles/ under the round's names
cd $(dirname $0)/..; F=gpurun_out/${1:-r06final}; R=${2:-r06}
rm -f profiles/${R}_*_kernel_stats.csv profiles/${R}_*_kernel_stats.source_sha profiles/${R}_*_pmc_traffic.json profiles/${R}_*_pmc_insts.json profiles/${R}_*_bench.json profiles/${R}_*_bench_under_rocprof.json
for r in head straight c2_1024 expert respawn expert_respawn c5_8x240 c5_8x72 c3_32768 c5_40x72; do for f in kernel_stats.csv kernel_stats.source_sha pmc_traffic.json pmc_insts.json bench.json bench_under_rocprof.json; do [ -f $F/$r/$f ] && cp $F/$r/$f profiles/${R}_${r}_$f; done; done
cp $F/bench.json profiles/${R}_bench.json
cp $F/topdown_kernel_stats.csv profiles/${R}_topdown_kernel_stats.csv; cp $F/source_sha.txt profiles/${R}_topdown_kernel_stats.source_sha
[ -f $F/topdown_u8_kernel_stats.csv ] && cp $F/topdown_u8_kernel_stats.csv profiles/${R}_topdown_u8_kernel_stats.csv && cp $F/source_sha.txt profiles/${R}_topdown_u8_kernel_stats.source_sha
cp $F/policy_kernel_stats.csv profiles/${R}_policy_kernel_stats.csv; cp $F/source_sha.txt profiles/${R}_policy_kernel_stats.source_sha
for w in metric expert respawn; do grep -v amdgpu.ids $F/wave_life_$w.txt > profiles/${R}_wave_life_${w}.txt; done
cp $F/mlp_bench.txt profiles/${R}_mlp_bench.txt; tail -12 $F/pytest.log > profiles/${R}_gpu_suite.log; cp $F/campaign.json profiles/${R}_parity_campaign.json
cat $F/source_sha.txt
