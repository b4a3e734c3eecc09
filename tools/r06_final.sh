#!/bin/bash
# round-6 final pass: GPU suite, the default bench line, tools/row_pass.sh for every row of the line (kernel stats, FETCH_SIZE /
# WRITE_SIZE, SQ_INSTS_* -- every summary stamped with the binary's source sha), the top-down kernel's stats, the env-count sweep
# SURVEY 8(d) asks for (default and throughput mode), the per-wave life histograms of the loaded rows, the policy kernel alone
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06final}
O=$R/gpurun_out/$TAG; mkdir -p $O
if [ -z "$NO_SUITE" ]; then ( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log; fi
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null ) 2> $O/bench.time; tail -c 300 $O/bench.json; echo; cat $O/bench.time
bash tools/row_pass.sh $TAG/head c3 4096 uniform trigger 1 240
bash tools/row_pass.sh $TAG/straight c3 4096 straight trigger 1 240
bash tools/row_pass.sh $TAG/c2_1024 c3 1024 uniform trigger 1 0 --traffic 0
bash tools/row_pass.sh $TAG/expert c3 4096 expert trigger 1 240
bash tools/row_pass.sh $TAG/respawn c3 4096 uniform respawn 1 240
bash tools/row_pass.sh $TAG/expert_respawn c3 4096 expert respawn 1 240
bash tools/row_pass.sh $TAG/c5_8x240 c5 4096 uniform trigger 8 240
bash tools/row_pass.sh $TAG/c5_8x72 c5 4096 uniform trigger 8 72
bash tools/row_pass.sh $TAG/c3_32768 c3 32768 uniform trigger 1 240
bash tools/row_pass.sh $TAG/c5_40x72 c5 4096 uniform trigger 40 72
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/td -- python $R/bench.py --no-rows --no-cpu-baseline --topdown --exact --warmup 300 --steps 400 --windows 1 > $O/topdown_bench_under_rocprof.json 2> /dev/null < /dev/null
for f in $(find $O/td -name "*kernel_stats.csv"); do cp $f $O/topdown_kernel_stats.csv; head -4 $f; done; rm -rf $O/td
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/td8 -- python $R/bench.py --no-rows --no-cpu-baseline --topdown --topdown-u8 --exact --warmup 300 --steps 400 --windows 1 > $O/topdown_u8_bench_under_rocprof.json 2> /dev/null < /dev/null
for f in $(find $O/td8 -name "*kernel_stats.csv"); do cp $f $O/topdown_u8_kernel_stats.csv; head -3 $f; done; rm -rf $O/td8
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pol -- python tools/mlp_bench.py 4096 ) > $O/mlp_bench_under_rocprof.txt 2> /dev/null < /dev/null
for f in $(find $O/pol -name "*kernel_stats.csv"); do cp $f $O/policy_kernel_stats.csv; head -4 $f; done; rm -rf $O/pol
cd $R
( python -c "from pgdrive_amd import build; print(build.source_sha())" ) > $O/source_sha.txt
bash tools/pack_sweep.sh 4096 16384 32768 262144 2>&1 | tee $O/sweep.txt
TRAFFIC=respawn timeout 600 python tools/wave_life.py uniform 4096 24 > $O/wave_life_respawn.txt 2>&1; head -6 $O/wave_life_respawn.txt
timeout 600 python tools/wave_life.py expert 4096 24 > $O/wave_life_expert.txt 2>&1; head -6 $O/wave_life_expert.txt
timeout 600 python tools/wave_life.py uniform 4096 24 > $O/wave_life_metric.txt 2>&1; head -6 $O/wave_life_metric.txt
timeout 300 python tools/mlp_bench.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/mlp_bench.txt
( time timeout 3000 python tests/parity_campaign.py ) > $O/campaign.log 2>&1; tail -6 $O/campaign.log | cut -c1-200; cp gpurun_out/campaign.json $O/campaign.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
