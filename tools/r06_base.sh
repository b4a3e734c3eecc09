#!/bin/bash
# round-6 first pass: the default bench line of the round-5 kernels on today's box + the per-wave life histograms of the loaded rows
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06base}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null ) 2> $O/bench.time; tail -c 300 $O/bench.json; echo; cat $O/bench.time
TRAFFIC=respawn timeout 600 python tools/wave_life.py uniform 4096 24 > $O/wave_life_respawn.txt 2>&1; head -8 $O/wave_life_respawn.txt
timeout 600 python tools/wave_life.py expert 4096 24 > $O/wave_life_expert.txt 2>&1; head -8 $O/wave_life_expert.txt
timeout 600 python tools/wave_life.py uniform 4096 24 > $O/wave_life_metric.txt 2>&1; head -8 $O/wave_life_metric.txt
