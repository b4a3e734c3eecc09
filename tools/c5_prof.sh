#!/bin/bash
# rocprofv3 kernel stats + bench line of the multi-agent workload (C5: 4096 envs x 8 agents, 72 beams)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/c5; rm -rf $O; mkdir -p $O
timeout 300 python $R/bench.py --workload c5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --workload c5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/err.txt < /dev/null
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -4 $f; done
tail -c 600 $O/bench.json
