#!/bin/bash
# rocprofv3 kernel stats of the top-down observation bench (pgd_step + pgd_observe_topdown per step)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/topdown; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --topdown --exact --steps 300 --warmup 300 --no-cpu-baseline > $O/bench.json 2> $O/err.txt < /dev/null
for f in $(find $O -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -5 $f; done
