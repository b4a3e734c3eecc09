#!/bin/bash
# usage: gpu_pass.sh TAG [pytest-args...]   -- GPU test suite + default bench + rocprofv3 kernel stats into gpurun_out/TAG
R=$GRAFT_REPO_ROOT; TAG=${1:-pass}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( time timeout 2400 python -m pytest tests -m gpu -x -q "$@" ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err < /dev/null
tail -c 1500 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $O/bench_under_rocprof.json 2> $O/stats.err < /dev/null
for f in $(find $O/stats -name "*kernel_stats.csv"); do head -5 $f; cp $f $O/kernel_stats.csv; done
