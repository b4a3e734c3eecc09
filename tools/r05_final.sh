#!/bin/bash
# round-5 final pass: GPU suite, the default bench line, the multi-agent rows' profile passes again (their kernels changed after the
# mid-round pass of tools/r05_pass.sh; the single-agent kernels did not), the top-down kernel's stats
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r05final}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null ) 2> $O/bench.time; tail -c 300 $O/bench.json; echo; cat $O/bench.time
bash tools/row_pass.sh $TAG/c5_40x72 c5 4096 uniform trigger 40 72
bash tools/row_pass.sh $TAG/c5_8x72 c5 4096 uniform trigger 8 72
bash tools/row_pass.sh $TAG/c5_8x240 c5 4096 uniform trigger 8 240
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/td -- python $R/bench.py --no-rows --no-cpu-baseline --topdown --exact --warmup 300 --steps 400 --windows 1 > $O/topdown_bench_under_rocprof.json 2> /dev/null < /dev/null
for f in $(find $O/td -name "*kernel_stats.csv"); do cp $f $O/topdown_kernel_stats.csv; head -4 $f; done; rm -rf $O/td
