#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p11}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null ) 2> $O/bench.time; tail -c 200 $O/bench.json; echo; cat $O/bench.time
export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pol -- python tools/mlp_bench.py 4096 ) > $O/mlp_bench_under_rocprof.txt 2> $O/pol.err < /dev/null
for f in $(find $O/pol -name "*kernel_stats.csv"); do cp $f $O/policy_kernel_stats.csv; head -5 $f; done; rm -rf $O/pol
( time timeout 3000 python tests/parity_campaign.py ) > $O/campaign.log 2>&1; tail -8 $O/campaign.log; cp gpurun_out/campaign.json $O/campaign.json
