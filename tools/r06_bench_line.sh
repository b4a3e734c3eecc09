#!/bin/bash
# the default bench line as the driver runs it (profiles/r06_bench.json): with the round's stamped counter passes committed, `stale: false`
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06bench}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err < /dev/null ) 2> $O/bench.time; tail -c 200 $O/bench.json; echo; cat $O/bench.time
