#!/bin/bash
# round-5 measurement pass: GPU suite, the default bench line (with rows), profile passes of every row bench.py matches counters for
# (kernel stats, FETCH_SIZE / WRITE_SIZE, SQ_INSTS_* + wave-life shares: tools/row_pass.sh)
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r05}
O=$R/gpurun_out/$TAG; mkdir -p $O
if [ -z "$NO_SUITE" ]; then ( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log; fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; tail -c 300 $O/bench.json; echo
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
