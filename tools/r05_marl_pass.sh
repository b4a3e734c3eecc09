#!/bin/bash
# the multi-agent rows again after a change of their kernels: GPU suite, default bench line, row passes of the three c5 rows
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r05m}
O=$R/gpurun_out/$TAG; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
bash tools/row_pass.sh $TAG/c5_8x240 c5 4096 uniform trigger 8 240
bash tools/row_pass.sh $TAG/c5_8x72 c5 4096 uniform trigger 8 72
bash tools/row_pass.sh $TAG/c5_40x72 c5 4096 uniform trigger 40 72
