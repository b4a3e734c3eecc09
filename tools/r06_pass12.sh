#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p12}
O=$R/gpurun_out/$TAG; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "lane_keep or scripted or traffic_band or default_configuration or instantiations" ) > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
bash tools/ab3.sh -r 2 pinned lk 2>&1 | tee $O/ab3_lk.txt
