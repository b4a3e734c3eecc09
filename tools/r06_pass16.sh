#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p16}
O=$R/gpurun_out/$TAG; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -s -k "steer_lag or lateral_behaviour" ) > $O/pytest_sel.log 2>&1; grep -E "IDM traffic|steer lag parity|passed|failed|Error|assert" $O/pytest_sel.log | cut -c1-900
