#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p6}
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 300 python tools/mlp_bench.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/mlp_bench.txt
timeout 300 python tools/mlp_bench.py 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/mlp_bench.txt
bash tools/ab3.sh -r 2 nopf pf 2>&1 | tee $O/ab3_pf.txt
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "idm or traffic or default_configuration or parity_campaign or free_running" ) > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
