#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p18}
O=$R/gpurun_out/$TAG; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x -k "run_time_kernel or instantiations or default_configuration" ) > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
timeout 900 python tools/variant_ab.py 4096 3000 2>&1 | grep -v amdgpu.ids | tee $O/variant_ab.txt
