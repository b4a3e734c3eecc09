#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p2}
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/variant_ab.py 4096 3000 > $O/variant_ab.txt 2>&1; cat $O/variant_ab.txt | grep -v amdgpu.ids
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x -k "instantiations or reused_address or gym_make or throughput_mode or zero_row or default_configuration or world_8 or eight" ) > $O/pytest_sel.log 2>&1; tail -5 $O/pytest_sel.log
