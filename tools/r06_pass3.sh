#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p3}
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 600 python tools/variant_ab.py 4096 3000 > $O/variant_ab.txt 2>&1; cat $O/variant_ab.txt | grep -v amdgpu.ids
( time timeout 900 python -m pytest tests -m gpu -q --timeout 900 -x -k "instantiations" ) > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
( time timeout 900 python bench.py --no-cpu-baseline --rows c3_policy,c5_40x72,c5_8x72,c3_topdown > $O/bench_rows.json 2> $O/bench_rows.err < /dev/null ) 2> $O/bench.time; tail -c 2500 $O/bench_rows.json; echo; cat $O/bench.time; tail -5 $O/bench_rows.err
timeout 600 python tools/phase_profile_c5.py 4096 40 72 > $O/phase_c5_40.txt 2>&1; grep -v amdgpu.ids $O/phase_c5_40.txt
