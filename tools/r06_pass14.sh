#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p14}
O=$R/gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do for v in pinned lk2 nowrappin; do
  PGD_LIB=$PWD/scratch/lib_$v.so timeout 200 python bench.py --no-rows --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s' % '$v', round(d['value']/1e6,2), 'M/s  k_step', round(d['roofline']['k_step_ms']*1000,3))"
done; done | tee $O/ab_metric.txt
