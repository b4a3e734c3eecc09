#!/bin/bash
# k_topdown time of several scratch/lib_<name>.so builds (rocprofv3 kernel stats of the top-down bench)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/td_$v; rm -rf $O; mkdir -p $O
  PGD_LIB=$R/scratch/lib_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --topdown --exact --steps 200 --warmup 100 --no-cpu-baseline > $O/bench.json 2> $O/err.txt < /dev/null
  for f in $(find $O -name "*kernel_stats.csv"); do python3 -c "import csv,sys; [print(\"$v\", r[\"Calls\"], round(float(r[\"AverageNs\"])/1000,1), \"us\") for r in csv.DictReader(open(\"$f\")) if r[\"Name\"].startswith(\"k_topdown(\")]"; done
done
