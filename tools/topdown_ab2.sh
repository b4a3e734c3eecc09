#!/bin/bash
# top-down A/B: one kernel (default) vs prologue + banded image kernels (PGD_TD_SPLIT=1); rocprofv3 kernel stats of the top-down bench
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in one split; do
  O=$R/gpurun_out/td_$v; rm -rf $O; mkdir -p $O
  if [ $v = split ]; then export PGD_TD_SPLIT=1; else unset PGD_TD_SPLIT; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --topdown --exact --steps 300 --warmup 200 --no-cpu-baseline > $O/bench.json 2> $O/err.txt < /dev/null
  for f in $(find $O -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/td_${v}_kernel_stats.csv; python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'topdown' in r['Name'] or 'k_step' in r['Name']: print('$v', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1000,1), 'us')"; done
  tail -1 $O/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), 'M env-steps/s', round(d['ms_per_step']*1e3,1), 'us/step')"
  find $O -name "*kernel_trace.csv" -delete
done
