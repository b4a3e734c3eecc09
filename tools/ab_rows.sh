#!/bin/bash
# A/B of the one-wave k_step (PGD_TWO_WAVE=0) and the two-wave k_step2 (PGD_TWO_WAVE=1) on the workload rows.
# usage: bash tools/ab_rows.sh [rows...]   rows like "uniform:trigger"
mkdir -p gpurun_out
ROWS=${@:-"uniform:trigger straight:trigger expert:trigger expert:respawn uniform:respawn"}
for row in $ROWS; do
  a=${row%%:*}; m=${row##*:}
  for tw in 0 1; do
    PGD_TWO_WAVE=$tw python bench.py --no-cpu-baseline --actions $a --traffic-mode $m $AB_ARGS 2>gpurun_out/ab.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('%-9s %-8s two_wave=$tw  %7.1f M env-steps/s  k_step %6.2f us  driving %5.2f  envs with traffic %4.2f' % ('$a','$m',d['value']/1e6,r['k_step_ms']*1e3,c.get('driving_traffic_mean',-1),c.get('envs_with_traffic_frac',-1)))" || tail -3 gpurun_out/ab.err
  done
done
