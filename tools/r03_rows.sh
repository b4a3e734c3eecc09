#!/bin/bash
# The workload rows of profiles/r03_sweep.md: C3 idle (the metric), drive-straight, expert (trigger traffic), expert + respawn
# traffic (dense), each with bench.py's own HIP-event kernel time.  usage: bash tools/r03_rows.sh TAG [extra bench args]
TAG=${1:-rows}; shift
mkdir -p gpurun_out
for row in "uniform trigger" "straight trigger" "expert trigger" "expert respawn" "uniform respawn"; do
  set -- $row
  python bench.py --no-cpu-baseline --actions $1 --traffic-mode $2 > gpurun_out/${TAG}_$1_$2.json 2> gpurun_out/${TAG}_$1_$2.err || tail -5 gpurun_out/${TAG}_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${TAG}_$1_$2.json"))
    c=d["config"]; r=d["roofline"]
    print("%-9s %-8s %7.1f M env-steps/s  k_step %6.2f us  driving traffic %5.2f  envs with traffic %4.2f  ego %5.1f km/h  ep step %6.0f  frac %.3f frac_active %.3f" % (
        "$1","$2",d["value"]/1e6,r["k_step_ms"]*1e3,c["driving_traffic_mean"],c["envs_with_traffic_frac"],c["ego_speed_kmh_mean"],c["episode_step_mean"],r["frac"],r.get("frac_active",0)))
except Exception as e:
    print("$1 $2 failed", e)
PY
done
