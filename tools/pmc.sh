#!/bin/bash
# usage: pmc.sh tag "COUNTER1 COUNTER2 ..."
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$1
timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1 -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pmc_$1/log.txt 2>&1 < /dev/null
python - <<PY
import csv,glob,collections
fs=glob.glob("$R/gpurun_out/pmc_$1/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(lambda:[0.0,0])
for f in fs:
    for row in csv.DictReader(open(f)):
        if 'k_step' in row['Kernel_Name']:
            a=acc[row['Counter_Name']]; a[0]+=float(row['Counter_Value']); a[1]+=1
for k,(v,n) in sorted(acc.items()): print("$1", k, "per launch", round(v/n,1), "launches", n)
PY
