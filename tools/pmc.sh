#!/bin/bash
# usage: pmc.sh TAG "COUNTER1 COUNTER2 ..." [bench args]  -- one rocprofv3 --pmc pass, averaged per k_step dispatch (steady state)
R=$GRAFT_REPO_ROOT; TAG=$1; CNT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$TAG
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python $R/bench.py --exact --steps 100 --warmup 1500 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$TAG/log.txt 2>&1 < /dev/null
python - <<PY
import csv,glob,collections
fs=glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(list)
for f in fs:
    for row in csv.DictReader(open(f)):
        if 'k_step' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in sorted(acc.items()):
    v=v[len(v)//2:]
    print("$TAG", k, "per launch", round(sum(v)/len(v),1), "launches", len(v))
PY
