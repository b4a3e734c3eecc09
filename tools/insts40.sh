#!/bin/bash
# instruction counts per launch of the multi-agent kernels (40 slots x 72 beams), by class: one rocprofv3 --pmc pass over the bench row
R=$GRAFT_REPO_ROOT; AG=${1:-40}; cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/xi40; rm -rf $O; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH"; do
(cd $R && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -- python bench.py --no-rows --no-cpu-baseline --exact --workload c5 --envs 4096 --agents $AG --lasers 72 --warmup 1600 --steps 100 > $O/log.txt 2>&1 < /dev/null)
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name']
        if 'k_step' in k or 'k_observe_env' in k: acc[k.split('(')[0][:40]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,d in acc.items():
    print(k, {c:round(sum(v[-100:])/100/4096,1) for c,v in sorted(d.items())}, "per env (launch / 4096)")
PY
rm -rf $O/*
done
