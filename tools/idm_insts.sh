#!/bin/bash
# instruction counts of the IDM policy's parts: exit-profile build under rocprofv3 --pmc, waves return after policy + dynamics +
# contacts (mark 4) with parts of the policy switched off (bit 0 broad phase, 1 front / back search, 2 lane-change logic, 3 PID + IDM law)
R=$GRAFT_REPO_ROOT; MODE=${1:-dense}; cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/libpgd_exit.so
for bits in 0 1 2 4 8 15; do
  pt=$(( 4 + bits * 256 ))
  O=$R/gpurun_out/xi_$pt; rm -rf $O; mkdir -p $O
  (cd $R && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O -- python tools/exit_insts.py $pt $MODE > $O/log.txt 2>&1 < /dev/null)
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_step' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
out={k:sum(v[-80:])/80/4096 for k,v in acc.items()}
print("skip bits %2d" % $bits, {k:round(v,1) for k,v in sorted(out.items())}, "total", round(sum(out.values()),1))
PY
  rm -rf $O
done
