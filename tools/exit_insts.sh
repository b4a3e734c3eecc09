#!/bin/bash
# instruction counts per wave up to each exit mark of k_step (PGD_EXITAT build): one rocprofv3 --pmc run per mark
R=$GRAFT_REPO_ROOT; MODE=${1:-uniform}; cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/libpgd_exit.so
for pt in 99 13 0 1 4 5 6 7 8 20 21 22 23 14 -1; do
  O=$R/gpurun_out/xi_$pt; rm -rf $O; mkdir -p $O
  (cd $R && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O -- python tools/exit_insts.py $pt $MODE > $O/log.txt 2>&1 < /dev/null)
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if 'k_step' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
out={k:sum(v[-80:])/80/4096 for k,v in acc.items()}
print("mark %4d" % $pt, {k:round(v,1) for k,v in sorted(out.items())}, "total", round(sum(out.values()),1))
PY
  rm -rf $O
done
