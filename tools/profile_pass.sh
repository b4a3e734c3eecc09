#!/bin/bash
# final profile passes for profiles/: kernel stats + FETCH_SIZE + WRITE_SIZE (separate passes) + calibration
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/final; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/profiles/r01_calib.hip -o /tmp/pgd_calib 2>/dev/null
timeout 300 python $R/bench.py > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/fetch.err < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2> $O/write.err < /dev/null
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- /tmp/pgd_calib > $O/calib.txt 2> $O/cal.err < /dev/null
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- /tmp/pgd_calib > /dev/null 2>> $O/cal.err < /dev/null
python - <<PY
import csv,glob,collections,json
def pmc(d):
    acc=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for row in csv.DictReader(open(f)):
            a=acc[(row['Kernel_Name'].split('(')[0][:40],row['Counter_Name'])]; a[0]+=float(row['Counter_Value']); a[1]+=1
    return {k:(v/n,n) for k,(v,n) in acc.items()}
for d in ('fetch','write','cal_fetch','cal_write'):
    for k,v in sorted(pmc(d).items()): print(d,k,'avg per launch',round(v[0],2),'launches',v[1])
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
print(open("$O/bench.json").read())
PY
