#!/bin/bash
# the profile pass behind profiles/rNN_*: default bench.py line, rocprofv3 kernel stats of the same command, FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (+ the calibration kernels of profiles/r01_calib.hip), summarised by tools/pmc_traffic.py
# usage: profile_pass.sh TAG      (outputs under gpurun_out/TAG)
R=$GRAFT_REPO_ROOT; TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/profiles/r01_calib.hip -o /tmp/pgd_calib 2>/dev/null
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rows > $O/bench_under_rocprof.json 2> $O/stats.err < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py --exact --steps 200 --warmup 1500 --no-cpu-baseline > /dev/null 2> $O/fetch.err < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py --exact --steps 200 --warmup 1500 --no-cpu-baseline > /dev/null 2> $O/write.err < /dev/null
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- /tmp/pgd_calib > $O/calib.txt 2> $O/cal.err < /dev/null
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- /tmp/pgd_calib > /dev/null 2>> $O/cal.err < /dev/null
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -4 $f; done
python $R/tools/pmc_traffic.py $O > $O/pmc_traffic.json; cat $O/pmc_traffic.json | head -30
tail -c 900 $O/bench.json
