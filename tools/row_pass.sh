#!/bin/bash
# Profile pass of one workload row of bench.py: the row's bench line, rocprofv3 --kernel-trace --stats of the same command,
# FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes (+ the calibration kernels), summarised by tools/pmc_traffic.py into the
# traffic record bench.py matches by workload (profiles/rNN_<row>_pmc_traffic.json).
# usage: row_pass.sh TAG WORKLOAD(c3|c5) ENVS ACTIONS MODE AGENTS LASERS [more bench args]     outputs under gpurun_out/TAG
R=$GRAFT_REPO_ROOT; TAG=$1; WL=$2; N=$3; ACT=$4; MODE=$5; AG=$6; NL=$7; shift 7
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
TR=16; [ "$WL" = "c5" ] && TR=0; prev=""; for a in "$@"; do case "$prev" in --traffic) TR=$a;; esac; prev=$a; done
ARGS="--no-rows --no-cpu-baseline --workload $WL --envs $N --actions $ACT --traffic-mode $MODE --agents $AG --lasers $NL $@"
[ -x /tmp/pgd_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/profiles/r01_calib.hip -o /tmp/pgd_calib 2>/dev/null
# (1500 + 3 x 4096 steps: the default N = 1 run times 10 k + 3 x 100 k, too long a trace for the kernel statistics)
timeout 600 python $R/bench.py $ARGS --exact --warmup 1500 --steps 4096 > $O/bench.json 2> $O/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS --exact --warmup 1500 --steps 4096 > $O/bench_under_rocprof.json 2> $O/stats.err < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py $ARGS --exact --steps 200 --warmup 1500 --windows 1 > /dev/null 2> $O/fetch.err < /dev/null
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py $ARGS --exact --steps 200 --warmup 1500 --windows 1 > /dev/null 2> $O/write.err < /dev/null
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- /tmp/pgd_calib > $O/calib.txt 2> $O/cal.err < /dev/null
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- /tmp/pgd_calib > /dev/null 2>> $O/cal.err < /dev/null
# instruction counters by class + the wave-life shares (three more passes; bench.py's roofline.issue reads the summary)
if [ -z "$NO_INSTS" ]; then
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/insts_a -- python $R/bench.py $ARGS --exact --steps 200 --warmup 1500 --windows 1 > /dev/null 2> $O/insts.err < /dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/insts_b -- python $R/bench.py $ARGS --exact --steps 200 --warmup 1500 --windows 1 > /dev/null 2>> $O/insts.err < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/insts_c -- python $R/bench.py $ARGS --exact --steps 200 --warmup 1500 --windows 1 > /dev/null 2>> $O/insts.err < /dev/null
python $R/tools/pmc_insts.py $O $N $ACT $MODE $WL $AG $NL $TR > $O/pmc_insts.json; grep -E "insts_per_wave|waves_per_launch|share" $O/pmc_insts.json
rm -rf $O/insts_a $O/insts_b $O/insts_c
fi
for f in $(find $O/stats -name "*kernel_stats.csv"); do cp $f $O/kernel_stats.csv; head -3 $f; done
# the binary the passes ran on (sidecar of the kernel statistics; the pmc summaries carry the same stamp as `source_sha`)
( cd $R && python -c "from pgdrive_amd import build; print(build.source_sha())" ) > $O/kernel_stats.source_sha
python $R/tools/pmc_traffic.py $O $N $ACT $MODE $WL $AG $NL $TR > $O/pmc_traffic.json; grep -E "bytes_per_env_step|FETCH_SIZE_KB|WRITE_SIZE_KB|fetch_correction" $O/pmc_traffic.json
tail -c 400 $O/bench.json; echo
# keep only the summaries (the raw traces are large)
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
rm -rf $O/fetch $O/write $O/cal_fetch $O/cal_write $O/stats
