#!/bin/bash
# throughput mode A/B at large N: default kernel (PGD_PACK=0) vs packed (PGD_PACK=1)
mkdir -p gpurun_out
for n in ${@:-4096 16384 32768 262144}; do for pk in 0 1; do
  st=$(( n >= 100000 ? 300 : 1000 ))
  PGD_PACK=$pk timeout 300 python bench.py --no-cpu-baseline --envs $n --exact --warmup 1500 --steps $st $AB_ARGS 2>gpurun_out/pack.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('envs %7d pack=$pk  %7.1f M env-steps/s  step %8.2f us  k_step %8.2f us  frac %.3f' % ($n, d['value']/1e6, d['ms_per_step']*1e3, r['k_step_ms']*1e3, r['frac']))" || tail -3 gpurun_out/pack.err
done; done
