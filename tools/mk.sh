#!/bin/bash
# build an experimental variant of the engine: tools/mk.sh NAME [-DFLAG ...]  -> scratch/lib_NAME.so
N=$1; shift; mkdir -p /root/repo/scratch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -mllvm -bonus-inst-threshold=4 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -funsafe-math-optimizations -shared -fPIC "$@" -o /root/repo/scratch/lib_$N.so /root/repo/pgdrive_amd/csrc/pgd_engine.hip 2>&1 | grep -E "error" -A3
