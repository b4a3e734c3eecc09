#!/bin/bash
# register / spill / LDS metadata of every kernel in the engine (cross-compiles for gfx950; no GPU needed)
mkdir -p /tmp/isa && cd /tmp/isa && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -mllvm -bonus-inst-threshold=4 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero -funsafe-math-optimizations -shared -fPIC --save-temps "$@" -o /tmp/isa/lib.so /root/repo/pgdrive_amd/csrc/pgd_engine.hip 2>&1 | grep -E "error" -A3
python3 - <<'PY'
import re
s=open('/tmp/isa/pgd_engine-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', s, re.S):
    blk=m.group(0)
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,blk).group(1)
    print(name[:52].ljust(52), 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'sspill',g('sgpr_spill_count'),'vspill',g('vgpr_spill_count'),'lds',g('group_segment_fixed_size'),'scratch',g('private_segment_fixed_size'))
PY
