#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; TAG=${1:-r06p19}
O=$R/gpurun_out/$TAG; mkdir -p $O
cat > /tmp/jf.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario
from pgdrive_amd.engine import Engine
N=4096
descs = bank.get_descriptions(range(1000, 1100)); mb = mapdata.MapBank(descs)
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
for mode in ("trigger","respawn"):
    sb = scenario.ScenarioBank(descs, [d["seed"] for d in descs], num_agents=1, num_traffic=16, traffic_mode=mode)
    for rep in range(2):
        for force in (False, True):
            if force: os.environ["PGD_JIT_FORCE"]="1"
            else: os.environ.pop("PGD_JIT_FORCE", None)
            eng = Engine(_abi.make_config(N, auto_reset=1, seed=1234), mb, sb)
            if force: eng.specialise(wait=True)
            eng.reset(np.arange(N) % 100)
            with torch.cuda.stream(eng.stream):
                for k in range(3000): eng.step(acts[k % 64])
                eng.sync(); eng.profile_begin(6000 // 64 + 2, stride=64)
                for k in range(6000): eng.step(acts[k % 64])
                eng.sync(); p = eng.profile_end()
            print(mode, "forced run-time kernel" if force else "AOT instantiation   ", "k_step %.3f us" % (p["k_step_ms"]*1e3), eng.describe_step()[:80], flush=True)
            eng.close()
PY
cp /tmp/jf.py $O/jf.py; timeout 900 python $O/jf.py 2>&1 | grep -v amdgpu.ids | tee $O/jit_force.txt
