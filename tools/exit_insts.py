"""Instructions a wave executes up to each top-level point of the step (-DPGD_EXITAT build under rocprofv3 --pmc): run once per
exit mark by tools/exit_insts.sh.  usage: exit_insts.py MARK [mode]"""
import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario, build
lib = os.path.join("gpurun_out", "libpgd_exit.so")
if not os.path.exists(lib):
    subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_EXITAT',
                           '-o', lib, build.SRC])
from pgdrive_amd import engine
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_exit_at.argtypes = [C.c_void_p, C.c_int]
pt = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else 'uniform'
N = 4096
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs, [d['seed'] for d in descs], traffic_mode='respawn' if mode == 'dense' else 'trigger')
eng = engine.Engine(_abi.make_config(N, seed=1234), mb, sb)
eng.reset(np.arange(N) % 100)
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
with torch.cuda.stream(eng.stream):
    for k in range(1500): eng.step(acts[k % 64])
    eng.sync()
    L.pgd_debug_exit_at(eng.h, pt)
    for k in range(100): eng.step(acts[0])
    eng.sync()
