"""Timing-only ablations of the IDM policy on the steady-state snapshot (build -DPGD_EXITAT: waves return at mark 4 = after
policy + dynamics + contacts, nothing is stored).  Bits: 1 broad phase off, 2 neighbour search off, 4 lane-change logic off,
8 PID + IDM law off.  usage: idm_ablation.py [N]"""
import sys, os, ctypes as C, numpy as np, subprocess, time
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario, build
lib = os.path.join("gpurun_out", "libpgd_exit.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_EXITAT', '-o', lib, build.SRC])
from pgdrive_amd import engine
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_exit_at.argtypes = [C.c_void_p, C.c_int]
L.pgd_debug_step_many.argtypes = [C.c_void_p] * 6 + [C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs, [d['seed'] for d in descs])
eng = engine.Engine(_abi.make_config(N, seed=1234), mb, sb)
eng.reset(np.arange(N) % 100)
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
with torch.cuda.stream(eng.stream):
    for k in range(1500): eng.step(acts[k % 64])
    eng.sync()
    for mark in (4, 255):
        for bits, name in ((0, 'all on'), (1, 'broad phase off'), (2, 'search off'), (3, 'broad + search off'), (4, 'lane change off'), (8, 'pid + law off'), (15, 'whole policy off')):
            L.pgd_debug_exit_at(eng.h, mark | (bits << 8))
            ptrs = [C.c_void_p(t.data_ptr()) for t in (acts[0], eng.obs, eng.reward, eng.done, eng.flags)]
            L.pgd_debug_step_many(eng.h, *ptrs, 50); eng.sync()
            t0 = time.perf_counter()
            L.pgd_debug_step_many(eng.h, *ptrs, 1000); eng.sync()
            print('exit at %3d  %-20s %6.2f us' % (mark, name, (time.perf_counter() - t0) / 1000 * 1e6))
            if mark == 255: break  # a full step stores: only the unablated line is meaningful there
