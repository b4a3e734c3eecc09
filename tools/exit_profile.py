"""Launch time of k_step up to each top-level point of the step (build -DPGD_EXITAT: every wave returns at the chosen mark,
nothing is stored, so the state stays the steady-state snapshot reached by the warm-up).  usage: [TRAFFIC=respawn] exit_profile.py [N] [uniform|straight|expert]"""
import sys, os, ctypes as C, numpy as np, subprocess, time
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, bank, mapdata, scenario, build
lib = os.path.join("gpurun_out", "libpgd_exit.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_EXITAT',
                       '-o', lib, build.SRC] + os.environ.get('PGD_EXTRA', '').split())
from pgdrive_amd import engine
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_exit_at.argtypes = [C.c_void_p, C.c_int]
L.pgd_debug_step_many.argtypes = [C.c_void_p] * 6 + [C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else 'uniform'
descs = bank.get_descriptions(range(1000, 1100))
mb = mapdata.MapBank(descs); sb = scenario.ScenarioBank(descs, [d['seed'] for d in descs], traffic_mode=os.environ.get('TRAFFIC', 'trigger'))
eng = engine.Engine(_abi.make_config(N, seed=1234), mb, sb)
eng.reset(np.arange(N) % 100)
rng = np.random.default_rng(0)
if mode in ('uniform', 'expert'):
    acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, 1, 2)).astype(np.float32)).cuda()
else:
    a = np.zeros((64, N, 1, 2), np.float32); a[..., 1] = 1.0; a[..., 0] = rng.normal(0, 0.05, size=(64, N, 1)); acts = torch.from_numpy(a).cuda()
names = {99: 'entry', 13: 'loads issued+staged', 0: 'trigger', 1: 'snapshot', 4: 'policy+dynamics+crash', 5: 'after_step+state_check',
         6: 'reward/done', 7: 'reset', 8: 'store', 20: 'obs publish', 14: 'obs done', -1: 'full'}
with torch.cuda.stream(eng.stream):
    abuf = torch.zeros((N, 1, 2), dtype=torch.float32, device='cuda')
    for k in range(1500):
        if mode == 'expert' and k > 0:
            eng.lane_keep_actions(abuf, k); eng.step(abuf)
        else:
            eng.step(acts[k % 64])
    eng.sync()
    if mode == 'expert': acts[0].copy_(abuf)
    f, i, ei = eng.get_state()
    print('driving traffic per env %.2f, ego km/h %.1f' % ((i[0, :, 1:] == 2).sum(1).mean(), abs(f[3, :, 0]).mean() * 3.6))
    prev = 0.0
    for pt in (99, 13, 0, 1, 4, 5, 6, 7, 8, 20, 14, -1):
        L.pgd_debug_exit_at(eng.h, pt)
        for k in range(50): eng.step(acts[0])
        eng.sync()
        ptrs = [C.c_void_p(t.data_ptr()) for t in (acts[0], eng.obs, eng.reward, eng.done, eng.flags)]
        t0 = time.perf_counter()
        L.pgd_debug_step_many(eng.h, *ptrs, 1000)
        eng.sync()
        us = (time.perf_counter() - t0) / 1000 * 1e6
        print('exit at %-26s %6.2f us   (+%.2f)' % (names[pt], us, us - prev))
        prev = us
