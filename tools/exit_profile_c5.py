"""Launch time of the multi-agent k_step up to each top-level point of the step (-DPGD_EXITAT build: every wave returns at the chosen
mark, nothing is stored: the state stays the snapshot reached by the warm-up).  usage: exit_profile_c5.py [N] [AGENTS] [BEAMS] [WARMUP]"""
import sys, os, ctypes as C, numpy as np, subprocess, time
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, mapdata, scenario, build, mapgen
lib = os.path.join("gpurun_out", "libpgd_exit.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_EXITAT',
                       '-o', lib, build.SRC] + os.environ.get('PGD_EXTRA', '').split())
from pgdrive_amd import engine
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_exit_at.argtypes = [C.c_void_p, C.c_int]
L.pgd_debug_step_many.argtypes = [C.c_void_p] * 6 + [C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lasers = int(sys.argv[3]) if len(sys.argv) > 3 else 72
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
descs = [mapgen.generate_ma_roundabout()]
mb = mapdata.MapBank(descs)
sb = scenario.MarlScenarioBank(descs[0], num_agents=A, n_variants=16, seed=0)
cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=lasers, num_others=0, lidar_dist=40.0, multi_agent=True, horizon=1000,
                       agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn, out_of_road_penalty=10.0, crash_vehicle_penalty=10.0,
                       crash_object_penalty=10.0, delay_done=25, auto_reset=1, resample_scenario=1, seed=1234)
eng = engine.Engine(cfg, mb, sb)
eng.reset(np.arange(N) % len(sb.scenarios))
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, A, 2)).astype(np.float32)).cuda()
names = {99: 'entry', 13: 'loads issued+staged', 0: 'trigger', 1: 'snapshot', 4: 'policy+dynamics+crash', 5: 'after_step+state_check',
         6: 'reward/done/respawn', 7: 'reset', 8: 'store', 20: 'obs publish', 14: 'obs done', -1: 'full'}
with torch.cuda.stream(eng.stream):
    for k in range(warm): eng.step(acts[k % 64])
    eng.sync()
    f, i, ei = eng.get_state()
    print('agents alive per env %.2f' % ((i[0, :, :A] == _abi.ST_ACTIVE).sum(1).mean()))
    prev = 0.0
    for pt in (99, 13, 0, 1, 4, 5, 6, 7, 8, 14, -1):
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
