# per-phase cycles of the multi-agent k_step (BASELINE config 5: 4096 envs x A agents on the roundabout); -DPGD_PROF build
import sys, os, ctypes as C, numpy as np, subprocess
sys.path.insert(0, '.')
import torch
from pgdrive_amd import _abi, mapdata, scenario, build, mapgen
lib = os.path.join("gpurun_out", "libpgd_prof.so")
subprocess.check_call([build.hipcc(), '--offload-arch=gfx950', *build.OPT, '-std=c++17', *build.FAST_FP, '-shared', '-fPIC', '-DPGD_PROF', '-o', lib, build.SRC] + os.environ.get('PGD_EXTRA', '').split())
from pgdrive_amd import engine
engine._LIBH = None
L = engine.load_library(path=lib); engine._LIBH = L
L.pgd_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lasers = int(sys.argv[3]) if len(sys.argv) > 3 else 72
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
names = ['load', 'trig+snap', 'policy', 'dynamics', 'crash', 'linetest', 'tail_end', 'reset', 'store', 'i_route', 'i_search', 'i_lc', 'i_pid', 'ld_stage', 'obs',
         'WALL', 'as_route', 'as_getlane', 'as_local', 'as_side', 'eo_head', 'eo_init', 'eo_pairs', 'eo_cast', 'eo_rows', 'as_rest', 'm_reward', 'm_respawn', 'ko_load', 'KO_WALL']
out = (C.c_ulonglong * 64)()
with torch.cuda.stream(eng.stream):
    for k in range(1500): eng.step(acts[k % 64])
    L.pgd_debug_phase_cycles(eng.h, out, 1)
    for rep in range(2):
        for k in range(300): eng.step(acts[k % 64])
        L.pgd_debug_phase_cycles(eng.h, out, 1)
        nb = 300 * min(N, 8192)
        tot = sum(out[:15]) + sum(out[16:28])
        print('k_observe cycles/block (first 8192 rows):', {n: int(out[i] / (300 * 8192)) for i, n in enumerate(names) if i in (22, 23, 24, 28)}, 'wall us', round(out[29] / (300 * 8192) / 100, 2))
        print('cycles/block:', {n: int(out[i] / nb) for i, n in enumerate(names) if out[i]}, 'total', int(tot / nb), '=> us', round(out[15] / nb / 100, 2))
        f_, i_, ei_ = eng.get_state(); st_ = i_[_abi.SI['STATUS']][:, :A]
        print('   active agents per env %.2f, dying %.2f, max active %d' % ((st_ == _abi.ST_ACTIVE).sum(1).mean(), (st_ == _abi.ST_DYING).sum(1).mean(), (st_ == _abi.ST_ACTIVE).sum(1).max()))
        print('   MAX over blocks:', {n: int(out[32 + i] / 300) for i, n in enumerate(names) if out[i]})
