import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from pgdrive_amd import _abi, mapdata, scenario, mapgen
from pgdrive_amd.engine import Engine
N, A = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 40
d = [mapgen.generate_ma_roundabout()]
mb = mapdata.MapBank(d)
sb = scenario.MarlScenarioBank(d[0], num_agents=A, n_variants=16, seed=0)
cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=72, num_others=0, lidar_dist=40.0, multi_agent=True, horizon=1000,
                       agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn, out_of_road_penalty=10.0, crash_vehicle_penalty=10.0,
                       crash_object_penalty=10.0, delay_done=25, auto_reset=1, resample_scenario=1, seed=1234)
eng = Engine(cfg, mb, sb)
eng.reset(np.arange(N) % len(sb.scenarios))
rng = np.random.default_rng(0)
acts = torch.from_numpy(rng.uniform(-1, 1, size=(64, N, A, 2)).astype(np.float32)).cuda()
with torch.cuda.stream(eng.stream):
    for blk in range(32):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(100): eng.step(acts[k % 64])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100 * 1e6
        f, i, ei = eng.get_state()
        st = i[_abi.SI['STATUS']][:, :A]
        print('steps %4d: %.1f us/step, active %.2f dying %.2f, episodes %.2f' % ((blk + 1) * 100, dt, (st == _abi.ST_ACTIVE).sum(1).mean(), (st == _abi.ST_DYING).sum(1).mean(), ei[_abi.EI['EPISODES']].mean()))
