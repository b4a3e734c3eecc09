import sys; sys.path.insert(0,'.')
import numpy as np, torch
from pgdrive_amd import _abi, mapdata, scenario, mapgen
from pgdrive_amd.engine import Engine
N, A = 1024, 40
descs = [mapgen.generate_ma_roundabout()]
mb = mapdata.MapBank(descs); sb = scenario.MarlScenarioBank(descs[0], num_agents=A, n_variants=16, seed=0)
cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=72, num_others=0, lidar_dist=40.0, multi_agent=True, horizon=1000,
                       agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn, out_of_road_penalty=10.0, crash_vehicle_penalty=10.0,
                       crash_object_penalty=10.0, delay_done=25, auto_reset=1, resample_scenario=1, seed=1234)
eng = Engine(cfg, mb, sb); eng.reset(np.arange(N) % len(sb.scenarios))
rng = np.random.default_rng(0)
L = mb.lanes
for t in range(2400):
    act = rng.uniform(-1, 1, size=(N, A, 2)).astype(np.float32)
    eng.step(torch.from_numpy(act).cuda())
    if t % 300 == 299:
        eng.sync(); f, i, ei = eng.get_state()
        on = i[_abi.SI['STATUS']][:, :A] == _abi.ST_ACTIVE
        ln = i[_abi.SI['LANE']][:, :A][on]; x = f[_abi.SF['X']][:, :A][on]; y = f[_abi.SF['Y']][:, :A][on]
        hx = f[_abi.SF['HX']][:, :A][on]; hy = f[_abi.SF['HY']][:, :A][on]
        l = L[ln]; straight = l['dir'] == 0
        # straight: lateral offset and relative heading
        dx, dy = x - l['ax'], y - l['ay']
        lat_s = dy * l['bx'] - dx * l['by']; ca = np.abs(hx * l['bx'] + hy * l['by']); sa = np.abs(hy * l['bx'] - hx * l['by'])
        r = np.hypot(dx, dy); lat_c = l['dir'] * (l['bx'] - r)
        # circular: tangent direction = perpendicular to the radius
        tx, ty = -dy / np.maximum(r, 1e-6), dx / np.maximum(r, 1e-6)
        ca_c = np.abs(hx * tx + hy * ty); sa_c = np.abs(hy * tx - hx * ty)
        lat = np.where(straight, lat_s, lat_c); ca = np.where(straight, ca, ca_c); sa = np.where(straight, sa, sa_c)
        e_lat = 0.9 * ca + 2.2 * sa
        inside = np.abs(lat) + e_lat <= 0.5 * l['width'] - 0.15
        print("t=%d active/env %.1f  on straight %.2f  inside-lane-strip: straight %.2f circular %.2f  all %.2f" % (
            t, on.sum() / N, straight.mean(), inside[straight].mean(), inside[~straight].mean(), inside.mean()))
