"""Bit-level fingerprint of a free-running rollout (per step: sha1 of obs / reward / done / flags): two builds of the engine that
claim identical results are run on the same seeded inputs (PGD_LIB picks the build) and their fingerprint files compared."""
import sys, hashlib, json; sys.path.insert(0, '.')
import numpy as np, torch
from tests import util
from pgdrive_amd import _abi, bank, mapdata, scenario, mapgen
from pgdrive_amd.engine import Engine
out, mode = sys.argv[1], sys.argv[2]
rng = np.random.default_rng(5)
if mode == "c5":
    N, A, steps = 512, 40, 700
    descs = [mapgen.generate_ma_roundabout()]
    mb = mapdata.MapBank(descs); sb = scenario.MarlScenarioBank(descs[0], num_agents=A, n_variants=16, seed=0)
    cfg = _abi.make_config(N, num_agents=A, num_traffic=0, num_lasers=72, num_others=0, lidar_dist=40.0, multi_agent=True, horizon=1000,
                           agent_limit=A, respawn_places=sb.P, respawn_dests=sb.Dn, out_of_road_penalty=10.0, crash_vehicle_penalty=10.0,
                           crash_object_penalty=10.0, delay_done=25, auto_reset=1, resample_scenario=1, seed=1234)
    ids = np.arange(N) % len(sb.scenarios)
else:
    N, A, steps = 1024, 1, 400
    descs = bank.load_descriptions()
    mb, sb = util.make_banks(descs, n_maps=50, traffic_mode="respawn" if mode == "respawn" else "trigger")
    cfg = _abi.make_config(N, num_agents=1, num_traffic=16, num_lasers=240, auto_reset=1, seed=11)
    ids = np.arange(N) % 50
eng = Engine(cfg, mb, sb); eng.reset(ids)
fp = []
for t in range(steps):
    act = util.driving_actions(rng, N) if A == 1 else rng.uniform(-1, 1, size=(N, A, 2)).astype(np.float32)
    o, r, dn, fl = eng.step(torch.from_numpy(act).cuda()); eng.sync()
    h = hashlib.sha1()
    for x in (o, r, dn, fl): h.update(x.cpu().numpy().tobytes())
    fp.append(h.hexdigest())
    if len(sys.argv) > 3 and t < int(sys.argv[3]):
        np.savez(out + ".step%d.npz" % t, o=o.cpu().numpy()[:128], r=r.cpu().numpy()[:128], dn=dn.cpu().numpy()[:128], fl=fl.cpu().numpy()[:128])
json.dump(fp, open(out, 'w'))
print(mode, steps, fp[-1])
