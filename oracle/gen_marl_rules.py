"""Fixtures that pin the multi-agent ENV RULES of the oracle to the reference's own classes (TEST INFRASTRUCTURE; runs only in
the build container where /root/reference exists; writes tests/golden/marl_rules_v0.json.gz).

The rules are bookkeeping on top of the simulation, so they are pinned by REPLAY: the oracle plays an episode batch, the
event stream it produced (which agent was active on which block at which episode step; who finished; who was respawned
where) is fed to the reference's classes -- imported from /root/reference under the stubs of oracle/refstub.py, with
duck-typed vehicles -- and the reference's outputs are stored next to the events.  tests/test_marl_rules.py re-plays the same
oracle run, checks that it reproduces the stored events (so the comparison is about the same inputs) and then compares the
oracle's bookkeeping with what the reference's classes answered.

  tollgate   StayTimeManager.record / entry_time / exit_time / last_block (marl_tollgate.py:36-60), TollGateObservation's
             in_toll_time counter and its two toll floats (marl_tollgate.py:76-96), the stay-time clause of
             MultiAgentTollgateEnv.done_function (marl_tollgate.py:262-268)
  parking    ParkingLotSpawnManager: get_parking_space / after_vehicle_done / update_destination_for
             (marl_parking_lot.py:40-90): size of the free pool and who holds a space after every event (the reference
             draws WHICH free space from an unseeded stream, the engine from its counter RNG: the draw itself is not compared)
  respawn    how many agents ONE env.step respawns: SpawnManager.get_available_respawn_places (spawn_manager.py:157-207) offers a
             free place at most once per frame and MultiAgentPGDrive._respawn_vehicles / _respawn_single_vehicle
             (multi_agent_pgdrive.py:180-213) take one place per call -- both run here on occupancy patterns (a stand-in
             rect_region_detection answers from the pattern): newcomers of a frame, of a second call in the same frame and of
             the next frame, and whether the chosen place was a free one
"""
import gzip
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def toll_run():
    """The oracle run of tests/test_marl_rules.py::tollgate (kept in one place: both sides call this)."""
    from oracle import orc
    from pgdrive_amd import _abi, mapdata
    from tests import util
    TOLL = dict(tollgate=True, num_lasers=72, lidar_dist=20.0, side_lasers=72, side_dist=20.0, lane_line_lasers=4, lane_line_dist=20.0,
                min_pass_steps=30, overspeed_penalty=0.5)
    d, mb, sb = util.make_marl_banks(num_agents=40, n_variants=4, kind="tollgate")
    n_envs = 12
    cfg = util.marl_config(n_envs, sb, horizon=400, **TOLL)
    ora = orc.Oracle(cfg, mb, sb)
    ids = np.arange(n_envs) % 4
    ora.reset(ids)
    f, i, ei = ora.get_state()
    SF, SI = _abi.SF, _abi.SI
    rng = np.random.default_rng(3)
    for e in range(n_envs):  # half of the agents start at the mouth of the plaza: crawlers and rushers
        sp = sb.spawns[ids[e] * sb.stride:(ids[e] + 1) * sb.stride]
        for a in range(0, 40, 2):
            route = [int(r) for r in sp[a]["ckpt_road"][:sp[a]["n_ckpt"] - 1]]
            toll = [k for k, rid in enumerate(route) if d["roads"][rid]["block_id"] == "$"]
            if not toll:
                continue
            k = toll[0] - 1
            road = d["roads"][route[k]]
            lane_id = road["first_lane"] + 2 * int(rng.integers(0, road["n_lanes"] // 2 + road["n_lanes"] % 2))
            lane = d["lanes"][lane_id]
            lon = max(lane["length"] - rng.uniform(0.5, 6.0), 0.2)
            x, y = mapdata.lane_position(lane, lon, 0.0)
            th = mapdata.lane_heading_at(lane, lon)
            f[SF["X"], e, a], f[SF["Y"], e, a], f[SF["THETA"], e, a] = x, y, th
            f[SF["LASTX"], e, a], f[SF["LASTY"], e, a] = x, y
            f[SF["LASTHX"], e, a], f[SF["LASTHY"], e, a] = np.cos(th), np.sin(th)
            f[SF["SPEED"], e, a] = rng.choice([0.6, 6.0])
            i[SI["LANE"], e, a], i[SI["CK0"], e, a], i[SI["CK1"], e, a] = lane_id, k, k + 1
    ora.set_state(f, i, ei)
    events = []  # per step: per env: list of (agent_id, block id char of its current road) for the agents active AFTER the step
    book = []    # the oracle's bookkeeping after the step, same order: (in_toll_time, entry, exit, last block)
    rows = []    # per step: per env: for every agent that REPORTED: (agent_id, block during the step's observation, toll floats, done, out_of_road)
    block_of = lambda e, a, st_i: d["roads"][int(sb.spawns[ids[e] * sb.stride + int(st_i[SI["SPAWN"], e, a])]["ckpt_road"][int(st_i[SI["CK0"], e, a])])]["block_id"]
    for t in range(190):
        act = np.zeros((n_envs, 40, 2), dtype=np.float32)
        act[..., 0] = np.clip(rng.normal(0, 0.03, size=(n_envs, 40)), -1, 1)
        act[..., 1] = np.where(np.arange(40) % 4 == 0, 0.02, 0.4)[None, :]
        f0, i0, ei0 = ora.get_state()
        obs, rew, done, flags = ora.step(act)
        f1, i1, ei1 = ora.get_state()
        ev_t, bk_t, rw_t = [], [], []
        for e in range(n_envs):
            rst = bool((flags[e] & _abi.F_RESET).any())
            ev, bk, rw = [], [], []
            for a in range(40):
                if flags[e, a] & _abi.F_REPORT:
                    # the row was observed on the post-physics state of the agent that drove: its record before finish / respawn.
                    # The block it stood on is not in the post-step state any more when the slot was recycled, so it is taken
                    # from the toll floats' own input: the oracle's observation row ends with [in plaza, stayed long enough]
                    rw.append([int(f0[SF["AGENT_ID"], e, a]), float(obs[e, a, -2]), float(obs[e, a, -1]), int(done[e, a]),
                               int(bool(flags[e, a] & _abi.F_OUT_OF_ROAD))])
                if not rst and i1[SI["STATUS"], e, a] == _abi.ST_ACTIVE:
                    ev.append([int(f1[SF["AGENT_ID"], e, a]), block_of(e, a, i1)])
                    bk.append([float(f1[SF["PID_HP"], e, a]), float(f1[SF["PID_HI"], e, a]), float(f1[SF["PID_LP"], e, a]),
                               float(f1[SF["PID_LI"], e, a])])
            ev_t.append(dict(reset=rst, ep_steps=int(ei1[_abi.EI["EP_STEPS"], e]), active=ev))
            bk_t.append(bk)
            rw_t.append(rw)
        events.append(ev_t)
        book.append(bk_t)
        rows.append(rw_t)
    ora.close()
    return dict(n_envs=n_envs, events=events, book=book, rows=rows)


def toll_reference(run):
    """StayTimeManager + TollGateObservation of the reference on the oracle's event stream."""
    from oracle import refstub
    refstub.load()
    from pgdrive.envs.marl_envs.marl_tollgate import StayTimeManager, TollGateObservation
    out, obs_replay, stay_done = [], [], []
    mgrs = [StayTimeManager() for _ in range(run["n_envs"])]
    observers = [dict() for _ in range(run["n_envs"])]  # agent id -> TollGateObservation (one per agent, multi_agent_pgdrive.py:196-204)
    for ev_t in run["events"]:
        o_t, r_t, d_t = [], [], []
        for e, ev in enumerate(ev_t):
            # the stay-time clause of done_function (marl_tollgate.py:262-268) sees the manager as the PREVIOUS step left it
            d_t.append(sorted(int(k[5:]) for k in mgrs[e].entry_time
                              if k in mgrs[e].exit_time and mgrs[e].exit_time[k] - mgrs[e].entry_time[k] < 30))
            if ev["reset"]:
                mgrs[e].reset()  # MultiAgentTollgateEnv.reset (marl_tollgate.py:180-183)
                observers[e] = dict()
                o_t.append(None)
                r_t.append(None)
                continue
            agents, rr = {}, []
            for aid, blk in ev["active"]:
                v = types.SimpleNamespace(current_road=types.SimpleNamespace(block_ID=(lambda b=blk: b)), config=dict(min_pass_steps=30))
                agents["agent%d" % aid] = v
                if aid not in observers[e]:
                    o = TollGateObservation.__new__(TollGateObservation)
                    o.in_toll_time = 0
                    o.state_observe = lambda v: np.zeros(0)
                    o.lidar_observe = lambda v: []
                    observers[e][aid] = o
                row = TollGateObservation.observe(observers[e][aid], v)  # one call per step the agent is observed
                rr.append([float(row[-2]), float(row[-1]), int(observers[e][aid].in_toll_time)])
            mgrs[e].record(agents, ev["ep_steps"])
            o_t.append([[mgrs[e].entry_time.get("agent%d" % aid, -1), mgrs[e].exit_time.get("agent%d" % aid, -1),
                         mgrs[e].last_block.get("agent%d" % aid, None)] for aid, _ in ev["active"]])
            r_t.append(rr)
        out.append(o_t)
        obs_replay.append(r_t)
        stay_done.append(d_t)
    # TollGateObservation's counter and its two floats on a synthetic block sequence per agent (state / lidar parts patched out)
    obs_cases = []
    rng = np.random.default_rng(5)
    for case in range(12):
        seq = [">"] * int(rng.integers(1, 6)) + ["y"] * int(rng.integers(2, 8)) + ["$"] * int(rng.integers(5, 60)) + \
              ["Y"] * int(rng.integers(2, 8))
        o = TollGateObservation.__new__(TollGateObservation)
        o.in_toll_time = 0
        o.state_observe = lambda v: np.zeros(0)
        o.lidar_observe = lambda v: []
        outs = []
        for blk in seq:
            v = types.SimpleNamespace(current_road=types.SimpleNamespace(block_ID=(lambda b=blk: b)), config=dict(min_pass_steps=30))
            r = TollGateObservation.observe(o, v)
            outs.append([float(r[-2]), float(r[-1]), int(o.in_toll_time)])
        obs_cases.append(dict(seq=seq, out=outs))
    return dict(stay=out, obs_replay=obs_replay, stay_done=stay_done, obs_cases=obs_cases)


def parking_run():
    from oracle import orc
    from pgdrive_amd import _abi
    from tests import util
    d, mb, sb = util.make_marl_banks(num_agents=10, n_variants=4, kind="parking")
    n_envs = 6
    cfg = util.marl_config(n_envs, sb, horizon=300, parking=True, enable_reverse=True)
    ora = orc.Oracle(cfg, mb, sb)
    ids = np.arange(n_envs) % 4
    ora.reset(ids)
    SF, SI = _abi.SF, _abi.SI
    rng = np.random.default_rng(8)
    A = sb.A
    n_spaces = cfg.respawn_dests
    events = []  # per env: list of ("reset", holders) / ("take", agent_id) / ("done", agent_id, held) in the oracle's order
    pool = []    # the oracle's free-pool size and number of holders after every event
    f, i, ei = ora.get_state()

    def holders(f_, i_, e):
        return sorted(int(f_[SF["AGENT_ID"], e, a]) for a in range(A)
                      if i_[SI["STATUS"], e, a] in (_abi.ST_ACTIVE, _abi.ST_DYING) and f_[SF["PID_HP"], e, a] > 0)

    per_env = [[dict(kind="reset", holders=holders(f, i, e), free=bin(int(ei[_abi.EI["AUX"], e]) & ((1 << n_spaces) - 1)).count("1"))]
               for e in range(n_envs)]
    for t in range(260):
        act = util.marl_actions(rng, n_envs, A)
        if t % 3 == 0:
            act[:, ::2, 0] = 1.0  # hard steering: agents leave the road, finish, free their space, get respawned
        f0, i0, ei0 = ora.get_state()
        obs, rew, done, flags = ora.step(act)
        f1, i1, ei1 = ora.get_state()
        for e in range(n_envs):
            free1 = bin(int(ei1[_abi.EI["AUX"], e]) & ((1 << n_spaces) - 1)).count("1")
            if (flags[e] & _abi.F_RESET).any():
                per_env[e].append(dict(kind="reset", holders=holders(f1, i1, e), free=free1))
                continue
            for a in range(A):  # finishes first (AgentManager.finish), then respawns (multi_agent_pgdrive.py:126-141)
                if (flags[e, a] & _abi.F_REPORT) and done[e, a]:
                    per_env[e].append(dict(kind="done", agent=int(f0[SF["AGENT_ID"], e, a]), held=bool(f0[SF["PID_HP"], e, a] > 0)))
            for a in range(A):
                if flags[e, a] & _abi.F_NEW:
                    per_env[e].append(dict(kind="spawn", agent=int(f1[SF["AGENT_ID"], e, a]), takes=bool(f1[SF["PID_HP"], e, a] > 0)))
            per_env[e].append(dict(kind="check", holders=holders(f1, i1, e), free=free1))
    ora.close()
    return dict(n_envs=n_envs, n_spaces=int(n_spaces), events=per_env)


def parking_reference(run):
    """ParkingLotSpawnManager of the reference on the oracle's event stream: pool size / holders after every event."""
    from oracle import refstub
    refstub.load()
    from pgdrive.envs.marl_envs.marl_parking_lot import ParkingLotSpawnManager
    from pgdrive.component.road.road import Road
    spaces = [Road("P%da" % k, "P%db" % k) for k in range(run["n_spaces"])]
    out = []
    for ev_e in run["events"]:
        m = ParkingLotSpawnManager.__new__(ParkingLotSpawnManager)
        m._parking_spaces = list(spaces)
        m.v_dest_pair = {}
        m.parking_space_available = set(spaces)
        m.np_random = np.random.RandomState(0)
        res = []
        for ev in ev_e:
            if ev["kind"] == "reset":
                m.v_dest_pair = {}
                m.parking_space_available = set(spaces)
                for aid in ev["holders"]:
                    ParkingLotSpawnManager.get_parking_space(m, "agent%d" % aid)
            elif ev["kind"] == "done":
                ParkingLotSpawnManager.after_vehicle_done(m, "agent%d" % ev["agent"])
            elif ev["kind"] == "spawn":
                if ev["takes"]:
                    ParkingLotSpawnManager.get_parking_space(m, "agent%d" % ev["agent"])
            res.append([len(m.parking_space_available), sorted(int(k[5:]) for k in m.v_dest_pair)])
        out.append(res)
    return out


def respawn_reference():
    """The reference's own respawn loop on occupancy patterns over P = 8 places (which places' 8 m x 3 m regions hold a vehicle)."""
    from oracle import refstub
    refstub.load()
    import pgdrive.envs.marl_envs.marl_parking_lot  # noqa: F401  (the import order that avoids the reference's circular imports)
    from pgdrive.manager import spawn_manager as sm_mod
    from pgdrive.manager.spawn_manager import SpawnManager
    from pgdrive.envs.marl_envs.multi_agent_pgdrive import MultiAgentPGDrive
    from pgdrive.utils import Config

    P = 8
    occupied = set()

    class _Hit:
        def __init__(self, hit):
            self._hit = hit

        def hasHit(self):
            return self._hit

    class _Engine:
        global_config = dict(debug=False, debug_physics_world=False)

    def fake_region_detection(engine, position, heading, lon, lat, mask, *a, **k):
        assert (lon, lat) == (SpawnManager.RESPAWN_REGION_LONGITUDE, SpawnManager.RESPAWN_REGION_LATERAL) == (8.0, 3.0)
        return _Hit(int(round(position[0] / 100.0)) in occupied)  # place p sits at x = 100 p

    sm_mod.rect_region_detection = fake_region_detection
    sm_mod.get_engine = lambda: _Engine

    def fresh_manager():
        m = SpawnManager.__new__(SpawnManager)
        m.spawn_places_used = []
        m.safe_spawn_places = {
            "place%d" % p: Config(dict(identifier="place%d" % p, config=dict(spawn_lane_index=("a", "b", p), spawn_longitude=4.0,
                                                                           spawn_lateral=0.0),
                                       spawn_point_position=(100.0 * p, 0.0), spawn_point_heading=0.0), unchangeable=False)
            for p in range(P)}
        m.update_destination_for = lambda agent_id, cfg: cfg
        return m

    class _Vehicle:
        def __init__(self):
            self.config = {}

        def reset(self):
            pass

        def after_step(self):
            pass

    class _Obs:
        def observe(self, v):
            return "row"

    class _AgentManager:
        allow_respawn = True

        def __init__(self):
            self.count = 0

        def propose_new_vehicle(self):
            self.count += 1
            return "agent%d" % (100 + self.count), _Vehicle()

    class _Env:
        _DEBUG_RANDOM_SEED = None
        current_map = None
        _respawn_single_vehicle = MultiAgentPGDrive._respawn_single_vehicle

        def __init__(self, m):
            self.engine = types.SimpleNamespace(spawn_manager=m)
            self.agent_manager = _AgentManager()
            self.dones = {}

            class _D(dict):
                def __missing__(self, k):
                    return _Obs()
            self.observations = _D()

    cases = []
    rng = np.random.RandomState(5)
    patterns = [[], list(range(P)), [0], [1, 2, 3, 4, 5, 6, 7]] + [sorted(rng.choice(P, size=int(rng.randint(1, P)), replace=False).tolist())
                                                                  for _ in range(20)]
    for occ in patterns:
        occupied.clear()
        occupied.update(occ)
        m = fresh_manager()
        env = _Env(m)
        first = MultiAgentPGDrive._respawn_vehicles(env, randomize_position=False)
        chosen = [int(v.config["spawn_lane_index"][2]) for v in []]  # (the vehicles are not kept: read the places from the manager)
        offered = [int(b[5:]) for b in m.spawn_places_used]
        again = MultiAgentPGDrive._respawn_vehicles(env, randomize_position=False)   # same frame: every free place was offered already
        SpawnManager.step(m)                                                          # next frame
        nxt = MultiAgentPGDrive._respawn_vehicles(env, randomize_position=False)
        cases.append(dict(places=P, occupied=list(occ), newcomers=len(first), offered_in_first_call=sorted(offered),
                          newcomers_second_call_same_frame=len(again), newcomers_next_frame=len(nxt)))
        del chosen
    return cases


def main():
    t = toll_run()
    tr = toll_reference(t)
    p = parking_run()
    pr = parking_reference(p)
    out = dict(toll=dict(events=t["events"], reference=tr), parking=dict(events=p["events"], n_spaces=p["n_spaces"], reference=pr),
               respawn=respawn_reference())
    path = os.path.join(ROOT, "tests", "golden", "marl_rules_v0.json.gz")
    with gzip.open(path, "wt") as fh:
        json.dump(out, fh)
    print("wrote", path, os.path.getsize(path), "bytes;",
          "toll steps", len(t["events"]), "parking events", sum(len(e) for e in p["events"]))


if __name__ == "__main__":
    main()
