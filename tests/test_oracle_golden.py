"""Pins the CPU oracle (oracle/pgd_oracle.c) against golden vectors produced by the reference's own Python
(oracle/gen_golden.py -> tests/golden/routines_v0.npz, scenes_v0.json).  fp64 oracle vs fp64 reference: 1e-9 abs
(SURVEY.md §8c).  Every geometry-dependent test runs twice: on the oracle's float64 table path (orc.Oracle(..., f64=True): the
float fields of lanes, boxes, map header and spawn records keep the float64 values of the map description) at 1e-9, and on the
ABI's float32 records -- the numbers the engine is given, what the GPU parity tests compare with -- at 1e-4 m / 2e-6."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from pgdrive_amd import _abi, mapdata, scenario

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    from oracle import orc
    orc.build()
    return orc.lib()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "routines_v0.npz"))


@pytest.fixture(scope="module")
def scenes():
    with open(os.path.join(GOLD, "scenes_v0.json")) as f:
        return json.load(f)


def test_scalar_helpers(L, gold):
    """math_utils.wrap_to_pi / not_zero / norm / clip (utils/math_utils.py:32-94, cutils.pyx:147-154)."""
    y = np.array([L.orc_wrap_to_pi(float(v)) for v in gold["wrap_x"]])
    assert np.abs(y - gold["wrap_y"]).max() < 1e-12
    y = np.array([L.orc_not_zero(float(v), 1e-2) for v in gold["nz_x"]])
    assert (y == gold["nz_y"]).all()
    y = np.array([L.orc_not_zero(float(v), 0.0) for v in gold["nz_x"]])
    assert (y == gold["nz0_y"]).all()
    y = np.array([L.orc_norm(float(a), float(b)) for a, b in gold["norm_x"]])
    assert np.abs(y - gold["norm_y"]).max() < 1e-12
    y = np.array([L.orc_clip(float(v), -1.0, 1.0) for v in gold["clip_x"]])
    assert (y == gold["clip_y"]).all()


@pytest.mark.parametrize("seed", [1000, 1003])
def test_lane_closed_forms(L, gold, descs, seed):
    """StraightLane / CircularLane position, local_coordinates, heading_at, distance on every lane of two maps."""
    d = [m for m in descs if m["seed"] == seed][0]
    lanes = mapdata.pack_lanes(d, mapdata.build_successors(d))
    gin, gout = gold["lane_%d_in" % seed], gold["lane_%d_out" % seed]
    buf = (C.c_double * 2)()
    worst = 0.0
    for row, ref in zip(gin, gout):
        lp = lanes[int(row[0]):int(row[0]) + 1].ctypes.data_as(C.c_void_p)
        L.orc_lane_position(lp, float(row[1]), float(row[2]), buf)
        worst = max(worst, abs(buf[0] - ref[0]), abs(buf[1] - ref[1]))
        L.orc_lane_local(lp, float(ref[0]), float(ref[1]), buf)
        worst = max(worst, abs(buf[0] - ref[2]), abs(buf[1] - ref[3]))
        h = L.orc_lane_heading(lp, float(row[1]))
        worst = max(worst, abs(h - ref[4]))
        L.orc_lane_local(lp, float(row[3]), float(row[4]), buf)
        worst = max(worst, abs(buf[0] - ref[5]), abs(buf[1] - ref[6]))
        worst = max(worst, abs(L.orc_lane_distance(lp, float(row[3]), float(row[4])) - ref[7]))
    # lane records are float32: 500 m * 2^-24 = 3e-5 m
    assert worst < 1e-4, worst
    # the same closed forms on the float64 lane parameters: the restatement itself, at SURVEY 8c's 1e-9
    l64 = mapdata.pack_lanes_f64(d)
    worst = 0.0
    for row, ref in zip(gin, gout):
        lp = l64[int(row[0]):int(row[0]) + 1].ctypes.data_as(C.c_void_p)
        L.orc_lane_position_f64(lp, float(row[1]), float(row[2]), buf)
        worst = max(worst, abs(buf[0] - ref[0]), abs(buf[1] - ref[1]))
        L.orc_lane_local_f64(lp, float(ref[0]), float(ref[1]), buf)
        worst = max(worst, abs(buf[0] - ref[2]), abs(buf[1] - ref[3]))
        worst = max(worst, abs(L.orc_lane_heading_f64(lp, float(row[1])) - ref[4]))
        L.orc_lane_local_f64(lp, float(row[3]), float(row[4]), buf)
        worst = max(worst, abs(buf[0] - ref[5]), abs(buf[1] - ref[6]))
        worst = max(worst, abs(L.orc_lane_distance_f64(lp, float(row[3]), float(row[4])) - ref[7]))
    print("lane closed forms, float64 parameters: worst", worst)
    assert worst < 1e-9, worst


def test_pid(L, gold):
    """PIDController.get_result (vehicle_module/PID_controller.py:10-17) with the IDM gains (idm_policy.py:187-188)."""
    for name, gains in (("h", (1.7, 0.01, 3.5)), ("l", (0.3, 0.002, 0.05))):
        st = (C.c_double * 2)(0.0, 0.0)
        y = np.array([L.orc_pid(st, gains[0], gains[1], gains[2], float(e)) for e in gold["pid_%s_err" % name]])
        assert np.abs(y - gold["pid_%s_out" % name]).max() < 1e-12


def test_idm_law(L, gold):
    """IDMPolicy.acceleration / desired_gap (policy/idm_policy.py:254-271)."""
    y = np.array([L.orc_idm_acc(r[0], r[1], int(r[2]), r[3], r[4], r[5], r[6]) for r in gold["idm_in"]])
    ref = gold["idm_out"]
    assert np.abs(y - ref).max() / max(1.0, np.abs(ref).max()) < 1e-12
    assert np.allclose(y, ref, rtol=1e-10, atol=1e-9)


def test_bicycle(L, gold):
    """highway_vehicle.kinematics.Vehicle.step (kinematics.py:134-156): the dynamics oracle."""
    for traj, acts in zip(gold["bike_traj"], gold["bike_act"]):
        s = (C.c_double * 4)(*traj[0])
        for t, a in enumerate(acts):
            L.orc_bicycle(s, float(a[0]), float(a[1]), 2.46894, 0.02)
            assert np.abs(np.array(s[:]) - traj[t + 1]).max() < 1e-9


def test_before_step_and_energy(L, descs):
    """BaseVehicle.before_step / _set_action / _set_incremental_action / _apply_throttle_brake (base_vehicle.py:238-253,
    343-376) and _update_energy_consumption + the step-info floats of after_step (:255-290): the reference's own methods on
    a recording `system` (tests/golden/stepinfo_v0.json, oracle/gen_golden.py::gen_stepinfo) against the oracle's
    before_step_vehicle, action_forces (what `dynamics` applies) and energy_step (what after_step_vehicle accumulates)."""
    from oracle import orc
    with open(os.path.join(GOLD, "stepinfo_v0.json")) as f:
        g = json.load(f)
    SF = _abi.SF
    d = descs[0]
    mb = mapdata.MapBank([d])
    sb = scenario.ScenarioBank([d], [d["seed"]], num_agents=1, num_traffic=0)
    o = orc.Oracle(_abi.make_config(1, num_agents=1, num_traffic=0, num_lasers=0), mb, sb)
    o.reset(np.zeros(1, dtype=np.int32))
    out2 = (C.c_double * 2)()
    for c in g["before_step"]:
        f, i, ei = o.get_state()
        f[SF["X"], 0, 0], f[SF["Y"], 0, 0], f[SF["THETA"], 0, 0] = c["x"], c["y"], c["theta"]
        f[SF["SPEED"], 0, 0] = c["speed_kmh"] / 3.6
        f[SF["STEER"], 0, 0] = c["steering0"]
        f[SF["ACT1S"], 0, 0], f[SF["ACT1T"], 0, 0] = c["prev_action"]
        o.set_state(f, i, ei)
        L.orc_before_step(o.h, 0, 0, c["action"][0], c["action"][1], int(c["increment_steering"]))
        f, i, ei = o.get_state()
        got = lambda k: f[SF[k], 0, 0]  # noqa: E731
        assert abs(got("STEER") - c["steering"]) < 1e-12 and abs(got("THROTTLE") - c["throttle_brake"]) < 1e-12
        assert [got("LASTX"), got("LASTY")] == pytest.approx(c["last_position"], abs=1e-12)
        assert [got("LASTHX"), got("LASTHY")] == pytest.approx(c["last_heading_dir"], abs=1e-12)
        assert [[got("ACT0S"), got("ACT0T")], [got("ACT1S"), got("ACT1T")]] == c["deque"]
        assert c["raw_action"] == c["action"]
        # what Bullet's raycast vehicle is handed: the steering angle on both front wheels, engine force and brake on all four
        assert c["steer_value_deg"][0] == c["steer_value_deg"][1] == pytest.approx(c["steering"] * c["max_steering_deg"], abs=1e-12)
        L.orc_action_forces(c["max_engine_force"], c["max_brake_force"], c["max_speed"], c["speed_kmh"] / 3.6,
                            c["throttle_brake"], int(c["enable_reverse"]), out2)
        assert all(abs(v - out2[0]) < 1e-9 for v in c["engine_force"]) and all(abs(v - out2[1]) < 1e-9 for v in c["brake"])
    for c in g["energy"]:
        step = L.orc_energy_step(c["speed_kmh"], c["last"][0] - c["pos"][0], c["last"][1] - c["pos"][1])
        assert abs(step - c["step_energy"]) < 1e-12 * max(1.0, c["step_energy"])
        assert abs(c["e0"] + step - c["episode_energy"]) < 1e-9
    # the running sum over a trajectory, through the oracle's real after_step (orc_refresh = engine.after_step on a state):
    # the golden displacement and speed of every step placed at the spawn point of a real map
    f0, i0, ei0 = o.get_state()
    x0, y0 = f0[SF["X"], 0, 0], f0[SF["Y"], 0, 0]
    total, last = 0.0, (0.0, 0.0)
    for c in g["trajectory"]:
        f, i, ei = o.get_state()
        f[SF["X"], 0, 0], f[SF["Y"], 0, 0] = x0, y0
        f[SF["LASTX"], 0, 0], f[SF["LASTY"], 0, 0] = x0 - (c["pos"][0] - last[0]), y0 - (c["pos"][1] - last[1])
        f[SF["SPEED"], 0, 0] = c["speed_kmh"] / 3.6
        f[SF["ENERGY"], 0, 0] = total
        o.set_state(f, i, ei)
        o.refresh()
        f, i, ei = o.get_state()
        assert abs(f[SF["ENERGY"], 0, 0] - total - c["step_energy"]) < 1e-9
        total = f[SF["ENERGY"], 0, 0]
        last = c["pos"]
        assert abs(total - c["episode_energy"]) < 1e-7 and c["velocity"] == c["speed_kmh"]
    o.close()


def test_topdown_reference_arithmetic(L, descs):
    """The arithmetic of the reference's top-down observation, recorded from its own Python under a pygame that only records
    (oracle/gen_topdown.py -> tests/golden/topdown_v0.json), against the oracle's top-down restatement (which the GPU rasteriser
    is compared with pixel by pixel in tests/test_topdown_gpu.py): frame / past-position history and stack order, grey levels,
    window scale, heading snap, the past-position transform.  pygame's rasterisation rules themselves stay unpinned."""
    from oracle import orc
    with open(os.path.join(GOLD, "topdown_v0.json")) as f:
        g = json.load(f)
    c = g["config"]
    R, DIST, FS, PS, SKIP = c["resolution"], c["distance"], c["frame_stack"], c["post_stack"], c["frame_skip"]
    SF, SI = _abi.SF, _abi.SI
    # (1) history lengths and which entries are shown: ages 0, skip, 2 skip ... of what has been recorded so far
    assert g["deque_maxlen"] == dict(traffic=(FS - 1) * SKIP + 1, past_pos=(PS - 1) * SKIP + 1)
    for case in g["stack_indices"]:
        n, k = case["length"], case["frame_skip"]
        ages = [n - 1 - i for i in case["indices"]]
        assert ages == [q * k for q in range(n) if q * k < n]
    # (2) grey levels: the engine's three constants are the reference's _transform of its own colours (road channel doubled)
    gr = g["grey"]
    TD_LINE, TD_NAVI, TD_VEH = 2 * 35 / 255, 2 * 64 / 255, (0.299 * 100 + 0.587 * 200 + 0.114 * 255) / 255
    assert abs(2 * gr["lane_line_clip"] - TD_LINE) < 1e-7 and abs(2 * gr["navigation_clip"] - TD_NAVI) < 1e-7
    assert abs(gr["vehicle_blue_clip"] - TD_VEH) < 1e-7 and g["geometry"][0]["navigation_color"] == [64, 64, 64]
    assert all(s["road"] == pytest.approx(TD_LINE, abs=1e-7) for s in g["observe"]["steps"])  # road_network * 2 of a (35, 35, 35) canvas
    # (3) window scale: px per metre of the cropped, rotated, zoomed window on the reference's own road networks
    for geo in g["geometry"]:
        for ch in ("traffic", "road"):
            assert abs(R / (2.0 * DIST) / geo["window_px_per_m"][ch] - 1.0) < 0.006  # int() truncations of the canvas sizes: < 0.6 %
        assert all(abs(rz["angle"] - (np.rad2deg(geo["heading"]) + 90.0)) < 1e-9 for rz in geo["rotozoom"])
        assert geo["crop"][1][2:] == [R, R] and geo["crop"][1][0] == geo["crop"][1][1]  # centre crop
    # (4) heading snap: others and past positions snapped at 2 degrees, the window rotation is not
    for sc in g["scene"]:
        fr = sc["frames"][0]
        assert [L.orc_topdown_snap(h) for h in sc["other_headings"]] == fr["vehicle_headings"]
        assert fr["window_heading"] == sc["ego_heading"] and fr["vehicle_color"] == [100, 200, 255]
        assert all(r["angle"] == pytest.approx(np.rad2deg(L.orc_topdown_snap(sc["ego_heading"])) + 90.0) for f_ in sc["frames"] for r in f_["rotate"])
        assert sc["past_pos_scaling"] == R / DIST
    # the oracle itself, one env on a real map
    d = descs[0]
    mb = mapdata.MapBank([d])
    sb = scenario.ScenarioBank([d], [d["seed"]], num_agents=1, num_traffic=1, density=0.0)
    o = orc.Oracle(_abi.make_config(1, num_agents=1, num_traffic=1, num_lasers=0, auto_reset=0), mb, sb)
    o.enable_topdown(_abi.make_topdown_config(R, float(DIST), FS, PS, SKIP))
    o.reset(np.zeros(1, dtype=np.int32))
    f0, i0, ei0 = o.get_state()
    ex, ey, eth = f0[SF["X"], 0, 0], f0[SF["Y"], 0, 0], f0[SF["THETA"], 0, 0]

    def place(f, i, slot, x, y, th):
        f[SF["X"], 0, slot], f[SF["Y"], 0, slot], f[SF["THETA"], 0, slot] = x, y, th
        f[SF["HX"], 0, slot] = f[SF["HY"], 0, slot] = 0.0
        i[SI["STATUS"], 0, slot] = _abi.ST_PENDING if slot else _abi.ST_ACTIVE
        i[SI["SPAWN"], 0, slot] = slot

    # (5) frame history through the images: the traffic vehicle stands d(t) metres ahead of a standing ego at step t; the row of
    # its box in channel 2 + k tells which step's frame the channel shows -- the table the reference's observe() produced
    sb.spawns["length"][1], sb.spawns["width"][1], sb.spawns["lane"][1] = 4.5, 1.8, sb.spawns["lane"][0]
    o.L.orc_upload_scenarios(o.h, sb.scenarios.ctypes.data_as(C.c_void_p), len(sb.scenarios), sb.spawns.ctypes.data_as(C.c_void_p))
    dist_of = lambda t: 4.0 + 1.5 * (t % 16)  # noqa: E731
    spx = R / (2.0 * DIST)
    for st in g["observe"]["steps"]:
        t = st["t"]
        if st["reset"]:
            o.reset(np.zeros(1, dtype=np.int32))
        f, i, ei = o.get_state()
        place(f, i, 0, ex, ey, eth)
        dd = dist_of(t)
        place(f, i, 1, ex + dd * np.cos(eth), ey + dd * np.sin(eth), eth)
        o.set_state(f, i, ei)
        img = o.observe_topdown()[0]
        for k in range(FS):
            rows = np.nonzero(img[:, R // 2, 2 + k])[0]
            assert len(rows) > 0 and abs(img[rows[0], R // 2, 2 + k] - TD_VEH) < 1e-12
            d_est = (R / 2 - 0.5 - rows.mean()) / spx
            cands = [s_ for s_ in range(max(0, t - (FS - 1) * SKIP), t + 1)]
            src = min(cands, key=lambda s_: abs(dist_of(s_) - d_est))
            assert abs(dist_of(src) - d_est) < 0.6 and src == st["traffic_source"][k], (t, k, src, st["traffic_source"])
    # (6) past positions: the ego along the fixture's track, heading -90 / +90 degrees (rotation by 0 / 180 degrees)
    for sc in g["scene"][:2]:
        o.reset(np.zeros(1, dtype=np.int32))
        for fr in sc["frames"]:
            f, i, ei = o.get_state()
            # (the fixture's track is in free space: moved onto the map by the offset of its first point)
            x = ex + fr["ego"][0] - sc["frames"][0]["ego"][0]
            y = ey + fr["ego"][1] - sc["frames"][0]["ego"][1]
            place(f, i, 0, x, y, sc["ego_heading"])
            place(f, i, 1, ex + 500.0, ey + 500.0, 0.0)
            o.set_state(f, i, ei)
            img = o.observe_topdown()[0]
            want = {(int(np.floor(p[1])), int(np.floor(p[0]))) for p in fr["filled"]
                    if 0 <= p[0] < R and 0 <= p[1] < R}  # surface (x, y) -> image [row = y][col = x]
            got = {(int(a), int(b)) for a, b in zip(*np.nonzero(img[:, :, 1]))}
            assert got == want, (sc["label"], fr["ego"], got, want)
            assert len(fr["filled"]) == len([q for q in range(PS) if q * SKIP < fr["deque_len"]])
    o.close()


def test_ray_box_known_answers(L):
    """SURVEY §8c known answers: empty -> 1.0; a box straight ahead at distance d -> (d - L_other/2)/50 on beam 0."""
    f = L.orc_ray_box(20.0, 0.0, 0.0, 2.25, 0.9, 0.0, 0.0, 50.0, 0.0)
    assert abs(f - (20.0 - 2.25) / 50.0) < 1e-12
    assert L.orc_ray_box(20.0, 0.0, 0.0, 2.25, 0.9, 0.0, 0.0, 0.0, 50.0) == 1.0  # perpendicular beam misses
    assert L.orc_ray_box(20.0, 0.0, 0.0, 2.25, 0.9, 0.0, 0.0, -50.0, 0.0) == 1.0  # behind
    assert L.orc_ray_box(100.0, 0.0, 0.0, 2.25, 0.9, 0.0, 0.0, 50.0, 0.0) == 1.0  # out of range
    assert L.orc_ray_box(1.0, 0.0, 0.3, 2.25, 0.9, 0.0, 0.0, 50.0, 0.0) == 1.0  # origin inside the box: no hit, as Bullet's convex ray cast


# ----------------------------------------------------------------------------------------------------------------------
# full scenes
# ----------------------------------------------------------------------------------------------------------------------
SPAWN_CONST = (2.46894, 1100.0, 800.0, 130.0, 0.9, float(np.deg2rad(40)), 80.0)  # wheelbase .. max_speed of the scenes' vehicles


def spawns_f64(vehicles):
    """[n, 12] float64 values of the spawn records' float fields (ORC_SPAWN_F64 order) for scene vehicles"""
    return np.array([[v["x"], v["y"], v["theta"], v["length"], v["width"], *SPAWN_CONST] for v in vehicles], dtype=np.float64)


PREC = pytest.mark.parametrize("f64", [True, False], ids=["f64-tables", "f32-abi-tables"])


def _scene_engine(descs, sc, f64=False, **cfg_kw):
    """Oracle holding exactly the scene's vehicles (slot 0 = ego) with their routes; state overwritten from the scene."""
    from oracle import orc
    d = [m for m in descs if m["seed"] == sc["seed"]][0]
    mb = mapdata.MapBank([d])
    V = len(sc["vehicles"])
    spawns = np.zeros(V, dtype=scenario.SPAWN_DT)
    rl = mapdata.road_lookup(d)
    for k, v in enumerate(sc["vehicles"]):
        r = spawns[k]
        r["x"], r["y"], r["heading"] = v["x"], v["y"], v["theta"]
        r["length"], r["width"] = v["length"], v["width"]
        r["wheelbase"], r["mass"], r["max_engine_force"], r["max_brake_force"] = 2.46894, 1100, 800, 130
        r["friction"], r["max_steer"], r["max_speed"] = 0.9, np.deg2rad(40), 80
        r["lane"], r["group"], r["n_ckpt"] = v["lane"], -1, len(v["ckpt"])
        r["ckpt"][:] = -1
        r["ckpt_road"][:] = -1
        r["ckpt"][:len(v["ckpt"])] = v["ckpt"]
        roads = [rl[(v["ckpt"][j], v["ckpt"][j + 1])] for j in range(len(v["ckpt"]) - 1)]
        r["ckpt_road"][:len(roads)] = roads
        fr = d["roads"][roads[-1]]
        r["dest_lane"] = fr["first_lane"] + fr["n_lanes"] - 1
    sb = scenario.ScenarioBank.__new__(scenario.ScenarioBank)
    scen = np.zeros(1, dtype=scenario.SCEN_DT)
    scen["trigger_road"][:] = -1
    sb.scenarios, sb.spawns, sb.V, sb.info = scen, spawns, V, []
    sb.spawns64 = spawns_f64(sc["vehicles"])
    cfg = _abi.make_config(1, num_agents=1, num_traffic=V - 1, **cfg_kw)
    o = orc.Oracle(cfg, mb, sb, f64=f64)
    o.reset(np.zeros(1, dtype=np.int32))
    f, i, ei = o.get_state()
    SF, SI = _abi.SF, _abi.SI
    for k, v in enumerate(sc["vehicles"]):
        f[SF["X"], 0, k], f[SF["Y"], 0, k], f[SF["THETA"], 0, k] = v["x"], v["y"], v["theta"]
        f[SF["SPEED"], 0, k] = v["speed_kmh"] / 3.6
        i[SI["STATUS"], 0, k] = _abi.ST_ACTIVE
        i[SI["LANE"], 0, k] = v["lane"]
        i[SI["CK0"], 0, k], i[SI["CK1"], 0, k] = v["idx"]
        i[SI["RLANE"], 0, k] = -1
        f[SF["TARGET_SPEED"], 0, k] = 30.0
    e = sc["ego"]
    f[SF["STEER"], 0, 0] = e["steering"]
    f[SF["ACT0S"], 0, 0], f[SF["ACT0T"], 0, 0] = e["act0"]
    f[SF["LASTHX"], 0, 0], f[SF["LASTHY"], 0, 0] = e["last_heading"]
    f[SF["LASTX"], 0, 0], f[SF["LASTY"], 0, 0] = e["last_position"]
    return o, f, i, ei, d


@PREC
def test_scene_observation(L, descs, scenes, f64):
    """navi info (navigation.py:213-260), vehicle_state (state_obs.py:58-106), neighbour info (lidar.py:55-77) and the
    240-beam fan cast with the reference test's exact intersector (test_detector_mask.py:132-154)."""
    SF = _abi.SF
    worst = dict(lr=0.0, state=0.0, navi=0.0, others=0.0, lidar=0.0)
    flips = 0
    for sc in scenes["scenes"]:
        o, f, i, ei, d = _scene_engine(descs, sc, f64=f64)
        o.set_state(f, i, ei)
        lr = (C.c_double * 2)()
        L.orc_dist_left_right(o.h, 0, 0, lr)
        worst["lr"] = max(worst["lr"], abs(lr[0] - sc["left"]), abs(lr[1] - sc["right"]))
        f[SF["DIST_LEFT"], 0, 0], f[SF["DIST_RIGHT"], 0, 0] = lr[0], lr[1]
        o.set_state(f, i, ei)
        obs = o.observe()[0, 0]
        worst["state"] = max(worst["state"], np.abs(obs[:8] - np.array(sc["state"])).max())
        worst["navi"] = max(worst["navi"], np.abs(obs[8:18] - np.array(sc["navi"])).max())
        worst["others"] = max(worst["others"], np.abs(obs[18:34] - np.array(sc["others"])).max())
        dl = np.abs(obs[34:] - np.array(sc["cloud"]))
        flips += int((dl > 1e-6).sum())  # a beam through a box corner may differ (the reference helper pads edges by 1e-5)
        worst["lidar"] = max(worst["lidar"], float(dl[dl <= 1e-6].max()))
        o.close()
    print(worst, "corner beams", flips)
    if f64:  # fp64 restatement vs the reference's fp64 Python <= 1e-9 (corner beams counted, as below)
        assert max(worst.values()) < 1e-9 and flips <= 3
    assert worst["lr"] < 1e-4 and worst["state"] < 2e-6 and worst["navi"] < 2e-6 and worst["others"] < 2e-6
    assert worst["lidar"] < 1e-6 and flips <= 3


def agents_scene_banks(descs, sc, **cfg_kw):
    """Banks + config + state of a scene whose vehicles are ALL agents of one multi-agent env (tests/golden/maround_v0.json);
    shared with the GPU variant of the test."""
    d = [m for m in descs if m["seed"] == sc["seed"]][0]
    mb = mapdata.MapBank([d])
    V = len(sc["vehicles"])
    spawns = np.zeros(V, dtype=scenario.SPAWN_DT)
    rl = mapdata.road_lookup(d)
    for k, v in enumerate(sc["vehicles"]):
        r = spawns[k]
        r["x"], r["y"], r["heading"] = v["x"], v["y"], v["theta"]
        r["length"], r["width"] = v["length"], v["width"]
        r["wheelbase"], r["mass"], r["max_engine_force"], r["max_brake_force"] = 2.46894, 1100, 800, 130
        r["friction"], r["max_steer"], r["max_speed"] = 0.9, np.deg2rad(40), 80
        r["lane"], r["group"], r["n_ckpt"] = v["lane"], -1, len(v["ckpt"])
        r["ckpt"][:] = -1
        r["ckpt_road"][:] = -1
        r["ckpt"][:len(v["ckpt"])] = v["ckpt"]
        roads = [rl[(v["ckpt"][j], v["ckpt"][j + 1])] for j in range(len(v["ckpt"]) - 1)]
        r["ckpt_road"][:len(roads)] = roads
        fr = d["roads"][roads[-1]]
        r["dest_lane"] = fr["first_lane"] + fr["n_lanes"] - 1
    sb = scenario.ScenarioBank.__new__(scenario.ScenarioBank)
    scen = np.zeros(1, dtype=scenario.SCEN_DT)
    scen["trigger_road"][:] = -1
    sb.scenarios, sb.spawns, sb.V, sb.info = scen, spawns, V, []
    sb.spawns64 = spawns_f64(sc["vehicles"])
    cfg = _abi.make_config(1, num_agents=V, num_traffic=0, multi_agent=True, agent_limit=V, respawn_places=0, respawn_dests=0,
                           allow_respawn=False, auto_reset=0, **cfg_kw)
    return mb, sb, cfg


def agents_scene_state(sc, f, i):
    SF, SI = _abi.SF, _abi.SI
    for k, v in enumerate(sc["vehicles"]):
        f[SF["X"], 0, k], f[SF["Y"], 0, k], f[SF["THETA"], 0, k] = v["x"], v["y"], v["theta"]
        f[SF["HX"], 0, k], f[SF["HY"], 0, k] = np.cos(v["theta"]), np.sin(v["theta"])
        f[SF["SPEED"], 0, k] = v["speed_kmh"] / 3.6
        i[SI["STATUS"], 0, k] = _abi.ST_ACTIVE
        i[SI["LANE"], 0, k] = v["lane"]
        i[SI["CK0"], 0, k], i[SI["CK1"], 0, k] = v["idx"]
        f[SF["STEER"], 0, k] = v["steering"]
        f[SF["ACT0S"], 0, k], f[SF["ACT0T"], 0, k] = v["act0"]
        f[SF["LASTHX"], 0, k], f[SF["LASTHY"], 0, k] = v["last_heading"]
        f[SF["LASTX"], 0, k], f[SF["LASTY"], 0, k] = v["last_position"]
        f[SF["DIST_LEFT"], 0, k], f[SF["DIST_RIGHT"], 0, k] = v["left"], v["right"]


def compare_maround_rows(sc, obs, worst, beam_tol=1e-6):
    """rows [state 18 | 4 x neighbour state 18 | 240 beams] of the reference's LidarStateObservationMARound; returns corner beams"""
    flips = 0
    for r in sc["rows"]:
        ref = np.array(r["row"])
        got = obs[r["slot"]].astype(np.float64)
        assert got.shape == ref.shape
        worst["state"] = max(worst["state"], float(np.abs(got[:18] - ref[:18]).max()))
        worst["others"] = max(worst["others"], float(np.abs(got[18:90] - ref[18:90]).max()))
        dl = np.abs(got[90:] - ref[90:])
        flips += int((dl > beam_tol).sum())
        worst["lidar"] = max(worst["lidar"], float(dl[dl <= beam_tol].max()))
        worst["rows"] += 1
        worst["absent"] += int(sum(1 for q in range(4) if not ref[18 + 18 * q:36 + 18 * q].any()))
    return flips


@PREC
def test_maround_neighbour_state_rows(L, descs, f64):
    """LidarStateObservationMARound.observe run as-is by oracle/gen_golden.py::gen_maround (marl_inout_roundabout.py:66-122):
    the num_others nearest detected vehicles contribute their OWN state vectors in distance order, absent ranks are zeros."""
    from oracle import orc
    with open(os.path.join(GOLD, "maround_v0.json")) as fh:
        gold = json.load(fh)
    worst = dict(state=0.0, others=0.0, lidar=0.0, rows=0, absent=0)
    flips = 0
    for sc in gold["cases"]:
        mb, sb, cfg = agents_scene_banks(descs, sc, num_lasers=240, lidar_dist=50.0, num_others=gold["num_others"], others_state=True)
        o = orc.Oracle(cfg, mb, sb, f64=f64)
        o.reset(np.zeros(1, dtype=np.int32))
        f, i, ei = o.get_state()
        agents_scene_state(sc, f, i)
        o.set_state(f, i, ei)
        flips += compare_maround_rows(sc, o.observe()[0], worst)
        o.close()
    print("MARound rows:", worst, "corner beams", flips)
    assert worst["rows"] >= 50 and worst["absent"] > 20
    if f64:
        assert worst["state"] < 1e-9 and worst["others"] < 1e-9 and worst["lidar"] < 1e-9 and flips <= 3
    assert worst["state"] < 2e-6 and worst["others"] < 2e-6 and worst["lidar"] < 1e-6 and flips <= 3


@PREC
def test_side_and_lane_line_detectors(L, descs, f64):
    """SideDetector / LaneLineDetector fans spliced into vehicle_state by the reference's own StateObservation
    (state_obs.py:64-71,96-105; distance_detector.py:137-152), cast against the reference-recorded line boxes."""
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "detectors_v0.json")))["cases"]
    SF = _abi.SF
    worst, flips, total = 0.0, 0, 0
    for sc in cases:
        (ks, ds), (km, dm) = sc["side"], sc["lane_line"]
        ram = sc.get("random_agent_model", False)
        o, f, i, ei, d = _scene_engine(descs, sc, f64=f64, side_lasers=ks, side_dist=ds, lane_line_lasers=km, lane_line_dist=dm,
                                       random_agent_model=ram)
        f[SF["DIST_LEFT"], 0, 0], f[SF["DIST_RIGHT"], 0, 0] = sc["left"], sc["right"]
        o.set_state(f, i, ei)
        obs = o.observe()[0, 0]
        n = (ks or 2) + 6 + km + (2 if ram else 0)
        assert obs.shape[0] == n + 10 + 16 + 240 and len(sc["state"]) == n
        dl = np.abs(obs[:n] - np.array(sc["state"]))
        bad = dl > (1e-9 if f64 else 2e-6)
        flips += int(bad.sum())  # a beam through a box corner (the reference helper pads edges by 1e-5)
        total += n
        worst = max(worst, float(dl[~bad].max()))
        o.close()
    print("detector floats", total, "worst", worst, "corner beams", flips)
    assert worst < (1e-9 if f64 else 2e-6) and flips <= 3


@PREC
def test_scene_reward_done(L, descs, scenes, f64):
    """PGDriveEnv.reward_function / done_function (envs/pgdrive_env.py:162-258) over 8 flag combinations per scene."""
    SI = _abi.SI
    bits = {1: _abi.F_ON_YELLOW, 2: _abi.F_CRASH_VEHICLE, 4: _abi.F_CRASH_SIDEWALK}
    worst = 0.0
    for sc in scenes["scenes"]:
        o, f, i, ei, d = _scene_engine(descs, sc, f64=f64)
        for combo, r_ref, d_ref, arrive, oor, crash in sc["rewards"]:
            fl = 0
            for b, m in bits.items():
                if int(combo) & b:
                    fl |= m
            i[SI["VFLAGS"], 0, 0] = fl
            o.set_state(f, i, ei)
            out = (C.c_double * 3)()
            L.orc_reward_done(o.h, 0, 0, out)
            worst = max(worst, abs(out[0] - r_ref))
            assert int(out[1]) == int(d_ref)
            fo = int(out[2])
            assert bool(fo & _abi.F_ARRIVE) == bool(arrive)
            assert bool(fo & _abi.F_OUT_OF_ROAD) == bool(oor)
            assert bool(fo & _abi.F_CRASH_VEHICLE) == bool(crash)
        o.close()
    print("reward worst", worst)
    assert worst < (1e-9 if f64 else 1e-4), worst


@PREC
def test_scene_idm(L, descs, scenes, f64):
    """FrontBackObjects.get_find_front_back_objs (idm_policy.py:82-133) and the full IDMPolicy.act (idm_policy.py:190-353:
    move_to_next_road, lane_change_policy, steering PIDs, IDM law) for every traffic vehicle of every scene."""
    SI, SF = _abi.SI, _abi.SF
    n_checked = 0
    worst = worst_d = 0.0
    for sc in scenes["scenes"]:
        for rec in sc["idm"]:
            o, f, i, ei, d = _scene_engine(descs, sc, f64=f64)
            s = rec["slot"]
            i[SI["TIMER"], 0, s] = rec["timer0"]
            o.set_state(f, i, ei)
            objs = (C.c_int * 6)()
            dist = (C.c_double * 6)()
            L.orc_find_front_back(o.h, 0, s, sc["vehicles"][s]["lane"], 1 if rec["in_ref"] else 0, objs, dist)
            assert list(objs[:3]) == rec["front"] and list(objs[3:]) == rec["back"], (sc["seed"], s)
            ref_d = np.array(rec["fd"] + rec["bd"])
            worst_d = max(worst_d, float(np.abs(np.array(dist[:]) - ref_d).max()))
            out = (C.c_double * 2)()
            L.orc_idm_act(o.h, 0, s, out)
            assert not rec["fallback"]
            worst = max(worst, abs(out[0] - rec["act"][0]), abs(out[1] - rec["act"][1]) / max(1.0, abs(rec["act"][1])))
            f2, i2, _ = o.get_state()
            assert i2[SI["RLANE"], 0, s] == rec["rlane"]
            assert i2[SI["TIMER"], 0, s] == rec["timer1"]
            assert f2[SF["TARGET_SPEED"], 0, s] == rec["target"]
            n_checked += 1
            o.close()
    print("idm vehicles checked:", n_checked, "worst act", worst, "worst neighbour distance", worst_d)
    assert n_checked > 50 and worst < (1e-9 if f64 else 1e-4) and worst_d < (1e-9 if f64 else 1e-4)


def test_checkpoint_update(L, descs, scenes):
    """Navigation._update_target_checkpoints (navigation.py:262-282)."""
    SI = _abi.SI
    by_seed = {}
    for row in scenes["checkpoints"]:
        by_seed.setdefault(row["seed"], []).append(row)
    for seed, rows in by_seed.items():
        sc = dict(seed=seed, vehicles=[dict(x=0, y=0, theta=0, speed_kmh=0, length=4.5, width=1.8, lane=0,
                                            ckpt=rows[0]["ckpt"], idx=[0, 0])],
                  ego=dict(steering=0, act0=[0, 0], last_heading=[1, 0], last_position=[0, 0]))
        o, f, i, ei, d = _scene_engine(descs, sc)
        for row in rows:
            i[SI["CK0"], 0, 0], i[SI["CK1"], 0, 0] = row["idx"]
            o.set_state(f, i, ei)
            L.orc_update_checkpoints(o.h, 0, 0, row["lane"], float(row["lon"]))
            _, i2, _ = o.get_state()
            assert [int(i2[SI["CK0"], 0, 0]), int(i2[SI["CK1"], 0, 0])] == row["out"], row
        o.close()
