"""Host-side mirror of the reference's env surface that does not need a GPU: config merging, spaces, gym ids, sharding."""
import json
import numpy as np
import pytest

from pgdrive_amd import dist as pdist
from pgdrive_amd import env as penv
from pgdrive_amd import _abi, spaces, vec_env


def test_unknown_config_key_raises_like_reference():
    """Config.update(allow_add_new_key=False) raises KeyError on unknown keys (utils/config.py:115-125)."""
    with pytest.raises(KeyError):
        vec_env.merge_config(vec_env.DEFAULT_CONFIG, dict(not_a_key=1))
    with pytest.raises(KeyError):
        vec_env.merge_config(vec_env.DEFAULT_CONFIG, dict(vehicle_config=dict(lidar=dict(beams=3))))
    c = vec_env.merge_config(vec_env.DEFAULT_CONFIG, dict(vehicle_config=dict(lidar=dict(num_lasers=72))))
    assert c["vehicle_config"]["lidar"]["num_lasers"] == 72 and c["vehicle_config"]["lidar"]["distance"] == 50
    assert vec_env.DEFAULT_CONFIG["vehicle_config"]["lidar"]["num_lasers"] == 240  # defaults untouched


def test_default_config_mirrors_reference_values():
    c = vec_env.DEFAULT_CONFIG
    assert c["decision_repeat"] == 5 and c["physics_world_step_size"] == 2e-2  # base_env.py:33,70
    assert c["traffic_density"] == 0.1 and c["traffic_mode"] == "trigger"  # pgdrive_env.py:47-48
    assert (c["success_reward"], c["out_of_road_penalty"], c["crash_vehicle_penalty"], c["driving_reward"],
            c["speed_reward"]) == (10.0, 5.0, 5.0, 1.0, 0.1)  # pgdrive_env.py:91-101
    assert c["vehicle_config"]["lidar"] == dict(num_lasers=240, distance=50, num_others=4, gaussian_noise=0.0,
                                                dropout_prob=0.0)


def test_spaces():
    b = spaces.Box(-1.0, 1.0, (2, ), np.float32)
    assert b.contains(np.array([0.3, -1.0], dtype=np.float32)) and not b.contains(np.array([1.5, 0.0]))
    assert b.sample().shape == (2, ) and b.contains(b.sample())
    o = spaces.Box(-0.0, 1.0, (274, ), np.float32)
    assert o.contains(np.zeros(274, dtype=np.float32)) and not o.contains(np.zeros(273))


def test_gym_id_table():
    """register.py:5-38"""
    assert penv.ENV_IDS["PGDrive-v0"] == dict(start_seed=1000, environment_num=100)
    assert penv.ENV_IDS["PGDrive-test-v0"] == dict(start_seed=0, environment_num=200)
    assert len(penv.ENV_IDS) == 8


def test_shard_ranges_partition_envs():
    for n, w in ((32768, 8), (4096, 1), (10, 4), (7, 8)):
        got = []
        for r in range(w):
            lo, hi = pdist.shard_range(n, r, w)
            got.extend(range(lo, hi))
        assert got == list(range(n))
    ids = np.concatenate([pdist.scenario_ids_for(*pdist.shard_range(1000, r, 4), 100) for r in range(4)])
    assert (ids == np.arange(1000) % 100).all()  # env -> scenario mapping does not depend on the world size


def test_oracle_multithreaded_step_is_identical(descs):
    """orc_step_mt (OpenMP over envs; the all-cores CPU baseline of bench.py) returns exactly what orc_step returns."""
    from oracle import orc
    from tests import util
    mb, sb = util.make_banks(descs, n_maps=4)
    cfg = _abi.make_config(32, num_agents=1, num_traffic=16, num_lasers=30)
    a, b = orc.Oracle(cfg, mb, sb), orc.Oracle(cfg, mb, sb)
    ids = np.arange(32) % 4
    a.reset(ids)
    b.reset(ids)
    rng = np.random.default_rng(3)
    for t in range(40):
        act = util.driving_actions(rng, 32)
        ra, rb = a.step(act), b.step(act, threads=4)
        for x, y in zip(ra, rb):
            assert np.array_equal(x, y)
    a.close()
    b.close()


def test_lidar_noise_and_dropout_statistics(descs):
    """LidarStateObservation._add_noise_to_cloud_points (state_obs.py:172-182): clip(beam + N(0, sigma), 0, 1), then zero with
    probability dropout -- checked on the oracle as a distribution (the reference draws from the global numpy RNG)."""
    from oracle import orc
    from tests import util
    mb, sb = util.make_banks(descs, n_maps=4, num_traffic=0)
    clean = orc.Oracle(_abi.make_config(64, num_agents=1, num_traffic=0, num_lasers=240), mb, sb)
    noisy = orc.Oracle(_abi.make_config(64, num_agents=1, num_traffic=0, num_lasers=240, lidar_gaussian_noise=0.05,
                                        lidar_dropout_prob=0.1, seed=5), mb, sb)
    ids = np.arange(64) % 4
    a = clean.reset(ids)[:, 0, -240:]
    b = noisy.reset(ids)[:, 0, -240:]
    assert (a == 1.0).all()  # no traffic: every beam misses
    dropped = b == 0.0
    assert abs(dropped.mean() - 0.1) < 0.01
    kept = b[~dropped]
    # clip(1 + N(0, 0.05), 0, 1): half of the mass sits at 1, the rest is the lower half-normal
    assert abs((kept == 1.0).mean() - 0.5) < 0.02
    low = 1.0 - kept[kept < 1.0]
    assert abs(low.mean() - 0.05 * np.sqrt(2 / np.pi)) < 0.002 and abs(np.sqrt((low ** 2).mean()) - 0.05) < 0.002
    again = noisy.reset(ids)[:, 0, -240:]
    assert np.array_equal(again, b)  # same (seed, env, agent, beam, step) -> same draw
    act = np.zeros((64, 1, 2), dtype=np.float32)
    c = noisy.step(act)[0][:, 0, -240:]
    assert not np.array_equal(c, b)  # a new step draws afresh
    clean.close()
    noisy.close()


def test_reference_config_keys_are_accepted_or_refused_by_name():
    """A config written for the reference passes: visual / debugging keys are dropped, out-of-scope features are accepted at
    their neutral value and refused by name otherwise; unknown keys still raise KeyError (utils/config.py:115-125)."""
    import pytest
    from pgdrive_amd.vec_env import DEFAULT_CONFIG, merge_config, strip_reference_only_keys
    ref_style = dict(
        start_seed=5, environment_num=3, use_render=False, manual_control=False, window_size=(1200, 900), camera_height=1.8,
        pstats=False, load_map_from_json=True, use_topdown=False, rgb_clip=True, general_penalty=0.0, num_agents=1,
        vehicle_config=dict(show_navi_mark=True, show_lidar=False, rgb_camera=(84, 84), overtake_stat=False, extra_action_dim=0,
                            lidar=dict(num_lasers=120)), traffic_density=0.2)
    c = merge_config(DEFAULT_CONFIG, strip_reference_only_keys(ref_style))
    assert c["start_seed"] == 5 and c["traffic_density"] == 0.2 and c["vehicle_config"]["lidar"]["num_lasers"] == 120
    assert c["vehicle_config"]["lidar"]["distance"] == 50 and "use_render" not in c and "show_lidar" not in c["vehicle_config"]
    for bad in (dict(use_render=True), dict(manual_control=True),
                dict(vehicle_config=dict(overtake_stat=True)), dict(num_agents=4)):
        with pytest.raises(NotImplementedError):
            strip_reference_only_keys(bad)
    with pytest.raises(KeyError):
        merge_config(DEFAULT_CONFIG, strip_reference_only_keys(dict(not_a_key=1)))


def test_map_choice_short_and_long_form():
    """parse_map_config (base_map.py:16-35) as tests/test_functionality/test_config_consistency.py:13-56 exercises it: `map` (int =
    number of blocks, str = block sequence) or map_config["config"] (+ "type"); both forms end up in the config."""
    from pgdrive_amd.vec_env import DEFAULT_CONFIG, merge_config, resolve_map_choice
    for user, m, typ in (({}, 3, "block_num"), ({"map": 11}, 11, "block_num"), ({"map": "OO"}, "OO", "block_sequence"),
                         ({"map_config": {"config": 11}}, 11, "block_num"),
                         ({"map_config": {"config": "OO", "type": "block_sequence"}}, "OO", "block_sequence")):
        c = merge_config(DEFAULT_CONFIG, user)
        assert resolve_map_choice(c) == m and c["map_config"]["config"] == m and c["map_config"]["type"] == typ
    with pytest.raises(ValueError):
        resolve_map_choice(merge_config(DEFAULT_CONFIG, {"map": 5, "map_config": {"config": 7}}))
    with pytest.raises(ValueError):
        resolve_map_choice(merge_config(DEFAULT_CONFIG, {"map_config": {"config": 7, "type": "block_sequence"}}))
    with pytest.raises(ValueError):
        resolve_map_choice(merge_config(DEFAULT_CONFIG, {"map": 2.5}))


def test_bench_prints_what_it_has_when_the_launcher_terminates_it():
    """N > 1: a rank that dies hard under a transport makes the launcher SIGTERM the others.  Rank 0 then still prints the line
    measured so far -- even while its main thread sits in a blocking call (bench.py, Watchdog.on_termination: the C-level handler feeds
    a wake-up descriptor, a helper thread prints and leaves)."""
    import json
    import os
    import signal
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        import bench
        bench.WATCHDOG.rank = 0
        bench.WATCHDOG.partial = {"metric": "x", "value": 1.0, "value_by_transport": {"root": {"value": 1.0, "gather_ok": True}}}
        bench.WATCHDOG.label = "transport peer+graph"
        bench.WATCHDOG.on_termination()
        print("armed", flush=True)
        time.sleep(60)
    """ % root)
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "armed"
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=30)
    line = json.loads(out.strip().splitlines()[-1])
    assert line["value"] == 1.0 and "peer+graph" in line["aborted"] and line["value_by_transport"]["root"]["gather_ok"] is True
    assert p.returncode == 1


def test_nested_config_update_rules():
    """tests/test_functionality/test_nested_config.py:4-49 on the env surface's merge_config (a user config is merged like
    Config.update(..., allow_add_new_key=False), base_env.py:100-107): nested values update in place, an unknown key at any depth is a
    KeyError, a dict item replaced by a non-dict a TypeError, siblings keep their values."""
    import pytest
    from pgdrive_amd.vec_env import merge_config
    base = {"aa": {"bb": {"cc": 100}}, "x": 1}
    c = merge_config(base, {"aa": {"bb": {"cc": 101}}})
    assert c["aa"]["bb"]["cc"] == 101 and c["x"] == 1 and base["aa"]["bb"]["cc"] == 100
    with pytest.raises(TypeError):
        merge_config(base, {"aa": {"bb": 102}})
    with pytest.raises(KeyError):
        merge_config(base, {"aa": {"bbd": 102}})
    with pytest.raises(KeyError):
        merge_config(base, {"aa": {"bb": {"dd": 101}}})
    c = merge_config({"aa": {"bb": {"cc": 100, "dd": 1}}}, {"aa": {"bb": {"dd": 101}}})
    assert c["aa"]["bb"] == {"cc": 100, "dd": 101}


def test_touched_source_makes_the_roofline_say_stale(tmp_path, monkeypatch):
    """the counter figures of the bench line (`roofline.traffic`, `roofline.issue`) come from committed
    rocprofv3 passes; every pass is stamped with the sha of the sources its binary was built from (pgdrive_amd/build.py
    source_sha, compiled into the library as pgd_source_sha).  A pass whose stamp equals the library's is quoted; touch a source and
    the same pass is reported as stale, figures null."""
    import bench
    from pgdrive_amd import build
    src = tmp_path / "kernel.hip"
    src.write_text("// step kernel, version 1\n")
    monkeypatch.setattr(build, "DEPS", [str(src)])
    sha1 = build.source_sha()
    assert len(sha1) == 16 and sha1 == build.source_sha()
    prof_dir = tmp_path / "profiles"
    prof_dir.mkdir()
    wl = dict(envs=4096, traffic=16, lasers=240, actions="uniform", traffic_mode="trigger", workload="c3", agents=1)
    (prof_dir / "r06_head_pmc_traffic.json").write_text(json.dumps(dict(wl, bytes_per_launch=18.0e6, source_sha=sha1)))
    (prof_dir / "r06_head_pmc_insts.json").write_text(json.dumps(dict(
        wl, source_sha=sha1, k_step=dict(insts_per_wave=2000.0, waves_per_launch=4096, valu=1300.0, salu=650.0, lds=50.0, vmem=60.0, smem=40.0))))
    (prof_dir / "r06_issue_rate.json").write_text(json.dumps(dict(
        clock_ghz=2.4, mixes=[dict(name="2 VALU : 1 SALU", valu_per_salu=2.0, ns_per_inst_per_simd={"4": 0.885})])))
    monkeypatch.setattr(bench, "PROFILES_DIR", str(prof_dir))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    args = bench.parse_args([])
    prof = dict(k_step_ms=0.017, k_observe_ms=0.0, count=100)
    r = bench.make_roofline(prof, 64, None, args, 4096, 1, 274, lib_sha=build.source_sha())
    assert r["stale"] is False and r["traffic"] == 18.0e6 and r["issue"] is not None
    assert r["source_sha"] == r["profile_source_sha"] == sha1 and r["frac_moved"] is not None
    # a kernel change without a re-profile
    src.write_text("// step kernel, version 2\n")
    sha2 = build.source_sha()
    assert sha2 != sha1
    r2 = bench.make_roofline(prof, 64, None, args, 4096, 1, 274, lib_sha=sha2)
    assert r2["stale"] is True and r2["traffic"] is None and r2["issue"] is None
    assert r2["source_sha"] == sha2 and r2["profile_source_sha"] == sha1 and sha1 in r2["stale_reason"]
    assert r2["frac"] == r["frac"]  # the contract's figure comes from this run's own HIP events either way
    # a pass from before the stamps existed (rounds 1 - 5) is stale by construction
    (prof_dir / "r06_head_pmc_traffic.json").write_text(json.dumps(dict(wl, bytes_per_launch=18.0e6)))
    r3 = bench.make_roofline(prof, 64, None, args, 4096, 1, 274, lib_sha=sha2)
    assert r3["stale"] is True and r3["traffic"] is None


def test_the_built_library_carries_the_stamp_of_its_sources():
    """pgd_source_sha of the in-tree library equals build.source_sha() of the tree it was built from (build() passes it as a
    define); needs the library, not a GPU."""
    from pgdrive_amd import build, engine
    build.build()
    assert engine.load_library().pgd_source_sha().decode() == build.source_sha()


_GYM_PROBE = """
import json, sys
sys.path.insert(0, %r)
import gym
import pgdrive_amd
from pgdrive_amd import spaces, env, marl_env, vec_env
reg = gym.envs.registration.registry
out = dict(
    gym_found=spaces.GYM is gym,
    single=issubclass(env.PGDriveEnv, gym.Env) and issubclass(env.SafePGDriveEnv, gym.Env) and issubclass(env.TopDownPGDriveEnv, gym.Env),
    marl=all(issubclass(getattr(marl_env, n), gym.Env) for n in ("MultiAgentRoundaboutEnv", "MultiAgentIntersectionEnv",
             "MultiAgentBottleneckEnv", "MultiAgentTollgateEnv", "MultiAgentParkingLotEnv", "MultiAgentPGDrive")),
    spaces=(spaces.Box is gym.spaces.Box) and (spaces.MultiDiscrete is gym.spaces.MultiDiscrete) and (spaces.Dict is gym.spaces.Dict),
    ids={k: [str(getattr(v, "entry_point", None) or v["entry_point"]), (getattr(v, "kwargs", None) or v["kwargs"])["config"]]
         for k, v in (getattr(reg, "env_specs", reg)).items() if k.startswith("PGDrive-")},
    again=pgdrive_amd.spaces.register_gym_ids(),
)
print(json.dumps(out))
"""


def _check_gym_probe(d):
    assert d["gym_found"] and d["single"] and d["marl"] and d["spaces"]
    # pgdrive/register.py:5-41: eight ids, each PGDriveEnv(config=dict(start_seed, environment_num))
    assert d["ids"] == {k: ["pgdrive_amd.env:PGDriveEnv", dict(v)] for k, v in penv.ENV_IDS.items()} and len(d["ids"]) == 8
    assert d["ids"]["PGDrive-v0"][1] == dict(start_seed=1000, environment_num=100)
    assert d["again"] == []  # importing twice / calling again registers nothing twice


def test_gym_identity_with_a_gym_on_the_path(tmp_path):
    """with `gym` importable the envs ARE gym.Env subclasses (envs/base_env.py:93), their spaces are gym's, and
    the eight ids of pgdrive/register.py:5-41 are registered at `import pgdrive_amd`.  This image has no gym: a minimal package of
    that name (tests/util.py write_fake_gym) goes on the PYTHONPATH of a subprocess -- the decision is made at import."""
    import os
    import subprocess
    import sys
    from tests import util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = util.write_fake_gym(tmp_path)
    env = dict(os.environ, PYTHONPATH=fake + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", _GYM_PROBE % root], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    _check_gym_probe(json.loads(out.stdout.strip().splitlines()[-1]))


def test_gym_identity_with_the_real_gym():
    """The same with a real gym, where one is installed (skipped here)."""
    pytest.importorskip("gym")
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _GYM_PROBE % root], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    _check_gym_probe(json.loads(out.stdout.strip().splitlines()[-1]))


def test_without_gym_the_stand_ins_keep_the_surface():
    """No gym on the path (this image): the stand-in spaces, plain classes, nothing registered -- the env surface is unchanged."""
    if spaces.GYM is not None:
        pytest.skip("gym is installed")
    assert penv.PGDriveEnv.__mro__[1] is object and spaces.register_gym_ids() == []
    b = spaces.Box(-1.0, 1.0, (2, ), np.float32)
    assert b.contains(np.zeros(2, np.float32)) and not b.contains(np.full(2, 2.0, np.float32))


def test_run_time_kernel_builds_without_a_gpu(tmp_path, monkeypatch):
    """pgdrive_amd/jit.py: the code object of a step kernel with one configuration compiled in is built by hipcc (a cross-compile: no
    GPU), holds the kernel under the name pgd_set_step_module looks up, and is found in the cache the second time."""
    import time
    from pgdrive_amd import jit
    monkeypatch.setenv("PGD_JIT_DIR", str(tmp_path))
    cfg = _abi.make_config(512, num_traffic=12, num_lasers=72, num_others=2, discrete_action=True)
    geom = dict(N=512, A=1, T=12, V=13, D=_abi.obs_dim(cfg), NV=512 * 13, epw=1, sub=4, pack_obs=0, sstride=13, use_imask=0)
    text = jit.header_text(cfg, geom, False, True)
    assert "F(d.V, 13)" in text and "F(c.num_lasers, 72)" in text and "F(c.discrete_action, 1)" in text and "PGD_JIT_STD true" in text
    assert "F(c.lidar_dist, 0x1.9000000000000p+5f)" in text  # 50.0 as an exact hexadecimal float literal
    path = jit.build_module(cfg, geom, False, True)
    blob = open(path, "rb").read()
    assert b"_Z6k_stepILb1ELb0ELb0ELb1ELi9ELb0EEv6PgdDevPKfPfPhPjS3_7PgdCold" in blob
    t0 = time.time()
    assert jit.build_module(cfg, geom, False, True) == path and time.time() - t0 < 0.5  # (cached)
    cfg2 = _abi.make_config(512, num_traffic=12, num_lasers=96, num_others=2, discrete_action=True)
    assert jit.header_text(cfg2, dict(geom, D=_abi.obs_dim(cfg2)), False, True) != text


def test_bench_rows_are_well_formed():
    """bench.py's ROWS: names unique, every override names an option parse_args knows (a typo would silently time the default
    workload), and a row that replays per-group HIP graphs times a whole number of replays per window."""
    import bench
    base = vars(bench.parse_args([]))
    names = [n for n, _ in bench.ROWS]
    assert len(names) == len(set(names)) and "c5_40x72" in names and "c5_40x72_two_groups" in names and "c3_topdown_u8" in names
    for name, over in bench.ROWS:
        for k in over:
            assert k in base, "row %s overrides %r, which is not an option of bench.py" % (name, k)
        gg = over.get("groups_graph", 0)
        if gg:
            assert over.get("groups", 1) > 1 and over["steps"] % gg == 0, name
        if over.get("topdown_u8"):
            assert over.get("topdown")
    a = bench.parse_args(["--groups", "2", "--groups-graph", "64", "--topdown", "--topdown-u8"])
    assert a.groups == 2 and a.groups_graph == 64 and a.topdown_u8 and a.lasers == 0
