"""Top-down multi-channel observation (obs/top_down_obs_multi_channel.py as a rasteriser kernel, pgdrive_amd/csrc/pgd_topdown.h):
GPU vs the oracle's brute-force fp64 restatement of the same definition, known answers, frame stacking and reset behaviour.
(pygame, which the reference rasterises with, exists neither here nor on the GPU box: pixel parity with it is unpinned.)"""
import numpy as np
import pytest

from pgdrive_amd import _abi
from tests import util

pytestmark = pytest.mark.gpu
TD_LINE, TD_NAVI, TD_VEH = 2 * 35 / 255, 2 * 64 / 255, (0.299 * 100 + 0.587 * 200 + 0.114 * 255) / 255


def _setup(descs, n_envs, td=None, **kw):
    import torch
    from oracle import orc
    from pgdrive_amd.engine import Engine
    mb, sb = util.make_banks(descs, n_maps=8)
    cfg = _abi.make_config(n_envs, num_agents=1, num_traffic=16, num_lasers=0, auto_reset=kw.get("auto_reset", 1), seed=3)
    eng, ora = Engine(cfg, mb, sb), orc.Oracle(cfg, mb, sb)
    td = td or _abi.make_topdown_config()
    eng.enable_topdown(td)
    ora.enable_topdown(td)
    return torch, eng, ora, mb, sb


def test_topdown_parity_with_the_oracle(descs):
    """84 x 84 x 5 images (TopDownPGDriveEnv defaults) of 48 envs over 70 teacher-forced steps incl. auto-resets: every pixel is one
    of the channel's few exact values, so the comparison is exact except where a pixel centre sits on an edge (fp32 vs fp64 inside
    / outside); such pixels are counted.  Road channel (round 5: one 0.5 m cell of the pre-averaged raster per pixel): a centre
    within an fp32 ulp of a cell border (1.5e-5 m at 200 m from the origin, two axes, 0.5 m cells) reads the neighbouring cell,
    whose counts differ wherever a line or a lane edge is near: 1e-4 of the road channel's pixels, counted on their own."""
    n = 32
    torch, eng, ora, mb, sb = _setup(descs, n)
    ids = np.arange(n) % 8
    ora.reset(ids)
    eng.reset(ids)
    rng = np.random.default_rng(2)
    tot = diff = diff_road = 0
    seen = dict(line=0, navi=0, veh=0, past=0, old_frames=0, resets=0)
    # waiting traffic moved into view: a few vehicles around every ego (random offsets / headings), so that the traffic
    # channels have something to show from the first frame on
    f, i, ei = ora.get_state()
    SF, SI = _abi.SF, _abi.SI
    for k in range(1, 6):
        th = f[SF["THETA"], :, 0]
        fw, lt = rng.uniform(6, 26, n), rng.uniform(-9, 9, n)
        f[SF["X"], :, k] = f[SF["X"], :, 0] + fw * np.cos(th) - lt * np.sin(th)
        f[SF["Y"], :, k] = f[SF["Y"], :, 0] + fw * np.sin(th) + lt * np.cos(th)
        f[SF["THETA"], :, k] = th + rng.uniform(-3.0, 3.0, n) * (k % 2)  # every other one keeps the ego's heading (snapped when ~0)
        f[SF["HX"], :, k] = f[SF["HY"], :, k] = 0.0
        i[SI["STATUS"], :, k] = _abi.ST_PENDING
    f32 = util.round_state_f32(f)
    ora.set_state(f32, i, ei)
    eng.set_state(f32, i, ei)

    def compare():
        nonlocal tot, diff, diff_road
        g = eng.observe_topdown().cpu().numpy().astype(np.float64)
        o = ora.observe_topdown()
        assert g.shape == o.shape == (n, 84, 84, 5)
        # road channel: the mean over the 2 x 2 texels of the pre-averaged raster's cell, each line / route lane / nothing
        levels = {round((k * TD_LINE + j * TD_NAVI) / 4, 6) for k in range(5) for j in range(5 - k)}
        assert set(np.unique(np.round(g[..., 0], 6))) <= levels
        assert set(np.unique(g[..., 1])) <= {0.0, 1.0}
        assert set(np.unique(np.round(g[..., 2:], 6))) <= {0.0, round(TD_VEH, 6)}
        d = np.abs(g - o) > 1e-6
        tot += d.size
        diff += int(d[..., 1:].sum())
        diff_road += int(d[..., 0].sum())
        seen["line"] += int((np.abs(o[..., 0] - TD_LINE) < 1e-9).sum()); seen["navi"] += int((np.abs(o[..., 0] - TD_NAVI) < 1e-9).sum())
        seen["edge"] = seen.get("edge", 0) + int(((o[..., 0] > 0) & (np.abs(o[..., 0] - TD_LINE) > 1e-9) & (np.abs(o[..., 0] - TD_NAVI) > 1e-9)).sum())
        seen["veh"] += int((o[..., 2] > 0).sum()); seen["past"] += int(o[..., 1].sum())
        seen["old_frames"] += int((np.abs(o[..., 2] - o[..., 4]) > 0).any(axis=(1, 2)).sum())
        return g

    compare()
    for t in range(56):
        act = util.driving_actions(rng, n)
        act[:, 0, 1] = np.abs(act[:, 0, 1]) * 0.7 + 0.3
        if t % 4 == 0:
            act[::3, 0, 0] = 1.0
        o_out = ora.step(act)
        eng.step(torch.from_numpy(act).to(eng.device))
        eng.sync()
        seen["resets"] += int(o_out[2].sum())
        f, i, ei = ora.get_state()
        f32 = util.round_state_f32(f)
        ora.set_state(f32, i, ei)
        eng.set_state(f32, i, ei)
        compare()
    print("top-down parity: pixels", tot, "edge pixels that differ: road channel", diff_road, "other channels", diff, seen)
    assert diff <= 2e-5 * tot and diff_road <= 2.5e-4 * (tot / 5)
    assert seen["line"] > 20000 and seen["navi"] > 400000 and seen["edge"] > 100000 and seen["veh"] > 5000 and seen["past"] > 2000
    assert seen["old_frames"] > 100 and seen["resets"] > 5
    eng.close()


def test_topdown_known_answers(descs):
    """Right after a reset: the ego stands on its route (centre pixel = navigation grey, ego not drawn in the traffic channels),
    the three traffic frames are identical (the history is the first frame repeated), past_pos has exactly the centre pixel;
    a vehicle teleported 10 m ahead appears as a blob of its box size centred 14 px above the centre; after six steps the
    newest frame differs from the oldest for a moving ego."""
    n = 8
    torch, eng, ora, mb, sb = _setup(descs, n, auto_reset=0)
    ids = np.arange(n) % 8
    eng.reset(ids)
    ora.reset(ids)
    g = eng.observe_topdown().cpu().numpy()
    c = 42
    assert (g[:, c, c, 0] >= 0.25 * TD_LINE - 1e-6).all()  # on its route: route grey, or a mix with a lane line under the car
    assert (g[:, c - 1:c + 1, c - 1:c + 1, 2:] == 0).all()
    assert (g[..., 2] == g[..., 3]).all() and (g[..., 3] == g[..., 4]).all()
    assert (g[..., 1].sum(axis=(1, 2)) == 1).all() and (g[:, c, c, 1] == 1).all()
    # a vehicle 10 m straight ahead of the ego
    f, i, ei = ora.get_state()
    SF, SI = _abi.SF, _abi.SI
    k = 1
    th = f[SF["THETA"], :, 0]
    f[SF["X"], :, k] = f[SF["X"], :, 0] + 10 * np.cos(th)
    f[SF["Y"], :, k] = f[SF["Y"], :, 0] + 10 * np.sin(th)
    f[SF["THETA"], :, k] = th
    f[SF["HX"], :, k] = f[SF["HY"], :, k] = 0.0
    i[SI["STATUS"], :, k] = _abi.ST_PENDING
    eng.set_state(f, i, ei)
    eng.reset  # (no reset: the history keeps the old frames, the new frame shows the vehicle)
    g = eng.observe_topdown().cpu().numpy()
    blob = g[..., 2] > 0
    rows = np.array([np.nonzero(blob[e].any(axis=1))[0].mean() for e in range(n)])
    cols = np.array([np.nonzero(blob[e].any(axis=0))[0].mean() for e in range(n)])
    assert np.abs(rows - (c - 0.5 - 14)).max() < 1.2 and np.abs(cols - (c - 0.5)).max() < 1.2  # 10 m * 1.4 px/m ahead = up
    sp = sb.spawns[ids * sb.V + k]
    area = blob.sum(axis=(1, 2))
    assert np.all(np.abs(area - sp["length"] * sp["width"] * 1.4 * 1.4) < 6)
    assert (g[..., 3][blob] == 0).all()  # the older frames do not have it
    # drive: after 6 steps the frame of 5 steps ago is a different picture from the newest one
    act = np.zeros((n, 1, 2), np.float32)
    act[..., 1] = 1.0
    for t in range(12):
        eng.step(torch.from_numpy(act).to(eng.device))
        g = eng.observe_topdown().cpu().numpy()
    assert (g[..., 2] != g[..., 3]).any(axis=(1, 2)).all()
    assert (g[..., 1].sum(axis=(1, 2)) >= 2).all()  # several past positions marked by now
    eng.close()


def test_topdown_env_class():
    """TopDownPGDriveEnv mirrors envs/top_down_env.py:28-42: image observation space [84, 84, 5], lidar removed."""
    from pgdrive_amd.env import TopDownPGDriveEnv
    env = TopDownPGDriveEnv(dict(start_seed=1000, environment_num=4))
    try:
        assert env.observation_space.shape == (84, 84, 5)
        o = env.reset()
        assert o.shape == (84, 84, 5) and o.dtype == np.float32 and o.min() >= 0.0 and o.max() <= 1.0
        tot = 0
        for t in range(30):
            o, r, d, info = env.step([0.0, 1.0])
            assert o.shape == (84, 84, 5) and env.observation_space.contains(o)
            tot += r
            if d:
                o = env.reset()
        assert tot > 1.0
    finally:
        env.close()


def test_top_down_rendering_like_upstream():
    """tests/test_env/test_top_down_env.py:6-26: the three env classes in the four configurations upstream runs (map "C", traffic
    density 1.0, frame_stack / post_stack overrides): every observation of 5 episodes x 20 steps shows something."""
    from pgdrive_amd.env import TopDownSingleFramePGDriveEnv, TopDownPGDriveEnv, TopDownPGDriveEnvV2
    for cls, cfg, shape in ((TopDownSingleFramePGDriveEnv, dict(environment_num=5, map="C", traffic_density=1.0), (200, 200, 3)),
                            (TopDownPGDriveEnv, dict(environment_num=5, map="C", traffic_density=1.0), (84, 84, 5)),
                            (TopDownPGDriveEnv, dict(environment_num=5, map="C", frame_stack=1, post_stack=2), (84, 84, 3)),
                            (TopDownPGDriveEnvV2, dict(environment_num=5, map="C", frame_stack=1, post_stack=2), (84, 84, 3))):
        env = cls(cfg)
        try:
            assert env.observation_space.shape == shape
            for _ in range(5):
                o = env.reset()
                assert o.shape == shape and np.mean(o) > 0.0
                for a in ([0, 1], [-0.05, 1]):
                    for _ in range(10):
                        o, *_ = env.step(a)
                        assert np.mean(o) > 0.0
        finally:
            env.close()


def test_topdown_rgb_single_frame(descs):
    """TopDownObservation (obs/top_down_obs.py:22-240; TopDownSingleFramePGDriveEnv, top_down_env.py:8-26): one RGB frame
    [200, 200, 3] / 255 -- lane lines (35, 35, 35), the ego GREEN (50, 200, 0) at the centre heading up, the other vehicles BLUE
    (100, 200, 255) over it, +-30 m, no route colouring, no history.  GPU vs the oracle's brute-force restatement over teacher-forced
    steps with resets (edge pixels counted), known answers, and the env classes."""
    n = 24
    td = _abi.make_topdown_config(resolution=200, distance=30.0, mode=1)
    torch, eng, ora, mb, sb = _setup(descs, n, td=td)
    assert eng.img.shape == (n, 200, 200, 3)
    ids = np.arange(n) % 8
    ora.reset(ids); eng.reset(ids)
    rng = np.random.default_rng(5)
    f, i, ei = ora.get_state()
    SF, SI = _abi.SF, _abi.SI
    for k in range(1, 6):  # some waiting traffic moved into view, one of them overlapping the ego's box
        th = f[SF["THETA"], :, 0]
        fw, lt = (rng.uniform(6, 26, n), rng.uniform(-9, 9, n)) if k > 1 else (np.full(n, 3.0), np.full(n, 1.0))
        f[SF["X"], :, k] = f[SF["X"], :, 0] + fw * np.cos(th) - lt * np.sin(th)
        f[SF["Y"], :, k] = f[SF["Y"], :, 0] + fw * np.sin(th) + lt * np.cos(th)
        f[SF["THETA"], :, k] = th + rng.uniform(-3.0, 3.0, n) * (k % 2)
        f[SF["HX"], :, k] = f[SF["HY"], :, k] = 0.0
        i[SI["STATUS"], :, k] = _abi.ST_PENDING
    f32 = util.round_state_f32(f)
    ora.set_state(f32, i, ei); eng.set_state(f32, i, ei)
    GREEN, BLUE, LINE = (50 / 255, 200 / 255, 0.0), (100 / 255, 200 / 255, 1.0), (35 / 255,) * 3
    tot = diff = 0
    seen = dict(green=0, blue=0, line=0, resets=0)

    def compare():
        nonlocal tot, diff
        g = eng.observe_topdown().cpu().numpy().astype(np.float64)
        o = ora.observe_topdown()
        px = np.round(g.reshape(-1, 3), 5)
        allowed = {tuple(np.round(c, 5)) for c in (GREEN, BLUE, LINE, (0.0, 0.0, 0.0))}
        assert {tuple(c) for c in np.unique(px, axis=0)} <= allowed
        d = (np.abs(g - o) > 1e-6).any(axis=-1)
        tot += d.size; diff += int(d.sum())
        seen["green"] += int((np.abs(o - GREEN) < 1e-9).all(-1).sum()); seen["blue"] += int((np.abs(o - BLUE) < 1e-9).all(-1).sum())
        seen["line"] += int((np.abs(o - LINE) < 1e-9).all(-1).sum())
        return g

    g = compare()
    # the ego: a green box of its size at the centre, long axis up (3.33 px / m), partly covered by the blue vehicle 3 m ahead
    c = 100
    assert (np.abs(g[:, c + 4:c + 6, c - 2:c + 2] - GREEN) < 1e-6).all()  # behind the centre: ego only
    area_g = (np.abs(g - GREEN) < 1e-6).all(-1).sum(axis=(1, 2))
    sp0 = sb.spawns[ids * sb.V]
    full = sp0["length"] * sp0["width"] * (200 / 60.0) ** 2
    assert np.all(area_g < full + 8) and np.all(area_g > 0.3 * full)  # never more than its own box, some of it hidden by the other
    for t in range(40):
        act = util.driving_actions(rng, n)
        act[:, 0, 1] = np.abs(act[:, 0, 1]) * 0.7 + 0.3
        if t % 4 == 0:
            act[::3, 0, 0] = 1.0
        o_out = ora.step(act)
        eng.step(torch.from_numpy(act).to(eng.device)); eng.sync()
        seen["resets"] += int(o_out[2].sum())
        f, i, ei = ora.get_state()
        f32 = util.round_state_f32(f)
        ora.set_state(f32, i, ei); eng.set_state(f32, i, ei)
        compare()
    print("top-down RGB frame: pixels", tot, "edge pixels that differ", diff, seen)
    assert diff <= 2e-5 * tot and seen["green"] > 20000 and seen["blue"] > 5000 and seen["line"] > 100000 and seen["resets"] > 3
    eng.close()
    from pgdrive_amd.env import TopDownSingleFramePGDriveEnv, TopDownPGDriveEnvV2
    env = TopDownSingleFramePGDriveEnv(dict(start_seed=1000, environment_num=2))
    try:
        assert env.observation_space.shape == (200, 200, 3)
        o = env.reset()
        assert o.shape == (200, 200, 3) and o.dtype == np.float32 and (np.abs(o[100, 100] - GREEN) < 1e-6).all()
        for t in range(5):
            o, r, d, info = env.step([0.0, 1.0])
            assert env.observation_space.contains(o)
    finally:
        env.close()
    env = TopDownPGDriveEnvV2(dict(start_seed=1000, environment_num=2))
    try:
        assert env.observation_space.shape == (84, 84, 5) and env.reset().shape == (84, 84, 5)
    finally:
        env.close()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_topdown_uint8_images(descs, mode):
    """rgb_clip=False (pgdrive_env.py:133-141; top_down_obs_multi_channel.py:208-211, 277-280): the image as the uint8 pygame values
    instead of float32 / 255 -- pgd_observe_topdown_u8.  Two engines through the same 60 steps with auto-resets, one with float images,
    one with bytes: every byte is the float value x 255 truncated (numpy's astype(uint8)), every pixel of every image; the values
    are the few the definition allows; the image of a reset env is refilled like the float one."""
    import torch
    from pgdrive_amd.engine import Engine
    n = 40
    mb, sb = util.make_banks(descs, n_maps=8)
    cfg = _abi.make_config(n, num_agents=1, num_traffic=16, num_lasers=0, auto_reset=1, seed=3)
    # mode 2: an odd resolution (85 x 85 x 5 bytes per image: no image starts on a 16-byte boundary, the byte-by-byte stream-out);
    # mode 3: 96 x 96 x 4 at 25 m (two stacked frames; an image of two bands of rows: the band loop of the gather and of the stream-out)
    td = {0: _abi.make_topdown_config(), 1: _abi.make_topdown_config(resolution=200, distance=30.0, mode=1),
          2: _abi.make_topdown_config(resolution=85), 3: _abi.make_topdown_config(96, 25.0, 2, 3, 4)}[mode]
    a, b = Engine(cfg, mb, sb), Engine(cfg, mb, sb)
    a.enable_topdown(td)
    b.enable_topdown(td, uint8=True)
    if mode >= 2:
        assert b.img.shape == {2: (n, 85, 85, 5), 3: (n, 96, 96, 4)}[mode]
    assert b.img.dtype == torch.uint8 and b.img.shape == a.img.shape
    ids = np.arange(n) % 8
    a.reset(ids); b.reset(ids)
    rng = np.random.default_rng(2)
    f, i, ei = a.get_state()
    SF, SI = _abi.SF, _abi.SI
    for k in range(1, 5):  # some waiting traffic moved into the window
        th = f[SF["THETA"], :, 0]
        fw, lt = rng.uniform(6, 26, n), rng.uniform(-9, 9, n)
        f[SF["X"], :, k] = f[SF["X"], :, 0] + fw * np.cos(th) - lt * np.sin(th)
        f[SF["Y"], :, k] = f[SF["Y"], :, 0] + fw * np.sin(th) + lt * np.cos(th)
        f[SF["THETA"], :, k] = th + rng.uniform(-3.0, 3.0, n) * (k % 2)
        f[SF["HX"], :, k] = f[SF["HY"], :, k] = 0.0
        i[SI["STATUS"], :, k] = _abi.ST_PENDING
    f32 = util.round_state_f32(f)
    a.set_state(f32, i, ei); b.set_state(f32, i, ei)
    n_done = 0
    seen = set()
    for t in range(60):
        if t:
            act = util.driving_actions(rng, n)
            act[::3, 0, 1] = 1.0
            at = torch.from_numpy(act).to(a.device)
            _, _, dn, _ = a.step(at)
            b.step(at)
            n_done += int(dn.sum())
        f = a.observe_topdown().cpu().numpy()
        u = b.observe_topdown().cpu().numpy()
        assert u.dtype == np.uint8
        want = np.floor(f.astype(np.float64) * 255.0 + 1e-3).astype(np.uint8)
        assert np.array_equal(u, want), "step %d: %d bytes differ" % (t, int((u != want).sum()))
        seen |= set(np.unique(u).tolist())
    assert n_done > 5
    if mode == 1:
        assert seen <= {0, 35, 50, 100, 200, 255} and {35, 50, 200} <= seen
    else:  # road channel: (lines x 35 + route texels x 64) / 2 of a 2 x 2 cell, truncated; past positions 255; vehicle boxes 176
        road = {(nl * 35 + nn * 64) >> 1 for nl in range(5) for nn in range(5) if nl + nn <= 4}
        assert seen <= road | {255, 176} and {255, 176, 128} <= seen and len(seen & road) >= 5
    # a caller's own byte buffer, and the float image into a float buffer of the caller's, from the same handle
    own = torch.zeros_like(b.img)
    b.step(at)
    a.step(at)
    assert torch.equal(b.observe_topdown(out=own), own) and own.any()
    f = a.observe_topdown().cpu().numpy()
    assert np.array_equal(own.cpu().numpy(), np.floor(f.astype(np.float64) * 255.0 + 1e-3).astype(np.uint8))
    a.close(); b.close()


def test_topdown_env_uint8():
    """TopDownPGDriveEnv(dict(rgb_clip=False)): Box(0, 255, uint8) observations (top_down_obs_multi_channel.py:277-280)."""
    from pgdrive_amd.env import TopDownPGDriveEnv
    env = TopDownPGDriveEnv(dict(start_seed=1000, environment_num=4, rgb_clip=False))
    try:
        assert env.observation_space.shape == (84, 84, 5) and env.observation_space.dtype == np.uint8
        o = env.reset()
        assert o.shape == (84, 84, 5) and o.dtype == np.uint8 and o.max() == 255
        for t in range(12):
            o, r, d, info = env.step([0.0, 1.0])
            assert o.dtype == np.uint8 and env.observation_space.contains(o) and o[..., 0].max() > 0 and o[..., 1].max() == 255
    finally:
        env.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_topdown_tile_occupancy_changes_nothing(descs, monkeypatch, mode):
    """k_topdown gathers an 8 x 8 pixel tile of the window only if the bounding box of its pixel centres touches a raster line in which
    something is drawn (TopDown::occ, one byte per 64-byte line, built with the rasters); PGD_TD_NO_OCC=1 gathers every tile.  Same
    images, every pixel, over 50 steps with resets on all eight maps -- float and byte images, both modes -- and most of what the
    occupancy skips is there to be skipped (the mean image is mostly empty)."""
    import torch
    from pgdrive_amd.engine import Engine
    n = 48
    mb, sb = util.make_banks(descs, n_maps=8)
    cfg = _abi.make_config(n, num_agents=1, num_traffic=16, num_lasers=0, auto_reset=1, seed=5)
    td = _abi.make_topdown_config(resolution=200, distance=30.0, mode=1) if mode else _abi.make_topdown_config()
    monkeypatch.delenv("PGD_TD_NO_OCC", raising=False)
    a, a8 = Engine(cfg, mb, sb), Engine(cfg, mb, sb)
    a.enable_topdown(td); a8.enable_topdown(td, uint8=True)
    ids = np.arange(n) % 8
    for e in (a, a8):
        e.reset(ids)
        e.observe_topdown()  # (the rasters and their occupancy bytes are built by the first image)
    monkeypatch.setenv("PGD_TD_NO_OCC", "1")
    b, b8 = Engine(cfg, mb, sb), Engine(cfg, mb, sb)
    b.enable_topdown(td); b8.enable_topdown(td, uint8=True)
    for e in (b, b8):
        e.reset(ids)
        e.observe_topdown()
    monkeypatch.delenv("PGD_TD_NO_OCC", raising=False)
    assert torch.equal(a.img, b.img) and torch.equal(a8.img, b8.img)
    rng = np.random.default_rng(8)
    n_done = 0
    filled = 0.0
    for t in range(50):
        act = util.driving_actions(rng, n)
        act[::2, 0, 1] = 1.0
        at = torch.from_numpy(act).to(a.device)
        for e in (a, a8, b, b8):
            _, _, dn, _ = e.step(at)
        n_done += int(dn.sum())
        ia, ib, ia8, ib8 = a.observe_topdown(), b.observe_topdown(), a8.observe_topdown(), b8.observe_topdown()
        assert torch.equal(ia, ib), "step %d: %d float pixels differ" % (t, int((ia != ib).sum()))
        assert torch.equal(ia8, ib8), "step %d: %d bytes differ" % (t, int((ia8 != ib8).sum()))
        filled += float((ia[..., 0] > 0).float().mean())
    assert n_done > 5 and 0.02 < filled / 50 < 0.6
    for e in (a, a8, b, b8):
        e.close()
