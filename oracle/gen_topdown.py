"""Golden vectors for the ARITHMETIC of the reference's top-down observation (runs only in the build container).

    PYTHONHASHSEED=0 PYTHONDONTWRITEBYTECODE=1 python oracle/gen_topdown.py  ->  tests/golden/topdown_v0.json

`TopDownMultiChannel` (pgdrive/obs/top_down_obs_multi_channel.py:18-280) and `ObservationWindow` / `WorldSurface`
(obs/top_down_obs_impl.py:15-200, 401-445) are imported from where they lie, with pygame replaced by a module that only RECORDS
what it is handed (surface sizes, blit / crop rectangles, the angle and zoom given to rotozoom, the headings and colours given to
the vehicle drawing routine, the vectors and angles given to Vector2.rotate, the pixel filled for a past position).  Nothing of
pygame's rasterisation is re-implemented: `Vector2.rotate` answers only for multiples of 180 degrees (sign flips, the same under
every convention), `smoothscale` / `array3d` pass arrays through.  Pinned this way:
  stack_indices   _get_stack_indices: which entries of the frame / past-position deques are shown
  grey            _transform: the grey value of every colour the observation draws with (both clip modes), and the * 2 of the road channel
  observe         TopDownMultiChannel.observe run on injected per-step channel images: deque lengths, refill after a reset, stack
                  order, clip, transpose -- as the table "channel k at step t shows the traffic frame of step ..."
  geometry        draw_map + ObservationWindow.reset / render on the reference's own road networks: canvas px per metre, receptive
                  fields, the rotation angle (heading -> degrees + 90) and zoom handed to rotozoom, the centre crop: the effective
                  window scale in px per metre for each channel
  scene           draw_scene: the 2-degree heading snap (vehicles and past positions: snapped; window rotation: NOT snapped), the
                  past-position transform (scale resolution / max_distance, axis swap, rotation angle, + resolution / 2, clip) and
                  the pixel it fills, for ego headings of -90 and +90 degrees (rotation by 0 / 180: convention-free)
What stays UNPINNED (pygame absent): polygon fill / line width rules, the resampling of rotozoom and smoothscale (the reference
renders the road channel at twice the resolution and smooth-scales it down: anti-aliased greys), Rect truncation of float
coordinates, the sign convention of Vector2.rotate for other angles.
"""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import ref_export  # noqa: E402  (installs the stubs for panda3d / gym / ...)

REC = []


class Surface:
    def __init__(self, size, flags=0, surf=None):
        self._size = (int(size[0]), int(size[1]))
        self.name = None

    def get_size(self):
        return self._size

    def get_width(self):
        return self._size[0]

    def get_height(self):
        return self._size[1]

    def fill(self, color, rect=None):
        REC.append(("fill", self.name, tuple(color) if not isinstance(color, str) else color, rect))

    def blit(self, src, dest, area=None):
        REC.append(("blit", self.name, getattr(src, "name", None), tuple(dest), None if area is None else tuple(float(a) for a in area)))

    def set_clip(self, r):
        REC.append(("set_clip", self.name, tuple(float(a) for a in r)))

    def set_colorkey(self, c):
        pass


class Vector2:
    def __init__(self, *a):
        if len(a) == 1:
            a = tuple(a[0])
        self.x, self.y = float(a[0]), float(a[1])

    def rotate(self, angle):
        REC.append(("rotate", (self.x, self.y), float(angle)))
        k = (float(angle) / 180.0)
        if abs(k - round(k)) < 1e-9:
            return Vector2(self.x, self.y) if int(round(k)) % 2 == 0 else Vector2(-self.x, -self.y)
        return Vector2(float("nan"), float("nan"))  # the convention of other angles is pygame's own: not re-implemented

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __add__(self, o):
        return Vector2(self.x + o[0], self.y + o[1])

    def __iter__(self):
        return iter((self.x, self.y))


def _install_pygame():
    pg = types.ModuleType("pygame")
    pg.Surface = Surface
    pg.SurfaceType = Surface
    pg.Color = lambda name: name
    pg.init = lambda: None
    pg.display = types.SimpleNamespace(set_caption=lambda *a: None, set_mode=lambda *a: Surface((1, 1)), flip=lambda: None)
    pg.event = types.SimpleNamespace(get=lambda: [], EventType=object)
    pg.math = types.SimpleNamespace(Vector2=Vector2)

    def rotozoom(surf, angle, scale):
        REC.append(("rotozoom", getattr(surf, "name", None), float(angle), float(scale), surf.get_size()))
        out = Surface((surf.get_size()[0] * scale, surf.get_size()[1] * scale))  # (only the centre of it is cropped)
        out.name = "rotated"
        return out

    pg.transform = types.SimpleNamespace(rotozoom=rotozoom, smoothscale=lambda s, res: s, scale2x=lambda s, *a: s)
    pg.surfarray = types.SimpleNamespace(array3d=lambda s: s)
    pg.draw = types.SimpleNamespace(polygon=lambda *a, **k: REC.append(("polygon", )), line=lambda *a, **k: None)
    pg.KEYDOWN = pg.K_ESCAPE = pg.K_l = pg.K_o = pg.K_m = pg.K_k = 0
    sys.modules["pygame"] = pg
    for k in [k for k in sys.modules if k.startswith("pgdrive.obs.top_down")]:
        del sys.modules[k]
    return pg


def main():
    _install_pygame()
    from pgdrive.constants import DEFAULT_AGENT
    from pgdrive.obs import top_down_obs_impl as impl
    from pgdrive.obs.top_down_obs_multi_channel import TopDownMultiChannel
    from pgdrive.utils.math_utils import Vector

    R, DIST, FS, PS, SKIP = 84, 30, 3, 5, 5  # TopDownPGDriveEnv (envs/top_down_env.py:8-42)
    env = types.SimpleNamespace(config=dict(use_render=False))

    def make(clip=True, frame_stack=FS, post_stack=PS, frame_skip=SKIP):
        return TopDownMultiChannel({}, env, clip, frame_stack=frame_stack, post_stack=post_stack, frame_skip=frame_skip,
                                   resolution=(R, R), max_distance=DIST)

    out = dict(config=dict(resolution=R, distance=DIST, frame_stack=FS, post_stack=PS, frame_skip=SKIP))

    # ---- stack indices -------------------------------------------------------------------------------------------------
    o = make()
    out["stack_indices"] = [dict(length=n, frame_skip=k, indices=o._get_stack_indices(n, k)) for k in (1, 3, 5) for n in range(1, 26)]
    out["deque_maxlen"] = dict(traffic=o.stack_traffic_flow.maxlen, past_pos=o.stack_past_pos.maxlen)

    # ---- grey levels -----------------------------------------------------------------------------------------------------
    colors = dict(lane_line=impl.WorldSurface.LANE_LINE_COLOR, navigation=(64, 64, 64), vehicle_blue=impl.VehicleGraphics.BLUE,
                  white=(255, 255, 255), black=(0, 0, 0))
    grey = {}
    for clip in (True, False):
        oc = make(clip)
        for name, c in colors.items():
            img = np.array(c, dtype=np.uint8).reshape(1, 1, 3)
            grey["%s_%s" % (name, "clip" if clip else "u8")] = float(oc._transform(img)[0, 0])
    out["grey"] = grey
    out["colors"] = {k: list(v) for k, v in colors.items()}

    # ---- observe(): frame history, stack order, clip, transpose ------------------------------------------------------------
    # channel images are injected as arrays: value (t + 1) in the red channel of a 2 x 3 image whose [x][y] layout is marked by one
    # brighter pixel at x = 1, y = 2 (to pin the final transpose)
    for clip in (True, ):
        oc = make(clip)
        steps = []

        def frame(t):
            a = np.zeros((2, 3, 3), dtype=np.float64)
            a[..., :] = float(t + 1)
            a[1, 2, :] = float(t + 1) + 100.0
            return a

        state = dict(t=0)
        oc.render = lambda: None
        oc.get_observation_window = lambda: dict(road_network=np.full((2, 3, 3), 35.0), traffic_flow=frame(state["t"]),
                                                 target_vehicle=np.zeros((2, 3, 3)), past_pos=np.full((2, 3, 3), 255.0 if state["t"] % 2 else 0.0))
        reset_at = (0, 17)
        for t in range(40):
            state["t"] = t
            if t in reset_at:
                oc._should_fill_stack = True  # TopDownMultiChannel.reset (top_down_obs_multi_channel.py:68-73)
            img = oc.observe(None)
            assert img.shape == (3, 2, 2 + FS)
            src = [int(round(float(img[0, 0, 2 + k]) * 255.0)) - 1 for k in range(FS)]  # which step's traffic frame channel 2 + k shows
            steps.append(dict(t=t, reset=t in reset_at, traffic_source=src, road=float(img[0, 0, 0]), past=float(img[0, 0, 1]),
                              marked=[float(img[2, 1, 2]), float(img[1, 2, 2]) if img.shape[1] > 2 else None]))
        out["observe"] = dict(steps=steps, note="image [row][col] = channel[x = col][y = row] (np.transpose of surfarray's [x][y])")

    # ---- geometry: draw_map + ObservationWindow on the reference's own road networks -------------------------------------------
    impl.LaneGraphics.display = classmethod(lambda cls, *a, **k: None)
    geo = []
    for seed in (1000, 1003, 1017, 1042, 1077):
        m = ref_export.generate(seed, block_num=3)
        oc = make()
        for name in ("canvas_background", "canvas_navigation", "canvas_road_network", "canvas_runtime", "canvas_ego"):
            getattr(oc, name).name = name
        oc.road_network = m["net"]
        oc.draw_navigation = lambda canvas, color=(128, 128, 128): REC.append(("draw_navigation", tuple(color)))
        del REC[:]
        oc.draw_map()
        nav_colors = [r[1] for r in REC if r[0] == "draw_navigation"]
        cr = oc.canvas_runtime
        bb = m["net"].get_bounding_box()
        heading = 0.7
        pos = cr.pos2pix(bb[0] + 30.0, bb[2] + 25.0)
        del REC[:]
        oc.obs_window.render(canvas_dict=dict(road_network=oc.canvas_road_network, traffic_flow=oc.canvas_runtime,
                                              target_vehicle=oc.canvas_ego), position=pos, heading=heading)
        rz = [r for r in REC if r[0] == "rotozoom"]
        crops = [r for r in REC if r[0] == "blit" and r[2] == "rotated"]
        subs = oc.obs_window.sub_observations
        geo.append(dict(
            seed=seed, bounding_box=[float(v) for v in bb], canvas_px_per_m=float(cr.scaling), origin=[float(cr.origin[0]), float(cr.origin[1])],
            navigation_color=list(nav_colors[0]), pos_pix=[int(pos[0]), int(pos[1])],
            receptive_field=list(subs["traffic_flow"].receptive_field), receptive_field_double=list(subs["traffic_flow"].receptive_field_double),
            rotozoom=[dict(angle=r[2], zoom=r[3], src=list(r[4])) for r in rz], heading=heading,
            crop=[list(r[4]) for r in crops],
            # window px per metre: canvas px/m * zoom (road channel: rendered at 2 R, then smooth-scaled to R: half of it)
            window_px_per_m=dict(traffic=float(cr.scaling) * rz[1][3] if len(rz) > 1 else None,
                                 road=float(cr.scaling) * rz[0][3] / 2.0)))
    out["geometry"] = geo

    # ---- draw_scene: heading snap, past positions ------------------------------------------------------------------------------
    shown = []
    impl.VehicleGraphics.display = classmethod(lambda cls, vehicle, surface, color, heading, **k: shown.append((float(heading), tuple(color))))
    import pgdrive.obs.top_down_obs_multi_channel as mod
    mod.VehicleGraphics.display = impl.VehicleGraphics.display
    scenes = []
    m = ref_export.generate(1000, block_num=3)
    for ego_heading, label in ((-math.pi / 2, "rot0"), (math.pi / 2, "rot180"), (0.02, "snapped"), (0.6, "free")):
        oc = make()
        oc.road_network = m["net"]
        oc.draw_navigation = lambda *a, **k: None
        oc.draw_map()
        oc.canvas_past_pos.name = "past_pos"
        others = [types.SimpleNamespace(heading_theta=h, position=Vector((10.0 + k, 3.0)), WIDTH=1.8, LENGTH=4.5)
                  for k, h in enumerate((0.0, 0.03, -0.034, 0.0352, -0.5, 2.0))]
        ego = types.SimpleNamespace(heading_theta=ego_heading, position=Vector((20.0, 5.0)), WIDTH=1.852, LENGTH=4.51)
        oc.engine = types.SimpleNamespace(agents={DEFAULT_AGENT: ego}, traffic_manager=types.SimpleNamespace(vehicles=[ego] + others))
        rendered = []
        oc.obs_window.render = lambda canvas_dict, position, heading: (rendered.append((tuple(position), float(heading))), {})[1]
        track = [(20.0, 5.0), (20.0, 5.0 - 0.7), (20.0 + 0.4, 5.0 - 1.5), (20.0 + 1.0, 5.0 - 2.6), (20.0 + 1.1, 5.0 - 4.0), (20.0 + 3.0, 5.0 - 9.5),
                 (20.0 + 3.5, 5.0 - 12.0)]
        frames = []
        for k, pt in enumerate(track):
            ego.position = Vector(pt)
            del REC[:]
            del shown[:]
            oc.draw_scene()
            fills = [r for r in REC if r[0] == "fill" and r[1] == "past_pos" and r[3] is not None]
            rots = [r for r in REC if r[0] == "rotate"]
            frames.append(dict(ego=list(pt), deque_len=len(oc.stack_past_pos),
                               rotate=[dict(vec=list(r[1]), angle=r[2]) for r in rots],
                               filled=[[float(r[3][0][0]), float(r[3][0][1])] for r in fills],
                               vehicle_headings=[h for h, c in shown], vehicle_color=list(shown[0][1]),
                               window_heading=rendered[-1][1]))
        scenes.append(dict(label=label, ego_heading=ego_heading, other_headings=[o_.heading_theta for o_ in others], frames=frames,
                           past_pos_scaling=float(oc.scaling)))
    out["scene"] = scenes

    path = os.path.join(ROOT, "tests", "golden", "topdown_v0.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["stack_indices"]), "index cases,", len(geo), "maps,", len(scenes), "scenes")
    print("window px/m (traffic, road):", [(round(g["window_px_per_m"]["traffic"], 4), round(g["window_px_per_m"]["road"], 4)) for g in geo],
          "engine:", R / (2.0 * DIST))
    print("grey:", grey)


if __name__ == "__main__":
    main()
