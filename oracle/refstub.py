"""Import harness for the read-only PGDrive reference (TEST INFRASTRUCTURE — runs only in the build container).

The reference (`/root/reference/pgdrive`) cannot be imported as-is here: panda3d (Bullet), gym, seaborn, pygame,
cv2 ... are not installed.  Its pure-math modules (lanes, road network, BIG map generator, IDM policy, navigation,
observation, reward) import and run once those packages are replaced by permissive stubs.

This module installs the stubs and exposes `load()`.  It is used ONLY by `oracle/gen_golden.py` to produce the
fixtures under `tests/golden/` and the map bank; nothing at test/bench/run time imports it (the reference does not
exist on the GPU box).  No reference source is copied: the modules are imported from where they lie.
"""
import sys
import types

REF_ROOT = "/root/reference"


class _Meta(type):
    """Class whose unknown class-attributes are again stub classes (e.g. `ShowBase.ShowBase`)."""
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = _Meta(name, (_Stub, ), {})
        setattr(cls, name, sub)
        return sub


class _Stub(metaclass=_Meta):
    """Instance that swallows any call / attribute access / arithmetic."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Stub()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Stub()

    def __iter__(self):
        return iter(())

    def __len__(self):
        return 0

    def __bool__(self):
        return False

    def __or__(self, o):
        return self

    __ror__ = __and__ = __rand__ = __add__ = __radd__ = __sub__ = __mul__ = __or__

    def __index__(self):
        return 0


class _BitMask32(int):
    """`BitMask32.bit(i)` must be int-like with `|` and `getWord()` (constants.py:99-122,192-195)."""
    def __new__(cls, v=0):
        return int.__new__(cls, v)

    @classmethod
    def bit(cls, i):
        return cls(1 << i)

    @classmethod
    def allOn(cls):
        return cls(0xFFFFFFFF)

    @classmethod
    def allOff(cls):
        return cls(0)

    def getWord(self):
        return int(self)

    def __or__(self, o):
        return _BitMask32(int(self) | int(o))

    __ror__ = __or__


class Vec3(tuple):
    """Real 3-vector so box positions/half-extents survive (coordinates_shift.py:22-33)."""
    def __new__(cls, *a):
        if len(a) == 1:
            a = tuple(a[0])
        return tuple.__new__(cls, (float(a[0]), float(a[1]), float(a[2]) if len(a) > 2 else 0.0))


class LQuaternionf(tuple):
    def __new__(cls, *a):
        return tuple.__new__(cls, tuple(float(v) for v in a))


class BulletBoxShape:
    def __init__(self, half):
        self.half = tuple(float(v) for v in half)


class _BodyNode(_Stub):
    """Recording stand-in for Bullet body nodes: keeps name + shapes (base_block.py:286-464)."""
    def __init__(self, name="", *a, **k):
        self.__dict__["name"] = name
        self.__dict__["shapes"] = []

    def addShape(self, shape, *a):
        self.shapes.append(shape)

    def getName(self):
        return self.name


class BulletGhostNode(_BodyNode):
    pass


class BulletRigidBodyNode(_BodyNode):
    pass


RECORD = []  # every NodePath that wraps a body node, in creation order


class NodePath(_Stub):
    def __init__(self, node=None, *a, **k):
        d = self.__dict__
        d["_node"] = node
        d["pos"] = None
        d["quat"] = None
        d["scale"] = (1.0, 1.0, 1.0)
        if isinstance(node, _BodyNode):
            RECORD.append(self)

    def attachNewNode(self, node, *a, **k):
        return NodePath(node)

    def node(self):
        return self._node if self._node is not None else _Stub()

    def setPos(self, *a):
        self.__dict__["pos"] = Vec3(*a)

    def setQuat(self, q):
        self.__dict__["quat"] = tuple(q)

    def setScale(self, *a):
        self.__dict__["scale"] = tuple(float(v) for v in (a if len(a) == 3 else a[0]))


_OVERRIDES = {
    "BitMask32": _BitMask32, "Vec3": Vec3, "LQuaternionf": LQuaternionf, "BulletBoxShape": BulletBoxShape,
    "BulletGhostNode": BulletGhostNode, "BulletRigidBodyNode": BulletRigidBodyNode, "NodePath": NodePath,
}


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name in _OVERRIDES:
            return _OVERRIDES[name]
        cls = _Meta(name, (_Stub, ), {})
        setattr(self, name, cls)
        return cls


_STUBBED = [
    "panda3d", "panda3d.core", "panda3d.bullet", "gym", "gym.spaces", "gym.envs", "gym.envs.registration", "gym.utils",
    "seaborn", "pygame", "pygame.gfxdraw", "gltf", "direct", "direct.gui", "direct.gui.OnscreenImage",
    "direct.gui.OnscreenText", "direct.showbase", "direct.showbase.ShowBase", "direct.showbase.OnScreenDebug",
    "direct.controls", "direct.controls.InputState", "direct.filter", "direct.filter.FilterManager", "cv2",
    "simplepbr", "evdev", "PIL", "PIL.Image", "matplotlib", "matplotlib.pyplot",
]

_loaded = False


def load():
    """Install stubs, register an empty `pgdrive` package rooted at the reference, return nothing."""
    global _loaded
    if _loaded:
        return
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    import numpy as np
    for alias, typ in (("float", float), ("int", int), ("bool", bool)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)  # reference uses removed numpy aliases (lidar.py:83, math_utils.py:15)
    for name in _STUBBED:
        if name not in sys.modules:
            m = _StubModule(name)
            m.__path__ = []
            sys.modules[name] = m
    # real gym.Env base so `class BasePGDriveEnv(gym.Env)` is a plain class
    sys.modules["gym"].Env = type("Env", (), {})
    pkg = types.ModuleType("pgdrive")
    pkg.__path__ = [REF_ROOT + "/pgdrive"]  # skips pgdrive/__init__.py (imports gym + envs)
    sys.modules["pgdrive"] = pkg
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _loaded = True


class FakeWorld:
    """Stands in for a Bullet world: BIG only needs attach/remove (base_object.py:14-36)."""
    def attach(self, *a, **k):
        pass

    def remove(self, *a, **k):
        pass

    attachRigidBody = attachGhost = removeRigidBody = removeGhost = attach


class FakePhysicsWorld:
    def __init__(self):
        self.dynamic_world = FakeWorld()
        self.static_world = FakeWorld()
