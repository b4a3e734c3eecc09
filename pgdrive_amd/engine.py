"""ctypes binding of libpgdrive_hip.so (C ABI in include/pgdrive_hip.h) + torch buffers for device memory.

PyTorch is plumbing here (device allocation, streams, torch.distributed); all simulation runs in the HIP library.
There is NO CPU fallback: if the HIP library is missing or no GPU is visible, constructing an Engine raises.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .build import LIB, build

_LIBH = None

_SIGS = {
    "pgd_obs_dim": (C.c_int, [C.POINTER(_abi.PgdConfig)]),
    "pgd_create": (C.c_int, [C.POINTER(_abi.PgdConfig), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pgd_upload_maps": (C.c_int, [C.c_void_p] + [C.c_void_p, C.c_int] * 6),
    "pgd_upload_scenarios": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pgd_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pgd_step": (C.c_int, [C.c_void_p] * 6),
    "pgd_step_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgd_step_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgd_state_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pgd_get_state": (C.c_int, [C.c_void_p] * 4),
    "pgd_set_state": (C.c_int, [C.c_void_p] * 4),
    "pgd_observe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgd_describe_step": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "pgd_forget_rows": (C.c_int, [C.c_void_p]),
    "pgd_mlp_prepared_bytes": (C.c_size_t, [C.c_int]),
    "pgd_mlp_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "pgd_mlp_policy_prepared": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pgd_step_geometry": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pgd_set_step_module": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "pgd_mlp_policy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "pgd_step_lane_keep": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32] + [C.c_void_p] * 4),
    "pgd_lane_keep_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32]),
    "pgd_topdown_channels": (C.c_int, [C.POINTER(_abi.TopDownConfig)]),
    "pgd_topdown_enable": (C.c_int, [C.c_void_p, C.POINTER(_abi.TopDownConfig)]),
    "pgd_observe_topdown": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgd_observe_topdown_u8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgd_set_groups": (C.c_int, [C.c_void_p, C.c_int]),
    "pgd_step_group": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    "pgd_group_stream": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "pgd_group_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "pgd_last_step_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pgd_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "pgd_profile_begin_strided": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pgd_enable_step_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pgd_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "pgd_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgd_sync": (C.c_int, [C.c_void_p]),
    "pgd_destroy": (C.c_int, [C.c_void_p]),
    "pgd_version": (C.c_char_p, []),
    "pgd_source_sha": (C.c_char_p, []),
    "pgd_gather_create": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_void_p)]),
    "pgd_gather_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "pgd_gather_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgd_gather_connect": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "pgd_gather_push": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pgd_gather_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pgd_gather_release": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pgd_gather_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pgd_gather_mem_kind": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "pgd_gather_destroy": (C.c_int, [C.c_void_p]),
}
EXPORTS = tuple(_SIGS.keys())


def load_library(path=None):
    """dlopen the HIP engine; fails loudly when it has not been built (no fallback path exists)."""
    global _LIBH
    if _LIBH is not None and path is None:
        return _LIBH
    # torch bundles its own HIP runtime: import it first so this library binds to the same libamdhip64 (two runtimes
    # in one process do not see each other's devices / streams)
    import torch  # noqa: F401
    p = path or os.environ.get("PGD_LIB") or LIB  # PGD_LIB: A/B runs of experimental builds (tools/ab.sh)
    if not os.path.exists(p):
        raise RuntimeError(
            "pgdrive_amd: %s is missing — build it with `python -m pgdrive_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the step engine." % p
        )
    L = C.CDLL(p)
    for name, (res, args) in _SIGS.items():
        if path is None and os.environ.get("PGD_LIB") and not hasattr(L, name):
            continue  # A/B runs of an OLDER build (tools/ab.sh): entry points it does not have yet are simply absent
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _LIBH = L
    return L


def _np_p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class PgdError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise PgdError("%s failed with status %d" % (what, rc))


class Engine:
    """One simulation handle on one GPU: N envs x (A agents + T traffic slots)."""
    def __init__(self, cfg, bank, scen, device=0, stream=None, lib=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("pgdrive_amd.Engine needs a GPU (MI355X); no CPU fallback exists")
        self.torch = torch
        self.L = lib if lib is not None else load_library()  # `lib`: another build of the same ABI (tests: IEEE-math A/B)
        self.cfg = cfg
        self.N, self.A, self.T = cfg.num_envs, cfg.num_agents, cfg.num_traffic
        self.V = self.A + self.T
        self.D = _abi.obs_dim(cfg)
        assert scen.V == self.V, "scenario bank built for V=%d, engine has V=%d" % (scen.V, self.V)
        self.device = torch.device("cuda", device)
        self.bank, self.scen = bank, scen
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        # the engine enqueues on its own torch stream; step() orders it after / before the caller's current stream
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        sptr = C.c_void_p(self.stream.cuda_stream)
        _chk(self.L.pgd_create(C.byref(cfg), device, sptr, C.byref(h)), "pgd_create")
        self.h = h
        _chk(
            self.L.pgd_upload_maps(
                self.h, _np_p(bank.maps), len(bank.maps), _np_p(bank.lanes), len(bank.lanes), _np_p(bank.roads),
                len(bank.roads), _np_p(bank.boxes), len(bank.boxes), _np_p(bank.cell_start), len(bank.cell_start),
                _np_p(bank.cell_items), len(bank.cell_items)
            ), "pgd_upload_maps"
        )
        _chk(self.L.pgd_upload_scenarios(self.h, _np_p(scen.scenarios), len(scen.scenarios), _np_p(scen.spawns)),
             "pgd_upload_scenarios")
        dev = self.device
        self.obs = torch.zeros((self.N, self.A, self.D), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((self.N, self.A), dtype=torch.float32, device=dev)
        self.done = torch.zeros((self.N, self.A), dtype=torch.uint8, device=dev)
        self.flags = torch.zeros((self.N, self.A), dtype=torch.int32, device=dev)
        self._bound_stream = self.stream.cuda_stream  # the stream pgd_create was given
        self._own_ptrs = tuple(C.c_void_p(t.data_ptr()) for t in (self.obs, self.reward, self.done, self.flags))
        self._seen_out = []  # caller-supplied observation tensors of the last steps, kept alive (see step)
        torch.cuda.synchronize(dev)

    # -- reference surface ------------------------------------------------------------------------------------------
    def reset(self, scen_ids, env_ids=None):
        scen_ids = np.ascontiguousarray(scen_ids, dtype=np.int32)
        env_ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        self.torch.cuda.synchronize(self.device)
        _chk(self.L.pgd_reset(self.h, _np_p(env_ids), _np_p(scen_ids), len(scen_ids), C.c_void_p(self.obs.data_ptr())),
             "pgd_reset")
        self.sync()
        return self.obs

    def make_outputs(self):
        """A second set of output buffers (double buffering of the per-step gather, pgdrive_amd/dist.py)."""
        t, dev = self.torch, self.device
        return (t.zeros((self.N, self.A, self.D), dtype=t.float32, device=dev),
                t.zeros((self.N, self.A), dtype=t.float32, device=dev),
                t.zeros((self.N, self.A), dtype=t.uint8, device=dev),
                t.zeros((self.N, self.A), dtype=t.int32, device=dev))

    def step(self, actions, want_obs=True, out=None):
        """actions: float32 cuda tensor [N, A, 2] (contiguous). Asynchronous on the engine stream.
        `out` = (obs, reward, done, flags) tensors to write instead of the engine's own buffers.
        The returned obs IS the engine's buffer (or `out[0]`), not a copy.  Multi-agent engines zero the row of a seat that is not
        due ONCE and remember it per buffer (include/pgdrive_hip.h, pgd_step): a caller that edits the returned rows in place
        (normalisation, clamp_, noise) would see its edits persist in the rows of empty seats -- clone first, or create the engine
        with PGD_NO_ROWZ=1 in the environment (the zero rows are then rewritten by every call).  `out` tensors need not be long-lived
        or zero-initialised: a tensor the engine has not seen alive makes it forget its marks (pgd_forget_rows) before the step."""
        assert actions.is_cuda and actions.dtype == self.torch.float32 and actions.is_contiguous()
        assert actions.numel() == self.N * self.A * 2
        forget_first = False
        if out is None:  # the engine's own output buffers: their addresses never change
            obs, reward, done, flags = self.obs, self.reward, self.done, self.flags
            p_obs, p_rew, p_done, p_flags = self._own_ptrs
        else:
            obs, reward, done, flags = out
            p_obs, p_rew, p_done, p_flags = (C.c_void_p(obs.data_ptr()), C.c_void_p(reward.data_ptr()),
                                             C.c_void_p(done.data_ptr()), C.c_void_p(flags.data_ptr()))
            # The zero-row marks of a multi-agent engine name a buffer by its address: a NEW tensor may sit where a freed one sat
            # (torch's caching allocator; `out=torch.empty(...)` per step) and would inherit "known zero" marks for rows that hold
            # anything.  Tensors seen before are kept alive here (their memory cannot be handed out again); any other tensor makes
            # the engine forget its marks first.
            if want_obs and self.A > 1 and not any(t is obs for t in self._seen_out):
                forget_first = True
                self._seen_out.append(obs)
                del self._seen_out[:-4]
        # the engine follows the caller's current stream (like a torch op): no events, no cross-stream waits per step;
        # pgd_set_stream orders the hand-over when the stream changes
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        if forget_first:
            _chk(self.L.pgd_forget_rows(self.h), "pgd_forget_rows")
        _chk(
            self.L.pgd_step(self.h, C.c_void_p(actions.data_ptr()), p_obs if want_obs else None, p_rew, p_done, p_flags),
            "pgd_step"
        )
        return obs, reward, done, flags

    def step_packed(self, actions, rows):
        """Like step(), but the env's results go into `rows` [N, >= A*(D+2)] fp32 as [A*D obs | A reward | A done]: the
        row a per-step gather sends (pgdrive_amd/dist.py); `rows` may be a slice of the gather's receive buffer."""
        assert actions.is_cuda and actions.dtype == self.torch.float32 and actions.is_contiguous()
        assert rows.is_cuda and rows.dtype == self.torch.float32 and rows.dim() == 2 and rows.shape[0] == self.N
        assert rows.stride(1) == 1 and rows.stride(0) >= self.A * (self.D + 2)
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        p_obs, p_rew, p_done, p_flags = self._own_ptrs
        _chk(self.L.pgd_step_packed(self.h, C.c_void_p(actions.data_ptr()), C.c_void_p(rows.data_ptr()),
                                    int(rows.stride(0)), p_rew, p_done, p_flags), "pgd_step_packed")
        return rows, self.reward, self.done, self.flags

    def step_n(self, action_ring, first, n_steps, want_obs=True):
        """n_steps steps of `action_ring` [L, N, A, 2] (step k applies ring[(first + k) % L]) in one call (pgd_step_n): returns
        (obs after the last step or None, reward [n_steps, N, A], done, flags); the intermediate steps skip the observation."""
        t = self.torch
        assert action_ring.is_cuda and action_ring.dtype == t.float32 and action_ring.is_contiguous() and action_ring.dim() == 4
        assert tuple(action_ring.shape[1:]) == (self.N, self.A, 2)
        cur = t.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        rew = t.empty((n_steps, self.N, self.A), dtype=t.float32, device=self.obs.device)
        done = t.empty((n_steps, self.N, self.A), dtype=t.uint8, device=self.obs.device)
        flags = t.empty((n_steps, self.N, self.A), dtype=t.int32, device=self.obs.device)
        p_obs = self._own_ptrs[0] if want_obs else None
        _chk(self.L.pgd_step_n(self.h, C.c_void_p(action_ring.data_ptr()), int(action_ring.shape[0]), int(first), int(n_steps), p_obs,
                               C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), C.c_void_p(flags.data_ptr())), "pgd_step_n")
        return (self.obs if want_obs else None), rew, done, flags

    def mlp_prepare(self, weights):
        """The policy network's weights in the split-bf16 kernel's own layout (pgd_mlp_prepare): call once per policy update, hand the
        returned buffer to mlp_policy(prepared=...).  `weights` as for mlp_policy."""
        t = self.torch
        w1, b1, w2, b2, w3, b3 = weights
        for w in weights:
            assert w.is_cuda and w.dtype == t.float32 and w.is_contiguous()
        k = int(w1.shape[0])
        assert w1.shape == (k, 256) and w2.shape == (256, 256) and w3.shape[0] == 256 and w3.shape[1] >= 2
        buf = t.empty(int(self.L.pgd_mlp_prepared_bytes(k)), dtype=t.uint8, device=self.device)
        cur = t.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        _chk(self.L.pgd_mlp_prepare(self.h, k, 256, C.c_void_p(w1.data_ptr()), C.c_void_p(b1.data_ptr()), C.c_void_p(w2.data_ptr()),
                                    C.c_void_p(b2.data_ptr()), C.c_void_p(w3.data_ptr()), C.c_void_p(b3.data_ptr()), int(w3.shape[1]),
                                    C.c_void_p(buf.data_ptr())), "pgd_mlp_prepare")
        buf._pgd_in_dim = k
        return buf

    def specialise(self, wait=True, verbose=False):
        """Build (hipcc, cached) and load a step kernel with THIS engine's configuration compiled in (pgdrive_amd/jit.py):
        configurations the library has no instantiation for then step as fast as the reference's defaults.  wait=False: in a
        background thread, the general kernel steps meanwhile."""
        from . import jit
        return jit.specialise(self, wait=wait, verbose=verbose)

    def describe_step(self):
        """Which step kernel the last step call launched (pgd_describe_step)."""
        buf = C.create_string_buffer(512)
        _chk(self.L.pgd_describe_step(self.h, buf, 512), "pgd_describe_step")
        return buf.value.decode()

    def lane_keep_actions(self, out, tick, obs=None, k_lat=1.0, k_head=2.0, v_target_kmh=30.0, noise=0.05):
        """Scripted lane-keeping actions for the ego from the last observation (pgd_lane_keep_actions): an action stream that
        keeps the ego driving (bench.py --actions expert).  `out` = float32 cuda tensor [N, 1, 2]."""
        assert out.is_cuda and out.dtype == self.torch.float32 and out.is_contiguous() and out.numel() == self.N * 2
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        o = self.obs if obs is None else obs
        _chk(self.L.pgd_lane_keep_actions(self.h, C.c_void_p(o.data_ptr()), C.c_void_p(out.data_ptr()), k_lat, k_head,
                                          v_target_kmh, noise, int(tick) & 0xffffffff), "pgd_lane_keep_actions")
        return out

    def mlp_policy(self, weights, out, obs=None, group=-1, final_tanh=False, in_dim=None, prepared=None):
        """actions = tanh-MLP(observation rows) in one launch (pgd_mlp_policy; the network of examples/ppo_expert/numpy_expert.py).
        `weights` = (w1 [in, 256], b1, w2 [256, 256], b2, w3 [256, >= 2], b3): contiguous float32 cuda tensors, row-major [in][out].
        `obs`: [rows, stride] float32 cuda (default: the engine's own observation buffer); `out`: float32 cuda [N, A, 2].
        group >= 0: only the rows of that env group, on the group's stream (then `obs` / `out` are still the FULL buffers)."""
        t = self.torch
        if prepared is not None:  # split-bf16 kernel on weights prepared by mlp_prepare (`weights` is then ignored)
            o = self.obs if obs is None else obs
            o2 = o.view(-1, o.shape[-1])
            assert o2.is_cuda and o2.dtype == t.float32 and o2.stride(1) == 1 and o2.shape[0] == self.N * self.A
            assert out.is_cuda and out.dtype == t.float32 and out.is_contiguous() and out.numel() == self.N * self.A * 2
            k = int(in_dim if in_dim is not None else prepared._pgd_in_dim)
            if group < 0:
                cur = t.cuda.current_stream(self.device).cuda_stream
                if cur != self._bound_stream:
                    _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
                    self._bound_stream = cur
            _chk(self.L.pgd_mlp_policy_prepared(self.h, int(group), C.c_void_p(o2.data_ptr()), int(o2.stride(0)), k,
                                                C.c_void_p(prepared.data_ptr()), int(bool(final_tanh)), C.c_void_p(out.data_ptr())),
                 "pgd_mlp_policy_prepared")
            return out
        w1, b1, w2, b2, w3, b3 = weights
        o = self.obs if obs is None else obs
        o2 = o.view(-1, o.shape[-1])
        assert o2.is_cuda and o2.dtype == t.float32 and o2.stride(1) == 1 and o2.shape[0] == self.N * self.A
        for w in weights:
            assert w.is_cuda and w.dtype == t.float32 and w.is_contiguous()
        k = int(in_dim if in_dim is not None else w1.shape[0])
        assert w1.shape == (k, 256) and w2.shape == (256, 256) and w3.shape[0] == 256 and w3.shape[1] >= 2 and k <= o2.shape[1]
        assert out.is_cuda and out.dtype == t.float32 and out.is_contiguous() and out.numel() == self.N * self.A * 2
        if group < 0:
            cur = t.cuda.current_stream(self.device).cuda_stream
            if cur != self._bound_stream:
                _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
                self._bound_stream = cur
        _chk(self.L.pgd_mlp_policy(self.h, int(group), C.c_void_p(o2.data_ptr()), int(o2.stride(0)), k, 256,
                                   C.c_void_p(w1.data_ptr()), C.c_void_p(b1.data_ptr()), C.c_void_p(w2.data_ptr()), C.c_void_p(b2.data_ptr()),
                                   C.c_void_p(w3.data_ptr()), C.c_void_p(b3.data_ptr()), int(w3.shape[1]), int(bool(final_tanh)),
                                   C.c_void_p(out.data_ptr())), "pgd_mlp_policy")
        return out

    def step_lane_keep(self, tick, k_lat=1.0, k_head=2.0, v_target_kmh=30.0, noise=0.05):
        """One closed-loop step under the scripted lane-keeping policy (pgd_step_lane_keep): the policy reads the engine's own
        observation buffer (what the previous step / reset wrote), the step rewrites it -- one launch on engines with one env per
        wave.  Same results as lane_keep_actions(...) followed by step(...)."""
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        p_obs, p_rew, p_done, p_flags = self._own_ptrs
        _chk(self.L.pgd_step_lane_keep(self.h, k_lat, k_head, v_target_kmh, noise, int(tick) & 0xffffffff, p_obs, p_rew, p_done, p_flags),
             "pgd_step_lane_keep")
        return self.obs, self.reward, self.done, self.flags

    # -- top-down observation (obs/top_down_obs_multi_channel.py) -----------------------------------------------------------
    def enable_topdown(self, td_cfg=None, uint8=False):
        """Switch on the bird's-eye observation; `td_cfg` = _abi.make_topdown_config(...).  `uint8`: the engine's own image buffer holds
        bytes in [0, 255] (the reference's rgb_clip=False) instead of float32 in [0, 1]."""
        self.td_cfg = td_cfg or _abi.make_topdown_config()
        _chk(self.L.pgd_topdown_enable(self.h, C.byref(self.td_cfg)), "pgd_topdown_enable")
        R, Cn = self.td_cfg.resolution, self.L.pgd_topdown_channels(C.byref(self.td_cfg))
        self.img = self.torch.zeros((self.N, R, R, Cn), dtype=self.torch.uint8 if uint8 else self.torch.float32, device=self.device)

    def observe_topdown(self, out=None):
        """Image [N, R, R, C] of the present state -- float32 in [0, 1], or bytes in [0, 255] when the buffer (`out`, else the engine's
        own) is a uint8 tensor; call once after every step / reset (it advances the frame history)."""
        img = self.img if out is None else out
        assert img.is_cuda and img.is_contiguous() and img.dtype in (self.torch.float32, self.torch.uint8) and img.numel() == self.img.numel()
        cur = self.torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            _chk(self.L.pgd_set_stream(self.h, C.c_void_p(cur)), "pgd_set_stream")
            self._bound_stream = cur
        if img.dtype == self.torch.uint8:
            _chk(self.L.pgd_observe_topdown_u8(self.h, C.c_void_p(img.data_ptr())), "pgd_observe_topdown_u8")
        else:
            _chk(self.L.pgd_observe_topdown(self.h, C.c_void_p(img.data_ptr())), "pgd_observe_topdown")
        return img

    # -- asynchronous env groups ----------------------------------------------------------------------------------------
    def set_groups(self, n_groups):
        """Split the envs into `n_groups` equal contiguous groups with their own internal streams (pgd_set_groups)."""
        self.torch.cuda.synchronize(self.device)
        _chk(self.L.pgd_set_groups(self.h, int(n_groups)), "pgd_set_groups")
        self.n_groups = int(n_groups)
        self.group_streams = []
        for g in range(self.n_groups if n_groups > 1 else 0):
            p = C.c_void_p()
            _chk(self.L.pgd_group_stream(self.h, g, C.byref(p)), "pgd_group_stream")
            self.group_streams.append(self.torch.cuda.ExternalStream(p.value, device=self.device))

    def group_slice(self, g):
        n = self.N // self.n_groups
        return slice(g * n, (g + 1) * n)

    def step_group(self, g, actions):
        """Step only the envs of group g, asynchronously on the group's stream; `actions` is the full [N, A, 2] tensor.
        Returns views of the group's rows of the engine's output buffers."""
        assert actions.is_cuda and actions.dtype == self.torch.float32 and actions.is_contiguous()
        assert actions.numel() == self.N * self.A * 2
        p_obs, p_rew, p_done, p_flags = self._own_ptrs
        _chk(self.L.pgd_step_group(self.h, int(g), C.c_void_p(actions.data_ptr()), p_obs, p_rew, p_done, p_flags), "pgd_step_group")
        sl = self.group_slice(g)
        return self.obs[sl], self.reward[sl], self.done[sl], self.flags[sl]

    def group_sync(self, g):
        _chk(self.L.pgd_group_sync(self.h, int(g)), "pgd_group_sync")

    def observe(self):
        _chk(self.L.pgd_observe(self.h, C.c_void_p(self.obs.data_ptr())), "pgd_observe")
        self.sync()
        return self.obs

    def sync(self):
        _chk(self.L.pgd_sync(self.h), "pgd_sync")

    def enable_step_timing(self, on=True):
        _chk(self.L.pgd_enable_step_timing(self.h, int(bool(on))), "pgd_enable_step_timing")

    def last_step_ms(self):
        ms = C.c_float()
        _chk(self.L.pgd_last_step_ms(self.h, C.byref(ms)), "pgd_last_step_ms")
        return ms.value

    def profile_begin(self, capacity, stride=1):
        _chk(self.L.pgd_profile_begin_strided(self.h, int(capacity), int(stride)), "pgd_profile_begin_strided")

    def profile_end(self):
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        _chk(self.L.pgd_profile_end(self.h, C.byref(a), C.byref(b), C.byref(n)), "pgd_profile_end")
        return dict(k_step_ms=a.value, k_observe_ms=b.value, count=n.value)

    # -- checkpoint / resume ------------------------------------------------------------------------------------------
    def get_state(self):
        f = np.zeros((_abi.NF, self.N, self.V), dtype=np.float32)
        i = np.zeros((_abi.NI, self.N, self.V), dtype=np.int32)
        ei = np.zeros((_abi.NEI, self.N), dtype=np.int32)
        _chk(self.L.pgd_get_state(self.h, _np_p(f), _np_p(i), _np_p(ei)), "pgd_get_state")
        return f, i, ei

    def set_state(self, f, i, ei):
        f = np.ascontiguousarray(f, dtype=np.float32)
        i = np.ascontiguousarray(i, dtype=np.int32)
        ei = np.ascontiguousarray(ei, dtype=np.int32)
        assert f.shape == (_abi.NF, self.N, self.V) and i.shape == (_abi.NI, self.N, self.V)
        _chk(self.L.pgd_set_state(self.h, _np_p(f), _np_p(i), _np_p(ei)), "pgd_set_state")

    def close(self):
        t = getattr(self, "_jit_thread", None)
        if t is not None and t.is_alive():  # a run-time kernel still being built (specialise(wait=False))
            t.join(timeout=120)
        if getattr(self, "h", None):
            self.L.pgd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
