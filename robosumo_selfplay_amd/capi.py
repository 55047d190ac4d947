"""ctypes binding of the engine's C ABI (include/sumo_hip.h -> csrc/libsumo_hip.so).

This is the stub a maintainer of the reference would drop in next to subproc_vec_env.py.
There is NO CPU fallback: if the library is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

INFO_STRIDE = 8
NDIMS = 16
NSTATS = 13
_LIB = None


class SumoHipError(RuntimeError):
    pass


class Rollout(C.Structure):
    """``sumo_rollout`` of include/sumo_hip.h (device pointers as integers)."""
    _fields_ = [("learner_params", C.c_void_p), ("opponent_params", C.c_void_p), ("opponent_index", C.c_void_p),
                ("npool", C.c_int), ("ob_dim", C.c_int), ("ac_dim", C.c_int),
                ("T", C.c_int), ("Ntot", C.c_int), ("env_offset", C.c_int), ("s0", C.c_int), ("K", C.c_int),
                ("alpha", C.c_double), ("noise0", C.c_void_p), ("noise1", C.c_void_p),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("val", C.c_void_p), ("nlp", C.c_void_p), ("onlp", C.c_void_p),
                ("done", C.c_void_p), ("ep_done", C.c_void_p), ("ep_r", C.c_void_p), ("ep_l", C.c_void_p)]


class RolloutLstm(C.Structure):
    """``sumo_rollout_lstm`` of include/sumo_hip.h (``learner``: pointer to a host ``ppo_capi.LstmNet``; the rest device pointers)."""
    _fields_ = [("learner", C.c_void_p), ("opponents_dev", C.c_void_p), ("tile_net_dev", C.c_void_p), ("npool", C.c_int),
                ("state0", C.c_void_p), ("state1", C.c_void_p),
                ("T", C.c_int), ("Ntot", C.c_int), ("env_offset", C.c_int), ("s0", C.c_int), ("K", C.c_int),
                ("alpha", C.c_double), ("noise0", C.c_void_p), ("noise1", C.c_void_p),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("val", C.c_void_p), ("nlp", C.c_void_p), ("onlp", C.c_void_p),
                ("done", C.c_void_p), ("ep_done", C.c_void_p), ("ep_r", C.c_void_p), ("ep_l", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("SUMO_HIP_LIB") or _build.lib_path("libsumo_hip.so")  # override: profiling builds
        if not os.path.exists(path):
            raise SumoHipError("%s not found: build it with `python -m robosumo_selfplay_amd.build` "
                               "(the HIP engine has no CPU fallback)" % path)
        L = C.CDLL(path)
        L.sumo_last_error.restype = C.c_char_p
        vp, i32 = C.c_void_p, C.c_int
        L.sumo_create.argtypes = [vp, C.c_size_t, i32, i32, C.POINTER(vp)]
        L.sumo_destroy.argtypes = [vp]
        L.sumo_dims.argtypes = [vp, vp]
        L.sumo_reset.argtypes = [vp, vp, vp, vp, vp]
        L.sumo_step.argtypes = [vp] * 9
        L.sumo_rollout_steps.argtypes = [vp, C.POINTER(Rollout)] + [vp] * 8
        L.sumo_rollout_steps_lstm.argtypes = [vp, C.POINTER(RolloutLstm)] + [vp] * 8
        L.sumo_get_state.argtypes = [vp] * 5
        L.sumo_set_cfrc_mode.argtypes = [vp, i32]
        L.sumo_get_cfrc_ext.argtypes = [vp, vp]
        L.sumo_set_adjust_z.argtypes = [vp, C.c_double]
        L.sumo_set_state.argtypes = [vp] * 5
        L.sumo_debug_forward.argtypes = [vp] * 4
        L.sumo_stats.argtypes = [vp, vp]
        L.sumo_profile.argtypes = [vp, vp]
        L.sumo_debug_trace.argtypes = [vp, vp]
        L.sumo_debug_trace.restype = i32
        L.sumo_rollout_status.argtypes = [vp, vp]
        L.sumo_rollout_status.restype = i32
        L.sumo_debug_fault.argtypes = [vp, i32]
        L.sumo_debug_fault.restype = i32
        L.sumo_static_layout.argtypes = [vp]
        L.sumo_static_layout.restype = i32
        L.sumo_profile.restype = i32
        for n in ("sumo_create", "sumo_destroy", "sumo_dims", "sumo_reset", "sumo_step", "sumo_rollout_steps", "sumo_rollout_steps_lstm", "sumo_get_state",
                  "sumo_set_cfrc_mode", "sumo_get_cfrc_ext", "sumo_set_adjust_z", "sumo_set_state", "sumo_debug_forward", "sumo_stats"):
            getattr(L, n).restype = i32
        _LIB = L
    return _LIB


EXPORTS = ("sumo_last_error", "sumo_create", "sumo_destroy", "sumo_dims", "sumo_reset", "sumo_step", "sumo_rollout_steps",
           "sumo_rollout_steps_lstm", "sumo_set_cfrc_mode", "sumo_get_cfrc_ext", "sumo_set_adjust_z", "sumo_get_state", "sumo_set_state", "sumo_debug_forward", "sumo_stats", "sumo_profile", "sumo_debug_trace",
           "sumo_rollout_status", "sumo_debug_fault", "sumo_static_layout", "sumo_debug_layout", "sumo_debug_model_ints", "sumo_debug_dump")


def _np(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc):
    if rc != 0:
        raise SumoHipError("sumo_hip error %d: %s" % (rc, lib().sumo_last_error().decode()))


class Engine:
    """Thin owner of one ``sumo_handle_t``; device buffers are passed as raw pointers (ints)."""

    def __init__(self, model, num_envs, device=0):
        L = lib()
        blob = model.to_blob()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        h = C.c_void_p()
        _chk(L.sumo_create(C.cast(buf, C.c_void_p), len(blob), int(num_envs), int(device), C.byref(h)))
        self.h = h
        self.N = int(num_envs)
        d = np.zeros(NDIMS, np.int32)
        _chk(L.sumo_dims(self.h, _np(d)))
        (self.nq, self.nv, self.nu, self.nbody, self.njnt, self.ngeom, self.npair, self.nagent, self.obs_stride,
         self.act_stride, self.maxcon, self.maxefc, self.lds_bytes, self.state_stride, self.jbcap) = [int(x) for x in d[:15]]

    def close(self):
        if getattr(self, "h", None):
            lib().sumo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, obs_ptr, seeds=None, mask_ptr=None, stream=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        if s is not None and s.shape != (self.N,):
            raise ValueError("seeds must have shape (%d,)" % self.N)
        _chk(lib().sumo_reset(self.h, _np(s), mask_ptr, obs_ptr, stream))

    def step(self, actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream=None):
        _chk(lib().sumo_step(self.h, actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream))

    def rollout_steps(self, ro, actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream=None):
        """K fused self-play rollout steps (``sumo_rollout_steps``); ``ro`` is a filled :class:`Rollout`."""
        _chk(lib().sumo_rollout_steps(self.h, C.byref(ro), actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream))

    def rollout_steps_lstm(self, ro, actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream=None):
        """The same for recurrent policies (``sumo_rollout_steps_lstm``); ``ro`` is a filled :class:`RolloutLstm`."""
        _chk(lib().sumo_rollout_steps_lstm(self.h, C.byref(ro), actions_ptr, obs_ptr, info_ptr, done_ptr, ep_r_ptr, ep_dr_ptr, ep_l_ptr, stream))

    def set_cfrc_mode(self, mode):
        """'zero' (default, the reference's behaviour) or 'rne_post' (include/sumo_hip.h: cfrc_mode)."""
        _chk(lib().sumo_set_cfrc_mode(self.h, {"zero": 0, "rne_post": 1}[mode]))

    def set_adjust_z(self, adjust_z):
        """``Agent._adjust_z`` of the reference (agents.py:33,155-161; include/sumo_hip.h: sumo_set_adjust_z)."""
        _chk(lib().sumo_set_adjust_z(self.h, float(adjust_z)))

    def get_cfrc_ext(self):
        out = np.zeros((self.N, self.nbody, 6))
        _chk(lib().sumo_get_cfrc_ext(self.h, _np(out)))
        return out

    def get_state(self):
        qpos = np.zeros((self.N, self.nq))
        qvel = np.zeros((self.N, self.nv))
        warm = np.zeros((self.N, self.nv))
        cnt = np.zeros((self.N, 2), np.int32)
        _chk(lib().sumo_get_state(self.h, _np(qpos), _np(qvel), _np(warm), _np(cnt)))
        return qpos, qvel, warm, cnt

    def set_state(self, qpos=None, qvel=None, warm=None, counters=None):
        def f(a, dt, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            if a.shape != shape:
                raise ValueError("bad shape %s, expected %s" % (a.shape, shape))
            return a
        qpos = f(qpos, np.float64, (self.N, self.nq))
        qvel = f(qvel, np.float64, (self.N, self.nv))
        warm = f(warm, np.float64, (self.N, self.nv))
        counters = f(counters, np.int32, (self.N, 2))
        _chk(lib().sumo_set_state(self.h, _np(qpos), _np(qvel), _np(warm), _np(counters)))

    def debug_forward(self, ctrl):
        ctrl = np.ascontiguousarray(ctrl, np.float64)
        if ctrl.shape != (self.N, self.nu):
            raise ValueError("ctrl must have shape (%d, %d)" % (self.N, self.nu))
        qacc = np.zeros((self.N, self.nv))
        counts = np.zeros((self.N, 4), np.int32)
        _chk(lib().sumo_debug_forward(self.h, _np(ctrl), _np(qacc), _np(counts)))
        return qacc, counts

    def debug_trace(self, stamps_ptr):
        """Development: device uint64 [N][4] buffer that receives each env wave's start / end stamps and work counters (0 / None = off)."""
        _chk(lib().sumo_debug_trace(self.h, stamps_ptr or None))

    def profile(self):
        o = np.zeros(24)
        _chk(lib().sumo_profile(self.h, _np(o)))
        return o

    def stats(self):
        o = np.zeros(NSTATS)
        _chk(lib().sumo_stats(self.h, _np(o)))
        return dict(forward=o[0], newton=o[1], contacts=o[2], efc=o[3], max_ncon=o[4], max_nefc=o[5],
                    max_newton=o[6], dropped=o[7], diverged=o[8], rollout_aborts=o[9], handover_mismatches=o[10],
                    capsule_box_3=o[11], rod_endcap=o[12])

    def rollout_status(self):
        """Waits for the engine's most recent fused rollout launch and RAISES if it was cut short (``sumo_rollout_status``:
        expired hand-over wait or a hand-over tag / checksum mismatch).  Returns dict(aborted, tickets_drawn, tickets, mismatches)."""
        o = np.zeros(4, np.int64)
        _chk(lib().sumo_rollout_status(self.h, _np(o)))
        return dict(aborted=int(o[0]), tickets_drawn=int(o[1]), tickets=int(o[2]), mismatches=int(o[3]))

    def static_layout(self):
        """True if this engine runs static-Layout kernel variants (Ant-vs-Ant or Spider-vs-Spider at default settings; include/sumo_hip.h)."""
        return lib().sumo_static_layout(self.h) >= 1

    def debug_fault(self, env):
        """Tests: make the first hand-over of ``env`` in the following fused launches carry a wrong checksum (-1 = off)."""
        _chk(lib().sumo_debug_fault(self.h, int(env)))
