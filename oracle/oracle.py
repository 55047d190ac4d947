"""ctypes binding of the CPU oracle (oracle/libsumo_oracle.so).  TEST INFRASTRUCTURE ONLY:
import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
INFO_STRIDE = 8


def build(force=False):
    so = os.path.join(_HERE, "libsumo_oracle.so")
    src = os.path.join(_HERE, "sumo_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsumo_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.so_create.restype = C.c_void_p
        L.so_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.so_destroy.argtypes = [C.c_void_p]
        L.so_last_error.restype = C.c_char_p
        for name in ("so_dims", "so_reset", "so_step", "so_get_state", "so_set_state", "so_forward", "so_mj_step",
                     "so_get_array", "so_stats", "so_set_maxcon", "so_set_jbcap", "so_set_seeds", "so_set_cfrc_mode", "so_set_adjust_z"):
            getattr(L, name).restype = C.c_int
        L.so_set_adjust_z.argtypes = [C.c_void_p, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleSim:
    """N independent envs stepped serially (or with OpenMP threads) in float64."""

    def __init__(self, model, num_envs, maxcon=None, jbcap=None):
        self.L = lib()
        blob = model.to_blob()
        self._blob = (C.c_char * len(blob)).from_buffer_copy(blob)
        self.h = self.L.so_create(C.byref(self._blob), C.c_size_t(len(blob)), int(num_envs))
        if not self.h:
            raise RuntimeError(self.L.so_last_error().decode())
        self.h = C.c_void_p(self.h)
        dims = np.zeros(10, np.int32)
        self.L.so_dims(self.h, _p(dims))
        (self.nq, self.nv, self.nu, self.nbody, self.njnt, self.ngeom, self.npair, self.nagent, self.obs_stride,
         self.act_stride) = [int(x) for x in dims]
        self.N = int(num_envs)
        if maxcon is not None:
            assert self.L.so_set_maxcon(self.h, int(maxcon)) == 0
        if jbcap is not None:
            assert self.L.so_set_jbcap(self.h, int(jbcap)) == 0

    def set_cfrc_mode(self, mode):
        """'zero' (default: what the reference's MuJoCo 2.1 without force sensors yields) or 'rne_post' (cfrc_ext as
        mj_rnePostConstraint fills it at the start of the last mj_step of an env step)."""
        assert self.L.so_set_cfrc_mode(self.h, {"zero": 0, "rne_post": 1}[mode]) == 0

    def set_adjust_z(self, adjust_z):
        """``Agent._adjust_z`` (agents.py:33,155-161): shifts the observed own / opponent z and the lose test; 0 = training."""
        assert self.L.so_set_adjust_z(self.h, float(adjust_z)) == 0

    def __del__(self):
        try:
            if self.h:
                self.L.so_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self, seeds=None, mask=None):
        obs = np.zeros((self.N, 2, self.obs_stride), np.float32)
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.so_reset(self.h, _p(s), _p(mk), _p(obs))
        return obs

    def step(self, actions, nthreads=1):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.N, 2, self.act_stride)
        obs = np.zeros((self.N, 2, self.obs_stride), np.float32)
        info = np.zeros((self.N, 2, INFO_STRIDE), np.float64)
        done = np.zeros((self.N, 2), np.uint8)
        ep_r = np.zeros(self.N)
        ep_dr = np.zeros(self.N)
        ep_l = np.zeros(self.N, np.int32)
        self.L.so_step(self.h, _p(a), _p(obs), _p(info), _p(done), _p(ep_r), _p(ep_dr), _p(ep_l), int(nthreads))
        return obs, info, done, ep_r, ep_dr, ep_l

    def get_state(self):
        qpos = np.zeros((self.N, self.nq))
        qvel = np.zeros((self.N, self.nv))
        warm = np.zeros((self.N, self.nv))
        cnt = np.zeros((self.N, 2), np.int32)
        self.L.so_get_state(self.h, _p(qpos), _p(qvel), _p(warm), _p(cnt))
        return qpos, qvel, warm, cnt

    def set_state(self, qpos=None, qvel=None, warm=None, counters=None):
        f = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        qpos, qvel, warm, counters = f(qpos, np.float64), f(qvel, np.float64), f(warm, np.float64), f(counters, np.int32)
        self._keep = (qpos, qvel, warm, counters)
        self.L.so_set_state(self.h, _p(qpos), _p(qvel), _p(warm), _p(counters))

    def set_seeds(self, seeds):
        s = np.ascontiguousarray(seeds, np.uint64)
        self.L.so_set_seeds(self.h, _p(s))

    def forward(self, e=0, ctrl=None):
        c = None if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
        self.L.so_forward(self.h, int(e), _p(c))

    def mj_step(self, e=0, ctrl=None, n=1):
        c = None if ctrl is None else np.ascontiguousarray(ctrl, np.float64)
        self.L.so_mj_step(self.h, int(e), _p(c), int(n))

    def array(self, name, e=0, cap=1 << 16):
        out = np.zeros(cap)
        n = self.L.so_get_array(self.h, int(e), name.encode(), _p(out), cap)
        if n < 0:
            raise RuntimeError("buffer too small for %s" % name)
        return out[:n].copy()

    def stats(self):
        o = np.zeros(11)
        self.L.so_stats(self.h, _p(o))
        return dict(forward=o[0], newton=o[1], contacts=o[2], efc=o[3], max_ncon=o[4], max_nefc=o[5],
                    max_newton=o[6], dropped=o[7], diverged=o[8], capsule_box_3=o[9], rod_endcap=o[10])
