"""ctypes binding of include/sumo_ppo.h (csrc/libsumo_ppo.so).  No CPU fallback."""
import ctypes as C
import os

from . import build as _build

NSTATS = 8
FWD_PI, FWD_VF, FWD_TANH = 1, 2, 4
LSTM_GATES_IFOU, LSTM_GATES_IJFO = 0, 1
_LIB = None

EXPORTS = ("ppo_last_error", "ppo_param_count", "ppo_forward", "ppo_forward_filtered", "ppo_lstm_step", "ppo_lstm_step_pool", "ppo_lstm_step_save", "ppo_lstm_xproj", "ppo_lstm_step_save_z", "ppo_lstm_seq_forward", "ppo_lstm_seq_backward", "ppo_lstm_head_grad", "ppo_lstm_bwd_step", "ppo_lstm_wgrad_workspace_bytes", "ppo_lstm_wgrad", "ppo_selfplay_forward", "ppo_post_step", "ppo_reward_mix", "ppo_vtrace", "ppo_adv_moments", "ppo_adv_moments_ws", "ppo_adv_moments_workspace_bytes",
           "ppo_adv_normalize", "ppo_grad_workspace_bytes", "ppo_grad", "ppo_loss_stats", "ppo_clip_adam")


class PpoHipError(RuntimeError):
    pass


class LstmNet(C.Structure):
    """``ppo_lstm_net`` of include/sumo_ppo.h (device pointers as integers; 0 / None = absent)."""
    _fields_ = [("ob_dim", C.c_int), ("emb_dim", C.c_int), ("hidden", C.c_int), ("ac_dim", C.c_int), ("gate_order", C.c_int),
                ("forget_bias", C.c_float), ("obs_mean", C.c_void_p), ("obs_invstd", C.c_void_p), ("obs_clip", C.c_float),
                ("emb_w", C.c_void_p), ("emb_b", C.c_void_p), ("wx", C.c_void_p), ("wh", C.c_void_p), ("b", C.c_void_p),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("logstd", C.c_void_p), ("vf_w", C.c_void_p), ("vf_b", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("SUMO_PPO_LIB") or _build.lib_path("libsumo_ppo.so")
        if not os.path.exists(path):
            raise PpoHipError("%s not found: build it with `python -m robosumo_selfplay_amd.build`" % path)
        L = C.CDLL(path)
        vp, i32, f64, f32 = C.c_void_p, C.c_int, C.c_double, C.c_float
        L.ppo_last_error.restype = C.c_char_p
        L.ppo_param_count.argtypes = [i32, i32]
        L.ppo_forward.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_forward_filtered.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_step.argtypes = [C.POINTER(LstmNet), vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_step_pool.argtypes = [C.POINTER(LstmNet), vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_step_save.argtypes = [C.POINTER(LstmNet), vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        L.ppo_lstm_xproj.argtypes = [C.POINTER(LstmNet), vp, i32, i32, vp, vp]
        L.ppo_lstm_step_save_z.argtypes = [C.POINTER(LstmNet), vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_seq_forward.argtypes = [C.POINTER(LstmNet), vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_seq_backward.argtypes = [C.POINTER(LstmNet), i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_head_grad.argtypes = [C.POINTER(LstmNet), vp, i32, vp, vp, vp, vp, vp, f64, f32, f32, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_bwd_step.argtypes = [C.POINTER(LstmNet), i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_lstm_wgrad_workspace_bytes.argtypes = [i32, i32, i32]
        L.ppo_lstm_wgrad_workspace_bytes.restype = C.c_size_t
        L.ppo_lstm_wgrad.argtypes = [C.POINTER(LstmNet), i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp]
        L.ppo_reward_mix.argtypes = [vp, i32, f64, vp, i32, vp]
        L.ppo_selfplay_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), vp]
        L.ppo_post_step.argtypes = [vp, i32, f64, vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ppo_vtrace.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, f64, f64, f64, f64, vp, vp, vp, vp, vp]
        L.ppo_adv_moments.argtypes = [vp, vp, vp, i32, vp, vp]
        L.ppo_adv_moments_ws.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        L.ppo_adv_moments_workspace_bytes.argtypes = []
        L.ppo_adv_moments_workspace_bytes.restype = C.c_size_t
        L.ppo_adv_normalize.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        L.ppo_grad_workspace_bytes.argtypes = [i32, i32]
        L.ppo_grad_workspace_bytes.restype = C.c_size_t
        L.ppo_grad.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, f64, f32, f32, f32, vp, vp, vp, vp, vp]
        L.ppo_loss_stats.argtypes = [vp, vp, i32, vp, vp]
        L.ppo_clip_adam.argtypes = [vp, vp, vp, vp, i32, i32, f64, f64, f64, f64, f64, vp, vp]
        for n in EXPORTS:
            if n not in ("ppo_last_error", "ppo_grad_workspace_bytes", "ppo_lstm_wgrad_workspace_bytes", "ppo_adv_moments_workspace_bytes"):
                getattr(L, n).restype = i32
        _LIB = L
    return _LIB


def chk(rc):
    if rc != 0:
        raise PpoHipError("sumo_ppo error %d: %s" % (rc, lib().ppo_last_error().decode()))


def ptr(t):
    return None if t is None else t.data_ptr()
