"""Fixed opponents from the OpenAI RoboSumo policy zoo + the win/draw/lose evaluator, on the HIP forward kernel.

Reference:
  * robosumo/robosumo/policy_zoo/policy.py:23-91   ``MLPPolicy`` (tanh 64-64 value and policy trunks, state-independent
    logstd, running-mean observation filter clipped to +-5, value de-normalised by the return filter)
  * robosumo/robosumo/policy_zoo/utils.py:9-33     ``RunningMeanStd`` (sum / sumsq / count, std = sqrt(max(var, 1e-2)))
  * robosumo/robosumo/policy_zoo/utils.py:66-83    flat ``.npy`` parameter vector in TF variable-creation order
  * eval_robosumo_against_fix.py:150-230           evaluation loop (learner deterministic on obs[:,0], zoo opponent
    deterministic on obs[:,1,:-1], 'winner' bookkeeping)
  * alg_ppo.py:194-206                             ``opponent_mode='fix'``

MLP and LSTM zoo nets are built (policy.py:23-91 and :94-199); the LSTM one keeps a per-env recurrent state on the device.
The ``.npy`` files are loaded with ``numpy.load(allow_pickle=False)``.
"""
import numpy as np

from . import ppo_capi
from .policies import HIDDEN, flatten_params

# TF variable-creation order of MLPPolicy(normalize=True) -- policy.py:38-70
_ZOO_MLP_ORDER = ["retfilter/sum", "retfilter/sumsq", "retfilter/count", "obsfilter/sum", "obsfilter/sumsq", "obsfilter/count",
                  "vffc1/w", "vffc1/b", "vffc2/w", "vffc2/b", "vffinal/w", "vffinal/b",
                  "polfc1/w", "polfc1/b", "polfc2/w", "polfc2/b", "polfinal/w", "polfinal/b", "logstd"]


def zoo_mlp_shapes(ob_dim, ac_dim, hidden=HIDDEN):
    h = hidden
    return {"retfilter/sum": (), "retfilter/sumsq": (), "retfilter/count": (),
            "obsfilter/sum": (ob_dim,), "obsfilter/sumsq": (ob_dim,), "obsfilter/count": (),
            "vffc1/w": (ob_dim, h), "vffc1/b": (h,), "vffc2/w": (h, h), "vffc2/b": (h,), "vffinal/w": (h, 1), "vffinal/b": (1,),
            "polfc1/w": (ob_dim, h), "polfc1/b": (h,), "polfc2/w": (h, h), "polfc2/b": (h,), "polfinal/w": (h, ac_dim),
            "polfinal/b": (ac_dim,), "logstd": (1, ac_dim)}


def zoo_mlp_param_count(ob_dim, ac_dim, hidden=HIDDEN):
    return int(sum(int(np.prod(s)) for s in zoo_mlp_shapes(ob_dim, ac_dim, hidden).values()))


def infer_ob_dim(nparams, ac_dim, hidden=HIDDEN):
    """Observation width of a flat zoo MLP vector (the count is affine in ob_dim)."""
    c0, c1 = zoo_mlp_param_count(0, ac_dim, hidden), zoo_mlp_param_count(1, ac_dim, hidden)
    d, r = divmod(nparams - c0, c1 - c0)
    if r != 0 or d <= 0:
        raise ValueError("%d parameters do not fit a zoo MLP policy with %d actions" % (nparams, ac_dim))
    return int(d)


def split_zoo_mlp(flat, ac_dim, hidden=HIDDEN):
    """utils.py:70-83 ``set_from_flat``: consecutive slices in variable order."""
    flat = np.asarray(flat, np.float32).ravel()
    ob_dim = infer_ob_dim(flat.size, ac_dim, hidden)
    shapes = zoo_mlp_shapes(ob_dim, ac_dim, hidden)
    out, o = {}, 0
    for k in _ZOO_MLP_ORDER:
        n = int(np.prod(shapes[k]))
        out[k] = flat[o:o + n].reshape(shapes[k]).copy()
        o += n
    return ob_dim, out


def filter_stats(p, prefix):
    """RunningMeanStd.mean / .std (utils.py:30-32), float32 like the TF graph."""
    cnt = np.float32(p[prefix + "/count"])
    mean = (p[prefix + "/sum"] / cnt).astype(np.float32)
    var = (p[prefix + "/sumsq"] / cnt).astype(np.float32) - np.square(mean)
    std = np.sqrt(np.maximum(var, np.float32(1e-2))).astype(np.float32)
    return mean, std


class ZooMLPPolicy(object):
    """One zoo MLP net resident on the GPU.  ``act`` follows policy.py:72-79; ``step`` / ``value`` /
    ``action_probability`` follow the PolicyWithValue surface so the object can sit in ``Runner.models[1]``
    (``opponent_mode='fix'``).  Observations may carry extra trailing columns (the time feature): only the first
    ``ob_dim`` are read, which is the ``obs[:, 1, :-1]`` of eval_robosumo_against_fix.py:206."""

    initial_state = None
    recurrent = False

    def __init__(self, flat_params, ac_dim, device=0):
        import torch
        self._t = torch
        self.device = torch.device("cuda", int(device)) if not isinstance(device, torch.device) else device
        self.ac_dim = int(ac_dim)
        self.ob_dim, p = split_zoo_mlp(flat_params, ac_dim)
        self.tensors = p
        mean, std = filter_stats(p, "obsfilter")
        self.ret_mean, self.ret_std = [float(x) for x in filter_stats(p, "retfilter")]
        # the kernel's flat layout (sumo_ppo.h): pi trunk, vf trunk, pi head, logstd, vf head
        flat = flatten_params([p["polfc1/w"], p["polfc1/b"], p["polfc2/w"], p["polfc2/b"], p["vffc1/w"], p["vffc1/b"],
                               p["vffc2/w"], p["vffc2/b"], p["polfinal/w"], p["polfinal/b"], p["logstd"], p["vffinal/w"],
                               p["vffinal/b"]])
        self.params = torch.from_numpy(flat).to(self.device)
        self.obs_mean = torch.from_numpy(mean).to(self.device)
        self.obs_invstd = torch.from_numpy((np.float32(1.0) / std).astype(np.float32)).to(self.device)
        self.gen = torch.Generator(device=self.device)

    def seed(self, s):
        self.gen.manual_seed(int(s))

    def reset(self, **kwargs):   # policy.py:13-15
        pass

    def _prep(self, x, min_cols):
        t = self._t
        np_in = isinstance(x, np.ndarray) or not t.is_tensor(x)
        if np_in:
            x = np.asarray(x, np.float32)
            if x.ndim == 1:
                x = x[None]
            x = t.from_numpy(np.ascontiguousarray(x)).to(self.device)
        if x.dtype != t.float32 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1 or x.shape[1] < min_cols:
            raise ValueError("expected float32 [n, >=%d] observations with unit inner stride" % min_cols)
        return x, np_in

    def evaluate(self, obs, flags, given_action=None, deterministic=False, out=None):
        """Same contract as ``PolicyWithValue.evaluate`` (device tensors in, dict of device tensors out; ``out`` may hold
        preallocated outputs), so the device-mode Runner can drive a zoo opponent."""
        t = self._t
        ob, _ = self._prep(obs, self.ob_dim)
        n, A = ob.shape[0], self.ac_dim
        out = out or {}
        action = neglogp = value = noise = given = None
        if flags & ppo_capi.FWD_PI:
            action = out.get("action")
            if action is None:
                action = t.empty((n, A), dtype=t.float32, device=self.device)
            neglogp = out.get("neglogp")
            if neglogp is None:
                neglogp = t.empty(n, dtype=t.float32, device=self.device)
            if given_action is not None:
                given, _ = self._prep(given_action, A)
                given = given.contiguous()
            elif not deterministic:
                noise = t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32)
        if flags & ppo_capi.FWD_VF:
            value = out.get("value")
            if value is None:
                value = t.empty(n, dtype=t.float32, device=self.device)
        st = t.cuda.current_stream(self.device).cuda_stream
        ppo_capi.chk(ppo_capi.lib().ppo_forward_filtered(
            self.params.data_ptr(), ob.data_ptr(), n, ob.stride(0) if n > 1 else ob.shape[1], self.ob_dim, A, flags | ppo_capi.FWD_TANH,
            self.obs_mean.data_ptr(), self.obs_invstd.data_ptr(), 5.0, ppo_capi.ptr(noise), ppo_capi.ptr(given),
            ppo_capi.ptr(action), ppo_capi.ptr(neglogp), ppo_capi.ptr(value), None, st))
        if value is not None:
            value.mul_(self.ret_std).add_(self.ret_mean)          # policy.py:58-60
        return dict(action=action, neglogp=neglogp, value=value)

    @staticmethod
    def _np(x):
        return isinstance(x, np.ndarray) or not hasattr(x, "is_cuda")

    def _ret(self, x, np_in):
        return x.cpu().numpy() if np_in else x

    # ---- zoo surface (policy.py:72-79) ----------------------------------------------------------------------------
    def act(self, observation, stochastic=True):
        np_in = self._np(observation)
        single = np_in and np.asarray(observation).ndim == 1
        r = self.evaluate(observation, ppo_capi.FWD_PI | ppo_capi.FWD_VF, deterministic=not stochastic)
        a, v = self._ret(r["action"], np_in), self._ret(r["value"], np_in)
        return (a[0], {"vpred": v[0]}) if single else (a, {"vpred": v})

    # ---- PolicyWithValue surface (policies.py:84-128 of the reference) -------------------------------------------
    def step(self, observation, deterministic=False, **extra_feed):
        np_in = self._np(observation)
        r = self.evaluate(observation, ppo_capi.FWD_PI | ppo_capi.FWD_VF, deterministic=deterministic)
        return self._ret(r["action"], np_in), self._ret(r["value"], np_in), None, self._ret(r["neglogp"], np_in)

    def value(self, ob, *args, **kwargs):
        return self._ret(self.evaluate(ob, ppo_capi.FWD_VF)["value"], self._np(ob))

    def action_probability(self, observation, given_action=None, **extra_feed):
        return self._ret(self.evaluate(observation, ppo_capi.FWD_PI, given_action=given_action)["neglogp"], self._np(observation))


# TF variable-creation order of LSTMPolicy(hiddens=[E, H], normalize=True) -- policy.py:106-183
_ZOO_LSTM_ORDER = ["retfilter/sum", "retfilter/sumsq", "retfilter/count", "obsfilter/sum", "obsfilter/sumsq", "obsfilter/count",
                   "v/emb/w", "v/emb/b", "lstmv/kernel", "lstmv/bias", "v/out/w", "v/out/b",
                   "p/emb/w", "p/emb/b", "lstmp/kernel", "lstmp/bias", "p/out/w", "p/out/b", "logstd"]


def zoo_lstm_shapes(ob_dim, ac_dim, emb=HIDDEN, hidden=HIDDEN):
    e, h = emb, hidden
    return {"retfilter/sum": (), "retfilter/sumsq": (), "retfilter/count": (),
            "obsfilter/sum": (ob_dim,), "obsfilter/sumsq": (ob_dim,), "obsfilter/count": (),
            "v/emb/w": (ob_dim, e), "v/emb/b": (e,), "lstmv/kernel": (e + h, 4 * h), "lstmv/bias": (4 * h,), "v/out/w": (h, 1), "v/out/b": (1,),
            "p/emb/w": (ob_dim, e), "p/emb/b": (e,), "lstmp/kernel": (e + h, 4 * h), "lstmp/bias": (4 * h,), "p/out/w": (h, ac_dim),
            "p/out/b": (ac_dim,), "logstd": (1, ac_dim)}


def zoo_lstm_param_count(ob_dim, ac_dim, emb=HIDDEN, hidden=HIDDEN):
    return int(sum(int(np.prod(s)) for s in zoo_lstm_shapes(ob_dim, ac_dim, emb, hidden).values()))


def split_zoo_lstm(flat, ac_dim, emb=HIDDEN, hidden=HIDDEN):
    flat = np.asarray(flat, np.float32).ravel()
    c0, c1 = zoo_lstm_param_count(0, ac_dim, emb, hidden), zoo_lstm_param_count(1, ac_dim, emb, hidden)
    ob_dim, r = divmod(flat.size - c0, c1 - c0)
    if r != 0 or ob_dim <= 0:
        raise ValueError("%d parameters do not fit a zoo LSTM policy with %d actions" % (flat.size, ac_dim))
    shapes = zoo_lstm_shapes(int(ob_dim), ac_dim, emb, hidden)
    out, o = {}, 0
    for k in _ZOO_LSTM_ORDER:
        n = int(np.prod(shapes[k]))
        out[k] = flat[o:o + n].reshape(shapes[k]).copy()
        o += n
    return int(ob_dim), out


class ZooLSTMPolicy(object):
    """policy.py:94-199 on the device: observation filter -> relu embedding (64) -> BasicLSTMCell(64) -> head, separately
    for the value and the policy (two cells).  The recurrent state of every env lives in ``self.state`` ([4][n][64]:
    value c, value h, policy c, policy h -- the reference's ``zero_state`` order); ``reset(mask)`` zeroes the rows of
    finished episodes (the reference calls ``policy.reset()`` when an episode starts)."""

    recurrent = True

    def __init__(self, flat_params, ac_dim, device=0, emb=HIDDEN, hidden=HIDDEN):
        import torch
        self._t = torch
        self.device = torch.device("cuda", int(device)) if not isinstance(device, torch.device) else device
        self.ac_dim, self.emb, self.hidden = int(ac_dim), int(emb), int(hidden)
        self.ob_dim, p = split_zoo_lstm(flat_params, ac_dim, emb, hidden)
        self.tensors = p
        mean, std = filter_stats(p, "obsfilter")
        self.ret_mean, self.ret_std = [float(x) for x in filter_stats(p, "retfilter")]
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        self.dev = {k: dev(v) for k, v in p.items() if "/" in k and not k.startswith(("retfilter", "obsfilter"))}
        self.dev["logstd"] = dev(p["logstd"])
        self.obs_mean, self.obs_invstd = dev(mean), dev(np.float32(1.0) / std)
        self.gen = torch.Generator(device=self.device)
        self.state = None
        self._nets = {}
        for br, cell, head in (("v", "lstmv", None), ("p", "lstmp", "p/out")):
            n = ppo_capi.LstmNet()
            n.ob_dim, n.emb_dim, n.hidden, n.ac_dim = self.ob_dim, self.emb, self.hidden, self.ac_dim
            n.gate_order, n.forget_bias = ppo_capi.LSTM_GATES_IJFO, 1.0          # tf BasicLSTMCell: (i, j, f, o), forget_bias 1
            n.obs_mean, n.obs_invstd, n.obs_clip = self.obs_mean.data_ptr(), self.obs_invstd.data_ptr(), 5.0
            n.emb_w, n.emb_b = self.dev[br + "/emb/w"].data_ptr(), self.dev[br + "/emb/b"].data_ptr()
            k = self.dev[cell + "/kernel"]
            n.wx, n.wh, n.b = k.data_ptr(), k.data_ptr() + 4 * self.emb * 4 * self.hidden, self.dev[cell + "/bias"].data_ptr()
            if head:
                n.head_w, n.head_b, n.logstd = self.dev["p/out/w"].data_ptr(), self.dev["p/out/b"].data_ptr(), self.dev["logstd"].data_ptr()
            else:
                n.vf_w, n.vf_b = self.dev["v/out/w"].data_ptr(), self.dev["v/out/b"].data_ptr()
            self._nets[br] = n

    def seed(self, s):
        self.gen.manual_seed(int(s))

    def reset(self, mask=None, **kwargs):
        """policy.py:198-199; ``mask`` (bool / uint8 [n], host or device) restricts the reset to those rows."""
        if self.state is None:
            return
        if mask is None:
            self.state.zero_()
        else:
            m = self._t.as_tensor(mask, device=self.device).to(self._t.bool)
            self.state[:, m, :] = 0.0

    def act(self, observation, stochastic=True, want_value=False):
        t = self._t
        np_in = isinstance(observation, np.ndarray) or not t.is_tensor(observation)
        x = observation
        single = False
        if np_in:
            x = np.asarray(x, np.float32)
            single = x.ndim == 1
            x = t.from_numpy(np.ascontiguousarray(x[None] if single else x)).to(self.device)
        if x.dtype != t.float32 or x.dim() != 2 or x.stride(1) != 1 or x.shape[1] < self.ob_dim:
            raise ValueError("expected float32 [n, >=%d] observations with unit inner stride" % self.ob_dim)
        n, A, H = x.shape[0], self.ac_dim, self.hidden
        if self.state is None or self.state.shape[1] != n:
            self.state = t.zeros((4, n, H), dtype=t.float32, device=self.device)
        action = t.empty((n, A), dtype=t.float32, device=self.device)
        noise = t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32) if stochastic else None
        st = t.cuda.current_stream(self.device).cuda_stream
        stride = x.stride(0) if n > 1 else x.shape[1]
        L = ppo_capi.lib()
        import ctypes as C
        ppo_capi.chk(L.ppo_lstm_step(C.byref(self._nets["p"]), x.data_ptr(), n, stride, None, self.state[2].data_ptr(),
                                     self.state[3].data_ptr(), H, ppo_capi.ptr(noise), None, action.data_ptr(), None, None, None, st))
        info = {"state": self.state}
        if want_value:
            value = t.empty(n, dtype=t.float32, device=self.device)
            ppo_capi.chk(L.ppo_lstm_step(C.byref(self._nets["v"]), x.data_ptr(), n, stride, None, self.state[0].data_ptr(),
                                         self.state[1].data_ptr(), H, None, None, None, None, value.data_ptr(), None, st))
            value = value * self.ret_std + self.ret_mean
            info["vpred"] = value.cpu().numpy() if np_in else value
            if single:
                info["vpred"] = info["vpred"][0]
        a = action.cpu().numpy() if np_in else action
        return (a[0] if single else a), info


def load_zoo_policy(path, ac_dim, device=0, kind=None):
    """utils.py:66-67 ``load_params`` + policy construction; ``kind`` 'mlp' / 'lstm' (default: whichever layout fits the
    vector length)."""
    return load_zoo_policy_from_flat(np.load(path, allow_pickle=False), ac_dim, device=device, kind=kind)


def load_zoo_policy_from_flat(flat, ac_dim, device=0, kind=None):
    """The policy for a flat parameter vector already in memory (utils.py:70-83 ``set_from_flat``)."""
    if kind is None:
        try:
            infer_ob_dim(flat.size, ac_dim)
            kind = "mlp"
        except ValueError:
            kind = "lstm"
    if kind == "lstm":
        return ZooLSTMPolicy(flat, ac_dim, device=device)
    return ZooMLPPolicy(flat, ac_dim, device=device)


class FixedOpponentModel(object):
    """What alg_ppo.py:194-206 puts into ``runner.models[1]`` in ``opponent_mode='fix'``: a non-trainable model whose
    ``step`` / ``value`` / ``act_model.action_probability`` come from the zoo net."""

    trainable = False

    def __init__(self, policy):
        self.act_model = self.train_model = policy
        self.initial_state = None
        self.step = policy.step
        self.value = policy.value

    def load(self, path):
        raise RuntimeError("the fixed opponent is not replaced by checkpoints")


EVAL_ADJUST_Z = -0.5   # eval_robosumo_against_fix.py:108-115, play_fixed.py:23, compare_history_version.py:74


def evaluate_against(model, opponent, env, rounds, deterministic=True, adjust_z=EVAL_ADJUST_Z):
    """eval_robosumo_against_fix.py:196-230 on the device: ``model`` acts for agent 0 on obs[:, 0], ``opponent`` (zoo
    policy) for agent 1 on obs[:, 1, :ob_dim]; an episode counts as a win if agent 0 carries the 'winner' flag when it
    ends, a loss if agent 1 does, a draw otherwise.  Returns dict(win, draw, lose, rounds, steps).

    The reference's evaluator builds its envs with ``agent._adjust_z = -0.5`` on every agent (:108-115): observed heights
    and the lose test (sumo.py:147-160: ``z + adjust_z < 0.29``) are relative to a tatami surface at z = 0, which is what
    the zoo nets were trained on.  ``adjust_z`` is imposed on ``env`` for the evaluation and the env's own value restored
    afterwards (None: leave the env as it is)."""
    import torch
    prev_adjust = getattr(env, "adjust_z", 0.0)
    if adjust_z is not None and float(adjust_z) != prev_adjust:
        env.set_adjust_z(adjust_z)
    try:
        return _evaluate_against(model, opponent, env, rounds, deterministic)
    finally:
        if adjust_z is not None and float(adjust_z) != prev_adjust:
            torch.cuda.synchronize()
            env.set_adjust_z(prev_adjust)


def _evaluate_against(model, opponent, env, rounds, deterministic):
    import torch
    obs = env.reset_device()
    A0, A1 = env.model.act_dims
    D0 = env.model.obs_dims[0]
    acts = torch.zeros_like(env.act_dev)
    win = draw = lose = done_rounds = steps = 0
    while done_rounds < rounds:
        a0 = model.step(obs[:, 0, :D0], deterministic=deterministic)[0]
        a1 = opponent.act(obs[:, 1, :], stochastic=not deterministic)[0]
        acts[:, 0, :A0] = a0
        acts[:, 1, :A1] = a1
        obs, info, done, _, _, _ = env.step_device(acts)
        steps += 1
        fin = done[:, 0] != 0
        nfin = int(fin.sum())
        if nfin and getattr(opponent, "recurrent", False):
            opponent.reset(fin)
        if nfin:
            flags = info[:, :, 7].to(torch.int64)
            w0 = ((flags[:, 0] & 1) != 0) & fin
            w1 = ((flags[:, 1] & 1) != 0) & fin & ~w0
            nw, nl = int(w0.sum()), int(w1.sum())
            win += nw; lose += nl; draw += nfin - nw - nl
            done_rounds += nfin
    return dict(win=win / done_rounds, draw=draw / done_rounds, lose=lose / done_rounds, rounds=done_rounds, steps=steps)
