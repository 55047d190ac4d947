"""Policy / value network on the HIP kernels, behind the reference's ``build_policy`` / ``PolicyWithValue`` names.

Reference: policies.py:14-193 (this fork's multi-agent variant: ``env.observation_space[0]``), network
baselines/baselines/common/models.py:74-103 (``mlp``, 2 x 64, relu per defaults.py:23), separate value trunk
(``value_network='copy'``, policies.py:173-180), diagonal Gaussian head with state-independent logstd
(baselines/baselines/common/distributions.py:96-113), orthogonal init (baselines/baselines/a2c/utils.py:20-35).
"""
import numpy as np

from . import ppo_capi

HIDDEN = 64
PARAM_NAMES = ["pi/mlp_fc0/w", "pi/mlp_fc0/b", "pi/mlp_fc1/w", "pi/mlp_fc1/b", "vf/mlp_fc0/w", "vf/mlp_fc0/b",
               "vf/mlp_fc1/w", "vf/mlp_fc1/b", "pi/w", "pi/b", "pi/logstd", "vf/w", "vf/b"]


def param_shapes(ob_dim, ac_dim):
    h = HIDDEN
    return [(ob_dim, h), (h,), (h, h), (h,), (ob_dim, h), (h,), (h, h), (h,), (h, ac_dim), (ac_dim,), (1, ac_dim), (h, 1), (1,)]


def ortho_init(shape, scale, rng):
    """a2c/utils.py:20-35: numpy normal -> SVD -> orthogonal factor * scale (float32)."""
    a = rng.normal(0.0, 1.0, shape)
    u, _, v = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == shape else v
    return (scale * q[:shape[0], :shape[1]]).astype(np.float32)


def init_param_list(ob_dim, ac_dim, rng=None):
    """Tensors in TF variable-creation order (policies.py:156-190,50,70).  ``rng`` defaults to numpy's GLOBAL RNG, which
    is what the reference consumes after ``set_global_seeds`` (misc_util.py:48-62)."""
    rng = rng or np.random
    h = HIDDEN
    out = []
    for _ in range(2):
        out += [ortho_init((ob_dim, h), np.sqrt(2), rng), np.zeros(h, np.float32), ortho_init((h, h), np.sqrt(2), rng),
                np.zeros(h, np.float32)]
    out += [ortho_init((h, ac_dim), 0.01, rng), np.zeros(ac_dim, np.float32), np.zeros((1, ac_dim), np.float32),
            ortho_init((h, 1), 1.0, rng), np.zeros(1, np.float32)]
    return out


def flatten_params(plist):
    return np.concatenate([np.asarray(p, np.float32).ravel() for p in plist])


def unflatten_params(flat, ob_dim, ac_dim):
    out, o = [], 0
    for s in param_shapes(ob_dim, ac_dim):
        n = int(np.prod(s))
        out.append(np.asarray(flat[o:o + n], np.float32).reshape(s).copy())
        o += n
    return out


class PolicySpec(object):
    """What ``build_policy`` returns here: the architecture description that PPOModel instantiates."""

    def __init__(self, ob_dim, ac_dim, network="mlp", value_network="copy", num_hidden=HIDDEN, num_layers=2, activation="relu"):
        act = getattr(activation, "__name__", activation)
        if network != "mlp" or value_network != "copy" or num_hidden != HIDDEN or num_layers != 2 or act != "relu":
            raise NotImplementedError("only network='mlp'(2x64, relu) with value_network='copy' is built so far "
                                      "(reference defaults.py:8-26); got %r/%r/%r/%r/%r"
                                      % (network, value_network, num_hidden, num_layers, act))
        self.ob_dim, self.ac_dim = int(ob_dim), int(ac_dim)


def build_policy(env, policy_network="mlp", value_network=None, normalize_observations=False, estimate_q=False,
                 **policy_kwargs):
    """policies.py:136-193.  ``value_network=None`` means a shared trunk in baselines; the reference's RoboSumo
    defaults always pass 'copy' (defaults.py:22)."""
    if normalize_observations or estimate_q:
        raise NotImplementedError("normalize_observations / estimate_q are off in the reference defaults")
    if policy_network == "lstm":                      # models.py:131-183; recurrent nets share the latent (policies.py:176-181)
        from .lstm_model import LstmSpec
        if value_network not in (None, "shared"):
            raise NotImplementedError("recurrent architectures are not supported with value_network=copy (reference policies.py:179)")
        return LstmSpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], nlstm=policy_kwargs.get("nlstm", 128))
    return PolicySpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], policy_network,
                      value_network if value_network is not None else "shared", **policy_kwargs)


class PolicyWithValue(object):
    """Device evaluation of one parameter vector.  numpy in -> numpy out (reference contract, policies.py:84-128);
    torch CUDA tensors in -> torch out (no host round trip)."""

    class _X:
        class dtype:
            name = "float32"

    X = _X()
    initial_state = None

    def __init__(self, spec, params, device):
        import torch
        self._t = torch
        self.spec, self.params, self.device = spec, params, device
        self.gen = torch.Generator(device=device)

    def seed(self, s):
        self.gen.manual_seed(int(s))

    def _prep(self, x, cols):
        t = self._t
        if isinstance(x, np.ndarray) or not t.is_tensor(x):
            return t.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.device).reshape(-1, cols), True
        if x.dtype != t.float32 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
            raise ValueError("expected a float32 CUDA matrix with unit inner stride")
        return x, False

    def evaluate(self, obs, flags, given_action=None, deterministic=False, out=None):
        """Returns dict(action, neglogp, value).  ``out`` may hold preallocated output tensors."""
        t = self._t
        D, A = self.spec.ob_dim, self.spec.ac_dim
        ob, _ = self._prep(obs, D)
        n = ob.shape[0]
        out = out or {}
        st = t.cuda.current_stream(self.device).cuda_stream
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
                if not given.is_contiguous():
                    given = given.contiguous()
            elif not deterministic:
                noise = t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32)
        if flags & ppo_capi.FWD_VF:
            value = out.get("value")
            if value is None:
                value = t.empty(n, dtype=t.float32, device=self.device)
        ppo_capi.chk(ppo_capi.lib().ppo_forward(self.params.data_ptr(), ob.data_ptr(), n, ob.stride(0), D, A, flags,
                                                ppo_capi.ptr(noise), ppo_capi.ptr(given), ppo_capi.ptr(action),
                                                ppo_capi.ptr(neglogp), ppo_capi.ptr(value), None, st))
        return dict(action=action, neglogp=neglogp, value=value)

    # ---- reference surface ------------------------------------------------------------------------------------
    def _ret(self, x, as_numpy):
        return x.cpu().numpy() if as_numpy else x

    def step(self, observation, deterministic=False, **extra_feed):
        np_in = isinstance(observation, np.ndarray)
        r = self.evaluate(observation, ppo_capi.FWD_PI | ppo_capi.FWD_VF, deterministic=deterministic)
        return self._ret(r["action"], np_in), self._ret(r["value"], np_in), None, self._ret(r["neglogp"], np_in)

    def value(self, ob, *args, **kwargs):
        return self._ret(self.evaluate(ob, ppo_capi.FWD_VF)["value"], isinstance(ob, np.ndarray))

    def action_probability(self, observation, given_action=None, **extra_feed):
        """policies.py:122-123: NEGATIVE log-probability of ``given_action``."""
        return self._ret(self.evaluate(observation, ppo_capi.FWD_PI, given_action=given_action)["neglogp"],
                         isinstance(observation, np.ndarray))

    def value_and_neglogp(self, observation, given_action=None, **extra_feed):
        r = self.evaluate(observation, ppo_capi.FWD_PI | ppo_capi.FWD_VF, given_action=given_action)
        np_in = isinstance(observation, np.ndarray)
        return self._ret(r["value"], np_in), self._ret(r["neglogp"], np_in)


# ---------------------------------------------------------------------------------------------------------------------
# recurrent policy: baselines ``lstm(nlstm=128)`` (baselines/baselines/common/models.py:131-183, a2c/utils.py:82-103) with
# the shared latent feeding both heads (policies.py:160-181 of the reference: ``value_network='copy'`` is not available
# for recurrent nets).  ACTING ONLY so far: the reference's recurrent training branch is dead code (it passes the states
# as IS_weight, alg_ppo.py:408-421), and the BPTT kernel is not built yet -- PPOModel refuses to train this policy.
# ---------------------------------------------------------------------------------------------------------------------
LSTM_PARAM_NAMES = ["pi/lstm/wx", "pi/lstm/wh", "pi/lstm/b", "pi/w", "pi/b", "pi/logstd", "vf/w", "vf/b"]


def lstm_param_shapes(ob_dim, ac_dim, nlstm=128):
    return [(ob_dim, 4 * nlstm), (nlstm, 4 * nlstm), (4 * nlstm,), (nlstm, ac_dim), (ac_dim,), (1, ac_dim), (nlstm, 1), (1,)]


def init_lstm_param_list(ob_dim, ac_dim, nlstm=128, rng=None):
    """a2c/utils.py:85-88 (ortho_init(1.0) for wx / wh, zero bias) + the heads of policies.py:50,70."""
    rng = rng or np.random
    return [ortho_init((ob_dim, 4 * nlstm), 1.0, rng), ortho_init((nlstm, 4 * nlstm), 1.0, rng), np.zeros(4 * nlstm, np.float32),
            ortho_init((nlstm, ac_dim), 0.01, rng), np.zeros(ac_dim, np.float32), np.zeros((1, ac_dim), np.float32),
            ortho_init((nlstm, 1), 1.0, rng), np.zeros(1, np.float32)]


class LstmPolicyWithValue(object):
    """Device evaluation of one baselines-LSTM parameter list.  ``step(obs, S=state, M=done)`` / ``value`` follow
    policies.py:84-128 with the extra feeds of models.py:163-170: state [n, 2*nlstm] = (c | h), M = done flags of the
    previous step (state rows are zeroed where M is set before the cell runs)."""

    def __init__(self, ob_dim, ac_dim, param_list, nlstm=128, device=0):
        import ctypes as C
        import torch
        self._t, self._C = torch, C
        self.device = torch.device("cuda", int(device))
        self.ob_dim, self.ac_dim, self.nlstm = int(ob_dim), int(ac_dim), int(nlstm)
        shapes = lstm_param_shapes(ob_dim, ac_dim, nlstm)
        assert len(param_list) == len(shapes) and all(tuple(np.shape(p)) == s for p, s in zip(param_list, shapes))
        self.tensors = [torch.from_numpy(np.ascontiguousarray(p, np.float32)).to(self.device) for p in param_list]
        wx, wh, b, pw, pb, logstd, vw, vb = self.tensors
        n = ppo_capi.LstmNet()
        n.ob_dim, n.emb_dim, n.hidden, n.ac_dim = self.ob_dim, 0, self.nlstm, self.ac_dim
        n.gate_order, n.forget_bias = ppo_capi.LSTM_GATES_IFOU, 0.0
        n.wx, n.wh, n.b = wx.data_ptr(), wh.data_ptr(), b.data_ptr()
        n.head_w, n.head_b, n.logstd, n.vf_w, n.vf_b = pw.data_ptr(), pb.data_ptr(), logstd.data_ptr(), vw.data_ptr(), vb.data_ptr()
        self._net = n
        self.gen = torch.Generator(device=self.device)

    def initial_state(self, nenv):
        return np.zeros((nenv, 2 * self.nlstm), np.float32)                      # models.py:176

    def seed(self, s):
        self.gen.manual_seed(int(s))

    def _run(self, obs, S, M, given_action=None, deterministic=False, keep_state=True):
        t = self._t
        np_in = isinstance(obs, np.ndarray)
        dev = lambda a, dt=np.float32: t.from_numpy(np.ascontiguousarray(a, dt)).to(self.device) if not t.is_tensor(a) else a
        x, st = dev(obs).reshape(-1, self.ob_dim), dev(S).clone() if not keep_state or not t.is_tensor(S) else S
        n, A, H = x.shape[0], self.ac_dim, self.nlstm
        mask = None if M is None else dev(np.asarray(M, np.float32) if not t.is_tensor(M) else M.to(t.float32))
        action = t.empty((n, A), dtype=t.float32, device=self.device)
        neglogp = t.empty(n, dtype=t.float32, device=self.device)
        value = t.empty(n, dtype=t.float32, device=self.device)
        given = None if given_action is None else dev(given_action).reshape(n, A).contiguous()
        noise = None if (deterministic or given is not None) else t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32)
        ppo_capi.chk(ppo_capi.lib().ppo_lstm_step(self._C.byref(self._net), x.data_ptr(), n, x.stride(0) if n > 1 else x.shape[1],
                                                  ppo_capi.ptr(mask), st.data_ptr(), st.data_ptr() + 4 * H, 2 * H, ppo_capi.ptr(noise),
                                                  ppo_capi.ptr(given), action.data_ptr(), neglogp.data_ptr(), value.data_ptr(), None,
                                                  t.cuda.current_stream(self.device).cuda_stream))
        out = lambda z: z.cpu().numpy() if np_in else z
        return out(action), out(value), out(st), out(neglogp)

    def step(self, observation, S=None, M=None, deterministic=False, **extra_feed):
        return self._run(observation, S, M, deterministic=deterministic)

    def value(self, ob, S=None, M=None, **kwargs):
        return self._run(ob, S, M, deterministic=True, keep_state=False)[1]

    def action_probability(self, observation, given_action=None, S=None, M=None, **extra_feed):
        return self._run(observation, S, M, given_action=given_action, keep_state=False)[3]
