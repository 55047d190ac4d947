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
