"""Deterministic duck-typed stand-ins for the env and models that the reference's Runner drives (runner.py:7-252).
Written for this repo (not taken from the reference): used by tests/golden/make_runner_golden.py to drive the
REFERENCE Runner when generating golden vectors, and by the tests to drive OUR Runner / oracle on the same streams."""
import numpy as np


class _Space:
    def __init__(self, shape):
        self.shape = shape


class _SpaceTuple:
    def __init__(self, spaces):
        self.spaces = spaces

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class FakeEnv:
    """Vector env with pre-drawn streams. done pattern: Bernoulli(p_done) per env (both agents equal)."""

    def __init__(self, nenv, ob_dim, ac_dim, seed, p_done=0.1, with_info=True):
        self.num_envs = nenv
        self.observation_space = _SpaceTuple([_Space((ob_dim,)), _Space((ob_dim,))])
        self.action_space = _SpaceTuple([_Space((ac_dim,)), _Space((ac_dim,))])
        self.rng = np.random.RandomState(seed)
        self.ob_dim, self.ac_dim, self.p_done, self.with_info = ob_dim, ac_dim, p_done, with_info
        self.t = 0
        self.action_log = []

    def reset(self):
        return self.rng.standard_normal((self.num_envs, 2, self.ob_dim)).astype(np.float32)

    def step(self, actions):
        self.action_log.append(np.asarray(actions).copy())
        n = self.num_envs
        obs = self.rng.standard_normal((n, 2, self.ob_dim)).astype(np.float32)
        d = self.rng.uniform(size=n) < self.p_done
        dones = np.stack([d, d], axis=1)
        rews = self.rng.standard_normal((n, 2))
        shaping = self.rng.standard_normal((n, 2)) * 3.0
        main = np.where(dones, self.rng.choice([-2000.0, 2000.0, -1000.0], size=(n, 2)), 0.0)
        infos = []
        for e in range(n):
            per = []
            for a in range(2):
                info = {}
                if self.with_info:
                    info["shaping_reward"] = float(shaping[e, a])
                    info["main_reward"] = float(main[e, a])
                per.append(info)
            if d[e]:
                per[0]["episode"] = {"r": float(self.t + e), "l": int(self.t + 1), "t": 0.5}
            infos.append(tuple(per))
        self.t += 1
        return obs, rews, dones, tuple(infos)


class _X:
    class dtype:
        name = "float32"


class _TrainModel:
    X = _X()


class _ActModel:
    def __init__(self, owner):
        self.owner = owner

    def action_probability(self, observation, given_action=None, **kw):
        return self.owner._neglogp(observation, given_action)


class FakeModel:
    """Linear-Gaussian 'policy': mean = obs @ W, value = obs @ v, unit-ish std; float32 like the TF model."""
    initial_state = None

    def __init__(self, ob_dim, ac_dim, seed):
        r = np.random.RandomState(seed)
        self.W = (r.standard_normal((ob_dim, ac_dim)) * 0.3).astype(np.float32)
        self.v = (r.standard_normal(ob_dim) * 0.5).astype(np.float32)
        self.logstd = (r.standard_normal(ac_dim) * 0.2).astype(np.float32)
        self.noise = np.random.RandomState(seed + 1000)
        self.train_model = _TrainModel()
        self.act_model = _ActModel(self)

    def _neglogp(self, obs, a):
        obs = np.asarray(obs, np.float32)
        mean = obs @ self.W
        std = np.exp(self.logstd)
        return (0.5 * np.sum(np.square((a - mean) / std), axis=-1) + 0.5 * np.log(2.0 * np.pi) * a.shape[-1]
                + np.sum(self.logstd)).astype(np.float32)

    def step(self, obs, S=None, M=None, **kw):
        obs = np.asarray(obs, np.float32)
        mean = obs @ self.W
        a = (mean + np.exp(self.logstd) * self.noise.standard_normal(mean.shape).astype(np.float32)).astype(np.float32)
        return a, self.value(obs), None, self._neglogp(obs, a)

    def value(self, obs, S=None, M=None, **kw):
        return (np.asarray(obs, np.float32) @ self.v).astype(np.float32)


CASES = [
    # name, nenv, nsteps, ob_dim, ac_dim, update, anneal_bound, gamma, lam, rho_bar, c_bar, p_done, with_info
    ("basic", 3, 4, 5, 2, 1, 500, 0.995, 1.0, 10.0, 1.0, 0.2, True),
    ("mid_anneal", 5, 16, 7, 3, 250, 500, 0.99, 0.95, 1.0, 1.0, 0.1, True),
    ("past_anneal", 4, 9, 6, 2, 777, 500, 0.995, 1.0, 10.0, 1.0, 0.15, True),
    ("no_done", 2, 6, 4, 2, 3, 10, 0.9, 0.8, 2.0, 0.5, 0.0, True),
    ("all_done", 3, 5, 4, 1, 2, 1000, 0.995, 1.0, 10.0, 1.0, 1.0, True),
    ("env_rewards", 4, 8, 5, 2, 1, 500, 0.995, 0.9, 10.0, 1.0, 0.2, False),   # infos lack shaping_reward (runner.py:144)
    ("single_step", 6, 1, 3, 2, 5, 500, 0.995, 1.0, 10.0, 1.0, 0.3, True),
]


def make_case(c):
    name, nenv, nsteps, ob, ac, update, anneal, gamma, lam, rho, cb, pd, wi = c
    env = FakeEnv(nenv, ob, ac, seed=hash(name) % 1000 if False else sum(map(ord, name)), p_done=pd, with_info=wi)
    models = [FakeModel(ob, ac, seed=11 + sum(map(ord, name))), FakeModel(ob, ac, seed=23 + sum(map(ord, name)))]
    kw = dict(nsteps=nsteps, nagent=2, gamma=gamma, lam=lam, rho_bar=rho, c_bar=cb, anneal_bound=anneal)
    return env, models, kw, update
