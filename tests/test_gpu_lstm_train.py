"""Recurrent PPO training on the GPU (robosumo_selfplay_amd/lstm_model.py) against the finite-difference-verified numpy BPTT in
oracle/ppo_oracle.py.  Tolerances: float32 MFMA / hipBLASLt GEMMs vs float64 numpy -> 3e-4 of each tensor's gradient scale."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import lstm_model
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    from oracle import ppo_oracle as po


def _batch(rng, T, n, D, A, H):
    obs = rng.normal(0, 1, (n, T, D)).astype(np.float32)               # env-major, as the Runner's sf01 produces
    masks = (rng.random((n, T)) < 0.2)
    actions = rng.normal(0, 0.7, (n, T, A)).astype(np.float32)
    returns = rng.normal(0, 1, (n, T)).astype(np.float32)
    values = rng.normal(0, 1, (n, T)).astype(np.float32)
    S0 = rng.normal(0, 0.5, (n, 2 * H)).astype(np.float32)
    return obs, masks, actions, returns, values, S0


@pytest.mark.parametrize("T,n,D,A,H", [(7, 20, 13, 3, 64), (5, 33, 121, 8, 128)])
def test_bptt_gradients_match_oracle(T, n, D, A, H):
    rng = np.random.default_rng(0)
    np.random.seed(1)
    m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
    pl = [p + rng.normal(0, 0.05, p.shape).astype(np.float32) for p in m.get_param_list()]
    pl[5] = rng.normal(-0.3, 0.1, (1, A)).astype(np.float32)
    m.set_param_list(pl)
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    tm = lambda x: np.swapaxes(x, 0, 1)                                  # [n, T, ...] -> [T, n, ...]
    # old neglogp: current policy's, perturbed so that some ratios are clipped
    state = S0.copy()
    old = np.zeros((T, n), np.float32)
    for t in range(T):
        h, state = po.lstm_step_baselines(pl[0], pl[1], pl[2], tm(obs)[t], state, tm(masks)[t].astype(np.float32))
        old[t] = po.neglogp(h @ pl[3] + pl[4], pl[5], tm(actions)[t])
    old = (old + rng.normal(0, 0.25, (T, n))).astype(np.float32)
    advs = rng.normal(0, 1, (n, T)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (n, T)).astype(np.float32)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    m.loss_and_grads(0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(advs), flat(np.swapaxes(old, 0, 1)), flat(w), S0, T)
    torch.cuda.synchronize()
    loss, stats, grads, _ = po.lstm_ppo_loss_and_grads(pl, tm(obs), tm(masks), tm(actions), tm(advs), tm(returns), old, tm(w), S0, 0.2,
                                                       0.01, 0.5)
    s = m.stats.cpu().numpy()
    assert s[6] == T * n
    assert np.allclose([s[0] / s[6], s[1] / s[6], s[3] / s[6], s[4] / s[6]], [stats[0], stats[1], stats[3], stats[4]], rtol=2e-4, atol=2e-5)
    for name, gv, go in zip(lstm_model.policies.LSTM_PARAM_NAMES, m.gviews, grads):
        g = gv.cpu().numpy().astype(np.float64)
        scale = np.abs(go).max() + 1e-8
        assert np.abs(g - go.reshape(g.shape)).max() < 3e-4 * scale + 1e-7, (name, np.abs(g - go.reshape(g.shape)).max(), scale)


def test_train_step_matches_oracle_adam_and_reduces_loss():
    rng = np.random.default_rng(5)
    T, n, D, A, H = 8, 24, 19, 4, 64
    np.random.seed(2)
    m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
    pl = m.get_param_list()
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    tm = lambda x: np.swapaxes(x, 0, 1)
    # on-policy old neglogp
    state = S0.copy()
    old = np.zeros((T, n), np.float32)
    for t in range(T):
        h, state = po.lstm_step_baselines(pl[0], pl[1], pl[2], tm(obs)[t], state, tm(masks)[t].astype(np.float32))
        old[t] = po.neglogp(h @ pl[3] + pl[4], pl[5], tm(actions)[t])
    oldf = flat(np.swapaxes(old, 0, 1))
    w = np.ones(n * T, np.float32)
    out = m.train(1e-3, 0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(values), oldf, None, w, states=S0)
    advs = po.normalize_advantages(flat(returns), flat(values)).reshape(n, T)
    loss, stats, grads, _ = po.lstm_ppo_loss_and_grads(pl, tm(obs), tm(masks), tm(actions), tm(advs), tm(returns), old, np.ones((T, n)), S0,
                                                       0.2, 0.0, 0.5)
    assert np.allclose(out[:2], stats[:2], rtol=5e-4, atol=1e-5) and abs(out[3]) < 1e-4 and out[4] == 0.0
    gc, _ = po.clip_by_global_norm(grads, 0.5)
    newp, _, _ = po.adam_step([p.astype(np.float64) for p in pl], gc, [np.zeros_like(g) for g in gc], [np.zeros_like(g) for g in gc], 1, 1e-3)
    for a, b in zip(m.get_param_list(), newp):
        assert np.abs(a - b.reshape(a.shape)).max() < 2e-5
    # a few more steps on the same batch drive the value loss down
    v0 = out[1]
    for _ in range(30):
        out = m.train(3e-3, 0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(values), oldf, None, w, states=S0)
    assert out[1] < 0.8 * v0


def test_graph_replay_matches_eager_recurrent_steps():
    """The recurrent minibatch step replayed from a HIP graph (device tensors in, single GPU) leaves the same parameters
    and loss statistics as the eager launches, over several steps with changing inputs."""
    rng = np.random.default_rng(11)
    T, n, D, A, H = 6, 20, 23, 5, 64
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    dev = lambda x, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(x)).to("cuda", dt)
    base = [dev(flat(obs)), dev(flat(returns)), dev(flat(masks).astype(np.float32)), dev(flat(actions)), dev(flat(values))]
    old = dev(rng.normal(5.0, 0.1, n * T)); w = dev(np.ones(n * T)); S = dev(S0)
    res = []
    for use_graph in (True, False):
        np.random.seed(3)
        m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
        m.use_graph = use_graph
        outs = []
        for k in range(4):
            o = base[0] + 0.01 * k                    # new input values every step: the graph must read its copies, not stale data
            outs.append(m.train(1e-3, 0.2, o, base[1], base[2], base[3], base[4], old, None, w, states=S)[:5])
        assert (len(m._graphs) == 1) == use_graph
        res.append((m.params.clone(), np.asarray(outs, np.float64)))
    assert torch.equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])


def test_learn_with_lstm_policy_smoke(tmp_path):
    """Config-5-like plumbing on one GPU: Ant-vs-Ant, LSTM(128) learner and opponent, whole-sequence minibatches."""
    from robosumo_selfplay_amd import alg_ppo
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=8, seed=4)
    model = alg_ppo.learn(network="lstm", env=env, seed=4, total_timesteps=8 * 16 * 2, nagent=2, log_dir=str(tmp_path), verbose=False,
                          nsteps=16, nminibatches=4, noptepochs=2, lr=1e-3, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0,
                          opponent_mode="latest", nlstm=128, anneal_bound=1000)
    assert isinstance(model, lstm_model.LstmPPOModel) and model.t == 2 * 2 * 4
    assert len(model.history["lossvals"]) == 2 and all(np.isfinite(l).all() for l in model.history["lossvals"])
    assert torch.isfinite(model.params).all()
    assert len(os.listdir(os.path.join(str(tmp_path), "checkpoints"))) == 3
    env.close()


def test_device_and_host_recurrent_rollouts_agree():
    """The device-mode recurrent Runner (states, masks and buffers stay in HBM) reproduces the host-mode one, which follows the
    reference Runner line by line (runner.py:62-151 with the S / M feeds)."""
    from robosumo_selfplay_amd.runner import Runner

    class HostOnly(object):                      # hides step_device so that the Runner takes the reference (numpy) path
        def __init__(self, env):
            self._e = env
            self.num_envs, self.observation_space, self.action_space = env.num_envs, env.observation_space, env.action_space
        def reset(self):
            return self._e.reset()
        def step(self, a):
            return self._e.step(a)

    outs = []
    for host in (False, True):
        env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=8, seed=12)
        np.random.seed(6)
        spec = lstm_model.LstmSpec(121, 8, 64)
        ms = [lstm_model.LstmPPOModel(policy=spec, nbatch_act=8, nsteps=6, trainable=(i == 0)) for i in range(2)]
        ms[1].set_param_list([p + 0.01 for p in ms[0].get_param_list()])
        for i, m in enumerate(ms):
            m.seed(100 + i)
        r = Runner(env=HostOnly(env) if host else env, models=ms, nsteps=6, nagent=2, gamma=0.99, lam=0.95, rho_bar=1.0, c_bar=1.0)
        assert r.device_mode == (not host)
        out = r.run(1)
        cv = lambda x: x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
        outs.append([cv(out[k]) for k in (0, 1, 3, 4, 5, 7, 10)])
        env.close()
    for a, b in zip(*outs):
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-5, atol=1e-6), np.abs(a - b).max()
