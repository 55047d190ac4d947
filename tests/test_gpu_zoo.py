"""GPU tests of the policy-zoo opponents (robosumo_selfplay_amd/policy_zoo.py): the filtered tanh forward kernel against
the numpy restatement in oracle/ppo_oracle.py, the win/draw/lose evaluator, and ``opponent_mode='fix'``.
Tolerance: float32 MFMA + tanhf vs float32 numpy -> 2e-5 absolute on O(1) outputs."""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import policy_zoo, ppo_capi, policies, model as model_mod
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    from oracle import ppo_oracle as po


def _synthetic_flat(D, A, seed):
    """A zoo-shaped vector with a non-trivial observation filter (counts, sums) and O(1) weights."""
    rng = np.random.default_rng(seed)
    sh = policy_zoo.zoo_mlp_shapes(D, A)
    cnt = 1000.0
    parts = []
    for k in policy_zoo._ZOO_MLP_ORDER:
        s = sh[k]
        if k.endswith("/count"):
            v = np.array(cnt)
        elif k.endswith("/sum"):
            v = cnt * rng.normal(0, 0.5, s)
        elif k.endswith("/sumsq"):
            v = cnt * (0.25 + rng.uniform(0.0, 2.0, s))          # some variances below the 1e-2 floor after - mean^2
        elif k == "logstd":
            v = rng.normal(-1.0, 0.3, s)
        elif k.endswith("/w"):
            v = rng.normal(0, 1.0 / np.sqrt(s[0]), s)
        else:
            v = rng.normal(0, 0.1, s)
        parts.append(np.asarray(v, np.float32).ravel())
    return np.concatenate(parts)


@pytest.mark.parametrize("D,A,n,extra", [(120, 8, 300, 1), (208, 16, 70, 1), (164, 12, 16, 0)])
def test_zoo_forward_matches_oracle(D, A, n, extra):
    flat = _synthetic_flat(D, A, 3)
    pol = policy_zoo.ZooMLPPolicy(flat, A)
    _, p = policy_zoo.split_zoo_mlp(flat, A)
    rng = np.random.default_rng(1)
    obs = (rng.standard_normal((n, D + extra)) * 3.0).astype(np.float32)     # wide enough to hit the +-5 clip
    mean_o, v_o, logstd = po.zoo_mlp_forward(p, obs[:, :D])
    a, info = pol.act(obs, stochastic=False)
    assert a.shape == (n, A) and np.abs(a - mean_o).max() < 2e-5
    assert np.abs(info["vpred"] - v_o).max() < 2e-5 * (1 + np.abs(v_o).max())
    # neglogp of a given action under the diagonal Gaussian (policy.py:66: DiagonalGaussian(mean, logstd))
    given = (mean_o + rng.standard_normal((n, A)).astype(np.float32) * np.exp(logstd)).astype(np.float32)
    nlp = pol.action_probability(obs, given_action=given)
    ref = 0.5 * np.sum(((given - mean_o) / np.exp(logstd)) ** 2, axis=1) + 0.5 * np.log(2 * np.pi) * A + logstd.sum()
    assert np.abs(nlp - ref).max() < 1e-3 * (1 + np.abs(ref).max())
    # single observation, the reference's calling convention (policy.py:72-79)
    a1, i1 = pol.act(obs[0], stochastic=False)
    assert a1.shape == (A,) and np.allclose(a1, a[0], atol=1e-6) and np.isscalar(float(i1["vpred"]))
    # stochastic sampling: mean + std * N(0,1)
    pol.seed(5)
    s = np.stack([pol.act(obs[:4], stochastic=True)[0] for _ in range(200)])
    assert np.abs(s.mean(0) - mean_o[:4]).max() < 0.25 and np.abs(s.std(0) / np.exp(logstd) - 1).max() < 0.35


def test_evaluator_and_fixed_opponent_training(tmp_path):
    """eval_robosumo_against_fix.py loop + alg_ppo 'fix' mode with a synthetic zoo file (the shipped ones do not travel)."""
    from robosumo_selfplay_amd import alg_ppo
    D, A = 120, 8
    path = os.path.join(str(tmp_path), "agent-params-test.npy")
    np.save(path, _synthetic_flat(D, A, 7))
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=64, seed=3)
    opp = policy_zoo.load_zoo_policy(path, A)
    np.random.seed(0)
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    learner = model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False)
    r = policy_zoo.evaluate_against(learner, opp, env, rounds=64)
    assert r["rounds"] >= 64 and abs(r["win"] + r["draw"] + r["lose"] - 1.0) < 1e-12 and r["steps"] <= 501 * 3
    env.close()
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=1)
    model = alg_ppo.learn(network="mlp", env=env, seed=1, total_timesteps=16 * 16 * 2, nagent=2, log_dir=os.path.join(str(tmp_path), "log"),
                          verbose=False, nsteps=16, nminibatches=4, noptepochs=2, lr=1e-3, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0,
                          opponent_mode="fix", fix_opponent_path=path, value_network="copy", num_hidden=64, activation="relu",
                          anneal_bound=1000)
    assert len(model.history["lossvals"]) == 2 and all(np.isfinite(l).all() for l in model.history["lossvals"])
    assert torch.isfinite(model.params).all()
    env.close()


def _synthetic_lstm_flat(D, A, seed):
    rng = np.random.default_rng(seed)
    sh = policy_zoo.zoo_lstm_shapes(D, A)
    cnt = 500.0
    parts = []
    for k in policy_zoo._ZOO_LSTM_ORDER:
        s = sh[k]
        if k.endswith("/count"):
            v = np.array(cnt)
        elif k.endswith("/sum"):
            v = cnt * rng.normal(0, 0.5, s)
        elif k.endswith("/sumsq"):
            v = cnt * (0.25 + rng.uniform(0.0, 2.0, s))
        elif k == "logstd":
            v = rng.normal(-1.0, 0.3, s)
        elif k.endswith(("/w", "/kernel")):
            v = rng.normal(0, 1.0 / np.sqrt(s[0]), s)
        else:
            v = rng.normal(0, 0.1, s)
        parts.append(np.asarray(v, np.float32).ravel())
    return np.concatenate(parts)


@pytest.mark.parametrize("D,A,n", [(120, 8, 100), (208, 16, 33)])
def test_zoo_lstm_rollout_matches_oracle(D, A, n):
    """Six recurrent steps with an episode reset in between: actions (deterministic), values and the carried state."""
    flat = _synthetic_lstm_flat(D, A, 11)
    pol = policy_zoo.ZooLSTMPolicy(flat, A)
    _, p = policy_zoo.split_zoo_lstm(flat, A)
    rng = np.random.default_rng(2)
    state = np.zeros((4, n, 64), np.float32)
    for t in range(6):
        obs = (rng.standard_normal((n, D + 1)) * 2.0).astype(np.float32)
        if t == 3:
            fin = rng.random(n) < 0.4
            pol.reset(fin)
            state[:, fin, :] = 0
        mean_o, v_o, state = po.zoo_lstm_step(p, obs[:, :D], state)
        a, info = pol.act(obs, stochastic=False, want_value=True)
        assert np.abs(a - mean_o).max() < 5e-5, (t, np.abs(a - mean_o).max())
        assert np.abs(info["vpred"] - v_o).max() < 5e-5 * (1 + np.abs(v_o).max())
        assert np.abs(pol.state.cpu().numpy() - state).max() < 5e-5
    pol.reset()
    assert float(pol.state.abs().max()) == 0.0


@pytest.mark.parametrize("D,A,n,nlstm", [(121, 8, 70, 128), (30, 3, 16, 64)])
def test_baselines_lstm_step_matches_oracle(D, A, n, nlstm):
    """models.py:131-183 / a2c/utils.py:82-103: state (c | h), masks zero the state, shared latent feeds both heads."""
    rng = np.random.default_rng(4)
    pl = policies.init_lstm_param_list(D, A, nlstm, rng=np.random.RandomState(0))
    pl = [q + rng.normal(0, 0.05, q.shape).astype(np.float32) for q in pl]
    pol = policies.LstmPolicyWithValue(D, A, pl, nlstm=nlstm)
    wx, wh, b, pw, pb, logstd, vw, vb = pl
    S = pol.initial_state(n)
    So = S.copy()
    for t in range(5):
        obs = rng.standard_normal((n, D)).astype(np.float32)
        M = (rng.random(n) < 0.3).astype(np.float32) if t else np.zeros(n, np.float32)
        h, So2 = po.lstm_step_baselines(wx, wh, b, obs, So, M)
        mean_o, v_o = h @ pw + pb, (h @ vw + vb)[:, 0]
        v_only = pol.value(obs, S=S, M=M)
        a, v, S2, nlp = pol.step(obs, S=S, M=M, deterministic=True)
        assert np.abs(a - mean_o).max() < 5e-5 and np.abs(v - v_o).max() < 5e-5 and np.abs(v_only - v_o).max() < 5e-5
        assert np.abs(S2 - So2).max() < 5e-5
        ref = 0.5 * np.log(2 * np.pi) * A + logstd.sum()          # deterministic action = mean
        assert np.abs(nlp - ref).max() < 1e-4
        given = (mean_o + 0.3 * rng.standard_normal((n, A))).astype(np.float32)
        nlp_g = pol.action_probability(obs, given_action=given, S=S, M=M)
        ref_g = 0.5 * np.sum(((given - mean_o) / np.exp(logstd)) ** 2, axis=1) + ref
        assert np.abs(nlp_g - ref_g).max() < 1e-3 * (1 + np.abs(ref_g).max())
        S, So = S2, So2


def test_evaluator_with_recurrent_opponent(tmp_path):
    D, A = 120, 8
    path = os.path.join(str(tmp_path), "agent-lstm.npy")
    np.save(path, _synthetic_lstm_flat(D, A, 5))
    opp = policy_zoo.load_zoo_policy(path, A)
    assert isinstance(opp, policy_zoo.ZooLSTMPolicy)
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=32, seed=9)
    np.random.seed(0)
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    learner = model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False)
    r = policy_zoo.evaluate_against(learner, opp, env, rounds=32)
    assert r["rounds"] >= 32 and abs(r["win"] + r["draw"] + r["lose"] - 1.0) < 1e-12
    env.close()
