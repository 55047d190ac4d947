"""Recurrent PPO training on the GPU (robosumo_selfplay_amd/lstm_model.py) against the finite-difference-verified numpy BPTT in
oracle/ppo_oracle.py.  Tolerances: float32 MFMA kernels (forward, BPTT, the split-K weight-gradient kernel ppo_lstm_wgrad) vs float64
numpy -> 3e-4 of each tensor's gradient scale."""
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


@pytest.mark.parametrize("T,n,D,A,H", [(6, 37, 121, 8, 128), (4, 16, 13, 3, 64), (9, 130, 209, 16, 128)])
def test_hoisted_input_block_is_bit_identical(T, n, D, A, H):
    """The training forward with x * wx of all time steps computed up front (ppo_lstm_xproj + ppo_lstm_step_save_z) against the
    step-by-step forward from the observations (ppo_lstm_step_save): the gate sums continue the same accumulation, so gradients and
    loss sums agree bit for bit."""
    rng = np.random.default_rng(2)
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    old = rng.normal(8, 1, (n, T)).astype(np.float32)
    advs = rng.normal(0, 1, (n, T)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (n, T)).astype(np.float32)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    res = []
    for xproj in (True, False):
        np.random.seed(3)
        m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
        assert m.xproj
        m.xproj = xproj
        m.seq_kernels = False                    # step-by-step launches on both sides (the whole-sequence kernels have their own test)
        m.loss_and_grads(0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(advs), flat(old), flat(w), S0, T)
        torch.cuda.synchronize()
        res.append((m.grads.clone(), m.stats.clone()))
    P = res[0][0].numel() - res[0][1].numel()
    assert torch.equal(res[0][0][:P], res[1][0][:P]) and res[0][0][:P].abs().max() > 0
    # (the loss sums are float64 atomics over the rows: equal up to the order of the additions)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("T,n,D,A", [(6, 37, 121, 8), (33, 128, 121, 8), (5, 16, 13, 3)])
def test_sequence_kernels_match_step_kernels(T, n, D, A):
    """Whole-sequence forward / BPTT launches (ppo_lstm_seq_forward / _backward: recurrent weights resident in registers, one launch for
    all T steps) against the launch-per-step kernels: the forward is bit-identical (final state, loss sums up to the order of the
    float64 atomics); the backward sums dz * wh^T in one chain instead of four partial tiles -> gradients agree to 2e-6 of each
    tensor's scale."""
    H = 128
    rng = np.random.default_rng(5)
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    old = rng.normal(8, 1, (n, T)).astype(np.float32)
    advs = rng.normal(0, 1, (n, T)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (n, T)).astype(np.float32)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    res = []
    for seq in (True, False):
        np.random.seed(3)
        m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
        pl = [p + rng.normal(0, 0.05, p.shape).astype(np.float32) for p in m.get_param_list()] if seq else pl
        m.set_param_list(pl)
        assert m.seq_kernels and m.xproj
        m.seq_kernels = seq
        state = m.loss_and_grads(0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(advs), flat(old), flat(w), S0, T)
        torch.cuda.synchronize()
        res.append(([g.clone() for g in m.gviews], m.stats.clone(), state.clone()))
    assert torch.equal(res[0][2], res[1][2])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-12, atol=1e-12)
    for name, a, b in zip(lstm_model.policies.LSTM_PARAM_NAMES, res[0][0], res[1][0]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) < 2e-6 * scale, (name, float((a - b).abs().max()), scale)
        if name in ("pi/w", "pi/b", "pi/logstd", "vf/w", "vf/b"):                       # head gradients depend on the forward only
            assert torch.equal(a, b), name


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


def test_lstm_pool_step_equals_per_snapshot_steps():
    """ppo_lstm_step_pool (every 16-row tile evaluated with its own frozen snapshot, one launch) against ppo_lstm_step with each
    snapshot on that snapshot's tiles: actions, values, neglogps and new states bit-identical."""
    from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
    D, A, H, n, K = 121, 8, 128, 96, 4
    rng = np.random.default_rng(2)
    spec = lstm_model.LstmSpec(D, A, H)
    np.random.seed(9)
    singles = []
    pool = LstmOpponentPool(spec, K, n, torch.device("cuda", 0))
    for k in range(3):
        m = lstm_model.LstmPPOModel(policy=spec, nbatch_act=n, nsteps=4, trainable=False)
        m.set_param_list([p + rng.normal(0, 0.05, p.shape).astype(np.float32) for p in m.get_param_list()])
        singles.append(m)
        pool.set_snapshot(k, m.get_param_list(), label="v%d" % k)
    tiles = np.array([0, 2, 1, 1, 0, 2])
    pool.assign(tiles)
    assert np.array_equal(pool.counts(), [32, 32, 32, 0]) and np.array_equal(pool.index.cpu().numpy(), np.repeat(tiles, 16))
    obs = torch.from_numpy(rng.normal(0, 1, (n, D)).astype(np.float32)).cuda()
    S = torch.from_numpy(rng.normal(0, 0.5, (n, 2 * H)).astype(np.float32)).cuda()
    M = torch.from_numpy((rng.random(n) < 0.3).astype(np.uint8)).cuda()
    given = torch.from_numpy(rng.normal(0, 1, (n, A)).astype(np.float32)).cuda()
    # scoring a given action (zero state, as the Runner's call) and a deterministic step with carried state
    nlp_pool = pool.action_probability(obs, given_action=given)
    a_pool, v_pool, S_pool, n_pool = pool.step(obs, S=S, M=M, deterministic=True)
    # a group's slice (first_env) sees its own tiles
    a_half, _, _, _ = pool.step(obs[48:], S=S[48:], M=M[48:], deterministic=True, first_env=48)
    assert torch.equal(a_half, a_pool[48:])
    for k, m in enumerate(singles):
        rows = torch.from_numpy(np.nonzero(np.repeat(tiles, 16) == k)[0]).cuda()
        nlp_k = m.action_probability(obs[rows], given_action=given[rows])
        a_k, v_k, S_k, n_k = m.step(obs[rows], S=S[rows], M=M[rows], deterministic=True)
        assert torch.equal(nlp_pool[rows], nlp_k) and torch.equal(a_pool[rows], a_k) and torch.equal(v_pool[rows], v_k)
        assert torch.equal(S_pool[rows], S_k) and torch.equal(n_pool[rows], n_k)
    with pytest.raises(ValueError):
        pool.assign(np.full(6, 3))                         # slot 3 was never filled


def test_config5_shard_pool_rollout_and_bptt_update_vs_oracle(tmp_path):
    """BASELINE config 5 on its one-GPU shard: Ant-vs-Ant, 1024 envs, LSTM(128) learner, a pool of 16 frozen LSTM snapshots (one per
    16-env tile), rollout in the device-mode recurrent Runner, then ONE whole-sequence minibatch update (128 env sequences x 32
    steps) checked against the numpy BPTT + TF1-Adam restatement (oracle/ppo_oracle.py, finite-difference verified).
    Tolerances: loss terms 1e-3 relative, parameters after the step 3e-5 absolute (float32 MFMA kernels vs float64)."""
    from robosumo_selfplay_amd import alg_ppo
    from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
    from robosumo_selfplay_amd.runner import Runner
    N, T, H, K = 1024, 32, 128, 16
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=50, groups=2)
    spec = lstm_model.LstmSpec(121, 8, H)
    np.random.seed(4)
    learner = lstm_model.LstmPPOModel(policy=spec, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=N, nsteps=T)
    learner.seed(7)
    pool = LstmOpponentPool(spec, K, N, env.device)
    rng = np.random.default_rng(3)
    base = learner.get_param_list()
    for k in range(K):
        pool.set_snapshot(k, [p + rng.normal(0, 0.02, p.shape).astype(np.float32) for p in base], label="v%d" % k)
    pool.assign_round_robin()
    pool.seed(8)
    assert (pool.counts() == 64).all()
    r = Runner(env=env, models=[learner, pool], nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, anneal_bound=500)
    assert r.device_mode and r.recurrent
    obs, returns, masks, actions, values, nlp, rewards, onlp, _, _, states0, epinfos, opr, oer, tot = r.run(1)
    torch.cuda.synchronize()
    assert obs.shape == (2, N * T, 121) and torch.isfinite(returns).all() and torch.isfinite(nlp).all() and torch.isfinite(onlp).all()
    assert env.stats()["diverged"] == 0
    # envs of different tiles really faced different opponents: the opponent's neglogp of the learner's actions at step 0 uses the
    # tile's snapshot -> equals a single-snapshot evaluation only on that snapshot's tiles
    o0 = obs[0].reshape(N, T, 121)[:, 0].contiguous()
    a0 = actions[0].reshape(N, T, 8)[:, 0].contiguous()
    ref = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=T, trainable=False)
    ref.params.copy_(pool.params[3])
    same = (ref.action_probability(o0, given_action=a0) == onlp[0].reshape(N, T)[:, 0])
    assert same[pool.index == 3].all() and not same[pool.index != 3].all()
    # one minibatch = 128 whole env sequences (nenvs / nminibatches = 1024 / 8), as learn() forms them
    mbenv = np.sort(rng.choice(N, 128, replace=False))
    flatinds = torch.from_numpy((mbenv[:, None] * T + np.arange(T)[None, :]).ravel()).cuda()
    mb = lambda x: x[flatinds]
    S0 = states0[torch.from_numpy(mbenv).cuda()]
    pl = [p.copy() for p in learner.get_param_list()]
    w = torch.ones(128 * T, dtype=torch.float32, device="cuda")
    out = learner.train(3e-4, 0.2, mb(obs[0]), mb(returns[0]), mb(masks[0]), mb(actions[0]), mb(values[0]), mb(nlp[0]), None, w, S0, nsteps=T)
    torch.cuda.synchronize()
    cv = lambda x: x.cpu().numpy()
    tm = lambda x: np.swapaxes(cv(mb(x)).reshape(128, T, *x.shape[1:]), 0, 1)            # env-major rows -> [T, n, ...]
    advs = po.normalize_advantages(cv(mb(returns[0])), cv(mb(values[0]))).reshape(128, T)
    loss, stats, grads, _ = po.lstm_ppo_loss_and_grads(pl, tm(obs[0]), tm(masks[0]).astype(np.float32), tm(actions[0]), np.swapaxes(advs, 0, 1),
                                                       tm(returns[0]), tm(nlp[0]), np.ones((T, 128)), cv(S0), 0.2, 0.01, 0.5)
    got = np.array([float(x) for x in out[:5]])
    assert np.allclose(got[:2], stats[:2], rtol=1e-3, atol=1e-5), (got, stats)
    assert abs(got[3]) < 1e-4 and got[4] == 0.0                      # on-policy first step: ratio 1 (the stored neglogp is the acting net's own)
    gc, _ = po.clip_by_global_norm(grads, 0.5)
    newp, _, _ = po.adam_step([p.astype(np.float64) for p in pl], gc, [np.zeros_like(g) for g in gc], [np.zeros_like(g) for g in gc], 1, 3e-4)
    for name, a, b in zip(lstm_model.policies.LSTM_PARAM_NAMES, learner.get_param_list(), newp):
        assert np.abs(a - b.reshape(a.shape)).max() < 3e-5, (name, np.abs(a - b.reshape(a.shape)).max())
    env.close()
    # and the whole thing through learn(): two updates, the second one draws its 16 snapshots from the checkpoint directory
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=256, seed=51, groups=2)
    model = alg_ppo.learn(network="lstm", env=env, seed=4, total_timesteps=256 * 8 * 3, nagent=2, log_dir=str(tmp_path), verbose=False, nsteps=8,
                          nminibatches=4, noptepochs=1, lr=3e-4, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0, opponent_mode="random",
                          nlstm=128, anneal_bound=1000, opponent_pool=16)
    assert len(model.history["opponent_versions"]) == 3 and len(model.history["opponent_versions"][2]) == 16
    assert all(np.isfinite(l).all() for l in model.history["lossvals"]) and torch.isfinite(model.params).all()
    env.close()
    # the 'ours' selector (ratio-divergence sampling, alg_ppo.py:227-244) with a recurrent pool: its reference opponent is the
    # single-snapshot model, which holds checkpoint 00000 at update 2 (alg_ppo.py:208), not its own random initialisation
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=64, seed=52)
    seen = {}
    orig_sel = alg_ppo.selection_probs

    def spy(ap, naps):
        cv = lambda x: x.detach().cpu().numpy().copy() if torch.is_tensor(x) else np.asarray(x).copy()
        seen.setdefault("calls", []).append((cv(ap), [cv(x) for x in naps]))
        return orig_sel(ap, naps)
    alg_ppo.selection_probs = spy
    try:
        model = alg_ppo.learn(network="lstm", env=env, seed=5, total_timesteps=64 * 8 * 3, nagent=2, log_dir=str(tmp_path / "ours"), verbose=False,
                              nsteps=8, nminibatches=4, noptepochs=1, lr=3e-4, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0, opponent_mode="ours",
                              nlstm=128, anneal_bound=1000, opponent_pool=4)
    finally:
        alg_ppo.selection_probs = orig_sel
    assert len(seen["calls"]) == 2                                   # updates 2 and 3
    ap, naps = seen["calls"][0]
    # at update 2 the candidates are checkpoints 00000 and 00001; the reference opponent IS 00000: its likelihoods equal candidate 0's
    assert len(naps) == 2 and np.allclose(ap, naps[0], rtol=1e-5, atol=1e-5) and not np.allclose(ap, naps[1], rtol=1e-3, atol=1e-3)
    assert all(np.isfinite(l).all() for l in model.history["lossvals"])
    env.close()


@pytest.mark.parametrize("T,n,D,A,H", [(5, 33, 121, 8, 128), (9, 40, 209, 16, 64), (3, 7, 13, 3, 64)])
def test_native_weight_gradients_match_library_gemms(T, n, D, A, H):
    """ppo_lstm_wgrad (split-K MFMA kernel + fixed-order slab reduction) against the same products through hipBLASLt
    (SUMO_LSTM_WGRAD=blas path): every gradient tensor within 2e-5 of its scale (float32 summation order differs)."""
    rng = np.random.default_rng(21)
    obs, masks, actions, returns, values, S0 = _batch(rng, T, n, D, A, H)
    flat = lambda x: np.ascontiguousarray(x).reshape(n * T, *x.shape[2:])
    advs = rng.normal(0, 1, (n, T)).astype(np.float32)
    old = rng.normal(4.0, 0.3, (n, T)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, (n, T)).astype(np.float32)
    res = []
    for native in (True, False):
        np.random.seed(1)
        m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(D, A, H), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, nbatch_act=n, nsteps=T)
        m.wgrad_native = native
        m.loss_and_grads(0.2, flat(obs), flat(returns), flat(masks), flat(actions), flat(advs), flat(old), flat(w), S0, T)
        torch.cuda.synchronize()
        res.append([g.cpu().numpy().astype(np.float64) for g in m.gviews])
    for name, a, b in zip(lstm_model.policies.LSTM_PARAM_NAMES, *res):
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-5 * (np.abs(b).max() + 1e-6), (name, np.abs(a - b).max(), np.abs(b).max())
