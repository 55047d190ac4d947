"""Fused recurrent rollout (sumo_rollout_steps_lstm: LSTM policies evaluated inside the env engine's persistent rollout launch)
against the launch-per-evaluation recurrent branch of the device-mode Runner (ppo_lstm_step / ppo_lstm_step_pool + sumo_step +
ppo_post_step per step).  Bit-exact: both paths run the same cell / head device functions, and the in-wave gate sums follow the
MFMA tiles' accumulation order."""
import ctypes as C

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import capi, lstm_model
    from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv

NAMES = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opp_neglogpacs", "opp_obs", "opp_actions", "states",
         "epinfos", "off_policy_ratio", "off_env_ratio", "total_ratio"]


def _models(N, T, H, pool_k, seed, D=121, A=8):
    spec = lstm_model.LstmSpec(D, A, H)
    np.random.seed(seed)
    learner = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=T, trainable=False)
    rng = np.random.default_rng(seed)
    base = learner.get_param_list()
    # livelier heads than the 0.01-scaled initialisation: actions large enough for falls inside a short rollout, non-zero biases / logstd
    grow = lambda pl, sc: [p + rng.normal(0, sc if p.ndim == 2 and p.shape[0] == H else 0.02, p.shape).astype(np.float32) for p in pl]
    learner.set_param_list(grow(base, 0.3))
    if pool_k:
        opp = LstmOpponentPool(spec, pool_k + 1, N, torch.device("cuda", 0))       # one slot stays empty
        for k in range(pool_k):
            opp.set_snapshot(k, grow(base, 0.3), label="v%d" % k)
        opp.assign(rng.integers(0, pool_k, N // 16))
    else:
        opp = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=T, trainable=False)
        opp.set_param_list(grow(base, 0.3))
    learner.seed(11); opp.seed(12)
    return learner, opp


def _run_pair(N, T, groups, pool_k, fused, H=128, chunk=0, seed=5, env_id="RoboSumo-Ant-vs-Ant-v0"):
    env = SumoVecEnv(env_id, num_envs=N, seed=21, groups=groups)
    learner, opp = _models(N, T, H, pool_k, seed, env.observation_space[0].shape[0], env.action_space[0].shape[0])
    r = Runner(env=env, models=[learner, opp], nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, anneal_bound=500)
    # (after the Runner's reset) a third of the envs continue close to the time limit: their episodes end -- done flags -> state
    # masks, auto-reset -- within the rollout
    for E in env.engines:
        qpos, qvel, warm, cnt = E.get_state()
        cnt[::3, 0] = 500 - 1 - (np.arange(len(cnt[::3])) % 7)
        E.set_state(qpos, qvel, warm, cnt)
    assert r.device_mode and r.recurrent
    r.fused_rollout = fused
    r.rollout_chunk = chunk
    assert r.fused_lstm_ok() == (fused and H == 128)
    outs = [r.run(250), r.run(251)]
    torch.cuda.synchronize()
    states = [s.clone() for s in r.states]
    st = [E.get_state() for E in env.engines]
    stats = env.stats()
    env.close()
    return outs, states, st, stats


def _assert_same(a, b):
    (fo, fS, fs, fstat), (so, sS, ss, sstat) = a, b
    for f, s_ in zip(fo, so):
        for k, (x, y) in enumerate(zip(f, s_)):
            if torch.is_tensor(x):
                assert torch.equal(x, y), (NAMES[k], (x != y).sum().item())
            else:
                assert x == y, NAMES[k]
    for x, y in zip(fS, sS):
        assert torch.equal(x, y)
    for (p, q) in zip(fs, ss):
        for x, y in zip(p, q):
            assert np.array_equal(x, y)
    for k in ("forward", "newton", "contacts", "efc", "dropped", "diverged", "rollout_aborts"):
        assert fstat[k] == sstat[k], k


@pytest.mark.parametrize("N,T,groups,pool_k,chunk", [(48, 14, 1, 0, 0), (64, 10, 2, 3, 0), (32, 9, 1, 2, 4),
                                                   (1024, 8, 2, 16, 0)])   # config 5's shard: envs migrate between waves and XCDs
def test_fused_recurrent_rollout_matches_stepwise_path(N, T, groups, pool_k, chunk):
    fused = _run_pair(N, T, groups, pool_k, True, chunk=chunk)
    step = _run_pair(N, T, groups, pool_k, False)
    _assert_same(fused, step)
    outs = fused[0]
    assert outs[0][2].any() and len(outs[0][11]) > 0          # episodes ended inside the launch: masked states + auto-reset covered
    assert torch.isfinite(outs[1][4]).all() and fused[3]["diverged"] == 0


@pytest.mark.parametrize("env_id", ["RoboSumo-Spider-vs-Spider-v0", "RoboSumo-Bug-vs-Bug-v0"])
def test_fused_recurrent_rollout_other_scenes(env_id):
    """The widest scene (Spider: 16 action dimensions = the whole head tile, one wave per SIMD kernel variant) and the Bug scene."""
    fused = _run_pair(32, 7, 1, 2, True, env_id=env_id)
    step = _run_pair(32, 7, 1, 2, False, env_id=env_id)
    _assert_same(fused, step)
    assert torch.isfinite(fused[0][1][4]).all() and fused[3]["diverged"] == 0


def test_fused_recurrent_rollout_falls_back_for_other_widths():
    """nlstm = 64 is outside what the in-wave evaluation is built for: the Runner keeps the launch-per-evaluation path (and the C ABI
    refuses such a net)."""
    outs, _, _, _ = _run_pair(32, 5, 1, 0, True, H=64)
    assert torch.isfinite(outs[0][4]).all()
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=2)
    m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(121, 8, 64), nbatch_act=16, nsteps=4, trainable=False)
    ro = capi.RolloutLstm()
    ro.learner = C.addressof(m._net)
    z = torch.zeros(16 * 4 * 2 * 121, dtype=torch.float32, device="cuda")
    for f in ("opponents_dev", "state0", "state1", "noise0", "noise1", "obs", "act", "rew", "val", "nlp", "onlp", "done", "ep_done", "ep_r", "ep_l"):
        setattr(ro, f, z.data_ptr())
    ro.npool, ro.T, ro.Ntot, ro.env_offset, ro.s0, ro.K, ro.alpha = 1, 4, 16, 0, 0, 4, 0.5
    with pytest.raises(capi.SumoHipError, match="hidden 128"):
        env.rollout_steps_lstm_group(0, ro)
    ro.K = 5
    with pytest.raises(capi.SumoHipError, match="outside the rollout buffers"):
        env.rollout_steps_lstm_group(0, ro)
    env.close()


def test_learn_lstm_uses_fused_rollout(tmp_path, monkeypatch):
    """learn(network='lstm') on the device: the rollouts go through the fused launch (counted), training stays finite."""
    from robosumo_selfplay_amd import alg_ppo
    calls = {"n": 0}
    orig = SumoVecEnv.rollout_steps_lstm_group

    def counted(self, g, ro):
        calls["n"] += 1
        return orig(self, g, ro)
    monkeypatch.setattr(SumoVecEnv, "rollout_steps_lstm_group", counted)
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=64, seed=8, groups=2)
    model = alg_ppo.learn(network="lstm", env=env, seed=3, total_timesteps=64 * 8 * 3, nagent=2, log_dir=str(tmp_path), verbose=False, nsteps=8,
                          nminibatches=4, noptepochs=1, lr=3e-4, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, opponent_mode="random",
                          nlstm=128, anneal_bound=1000, opponent_pool=4)
    assert calls["n"] == 3 * 2                                   # updates x env groups
    assert all(np.isfinite(l).all() for l in model.history["lossvals"]) and torch.isfinite(model.params).all()
    env.close()
