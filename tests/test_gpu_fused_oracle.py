"""The bench default tested DIRECTLY against the oracle: one fused rollout launch (sumo_rollout_steps / sumo_rollout_steps_lstm:
policies, env steps, auto-resets and buffer appends of K steps of every env in ONE launch, envs migrating between waves) against
oracle.OracleSim replaying the actions the launch recorded, from the same start state -- not transitively through the
step-by-step kernels.  And the launch's failure path: a hand-over whose checksum does not match must raise out of Runner.run.

Tolerances.  Both sides integrate the same float64 recurrences with different summation orders (1e-9 per env step, see
test_gpu_env_parity.py) and the contact dynamics amplify differences along a free-running trajectory, so the comparison is
per step with a growing allowance: 1e-6 for the first 4 steps, 1e-4 (north_star's tolerance) to the end of the launch; done
flags, episode lengths and reset counters bit-exact.  The MLP policy phase is checked against the numpy restatement of the nets
(oracle/ppo_oracle.py) on the recorded observations: value and both likelihoods of the recorded actions to 2e-4."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import capi, lstm_model, policies
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    from oracle import ppo_oracle as po
    from oracle.oracle import OracleSim


def _stagger(env, rng):
    """After the Runner's reset: a third of the envs close to the time limit (timeouts -> auto-reset inside the launch), two
    envs with an agent outside the ring (win / lose at the first step)."""
    for E in env.engines:
        qpos, qvel, warm, cnt = E.get_state()
        cnt[::3, 0] = 500 - 1 - (np.arange(len(cnt[::3])) % 5)
        aq = [int(x) for x in env.model.agent_qposadr]
        qpos[1, aq[0]] = 2.6
        qpos[2, aq[1] + 1] = -2.7
        E.set_state(qpos, qvel, warm, cnt)


def _replay_and_compare(env, start, out, T, alpha, adjust_z=0.0):
    """Oracle replay of the recorded actions; returns the per-step max observation error."""
    m, N = env.model, env.num_envs
    D, A = m.obs_dims[0], m.act_dims[0]
    ora = OracleSim(m, N, maxcon=env.engine.maxcon, jbcap=env.engine.jbcap)
    ora.set_adjust_z(adjust_z)
    ora.set_seeds(env.seeds)
    ora.set_state(*start)
    obs_rec = out[0].reshape(2, N, T, D).cpu().numpy()            # observation BEFORE each step (runner.py:98-101), env-major rows
    act_rec = out[3].reshape(2, N, T, A).cpu().numpy()
    rew_rec = out[6].reshape(2, N, T).cpu().numpy()
    msk_rec = out[2].reshape(2, N, T).cpu().numpy()
    prev_done = np.zeros((N, 2), np.uint8)
    errs, ep_seen = [], []
    oobs = None
    for t in range(T):
        if oobs is not None:
            e = np.abs(obs_rec[:, :, t].transpose(1, 0, 2) - oobs[:, :, :D]).max()
            errs.append(float(e))
            assert np.array_equal(msk_rec[:, :, t].T.astype(np.uint8), prev_done), t       # mb_dones: the flags before the step
        a = np.zeros((N, 2, ora.act_stride), np.float32)
        a[:, :, :A] = act_rec[:, :, t].transpose(1, 0, 2)
        oobs, oinfo, odone, oer, oedr, oel = ora.step(a, nthreads=8)
        want = alpha * oinfo[:, :, 6] + (1.0 - alpha) * oinfo[:, :, 3]                       # runner.py:130-134
        got = rew_rec[:, :, t].T
        assert np.abs(got - want).max() <= 1e-4 * (1.0 + np.abs(want).max()), t
        prev_done = odone
        ep_seen.extend((float(oer[e_]), int(oel[e_])) for e_ in range(N) if odone[e_, 0])
    final = env.obs_dev.cpu().numpy()
    errs.append(float(np.abs(final - oobs).max()))
    assert np.array_equal(env.done_dev.cpu().numpy(), odone)
    gs, os_ = env.engine.get_state(), ora.get_state()
    assert np.array_equal(gs[3], os_[3])                                                    # num_steps, reset_count of every env
    # episode records harvested by the launch (monitor.py:63-78): same episodes, same lengths, returns to 1e-6 relative
    eps = sorted((int(e["l"]), float(e["r"])) for e in out[11])
    ref = sorted((l, r) for r, l in ep_seen)
    assert [l for l, _ in eps] == [l for l, _ in ref]
    assert np.allclose([r for _, r in eps], [r for _, r in ref], rtol=1e-6, atol=2e-6)      # EpInfoList rounds 'r' to 6 decimals like monitor.py:66
    return errs, len(eps)


@pytest.mark.parametrize("env_id,N,T,adjust_z", [("RoboSumo-Ant-vs-Ant-v0", 64, 24, 0.0), ("RoboSumo-Spider-vs-Spider-v0", 32, 12, 0.0),
                                                 ("RoboSumo-Ant-vs-Ant-v0", 32, 10, -0.5)])
def test_fused_rollout_matches_oracle(env_id, N, T, adjust_z):
    env = SumoVecEnv(env_id, num_envs=N, seed=5, adjust_z=adjust_z)
    D, A = env.observation_space[0].shape[0], env.action_space[0].shape[0]
    spec = policies.PolicySpec(D, A, value_network="copy", activation="relu")
    np.random.seed(3)
    models = [PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    rng = np.random.RandomState(1)
    plists = []
    for m_ in models:     # livelier than the 0.01-scaled initial heads
        pl = [p + rng.normal(0, 0.1, p.shape).astype(np.float32) for p in m_.get_param_list()]
        m_.set_param_list(pl)
        plists.append(pl)
    models[0].act_model.seed(7); models[1].act_model.seed(8)
    r = Runner(env=env, models=models, nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, anneal_bound=500)
    assert r.fused_ok()
    _stagger(env, rng)
    start = env.engine.get_state()
    update = 250
    out = r.run(update)
    torch.cuda.synchronize()
    alpha = float(np.linspace(1, 0, 500)[update - 1])
    errs, neps = _replay_and_compare(env, start, out, T, alpha, adjust_z)
    assert neps >= N // 3                                               # timeouts, the two ring-outs: auto-resets inside the launch
    assert max(errs[:4]) < 1e-6 and max(errs) < 1e-4, errs
    st = env.stats()
    assert st["rollout_aborts"] == 0 and st["handover_mismatches"] == 0 and st["diverged"] == 0
    assert env.engine.rollout_status()["aborted"] == 0
    # the policy phase against the numpy nets, on the recorded observations of every step
    obs = out[0].cpu().numpy().astype(np.float64)                       # [2, N*T, D]
    act, val, nlp, onlp = [out[k].cpu().numpy() for k in (3, 4, 5, 7)]
    for g in range(2):
        mean_l, v_l, _ = po.forward(plists[0], obs[g])
        mean_o, _, _ = po.forward(plists[1], obs[g])
        ls_l, ls_o = plists[0][10].astype(np.float64), plists[1][10].astype(np.float64)
        a = act[g].astype(np.float64)
        assert np.abs(val[g] - v_l).max() < 2e-4 * (1 + np.abs(v_l).max())                   # models[0].value on both sides (runner.py:69,89)
        assert np.abs(nlp[g] - po.neglogp(mean_l, ls_l, a)).max() < 2e-4 * (1 + np.abs(nlp[g]).max())    # learner's likelihood
        assert np.abs(onlp[g] - po.neglogp(mean_o, ls_o, a)).max() < 2e-4 * (1 + np.abs(onlp[g]).max())  # opponent's likelihood
    env.close()


def test_fused_recurrent_rollout_matches_oracle():
    """LSTM(128) policies against a pool of 3 snapshots (one per 16-env tile): the env side of sumo_rollout_steps_lstm."""
    N, T, H = 64, 12, 128
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=21)
    spec = lstm_model.LstmSpec(121, 8, H)
    np.random.seed(5)
    learner = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=T, trainable=False)
    rng = np.random.default_rng(5)
    base = learner.get_param_list()
    grow = lambda pl, sc: [p + rng.normal(0, sc if p.ndim == 2 and p.shape[0] == H else 0.02, p.shape).astype(np.float32) for p in pl]
    learner.set_param_list(grow(base, 0.3))
    pool = LstmOpponentPool(spec, 4, N, torch.device("cuda", 0))
    for k in range(3):
        pool.set_snapshot(k, grow(base, 0.3), label="v%d" % k)
    pool.assign(rng.integers(0, 3, N // 16))
    learner.seed(11); pool.seed(12)
    r = Runner(env=env, models=[learner, pool], nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, anneal_bound=500)
    assert r.fused_lstm_ok()
    _stagger(env, np.random.RandomState(2))
    start = env.engine.get_state()
    out = r.run(250)
    torch.cuda.synchronize()
    errs, neps = _replay_and_compare(env, start, out, T, float(np.linspace(1, 0, 500)[249]))
    assert neps >= N // 3 and max(errs[:4]) < 1e-6 and max(errs) < 1e-4, errs
    assert env.stats()["rollout_aborts"] == 0 and env.stats()["handover_mismatches"] == 0
    env.close()


def test_cut_short_launch_raises_out_of_runner():
    """Fault injection (sumo_debug_fault): env 5's first hand-over carries a wrong checksum.  The wave that takes the env over
    notices, the launch drains, Runner.run raises (the reference raises MujocoException out of env.step on a MuJoCo fault,
    mujoco-py/mujoco_py/builder.py:351-369) -- it does not hand unwritten rollout rows to the optimiser."""
    N, T = 32, 6
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=1)
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    models = [PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    r = Runner(env=env, models=models, nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    r.run(1)                                                            # a clean launch first
    assert env.engine.rollout_status() == dict(aborted=0, tickets_drawn=env.engine.rollout_status()["tickets_drawn"], tickets=N * T, mismatches=0)
    env.engine.debug_fault(5)
    with pytest.raises(capi.SumoHipError, match="cut short"):
        r.run(2)
    st = env.stats()
    assert st["handover_mismatches"] == 1
    env.engine.debug_fault(-1)
    env._needs_seed = True
    env.reset_device()                                                  # the documented recovery: reset, then go on (r.obs IS env.obs_dev)
    env.done_dev.zero_()
    out = r.run(3)
    torch.cuda.synchronize()
    assert torch.isfinite(out[1]).all() and env.engine.rollout_status()["aborted"] == 0
    env.close()
