"""GPU parity tests for the PPO2 rollout/update kernels (include/sumo_ppo.h) against oracle/ppo_oracle.py and the golden
vectors produced by the reference Runner.  Float tolerances (stated per test): float32 MFMA vs float64 numpy."""
import os

import numpy as np
import pytest

from conftest import ROOT, has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import model as model_mod, policies, ppo_capi
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    from oracle import ppo_oracle as po
    from fake_rollout import CASES, make_case

    DEV = torch.device("cuda:0")


def _model(ob, ac, seed=0, trainable=True, **kw):
    np.random.seed(seed)
    spec = policies.PolicySpec(ob, ac, value_network="copy", activation="relu")
    return model_mod.PPOModel(policy=spec, ent_coef=kw.pop("ent_coef", 0.0), vf_coef=0.5, max_grad_norm=kw.pop("max_grad_norm", 0.5),
                              trainable=trainable, **kw)


def _perturb(m, rng, scale=0.1):
    pl = [p + rng.normal(0, scale, p.shape).astype(np.float32) for p in m.get_param_list()]
    m.set_param_list(pl)
    return pl


@pytest.mark.parametrize("ob,ac,n", [(121, 8, 300), (209, 16, 100), (165, 12, 17), (5, 1, 16)])
def test_forward_matches_oracle(ob, ac, n):
    """tolerance 2e-5 relative (f32 MFMA chain vs f64)."""
    rng = np.random.RandomState(0)
    m = _model(ob, ac, trainable=False)
    pl = _perturb(m, rng)
    obs = rng.normal(0, 1, (n, ob)).astype(np.float32)
    given = rng.normal(0, 1, (n, ac)).astype(np.float32)
    mean, value, _ = po.forward(pl, obs)
    logstd = pl[10].astype(np.float64)
    a_det, v, _, nlp_det = m.step(obs, deterministic=True)
    assert a_det.shape == (n, ac) and v.shape == (n,) and a_det.dtype == np.float32
    assert np.allclose(a_det, mean, rtol=2e-5, atol=2e-5) and np.allclose(v, value, rtol=2e-5, atol=2e-5)
    assert np.allclose(nlp_det, po.neglogp(mean, logstd, a_det.astype(np.float64)), rtol=2e-5, atol=2e-5)
    assert np.allclose(m.value(obs), value, rtol=2e-5, atol=2e-5)
    nlp = m.act_model.action_probability(obs, given_action=given)
    assert np.allclose(nlp, po.neglogp(mean, logstd, given), rtol=2e-5, atol=1e-4)
    # stochastic step: action = mean + std * noise with the model's generator; neglogp consistent with that action
    m.act_model.seed(5)
    a, v2, s, nl = m.step(obs)
    assert s is None and np.allclose(v2, value, rtol=2e-5, atol=2e-5)
    assert np.allclose(nl, po.neglogp(mean, logstd, a.astype(np.float64)), rtol=2e-5, atol=1e-4)
    z = (a - mean) / np.exp(logstd)
    assert abs(z.mean()) < 0.2 and 0.8 < z.std() < 1.2
    m.act_model.seed(5)
    assert np.array_equal(m.step(obs)[0], a)


@pytest.mark.parametrize("case", CASES if has_gpu() else [], ids=[c[0] for c in CASES] if has_gpu() else [])
def test_runner_host_mode_matches_reference_golden(case):
    """Our Runner (reward mix + IS ratios + V-trace on the GPU) driven by the same fake env/models as the reference
    Runner.  Everything that does not pass through exp() is bit-exact; ratios / agent-1 returns to 2e-6 relative
    (device expf vs numpy exp)."""
    gold = np.load(os.path.join(ROOT, "tests", "golden", "runner_%s.npz" % case[0]))
    env, models, kw, update = make_case(case)
    r = Runner(env=env, models=models, **kw)
    names = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opponent_neglogpacs", "opponent_obs",
             "opponent_actions", "states", "epinfos", "off_policy_ratio", "off_env_ratio", "ratio"]
    for call in range(2):
        res = r.run(update + call)
        assert len(res) == 15 and res[10] is None
        for nm, v in zip(names, res):
            if nm == "states":
                continue
            if nm == "epinfos":
                assert [e["r"] for e in v] == gold["c%d_epinfo_r" % call].tolist()
                continue
            g = gold["c%d_%s" % (call, nm)]
            v = np.asarray(v)
            assert v.shape == g.shape and v.dtype == g.dtype, (nm, v.shape, g.shape, v.dtype, g.dtype)
            if nm in ("off_policy_ratio", "off_env_ratio", "ratio"):
                assert np.allclose(v, g, rtol=2e-6, atol=0), nm
            elif nm == "returns":
                assert np.array_equal(v[0], g[0]), "agent 0 returns must be bit-exact"
                assert np.allclose(v[1], g[1], rtol=1e-5, atol=1e-4), np.abs(v[1] - g[1]).max()
            else:
                assert np.array_equal(v, g), nm


def test_vtrace_kernel_vs_oracle_large():
    rng = np.random.RandomState(0)
    T, N = 64, 777
    rew = rng.normal(0, 3, (2, T, N)).astype(np.float32)
    val = rng.normal(0, 5, (2, T, N)).astype(np.float32)
    nlp = rng.normal(10, 1, (2, T, N)).astype(np.float32)
    onlp = (nlp + rng.normal(0, 0.5, (2, T, N))).astype(np.float32)
    dones = rng.uniform(size=(2, T, N)) < 0.05
    dones[1] = dones[0]
    last_d = rng.uniform(size=(N, 2)) < 0.05
    last_v = rng.normal(0, 5, (2, N)).astype(np.float32)
    up = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x.astype(dt))).to(DEV)
    ret = torch.empty((2, T, N), dtype=torch.float32, device=DEV)
    r1, r2, r3 = (torch.empty((T, N), dtype=torch.float32, device=DEV) for _ in range(3))
    keep = [up(rew, np.float32), up(val, np.float32), up(nlp, np.float32), up(onlp, np.float32), up(dones, np.uint8),
            up(last_d, np.uint8), up(last_v, np.float32)]      # hold references: data_ptr() of a temporary dangles
    ppo_capi.chk(ppo_capi.lib().ppo_vtrace(*[k.data_ptr() for k in keep], T, N, 0.995, 0.95, 10.0, 1.0, ret.data_ptr(),
                                           r1.data_ptr(), r2.data_ptr(), r3.data_ptr(), None))
    torch.cuda.synchronize()
    opr = np.exp(onlp[1] - nlp[1]); oer = np.exp(nlp[0] - onlp[0]); ratio = opr * oer
    ones = np.ones_like(ratio)
    e0 = po.vtrace_returns(rew[0], val[0], dones[0], last_d[:, 0], last_v[0], ones, ones * np.float32(0.95), 0.995)
    e1 = po.vtrace_returns(rew[1], val[1], dones[1], last_d[:, 1], last_v[1], np.clip(ratio, None, 10.0),
                           np.clip(ratio, None, 1.0) * np.float32(0.95), 0.995)
    g = ret.cpu().numpy()
    assert np.array_equal(g[0], e0)
    assert np.allclose(g[1], e1, rtol=1e-5, atol=1e-4) and np.allclose(r3.cpu().numpy(), ratio, rtol=2e-6)


@pytest.mark.parametrize("ob,ac,n,use_idx", [(121, 8, 1000, True), (209, 16, 160, False), (30, 3, 37, True)])
def test_gradients_match_oracle(ob, ac, n, use_idx):
    """d(loss)/d(theta) from the MFMA fwd+bwd kernel vs the numpy backprop (itself checked by finite differences):
    relative error of every tensor < 2e-4 (float32 accumulation over n rows), loss statistics < 1e-4."""
    rng = np.random.RandomState(1)
    m = _model(ob, ac, ent_coef=0.01)
    pl = _perturb(m, rng)
    NB = n * 2 if use_idx else n
    obs = rng.normal(0, 1, (NB, ob)).astype(np.float32)
    act = rng.normal(0, 1, (NB, ac)).astype(np.float32)
    ret = rng.normal(0, 2, NB).astype(np.float32)
    val = rng.normal(0, 2, NB).astype(np.float32)
    mean, _, _ = po.forward(pl, obs)
    old = (po.neglogp(mean, pl[10].astype(np.float64), act) + rng.normal(0, 0.3, NB)).astype(np.float32)
    w = rng.uniform(0.5, 2.0, NB).astype(np.float32)
    idx = rng.permutation(NB)[:n].astype(np.int32) if use_idx else np.arange(n, dtype=np.int32)
    advs = po.normalize_advantages(ret[idx], val[idx])
    _, stats, lr, grads = po.ppo_loss_and_grads(pl, obs[idx], act[idx], advs, ret[idx], old[idx], w[idx], 0.2, 0.01, 0.5)
    up = lambda x: torch.as_tensor(x).to(DEV)
    L = ppo_capi.lib()
    d_ret, d_val = up(ret), up(val)
    d_idx = up(idx) if use_idx else None
    mom = torch.zeros(3, dtype=torch.float64, device=DEV)
    ppo_capi.chk(L.ppo_adv_moments(d_ret.data_ptr(), d_val.data_ptr(), ppo_capi.ptr(d_idx), n, mom.data_ptr(), None))
    adv = torch.empty(n, dtype=torch.float32, device=DEV)
    ppo_capi.chk(L.ppo_adv_normalize(d_ret.data_ptr(), d_val.data_ptr(), ppo_capi.ptr(d_idx), n, mom.data_ptr(), adv.data_ptr(), None))
    assert np.allclose(adv.cpu().numpy(), advs, rtol=1e-5, atol=1e-5)
    g = torch.zeros(m.P, dtype=torch.float32, device=DEV)
    st = torch.zeros(8, dtype=torch.float64, device=DEV)
    lrat = torch.empty(n, dtype=torch.float32, device=DEV)
    d_obs, d_act, d_old, d_w = up(obs), up(act), up(old), up(w)
    ppo_capi.chk(L.ppo_grad(m.params.data_ptr(), d_obs.data_ptr(), ob, ob, ac, d_act.data_ptr(), adv.data_ptr(), d_ret.data_ptr(),
                            d_old.data_ptr(), d_w.data_ptr(), ppo_capi.ptr(d_idx), n, 1.0 / n, 0.2, 0.01, 0.5, g.data_ptr(),
                            st.data_ptr(), lrat.data_ptr(), m.workspace.data_ptr(), None))
    gl = policies.unflatten_params(g.cpu().numpy(), ob, ac)
    for k, (a, b) in enumerate(zip(gl, grads)):
        b = np.asarray(b).reshape(a.shape)
        err = np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
        assert err < 2e-4, (policies.PARAM_NAMES[k], err)
    s = st.cpu().numpy()
    assert s[6] == n
    assert s[0] / n == pytest.approx(stats[0], rel=1e-4, abs=1e-6) and s[1] / n == pytest.approx(stats[1], rel=1e-4)
    assert s[3] / n == pytest.approx(stats[3], rel=1e-3, abs=1e-6) and s[4] / n == pytest.approx(stats[4], abs=2.0 / n)
    assert np.allclose(lrat.cpu().numpy(), lr, rtol=1e-4, atol=1e-4)


def test_clip_adam_matches_tf1_formulation():
    rng = np.random.RandomState(3)
    P = 24529
    p0 = rng.normal(0, 1, P).astype(np.float32)
    m0 = rng.normal(0, 0.1, P).astype(np.float32)
    v0 = np.abs(rng.normal(0, 0.1, P)).astype(np.float32)
    g = rng.normal(0, 0.05, P).astype(np.float32)
    up = lambda x: torch.as_tensor(x.copy()).to(DEV)
    for max_norm in (0.5, 0.0):
        p, m, v = up(p0), up(m0), up(v0)
        st = torch.zeros(8, dtype=torch.float64, device=DEV)
        dg = up(g)
        ppo_capi.chk(ppo_capi.lib().ppo_clip_adam(p.data_ptr(), dg.data_ptr(), m.data_ptr(), v.data_ptr(), P, 7, 1e-3, 0.9, 0.999,
                                                  1e-5, max_norm, st.data_ptr(), None))
        gc, norm = po.clip_by_global_norm([g.astype(np.float64)], max_norm) if max_norm > 0 else ([g.astype(np.float64)], np.linalg.norm(g))
        ep, em, ev = po.adam_step([p0.astype(np.float64)], gc, [m0.astype(np.float64)], [v0.astype(np.float64)], 7, 1e-3)
        assert st.cpu().numpy()[7] == pytest.approx(norm, rel=1e-5)
        assert np.allclose(p.cpu().numpy(), ep[0], rtol=1e-5, atol=1e-6)
        assert np.allclose(m.cpu().numpy(), em[0], rtol=1e-5, atol=1e-7) and np.allclose(v.cpu().numpy(), ev[0], rtol=1e-5, atol=1e-9)


def test_model_train_step_matches_oracle():
    """PPOModel.train (model.py:179-213 contract): one optimiser step; parameters afterwards within 5e-6 of the oracle's
    step (Adam's first steps move every weight by ~lr, so this is a tight check of the gradient SIGN structure too)."""
    rng = np.random.RandomState(4)
    ob, ac, n = 121, 8, 512
    m = _model(ob, ac)
    pl = _perturb(m, rng, 0.05)
    obs = rng.normal(0, 1, (n, ob)).astype(np.float32)
    act = rng.normal(0, 1, (n, ac)).astype(np.float32)
    ret = rng.normal(0, 2, n).astype(np.float32)
    val = rng.normal(0, 2, n).astype(np.float32)
    mean, _, _ = po.forward(pl, obs)
    old = (po.neglogp(mean, pl[10].astype(np.float64), act) + rng.normal(0, 0.2, n)).astype(np.float32)
    w = np.ones(n, np.float32)
    out = m.train(1e-3, 0.2, obs, ret, np.zeros(n, bool), act, val, old, ret, w)
    assert len(out) == 7 and out[5].shape == (n,)
    advs = po.normalize_advantages(ret, val)
    _, stats, lr, grads = po.ppo_loss_and_grads(pl, obs, act, advs, ret, old, w, 0.2, 0.0, 0.5)
    gc, _ = po.clip_by_global_norm([np.asarray(g, np.float64) for g in grads], 0.5)
    p64 = [p.astype(np.float64) for p in pl]
    newp, _, _ = po.adam_step(p64, [g.reshape(p.shape) for g, p in zip(gc, p64)], [np.zeros_like(p) for p in p64],
                              [np.zeros_like(p) for p in p64], 1, 1e-3)
    for k, (a, b) in enumerate(zip(m.get_param_list(), newp)):
        assert np.allclose(a, b, rtol=0, atol=5e-6), (policies.PARAM_NAMES[k], np.abs(a - b).max())
    assert float(out[0]) == pytest.approx(stats[0], rel=1e-3, abs=1e-5) and float(out[1]) == pytest.approx(stats[1], rel=1e-4)
    assert float(out[2]) == pytest.approx(stats[2], rel=1e-6) and float(out[3]) == pytest.approx(stats[3], rel=1e-3, abs=1e-5)


def test_checkpoint_roundtrip(tmp_path):
    m = _model(121, 8, seed=1)
    path = str(tmp_path / "checkpoints" / "00000")
    m.save(path)
    import joblib
    lst = joblib.load(path)
    assert isinstance(lst, list) and [x.shape for x in lst] == policies.param_shapes(121, 8) and all(x.dtype == np.float32 for x in lst)
    m2 = _model(121, 8, seed=2, trainable=False)
    assert not np.array_equal(m2.get_param_list()[0], lst[0])
    m2.load(path)
    for a, b in zip(m2.get_param_list(), lst):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        _model(120, 8).set_param_list(lst)      # the 120-dim zoo/ckpt layout does not fit a 121-dim graph


def test_device_rollout_and_update_smoke():
    """Config-1-sized plumbing on the real env: 8 envs, nsteps 32 -> run() 15-tuple (CUDA tensors, env-major), V-trace
    consistent with the oracle on the downloaded buffers, then minibatch updates change the weights and keep them finite."""
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=8, seed=42)
    spec = policies.build_policy(env, "mlp", value_network="copy", num_hidden=64, activation="relu")
    np.random.seed(0)
    learner = model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, model_scope="model_0")
    opp = model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False, model_scope="model_1")
    opp.set_param_list(learner.get_param_list())
    T, N = 32, 8
    r = Runner(env=env, models=[learner, opp], nsteps=T, nagent=2, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0, anneal_bound=1000)
    out = r.run(1)
    assert len(out) == 15
    obs, returns, masks, actions, values, nlp, rew, onlp, oobs, oact, states, epinfos, opr, oer, ratio = out
    assert obs.shape == (2, N * T, 121) and returns.shape == (2, N * T) and actions.shape == (2, N * T, 8) and masks.dtype == torch.bool
    assert opr.shape == (N * T,) and states is None and oobs.shape == (T, N * 121)
    for x in (obs, returns, values, nlp, rew, onlp, ratio):
        assert torch.isfinite(x).all()
    # opponent == learner weights, so both IS ratios are exactly 1 up to float32 rounding
    assert torch.allclose(ratio, torch.ones_like(ratio), atol=1e-4)
    # env-major flattening: row e*T + t
    v = values[0].reshape(N, T).cpu().numpy()
    rw = rew[0].reshape(N, T).cpu().numpy()
    d = masks[0].reshape(N, T).cpu().numpy()
    last_v = learner.value(r.obs[:, 0, :].contiguous()).cpu().numpy()
    exp0 = po.vtrace_returns(rw.T.copy(), v.T.copy(), d.T.copy(), r.dones[:, 0].cpu().numpy().astype(bool), last_v,
                             np.ones((T, N), np.float32), np.ones((T, N), np.float32), 0.995)
    assert np.allclose(returns[0].reshape(N, T).cpu().numpy(), exp0.T, rtol=1e-6, atol=1e-5)
    before = learner.params.clone()
    idx = torch.randperm(N * T, device=DEV).to(torch.int32)
    w = torch.ones(N * T, dtype=torch.float32, device=DEV)
    for k in range(0, N * T, 64):
        st = learner.train_indexed(1e-3, 0.2, obs[0], returns[0], actions[0], values[0], nlp[0], w, idx[k:k + 64].contiguous(), 64)
        assert np.isfinite(st[:5]).all()
    assert torch.isfinite(learner.params).all() and not torch.equal(before, learner.params)
    env.close()


def test_learn_plumbing_config1(tmp_path):
    """BASELINE config 1 ("8 envs, MLP, ~1k steps", plumbing): nsteps=128 -> nbatch 1024 = total_timesteps, one update of
    6 epochs x 32 minibatches of 32.  Pass = loop runs, finite losses, checkpoints 00000/00001."""
    from robosumo_selfplay_amd import alg_ppo, defaults
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=8, seed=42)
    kw = defaults.get_default_params("RoboSumo-Ant-vs-Ant-v0")
    kw.update(nsteps=128)
    model = alg_ppo.learn(network="mlp", env=env, seed=42, total_timesteps=1024, nagent=2, log_dir=str(tmp_path), verbose=False, **kw)
    h = model.history
    assert len(h["lossvals"]) == 1 and np.isfinite(h["lossvals"][0]).all() and model.t == 6 * 32
    assert sorted(os.listdir(os.path.join(str(tmp_path), "checkpoints"))) == ["00000", "00001"]
    assert h["total_ratio_mean"][0] == pytest.approx(1.0, abs=1e-3)      # update 1: opponent == checkpoint 00000 == learner
    assert torch.isfinite(model.params).all()
    env.close()


@pytest.mark.parametrize("env_id", ["RoboSumo-Spider-vs-Spider-v0", "RoboSumo-Bug-vs-Bug-v0"])
def test_learn_other_matchups(env_id, tmp_path):
    """BASELINE config 4 (Spider-vs-Spider: ob 209 / ac 16 -> the widest kernel instantiations) and the Bug pair
    through the same learn() loop: device rollout, V-trace, two updates, finite losses, parameters moved."""
    from robosumo_selfplay_amd import alg_ppo, defaults
    env = SumoVecEnv(env_id, num_envs=16, seed=3)
    kw = defaults.get_default_params(env_id)
    kw.update(nsteps=16, nminibatches=4, noptepochs=2)
    model = alg_ppo.learn(network="mlp", env=env, seed=3, total_timesteps=16 * 16 * 2, nagent=2, log_dir=str(tmp_path), verbose=False, **kw)
    assert model.spec.ob_dim == env.observation_space[0].shape[0] and model.spec.ac_dim == env.action_space[0].shape[0]
    assert len(model.history["lossvals"]) == 2 and all(np.isfinite(l).all() for l in model.history["lossvals"])
    assert model.t == 2 * 2 * 4 and torch.isfinite(model.params).all()
    env.close()


def test_learn_opponent_modes_and_opponent_data(tmp_path):
    from robosumo_selfplay_amd import alg_ppo
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=1)
    model = alg_ppo.learn(network="mlp", env=env, seed=1, total_timesteps=16 * 16 * 3, nagent=2, log_dir=str(tmp_path), verbose=False,
                          nsteps=16, nminibatches=4, noptepochs=2, lr=1e-3, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0,
                          opponent_mode="ours", use_opponent_data="both", value_network="copy", num_hidden=64, activation="relu",
                          anneal_bound=1000, kl_threshold=10.0)
    assert len(model.history["lossvals"]) == 3 and all(np.isfinite(l).all() for l in model.history["lossvals"])
    assert len(os.listdir(os.path.join(str(tmp_path), "checkpoints"))) == 4
    env.close()


def test_graph_replay_matches_eager_steps():
    """The HIP-graph path of train_indexed (asynchronous, single GPU, after begin_update) replays exactly the eager launches --
    also when the caller refills the SAME tensors in place between two updates (the batch is handed over explicitly by
    begin_update; nothing is inferred from addresses)."""
    rng = np.random.default_rng(3)
    D, A, nb, n = 121, 8, 4096, 512
    mk = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(DEV)
    batches = [(mk(nb, D), mk(nb, A), mk(nb), mk(nb)) for _ in range(2)]
    w = torch.ones(nb, dtype=torch.float32, device=DEV)
    res = []
    for use_graph in (False, True):
        m = _model(D, A, seed=5)
        m.use_graph = use_graph
        obs, act, ret, val = (x.clone() for x in batches[0])         # persistent caller buffers, overwritten per update
        outs = []
        r2 = np.random.default_rng(9)
        for upd in range(2):
            for dst, src in zip((obs, act, ret, val), batches[upd]):
                dst.copy_(src)                                            # same addresses, new content
            nlp = m.act_model.action_probability(obs, given_action=act) + 0.05
            if use_graph:
                m.begin_update(obs, ret, act, val, nlp, w)
            for k in range(4):
                idx = torch.from_numpy(r2.permutation(nb)[:n].astype(np.int32)).to(DEV)
                outs.append(m.train_indexed(1e-3, 0.2, obs, ret, act, val, nlp, w, idx, n, sync=False))
            m.end_update()
        torch.cuda.synchronize()
        assert (len(m._graphs) == 1) == use_graph
        res.append((m.params.clone(), torch.stack(outs).cpu().numpy(), m.t))
    assert res[0][2] == res[1][2] == 8
    assert torch.equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    # without a hand-over the asynchronous step stays on the eager launches (never on a stale private copy)
    m = _model(D, A, seed=5)
    obs, act, ret, val = batches[0]
    nlp = m.act_model.action_probability(obs, given_action=act) + 0.05
    m.train_indexed(1e-3, 0.2, obs, ret, act, val, nlp, w, torch.arange(n, dtype=torch.int32, device=DEV), n, sync=False)
    assert len(m._graphs) == 0


@pytest.mark.parametrize("ob,ac,n", [(121, 8, 100), (209, 16, 37)])
def test_selfplay_forward_equals_separate_evaluations(ob, ac, n):
    """ppo_selfplay_forward (one launch per rollout step) against the four ppo_forward launches it replaces: bit-identical
    actions, neglogps and values; observation / done records copied; env action buffer filled."""
    import ctypes as C
    from robosumo_selfplay_amd import ppo_capi
    learner, opp = _model(ob, ac, seed=1).act_model, _model(ob, ac, seed=2, trainable=False).act_model
    g = torch.Generator(device=DEV); g.manual_seed(3)
    stride = ob + 3
    obs = torch.randn((n, 2, stride), generator=g, device=DEV)
    noise = torch.randn((2, n, ac), generator=g, device=DEV)
    done = (torch.rand((n, 2), generator=g, device=DEV) < 0.3).to(torch.uint8)
    PI, VF = ppo_capi.FWD_PI, ppo_capi.FWD_VF
    L = ppo_capi.lib()
    st = torch.cuda.current_stream().cuda_stream
    ref = {}
    for side, actor, scorer in ((0, learner, opp), (1, opp, learner)):
        o = obs[:, side, :]
        a = torch.empty((n, ac), device=DEV); nl_a = torch.empty(n, device=DEV); nl_s = torch.empty(n, device=DEV); v = torch.empty(n, device=DEV)
        ppo_capi.chk(L.ppo_forward(actor.params.data_ptr(), o.data_ptr(), n, o.stride(0), ob, ac, PI, noise[side].data_ptr(), None,
                                   a.data_ptr(), nl_a.data_ptr(), None, None, st))
        ppo_capi.chk(L.ppo_forward(scorer.params.data_ptr(), o.data_ptr(), n, o.stride(0), ob, ac, PI, None, a.data_ptr(),
                                   None, nl_s.data_ptr(), None, None, st))
        ppo_capi.chk(L.ppo_forward(learner.params.data_ptr(), o.data_ptr(), n, o.stride(0), ob, ac, VF, None, None, None, None,
                                   v.data_ptr(), None, st))
        ref[side] = dict(act=a, nlp=nl_a if side == 0 else nl_s, onlp=nl_s if side == 0 else nl_a, val=v)
    outs = [torch.empty((n, ob), device=DEV), torch.empty((n, ob), device=DEV), torch.empty((n, ac), device=DEV), torch.empty((n, ac), device=DEV)]
    outs += [torch.empty(n, device=DEV) for _ in range(6)]
    douts = [torch.empty(n, dtype=torch.uint8, device=DEV) for _ in range(2)]
    act_env = torch.zeros((n, 2, ac), device=DEV)
    fp = (C.c_void_p * 10)(*[x.data_ptr() for x in outs])
    dp = (C.c_void_p * 2)(*[x.data_ptr() for x in douts])
    ppo_capi.chk(L.ppo_selfplay_forward(learner.params.data_ptr(), opp.params.data_ptr(), obs.data_ptr(), n, obs.stride(0), obs.stride(1),
                                        ob, ac, noise[0].data_ptr(), noise[1].data_ptr(), done.data_ptr(), act_env.data_ptr(), fp, dp, st))
    torch.cuda.synchronize()
    for side in (0, 1):
        assert torch.equal(outs[side], obs[:, side, :ob])
        assert torch.equal(outs[2 + side], ref[side]["act"]) and torch.equal(act_env[:, side], ref[side]["act"])
        assert torch.equal(outs[4 + side], ref[side]["nlp"]) and torch.equal(outs[6 + side], ref[side]["onlp"])
        assert torch.equal(outs[8 + side], ref[side]["val"])
        assert torch.equal(douts[side], done[:, side])


def test_runner_rejects_mixed_matchups():
    env = SumoVecEnv("RoboSumo-Ant-vs-Bug-v0", num_envs=4, seed=0)
    with pytest.raises(ValueError, match="share observation and action spaces"):
        Runner(env=env, models=[_model(121, 8, seed=1), _model(121, 8, seed=2, trainable=False)], nsteps=4, nagent=2, gamma=0.99,
               lam=0.95, rho_bar=1.0, c_bar=1.0)
    env.close()


def test_grouped_rollout_and_update():
    """Device-mode Runner over an env with groups: every group advances on its own stream; buffers, V-trace inputs and the
    update stay consistent (finite, right shapes, per-env episode bookkeeping intact)."""
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=64, seed=2, groups=4)
    learner = _model(121, 8, seed=1)
    opp = _model(121, 8, seed=2, trainable=False)
    learner.act_model.seed(1); opp.act_model.seed(2)
    r = Runner(env=env, models=[learner, opp], nsteps=24, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    assert r.device_mode
    for upd in (1, 2):
        out = r.run(upd)
        torch.cuda.synchronize()
        obs, returns, masks, actions, values, nlp = out[0], out[1], out[2], out[3], out[4], out[5]
        assert tuple(obs.shape) == (2, 64 * 24, 121) and tuple(actions.shape) == (2, 64 * 24, 8)
        for x in (obs, returns, actions, values, nlp):
            assert torch.isfinite(x).all()
        # the time feature of every env advances by 2/500 per step within an episode (sumo_env.py:40-72): no env was skipped
        tf = obs[0].reshape(64, 24, 121)[:, :, -1]
        m0 = masks[0].reshape(64, 24)
        d = tf[:, 1:] - tf[:, :-1]
        same_ep = ~m0[:, 1:]
        assert torch.allclose(d[same_ep], torch.full_like(d[same_ep], 2.0 / 500.0), atol=1e-6)
        w = torch.ones(64 * 24, dtype=torch.float32, device=DEV)
        st = learner.train_indexed(1e-3, 0.2, obs[0].contiguous(), returns[0], actions[0], values[0], nlp[0], w, None, 64 * 24)
        assert np.isfinite(st[:5]).all()
    env.close()


def _rollout_pair(env_id, N, T, groups, fused, seed=3, pool=None, opp_params=None):
    env = SumoVecEnv(env_id, num_envs=N, seed=11, groups=groups)
    D, A = env.observation_space[0].shape[0], env.action_space[0].shape[0]
    learner, opp = _model(D, A, seed=seed, trainable=False), _model(D, A, seed=seed + 1, trainable=False)
    _perturb(learner, np.random.RandomState(seed)); _perturb(opp, np.random.RandomState(seed + 1))
    if opp_params is not None:
        opp.params.copy_(opp_params)
    learner.act_model.seed(101); opp.act_model.seed(202)
    r = Runner(env=env, models=[learner, opp], nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, anneal_bound=500)
    r.fused_rollout = fused
    if pool is not None:
        r.opponent_pool = pool(learner.spec, N, env.device)
    outs = [r.run(250), r.run(251)]                     # two consecutive rollouts: episode state carries over
    torch.cuda.synchronize()
    st = [E.get_state() for E in env.engines]
    stats = env.stats()
    env.close()
    return outs, st, stats


@pytest.mark.parametrize("env_id,N,T,groups", [("RoboSumo-Ant-vs-Ant-v0", 96, 24, 1), ("RoboSumo-Ant-vs-Ant-v0", 64, 12, 2),
                                               ("RoboSumo-Spider-vs-Spider-v0", 32, 8, 1),
                                               ("RoboSumo-Ant-vs-Ant-v0", 4096, 16, 1)])   # BASELINE configs[1] size: 2 tickets per wave slot, envs migrate between waves and XCDs
def test_rollout_kernel_matches_stepwise_path(env_id, N, T, groups):
    """sumo_rollout_steps (policies + env steps + buffer appends of a whole rollout in ONE launch) against the step-by-step
    launches it replaces (ppo_selfplay_forward, sumo_step, ppo_post_step per step): every returned array of Runner.run, the
    episode records and the env states are bit-identical."""
    fo, fs, fstat = _rollout_pair(env_id, N, T, groups, True)
    so, ss, sstat = _rollout_pair(env_id, N, T, groups, False)
    names = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opp_neglogpacs", "opp_obs", "opp_actions", "states",
             "epinfos", "off_policy_ratio", "off_env_ratio", "total_ratio"]
    for f, s_ in zip(fo, so):
        for k, (x, y) in enumerate(zip(f, s_)):
            if torch.is_tensor(x):
                assert torch.equal(x, y), names[k]
            else:
                assert x == y, names[k]
    for (a, b) in zip(fs, ss):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    for k in ("forward", "newton", "contacts", "efc", "dropped", "diverged"):
        assert fstat[k] == sstat[k], k
    assert len(fo[1][11]) > 0 or T * N < 500            # episodes did end inside the fused launches (auto-reset path covered)


def test_rollout_kernel_opponent_pool_per_env():
    """Per-env opponent snapshots (BASELINE config 5's pool on one shard): a fused rollout against a pool of three snapshots
    equals, env by env, the step-by-step rollout against that env's snapshot alone."""
    from robosumo_selfplay_amd.opponent_pool import OpponentPool
    env_id, N, T = "RoboSumo-Ant-vs-Ant-v0", 48, 10
    rng = np.random.RandomState(9)
    snaps = []
    for k in range(3):
        m = _model(121, 8, seed=40 + k, trainable=False)
        _perturb(m, rng, 0.2)
        snaps.append(m.params.clone())
    idx = rng.randint(0, 3, N)

    def pool(spec, n, dev):
        p = OpponentPool(spec, 4, n, dev)
        for k, v in enumerate(snaps):
            p.set_snapshot(k, v, label="snap%d" % k)
        p.assign(idx)
        return p
    fo, _, _ = _rollout_pair(env_id, N, T, 1, True, pool=pool)
    assert (np.bincount(idx, minlength=3) > 0).all()
    for k in range(3):
        so, _, _ = _rollout_pair(env_id, N, T, 1, False, opp_params=snaps[k])
        cols = np.nonzero(idx == k)[0]
        rows = torch.from_numpy((cols[:, None] * T + np.arange(T)[None, :]).ravel()).to(DEV)     # env-major flattening (sf01)
        for j in (0, 1, 3, 4, 5, 6, 7):
            assert torch.equal(fo[0][j][:, rows], so[0][j][:, rows]), (k, j)
    with pytest.raises(ValueError):
        p = OpponentPool(policies.PolicySpec(121, 8, value_network="copy", activation="relu"), 4, N, DEV)
        p.set_snapshot(0, snaps[0])
        p.assign(np.full(N, 2))                           # empty slot


def test_learn_with_opponent_pool_matches_glue_oracle(tmp_path):
    """learn() with a device-resident pool of 4 snapshots (MLP, fused rollout) and opponent-data reuse: runs, draws 4 snapshots per
    update from the checkpoint directory, and the batch it hands to the optimiser equals the numpy restatement of reference
    alg_ppo.py:258-344 applied to the rollout it collected (ratio hygiene, usable rows, concat of agent-1 rows, weights)."""
    from robosumo_selfplay_amd import alg_ppo
    seen = {}
    orig = alg_ppo.assemble_update_batch

    def spy(obs, returns, masks, actions, values, neglogpacs, rewards, off_policy_ratio, total_ratio, **kw):
        out = orig(obs, returns, masks, actions, values, neglogpacs, rewards, off_policy_ratio, total_ratio, **kw)
        seen["in"] = [x.clone() for x in (obs, returns, masks, actions, values, neglogpacs, rewards, off_policy_ratio, total_ratio)]
        seen["kw"], seen["out"] = kw, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        return out
    alg_ppo.assemble_update_batch = spy
    try:
        env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=64, seed=31)
        model = alg_ppo.learn(network="mlp", env=env, seed=9, total_timesteps=64 * 16 * 3, nagent=2, log_dir=str(tmp_path), verbose=False, nsteps=16,
                              nminibatches=4, noptepochs=2, lr=3e-4, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0, opponent_mode="ours",
                              use_opponent_data="both", neglogp_threshold=14.0, opponent_pool=4, value_network="copy", num_hidden=64,
                              activation="relu")
    finally:
        alg_ppo.assemble_update_batch = orig
    h = model.history
    assert len(h["opponent_versions"]) == 3 and len(h["opponent_versions"][1]) == 4 and h["opponent_versions"][0] == [0]
    assert all(np.isfinite(l).all() for l in h["lossvals"]) and env.stats()["rollout_aborts"] == 0
    obs, returns, masks, actions, values, nlp, rewards, opr, tr = [x.cpu().numpy() for x in seen["in"]]
    want = po.update_batch(obs, returns, masks, actions, values, nlp, rewards, opr, opr, tr, nbatch=64 * 16, rho_bar=1.0,
                           neglogp_threshold=14.0, use_opponent_data="both")
    got = seen["out"]
    assert 0 < len(want["usable_index"]) < 64 * 16                      # the threshold really filters some opponent rows
    assert np.array_equal(got["usable_index"].cpu().numpy(), want["usable_index"])
    for k in ("obs", "returns", "actions", "values", "neglogpacs", "weights"):
        assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert abs(h["useful_ratio"][-1] - want["useful_ratio"]) < 1e-12
    env.close()


@pytest.mark.parametrize("N,T,chunk", [(1, 3, 0), (2, 1, 0), (5, 7, 3), (40, 12, 5)])
def test_rollout_kernel_edge_shapes_and_chunks(N, T, chunk):
    """Fused rollout launch on tiny batches (fewer envs than wave slots, one env, one step) and split into several launches per
    rollout (``rollout_chunk``: steps s0 > 0, the noise and the ring of buffers reused across launches): identical to the
    step-by-step path."""
    def run(fused):
        env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=3)
        learner, opp = _model(121, 8, seed=5, trainable=False), _model(121, 8, seed=6, trainable=False)
        learner.act_model.seed(1); opp.act_model.seed(2)
        r = Runner(env=env, models=[learner, opp], nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
        r.fused_rollout, r.rollout_chunk = fused, chunk
        out = [r.run(1), r.run(2)]
        torch.cuda.synchronize()
        st, aborts = env.engine.get_state(), env.stats()["rollout_aborts"]
        env.close()
        return out, st, aborts
    (fo, fs, fa), (so, ss, _) = run(True), run(False)
    assert fa == 0
    for f, s_ in zip(fo, so):
        for x, y in zip(f, s_):
            assert torch.equal(x, y) if torch.is_tensor(x) else x == y
    for x, y in zip(fs, ss):
        assert np.array_equal(x, y)
