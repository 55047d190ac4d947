"""Pins oracle/ppo_oracle.py: (i) RunnerOracle against the golden vectors produced by the reference's own Runner
(tests/golden/runner_*.npz <- /root/reference/runner.py, see make_runner_golden.py) -- bit-exact; (ii) the TF parts
(loss, gradients, Adam, clipping) by finite differences and known answers (parity unpinned for those)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from fake_rollout import CASES, make_case
from oracle import ppo_oracle as po

NAMES = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opponent_neglogpacs", "opponent_obs",
         "opponent_actions", "states", "epinfos", "off_policy_ratio", "off_env_ratio", "ratio"]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_runner_oracle_matches_reference_golden(case):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "runner_%s.npz" % case[0]))
    env, models, kw, update = make_case(case)
    r = po.RunnerOracle(env=env, models=models, **kw)
    for call in range(2):
        res = r.run(update + call)
        assert len(res) == 15 and res[10] is None
        for nm, v in zip(NAMES, res):
            if nm == "states":
                continue
            if nm == "epinfos":
                assert [e["r"] for e in v] == gold["c%d_epinfo_r" % call].tolist()
                assert [e["l"] for e in v] == gold["c%d_epinfo_l" % call].tolist()
                continue
            g = gold["c%d_%s" % (call, nm)]
            v = np.asarray(v)
            assert v.shape == g.shape and v.dtype == g.dtype, (nm, v.shape, g.shape, v.dtype, g.dtype)
            assert np.array_equal(v, g), (nm, np.abs(v.astype(np.float64) - g.astype(np.float64)).max())


def test_anneal_alpha():
    assert po.anneal_alpha(1, 500) == 1.0 and po.anneal_alpha(500, 500) == 0.0 and po.anneal_alpha(501, 500) == 0
    assert po.anneal_alpha(250, 500) == np.linspace(1, 0, 500)[249]


def test_neglogp_entropy_known_answers():
    mean = np.zeros((1, 8))
    logstd = np.zeros((1, 8))
    a = np.zeros((1, 8))
    assert po.neglogp(mean, logstd, a)[0] == pytest.approx(4 * np.log(2 * np.pi))
    assert po.entropy(logstd, 1)[0] == pytest.approx(8 * 0.5 * np.log(2 * np.pi * np.e))
    a[0, 0] = 2.0
    assert po.neglogp(mean, logstd, a)[0] == pytest.approx(4 * np.log(2 * np.pi) + 2.0)
    logstd[:] = 0.5
    assert po.neglogp(mean, logstd, a)[0] == pytest.approx(0.5 * 4 / np.exp(1.0) + 4 * np.log(2 * np.pi) + 4.0)


def test_init_params_layout_and_orthogonality():
    rng = np.random.RandomState(0)
    p = po.init_params(rng, 121, 8)
    assert [x.shape for x in p] == po.param_shapes(121, 8) and sum(x.size for x in p) == 24529
    assert all(x.dtype == np.float32 for x in p)
    w = p[0].astype(np.float64)                 # [121, 64], gain sqrt(2): columns orthogonal with norm sqrt(2)
    assert np.allclose(w.T @ w, 2 * np.eye(64), atol=1e-5)
    assert np.allclose(p[8].astype(np.float64).T @ p[8], 1e-4 * np.eye(8), atol=1e-9)
    assert np.all(p[10] == 0) and np.all(p[1] == 0)


def _rand_batch(rng, n, ob, ac):
    params = po.init_params(rng, ob, ac, 16)
    params = [x + rng.normal(0, 0.1, x.shape).astype(np.float32) for x in params]
    obs = rng.normal(0, 1, (n, ob)).astype(np.float32)
    act = rng.normal(0, 1, (n, ac)).astype(np.float32)
    adv = rng.normal(0, 1, n).astype(np.float32)
    ret = rng.normal(0, 2, n).astype(np.float32)
    mean, _, _ = po.forward(params, obs)
    old = (po.neglogp(mean, params[10].astype(np.float64), act) + rng.normal(0, 0.3, n)).astype(np.float32)
    w = rng.uniform(0.5, 2.0, n).astype(np.float32)
    return params, obs, act, adv, ret, old, w


def test_loss_gradients_by_finite_differences():
    rng = np.random.RandomState(1)
    params, obs, act, adv, ret, old, w = _rand_batch(rng, 40, 9, 3)
    args = (obs, act, adv, ret, old, w, 0.2, 0.01, 0.5)
    loss, stats, lr, grads = po.ppo_loss_and_grads(params, *args)
    assert loss == pytest.approx(stats[0] - 0.01 * stats[2] + 0.5 * stats[1])
    p64 = [x.astype(np.float64) for x in params]
    eps = 1e-6
    for k in range(13):
        flat = p64[k].ravel()
        for idx in rng.choice(flat.size, min(flat.size, 6), replace=False):
            save = flat[idx]
            flat[idx] = save + eps
            lp = po.ppo_loss_and_grads(p64, *args)[0]
            flat[idx] = save - eps
            lm = po.ppo_loss_and_grads(p64, *args)[0]
            flat[idx] = save
            fd = (lp - lm) / (2 * eps)
            assert np.asarray(grads[k]).ravel()[idx] == pytest.approx(fd, rel=2e-4, abs=1e-8), (k, idx)


def test_loss_quirks():
    """model.py:93-108: NaN ratio -> 2.0; first-order approxkl; IS weight on the pg term only; no value clipping."""
    rng = np.random.RandomState(2)
    params, obs, act, adv, ret, old, w = _rand_batch(rng, 16, 5, 2)
    old2 = old.copy()
    old2[3] = np.nan
    _, stats, lr, grads = po.ppo_loss_and_grads(params, obs, act, adv, ret, old2, w, 0.2, 0.0, 0.5)
    assert np.isnan(lr[3]) and np.isfinite(stats[0])
    mean, value, _ = po.forward(params, obs)
    nlp = po.neglogp(mean, params[10].astype(np.float64), act)
    _, stats, lr, _ = po.ppo_loss_and_grads(params, obs, act, adv, ret, old, w, 0.2, 0.0, 0.5)
    assert stats[3] == pytest.approx(np.mean(nlp - old)) and stats[1] == pytest.approx(0.5 * np.mean((value - ret) ** 2))
    _, stats2, _, _ = po.ppo_loss_and_grads(params, obs, act, adv, ret, old, 2 * w, 0.2, 0.0, 0.5)
    assert stats2[0] == pytest.approx(2 * stats[0]) and stats2[1] == pytest.approx(stats[1])


def test_adam_tf1_and_global_norm_clip():
    g = [np.array([3.0, 4.0]), np.array([[12.0]])]
    clipped, norm = po.clip_by_global_norm(g, 0.5)
    assert norm == pytest.approx(13.0) and np.allclose(clipped[0], np.array([3.0, 4.0]) * 0.5 / 13.0)
    same, _ = po.clip_by_global_norm([np.array([0.1])], 0.5)
    assert same[0][0] == 0.1
    p, m, v = [np.array([1.0])], [np.array([0.0])], [np.array([0.0])]
    p1, m1, v1 = po.adam_step(p, [np.array([2.0])], m, v, 1, 1e-3)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert m1[0][0] == pytest.approx(0.2) and v1[0][0] == pytest.approx(0.004)
    assert p1[0][0] == pytest.approx(1.0 - lr_t * 0.2 / (np.sqrt(0.004) + 1e-5))


def test_advantage_normalisation():
    r = np.array([1, 2, 3, 4], np.float32)
    v = np.array([0, 0, 1, 1], np.float32)
    a = po.normalize_advantages(r, v)
    adv = r - v
    assert np.allclose(a, (adv - adv.mean()) / (adv.std() + 1e-8)) and a.dtype == np.float32
