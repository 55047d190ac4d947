"""Policy-zoo import: the flat parameter layout against the files the reference ships (lengths / slices recorded in
tests/golden/zoo_mlp_layout.json by tests/golden/make_zoo_golden.py) and the host-side split logic."""
import json
import os

import numpy as np
import pytest

from robosumo_selfplay_amd import mjcf, policy_zoo

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zoo_mlp_layout.json")))
ENV_OF = {"ant": "RoboSumo-Ant-vs-Ant-v0", "bug": "RoboSumo-Bug-vs-Bug-v0", "spider": "RoboSumo-Spider-vs-Spider-v0"}


@pytest.mark.parametrize("key", sorted(GOLD))
def test_param_count_matches_shipped_files(key):
    g = GOLD[key]
    kind = key.split("-")[0]
    m = mjcf.load_model(ENV_OF[kind])
    # the zoo nets see the observation without the trailing time feature (eval_robosumo_against_fix.py:177-179,206)
    assert g["ob_dim"] == m.obs_dims[0] - 1 and g["ac_dim"] == m.act_dims[0]
    assert policy_zoo.zoo_mlp_param_count(g["ob_dim"], g["ac_dim"]) == g["nparams"]
    assert policy_zoo.infer_ob_dim(g["nparams"], g["ac_dim"]) == g["ob_dim"]
    assert g["dtype"] == "float32"
    # slices that only make sense if the variable order is right: filter counts are large sample counts, logstd of a
    # trained policy is negative and O(1)
    assert g["obs_count"] > 1e6 and g["ret_count"] > 1e6
    assert all(-6.0 < x < 0.5 for x in g["logstd"]) and len(g["logstd"]) == g["ac_dim"]
    assert all(s >= 0.1 - 1e-6 for s in g["obs_std_first4"])       # std = sqrt(max(var, 1e-2))


def test_split_roundtrip_and_errors():
    rng = np.random.default_rng(0)
    D, A = 120, 8
    n = policy_zoo.zoo_mlp_param_count(D, A)
    flat = rng.standard_normal(n).astype(np.float32)
    ob_dim, p = policy_zoo.split_zoo_mlp(flat, A)
    assert ob_dim == D
    assert np.array_equal(np.concatenate([np.ravel(p[k]) for k in policy_zoo._ZOO_MLP_ORDER]), flat)
    assert p["polfinal/w"].shape == (64, A) and p["logstd"].shape == (1, A) and p["obsfilter/sum"].shape == (D,)
    assert np.array_equal(p["logstd"].ravel(), flat[-A:]) and np.array_equal(np.ravel(p["retfilter/sum"]), flat[:1])
    with pytest.raises(ValueError):
        policy_zoo.split_zoo_mlp(flat[:-1], A)


def test_filter_stats_match_running_mean_std():
    p = {"f/sum": np.array([10.0, -4.0], np.float32), "f/sumsq": np.array([60.0, 8.0001], np.float32), "f/count": np.float32(2.0)}
    mean, std = policy_zoo.filter_stats(p, "f")
    assert np.allclose(mean, [5.0, -2.0])
    assert np.allclose(std, [np.sqrt(5.0), 0.1])      # second variance (5e-5) is floored at 1e-2


LGOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zoo_lstm_layout.json")))


@pytest.mark.parametrize("key", sorted(LGOLD))
def test_lstm_param_count_matches_shipped_files(key):
    g = LGOLD[key]
    m = mjcf.load_model(ENV_OF[key.split("-")[0]])
    assert g["ob_dim"] == m.obs_dims[0] - 1 and g["ac_dim"] == m.act_dims[0]
    assert policy_zoo.zoo_lstm_param_count(g["ob_dim"], g["ac_dim"]) == g["nparams"] and g["dtype"] == "float32"
    assert g["obs_count"] > 1e6 and all(-6.0 < x < 0.5 for x in g["logstd"]) and len(g["logstd"]) == g["ac_dim"]
    assert g["lstmp_bias_absmax"] < 5.0      # a bias vector, not a weight block, sits where the layout says


def test_lstm_split_roundtrip():
    rng = np.random.default_rng(1)
    D, A = 164, 12
    flat = rng.standard_normal(policy_zoo.zoo_lstm_param_count(D, A)).astype(np.float32)
    ob_dim, p = policy_zoo.split_zoo_lstm(flat, A)
    assert ob_dim == D and p["lstmv/kernel"].shape == (128, 256) and p["p/out/w"].shape == (64, A)
    assert np.array_equal(np.concatenate([np.ravel(p[k]) for k in policy_zoo._ZOO_LSTM_ORDER]), flat)
    with pytest.raises(ValueError):
        policy_zoo.split_zoo_lstm(flat[:-3], A)
