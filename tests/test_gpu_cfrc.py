"""cfrc_mode = rne_post (include/sumo_hip.h): the optional second launch of sumo_step that fills the contact-force
entries of the observations, against the oracle's restatement of mj_rnePostConstraint and against elementary statics.  Parity is
unpinned (no MuJoCo here, and the reference itself never sees non-zero entries: its scenes carry no force sensor); the default mode
('zero') is what every other test runs."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from robosumo_selfplay_amd import capi
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    from test_gpu_env_parity import Pair, relerr


def _pair(env_id, N):
    p = Pair(env_id, N)
    p.eng.set_cfrc_mode("rne_post")
    p.ora.set_cfrc_mode("rne_post")
    p.reset()
    return p


@pytest.mark.parametrize("env_id", ["RoboSumo-Ant-vs-Ant-v0", "RoboSumo-Spider-vs-Spider-v0", "RoboSumo-Ant-vs-Bug-v0"])
def test_cfrc_ext_matches_oracle(env_id):
    p = _pair(env_id, 32)
    rng = np.random.default_rng(3)
    nb = p.eng.nbody
    seen = 0
    for t in range(8):
        a = (rng.standard_normal((p.N, 2, p.eng.act_stride)) * 0.8).astype(np.float32)
        (gobs, ginfo, gdone, *_), (oobs, oinfo, odone, *_) = p.step(a)
        assert np.array_equal(gdone, odone)
        g = p.eng.get_cfrc_ext()
        for e in range(p.N):
            if gdone[e, 0]:
                continue                                  # reset observation: zeros on both sides (checked through obs below)
            o = p.ora.array("cfrc_ext", e).reshape(nb, 6)
            scale = 1.0 + np.abs(o).max()
            assert np.abs(g[e] - o).max() < 1e-7 * scale, (t, e, np.abs(g[e] - o).max())
            seen += int(np.abs(o).max() > 1.0)
        assert np.abs(gobs - oobs).max() < 2e-5           # the float32 observations, force entries included
        p.eng.set_state(*p.ora.get_state())               # resync: rounding drift must not mask per-step agreement
    assert seen > 50                                      # the agents stand on the tatami: most env steps carry contact forces


def test_cfrc_ext_statics_and_observation_entries():
    """An ant at rest: the contact forces on an agent's bodies add up to its weight (z) and to nothing sideways; the observation
    entries are |clip(., +-100)| of the per-body wrenches, for the own bodies and for the opponent's torso."""
    p = _pair("RoboSumo-Ant-vs-Ant-v0", 8)
    a = np.zeros((p.N, 2, p.eng.act_stride), np.float32)
    for t in range(250):
        (gobs, _, gdone, *_), _ = p.step(a)
        assert not gdone.any()
        p.eng.set_state(*p.ora.get_state())
    g = p.eng.get_cfrc_ext()
    m = p.m
    mass = np.asarray(m.tables["body_mass"], np.float64)
    grav = 9.81
    qvel = p.eng.get_state()[1]
    settled = 0
    for e in range(p.N):
        for ag in range(2):
            d0, nd = int(m.agent_dofadr[ag]), int(m.agent_nv[ag])
            if np.abs(qvel[e, d0:d0 + nd]).max() > 1e-3:
                continue                                  # still rocking on its legs
            settled += 1
            b0, n = int(m.agent_bodyadr[ag]), int(m.agent_nbody[ag])
            F = g[e, b0:b0 + n, 3:].sum(0)
            W = mass[b0:b0 + n].sum() * grav
            assert abs(F[2] - W) < 1e-2 * W and np.abs(F[:2]).max() < 1e-2 * W, (e, ag, F, W)
    assert settled >= 8
    # the observation entries are |clip(., +-100)| of exactly these numbers
    nq, nv = int(m.agent_nq[0]), int(m.agent_nv[0])
    own = gobs[:, 0, nq + nv:nq + nv + 6 * int(m.agent_nbody[0])].reshape(p.N, -1, 6)
    b0 = int(m.agent_bodyadr[0])
    assert np.allclose(own, np.abs(np.clip(g[:, b0:b0 + own.shape[1]], -100, 100)).astype(np.float32), rtol=0, atol=1e-6)
    opp_torso = gobs[:, 0, nq + nv + 6 * int(m.agent_nbody[0]) + 7:nq + nv + 6 * int(m.agent_nbody[0]) + 13]
    b1 = int(m.agent_bodyadr[1])
    assert np.allclose(opp_torso, np.abs(np.clip(g[:, b1], -100, 100)).astype(np.float32), rtol=0, atol=1e-6)


def test_cfrc_mode_default_is_zero_and_fused_rollout_refuses_rne_post():
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=3)
    assert env.cfrc_mode == "zero"
    obs = env.reset()
    for _ in range(3):
        obs, *_ = env.step(np.zeros((16, 2, 8), np.float32))
    assert np.all(obs[:, :, 29:107] == 0) and np.all(obs[:, :, 114:120] == 0)       # agents.py:190-214 with cfrc_ext == 0
    env.close()
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=3, cfrc_mode="rne_post")
    obs = env.reset()
    for _ in range(30):
        obs, *_ = env.step(np.zeros((16, 2, 8), np.float32))
    assert np.count_nonzero(obs[:, :, 29:107]) > 0
    ro = capi.Rollout()
    z = torch.zeros(16 * 4 * 2 * 121, dtype=torch.float32, device="cuda")
    for f in ("learner_params", "opponent_params", "noise0", "noise1", "obs", "act", "rew", "val", "nlp", "onlp", "done", "ep_done", "ep_r", "ep_l"):
        setattr(ro, f, z.data_ptr())
    ro.npool, ro.ob_dim, ro.ac_dim, ro.T, ro.Ntot, ro.env_offset, ro.s0, ro.K, ro.alpha = 1, 121, 8, 4, 16, 0, 0, 4, 0.5
    with pytest.raises(capi.SumoHipError, match="rne_post"):
        env.rollout_steps_group(0, ro)
    from robosumo_selfplay_amd import model as model_mod, policies
    from robosumo_selfplay_amd.runner import Runner
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    ms = [model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    r = Runner(env=env, models=ms, nsteps=4, nagent=2, gamma=0.99, lam=0.95, rho_bar=1.0, c_bar=1.0)
    assert r.device_mode and not r.fused_ok()
    out = r.run(1)                                       # step-by-step launches with the second launch per step
    assert torch.isfinite(out[0]).all() and (out[0][:, :, 29:107] != 0).any()
    env.close()
    with pytest.raises(ValueError):
        SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=16, seed=3, cfrc_mode="sensor")
