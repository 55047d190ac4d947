"""The CPU oracle is the checker for the HIP path, so it is pinned here against everything that CAN be pinned
offline: (i) the reference's own game logic (sumo.py / agents.py / sumo_env.py / subproc_vec_env.py, cited per
test), restated independently in numpy below; (ii) physics invariants and known answers.
MuJoCo trajectories themselves cannot be pinned (no binary, no fixtures in the reference): parity unpinned."""
import math

import numpy as np
import pytest

from robosumo_selfplay_amd import mjcf

G = 9.81


def _settled(oracle_lib, model, N=2, steps=60, seed=3):
    sim = oracle_lib.OracleSim(model, N)
    sim.reset(seeds=np.arange(N) + seed)
    z = np.zeros((N, 2, sim.act_stride), np.float32)
    for _ in range(steps):
        sim.step(z, nthreads=4)
    return sim


def test_mass_matrix_matches_jacobian_formulation(ant_model, spider_model, oracle_lib):
    for m in (ant_model, spider_model):
        sim = oracle_lib.OracleSim(m, 1)
        sim.reset(seeds=[11])
        qpos = sim.get_state()[0][0]
        sim.forward(0, np.zeros(m.nu))
        M = sim.array("M").reshape(m.nv, m.nv)
        Mnp, _, _ = mjcf.mass_matrix_np(m, qpos)
        assert np.abs(M - Mnp).max() < 1e-12


def test_free_flight_momentum(ant_model, oracle_lib):
    """No contacts, no active limits: each agent's linear momentum changes by exactly -m*g*h per mj_step in z and
    not at all in x, y (momentum = translational rows of M times qvel)."""
    m = ant_model
    sim = oracle_lib.OracleSim(m, 1)
    rng = np.random.default_rng(7)
    q = m.qpos0.copy()[None]
    q[0, 2] = q[0, 17] = 3.0
    for a in range(2):
        quat = rng.standard_normal(4)
        q[0, 15 * a + 3:15 * a + 7] = quat / np.linalg.norm(quat)
        for k in range(8):                       # mid-range of every hinge -> no limit rows
            lo, hi = m.jnt_range[1 + 9 * a + k]
            q[0, 15 * a + 7 + k] = 0.5 * (lo + hi)
    v = rng.standard_normal((1, m.nv)) * 0.5
    sim.set_state(q, v, np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))

    def momentum(qq, vv):
        M, _, _ = mjcf.mass_matrix_np(m, qq)
        Mv = M @ vv
        return np.stack([Mv[0:3], Mv[14:17]])
    p0 = momentum(q[0], v[0])
    sim.mj_step(0, np.zeros(m.nu), 1)
    assert sim.array("counts")[0] == 0 and sim.array("counts")[1] == 0
    q1, v1, _, _ = sim.get_state()
    p1 = momentum(q1[0], v1[0])
    mass = m.body_mass[1:14].sum()
    assert np.allclose(p1[:, :2], p0[:, :2], atol=1e-7)
    assert np.allclose(p1[:, 2] - p0[:, 2], -mass * G * 0.01, rtol=1e-5)


def test_resting_contact_supports_weight(ant_model, oracle_lib):
    sim = _settled(oracle_lib, ant_model, N=2, steps=120)
    q, v, _, _ = sim.get_state()
    assert np.abs(v).max() < 0.3
    sim.forward(0, np.zeros(ant_model.nu))
    qc = sim.array("qfrc_constraint")
    mg = ant_model.body_mass[1:14].sum() * G
    assert qc[2] == pytest.approx(mg, rel=5e-2) and qc[16] == pytest.approx(mg, rel=5e-2)
    f = sim.array("efc_force")
    assert np.all(f >= 0)
    # tatami top is z=0.5; standing ants keep their torso well above it
    assert 0.6 < q[0, 2] < 1.3


def test_joint_limit_pushes_back(ant_model, oracle_lib):
    m = ant_model
    sim = oracle_lib.OracleSim(m, 1)
    q = m.qpos0.copy()[None]
    q[0, 2] = 5.0
    q[0, 17] = 5.0                     # far from everything
    q[0, 7] = np.deg2rad(45.0)         # hip_1 range is +-30 deg -> upper limit violated
    sim.set_state(q, np.zeros((1, m.nv)), np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))
    sim.forward(0, np.zeros(m.nu))
    J = sim.array("efc_J").reshape(-1, m.nv)
    f = sim.array("efc_force")
    rows = np.nonzero(J[:, 6])[0]
    assert len(rows) == 1 and J[rows[0], 6] == -1.0 and f[rows[0]] > 0
    assert sim.array("qfrc_constraint")[6] < 0


def test_agent_swap_symmetry(ant_model, oracle_lib):
    """Exchanging the two (identical) agents' states and actions exchanges their outputs."""
    m = ant_model
    sim = _settled(oracle_lib, m, N=1, steps=20, seed=9)
    q, v, w, c = sim.get_state()
    sw = lambda x, n: np.concatenate([x[:, n:], x[:, :n]], axis=1)
    sim2 = oracle_lib.OracleSim(m, 1)
    sim2.set_state(sw(q, 15), sw(v, 14), sw(w, 14), c)
    a = np.random.default_rng(0).standard_normal((1, 2, 8)).astype(np.float32)
    o1, i1, d1, *_ = sim.step(a)
    o2, i2, d2, *_ = sim2.step(a[:, ::-1])
    assert np.allclose(o1[:, 0], o2[:, 1], atol=1e-5) and np.allclose(o1[:, 1], o2[:, 0], atol=1e-5)
    assert np.allclose(i1[:, 0], i2[:, 1], atol=1e-7)


def _expected_info(model, q_before, q_after, act, num_steps_after):
    """numpy restatement of sumo.py:120-202 + agents.py:216-223 for one env."""
    aq = model.agent_qposadr
    dt = model.opt[0] * model.frame_skip                                 # mujoco_env.py:121-123
    lim = model.tatami_size + 0.1                                          # sumo.py:55
    out = np.zeros((2, 8))
    lost = []
    for a in range(2):
        x, y, z = q_after[aq[a]:aq[a] + 3]
        lost.append(z < 0.29 or max(abs(x), abs(y)) >= lim)               # sumo.py:149-152
    done = False
    for a in range(2):
        o = 1 - a
        # float32 pairwise sum, then float64 product: numpy 1.18 (the reference's pin, requirements.txt) promotes
        # python-float * float32-scalar to float64; numpy >= 2 would not, hence the explicit float()
        ctrl = -0.1 * float(np.square(act[a]).sum())
        lose = -2000.0 if lost[a] else 0.0
        win = 2000.0 if lost[o] else 0.0
        main = win + lose
        done |= lost[a] or lost[o]
        if num_steps_after > model.timestep_limit:                         # sumo.py:165-167
            main += -1000.0
            done = True
        pb, pa, oa = q_before[aq[a]:aq[a] + 2], q_after[aq[a]:aq[a] + 2], q_after[aq[o]:aq[o] + 2]
        mv = (pa - pb) / dt
        direction = oa - pb
        direction = direction / np.linalg.norm(direction)
        move = max(np.sum(mv * direction), 0.0) * 0.1                      # sumo.py:194-198
        push = -10.0 * np.exp(-np.linalg.norm(oa))                         # sumo.py:200-202
        out[a, :7] = [ctrl, lose, win, main, move, push, float(ctrl) + push + move]
        out[a, 7] = 1.0 if lost[o] else 0.0
    return out, done


def test_rewards_and_info_follow_sumo_py(ant_model, oracle_lib):
    m = ant_model
    N = 6
    sim = oracle_lib.OracleSim(m, N)
    sim.reset(seeds=np.arange(N) + 40)
    rng = np.random.default_rng(2)
    for t in range(25):
        qb = sim.get_state()[0]
        cnt = sim.get_state()[3]
        act = rng.standard_normal((N, 2, 8)).astype(np.float32)
        obs, info, done, ep_r, ep_dr, ep_l = sim.step(act)
        qa = sim.get_state()[0]
        for e in range(N):
            if done[e, 0]:
                continue  # state was replaced by the reset state; terminal cases are covered below
            exp, d = _expected_info(m, qb[e], qa[e], act[e], cnt[e, 0] + 1)
            assert not d
            assert np.allclose(info[e], exp, rtol=1e-12, atol=1e-12)
            # observation layout agents.py:190-214 + time feature sumo_env.py:68-70
            for a in range(2):
                o = 1 - a
                ob = obs[e, a]
                assert np.array_equal(ob[:15], qa[e, 15 * a:15 * a + 15].astype(np.float32))
                assert np.array_equal(ob[107:114], qa[e, 15 * o:15 * o + 7].astype(np.float32))
                assert np.all(ob[29:107] == 0) and np.all(ob[114:120] == 0)
                assert ob[120] == np.float32(-1.0 + 2.0 * (cnt[e, 0] + 1) / 500.0)


def test_terminal_win_lose_and_autoreset(ant_model, oracle_lib):
    m = ant_model
    sim = _settled(oracle_lib, m, N=1, steps=30)
    q, v, w, c = sim.get_state()
    q[0, 0], q[0, 1] = 2.6, 0.0              # agent 0 far outside the ring (|x| >= 2.1) -> loses
    sim.set_state(q, v, w, c)
    obs, info, done, ep_r, ep_dr, ep_l = sim.step(np.zeros((1, 2, 8), np.float32))
    assert done.tolist() == [[1, 1]]
    assert info[0, 0, 1] == -2000 and info[0, 0, 2] == 0 and info[0, 0, 3] == -2000 and info[0, 0, 7] == 0
    assert info[0, 1, 1] == 0 and info[0, 1, 2] == 2000 and info[0, 1, 3] == 2000 and info[0, 1, 7] == 1   # 'winner'
    assert ep_l[0] == 31
    # auto-reset (subproc_vec_env.py:13-16): reset observation, counters cleared, fresh placement at r=1.15
    assert obs[0, 0, 120] == -1.0 and obs[0, 1, 120] == -1.0
    q2, v2, w2, c2 = sim.get_state()
    assert c2[0, 0] == 0 and c2[0, 1] == c[0, 1] + 1 and np.all(w2 == 0)
    assert abs(math.hypot(q2[0, 0], q2[0, 1]) - 1.15) < 0.15 and abs(q2[0, 2] - 1.25) <= 0.1
    assert np.allclose(np.linalg.norm(q2[0, 3:7]), 1.0) and np.allclose(np.linalg.norm(q2[0, 18:22]), 1.0)
    # the two agents start opposite each other (sumo.py:238-241), up to the +-0.1 noise
    assert np.allclose(q2[0, 0:2], -q2[0, 15:17], atol=0.2 + 1e-9)


def test_draw_penalty_and_timeout_flag(ant_model, oracle_lib):
    m = ant_model
    sim = _settled(oracle_lib, m, N=1, steps=10)
    q, v, w, c = sim.get_state()
    c[0, 0] = 500                              # next step makes _num_steps 501 > timestep_limit (sumo.py:165)
    sim.set_state(q, v, w, c)
    obs, info, done, ep_r, ep_dr, ep_l = sim.step(np.zeros((1, 2, 8), np.float32))
    assert done[0, 0] == 1 and info[0, 0, 3] == -1000 and info[0, 1, 3] == -1000
    assert int(info[0, 0, 7]) & 2 and int(info[0, 1, 7]) & 2          # 'timeout' (sumo_env.py:62-65)
    assert ep_l[0] == 501
    c[0, 0] = 499
    sim2 = oracle_lib.OracleSim(m, 1)
    sim2.set_state(q, v, w, c)
    _, info2, done2, *_ = sim2.step(np.zeros((1, 2, 8), np.float32))
    assert done2[0, 0] == 0 and info2[0, 0, 3] == 0


def test_episode_return_is_sum_of_agent0_rewards(ant_model, oracle_lib):
    """monitor.py:56-78 / sumo_env.py:44-58: 'r' sums reward[0] = main+shaping of agent 0, 'dr' the shaping part."""
    m = ant_model
    sim = oracle_lib.OracleSim(m, 3)
    sim.reset(seeds=[1, 2, 3])
    rng = np.random.default_rng(0)
    acc = np.zeros(3)
    accd = np.zeros(3)
    seen = 0
    for t in range(200):
        obs, info, done, ep_r, ep_dr, ep_l = sim.step(rng.standard_normal((3, 2, 8)).astype(np.float32) * 2)
        acc += info[:, 0, 3] + info[:, 0, 6]
        accd += info[:, 0, 6]
        for e in range(3):
            if done[e, 0]:
                assert ep_r[e] == pytest.approx(acc[e], rel=1e-12) and ep_dr[e] == pytest.approx(accd[e], rel=1e-12)
                acc[e] = accd[e] = 0
                seen += 1
    assert seen > 0


@pytest.mark.parametrize("env_id,n", [("RoboSumo-Ant-vs-Ant-v0", 8), ("RoboSumo-Bug-vs-Bug-v0", 12),
                                      ("RoboSumo-Spider-vs-Spider-v0", 16)])
def test_ctrl_reward_float32_pairwise(oracle_lib, env_id, n):
    m = mjcf.load_model(env_id)
    sim = oracle_lib.OracleSim(m, 4)
    sim.reset(seeds=[1, 2, 3, 4])
    act = (np.random.default_rng(5).standard_normal((4, 2, n)) * 3).astype(np.float32)
    _, info, *_ = sim.step(act)
    for e in range(4):
        for a in range(2):
            assert info[e, a, 0] == -0.1 * float(np.square(act[e, a]).sum())


def test_reset_distribution(ant_model, oracle_lib):
    """sumo.py:232-253: radius 1.15, z 1.25, opposite placement, U(-.1,.1) position noise, .1*N(0,1) velocity noise."""
    m = ant_model
    N = 512
    sim = oracle_lib.OracleSim(m, N)
    obs = sim.reset(seeds=np.arange(N))
    q, v, w, c = sim.get_state()
    assert np.all(c[:, 0] == 0) and np.all(c[:, 1] == 1) and np.all(w == 0)
    phi = np.arctan2(q[:, 1], q[:, 0])
    # heading uniform on the circle: all quadrants populated roughly equally
    hist = np.histogram(phi, bins=4, range=(-np.pi, np.pi))[0]
    assert hist.min() > N / 4 * 0.6
    hinge = np.r_[7:15, 22:30]
    assert np.all(np.abs(q[:, hinge]) <= 0.1) and q[:, hinge].std() == pytest.approx(0.2 / math.sqrt(12), rel=0.1)
    assert np.all(np.abs(q[:, 2] - 1.25) <= 0.1)
    assert v.std() == pytest.approx(0.1, rel=0.05) and abs(v.mean()) < 0.01
    assert np.allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0)
    # streams differ per seed and per reset index, and are reproducible
    sim2 = oracle_lib.OracleSim(m, N)
    sim2.reset(seeds=np.arange(N))
    assert np.array_equal(sim2.get_state()[0], q)
    sim2.reset()
    assert not np.allclose(sim2.get_state()[0], q)


def test_maxcon_cap_drops_in_order(spider_model, oracle_lib):
    sim = _settled(oracle_lib, spider_model, N=1, steps=40)
    sim.forward(0, np.zeros(spider_model.nu))
    ncon = int(sim.array("counts")[0])
    assert ncon >= 4
    full = sim.array("contacts").reshape(-1, 9)
    capped = oracle_lib.OracleSim(spider_model, 1, maxcon=ncon - 2)
    capped.set_state(*sim.get_state())
    capped.forward(0, np.zeros(spider_model.nu))
    cc = capped.array("contacts").reshape(-1, 9)
    assert len(cc) == ncon - 2 and np.array_equal(cc, full[:ncon - 2]) and capped.array("counts")[2] == 2


@pytest.mark.parametrize("poison", ["nan_qpos", "inf_qvel", "huge_qvel", "nan_warm"])
def test_divergence_guard_ends_the_episode(ant_model, oracle_lib, poison):
    """Bad-value guard (MuJoCo mj_checkPos/Vel/Acc: NaN or |x| > 1e10; the reference's mujoco-py turns it into a
    MujocoException, mujoco-py/mujoco_py/builder.py:351-369): the poisoned env reports done, zero rewards, info flag 4, a finite
    reset observation and is counted; its neighbours are untouched."""
    N = 4
    sim = oracle_lib.OracleSim(ant_model, N)
    ref = oracle_lib.OracleSim(ant_model, N)
    seeds = np.arange(N) + 21
    sim.reset(seeds=seeds); ref.reset(seeds=seeds)
    a = np.random.default_rng(0).standard_normal((N, 2, sim.act_stride)).astype(np.float32)
    for _ in range(3):
        sim.step(a, nthreads=2); ref.step(a, nthreads=2)
    q, v, w, c = sim.get_state()
    if poison == "nan_qpos":
        q[1, 5] = np.nan
    elif poison == "inf_qvel":
        v[1, 3] = np.inf
    elif poison == "huge_qvel":
        v[1, 20] = -3e10
    else:
        w[1, 0] = np.nan
    sim.set_state(q, v, w, c)
    obs, info, done, ep_r, ep_dr, ep_l = sim.step(a, nthreads=2)
    robs, rinfo, rdone, *_ = ref.step(a, nthreads=2)
    assert done[1].all() and np.all(info[1, :, :7] == 0) and np.all(info[1, :, 7] == 4)
    assert np.isfinite(obs).all() and obs[1, 0, -1] == -1.0 and ep_l[1] == 4           # reset observation, episode length reported
    assert np.isfinite(ep_r).all()
    keep = [0, 2, 3]
    assert np.array_equal(obs[keep], robs[keep]) and np.array_equal(info[keep], rinfo[keep]) and np.array_equal(done[keep], rdone[keep])
    assert sim.stats()["diverged"] == 1 and ref.stats()["diverged"] == 0
    q2, v2, w2, c2 = sim.get_state()
    assert np.isfinite(q2).all() and np.isfinite(v2).all() and np.isfinite(w2).all() and c2[1, 0] == 0
    obs, info, done, *_ = sim.step(a, nthreads=2)                                       # and it carries on like any fresh episode
    assert np.isfinite(obs).all() and np.isfinite(info).all() and sim.stats()["diverged"] == 1


# ----------------------------------------------------------------------------------------------------------------------
# SURVEY.md §7 S1 known answers that exercise the velocity-dependent terms (rne_bias / cdof_dot), the RK4 integrator and the
# constraint solver independently of both implementations.  All observables are computed in numpy from the body Jacobians of
# mjcf.mass_matrix_np (a different formulation from the oracle's composite-rigid-body / RNE passes).
# ----------------------------------------------------------------------------------------------------------------------
def _conservative_copy(model, timestep=None):
    """Test-only model edit: no joint damping, no joint limits (so free flight is a conservative system)."""
    import copy
    m = copy.deepcopy(model)
    m.tables["dof_damping"] = np.zeros_like(m.tables["dof_damping"])
    m.tables["jnt_limited"] = np.zeros_like(m.tables["jnt_limited"])
    # ... and no collisions between an agent's own geoms (limbs swinging past their ranges would touch the torso)
    b, root = m.geom_bodyid, m.body_rootid
    keep = np.array([b[g1] == 0 or b[g2] == 0 or root[b[g1]] != root[b[g2]] for g1, g2 in zip(m.pair_geom1, m.pair_geom2)])
    for k in [k for k in m.tables if k.startswith("pair_")]:
        m.tables[k] = m.tables[k][keep]
    m.npair = int(keep.sum())
    if timestep is not None:
        opt = m.tables["opt"].copy(); opt[0] = timestep; m.tables["opt"] = opt
    return m


def _agent_observables(m, q, v):
    """Per agent: total energy (kinetic incl. armature + potential) and angular momentum about its own centre of mass (world)."""
    M, jacp, jacr = mjcf.mass_matrix_np(m, q)
    xpos, xquat, _, _ = mjcf.kinematics_np(m, q)
    out = []
    for a in range(2):
        b0, nb = int(m.agent_bodyadr[a]), int(m.agent_nbody[a])
        d0, nd = int(m.agent_dofadr[a]), int(m.agent_nv[a])
        bodies = range(b0, b0 + nb)
        mass = np.array([m.body_mass[b] for b in bodies])
        xi = np.array([xpos[b] + mjcf.quat2mat(xquat[b]) @ m.body_ipos[b] for b in bodies])
        com = (mass[:, None] * xi).sum(0) / mass.sum()
        L = np.zeros(3)
        for k, b in enumerate(bodies):
            R = mjcf.quat2mat(mjcf.quat_mul(xquat[b], m.body_iquat[b]))
            Iw = R @ np.diag(m.body_inertia[b]) @ R.T
            L += Iw @ (jacr[b] @ v) + mass[k] * np.cross(xi[k] - com, jacp[b] @ v)
        va = v[d0:d0 + nd]
        ke = 0.5 * va @ M[d0:d0 + nd, d0:d0 + nd] @ va
        pe = G * (mass * xi[:, 2]).sum()
        out.append((ke + pe, L))
    return out


def _tumbling_state(m, rng, spin=3.0, joint_rate=4.0):
    q = m.qpos0.copy()
    v = np.zeros(m.nv)
    for a in range(2):
        qa, da, nqa, nva = int(m.agent_qposadr[a]), int(m.agent_dofadr[a]), int(m.agent_nq[a]), int(m.agent_nv[a])
        q[qa:qa + 3] = [(-1) ** a * 6.0, 0.0, 40.0]                   # far from the floor and from each other for the whole flight
        quat = rng.standard_normal(4); q[qa + 3:qa + 7] = quat / np.linalg.norm(quat)
        for k in range(nqa - 7):
            lo, hi = m.jnt_range[int(np.where(m.jnt_qposadr == qa + 7 + k)[0][0])]
            q[qa + 7 + k] = 0.5 * (lo + hi)
        v[da:da + 3] = rng.standard_normal(3)
        v[da + 3:da + 6] = rng.standard_normal(3) * spin
        v[da + 6:da + nva] = rng.standard_normal(nva - 6) * joint_rate
    return q, v


@pytest.mark.parametrize("which", ["ant", "spider"])
def test_free_flight_conserves_energy_and_angular_momentum(ant_model, spider_model, oracle_lib, which):
    """200 RK4 steps of tumbling free flight with every joint moving (no damping, no limits, gravity on): each agent's total
    energy and its angular momentum about its own centre of mass stay constant to the integrator's accuracy.  These are the
    quantities that depend on the Coriolis / centrifugal terms (the oracle's rne_bias and cdof_dot passes); a sign or frame
    error there shows up as O(1) drift within a few steps."""
    m = _conservative_copy(ant_model if which == "ant" else spider_model)
    sim = oracle_lib.OracleSim(m, 1)
    q, v = _tumbling_state(m, np.random.default_rng(5))
    sim.set_state(q[None], v[None], np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))
    obs0 = _agent_observables(m, q, v)
    worst_e = worst_l = 0.0
    for chunk in range(4):
        sim.mj_step(0, np.zeros(m.nu), 50)
        assert sim.array("counts")[0] == 0 and sim.array("counts")[1] == 0      # really free flight: no contact, no limit row
        q1, v1, _, _ = sim.get_state()
        for (e0, L0), (e1, L1) in zip(obs0, _agent_observables(m, q1[0], v1[0])):
            ke_scale = abs(e0 - G * 40.0 * m.body_mass[1:1 + int(m.agent_nbody[0])].sum()) + 1.0
            worst_e = max(worst_e, abs(e1 - e0) / ke_scale)
            worst_l = max(worst_l, np.abs(L1 - L0).max() / (np.abs(L0).max() + 1e-3))
    assert worst_e < 3e-3 and worst_l < 6e-3, (worst_e, worst_l)     # second-order drift at h = 0.01 (see the step-halving test below)


def test_free_flight_drift_shrinks_with_the_step(ant_model, oracle_lib):
    """Halving the time step divides the energy / angular-momentum drift over the same physical time by ~4.  MuJoCo's
    mj_RungeKutta applies the classical RK4 tableau but moves positions with mj_integratePos from the step's base point
    (SURVEY App. A.12 [EXT]); on the quaternion part that is a Lie-group Runge-Kutta without the dexp^-1 correction, which
    is second order -- the restatement shows exactly that order (ratios 3.99-4.09 measured), and every drift goes to zero
    with the step, i.e. stage states, quaternion update and bias forces are mutually consistent."""
    drifts = []
    for h, n in ((0.02, 50), (0.01, 100), (0.005, 200)):
        m = _conservative_copy(ant_model, timestep=h)
        sim = oracle_lib.OracleSim(m, 1)
        q, v = _tumbling_state(m, np.random.default_rng(8), spin=4.0, joint_rate=6.0)
        sim.set_state(q[None], v[None], np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))
        o0 = _agent_observables(m, q, v)
        sim.mj_step(0, np.zeros(m.nu), n)
        q1, v1, _, _ = sim.get_state()
        o1 = _agent_observables(m, q1[0], v1[0])
        drifts.append((abs(o1[0][0] - o0[0][0]), np.abs(o1[0][1] - o0[0][1]).max()))
    for k in range(2):
        re, rl = drifts[k][0] / drifts[k + 1][0], drifts[k][1] / drifts[k + 1][1]
        assert 3.0 < re < 5.0 and 3.0 < rl < 5.0, (drifts, re, rl)


def _advance_np(m, q, v, eps):
    """q (+) eps * v on the configuration manifold, in numpy (free joints: world-frame translation, body-frame rotation)."""
    q = q.copy()
    for j in range(m.njnt):
        qa, da = int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j])
        if m.jnt_type[j] == mjcf.JNT_FREE:
            q[qa:qa + 3] += eps * v[da:da + 3]
            w = eps * v[da + 3:da + 6]
            ang = np.linalg.norm(w)
            dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * w / ang]) if ang > 0 else np.array([1.0, 0, 0, 0])
            q[qa + 3:qa + 7] = mjcf.quat_mul(q[qa + 3:qa + 7], dq)
        else:
            q[qa] += eps * v[da]
    return q


@pytest.mark.parametrize("which", ["ant", "spider"])
def test_free_flight_acceleration_keeps_invariants_stationary(ant_model, spider_model, oracle_lib, which):
    """Integrator-independent form of the conservation test: with the oracle's qacc, d/dt of each agent's total energy and of
    its angular momentum about its centre of mass vanish (central differences of the numpy observables along (v, qacc)).
    Pins the velocity-dependent bias forces (Coriolis, centrifugal, gyroscopic: rne_bias / cdof_dot) to ~1e-8 of their size."""
    m = _conservative_copy(ant_model if which == "ant" else spider_model)
    sim = oracle_lib.OracleSim(m, 1)
    rng = np.random.default_rng(12)
    for trial in range(3):
        q, v = _tumbling_state(m, rng, spin=5.0, joint_rate=7.0)
        sim.set_state(q[None], v[None], np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))
        sim.forward(0, np.zeros(m.nu))
        assert sim.array("counts")[0] == 0 and sim.array("counts")[1] == 0
        a = sim.array("qacc")
        bias = sim.array("qfrc_bias")
        eps = 1e-5
        plus = _agent_observables(m, _advance_np(m, q, v, eps), v + eps * a)
        minus = _agent_observables(m, _advance_np(m, q, v, -eps), v - eps * a)
        M, _, _ = mjcf.mass_matrix_np(m, q)
        for ag in range(2):
            d0, nd = int(m.agent_dofadr[ag]), int(m.agent_nv[ag])
            power_scale = np.abs(v[d0:d0 + nd] * bias[d0:d0 + nd]).sum() + 1.0            # size of the terms that must cancel
            torque_scale = np.abs(bias[d0 + 3:d0 + 6]).max() + 1.0
            dE = (plus[ag][0] - minus[ag][0]) / (2 * eps)
            dL = (plus[ag][1] - minus[ag][1]) / (2 * eps)
            assert abs(dE) < 1e-6 * power_scale, (trial, ag, dE, power_scale)
            assert np.abs(dL).max() < 1e-6 * torque_scale, (trial, ag, dL, torque_scale)
        assert np.abs(bias).max() > 1.0                                                    # the terms under test are not trivially zero


def test_newton_solution_satisfies_kkt(ant_model, spider_model, oracle_lib):
    """The constrained acceleration the oracle's Newton solver returns is the optimum of MuJoCo's convex problem
    (SURVEY App. A.10-11): min_a 1/2 (a - a_smooth)' M (a - a_smooth) + sum_i s_i(J_i a - aref_i), s_i(x) = x^2 / (2 R_i) for x < 0
    else 0 (limit and pyramidal contact rows).  Checked in numpy on >= 50 contact-rich states, with M from the independent
    Jacobian formulation: stationarity M (a - a_smooth) = J' f, dual feasibility f >= 0, f_i = max(0, -jar_i) / R_i."""
    checked = most = 0
    worst = 0.0
    for m, N, steps in ((ant_model, 64, 50), (spider_model, 32, 40)):
        sim = oracle_lib.OracleSim(m, N)
        sim.reset(seeds=np.arange(N) + 77)
        rng = np.random.default_rng(2)
        for t in range(steps):      # thrash, then let most of them come down again (small actions) so that feet and bodies touch
            a = rng.standard_normal((N, 2, sim.act_stride)).astype(np.float32) * (1.0 if t < steps // 2 else 0.2)
            sim.step(a, nthreads=4)
        qs, vs, _, _ = sim.get_state()
        for e in range(N):
            ctrl = np.clip(rng.standard_normal(m.nu), -1, 1)
            sim.forward(e, ctrl)
            ncon, nefc = int(sim.array("counts", e)[0]), int(sim.array("counts", e)[1])
            if ncon < 2:
                continue
            J = sim.array("efc_J", e).reshape(nefc, m.nv)
            f, R, jar = sim.array("efc_force", e), sim.array("efc_R", e), sim.array("efc_jar", e)
            aref = sim.array("efc_aref", e)
            qacc, qs_ = sim.array("qacc", e), sim.array("qacc_smooth", e)
            M, _, _ = mjcf.mass_matrix_np(m, qs[e])
            assert np.allclose(jar, J @ qacc - aref, atol=1e-9 * (1 + np.abs(aref).max()))
            assert np.all(f >= 0) and np.all(R > 0)
            assert np.allclose(f, np.maximum(0.0, -jar) / R, rtol=1e-12, atol=1e-12)            # complementarity built into the force law
            res = M @ (qacc - qs_) - J.T @ f
            scale = np.abs(M @ (qacc - qs_)).max() + np.abs(J.T @ f).max() + 1.0
            worst = max(worst, np.abs(res).max() / scale)
            checked += 1
            most = max(most, ncon)
    assert checked >= 50 and most >= 8, (checked, most)
    assert worst < 1e-7, worst                 # solver tolerance 1e-8 on the scaled gradient norm (opt[4])


@pytest.mark.parametrize("armature", [0.0, 1.0])
def test_centrifugal_pendulum_period(ant_model, oracle_lib, armature):
    """Pendulum known answer for ONE hinge.  The scene has no world-fixed hinge, so the pendulum is a centrifugal one: the torso
    (test-only edit: 10^6 x heavier, every other joint frozen by a 10^9 armature, no gravity / damping / limits) spins at Omega
    about its vertical axis; hip_1's axis is parallel to it at distance r, so the leg swings about the radial direction with
    omega^2 = Omega^2 m r d / (I_hinge + armature)  (m, d: mass and centre-of-mass distance of the swinging limb, I_hinge its
    inertia about the hinge axis -- all computed here in numpy).  Only the centrifugal / Coriolis bias terms drive this motion
    and the armature enters as MuJoCo defines it (joint-space diagonal only), so the period pins both."""
    import copy
    m = _conservative_copy(ant_model)
    opt = m.tables["opt"].copy(); opt[1:4] = 0.0; m.tables["opt"] = opt
    arm = np.full(m.nv, 1e9); arm[:6] = 0.0; arm[14:20] = 0.0                    # free joints untouched, every hinge frozen ...
    hip = m.joint_names.index("ant0/hip_1")
    hd, hq = int(m.jnt_dofadr[hip]), int(m.jnt_qposadr[hip])
    arm[hd] = armature                                                            # ... except the pendulum's
    m.tables["dof_armature"] = arm
    torso = int(m.agent_torso[0])
    mass = m.tables["body_mass"].copy(); mass[torso] *= 1e6; m.tables["body_mass"] = mass
    sub = mass.copy()                                                             # the derived table that goes with body_mass
    for b in range(m.nbody - 1, 0, -1):
        sub[int(m.body_parentid[b])] += sub[b]
    m.tables["body_subtreemass"] = sub
    inert = m.tables["body_inertia"].copy(); inert[torso] *= 1e6; m.tables["body_inertia"] = inert
    q = m.qpos0.copy()
    q[0:3] = [0.0, 0.0, 40.0]; q[3:7] = [1, 0, 0, 0]
    q[15:18] = [30.0, 0.0, 40.0]
    for j in range(m.njnt):
        if m.jnt_type[j] != mjcf.JNT_FREE:
            q[int(m.jnt_qposadr[j])] = 0.5 * (m.jnt_range[j][0] + m.jnt_range[j][1])
    q[hq] = 0.0
    # the swinging limb: bodies hanging below hip_1
    limb = [b for b in range(m.nbody) if b == int(m.jnt_bodyid[hip]) or int(m.body_parentid[b]) == int(m.jnt_bodyid[hip])]
    xpos, xquat, xanchor, xaxis = mjcf.kinematics_np(m, q)
    assert abs(abs(xaxis[hip][2]) - 1.0) < 1e-12                                  # hinge axis parallel to the spin axis
    anchor = xanchor[hip][:2] - xpos[torso][:2]
    mb = np.array([m.body_mass[b] for b in limb])
    xi = np.array([xpos[b] + mjcf.quat2mat(xquat[b]) @ m.body_ipos[b] for b in limb])
    com = (mb[:, None] * xi).sum(0) / mb.sum()
    rel = com[:2] - xanchor[hip][:2]
    r, d = np.linalg.norm(anchor), np.linalg.norm(rel)
    assert abs(np.cross(anchor / r, rel / d)) < 1e-9                              # theta = 0 is the radial equilibrium
    I_h = 0.0
    for k, b in enumerate(limb):
        R = mjcf.quat2mat(mjcf.quat_mul(xquat[b], m.body_iquat[b]))
        I_h += (R @ np.diag(m.body_inertia[b]) @ R.T)[2, 2] + mb[k] * np.sum((xi[k][:2] - xanchor[hip][:2]) ** 2)
    Omega = 5.0
    omega = Omega * np.sqrt(mb.sum() * r * d / (I_h + armature))
    theta0 = 0.02
    q[hq] = theta0
    v = np.zeros(m.nv); v[5] = Omega
    sim = oracle_lib.OracleSim(m, 1)
    sim.set_state(q[None], v[None], np.zeros((1, m.nv)), np.zeros((1, 2), np.int32))
    period = 2 * np.pi / omega
    h = float(m.opt[0])
    n = int(2.6 * period / h)
    th = np.empty(n)
    for k in range(n):
        sim.mj_step(0, np.zeros(m.nu), 1)
        th[k] = sim.get_state()[0][0, hq]
    assert sim.array("counts")[0] == 0
    assert abs(th).max() < theta0 * 1.001 and th.min() < -0.99 * theta0           # a clean oscillation about the radial direction
    up = [k for k in range(n - 1) if th[k] < 0 <= th[k + 1]]                      # upward zero crossings, linearly interpolated
    t_cross = [(k + 1 + th[k] / (th[k] - th[k + 1])) * h for k in up]
    assert len(t_cross) >= 2
    measured = t_cross[1] - t_cross[0]
    assert measured == pytest.approx(period * (1 + theta0 ** 2 / 16), rel=2e-4), (measured, period)


def test_contact_jacobian_rows_match_independent_point_jacobians(ant_model, spider_model, oracle_lib):
    """Constraint Jacobian of the oracle against point Jacobians built in numpy from mjcf.mass_matrix_np's body Jacobians (a
    different formulation from the oracle's cdof-based one).  For every pyramidal contact (condim 3: rows n +- mu t1, n +- mu t2,
    MuJoCo's pyramid [EXT]): the mean of the four rows is the normal row n' (Jp_2 - Jp_1) at the contact point; the two
    half-differences are mu t' (Jp_2 - Jp_1) for unit tangents t1, t2 that are orthogonal to n and to each other.  Limit rows
    precede the contact rows and carry a single -+1."""
    checked = 0
    for m, N, steps in ((ant_model, 48, 50), (spider_model, 24, 40)):
        sim = oracle_lib.OracleSim(m, N)
        sim.reset(seeds=np.arange(N) + 5)
        rng = np.random.default_rng(4)
        for t in range(steps):
            sim.step(rng.standard_normal((N, 2, sim.act_stride)).astype(np.float32) * (1.0 if t < steps // 2 else 0.2), nthreads=4)
        qs = sim.get_state()[0]
        gbody = m.geom_bodyid
        for e in range(N):
            sim.forward(e, np.zeros(m.nu))
            ncon, nefc = int(sim.array("counts", e)[0]), int(sim.array("counts", e)[1])
            if ncon == 0:
                continue
            J = sim.array("efc_J", e).reshape(nefc, m.nv)
            con = sim.array("contacts", e).reshape(ncon, 9)
            nlim = nefc - 4 * ncon
            assert nlim >= 0
            for r in range(nlim):                                   # hinge-limit rows: one entry of magnitude 1
                nz = np.nonzero(J[r])[0]
                assert len(nz) == 1 and abs(abs(J[r, nz[0]]) - 1.0) < 1e-15
            M, jacp, jacr = mjcf.mass_matrix_np(m, qs[e])
            xpos, xquat, _, _ = mjcf.kinematics_np(m, qs[e])
            xipos = np.array([xpos[b] + mjcf.quat2mat(xquat[b]) @ m.body_ipos[b] for b in range(m.nbody)])

            def point_jac(b, p):                                     # 3 x nv translational Jacobian of the point p riding on body b
                if b == 0:
                    return np.zeros((3, m.nv))
                return jacp[b] + np.stack([np.cross(jacr[b][:, d], p - xipos[b]) for d in range(m.nv)], axis=1)
            for k in range(ncon):
                p, n = con[k, 1:4], con[k, 4:7]
                b1, b2 = int(gbody[int(con[k, 7])]), int(gbody[int(con[k, 8])])
                Jd = point_jac(b2, p) - point_jac(b1, p)
                rows = J[nlim + 4 * k:nlim + 4 * k + 4]
                scale = np.abs(Jd).max() + 1e-12
                assert abs(np.linalg.norm(n) - 1.0) < 1e-12
                assert np.abs(rows.mean(0) - n @ Jd).max() < 1e-10 * scale            # normal part
                assert np.abs((rows[0] + rows[1]) - (rows[2] + rows[3])).max() < 1e-10 * scale
                tang = []
                for a, b_ in ((0, 1), (2, 3)):
                    jt = 0.5 * (rows[a] - rows[b_])                                     # = mu t' Jd
                    t_mu, res, rank, _ = np.linalg.lstsq(Jd.T, jt, rcond=None)
                    if rank < 3:
                        continue                                                        # degenerate Jd (cannot recover t); rare
                    assert np.abs(Jd.T @ t_mu - jt).max() < 1e-9 * scale
                    mu = np.linalg.norm(t_mu)
                    assert mu > 0 and abs(t_mu @ n) < 1e-8 * mu
                    tang.append(t_mu / mu)
                    pair = m.pair_friction.reshape(-1, 3) if m.pair_friction.ndim == 1 else m.pair_friction
                    assert np.isclose(mu, pair[:, 0], rtol=1e-9).any()                  # a friction coefficient of the scene
                if len(tang) == 2:
                    assert abs(tang[0] @ tang[1]) < 1e-8
                checked += 1
    assert checked >= 100, checked


def test_capsule_on_tatami_contacts_match_geometry(ant_model, oracle_lib):
    """Narrow phase against elementary geometry computed in numpy: a leg capsule resting on the top face of the tatami box touches
    with one or both of its end spheres -- contact normal +-z, distance = (end-sphere centre z - radius) - top, contact point at the
    end sphere's xy, half-way between the two surfaces (MuJoCo's contact convention [EXT]); the torso sphere likewise."""
    m = ant_model
    sim = oracle_lib.OracleSim(m, 32)
    sim.reset(seeds=np.arange(32) + 40)
    z = np.zeros((32, 2, sim.act_stride), np.float32)
    for _ in range(60):
        sim.step(z, nthreads=4)
    qs = sim.get_state()[0]
    top = float(m.geom_pos[1][2] + m.geom_size[1][2])
    assert m.geom_names[1] == "tatami" and abs(top - 0.5) < 1e-12
    seen = 0
    for e in range(32):
        sim.forward(e, np.zeros(m.nu))
        ncon = int(sim.array("counts", e)[0])
        if not ncon:
            continue
        con = sim.array("contacts", e).reshape(ncon, 9)
        xpos, xquat, _, _ = mjcf.kinematics_np(m, qs[e])
        for k in range(ncon):
            g1, g2 = int(con[k, 7]), int(con[k, 8])
            if 1 not in (g1, g2) or abs(abs(con[k, 6]) - 1.0) > 1e-12:
                continue
            assert con[k, 6] == (-1.0 if g2 == 1 else 1.0)          # the normal points from geom 1 to geom 2
            g2 = g1 if g2 == 1 else g2                               # the moving geom
            b = int(m.geom_bodyid[g2])
            R = mjcf.quat2mat(mjcf.quat_mul(xquat[b], m.geom_quat[g2]))
            centre = xpos[b] + mjcf.quat2mat(xquat[b]) @ m.geom_pos[g2]
            r = float(m.geom_size[g2][0])
            if m.geom_type[g2] == mjcf.GEOM_SPHERE:
                ends = [centre]
            elif m.geom_type[g2] == mjcf.GEOM_CAPSULE:
                h = float(m.geom_size[g2][1])
                ends = [centre + R[:, 2] * h, centre - R[:, 2] * h]
            else:
                continue
            p = con[k, 1:4]
            end = min(ends, key=lambda c_: np.linalg.norm(c_[:2] - p[:2]))
            if np.linalg.norm(end[:2] - p[:2]) > 1e-9:
                continue                                            # an interior (edge-on) contact point of the capsule axis: not this test
            dist = end[2] - r - top
            assert abs(con[k, 0] - dist) < 1e-10, (con[k, 0], dist)
            assert abs(p[2] - (top + 0.5 * dist)) < 1e-10
            assert max(abs(p[0]), abs(p[1])) <= m.geom_size[1][0] + 1e-9    # on the top face
            seen += 1
    assert seen >= 40, seen


def test_oracle_cfrc_ext_against_generalised_constraint_force():
    """cfrc_mode = rne_post in the oracle (rne_post(): the restatement of mj_rnePostConstraint's contact part) against an independent
    route through the same solution: the per-body contact wrenches of an agent, summed, must be the generalised constraint force
    J^T f on that agent's free joint -- force on its three translational dofs (world axes), torque about the torso's origin on its
    three rotational dofs (body axes) -- because joint-limit rows only act on hinge dofs.  Also: resting weight, and the default
    mode's zeros in the observation."""
    from robosumo_selfplay_amd import mjcf
    from oracle.oracle import OracleSim
    m = mjcf.load_model("RoboSumo-Ant-vs-Ant-v0")
    sim = OracleSim(m, 6)
    obs0 = sim.reset(seeds=np.arange(6, dtype=np.uint64) + 40)
    rng = np.random.default_rng(8)
    nb = sim.nbody
    out = None
    for t in range(3):
        out = sim.step((rng.standard_normal((6, 2, sim.act_stride)) * 0.6).astype(np.float32))
    assert np.all(out[0][:, :, 29:107] == 0) and np.all(out[0][:, :, 114:120] == 0)          # default: cfrc_ext == 0 (agents.py:190-214 on MuJoCo 2.1)
    sim.set_cfrc_mode("rne_post")
    checked = 0
    for t in range(25):
        out = sim.step((rng.standard_normal((6, 2, sim.act_stride)) * 0.6).astype(np.float32))
        for e in range(6):
            sim.forward(e)
            cf = sim.array("cfrc_ext", e).reshape(nb, 6)
            qc = sim.array("qfrc_constraint", e)
            com = sim.array("subtree_com", e).reshape(nb, 3)
            xpos = sim.array("xpos", e).reshape(nb, 3)
            xquat = sim.array("xquat", e).reshape(nb, 4)
            for ag in range(2):
                b0, n, d0 = int(m.agent_bodyadr[ag]), int(m.agent_nbody[ag]), int(m.agent_dofadr[ag])
                F = cf[b0:b0 + n, 3:].sum(0)
                T = cf[b0:b0 + n, :3].sum(0)                                  # about the agent's subtree CoM, world axes
                scale = 1.0 + np.abs(F).max()
                assert np.abs(F - qc[d0:d0 + 3]).max() < 1e-9 * scale, (t, e, ag, F, qc[d0:d0 + 3])
                T_torso = T + np.cross(com[b0] - xpos[b0], F)                  # shifted to the torso's origin
                w, x, y, z = xquat[b0]
                R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                assert np.abs(R.T @ T_torso - qc[d0 + 3:d0 + 6]).max() < 1e-9 * (1.0 + np.abs(T_torso).max()), (t, e, ag)
                checked += int(np.abs(F).max() > 1.0)
    assert checked > 100
    # |clip| in the observation, own bodies and the opponent's torso
    cf = sim.array("cfrc_ext", 0)          # (of the forward just evaluated: not the step's; the step's is what obs shows)
    assert np.count_nonzero(out[0][:, :, 29:107]) > 0 and out[0][:, :, 29:107].min() >= 0 and out[0][:, :, 29:107].max() <= 100
