"""Zoo-vs-zoo play on the CPU oracle (TEST INFRASTRUCTURE): both agents are driven by policy-zoo nets (numpy restatement
oracle/ppo_oracle.py::zoo_mlp_forward / zoo_lstm_step of robosumo/robosumo/policy_zoo/policy.py:23-199) inside
oracle.OracleSim, the way the reference's evaluator plays them (eval_robosumo_against_fix.py:196-230: each net sees its
agent's observation without the time feature, envs built with ``_adjust_z = -0.5``).  Returns behaviour and observation
statistics that tests/test_zoo_validation.py compares with the observation-filter statistics the zoo files carry
(tests/golden/zoo_obsfilter_stats.json -- accumulated in real MuJoCo by the nets' authors)."""
import json
import os

import numpy as np

from oracle import ppo_oracle as po
from oracle.oracle import OracleSim
from robosumo_selfplay_amd import mjcf, policy_zoo

HERE = os.path.dirname(os.path.abspath(__file__))
ENV_OF = {"ant": "RoboSumo-Ant-vs-Ant-v0", "bug": "RoboSumo-Bug-vs-Bug-v0", "spider": "RoboSumo-Spider-vs-Spider-v0"}
AC = {"ant": 8, "bug": 12, "spider": 16}


def zoo_params(kind, net):
    """v3 parameter vector of the shipped zoo (fixture tests/golden/zoo_v3_params.npz, see make_zoo_stats.py)."""
    with np.load(os.path.join(HERE, "golden", "zoo_v3_params.npz"), allow_pickle=False) as z:
        return z["%s-%s-v3" % (kind, net)].copy()


def ref_stats(kind, net, v=3):
    with open(os.path.join(HERE, "golden", "zoo_obsfilter_stats.json")) as f:
        s = json.load(f)["%s-%s-v%d" % (kind, net, v)]
    return {k: (np.asarray(x) if isinstance(x, list) else x) for k, x in s.items()}


class NumpyZooNet:
    """act(obs [n, >= ob_dim]) -> actions [n, A]; keeps the LSTM state per row and zeroes it where ``reset(mask)`` says."""

    def __init__(self, kind, net, rng):
        flat = zoo_params(kind, net)
        self.net, self.A, self.rng = net, AC[kind], rng
        self.ob_dim, self.p = (policy_zoo.split_zoo_mlp if net == "mlp" else policy_zoo.split_zoo_lstm)(flat, self.A)
        self.state = None

    def act(self, obs, stochastic):
        x = np.ascontiguousarray(obs[:, :self.ob_dim], np.float32)
        if self.net == "mlp":
            mean, _, logstd = po.zoo_mlp_forward(self.p, x)
        else:
            if self.state is None:
                H = self.p["lstmp/bias"].size // 4
                self.state = np.zeros((4, x.shape[0], H), np.float32)
            mean, _, self.state = po.zoo_lstm_step(self.p, x, self.state)
            logstd = self.p["logstd"].ravel()
        if stochastic:
            mean = mean + np.exp(logstd) * self.rng.standard_normal(mean.shape).astype(np.float32)
        return mean.astype(np.float32)

    def reset(self, mask):
        if self.state is not None:
            self.state[:, mask.astype(bool), :] = 0.0


class ObsMoments:
    """Streaming per-entry mean / std of the observations a net is fed (what its obsfilter accumulated in training)."""

    def __init__(self, dim):
        self.n, self.s, self.ss = 0, np.zeros(dim), np.zeros(dim)

    def add(self, x):
        x = np.asarray(x, np.float64)
        self.n += x.shape[0]; self.s += x.sum(0); self.ss += (x * x).sum(0)

    def mean(self):
        return self.s / self.n

    def std(self):
        return np.sqrt(np.maximum(self.ss / self.n - self.mean() ** 2, 0.0))


def summarize(kind, mom, ep_len, n_win0, n_win1, n_draw, dist0, dist_t, extra=None):
    """Common result record of the CPU and the GPU play loops."""
    nb = {"ant": 13, "bug": 19, "spider": 25}[kind]
    nq, nv = 7 + (nb - 1) * 2 // 3, 6 + (nb - 1) * 2 // 3
    n_ep = n_win0 + n_win1 + n_draw
    out = dict(kind=kind, episodes=n_ep, decided=(n_win0 + n_win1) / max(1, n_ep), draws=n_draw,
               mean_len=float(np.mean(ep_len)) if len(ep_len) else float("nan"), dist0=float(dist0), dist_t=float(dist_t),
               obs_mean=mom.mean(), obs_std=mom.std(), samples=mom.n,
               blocks=dict(qpos=(0, nq), qvel=(nq, nq + nv), cfrc=(nq + nv, nq + nv + 6 * nb), opp_qpos=(nq + nv + 6 * nb, nq + nv + 6 * nb + 7),
                           opp_cfrc=(nq + nv + 6 * nb + 7, nq + nv + 6 * nb + 13)))
    if extra:
        out.update(extra)
    return out


def oracle_selfplay(kind, net, n_envs, steps, adjust_z=-0.5, cfrc_mode="rne_post", stochastic=True, seed=0, nthreads=8, dist_at=80, model=None):
    """v3-vs-v3 play of ``kind`` agents with ``net`` ('mlp' / 'lstm') nets on both sides, on the CPU oracle."""
    m = model if model is not None else mjcf.load_model(ENV_OF[kind])
    ora = OracleSim(m, n_envs)
    ora.set_adjust_z(adjust_z)
    ora.set_cfrc_mode(cfrc_mode)
    rng = np.random.default_rng(seed)
    nets = [NumpyZooNet(kind, net, rng), NumpyZooNet(kind, net, rng)]
    D = nets[0].ob_dim
    obs = ora.reset(seeds=np.arange(n_envs, dtype=np.uint64) + np.uint64(1000 * seed))
    mom = ObsMoments(D)
    ep_len, w0, w1, dr = [], 0, 0, 0
    aq = [int(x) for x in m.agent_qposadr]

    def torso_dist():
        q = ora.get_state()[0]
        return float(np.linalg.norm(q[:, aq[0]:aq[0] + 2] - q[:, aq[1]:aq[1] + 2], axis=1).mean())
    dist0, dist_t = torso_dist(), float("nan")
    ever_done = np.zeros(n_envs, bool)
    for t in range(steps):
        mom.add(obs[:, 0, :D]); mom.add(obs[:, 1, :D])
        a = np.zeros((n_envs, 2, ora.act_stride), np.float32)
        for g in range(2):
            a[:, g, :nets[g].A] = nets[g].act(obs[:, g], stochastic)
        obs, info, done, ep_r, ep_dr, ep_l = ora.step(a, nthreads=nthreads)
        fin = done[:, 0] != 0
        if fin.any():
            flags = info[:, :, 7].astype(np.int64)
            a0 = ((flags[:, 0] & 1) != 0) & fin
            a1 = ((flags[:, 1] & 1) != 0) & fin & ~a0
            w0 += int(a0.sum()); w1 += int(a1.sum()); dr += int(fin.sum() - a0.sum() - a1.sum())
            ep_len.extend(int(x) for x in ep_l[fin])
            for n_ in nets:
                n_.reset(fin)
            ever_done |= fin
        if t + 1 == dist_at:   # first-episode envs only: the others were re-placed by their reset
            q = ora.get_state()[0]
            d = np.linalg.norm(q[:, aq[0]:aq[0] + 2] - q[:, aq[1]:aq[1] + 2], axis=1)
            dist_t = float(d[~ever_done].mean()) if (~ever_done).any() else float("nan")
    return summarize(kind, mom, ep_len, w0, w1, dr, dist0, dist_t, dict(stats=ora.stats()))


def density10_ant_model():
    """Ant-vs-Ant scene compiled with ``agent_densities = [10, 10]`` -- construct_scene's own default (utils.py:97-99) instead
    of the registry's 13 (fixture tests/golden/ant_density10_model.json, derived tables only; see make_zoo_stats.py)."""
    with open(os.path.join(HERE, "golden", "ant_density10_model.json")) as f:
        return mjcf.SumoModel.from_json(f.read())


def gpu_selfplay(kind, net, n_envs, steps, adjust_z=-0.5, cfrc_mode="rne_post", stochastic=True, seed=0, dist_at=80, model=None):
    """The same play through the PRODUCT: ``SumoVecEnv`` (HIP engine) + ``policy_zoo.Zoo{MLP,LSTM}Policy`` (HIP forward kernels),
    everything resident in HBM; statistics accumulated on the device."""
    import torch
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    env = SumoVecEnv(ENV_OF[kind], num_envs=n_envs, seed=1000 * seed, cfrc_mode=cfrc_mode, adjust_z=adjust_z, model=model)
    A = AC[kind]
    flat = zoo_params(kind, net)
    pols = [policy_zoo.load_zoo_policy_from_flat(flat, A, kind=net) for _ in range(2)]
    for i, p in enumerate(pols):
        p.seed(17 + 2 * seed + i)
    D = pols[0].ob_dim
    obs = env.reset_device()
    dev = obs.device
    s = torch.zeros(D, dtype=torch.float64, device=dev)
    ss = torch.zeros(D, dtype=torch.float64, device=dev)
    nsamp = 0
    acts = torch.zeros_like(env.act_dev)
    aq = [int(x) for x in env.model.agent_qposadr]

    def torso_xy():
        q = torch.from_numpy(env.engine.get_state()[0])
        return torch.linalg.norm(q[:, aq[0]:aq[0] + 2] - q[:, aq[1]:aq[1] + 2], dim=1)
    dist0, dist_t = float(torso_xy().mean()), float("nan")
    ever_done = torch.zeros(n_envs, dtype=torch.bool, device=dev)
    w0 = w1 = dr = 0
    ep_len = []
    for t in range(steps):
        x = obs[:, :, :D].reshape(-1, D).to(torch.float64)
        s += x.sum(0); ss += (x * x).sum(0); nsamp += x.shape[0]
        for g in range(2):
            acts[:, g, :A] = pols[g].act(obs[:, g, :], stochastic=stochastic)[0]
        obs, info, done, ep_r, ep_dr, ep_l = env.step_device(acts)
        fin = done[:, 0] != 0
        nfin = int(fin.sum())
        if nfin:
            flags = info[:, :, 7].to(torch.int64)
            a0 = ((flags[:, 0] & 1) != 0) & fin
            a1 = ((flags[:, 1] & 1) != 0) & fin & ~a0
            w0 += int(a0.sum()); w1 += int(a1.sum()); dr += nfin - int(a0.sum()) - int(a1.sum())
            ep_len.extend(int(v) for v in ep_l[fin].cpu())
            for p in pols:
                if getattr(p, "recurrent", False):
                    p.reset(fin)
            ever_done |= fin
        if t + 1 == dist_at:
            d = torso_xy()
            keep = ~ever_done.cpu()
            dist_t = float(d[keep].mean()) if bool(keep.any()) else float("nan")
    mom = ObsMoments(D)
    mom.n, mom.s, mom.ss = nsamp, s.cpu().numpy(), ss.cpu().numpy()
    st = env.stats()
    env.close()
    return summarize(kind, mom, ep_len, w0, w1, dr, dist0, dist_t, dict(stats=st))


def block_report(r, ref):
    """Per-block deviations of simulated observation statistics from the zoo file's filter statistics."""
    out = {}
    for b, (lo, hi) in r["blocks"].items():
        dev = np.abs(r["obs_mean"][lo:hi] - ref["obs_mean"][lo:hi]) / ref["obs_std"][lo:hi]
        out[b] = dict(max_dev=float(dev.max()), mean_dev=float(dev.mean()))
    lo, hi = r["blocks"]["cfrc"]
    cs, cr = r["obs_mean"][lo:hi].reshape(-1, 6), ref["obs_mean"][lo:hi].reshape(-1, 6)
    out["force_ratio"] = float(cs[:, 3:].sum() / cr[:, 3:].sum())     # sum over bodies and axes of mean |clip(force)|
    out["torque_ratio"] = float(cs[:, :3].sum() / cr[:, :3].sum())
    return out
