"""Device-resident drop-in for the reference's ``SubprocVecEnv`` of RoboSumo envs.

Interface mirrored (same names, argument meaning, error behaviour):
  * baselines ``VecEnv`` contract  -- reference baselines/baselines/common/vec_env/vec_env.py:29-138
  * concrete behaviour             -- reference subproc_vec_env.py:35-116 (auto-reset on done[0], stacked
    observations ``[N, 2, D]``, ``infos`` = tuple over envs of tuple over agents of dicts)
  * what each "worker" wrapped     -- reference run.py:73-83: gym env -> SumoEnv wrapper (sumo_env.py) -> Monitor
    (baselines/baselines/bench/monitor.py:51-78)

One ``SumoVecEnv`` owns one GPU's shard of environments; all N envs advance in a single kernel launch
(csrc/sumo_engine.hip).  ``step_async``/``step_wait`` return numpy arrays like the reference; ``step_device`` keeps
everything in HBM for the device-side Runner.
"""
import time

import numpy as np

from . import capi, mjcf
from .spaces import Box, Tuple

_INFO_KEYS = ("ctrl_reward", "lose_penalty", "win_reward", "main_reward", "move_to_opp_reward", "push_opp_reward",
              "shaping_reward")


class VecEnv(object):
    """Subset of the baselines VecEnv ABC used by the hot path."""
    closed = False
    viewer = None
    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    @property
    def unwrapped(self):
        return self

    def render(self, mode="human"):
        raise NotImplementedError("rendering is outside the hot path (SURVEY.md §2.4)")

    def get_images(self):
        raise NotImplementedError


class EnvSpec(object):
    def __init__(self, env_id):
        self.id = env_id


class SumoVecEnv(VecEnv):
    def __init__(self, env_id="RoboSumo-Ant-vs-Ant-v0", num_envs=1, seed=0, device=0, asset_dir=None, model=None, groups=1,
                 cfrc_mode="zero", adjust_z=0.0):
        """``groups`` > 1 splits the envs into that many equal, contiguous groups, each with its own engine: a group's step is
        its own kernel launch (``step_device_group``), so a caller that advances the groups on separate streams is not held
        back by the slowest env of the whole batch at every step (the device-mode Runner does that).  All buffers stay
        unified ``[N, ...]`` tensors; ``step_device`` / the host API step every group."""
        import torch
        if not torch.cuda.is_available():
            raise capi.SumoHipError("SumoVecEnv needs a HIP device (there is no CPU fallback in the product path)")
        self._torch = torch
        self.model = model if model is not None else mjcf.load_model(env_id, asset_dir)
        self.device = torch.device("cuda", int(device))
        groups = int(groups)
        if groups < 1 or num_envs % groups:
            raise ValueError("num_envs %d is not divisible into %d groups" % (num_envs, groups))
        self.groups, self.group_size = groups, num_envs // groups
        self.engines = [capi.Engine(self.model, self.group_size, device=int(device)) for _ in range(groups)]
        # 'zero' = the reference's observations (its MuJoCo 2.1 scenes carry no force sensor: cfrc_ext == 0); 'rne_post' fills the contact-force
        # entries as a sensor-equipped MuJoCo would (second launch per step, no fused rollout; include/sumo_hip.h: cfrc_mode)
        if cfrc_mode not in ("zero", "rne_post"):
            raise ValueError("cfrc_mode must be 'zero' or 'rne_post'")
        self.cfrc_mode = cfrc_mode
        if cfrc_mode != "zero":
            for E_ in self.engines:
                E_.set_cfrc_mode(cfrc_mode)
        # Agent._adjust_z (agents.py:33,155-161): offset of the z an agent REPORTS (observations, lose test).  0 = training
        # (run.py:76-77); the reference's evaluation / play scripts set -0.5 (eval_robosumo_against_fix.py:108-115)
        self.adjust_z = float(adjust_z)
        if self.adjust_z != 0.0:
            for E_ in self.engines:
                E_.set_adjust_z(self.adjust_z)
        self.engine = self.engines[0]                                   # dimensions / limits (identical for every group)
        E = self.engine
        obs_dims, act_dims = self.model.obs_dims, self.model.act_dims
        ob_spaces = [Box(-np.inf * np.ones(d), np.inf * np.ones(d)) for d in obs_dims]           # agents.py:85-90
        lo = self.model.actuator_ctrlrange[:, 0]
        hi = self.model.actuator_ctrlrange[:, 1]
        ac_spaces = []
        for a in range(2):                                                                        # agents.py:92-115
            u0, nu = int(self.model.agent_uadr[a]), int(self.model.agent_nu[a])
            ac_spaces.append(Box(lo[u0:u0 + nu], hi[u0:u0 + nu]))
        VecEnv.__init__(self, int(num_envs), Tuple(ob_spaces), Tuple(ac_spaces))
        self.spec = EnvSpec(mjcf.canonical_id(env_id))
        self.agents = list(self.model.body_names[int(b)].split("/")[0] for b in self.model.agent_torso)
        self.seeds = (np.uint64(seed) + np.arange(num_envs, dtype=np.uint64))                     # run.py:144 (seed + i)
        N = self.num_envs
        f32, f64 = torch.float32, torch.float64
        self.obs_dev = torch.zeros((N, 2, E.obs_stride), dtype=f32, device=self.device)
        self.act_dev = torch.zeros((N, 2, E.act_stride), dtype=f32, device=self.device)
        self.info_dev = torch.zeros((N, 2, capi.INFO_STRIDE), dtype=f64, device=self.device)
        self.done_dev = torch.zeros((N, 2), dtype=torch.uint8, device=self.device)
        self.ep_r_dev = torch.zeros(N, dtype=f64, device=self.device)
        self.ep_dr_dev = torch.zeros(N, dtype=f64, device=self.device)
        self.ep_l_dev = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.waiting = False
        self.tstart = time.time()
        self._needs_seed = True
        self._same_dims = obs_dims[0] == obs_dims[1] and act_dims[0] == act_dims[1]

    # ---- device-side API -----------------------------------------------------------------------------------
    def _stream(self):
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def _gs(self, g):
        return slice(g * self.group_size, (g + 1) * self.group_size)

    def reset_device(self):
        self._assert_not_closed()
        for g, E in enumerate(self.engines):
            sl = self._gs(g)
            E.reset(self.obs_dev[sl].data_ptr(), seeds=self.seeds[sl] if self._needs_seed else None, stream=self._stream())
        self._needs_seed = False
        return self.obs_dev

    def step_device_group(self, g, actions):
        """Advance group ``g`` only (its slice of every buffer), on the current stream.  ``actions`` is the FULL [N, 2, A]
        action tensor; the group's rows are read."""
        sl = self._gs(g)
        self.engines[g].step(actions[sl].data_ptr(), self.obs_dev[sl].data_ptr(), self.info_dev[sl].data_ptr(),
                             self.done_dev[sl].data_ptr(), self.ep_r_dev[sl].data_ptr(), self.ep_dr_dev[sl].data_ptr(),
                             self.ep_l_dev[sl].data_ptr(), stream=self._stream())

    def rollout_steps_group(self, g, ro):
        """K fused self-play rollout steps of group ``g`` on the current stream (``capi.Engine.rollout_steps``): policies, env
        steps and the appends to the rollout buffers in one launch.  ``ro`` is a ``capi.Rollout`` whose ``env_offset`` is the
        group's first env."""
        sl = self._gs(g)
        self.engines[g].rollout_steps(ro, self.act_dev[sl].data_ptr(), self.obs_dev[sl].data_ptr(), self.info_dev[sl].data_ptr(),
                                      self.done_dev[sl].data_ptr(), self.ep_r_dev[sl].data_ptr(), self.ep_dr_dev[sl].data_ptr(),
                                      self.ep_l_dev[sl].data_ptr(), stream=self._stream())

    def rollout_steps_lstm_group(self, g, ro):
        """The same for recurrent policies (``capi.Engine.rollout_steps_lstm``, ``ro`` a ``capi.RolloutLstm``)."""
        sl = self._gs(g)
        self.engines[g].rollout_steps_lstm(ro, self.act_dev[sl].data_ptr(), self.obs_dev[sl].data_ptr(), self.info_dev[sl].data_ptr(),
                                           self.done_dev[sl].data_ptr(), self.ep_r_dev[sl].data_ptr(), self.ep_dr_dev[sl].data_ptr(),
                                           self.ep_l_dev[sl].data_ptr(), stream=self._stream())

    def step_device(self, actions):
        """actions: float32 CUDA tensor [N, 2, act_stride]. Returns (obs, info, done, ep_r, ep_dr, ep_l) tensors that
        are overwritten by the next call."""
        self._assert_not_closed()
        if actions.dtype != self._torch.float32 or not actions.is_cuda or not actions.is_contiguous() \
                or tuple(actions.shape) != tuple(self.act_dev.shape):
            raise ValueError("actions must be a contiguous float32 CUDA tensor of shape %s" % (tuple(self.act_dev.shape),))
        for g in range(self.groups):
            self.step_device_group(g, actions)
        return self.obs_dev, self.info_dev, self.done_dev, self.ep_r_dev, self.ep_dr_dev, self.ep_l_dev

    def set_adjust_z(self, adjust_z):
        """Counterpart of ``for agent in env.agents: agent._adjust_z = v`` (eval_robosumo_against_fix.py:110-112)."""
        self._assert_not_closed()
        self.adjust_z = float(adjust_z)
        for E_ in self.engines:
            E_.set_adjust_z(self.adjust_z)

    def stats(self):
        """Solver / contact statistics summed over the groups' engines (maxima for the ``max_*`` entries)."""
        out = None
        for E in self.engines:
            st = E.stats()
            if out is None:
                out = dict(st)
            else:
                for k, v in st.items():
                    out[k] = max(out[k], v) if k.startswith("max_") else out[k] + v
        return out

    # ---- reference (host) API ------------------------------------------------------------------------------------
    def _obs_host(self):
        o = self.obs_dev.cpu().numpy()
        if self._same_dims:
            return o[:, :, :self.model.obs_dims[0]].copy()
        # heterogeneous match-ups: the reference's np.stack of ragged tuples yields an object array
        out = np.empty((self.num_envs, 2), dtype=object)
        for e in range(self.num_envs):
            for a in range(2):
                out[e, a] = o[e, a, :self.model.obs_dims[a]].copy()
        return out

    def reset(self):
        self._assert_not_closed()
        self.reset_device()
        return self._obs_host()

    def step_async(self, actions):
        self._assert_not_closed()
        a = np.zeros(tuple(self.act_dev.shape), np.float32)
        for e in range(self.num_envs) if not self._same_dims else ():
            for g in range(2):
                a[e, g, :self.model.act_dims[g]] = np.asarray(actions[e][g], np.float32)
        if self._same_dims:
            act = np.asarray(actions, dtype=np.float32)
            if act.shape != a.shape:
                raise ValueError("actions must have shape %s, got %s" % (a.shape, act.shape))
            a = act
        self.act_dev.copy_(self._torch.from_numpy(np.ascontiguousarray(a)))
        self.step_device(self.act_dev)
        self.waiting = True

    def step_wait(self):
        self._assert_not_closed()
        self.waiting = False
        info = self.info_dev.cpu().numpy()
        done = self.done_dev.cpu().numpy().astype(bool)
        ep_r, ep_l = self.ep_r_dev.cpu().numpy(), self.ep_l_dev.cpu().numpy()
        obs = self._obs_host()
        rews = info[:, :, 3] + info[:, :, 6]                                      # sumo.py:186
        now = round(time.time() - self.tstart, 6)
        infos = []
        for e in range(self.num_envs):
            per_agent = []
            for g in range(2):
                d = {k: float(info[e, g, i]) for i, k in enumerate(_INFO_KEYS)}
                flags = int(info[e, g, 7])
                if flags & 1:
                    d["winner"] = True                                             # sumo.py:159
                if flags & 2:
                    d["timeout"] = True                                            # sumo_env.py:62-65
                if flags & 4:
                    d["diverged"] = True       # bad-value guard (include/sumo_hip.h): the reference raises MujocoException there
                per_agent.append(d)
            if done[e, 0]:                                                         # monitor.py:63-78 (agent 0 only)
                per_agent[0]["episode"] = {"r": round(float(ep_r[e]), 6), "l": int(ep_l[e]), "t": now}
            infos.append(tuple(per_agent))
        return obs, rews, done, tuple(infos)

    def close_extras(self):
        for E in self.engines:
            E.close()

    def _assert_not_closed(self):
        assert not self.closed, "Trying to operate on a SumoVecEnv after calling close()"

    def __del__(self):
        if not self.closed:
            try:
                self.close()
            except Exception:
                pass


def make_vec_env(env_id, num_env, seed, device=0, **kw):
    """Counterpart of reference run.py:124-146 ``build_env``: ``num_env`` envs seeded ``seed + i``."""
    return SumoVecEnv(env_id, num_envs=num_env, seed=seed, device=device, **kw)
