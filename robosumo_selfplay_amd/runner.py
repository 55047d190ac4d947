"""Rollout engine with the reference ``Runner``'s constructor and 15-tuple ``run(update)`` contract (runner.py:7-252).

Two modes, same arithmetic (csrc/ppo_kernels.hip: reward curriculum, IS ratios, V-trace):
  * device mode  -- ``env`` is a ``SumoVecEnv`` and the models are ``PPOModel``s: observations, actions, rewards and
    dones never leave HBM between the env step kernel and the policy kernels; ``run`` returns CUDA tensors.
  * host mode    -- any duck-typed VecEnv / models like the reference's: numpy per step (their API), then the collected
    buffers go through the same kernels; ``run`` returns numpy arrays (this is what the golden-vector tests drive).
"""
import os

import numpy as np

from . import ppo_capi


def sf01(arr):
    """runner.py:255-260 (numpy or torch)"""
    s = arr.shape
    return arr.swapaxes(1, 2).reshape(s[0], s[1] * s[2], *s[3:])


def sf0(arr):
    """runner.py:263-267"""
    return arr.swapaxes(0, 1).reshape(-1)


def anneal_alpha(update, anneal_bound):
    """runner.py:128-130"""
    if update <= anneal_bound:
        return float(np.linspace(1, 0, anneal_bound)[update - 1])
    return 0.0


class EpInfoList(object):
    """The rollout's episode records of agent 0 (monitor.py:63-78: ``{'r', 'l', 't'}`` per finished episode, in (step, env) order) as
    a read-only sequence that builds its dicts on demand: a 4096-env rollout ends ~10 000 episodes, the driver looks at the last
    100 (``epinfobuf = deque(maxlen=100)``, alg_ppo.py:160), and 10 000 Python dicts per update cost 8 ms of a 270 ms iteration.
    Behaves like the reference's list for ``len`` / iteration / indexing / slicing / ``deque.extend`` / ``==``."""

    def __init__(self, r, l):
        self._r = np.round(np.asarray(r, np.float64), 6)
        self._l = np.asarray(l, np.int64)

    def __len__(self):
        return int(self._r.shape[0])

    def _make(self, i):
        return {"r": float(self._r[i]), "l": int(self._l[i]), "t": 0.0}

    def __getitem__(self, i):
        if isinstance(i, slice):
            return EpInfoList(self._r[i], self._l[i])
        return self._make(range(len(self))[i])

    def __iter__(self):
        return (self._make(i) for i in range(len(self)))

    def __eq__(self, other):
        if isinstance(other, EpInfoList):
            return np.array_equal(self._r, other._r) and np.array_equal(self._l, other._l)
        return list(self) == list(other)

    def __repr__(self):
        return "EpInfoList(%d episodes)" % len(self)


class _nullctx(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class AbstractEnvRunner(object):
    def __init__(self, *, env, models, nsteps, nagent, anneal_bound):
        self.env, self.models = env, models
        self.nenv = env.num_envs
        self.nagent = nagent
        self.nsteps = nsteps
        self.anneal_bound = anneal_bound
        self.states = [m.initial_state for m in models]


class Runner(AbstractEnvRunner):
    def __init__(self, *, env, models, nsteps, nagent, gamma, lam, rho_bar, c_bar, anneal_bound=500, device=None):
        super().__init__(env=env, models=models, nsteps=nsteps, nagent=nagent, anneal_bound=anneal_bound)
        import torch
        if not torch.cuda.is_available():
            raise ppo_capi.PpoHipError("Runner needs a HIP device (no CPU fallback in the product path)")
        ppo_capi.lib()
        self._t = torch
        if nagent != 2:
            raise ValueError("two-agent self-play only (runner.py assumes agents 0 and 1)")
        self.lam, self.gamma, self.rho_bar, self.c_bar = lam, gamma, rho_bar, c_bar
        self._side = None
        self._gstreams = None
        self._gsides = None
        # device mode, MLP policies: the whole step loop of a rollout runs inside the env engine's fused rollout launch
        # (sumo_rollout_steps); SUMO_FUSED_ROLLOUT=0 keeps the step-by-step launches (same numbers, bit for bit)
        self.fused_rollout = os.environ.get("SUMO_FUSED_ROLLOUT", "1") != "0"
        self.rollout_chunk = int(os.environ.get("SUMO_ROLLOUT_CHUNK", "0"))   # steps per launch (0 = the whole rollout)
        self.opponent_pool = None     # opponent_pool.OpponentPool: frozen snapshots + per-env snapshot index (fused path)
        self.recurrent = all(getattr(m, "recurrent", False) for m in models)
        self.device_mode = hasattr(env, "step_device") and (self.recurrent or all(
            hasattr(m, "act_model") and hasattr(m.act_model, "evaluate") for m in models))
        self.device = getattr(env, "device", None) or device or torch.device("cuda", 0)
        ob_shape = env.observation_space[0].shape
        self.ob_dim = ob_shape[0]
        if len(env.observation_space) > 1 and (env.observation_space[1].shape != ob_shape or
                                               env.action_space[1].shape != env.action_space[0].shape):
            # the reference sizes its buffers and both policies from observation_space[0] as well (runner.py:14-16)
            raise ValueError("self-play rollouts need both agents to share observation and action spaces; mixed match-ups "
                             "(%s vs %s) run in the env engine and the evaluator only"
                             % (ob_shape, env.observation_space[1].shape))
        if self.device_mode:
            self.obs = env.reset_device()                               # [N, 2, D] float32, stays in HBM
            self.dones = torch.zeros((self.nenv, 2), dtype=torch.uint8, device=self.device)
            if self.recurrent:                                          # recurrent states live on the device as well
                self.states = [torch.as_tensor(np.asarray(st, np.float32)).to(self.device) for st in self.states]
        else:
            self.obs = np.zeros((self.nenv, len(env.observation_space)) + ob_shape,
                                dtype=models[0].train_model.X.dtype.name)
            self.obs[:] = env.reset()
            self.dones = np.array([[False for _ in range(nagent)] for _ in range(self.nenv)])

    # ---- shared tail: IS ratios + V-trace on the device ---------------------------------------------------------
    def _vtrace(self, rewards, values, nlp, onlp, dones, last_dones, last_values):
        t = self._t
        T, N = self.nsteps, self.nenv
        returns = t.empty((2, T, N), dtype=t.float32, device=self.device)
        opr = t.empty((T, N), dtype=t.float32, device=self.device)
        oer = t.empty_like(opr)
        ratio = t.empty_like(opr)
        st = t.cuda.current_stream(self.device).cuda_stream
        ppo_capi.chk(ppo_capi.lib().ppo_vtrace(rewards.data_ptr(), values.data_ptr(), nlp.data_ptr(), onlp.data_ptr(), dones.data_ptr(),
                                               last_dones.data_ptr(), last_values.data_ptr(), T, N, float(self.gamma), float(self.lam),
                                               float(self.rho_bar), float(self.c_bar), returns.data_ptr(), opr.data_ptr(),
                                               oer.data_ptr(), ratio.data_ptr(), st))
        return returns, opr, oer, ratio

    def run(self, update):
        return self._run_device(update) if self.device_mode else self._run_host(update)

    # ---- device mode ------------------------------------------------------------------------------------------
    def _alloc_device(self, T):
        """Rollout buffers [agent, time, env, ...] in HBM."""
        t = self._t
        N, D, dev = self.nenv, self.ob_dim, self.device
        A = self.env.action_space[0].shape[0]
        f32 = t.float32
        B = dict(T=T, A=A)
        B["obs"] = t.empty((2, T, N, D), dtype=f32, device=dev)
        B["act"] = t.empty((2, T, N, A), dtype=f32, device=dev)
        for k in ("rew", "val", "nlp", "onlp"):
            B[k] = t.empty((2, T, N), dtype=f32, device=dev)
        B["done"] = t.empty((2, T, N), dtype=t.uint8, device=dev)
        B["ep_done"] = t.empty((T, N), dtype=t.uint8, device=dev)
        B["ep_r"] = t.empty((T, N), dtype=t.float64, device=dev)
        B["ep_l"] = t.empty((T, N), dtype=t.int32, device=dev)
        B["scratch_a"] = t.empty((N, A), dtype=f32, device=dev)
        B["scratch_b"] = t.empty((N, A), dtype=f32, device=dev)
        return B

    def _step_device(self, B, s, alpha, env_events=None):
        """One rollout step (runner.py:62-151): 4 fused policy launches (the reference's 5 evaluations), env step, reward mix.
        With env groups (``SumoVecEnv(groups=G)``) every group runs this sequence on its own stream: a group's next step only
        waits for that group's envs, not for the slowest env of the whole batch (``join_groups`` before reading the buffers)."""
        t = self._t
        G = getattr(self.env, "groups", 1)
        if G == 1:
            self._step_group(B, s, alpha, None, slice(0, self.nenv), env_events)
            return
        if self._gstreams is None:
            cur = t.cuda.current_stream(self.device)
            self._gstreams = [t.cuda.Stream(device=self.device) for _ in range(G)]
            self._gsides = [None] * G
            for st in self._gstreams:
                st.wait_stream(cur)
        for g in range(G):
            with t.cuda.stream(self._gstreams[g]):
                self._step_group(B, s, alpha, g, self.env._gs(g), env_events if g == 0 else None)

    def _group_streams(self):
        t = self._t
        G = getattr(self.env, "groups", 1)
        if G > 1 and self._gstreams is None:
            cur = t.cuda.current_stream(self.device)
            self._gstreams = [t.cuda.Stream(device=self.device) for _ in range(G)]
            self._gsides = [None] * G
            for st in self._gstreams:
                st.wait_stream(cur)
        return self._gstreams

    def _draw_steps(self, gen, T, n, A):
        """Action noise [T, n, A] as T consecutive [n, A] draws of ``gen``: what ``step`` would draw call by call (so a one-group
        device rollout consumes the generator exactly like the reference-order host rollout)."""
        t = self._t
        buf = t.empty((T, n, A), dtype=t.float32, device=self.device)
        for k in range(T):
            t.randn((n, A), generator=gen, device=self.device, dtype=t.float32, out=buf[k])
        return buf

    def fused_lstm_ok(self):
        """The fused recurrent launch (``sumo_rollout_steps_lstm``) applies to device mode with the nets ``learn(network='lstm')``
        trains on both sides -- an ``LstmPPOModel`` learner against an ``LstmPPOModel`` or an ``LstmOpponentPool`` -- with
        nlstm = 128 and the env's own observation / action shape."""
        from .lstm_model import LstmPPOModel
        from .opponent_pool import LstmOpponentPool
        env = self.env
        if not (self.device_mode and self.fused_rollout and self.recurrent and hasattr(env, "rollout_steps_lstm_group")):
            return False
        if getattr(env, "cfrc_mode", "zero") != "zero":      # the force entries are filled in by a second launch per step
            return False
        m0, m1 = self.models[0], self.models[1]
        if type(m0) is not LstmPPOModel or type(m1) not in (LstmPPOModel, LstmOpponentPool):
            return False
        sp0, sp1 = m0.spec, m1.spec
        if isinstance(m1, LstmOpponentPool) and (m1.num_envs != self.nenv or any((env._gs(g).start % 16 or env._gs(g).stop % 16)
                                                                                 for g in range(getattr(env, "groups", 1)))):
            return False
        return (sp0.nlstm == sp1.nlstm == 128 and sp0.ob_dim == sp1.ob_dim == self.ob_dim and sp0.ac_dim == sp1.ac_dim
                and env.act_dev.shape[2] == sp0.ac_dim and env.obs_dev.stride(2) == 1)

    def _steps_fused_lstm(self, B, s0, K, alpha):
        """``_steps_fused`` for recurrent policies: one ``sumo_rollout_steps_lstm`` launch per env group; the recurrent states
        ``self.states`` are advanced in place.  Same numbers as the step-by-step recurrent branch of ``_step_group``."""
        import ctypes as C
        from . import capi
        from .opponent_pool import LstmOpponentPool
        t = self._t
        env = self.env
        m0, m1 = self.models
        G = getattr(env, "groups", 1)
        streams = self._group_streams() if G > 1 else [None]
        A, T, N = m0.spec.ac_dim, B["T"], self.nenv
        for g in range(G):
            sl = env._gs(g)
            n = sl.stop - sl.start
            ctx = t.cuda.stream(streams[g]) if streams[g] is not None else _nullctx()
            with ctx:
                key = ("noise", sl.start)
                if s0 == 0 or key not in B:
                    B[key] = (self._draw_steps(m0.gen, T, n, A), self._draw_steps(m1.gen, T, n, A))
                ro = capi.RolloutLstm()
                ro.learner = C.addressof(m0._net)
                if isinstance(m1, LstmOpponentPool):
                    ro.opponents_dev, ro.tile_net_dev, ro.npool = m1._nets_dev.data_ptr(), m1.tile_net.data_ptr(), m1.capacity
                else:
                    ro.opponents_dev, ro.tile_net_dev, ro.npool = m1.net_dev().data_ptr(), None, 1
                ro.state0, ro.state1 = self.states[0][sl].data_ptr(), self.states[1][sl].data_ptr()
                ro.T, ro.Ntot, ro.env_offset, ro.s0, ro.K, ro.alpha = T, N, sl.start, int(s0), int(K), float(alpha)
                ro.noise0, ro.noise1 = B[key][0].data_ptr(), B[key][1].data_ptr()
                for f in ("obs", "act", "rew", "val", "nlp", "onlp", "done", "ep_done", "ep_r", "ep_l"):
                    setattr(ro, f, B[f].data_ptr())
                env.rollout_steps_lstm_group(g, ro)
        self.obs, self.dones = env.obs_dev, env.done_dev

    def fused_ok(self):
        """The fused rollout launch applies to device mode with two plain MLP policies of the env's own observation / action shape."""
        if not (self.device_mode and self.fused_rollout and not self.recurrent and hasattr(self.env, "rollout_steps_group")):
            return False
        if getattr(self.env, "cfrc_mode", "zero") != "zero":
            return False
        m0, m1 = self.models[0], self.models[1]
        if not (hasattr(m0, "act_model") and hasattr(m1, "act_model")):
            return False
        return self._fused_ok(m0.act_model, m1.act_model)

    def _steps_fused(self, B, s0, K, alpha):
        """Rollout steps s0 .. s0+K-1 of every env group, one launch per group (``sumo_rollout_steps``): the five evaluations of
        runner.py:62-96, env.step, the reward curriculum and the appends to ``B`` all happen inside the env engine.  Noise comes
        from each acting model's own generator, drawn per group for the whole buffer when its first step is written -- the same
        draws as the step-by-step path, hence the same rollout bit for bit."""
        from . import capi
        t = self._t
        env = self.env
        learner, opp = self.models[0].act_model, self.models[1].act_model
        G = getattr(env, "groups", 1)
        streams = self._group_streams() if G > 1 else [None]
        pool = self.opponent_pool
        D, A, T, N = learner.spec.ob_dim, learner.spec.ac_dim, B["T"], self.nenv
        for g in range(G):
            sl = env._gs(g)
            n = sl.stop - sl.start
            ctx = t.cuda.stream(streams[g]) if streams[g] is not None else _nullctx()
            with ctx:
                key = ("noise", sl.start)
                if s0 == 0 or key not in B:
                    B[key] = (t.randn((T, n, A), generator=learner.gen, device=self.device, dtype=t.float32),
                              t.randn((T, n, A), generator=opp.gen, device=self.device, dtype=t.float32))
                ro = capi.Rollout()
                ro.learner_params = learner.params.data_ptr()
                if pool is None:
                    ro.opponent_params, ro.opponent_index, ro.npool = opp.params.data_ptr(), None, 1
                else:
                    ro.opponent_params, ro.opponent_index, ro.npool = pool.params.data_ptr(), pool.index[sl].data_ptr(), pool.capacity
                ro.ob_dim, ro.ac_dim, ro.T, ro.Ntot, ro.env_offset, ro.s0, ro.K, ro.alpha = D, A, T, N, sl.start, int(s0), int(K), float(alpha)
                ro.noise0, ro.noise1 = B[key][0].data_ptr(), B[key][1].data_ptr()
                for f in ("obs", "act", "rew", "val", "nlp", "onlp", "done", "ep_done", "ep_r", "ep_l"):
                    setattr(ro, f, B[f].data_ptr())
                env.rollout_steps_group(g, ro)
        self.obs, self.dones = env.obs_dev, env.done_dev

    def join_groups(self):
        """Make the current stream wait for every env group's stream (no-op without groups)."""
        if self._gstreams is not None:
            cur = self._t.cuda.current_stream(self.device)
            for st in self._gstreams:
                cur.wait_stream(st)

    def _step_group(self, B, s, alpha, g, sl, env_events):
        t = self._t
        env = self.env
        n = sl.stop - sl.start
        learner, opp = self.models[0].act_model, self.models[1].act_model
        ob, dn = env.obs_dev[sl], env.done_dev[sl]                      # [n, 2, D] / [n, 2]: the env's own buffers
        fused = self._fused_ok(learner, opp)
        if fused:
            self._selfplay_forward(B, s, learner, opp, ob, dn, sl)      # records obs / dones, evaluates, fills env.act_dev
        else:
            B["obs"][:, s, sl].copy_(ob.permute(1, 0, 2))               # one strided copy per array instead of one per agent
            B["done"][:, s, sl].copy_(dn.t())
        o0, o1 = B["obs"][0, s, sl], B["obs"][1, s, sl]
        act0, act1 = B["act"][0, s, sl], B["act"][1, s, sl]
        if fused:
            pass
        elif self.recurrent:
            # same five evaluations through the recurrent nets (runner.py:62-96 with the S / M feeds): each stream carries its
            # acting model's state; the scoring calls without a state feed start from zeros, as the reference's calls do
            m0, m1 = self.models
            nk0, nk1 = {}, {}
            if getattr(m0, "accepts_noise", False) and getattr(m1, "accepts_noise", False):
                # action noise of this group for the whole buffer, drawn at its first step from each acting model's generator
                # (as the MLP path does): the fused recurrent launch consumes the same draws
                key = ("noise", sl.start)
                if s == 0 or key not in B:
                    B[key] = (self._draw_steps(m0.gen, B["T"], n, act0.shape[-1]), self._draw_steps(m1.gen, B["T"], n, act0.shape[-1]))
                nk0, nk1 = dict(noise=B[key][0][s]), dict(noise=B[key][1][s])
            a0, v0, S0, n0 = m0.step(o0, S=self.states[0][sl], M=dn[:, 0], **nk0)
            self.states[0][sl] = S0
            act0.copy_(a0); B["val"][0, s, sl].copy_(v0); B["nlp"][0, s, sl].copy_(n0)
            pk = dict(first_env=sl.start) if hasattr(m1, "tile_net") else {}      # opponent pool: snapshots are indexed per env tile
            B["onlp"][0, s, sl].copy_(m1.act_model.action_probability(o0, given_action=a0, **pk))
            a1, _, S1, on1 = m1.step(o1, S=self.states[1][sl], M=dn[:, 1], **pk, **nk1)
            self.states[1][sl] = S1
            act1.copy_(a1); B["onlp"][1, s, sl].copy_(on1)
            B["val"][1, s, sl].copy_(m0.value(o1, S=S1, M=dn[:, 1]))
            B["nlp"][1, s, sl].copy_(m0.act_model.action_probability(o1, given_action=a1))
        else:
            self._policy_evals(B, s, learner, opp, o0, o1, sl, g)
        act = env.act_dev
        if not fused:
            act[sl].copy_(B["act"][:, s, sl].permute(1, 0, 2))
        if env_events is not None:
            env_events[0].record()
        if g is None:
            env.step_device(act)
        else:
            env.step_device_group(g, act)
        if env_events is not None:
            env_events[1].record()
        st = t.cuda.current_stream(self.device).cuda_stream
        # reward mix (runner.py:134) + the step's episode records, one launch
        ppo_capi.chk(ppo_capi.lib().ppo_post_step(env.info_dev[sl].data_ptr(), n, alpha, B["rew"][0, s, sl].data_ptr(), B["T"] * self.nenv,
                                                  env.done_dev[sl].data_ptr(), env.ep_r_dev[sl].data_ptr(), env.ep_l_dev[sl].data_ptr(),
                                                  B["ep_done"][s, sl].data_ptr(), B["ep_r"][s, sl].data_ptr(),
                                                  B["ep_l"][s, sl].data_ptr(), st))
        self.obs, self.dones = env.obs_dev, env.done_dev

    def _fused_ok(self, learner, opp):
        """One-launch evaluation (``ppo_selfplay_forward``) applies when both sides are plain MLP policies of the same shape."""
        from .policies import PolicyWithValue
        env = self.env
        return (not self.recurrent and type(learner) is PolicyWithValue and type(opp) is PolicyWithValue
                and learner.spec.ob_dim == opp.spec.ob_dim == self.ob_dim and learner.spec.ac_dim == opp.spec.ac_dim
                and env.act_dev.shape[2] == learner.spec.ac_dim and env.obs_dev.stride(2) == 1)

    def _selfplay_forward(self, B, s, learner, opp, ob, dn, sl):
        """runner.py:62-96 in one launch: learner acts on agent 0 (opponent scores it), opponent acts on agent 1 (learner scores
        it and evaluates the value); the kernel also records the observations / done flags and writes the env's action buffer.
        Noise comes from each acting model's own generator, as in ``PolicyWithValue.evaluate``."""
        import ctypes as C
        t = self._t
        n = sl.stop - sl.start
        D, A = learner.spec.ob_dim, learner.spec.ac_dim
        # action noise of this group for the whole buffer, drawn at its first step (two launches per rollout instead of two per
        # step between the env steps); each acting model's own generator, as in ``PolicyWithValue.evaluate``
        key = ("noise", sl.start)
        if s == 0 or key not in B:
            B[key] = (t.randn((B["T"], n, A), generator=learner.gen, device=self.device, dtype=t.float32),
                      t.randn((B["T"], n, A), generator=opp.gen, device=self.device, dtype=t.float32))
        noise0, noise1 = B[key][0][s], B[key][1][s]
        outs = [B["obs"][0, s, sl], B["obs"][1, s, sl], B["act"][0, s, sl], B["act"][1, s, sl], B["nlp"][0, s, sl], B["nlp"][1, s, sl],
                B["onlp"][0, s, sl], B["onlp"][1, s, sl], B["val"][0, s, sl], B["val"][1, s, sl]]
        fp = (C.c_void_p * 10)(*[x.data_ptr() for x in outs])
        dp = (C.c_void_p * 2)(B["done"][0, s, sl].data_ptr(), B["done"][1, s, sl].data_ptr())
        ppo_capi.chk(ppo_capi.lib().ppo_selfplay_forward(learner.params.data_ptr(), opp.params.data_ptr(), ob.data_ptr(), n, ob.stride(0),
                                                         ob.stride(1), D, A, noise0.data_ptr(), noise1.data_ptr(), dn.data_ptr(),
                                                         self.env.act_dev[sl].data_ptr(), fp, dp,
                                                         t.cuda.current_stream(self.device).cuda_stream))

    def _policy_evals(self, B, s, learner, opp, o0, o1, sl, g):
        """The two chains (learner acts on agent 0's stream and the opponent scores it; the opponent acts on agent 1's stream
        and the learner evaluates it) are independent: they run on two HIP streams and join before the env step."""
        t = self._t
        PI, VF = ppo_capi.FWD_PI, ppo_capi.FWD_VF
        sa, sb = B["scratch_a"][sl], B["scratch_b"][sl]
        if g is not None:
            # env groups already overlap with each other; more streams than hardware queues (4 by default) only serialise
            learner.evaluate(o0, PI | VF, out=dict(action=B["act"][0, s, sl], neglogp=B["nlp"][0, s, sl], value=B["val"][0, s, sl]))
            opp.evaluate(o0, PI, given_action=B["act"][0, s, sl], out=dict(neglogp=B["onlp"][0, s, sl], action=sa))
            opp.evaluate(o1, PI, out=dict(action=B["act"][1, s, sl], neglogp=B["onlp"][1, s, sl]))
            learner.evaluate(o1, PI | VF, given_action=B["act"][1, s, sl],
                             out=dict(neglogp=B["nlp"][1, s, sl], value=B["val"][1, s, sl], action=sb))
            return
        cur = t.cuda.current_stream(self.device)
        if self._side is None:
            self._side = t.cuda.Stream(device=self.device)
        side = self._side
        side.wait_stream(cur)
        # agent 0 acts with the learner; the opponent net scores that action (runner.py:67-85)
        learner.evaluate(o0, PI | VF, out=dict(action=B["act"][0, s, sl], neglogp=B["nlp"][0, s, sl], value=B["val"][0, s, sl]))
        opp.evaluate(o0, PI, given_action=B["act"][0, s, sl], out=dict(neglogp=B["onlp"][0, s, sl], action=sa))
        with t.cuda.stream(side):
            # agent 1 acts with the opponent; the learner net evaluates value and neglogp there (runner.py:86-96)
            opp.evaluate(o1, PI, out=dict(action=B["act"][1, s, sl], neglogp=B["onlp"][1, s, sl]))
            learner.evaluate(o1, PI | VF, given_action=B["act"][1, s, sl],
                             out=dict(neglogp=B["nlp"][1, s, sl], value=B["val"][1, s, sl], action=sb))
        cur.wait_stream(side)

    def _run_device(self, update):
        t = self._t
        T, N = self.nsteps, self.nenv
        B = self._alloc_device(T)
        alpha = anneal_alpha(update, self.anneal_bound)
        states0 = self.states[0].clone() if self.recurrent else None     # BPTT starts from the rollout's initial state
        if self._gstreams is not None:                                   # the group streams see the parameter update that preceded this rollout
            cur = t.cuda.current_stream(self.device)
            for st in self._gstreams:
                st.wait_stream(cur)
        if self.fused_ok() or self.fused_lstm_ok():
            steps = self._steps_fused_lstm if self.recurrent else self._steps_fused
            chunk = self.rollout_chunk if self.rollout_chunk > 0 else T
            for s0 in range(0, T, chunk):
                steps(B, s0, min(chunk, T - s0), alpha)
        else:
            if self.opponent_pool is not None:
                raise NotImplementedError("a per-env opponent pool needs the fused rollout path (MLP policies, SUMO_FUSED_ROLLOUT != 0)")
            for s in range(T):
                self._step_device(B, s, alpha)
        self.join_groups()
        learner = self.models[0].act_model
        last_values = t.empty((2, N), dtype=t.float32, device=self.device)
        if self.recurrent:
            for g in range(2):
                last_values[g].copy_(self.models[0].value(self.obs[:, g, :], S=self.states[g], M=self.dones[:, g]))
        else:
            learner.evaluate(self.obs[:, 0, :], ppo_capi.FWD_VF, out=dict(value=last_values[0]))   # runner.py:184: always models[0]
            learner.evaluate(self.obs[:, 1, :], ppo_capi.FWD_VF, out=dict(value=last_values[1]))
        returns, opr, oer, ratio = self._vtrace(B["rew"], B["val"], B["nlp"], B["onlp"], B["done"], self.dones.contiguous(), last_values)
        # A fused launch that was cut short (expired hand-over wait, hand-over tag / checksum mismatch) leaves unwritten rollout rows:
        # fail as loudly as a MuJoCo fault does in the reference (mujoco-py builder.py:351-369 raises out of env.step) instead of
        # training on them.  The call waits for the launch; this function synchronises right below anyway.
        if self.fused_ok() or self.fused_lstm_ok():
            for E in self.env.engines:
                E.rollout_status()
        # episode infos of agent 0 (monitor.py:63-78), harvested with one host sync per rollout
        d = B["ep_done"].cpu().numpy().astype(bool)
        rr, ll = B["ep_r"].cpu().numpy(), B["ep_l"].cpu().numpy()
        epinfos = EpInfoList(rr[d], ll[d])          # (step, env) order of np.nonzero, dicts built on demand
        return (sf01(B["obs"]), sf01(returns), sf01(B["done"].bool()), sf01(B["act"]), sf01(B["val"]), sf01(B["nlp"]), sf01(B["rew"]),
                sf01(B["onlp"]), sf01(B["obs"][1]), sf01(B["act"][1]), states0, epinfos, sf0(opr), sf0(oer), sf0(ratio))

    # ---- host mode ----------------------------------------------------------------------------------------------
    def _run_host(self, update):
        t = self._t
        T, N, A_ = self.nsteps, self.nenv, self.nagent
        mb_obs = [[] for _ in range(A_)]
        mb_actions = [[] for _ in range(A_)]
        mb_values = [[] for _ in range(A_)]
        mb_dones = [[] for _ in range(A_)]
        mb_nlp = [[] for _ in range(A_)]
        mb_onlp = [[] for _ in range(A_)]
        opp_obs, opp_act, epinfos = [], [], []
        info_steps, env_rew_steps = [], []
        use_info = None
        # recurrent nets: the state at the START of the rollout is what back-propagation through time begins from (the
        # reference returns the list slot after the loop, i.e. the final state -- its recurrent training path is dead code)
        mb_states0 = None if self.states[0] is None else np.array(self.states[0], copy=True)
        for _ in range(T):
            acts = []
            for agt in range(A_):
                o = self.obs[:, agt, :]
                a, v, self.states[agt], nlp = self.models[agt].step(o, S=self.states[agt], M=self.dones[:, agt])
                mb_obs[agt].append(o.copy())
                mb_actions[agt].append(a)
                mb_dones[agt].append(self.dones[:, agt])
                if agt == 0:
                    mb_values[0].append(v)
                    mb_nlp[0].append(nlp)
                    mb_onlp[0].append(self.models[1].act_model.action_probability(o, given_action=a))
                else:
                    mb_onlp[agt].append(nlp)
                    mb_values[agt].append(self.models[0].value(o, S=self.states[agt], M=self.dones[:, agt]))
                    mb_nlp[agt].append(self.models[0].act_model.action_probability(o, given_action=a))
                    opp_obs.append(self.obs[:, 1, :].copy())
                    opp_act.append(a)
                acts.append(a)
            self.obs[:], rewards, self.dones, infos = self.env.step(np.stack(acts, axis=1))
            if use_info is None:
                use_info = "shaping_reward" in infos[0][0]                # runner.py:127
            if use_info:
                arr = np.zeros((N, 2, 8))
                for e in range(N):
                    for agt in range(A_):
                        arr[e, agt, 6] = infos[e][agt]["shaping_reward"]
                        arr[e, agt, 3] = infos[e][agt]["main_reward"]
                info_steps.append(arr)
            else:
                env_rew_steps.append(np.asarray(rewards))
            for e in range(N):
                ep = infos[e][0].get("episode")
                if ep:
                    epinfos.append(ep)
        dev, f32 = self.device, t.float32
        up = lambda x, dt=np.float32: t.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=dt))).to(dev)
        mb_obs = np.asarray(mb_obs, dtype=self.obs.dtype)
        mb_actions = np.asarray(mb_actions)
        opp_obs = np.asarray(opp_obs, dtype=self.obs.dtype)
        opp_act = np.asarray(opp_act)
        mb_dones_np = np.asarray(mb_dones, dtype=bool)
        rew = t.empty((2, T, N), dtype=f32, device=dev)
        if use_info:
            alpha = anneal_alpha(update, self.anneal_bound)
            st = t.cuda.current_stream(dev).cuda_stream
            info_dev = up(np.stack(info_steps), np.float64)               # [T, N, 2, 8]
            for s in range(T):
                ppo_capi.chk(ppo_capi.lib().ppo_reward_mix(info_dev[s].data_ptr(), N, alpha, rew[0, s].data_ptr(), T * N, st))
        else:
            rew.copy_(up(np.stack(env_rew_steps).transpose(2, 0, 1)))    # runner.py:146: rewards[:, agt], cast to float32
        val, nlp, onlp = up(mb_values), up(mb_nlp), up(mb_onlp)
        last_values = up(np.stack([self.models[0].value(self.obs[:, agt, :], S=self.states[agt], M=self.dones[:, agt])
                                   for agt in range(A_)]))
        returns, opr, oer, ratio = self._vtrace(rew, val, nlp, onlp, up(mb_dones_np, np.uint8), up(self.dones, np.uint8),
                                                last_values)
        n = lambda x: x.cpu().numpy()
        mb_onlp_np = np.asarray(mb_onlp)
        return (*map(sf01, (mb_obs, n(returns), mb_dones_np, mb_actions, np.asarray(mb_values, np.float32),
                            np.asarray(mb_nlp, np.float32), n(rew), mb_onlp_np, opp_obs, opp_act)),
                mb_states0, epinfos, *map(lambda x: sf0(n(x)), (opr, oer, ratio)))
