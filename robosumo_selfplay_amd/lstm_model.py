"""Recurrent PPO model: baselines ``lstm(nlstm)`` policy (shared latent, two heads) trained by back-propagation through
time on the HIP kernels.

Reference surfaces mirrored: ``PPOModel`` (model.py:9-213: ``step`` / ``value`` / ``train(..., states)`` / ``save`` /
``load``), the recurrent network builder baselines/baselines/common/models.py:131-183 + a2c/utils.py:82-103 and the
recurrent minibatch convention of baselines ppo2 (whole env sequences per minibatch, env-major flattening; the reference
fork's own copy of that loop, alg_ppo.py:408-421, passes the states as ``IS_weight`` and cannot run -- SURVEY.md App. C.3).

Kernels (include/sumo_ppo.h): ``ppo_lstm_step`` (acting), ``ppo_lstm_step_save`` (unrolled forward that records gates),
``ppo_lstm_head_grad`` (loss + head gradients on the stored latents), ``ppo_lstm_bwd_step`` (one BPTT step; nlstm = 128: the whole
sequence in one launch each way, ``ppo_lstm_seq_forward`` / ``ppo_lstm_seq_backward``), the weight gradients by the hand-written
split-K MFMA kernel ``ppo_lstm_wgrad`` (``SUMO_LSTM_WGRAD=blas`` keeps ``torch.matmul`` = hipBLASLt as a cross-check only) and
``ppo_clip_adam``.
"""
import ctypes as C
import os

import numpy as np

from . import dist as sdist, hostcfg, policies, ppo_capi


class LstmSpec(object):
    def __init__(self, ob_dim, ac_dim, nlstm=128):
        if nlstm not in (64, 128):
            raise NotImplementedError("nlstm must be 64 or 128")
        self.ob_dim, self.ac_dim, self.nlstm = int(ob_dim), int(ac_dim), int(nlstm)


class LstmPPOModel(object):
    loss_names = ["policy_loss", "value_loss", "policy_entropy", "approxkl", "clipfrac"]
    recurrent = True
    accepts_noise = True  # step(..., noise=rows) takes the action noise from the caller (device-mode Runner: one draw per rollout)
    use_graph = True      # single-GPU training on device tensors: the ~2T+12 launches of a minibatch step replay from a HIP graph

    def __init__(self, *, policy, ob_space=None, ac_space=None, nbatch_act=None, nbatch_train=None, nsteps=None, ent_coef=0.0,
                 vf_coef=0.5, max_grad_norm=0.5, microbatch_size=None, trainable=True, model_scope="", device=0, comm=None):
        import torch
        if not torch.cuda.is_available():
            raise ppo_capi.PpoHipError("LstmPPOModel needs a HIP device (no CPU fallback in the product path)")
        self.comm = comm if trainable else None      # torch.distributed group: every rank trains on its own env sequences
        self._t = torch
        self.spec, self.scope, self.sess = policy, model_scope, None
        self.device = torch.device("cuda", int(device))
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), max_grad_norm
        self.trainable, self.nsteps, self.nenv = trainable, nsteps, nbatch_act
        D, A, H = policy.ob_dim, policy.ac_dim, policy.nlstm
        self.shapes = policies.lstm_param_shapes(D, A, H)
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.P = int(sum(self.sizes))
        flat = np.concatenate([np.asarray(p, np.float32).ravel() for p in policies.init_lstm_param_list(D, A, H)])
        self.params = torch.from_numpy(flat).to(self.device)
        self.views = self._split(self.params)
        self._net = self._make_net(self.views)
        self.gen = torch.Generator(device=self.device)
        self.act_model = self.train_model = self
        self.initial_state = None if nbatch_act is None else np.zeros((int(nbatch_act), 2 * H), np.float32)   # models.py:176
        if trainable:
            self.m = torch.zeros(self.P, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(self.P, dtype=torch.float32, device=self.device)
            self.t = 0
            self.grads = torch.zeros(self.P + ppo_capi.NSTATS, dtype=torch.float32, device=self.device)   # [flat grad | loss sums]
            self.gviews = self._split(self.grads)
            self.stats = torch.zeros(ppo_capi.NSTATS, dtype=torch.float64, device=self.device)
            self.moments = torch.zeros(3, dtype=torch.float64, device=self.device)
            self._graphs = {}
            self.wgrad_native = os.environ.get("SUMO_LSTM_WGRAD", "native") != "blas"
            self.seq_kernels = os.environ.get("SUMO_LSTM_SEQ", "1") != "0"    # whole-sequence forward / BPTT launches (nlstm 128)
            self.xproj = os.environ.get("SUMO_LSTM_XPROJ", "1") != "0"      # input block of the training forward hoisted out of the recurrence
            self.adv_ws = torch.zeros(ppo_capi.lib().ppo_adv_moments_workspace_bytes(), dtype=torch.uint8, device=self.device)
            self.wg_workspace = torch.empty(ppo_capi.lib().ppo_lstm_wgrad_workspace_bytes(D, H, A), dtype=torch.uint8, device=self.device)

    class _X:
        class dtype:
            name = "float32"
    X = _X()

    def _split(self, flat):
        out, o = [], 0
        for s, n in zip(self.shapes, self.sizes):
            out.append(flat[o:o + n].view(*s))
            o += n
        return out

    def _make_net(self, v):
        wx, wh, b, pw, pb, logstd, vw, vb = v
        n = ppo_capi.LstmNet()
        n.ob_dim, n.emb_dim, n.hidden, n.ac_dim = self.spec.ob_dim, 0, self.spec.nlstm, self.spec.ac_dim
        n.gate_order, n.forget_bias = ppo_capi.LSTM_GATES_IFOU, 0.0
        n.wx, n.wh, n.b = wx.data_ptr(), wh.data_ptr(), b.data_ptr()
        n.head_w, n.head_b, n.logstd, n.vf_w, n.vf_b = pw.data_ptr(), pb.data_ptr(), logstd.data_ptr(), vw.data_ptr(), vb.data_ptr()
        return n

    def seed(self, s):
        self.gen.manual_seed(int(s))

    # ---- checkpoints: list of the 8 tensors in TF variable order ---------------------------------------------------
    def get_param_list(self):
        return [v.cpu().numpy().copy() for v in self.views]

    def set_param_list(self, plist):
        assert len(plist) == len(self.shapes), "number of variables loaded mismatches len(variables)"
        for p, s in zip(plist, self.shapes):
            if tuple(np.shape(p)) != tuple(s):
                raise ValueError("parameter shape %s does not match %s" % (np.shape(p), s))
        self.params.copy_(self._t.from_numpy(np.concatenate([np.asarray(p, np.float32).ravel() for p in plist])))

    def save(self, save_path):
        import joblib
        d = os.path.dirname(save_path)
        if d:
            os.makedirs(d, exist_ok=True)
        joblib.dump(self.get_param_list(), save_path)

    def load(self, load_path):
        import joblib
        self.set_param_list(joblib.load(os.path.expanduser(load_path)))      # only files written by save()

    # ---- acting (policies.py:84-128 with the S / M feeds of models.py:163-170) --------------------------------------
    def _dev(self, x, dtype=np.float32):
        t = self._t
        return x if t.is_tensor(x) else t.from_numpy(np.ascontiguousarray(x, dtype)).to(self.device)

    def net_dev(self):
        """Device copy of this model's ``ppo_lstm_net`` (its pointers address ``self.params``, whose storage never moves): what the
        fused recurrent rollout reads when this model is the opponent."""
        if getattr(self, "_net_dev", None) is None:
            self._net_dev = self._t.from_numpy(np.frombuffer(bytes(self._net), dtype=np.uint8).copy()).to(self.device)
        return self._net_dev

    def _run(self, obs, S, M, given_action=None, deterministic=False, noise=None):
        t = self._t
        np_in = not t.is_tensor(obs)
        D, A, H = self.spec.ob_dim, self.spec.ac_dim, self.spec.nlstm
        x = self._dev(obs).reshape(-1, D)
        n = x.shape[0]
        st = t.zeros((n, 2 * H), dtype=t.float32, device=self.device) if S is None else self._dev(S).clone()
        mask = None if M is None else self._dev(np.asarray(M, np.float32) if not t.is_tensor(M) else M.to(t.float32))
        action = t.empty((n, A), dtype=t.float32, device=self.device)
        neglogp = t.empty(n, dtype=t.float32, device=self.device)
        value = t.empty(n, dtype=t.float32, device=self.device)
        given = None if given_action is None else self._dev(given_action).reshape(n, A).contiguous()
        if deterministic or given is not None:
            noise = None
        elif noise is None:                       # (the device-mode Runner hands in the rows of a per-rollout draw instead)
            noise = t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32)
        ppo_capi.chk(ppo_capi.lib().ppo_lstm_step(C.byref(self._net), x.data_ptr(), n, x.stride(0) if n > 1 else D, ppo_capi.ptr(mask),
                                                  st.data_ptr(), st.data_ptr() + 4 * H, 2 * H, ppo_capi.ptr(noise), ppo_capi.ptr(given),
                                                  action.data_ptr(), neglogp.data_ptr(), value.data_ptr(), None,
                                                  t.cuda.current_stream(self.device).cuda_stream))
        out = (lambda z: z.cpu().numpy()) if np_in else (lambda z: z)
        return out(action), out(value), out(st), out(neglogp)

    def step(self, observation, S=None, M=None, deterministic=False, noise=None, **extra_feed):
        return self._run(observation, S, M, deterministic=deterministic, noise=noise)

    def value(self, ob, S=None, M=None, **kwargs):
        return self._run(ob, S, M, deterministic=True)[1]

    def action_probability(self, observation, given_action=None, S=None, M=None, **extra_feed):
        """Without ``S`` the sequence start state (zeros) is used -- what the reference's Runner does for the opponent's
        likelihood of the learner's action (runner.py:85 passes no state)."""
        return self._run(observation, S, M, given_action=given_action)[3]

    # ---- training ---------------------------------------------------------------------------------------------------
    def loss_and_grads(self, cliprange, obs, returns, masks, actions, advs, neglogpacs, IS_weight, states, nsteps, world=1):
        """Forward over ``nsteps`` + BPTT for a minibatch of whole env sequences.  Flat inputs are env-major
        ([nenv_mb * nsteps, ...], baselines' ``sf01`` order); ``states`` [nenv_mb, 2H] is the state at the start of the
        sequences.  Leaves d(mean loss)/d(theta) in ``self.grads``, the un-normalised loss sums in ``self.stats``."""
        t = self._t
        L = ppo_capi.lib()
        D, A, H = self.spec.ob_dim, self.spec.ac_dim, self.spec.nlstm
        T = int(nsteps)
        f32 = t.float32
        tm = lambda x, *tail: self._dev(x).reshape(-1, T, *tail).transpose(0, 1).contiguous()     # env-major flat -> [T, n, ...]
        X, Mk, Ac = tm(obs, D), tm(np.asarray(masks, np.float32) if not t.is_tensor(masks) else masks.to(f32)), tm(actions, A)
        Ad, R, Old, W = tm(advs), tm(returns), tm(neglogpacs), tm(IS_weight)
        n = X.shape[1]
        rows = T * n
        st = t.cuda.current_stream(self.device).cuda_stream
        state = self._dev(states).reshape(n, 2 * H).clone()
        dev = self.device
        gates = t.empty((T, n, 4 * H), dtype=f32, device=dev)
        cprev, hprev, tanhc, lat = (t.empty((T, n, H), dtype=f32, device=dev) for _ in range(4))
        net = C.byref(self._net)
        # the input block x * wx of all T steps in one launch; the T sequential steps then only run the recurrent block on top of it
        # (same sums bit for bit, half the dependent chain per step).  The buffer is the one the backward sweep fills with dz later.
        dz = t.empty((T, n, 4 * H), dtype=f32, device=dev)
        if self.xproj:
            ppo_capi.chk(L.ppo_lstm_xproj(net, X.data_ptr(), rows, D, dz.data_ptr(), st))
        seq = self.xproj and self.seq_kernels and H == 128      # all T steps in one launch, recurrent weights resident in registers
        if seq:
            ppo_capi.chk(L.ppo_lstm_seq_forward(net, dz.data_ptr(), T, n, Mk.data_ptr(), state.data_ptr(), state.data_ptr() + 4 * H, 2 * H,
                                                gates.data_ptr(), cprev.data_ptr(), hprev.data_ptr(), tanhc.data_ptr(), lat.data_ptr(), st))
        for k in range(0 if seq else T):
            if self.xproj:
                ppo_capi.chk(L.ppo_lstm_step_save_z(net, dz[k].data_ptr(), n, Mk[k].data_ptr(), state.data_ptr(), state.data_ptr() + 4 * H, 2 * H,
                                                    gates[k].data_ptr(), cprev[k].data_ptr(), hprev[k].data_ptr(), tanhc[k].data_ptr(),
                                                    lat[k].data_ptr(), st))
            else:
                ppo_capi.chk(L.ppo_lstm_step_save(net, X[k].data_ptr(), n, D, Mk[k].data_ptr(), state.data_ptr(), state.data_ptr() + 4 * H,
                                                  2 * H, gates[k].data_ptr(), cprev[k].data_ptr(), hprev[k].data_ptr(), tanhc[k].data_ptr(), st))
                lat[k].copy_(state[:, H:])
        dlat = t.empty((T, n, H), dtype=f32, device=dev)
        dmean = t.empty((rows, A), dtype=f32, device=dev)
        dvalue = t.empty(rows, dtype=f32, device=dev)
        dls = t.empty((rows, A), dtype=f32, device=dev)
        self.stats.zero_()
        ppo_capi.chk(L.ppo_lstm_head_grad(net, lat.data_ptr(), rows, Ac.data_ptr(), Ad.data_ptr(), R.data_ptr(), Old.data_ptr(), W.data_ptr(),
                                          1.0 / (rows * world), float(cliprange), self.vf_coef, dlat.data_ptr(), dmean.data_ptr(), dvalue.data_ptr(),
                                          dls.data_ptr(), self.stats.data_ptr(), st))
        if seq:
            ppo_capi.chk(L.ppo_lstm_seq_backward(net, T, n, dlat.data_ptr(), Mk.data_ptr(), gates.data_ptr(), cprev.data_ptr(), tanhc.data_ptr(),
                                                 dz.data_ptr(), st))
        dh = t.zeros((n, H), dtype=f32, device=dev)
        dc = t.zeros((n, H), dtype=f32, device=dev)
        for k in range(-1 if seq else T - 1, -1, -1):
            ppo_capi.chk(L.ppo_lstm_bwd_step(net, n, dlat[k].data_ptr(), Mk[k].data_ptr(), gates[k].data_ptr(), cprev[k].data_ptr(),
                                             tanhc[k].data_ptr(), dh.data_ptr(), dc.data_ptr(), dz[k].data_ptr(), st))
        # weight gradients: [x | h_prev | 1]^T dz and [latent | 1]^T [dmean | dvalue | dlogstd rows] over all (time, env) rows, on the
        # split-K MFMA kernel (ppo_lstm_wgrad; SUMO_LSTM_WGRAD=blas keeps the library GEMMs for cross-checks)
        if self.wgrad_native:
            ppo_capi.chk(L.ppo_lstm_wgrad(net, rows, X.data_ptr(), hprev.data_ptr(), dz.data_ptr(), lat.data_ptr(), dmean.data_ptr(),
                                          dvalue.data_ptr(), dls.data_ptr(), -self.ent_coef / world, self.grads.data_ptr(),
                                          self.wg_workspace.data_ptr(), st))
        else:
            Z2, X2, H2, L2 = dz.view(rows, 4 * H), X.view(rows, D), hprev.view(rows, H), lat.view(rows, H)
            g = self.gviews
            t.matmul(X2.t(), Z2, out=g[0])
            t.matmul(H2.t(), Z2, out=g[1])
            t.sum(Z2, dim=0, out=g[2])
            t.matmul(L2.t(), dmean, out=g[3])
            t.sum(dmean, dim=0, out=g[4])
            g[5].copy_((dls.sum(dim=0) - self.ent_coef / world).view(1, A))     # the entropy term once in the global sum
            t.matmul(L2.t(), dvalue.view(rows, 1), out=g[6])
            g[7].copy_(dvalue.sum().view(1))
        return state

    def _loss_step(self, cliprange, obs, ret, val, masks, actions, neglogpacs, IS_weight, states, T, world):
        """Advantage normalisation (model.py:180-185) + loss_and_grads: everything of a minibatch step before the optimiser."""
        t = self._t
        L = ppo_capi.lib()
        nrow = ret.numel()
        st = t.cuda.current_stream(self.device).cuda_stream
        adv = t.empty(nrow, dtype=t.float32, device=self.device)
        ppo_capi.chk(L.ppo_adv_moments_ws(ret.data_ptr(), val.data_ptr(), None, nrow, self.moments.data_ptr(), self.adv_ws.data_ptr(), st))
        sdist.allreduce_moments(self.moments, self.comm)                                # global advantage normalisation
        ppo_capi.chk(L.ppo_adv_normalize(ret.data_ptr(), val.data_ptr(), None, nrow, self.moments.data_ptr(), adv.data_ptr(), st))
        self.loss_and_grads(cliprange, obs, ret, masks, actions, adv, neglogpacs, IS_weight, states, T, world=world)

    def _graph_step(self, cliprange, tensors, T):
        """Replay ``_loss_step`` from a HIP graph captured once per (minibatch shape, cliprange): the inputs are copied into
        buffers owned by the graph, the 2T+12 kernel launches and GEMMs of the step then cost one launch on the host.
        Returns False (after switching the graph path off) if capture is not possible here."""
        t = self._t
        key = (tuple(tuple(x.shape) for x in tensors), tuple(x.dtype for x in tensors), int(T), float(cliprange))
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= 2:
                hostcfg.drop_graphs(self._graphs)
            try:
                static = [t.empty(x.shape, dtype=x.dtype, device=self.device) for x in tensors]
                for d, x in zip(static, tensors):
                    d.copy_(x)

                def body():
                    obs, returns, masks, actions, values, neglogpacs, IS_weight, states = static
                    self._loss_step(cliprange, obs, returns.contiguous(), values.contiguous(), masks, actions, neglogpacs, IS_weight,
                                    states, T, 1)
                side = t.cuda.Stream(device=self.device)
                side.wait_stream(t.cuda.current_stream(self.device))
                with t.cuda.stream(side):          # warm-up outside the capture (GEMM workspaces, kernel attributes, allocator)
                    body()
                t.cuda.current_stream(self.device).wait_stream(side)
                t.cuda.synchronize(self.device)
                graph = t.cuda.CUDAGraph()
                with hostcfg.gc_paused(), t.cuda.graph(graph):
                    body()
                ent = self._graphs[key] = dict(graph=graph, static=static)
            except Exception as e:                 # capture unsupported here: eager launches from now on
                type(self).use_graph = False
                hostcfg.drop_graphs(self._graphs)
                import warnings
                warnings.warn("HIP graph capture of the recurrent PPO step failed (%r); using eager launches" % (e,))
                return False
        else:
            for d, x in zip(ent["static"], tensors):
                d.copy_(x)
        ent["graph"].replay()
        return True

    def train(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, rewards, IS_weight, states=None, nsteps=None):
        if not self.trainable:
            raise RuntimeError("model built with trainable=False")
        if states is None:
            raise ValueError("a recurrent model trains on whole sequences: pass the start states of the minibatch envs")
        t = self._t
        L = ppo_capi.lib()
        np_in = not t.is_tensor(obs)
        T = int(nsteps or self.nsteps)
        world = 1 if self.comm is None else t.distributed.get_world_size(self.comm)
        tensors = (obs, returns, masks, actions, values, neglogpacs, IS_weight, states)
        if not (self.use_graph and self.comm is None and all(t.is_tensor(x) for x in tensors)
                and self._graph_step(cliprange, tensors, T)):
            ret, val = self._dev(returns).contiguous(), self._dev(values).contiguous()
            self._loss_step(cliprange, obs, ret, val, masks, actions, neglogpacs, IS_weight, states, T, world)
        st = t.cuda.current_stream(self.device).cuda_stream
        if self.comm is not None:            # ONE fused collective per optimiser step: [flat grad | loss sums] (equal shards per rank)
            self.grads[self.P:] = self.stats.to(t.float32)
            sdist.allreduce_fused(self.grads, self.comm)
            self.stats.copy_(self.grads[self.P:].to(t.float64))
        entropy = float((self.views[5].double() + 0.5 * np.log(2.0 * np.pi * np.e)).sum().item())
        self.t += 1
        ppo_capi.chk(L.ppo_clip_adam(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.P, self.t,
                                     float(lr), 0.9, 0.999, 1e-5, float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0,
                                     self.stats.data_ptr(), st))
        s = self.stats.cpu().numpy()
        cnt = s[6]
        out = [s[0] / cnt, s[1] / cnt, entropy, s[3] / cnt, s[4] / cnt]
        return [np.float32(x) for x in out] + [None, None] if np_in else out + [None, None]
