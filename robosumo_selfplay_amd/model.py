"""``PPOModel`` on the HIP kernels: same constructor keywords, methods and return shapes as the reference's
TF1 class (model.py:9-213).  Loss / optimiser arithmetic: csrc/ppo_kernels.hip."""
import os

import numpy as np

from . import dist as sdist, hostcfg, policies, ppo_capi


class PPOModel(object):
    loss_names = ["policy_loss", "value_loss", "policy_entropy", "approxkl", "clipfrac"]   # model.py:138

    def __init__(self, *, policy, ob_space=None, ac_space=None, nbatch_act=None, nbatch_train=None, nsteps=None,
                 ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, microbatch_size=None, trainable=True, model_scope="",
                 device=0, comm=None):
        import torch
        if not torch.cuda.is_available():
            raise ppo_capi.PpoHipError("PPOModel needs a HIP device (no CPU fallback in the product path)")
        ppo_capi.lib()
        self._t = torch
        self.spec = policy
        self.scope = model_scope
        self.sess = None
        self.device = torch.device("cuda", int(device))
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), max_grad_norm
        self.trainable = trainable
        self.comm = comm            # torch.distributed process group (or None): gradient / moment all-reduce
        D, A = policy.ob_dim, policy.ac_dim
        self.P = ppo_capi.lib().ppo_param_count(D, A)
        flat = policies.flatten_params(policies.init_param_list(D, A))
        assert flat.size == self.P
        self.params = torch.from_numpy(flat).to(self.device)
        self.act_model = policies.PolicyWithValue(policy, self.params, self.device)
        self.train_model = self.act_model
        self.step = self.act_model.step
        self.value = self.act_model.value
        self.initial_state = None
        if trainable:
            self.m = torch.zeros(self.P, dtype=torch.float32, device=self.device)
            self.v = torch.zeros(self.P, dtype=torch.float32, device=self.device)
            self.t = 0
            self.grads = torch.zeros(self.P + ppo_capi.NSTATS * 2, dtype=torch.float32, device=self.device)
            self.stats = torch.zeros(ppo_capi.NSTATS, dtype=torch.float64, device=self.device)
            self.moments = torch.zeros(3, dtype=torch.float64, device=self.device)
            self.workspace = torch.zeros(ppo_capi.lib().ppo_grad_workspace_bytes(D, A), dtype=torch.uint8, device=self.device)   # zeroed once: sumo_ppo.h
            self.adv_ws = torch.zeros(ppo_capi.lib().ppo_adv_moments_workspace_bytes(), dtype=torch.uint8, device=self.device)   # this model's own (sumo_ppo.h)
            self._graphs = {}
            self._static = None
            self._epoch_moments = None

    # ---- checkpoints: list of 13 float32 arrays in TF variable order (model.py:153-177) -----------------------
    def get_param_list(self):
        return policies.unflatten_params(self.params.cpu().numpy(), self.spec.ob_dim, self.spec.ac_dim)

    def set_param_list(self, plist):
        shapes = policies.param_shapes(self.spec.ob_dim, self.spec.ac_dim)
        if isinstance(plist, dict):                                    # model.py:174-175: dict keyed by TF variable name
            plist = [plist["%s/%s:0" % (self.scope, n)] if "%s/%s:0" % (self.scope, n) in plist else plist[n]
                     for n in policies.PARAM_NAMES]
        assert len(plist) == len(shapes), "number of variables loaded mismatches len(variables)"
        for p, s in zip(plist, shapes):
            if tuple(np.shape(p)) != tuple(s):
                raise ValueError("checkpoint tensor shape %s does not match %s" % (np.shape(p), s))
        self.params.copy_(self._t.from_numpy(policies.flatten_params(plist)))

    def save(self, save_path):
        dirname = os.path.dirname(save_path)
        if dirname:
            os.makedirs(dirname, exist_ok=True)
        import joblib
        joblib.dump(self.get_param_list(), save_path)                  # same on-disk format as model.py:161

    def load(self, load_path):
        import joblib
        self.set_param_list(joblib.load(os.path.expanduser(load_path)))   # only files written by save()

    # ---- training (model.py:179-213) ---------------------------------------------------------------------------
    def _dev(self, x, dtype):
        t = self._t
        if t.is_tensor(x):
            return x
        return t.as_tensor(np.ascontiguousarray(x, dtype=dtype)).to(self.device)

    def train(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, rewards, IS_weight, states=None):
        t = self._t
        np_in = not t.is_tensor(obs)
        obs = self._dev(obs, np.float32)
        ret, val = self._dev(returns, np.float32), self._dev(values, np.float32)
        act, old, w = self._dev(actions, np.float32), self._dev(neglogpacs, np.float32), self._dev(IS_weight, np.float32)
        out = self.train_indexed(lr, cliprange, obs, ret, act, val, old, w, None, obs.shape[0])
        stats = out[:5]
        if np_in:
            return [np.float32(s) for s in stats] + [out[5].cpu().numpy(), None]
        return list(stats) + [out[5], None]

    # The launch-bound inner loop (noptepochs x nminibatches optimiser steps of ~7 small launches + tensor bookkeeping each)
    # is captured once per (batch arrays, minibatch size, cliprange) into a HIP graph and replayed with a fresh index
    # vector; only the Adam step stays outside (its step count is a host scalar).  Single-GPU, asynchronous path only.
    use_graph = os.environ.get("SUMO_PPO_GRAPH", "1") != "0"
    equal_counts = True   # multi-GPU: minibatches have the same size on every rank (learn() clears it for opponent-data reuse)

    def _launch_loss_grad(self, obs, returns, actions, values, neglogpacs, weights, idx, n, cliprange, adv, log_ratio, st, mom=None):
        """The launches of one optimiser step up to the gradient.  ``mom`` None: single GPU, the minibatch's own advantage moments.
        ``mom`` = device [3] float64 holding the GLOBAL moments of this minibatch (``prepare_epoch``): multi-GPU form -- normalise with
        them, scale by the global count, then ONE all-reduce of [flat grad | loss sums | count]."""
        L = ppo_capi.lib()
        D, A = self.spec.ob_dim, self.spec.ac_dim
        ip = idx.data_ptr()
        if mom is None:
            mom = self.moments
            ppo_capi.chk(L.ppo_adv_moments_ws(returns.data_ptr(), values.data_ptr(), ip, n, mom.data_ptr(), self.adv_ws.data_ptr(), st))
            count = float(n)
        else:
            count = float(n) * self._t.distributed.get_world_size(self.comm)       # equal shards: no host read-back of the reduced count
        ppo_capi.chk(L.ppo_adv_normalize(returns.data_ptr(), values.data_ptr(), ip, n, mom.data_ptr(), adv.data_ptr(), st))
        self.stats.zero_()
        ppo_capi.chk(L.ppo_grad(self.params.data_ptr(), obs.data_ptr(), obs.stride(0), D, A, actions.data_ptr(), adv.data_ptr(),
                                returns.data_ptr(), neglogpacs.data_ptr(), weights.data_ptr(), ip, n, 1.0 / count,
                                float(cliprange), self.ent_coef, self.vf_coef, self.grads.data_ptr(), self.stats.data_ptr(),
                                log_ratio.data_ptr(), self.workspace.data_ptr(), st))
        if self.comm is not None:
            self._allreduce_grad_and_stats()

    def _allreduce_grad_and_stats(self):
        """ONE fused collective per optimiser step: [flat grad | 8 loss sums] (the reference's only gradient collective,
        mpi_adam_optimizer.py:39, all-reduces the flat gradient alone)."""
        t = self._t
        self.grads[self.P:self.P + ppo_capi.NSTATS] = self.stats.to(t.float32)
        sdist.allreduce_fused(self.grads, self.comm)
        self.stats.copy_(self.grads[self.P:self.P + ppo_capi.NSTATS].to(t.float64))

    def prepare_epoch(self, inds, nbatch_train, returns=None, values=None):
        """Multi-GPU with equal shards: the advantage moments of ALL minibatches of the coming epoch -- rows
        ``inds[k * nbatch_train : (k + 1) * nbatch_train]`` of the ``begin_update`` batch (or of ``returns`` / ``values``) -- are summed
        locally (one small launch per minibatch, no host sync) and all-reduced ONCE as an [nmb, 3] float64 buffer, so that every optimiser
        step of the epoch issues exactly one collective.  Every rank passes the shuffle of ITS shard (the shuffles need not agree: the
        global moments of step k are the sum over the ranks of their k-th local minibatch); all ranks must run the same number of
        minibatches in lockstep.  Without a communicator, or with unequal shards (opponent-data reuse), this is a no-op and the steps all-reduce
        their own moments as before.  Steps pick their row with ``train_indexed(..., mb_index=k)``."""
        self._epoch_moments = None
        if self.comm is None or not self.equal_counts:
            return None
        t = self._t
        if returns is None:
            if self._static is None or not self._static["open"]:
                raise ValueError("prepare_epoch needs begin_update() or explicit returns / values")
            returns, values = self._static["bufs"][1], self._static["bufs"][3]
        nmb = -(-int(inds.numel()) // int(nbatch_train))
        mom = t.zeros((nmb, 3), dtype=t.float64, device=self.device)
        st = t.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        L = ppo_capi.lib()
        for k in range(nmb):
            mb = inds[k * nbatch_train:(k + 1) * nbatch_train]
            ppo_capi.chk(L.ppo_adv_moments_ws(returns.data_ptr(), values.data_ptr(), mb.data_ptr(), int(mb.numel()), mom[k].data_ptr(), self.adv_ws.data_ptr(), st))
        sdist.allreduce_moments(mom, self.comm)
        self._epoch_moments = mom
        return mom

    def begin_update(self, obs, returns, actions, values, neglogpacs, weights):
        """Hand the batch arrays of ONE update to the model: they are copied (six device copies per update, not per step) into
        buffers the model owns, so the captured step graph keeps valid pointers and is captured once per batch shape.  The
        asynchronous ``train_indexed(..., sync=False)`` steps that follow gather their rows from these copies until
        ``end_update()`` or the next ``begin_update``.  The hand-over is explicit on purpose: whether a caller's tensors still hold
        the same content cannot be told from their addresses (a caller that refills its buffers in place, or frees and
        re-allocates them, gets the same pointers back).  Without it ``train_indexed`` runs the eager launches on the arrays it
        is given."""
        t = self._t
        if obs.stride(1) != 1:
            raise ValueError("obs rows must have unit inner stride")
        arrs = (obs, returns, actions, values, neglogpacs, weights)
        shp = (tuple(obs.shape), obs.stride(0), tuple(weights.shape))
        if self._static is None or self._static["shape"] != shp:
            self._static = dict(shape=shp, bufs=[t.empty_like(x, memory_format=t.contiguous_format) for x in arrs])
            hostcfg.drop_graphs(self._graphs)
        for dst, x in zip(self._static["bufs"], arrs):
            dst.copy_(x)
        self._static["open"] = True
        self._epoch_moments = None

    def end_update(self):
        if self._static is not None:
            self._static["open"] = False
        self._epoch_moments = None

    def _graph_step(self, lr, cliprange, idx, n, mom=None):
        """``mom`` (multi-GPU: the minibatch's global advantage moments from ``prepare_epoch``): the captured step then holds
        adv_normalize -> ppo_grad -> fused all-reduce (RCCL calls capture on the stream) -> loss statistics, and the moments are copied
        into the graph's own buffer before every replay, like the index vector."""
        t = self._t
        A = self.spec.ac_dim
        obs, returns, actions, values, neglogpacs, weights = self._static["bufs"]
        key = (int(n), float(cliprange), mom is not None)
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 2:
                hostcfg.drop_graphs(self._graphs)
            try:
                ent = dict(idx=t.zeros(n, dtype=t.int32, device=self.device), adv=t.empty(n, dtype=t.float32, device=self.device),
                           log_ratio=t.empty(n, dtype=t.float32, device=self.device), keep=(obs, returns, actions, values, neglogpacs, weights),
                           mom=None if mom is None else t.zeros(3, dtype=t.float64, device=self.device))
                side = t.cuda.Stream(device=self.device)
                side.wait_stream(t.cuda.current_stream(self.device))
                with t.cuda.stream(side):       # warm-up outside the capture (one-time kernel attributes, allocator, communicator set-up)
                    ent["idx"].copy_(idx)
                    if mom is not None:
                        ent["mom"].copy_(mom)
                    self._launch_loss_grad(obs, returns, actions, values, neglogpacs, weights, ent["idx"], n, cliprange, ent["adv"],
                                           ent["log_ratio"], side.cuda_stream, mom=ent["mom"])
                t.cuda.current_stream(self.device).wait_stream(side)
                t.cuda.synchronize(self.device)
                graph = t.cuda.CUDAGraph()
                # with a collective inside, only this thread's calls belong to the capture (the process group's watchdog thread polls events)
                gkw = dict(capture_error_mode="thread_local") if mom is not None else {}
                with hostcfg.gc_paused(), t.cuda.graph(graph, **gkw):
                    cst = t.cuda.current_stream(self.device).cuda_stream
                    self._launch_loss_grad(obs, returns, actions, values, neglogpacs, weights, ent["idx"], n, cliprange, ent["adv"],
                                           ent["log_ratio"], cst, mom=ent["mom"])
                    ent["out"] = self._loss_stats(cst)
                ent["graph"] = graph
                g = self._graphs[key] = ent
            except Exception as e:                     # capture unsupported here: stay on the eager path for good
                if mom is not None:
                    type(self).use_comm_graph = False
                else:
                    type(self).use_graph = False
                hostcfg.drop_graphs(self._graphs)
                import warnings
                warnings.warn("HIP graph capture of the PPO step failed (%r); using eager launches" % (e,))
                return None
        g["idx"].copy_(idx)
        if mom is not None:
            g["mom"].copy_(mom)
        g["graph"].replay()
        self.t += 1
        ppo_capi.chk(ppo_capi.lib().ppo_clip_adam(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                                   self.P, self.t, float(lr), 0.9, 0.999, 1e-5,
                                                   float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0,
                                                   self.stats.data_ptr(), t.cuda.current_stream(self.device).cuda_stream))
        return g["out"].clone()

    # multi-GPU step graph (needs RCCL: gloo's collectives run on the host and cannot be captured)
    use_comm_graph = os.environ.get("SUMO_PPO_COMM_GRAPH", "1") != "0"

    def _comm_backend(self):
        try:
            return self._t.distributed.get_backend(self.comm)
        except Exception:
            return None

    def train_indexed(self, lr, cliprange, obs, returns, actions, values, neglogpacs, weights, idx, n, sync=True, mb_index=None):
        """One optimiser step on rows ``idx`` (int32 CUDA tensor or None) of device-resident batch arrays.
        ``sync=False`` skips the host read-back of the loss statistics (returns a device tensor
        [pg, vf, entropy, approxkl, clipfrac] instead) so consecutive minibatch steps queue without host stalls."""
        if not self.trainable:
            raise RuntimeError("model built with trainable=False")
        t = self._t
        L = ppo_capi.lib()
        D, A = self.spec.ob_dim, self.spec.ac_dim
        if obs.stride(1) != 1:
            raise ValueError("obs rows must have unit inner stride")
        em = self._epoch_moments if (self.comm is not None and self.equal_counts and mb_index is not None) else None
        mom_k = em[int(mb_index)] if em is not None else None                # this minibatch's global moments (prepare_epoch)
        if not sync and idx is not None and n > 0 and self._static is not None and self._static["open"]:
            if self.comm is None and self.use_graph:
                out = self._graph_step(lr, cliprange, idx, n)      # rows come from the begin_update() copies
                if out is not None:
                    return out
            elif mom_k is not None and self.use_graph and self.use_comm_graph and self._comm_backend() == "nccl":
                out = self._graph_step(lr, cliprange, idx, n, mom=mom_k)
                if out is not None:
                    return out
        st = t.cuda.current_stream(self.device).cuda_stream
        ip = ppo_capi.ptr(idx)
        if n == 0:
            # a rank whose shard ran out of rows (opponent-data reuse gives ranks different batch sizes) still takes part in
            # both collectives of the step with an empty contribution, so every rank issues the same sequence of all-reduces
            if self.comm is None or self.equal_counts:
                raise ValueError("empty minibatch")
            self.moments.zero_()
            sdist.allreduce_moments(self.moments, self.comm)
            self.grads.zero_()
            sdist.allreduce_fused(self.grads, self.comm)
            self.stats.copy_(self.grads[self.P:self.P + ppo_capi.NSTATS].to(t.float64))
            return self._finish_step(lr, sync, t.empty(0, dtype=t.float32, device=self.device), st)
        # advantages: returns - values, normalised over the (global) minibatch (model.py:180-185)
        if mom_k is not None:          # the epoch's moments were all-reduced in one collective (prepare_epoch): nothing to exchange here
            self.moments.copy_(mom_k)
        else:
            ppo_capi.chk(L.ppo_adv_moments_ws(returns.data_ptr(), values.data_ptr(), ip, n, self.moments.data_ptr(), self.adv_ws.data_ptr(), st))
            sdist.allreduce_moments(self.moments, self.comm)
        adv = t.empty(n, dtype=t.float32, device=self.device)
        ppo_capi.chk(L.ppo_adv_normalize(returns.data_ptr(), values.data_ptr(), ip, n, self.moments.data_ptr(), adv.data_ptr(), st))
        if self.comm is None:
            count = float(n)
        elif self.equal_counts:       # every rank contributes the same number of rows: no host read-back of the all-reduced count
            count = float(n) * self._t.distributed.get_world_size(self.comm)
        else:
            count = float(self.moments[2].item())
        self.stats.zero_()
        log_ratio = t.empty(n, dtype=t.float32, device=self.device)
        ppo_capi.chk(L.ppo_grad(self.params.data_ptr(), obs.data_ptr(), obs.stride(0), D, A, actions.data_ptr(), adv.data_ptr(),
                                returns.data_ptr(), neglogpacs.data_ptr(), weights.data_ptr(), ip, n, 1.0 / count,
                                float(cliprange), self.ent_coef, self.vf_coef, self.grads.data_ptr(), self.stats.data_ptr(),
                                log_ratio.data_ptr(), self.workspace.data_ptr(), st))
        if self.comm is not None:
            self._allreduce_grad_and_stats()
        return self._finish_step(lr, sync, log_ratio, st)

    def _loss_stats(self, st):
        """[policy loss, value loss, entropy, approxkl, clipfrac] of the step whose sums sit in ``self.stats`` (one launch)."""
        t = self._t
        A = self.spec.ac_dim
        out5 = t.empty(5, dtype=t.float64, device=self.device)
        off = self.P - 1 - policies.HIDDEN - A                       # pi/logstd inside the flat parameter vector (checkpoint order)
        ppo_capi.chk(ppo_capi.lib().ppo_loss_stats(self.stats.data_ptr(), self.params.data_ptr() + 4 * off, A, out5.data_ptr(), st))
        return out5

    def _finish_step(self, lr, sync, log_ratio, st):
        t = self._t
        A = self.spec.ac_dim
        # loss means + entropy of the distribution the loss was evaluated with (before the parameter update), model.py:69
        out5 = self._loss_stats(st)
        self.t += 1
        ppo_capi.chk(ppo_capi.lib().ppo_clip_adam(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.P,
                                                   self.t, float(lr), 0.9, 0.999, 1e-5,
                                                   float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0,
                                                   self.stats.data_ptr(), st))
        if not sync:
            return out5
        o = out5.cpu().numpy()
        return [o[0], o[1], o[2], o[3], o[4], log_ratio, float(self.stats[7].item())]
