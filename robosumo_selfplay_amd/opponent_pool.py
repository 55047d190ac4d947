"""Device-resident pool of frozen opponent snapshots with a per-env snapshot index (BASELINE config 5).

The reference keeps its opponent pool on disk -- the checkpoint directory -- and loads ONE snapshot for all parallel envs
before every update (alg_ppo.py:191-247, "all parallel environments get same opponent model", :214).  Here up to ``capacity``
snapshots stay in HBM as rows of one ``[capacity, P]`` float32 matrix in checkpoint order (model.py:153-177) and ``index[e]``
names the snapshot env ``e`` plays against; the fused rollout launch (include/sumo_hip.h ``sumo_rollout_steps``) reads each
env's opponent weights through that index, so a rollout can face the whole pool at once.  With ``capacity == 1`` this is the
reference's behaviour.
"""
import numpy as np

from . import policies


class OpponentPool(object):
    def __init__(self, spec, capacity, num_envs, device):
        import torch
        self._t = torch
        self.spec = spec
        self.capacity = int(capacity)
        if self.capacity < 1:
            raise ValueError("capacity must be >= 1")
        self.num_envs = int(num_envs)
        self.device = device
        D, A = spec.ob_dim, spec.ac_dim
        self.P = int(policies.flatten_params(policies.init_param_list(D, A)).size)
        self.params = torch.zeros((self.capacity, self.P), dtype=torch.float32, device=device)
        self.index = torch.zeros(self.num_envs, dtype=torch.int32, device=device)
        self.filled = np.zeros(self.capacity, bool)
        self.labels = [None] * self.capacity          # what each slot holds (checkpoint path / version), for logs

    def set_snapshot(self, k, params, label=None):
        """Fill slot ``k`` from a flat float32 vector (tensor / array), a 13-array checkpoint list (model.py:153-177) or a path
        written by ``PPOModel.save``."""
        t = self._t
        if not 0 <= k < self.capacity:
            raise IndexError("slot %d outside the pool of %d" % (k, self.capacity))
        if isinstance(params, str):
            import joblib
            label = label or params
            params = joblib.load(params)                       # only files written by PPOModel.save()
        if isinstance(params, (list, tuple)):
            shapes = policies.param_shapes(self.spec.ob_dim, self.spec.ac_dim)
            if len(params) != len(shapes) or any(tuple(np.shape(p)) != tuple(s) for p, s in zip(params, shapes)):
                raise ValueError("checkpoint does not match the pool's policy shape")
            params = policies.flatten_params(list(params))
        v = params if t.is_tensor(params) else t.from_numpy(np.ascontiguousarray(params, np.float32))
        if v.numel() != self.P:
            raise ValueError("snapshot has %d parameters, the pool's policy %d" % (v.numel(), self.P))
        self.params[k].copy_(v.reshape(-1).to(self.device, t.float32))
        self.filled[k] = True
        self.labels[k] = label

    def assign(self, index):
        """Per-env snapshot index (length ``num_envs``); every referenced slot must be filled."""
        idx = np.asarray(index.cpu().numpy() if self._t.is_tensor(index) else index).astype(np.int64).reshape(-1)
        if idx.shape[0] != self.num_envs:
            raise ValueError("index has %d entries for %d envs" % (idx.shape[0], self.num_envs))
        if idx.min() < 0 or idx.max() >= self.capacity or not self.filled[np.unique(idx)].all():
            raise ValueError("index refers to an empty or non-existent pool slot")
        self.index.copy_(self._t.from_numpy(idx.astype(np.int32)))

    def assign_round_robin(self, slots=None):
        """Spread the envs evenly over the filled slots (or the given ones): env e -> slots[e mod len(slots)]."""
        slots = np.nonzero(self.filled)[0] if slots is None else np.asarray(slots)
        self.assign(slots[np.arange(self.num_envs) % len(slots)])

    def counts(self):
        return np.bincount(self.index.cpu().numpy(), minlength=self.capacity)


class LstmOpponentPool(object):
    """Pool of frozen baselines-LSTM opponents (BASELINE config 5: LSTM(128), 16 snapshots).  Stands where the Runner's
    ``models[1]`` stands in the recurrent device path (runner.py:62-96 with the S / M feeds): ``step`` acts for agent 1 and
    carries the recurrent state, ``act_model.action_probability`` scores agent 0's actions -- every 16-env tile with its own
    snapshot, all tiles in ONE launch (``ppo_lstm_step_pool``).  The snapshot index is per tile of 16 consecutive envs (an MFMA
    tile shares its weight operands), so ``assign`` takes ``num_envs / 16`` entries; ``index`` expands it per env."""
    recurrent = True
    accepts_noise = True

    def __init__(self, spec, capacity, num_envs, device):
        import ctypes as C
        import torch
        from . import ppo_capi
        self._t, self._C, self._capi = torch, C, ppo_capi
        if num_envs % 16:
            raise ValueError("an LSTM opponent pool needs num_envs to be a multiple of 16 (one snapshot per 16-env tile)")
        self.spec, self.capacity, self.num_envs, self.device = spec, int(capacity), int(num_envs), device
        D, A, H = spec.ob_dim, spec.ac_dim, spec.nlstm
        self.shapes = policies.lstm_param_shapes(D, A, H)
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.P = int(sum(self.sizes))
        self.params = torch.zeros((self.capacity, self.P), dtype=torch.float32, device=device)
        nets = (ppo_capi.LstmNet * self.capacity)()
        for k in range(self.capacity):
            self._fill_net(nets[k], self.params[k].data_ptr())
        self._proto = ppo_capi.LstmNet()
        self._fill_net(self._proto, self.params[0].data_ptr())
        self._nets_dev = torch.from_numpy(np.frombuffer(bytes(nets), dtype=np.uint8).copy()).to(device)
        self.tile_net = torch.zeros(self.num_envs // 16, dtype=torch.int32, device=device)
        self.filled = np.zeros(self.capacity, bool)
        self.labels = [None] * self.capacity
        self.gen = torch.Generator(device=device)
        self.act_model = self.train_model = self
        self.initial_state = np.zeros((self.num_envs, 2 * H), np.float32)            # models.py:176

    def _fill_net(self, n, base):
        D, A, H = self.spec.ob_dim, self.spec.ac_dim, self.spec.nlstm
        o = np.concatenate([[0], np.cumsum(self.sizes)]) * 4
        n.ob_dim, n.emb_dim, n.hidden, n.ac_dim = D, 0, H, A
        n.gate_order, n.forget_bias = self._capi.LSTM_GATES_IFOU, 0.0
        n.wx, n.wh, n.b = base + int(o[0]), base + int(o[1]), base + int(o[2])
        n.head_w, n.head_b, n.logstd, n.vf_w, n.vf_b = base + int(o[3]), base + int(o[4]), base + int(o[5]), base + int(o[6]), base + int(o[7])

    @property
    def index(self):
        return self.tile_net.repeat_interleave(16)

    def seed(self, s):
        self.gen.manual_seed(int(s))

    def set_snapshot(self, k, params, label=None):
        t = self._t
        if not 0 <= k < self.capacity:
            raise IndexError("slot %d outside the pool of %d" % (k, self.capacity))
        if isinstance(params, str):
            import joblib
            label = label or params
            params = joblib.load(params)                       # only files written by LstmPPOModel.save()
        if isinstance(params, (list, tuple)):
            if len(params) != len(self.shapes) or any(tuple(np.shape(p)) != tuple(s) for p, s in zip(params, self.shapes)):
                raise ValueError("checkpoint does not match the pool's policy shape")
            params = np.concatenate([np.asarray(p, np.float32).ravel() for p in params])
        v = params if t.is_tensor(params) else t.from_numpy(np.ascontiguousarray(params, np.float32))
        if v.numel() != self.P:
            raise ValueError("snapshot has %d parameters, the pool's policy %d" % (v.numel(), self.P))
        self.params[k].copy_(v.reshape(-1).to(self.device, t.float32))
        self.filled[k] = True
        self.labels[k] = label

    def assign(self, tile_index):
        idx = np.asarray(tile_index.cpu().numpy() if self._t.is_tensor(tile_index) else tile_index).astype(np.int64).reshape(-1)
        if idx.shape[0] != self.num_envs // 16:
            raise ValueError("tile index has %d entries for %d tiles" % (idx.shape[0], self.num_envs // 16))
        if idx.min() < 0 or idx.max() >= self.capacity or not self.filled[np.unique(idx)].all():
            raise ValueError("index refers to an empty or non-existent pool slot")
        self.tile_net.copy_(self._t.from_numpy(idx.astype(np.int32)))

    def assign_round_robin(self, slots=None):
        slots = np.nonzero(self.filled)[0] if slots is None else np.asarray(list(slots))
        self.assign(slots[np.arange(self.num_envs // 16) % len(slots)])

    def counts(self):
        return np.bincount(self.tile_net.cpu().numpy(), minlength=self.capacity) * 16

    def _run(self, obs, S, M, given_action=None, deterministic=False, first_env=0, noise=None):
        t, C, cap = self._t, self._C, self._capi
        D, A, H = self.spec.ob_dim, self.spec.ac_dim, self.spec.nlstm
        if not t.is_tensor(obs):
            raise TypeError("the opponent pool lives on the device: pass CUDA tensors (device-mode Runner)")
        x = obs.reshape(-1, D)
        n = x.shape[0]
        if first_env % 16 or n % 16 or first_env + n > self.num_envs:
            raise ValueError("rows must be whole 16-env tiles of the pool's envs")
        st = t.zeros((n, 2 * H), dtype=t.float32, device=self.device) if S is None else S.clone()
        mask = None if M is None else M.to(t.float32)
        action = t.empty((n, A), dtype=t.float32, device=self.device)
        neglogp = t.empty(n, dtype=t.float32, device=self.device)
        value = t.empty(n, dtype=t.float32, device=self.device)
        given = None if given_action is None else given_action.reshape(n, A).contiguous()
        if deterministic or given is not None:
            noise = None
        elif noise is None:
            noise = t.randn((n, A), generator=self.gen, device=self.device, dtype=t.float32)
        tiles = self.tile_net[first_env // 16:(first_env + n) // 16]
        cap.chk(cap.lib().ppo_lstm_step_pool(C.byref(self._proto), self._nets_dev.data_ptr(), tiles.data_ptr(), x.data_ptr(), n,
                                             x.stride(0) if n > 1 else D, cap.ptr(mask), st.data_ptr(), st.data_ptr() + 4 * H, 2 * H,
                                             cap.ptr(noise), cap.ptr(given), action.data_ptr(), neglogp.data_ptr(), value.data_ptr(), None,
                                             t.cuda.current_stream(self.device).cuda_stream))
        return action, value, st, neglogp

    # the Runner's recurrent device path evaluates one env group at a time: ``first_env`` names the group's first env
    def step(self, observation, S=None, M=None, deterministic=False, first_env=0, noise=None, **extra_feed):
        return self._run(observation, S, M, deterministic=deterministic, first_env=first_env, noise=noise)

    def value(self, ob, S=None, M=None, first_env=0, **kwargs):
        return self._run(ob, S, M, deterministic=True, first_env=first_env)[1]

    def action_probability(self, observation, given_action=None, S=None, M=None, first_env=0, **extra_feed):
        return self._run(observation, S, M, given_action=given_action, first_env=first_env)[3]
