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
