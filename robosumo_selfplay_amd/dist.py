"""Multi-GPU sharding of the hot path: one process per GPU, envs partitioned, RCCL over xGMI.

The reference's only gradient-collective design is the vendored MPI data-parallel Adam
(baselines/baselines/common/mpi_adam_optimizer.py:26-67: Allreduce(SUM)/divide of the flat gradient, root broadcast at
init via mpi_util.sync_from_root (mpi_util.py:24), periodic sync assertion).  Here: the rollout needs no communication;
each optimiser step issues ONE all-reduce of a fused buffer ``[flat grad | loss sums | count]`` plus one 3-double
all-reduce of the advantage moments.  Messages are tiny (98 KB), i.e. latency-bound: fusing is what matters, not bandwidth.

All helpers take any torch tensors, so the N>1 logic is exercised with the gloo backend on CPU (tests/test_dist_gloo.py).
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """backend 'nccl' IS RCCL on ROCm; 'gloo' for CPU rehearsal."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:      # SUMO_DIST_BACKEND=gloo: rehearse N ranks on one GPU (RCCL refuses two ranks on one device)
        backend = os.environ.get("SUMO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    elif torch.cuda.is_available():
        local_rank = local_rank % torch.cuda.device_count()
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def shard_envs(total_envs, rank, world):
    """Contiguous partition of env ids; every rank gets total/world (weak scaling asks for equal shards)."""
    if total_envs % world:
        raise ValueError("total_envs %d not divisible by world size %d" % (total_envs, world))
    per = total_envs // world
    return rank * per, per


def broadcast_params(params, group, src=0):
    import torch.distributed as dist
    if group is not None:
        dist.broadcast(params, src=src, group=group)
    return params


def allreduce_moments(moments, group):
    """moments = [sum adv, sum adv^2, count] (float64): global advantage normalisation (model.py:182-185)."""
    import torch.distributed as dist
    if group is not None:
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


def allreduce_fused(grad_and_stats, group):
    """ONE collective per optimiser step over [flat grad | stats]; gradients are already scaled by 1/global_count,
    so SUM gives the gradient of the global mean loss (no divide, unlike mpi_adam_optimizer.py:39-40)."""
    import torch.distributed as dist
    if group is not None:
        dist.all_reduce(grad_and_stats, op=dist.ReduceOp.SUM, group=group)
    return grad_and_stats


def agree_max(value, group, device=None):
    """MAX of a host integer over the ranks (e.g. the number of minibatch steps of an epoch when opponent-data reuse gives the
    ranks different batch sizes: every rank must issue the same number of collectives)."""
    import torch
    import torch.distributed as dist
    if group is None:
        return int(value)
    v = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    return int(v.item())


def assert_synced(params, group):
    """Counterpart of MpiAdamOptimizer.check_synced (mpi_adam_optimizer.py:54-67): every rank must hold identical weights."""
    import torch
    import torch.distributed as dist
    if group is None:
        return True
    s = params.double().sum().reshape(1)
    lo, hi = s.clone(), s.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise AssertionError("parameters diverged across ranks: %r vs %r" % (lo.item(), hi.item()))
    return True
