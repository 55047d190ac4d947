"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding / collective helpers that the GPU run uses with
RCCL (pattern: reference baselines/baselines/common/tests/test_with_mpi.py:14-38 -- single host, multiple ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from robosumo_selfplay_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    group = sdist.init_process_group("gloo")
    out = {}
    out["shard"] = sdist.shard_envs(8192, rank, world)
    # parameters: rank 0's values win
    params = torch.full((24529,), float(rank + 1))
    sdist.broadcast_params(params, group)
    out["params_ok"] = bool((params == 1.0).all())
    out["synced"] = sdist.assert_synced(params, group)
    # moments: [sum, sumsq, n] of local advantages -> global mean/std identical on all ranks
    rng = np.random.RandomState(rank)
    adv = rng.normal(rank, 1.0, 1000 + 10 * rank)
    mom = torch.tensor([adv.sum(), (adv ** 2).sum(), float(adv.size)], dtype=torch.float64)
    sdist.allreduce_moments(mom, group)
    out["mom"] = mom.numpy().copy()
    # fused gradient + stats buffer (gradients pre-scaled by 1/global_count -> SUM is the global mean gradient)
    buf = torch.arange(24529 + 8, dtype=torch.float32) * (rank + 1)
    sdist.allreduce_fused(buf, group)
    out["fused_ok"] = bool(torch.equal(buf, torch.arange(24529 + 8, dtype=torch.float32) * sum(range(1, world + 1))))
    # divergence is detected
    bad = torch.full((16,), float(rank))
    try:
        sdist.assert_synced(bad, group)
        out["detects"] = False
    except AssertionError:
        out["detects"] = True
    # opponent-data reuse: the ranks hold different numbers of rows (alg_ppo.py:331-335 filters per rank) -> they agree on the
    # number of minibatch steps and the short rank finishes the epoch with empty contributions, so the collectives pair up
    nbatch_train, nsamp = 64, (256 + 130 * rank)               # rank 0: 4 steps of its own, rank 1: ceil(386 / 64) = 7
    steps = sdist.agree_max(-(-nsamp // nbatch_train), group)
    out["steps"] = steps
    seen = 0
    for ii in range(steps):
        rows = max(0, min(nbatch_train, nsamp - ii * nbatch_train))
        mom = torch.tensor([0.0, 0.0, float(rows)], dtype=torch.float64)
        sdist.allreduce_moments(mom, group)                     # would hang here if the ranks disagreed on the step count
        fused = torch.full((10,), float(rows))
        sdist.allreduce_fused(fused, group)
        seen += int(mom[2].item())
    out["rows_seen"] = seen
    q.put((rank, out))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_collectives():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["shard"] == (0, 4096) and res[1]["shard"] == (4096, 4096)
    all_adv = np.concatenate([np.random.RandomState(r).normal(r, 1.0, 1000 + 10 * r) for r in range(world)])
    for r in range(world):
        o = res[r]
        assert o["params_ok"] and o["synced"] and o["fused_ok"] and o["detects"]
        assert o["steps"] == 7 and o["rows_seen"] == 256 + 386       # every row of both ranks counted once, same step count
        s, s2, n = o["mom"]
        assert n == all_adv.size and s / n == pytest.approx(all_adv.mean()) and np.sqrt(s2 / n - (s / n) ** 2) == pytest.approx(all_adv.std())


def test_shard_envs_rejects_ragged():
    with pytest.raises(ValueError):
        sdist.shard_envs(4097, 0, 2)
    assert sdist.shard_envs(32768, 7, 8) == (28672, 4096)


def test_single_process_is_passthrough():
    t = torch.ones(4)
    assert sdist.allreduce_fused(t, None) is t and sdist.assert_synced(t, None)
