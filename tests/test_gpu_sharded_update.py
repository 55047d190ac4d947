"""§8(e) on one GPU: two ranks (gloo collectives over CUDA tensors, both on cuda:0) each train on half of a batch; the
resulting parameters must be identical on both ranks and match a single-process update on the whole batch (pattern:
reference baselines/baselines/ppo2/test_microbatches.py:12-32, whose tolerance is atol 3e-3; here 2e-5)."""
import os
import socket

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

OB, AC, N = 121, 8, 2048


def _batch():
    rng = np.random.RandomState(7)
    obs = rng.normal(0, 1, (N, OB)).astype(np.float32)
    act = rng.normal(0, 1, (N, AC)).astype(np.float32)
    ret = rng.normal(0, 2, N).astype(np.float32)
    val = rng.normal(0, 2, N).astype(np.float32)
    old = rng.normal(11, 1, N).astype(np.float32)
    return obs, act, ret, val, old


def _train(comm, lo, hi, steps=3):
    import torch
    from robosumo_selfplay_amd import dist as sdist, model as model_mod, policies
    np.random.seed(3)
    spec = policies.PolicySpec(OB, AC, value_network="copy", activation="relu")
    m = model_mod.PPOModel(policy=spec, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, comm=comm)
    sdist.broadcast_params(m.params, comm)
    obs, act, ret, val, old = (torch.as_tensor(x[lo:hi]).cuda() for x in _batch())
    w = torch.ones(hi - lo, dtype=torch.float32, device="cuda")
    stats = None
    for _ in range(steps):
        stats = m.train_indexed(1e-3, 0.2, obs, ret, act, val, old, w, None, hi - lo)
    sdist.assert_synced(m.params, comm)
    return m.params.cpu().numpy(), np.array(stats[:5], dtype=np.float64)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N // world
    p, st = _train(dist.group.WORLD, rank * per, (rank + 1) * per)
    q.put((rank, p, st))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_two_shards_match_single_process():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, p, st = q.get(timeout=300)
        res[r] = (p, st)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref_p, ref_st = _train(None, 0, N)
    assert np.array_equal(res[0][0], res[1][0])                      # ranks stay bit-identical
    assert np.allclose(res[0][0], ref_p, rtol=0, atol=2e-5), np.abs(res[0][0] - ref_p).max()
    assert np.allclose(res[0][1], ref_st, rtol=1e-3, atol=1e-5)      # loss statistics are global means
    assert np.allclose(res[1][1], ref_st, rtol=1e-3, atol=1e-5)


def _uneven_worker(rank, world, port, q):
    """Opponent-data reuse shape (alg_ppo.py:331-335): rank 0 holds 1536 rows, rank 1 512; minibatches of 512 -> rank 1 runs out
    after one step and joins the remaining two with empty minibatches (model.train_indexed n == 0)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from robosumo_selfplay_amd import dist as sdist, model as model_mod, policies
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = dist.group.WORLD
    np.random.seed(3)
    m = model_mod.PPOModel(policy=policies.PolicySpec(OB, AC, value_network="copy", activation="relu"), ent_coef=0.01, vf_coef=0.5,
                           max_grad_norm=0.5, comm=comm)
    m.equal_counts = False
    sdist.broadcast_params(m.params, comm)
    lo, hi = (0, 1536) if rank == 0 else (1536, 2048)
    obs, act, ret, val, old = (torch.as_tensor(x[lo:hi]).cuda() for x in _batch())
    w = torch.ones(hi - lo, dtype=torch.float32, device="cuda")
    steps = sdist.agree_max(-(-(hi - lo) // 512), comm, device=torch.device("cuda", 0))
    idx = torch.arange(hi - lo, dtype=torch.int32, device="cuda")
    for ii in range(steps):
        mb = idx[ii * 512:(ii + 1) * 512]
        st = m.train_indexed(1e-3, 0.2, obs, ret, act, val, old, w, mb, int(mb.numel()))
    sdist.assert_synced(m.params, comm)
    q.put((rank, m.params.cpu().numpy(), steps, np.array(st[:5], dtype=np.float64)))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_uneven_shards_keep_collectives_paired():
    import torch
    import torch.multiprocessing as mp
    from robosumo_selfplay_amd import model as model_mod, policies
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, p, steps, st = q.get(timeout=300)
        res[r] = (p, steps, st)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 3
    assert np.array_equal(res[0][0], res[1][0])
    # single process, same global minibatches: step 0 = rows [0,512) + [1536,2048), steps 1,2 = rows [512,1024), [1024,1536)
    np.random.seed(3)
    m = model_mod.PPOModel(policy=policies.PolicySpec(OB, AC, value_network="copy", activation="relu"), ent_coef=0.01, vf_coef=0.5,
                           max_grad_norm=0.5)
    obs, act, ret, val, old = (torch.as_tensor(x).cuda() for x in _batch())
    w = torch.ones(N, dtype=torch.float32, device="cuda")
    for rows in (list(range(0, 512)) + list(range(1536, 2048)), list(range(512, 1024)), list(range(1024, 1536))):
        mb = torch.tensor(rows, dtype=torch.int32, device="cuda")
        st = m.train_indexed(1e-3, 0.2, obs, ret, act, val, old, w, mb, len(rows))
    assert np.allclose(res[0][0], m.params.cpu().numpy(), rtol=0, atol=2e-5)
    assert np.allclose(res[1][2], np.array(st[:5], dtype=np.float64), rtol=1e-3, atol=1e-5)   # the empty rank reports the global means too


def _rccl_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from robosumo_selfplay_amd import dist as sdist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # the call bench.py / run.py make
    comm = dist.group.WORLD
    p, st = _train(comm, 0, N)
    buf = torch.arange(24529 + 8, dtype=torch.float32, device="cuda")
    sdist.allreduce_fused(buf, comm)
    ok = bool(torch.equal(buf.cpu(), torch.arange(24529 + 8, dtype=torch.float32))) and sdist.agree_max(5, comm, device=torch.device("cuda", 0)) == 5
    q.put((p, st, ok, dist.get_backend(comm)))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_rccl_process_group_world_size_one():
    """The RCCL ('nccl') initialisation and collective path of dist.py / bench.py executes on hardware: one rank, real
    all-reduces of the fused gradient buffer and the advantage moments, same update as without a communicator."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(port, q))
    pr.start()
    p, st, ok, backend = q.get(timeout=300)
    pr.join(timeout=120)
    assert pr.exitcode == 0 and ok and backend == "nccl"
    ref_p, ref_st = _train(None, 0, N)
    assert np.allclose(p, ref_p, rtol=0, atol=2e-6) and np.allclose(st, ref_st, rtol=1e-4, atol=1e-6)


# ---- the epoch's advantage moments in ONE all-reduce, then exactly one collective per optimiser step ------------
def _train_epochs(comm, lo, hi, nmb=4, epochs=2, count_calls=None):
    """begin_update -> per epoch: shuffle, prepare_epoch, nmb asynchronous steps (the loop of alg_ppo.learn / bench.py)."""
    import torch
    from robosumo_selfplay_amd import dist as sdist, model as model_mod, policies
    np.random.seed(3)
    spec = policies.PolicySpec(OB, AC, value_network="copy", activation="relu")
    m = model_mod.PPOModel(policy=spec, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, comm=comm)
    sdist.broadcast_params(m.params, comm)
    obs, act, ret, val, old = (torch.as_tensor(x[lo:hi]).cuda() for x in _batch())
    n = hi - lo
    w = torch.ones(n, dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)                                   # the SAME local permutation on every rank: rank r's k-th minibatch is then the
    m.begin_update(obs, ret, act, val, old, w)           # r-th part of the single process's k-th minibatch (see _ref_epochs)
    outs = []
    for ep in range(epochs):
        inds = torch.randperm(n, device="cuda", generator=gen).to(torch.int32)
        m.prepare_epoch(inds, n // nmb)
        for k in range(nmb):
            mb = inds[k * (n // nmb):(k + 1) * (n // nmb)]
            outs.append(m.train_indexed(1e-3, 0.2, obs, ret, act, val, old, w, mb, int(mb.numel()), sync=False, mb_index=k))
    m.end_update()
    torch.cuda.synchronize()
    sdist.assert_synced(m.params, comm)
    return m.params.cpu().numpy(), torch.stack(outs).cpu().numpy(), sorted(k for k in m._graphs)


def _ref_epochs(world, nmb=4, epochs=2):
    """Single process on the whole batch with the minibatches the `world` ranks form together: rank r holds rows [r per, (r+1) per) and
    every rank draws the same local permutation, so global minibatch k = union over r of (r per + local k-th slice)."""
    import torch
    from robosumo_selfplay_amd import model as model_mod, policies
    np.random.seed(3)
    spec = policies.PolicySpec(OB, AC, value_network="copy", activation="relu")
    m = model_mod.PPOModel(policy=spec, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5)
    obs, act, ret, val, old = (torch.as_tensor(x).cuda() for x in _batch())
    w = torch.ones(N, dtype=torch.float32, device="cuda")
    per = N // world
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    m.begin_update(obs, ret, act, val, old, w)
    for ep in range(epochs):
        inds = torch.randperm(per, device="cuda", generator=gen).to(torch.int32)
        for k in range(nmb):
            loc = inds[k * (per // nmb):(k + 1) * (per // nmb)]
            mb = torch.cat([loc + r * per for r in range(world)]).contiguous()
            m.train_indexed(1e-3, 0.2, obs, ret, act, val, old, w, mb, int(mb.numel()), sync=False)
    m.end_update()
    torch.cuda.synchronize()
    return m.params.cpu().numpy()


def _epoch_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    orig = dist.all_reduce

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    dist.all_reduce = counted
    per = N // world
    p, st, _ = _train_epochs(dist.group.WORLD, rank * per, (rank + 1) * per)
    dist.all_reduce = orig
    q.put((rank, p, st, calls["n"]))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_one_collective_per_step_with_epoch_moments():
    """2 ranks: per epoch ONE all-reduce of the [nmb, 3] advantage moments, then one fused all-reduce per optimiser step (was two per
    step); parameters equal a single-process run on the minibatches the two ranks form together (2e-5, as test_microbatches.py does
    for its full-batch / micro-batch pair at 3e-3)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_epoch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, p, st, ncalls = q.get(timeout=300)
        res[r] = (p, st, ncalls)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    nmb, epochs = 4, 2
    # per epoch: 1 (moments) + nmb (fused gradient) all-reduces; + the two of assert_synced
    assert res[0][2] == res[1][2] == epochs * (1 + nmb) + 2, res[0][2]
    assert np.array_equal(res[0][0], res[1][0])
    ref = _ref_epochs(2)
    assert np.allclose(res[0][0], ref, rtol=0, atol=2e-5), np.abs(res[0][0] - ref).max()
    assert np.isfinite(res[0][1]).all() and np.allclose(res[0][1], res[1][1])          # loss statistics are global: same on both ranks


def _rccl_graph_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    p, st, keys = _train_epochs(dist.group.WORLD, 0, N)
    q.put((p, st, keys))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_rccl_step_graph_holds_the_collective():
    """RCCL ('nccl', one rank -- a second rank needs a second GPU): the multi-GPU optimiser step -- adv_normalize, ppo_grad, the fused
    all-reduce, loss statistics -- is captured into ONE HIP graph with the collective inside and replayed per minibatch; same update as
    the single-GPU graph."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_graph_worker, args=(port, q))
    pr.start()
    p, st, keys = q.get(timeout=300)
    pr.join(timeout=120)
    assert pr.exitcode == 0
    assert keys and all(k[2] for k in keys), keys                    # the comm variant of the step graph was captured (no eager fallback)
    ref = _ref_epochs(1)
    assert np.allclose(p, ref, rtol=0, atol=2e-6), np.abs(p - ref).max()
    assert np.isfinite(st).all()


# ---- the same for the recurrent model: each rank back-propagates through its own env sequences -----------------------
LT, LN, LD, LA, LH = 6, 16, 17, 3, 64


def _lstm_batch():
    rng = np.random.RandomState(11)
    obs = rng.normal(0, 1, (LN, LT, LD)).astype(np.float32)
    masks = rng.rand(LN, LT) < 0.2
    act = rng.normal(0, 0.7, (LN, LT, LA)).astype(np.float32)
    ret = rng.normal(0, 1, (LN, LT)).astype(np.float32)
    val = rng.normal(0, 1, (LN, LT)).astype(np.float32)
    old = rng.normal(3.5, 0.3, (LN, LT)).astype(np.float32)
    S0 = rng.normal(0, 0.5, (LN, 2 * LH)).astype(np.float32)
    return obs, masks, act, ret, val, old, S0


def _lstm_train(comm, lo, hi, steps=3):
    from robosumo_selfplay_amd import dist as sdist, lstm_model
    np.random.seed(3)
    m = lstm_model.LstmPPOModel(policy=lstm_model.LstmSpec(LD, LA, LH), ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, comm=comm,
                                nbatch_act=hi - lo, nsteps=LT)
    sdist.broadcast_params(m.params, comm)
    obs, masks, act, ret, val, old, S0 = (x[lo:hi] for x in _lstm_batch())
    n = hi - lo
    flat = lambda x: np.ascontiguousarray(x).reshape(n * LT, *x.shape[2:])
    out = None
    for _ in range(steps):
        out = m.train(1e-3, 0.2, flat(obs), flat(ret), flat(masks), flat(act), flat(val), flat(old), None, np.ones(n * LT, np.float32),
                      states=S0)
    sdist.assert_synced(m.params, comm)
    return m.params.cpu().numpy(), np.array([float(x) for x in out[:5]])


def _lstm_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = LN // world
    p, st = _lstm_train(dist.group.WORLD, rank * per, (rank + 1) * per)
    q.put((rank, p, st))
    dist.destroy_process_group()


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_two_shards_match_single_process_recurrent():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_lstm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, p, st = q.get(timeout=300)
        res[r] = (p, st)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref_p, ref_st = _lstm_train(None, 0, LN)
    assert np.array_equal(res[0][0], res[1][0])
    assert np.allclose(res[0][0], ref_p, rtol=0, atol=3e-5), np.abs(res[0][0] - ref_p).max()
    assert np.allclose(res[0][1], ref_st, rtol=2e-3, atol=1e-5) and np.allclose(res[1][1], ref_st, rtol=2e-3, atol=1e-5)


@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
def test_run_py_two_ranks_end_to_end(tmp_path):
    """The whole distributed path through the CLI: `torch.distributed.run` starts two ranks of run.py (gloo over CUDA tensors, both on
    cuda:0 -- RCCL refuses two ranks on one device), each rank owns a shard of the envs, rolls out in its own fused launch and takes
    part in the per-step collectives; every rank ends with the same parameters as rank 0's checkpoints."""
    import subprocess
    import sys
    import joblib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SUMO_DIST_BACKEND="gloo", OPENAI_LOGDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(root, "run.py"), "--env", "RoboSumo-Ant-vs-Ant-v0", "--num_env", "32", "--num_timesteps", str(32 * 16 * 2),
           "--log_path", str(tmp_path), "--nsteps=16", "--nminibatches=4", "--noptepochs=2", "--log_interval=1", "--opponent_mode=latest"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    ck = os.path.join(str(tmp_path), "RoboSumo-Ant-vs-Ant-v0-0", "checkpoints")
    assert sorted(os.listdir(ck)) == ["00000", "00001", "00002"]
    p1, p2 = joblib.load(os.path.join(ck, "00001")), joblib.load(os.path.join(ck, "00002"))
    assert all(np.isfinite(a).all() for a in p2) and any(not np.array_equal(a, b) for a, b in zip(p1, p2))
    assert "update 2/2" in res.stdout
