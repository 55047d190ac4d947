#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: vectorised RoboSumo env steps per second.

A "step" is ONE pass of the hot path over one batch: one rollout step of the PPO2 self-play Runner -- the 5 policy /
value evaluations of reference runner.py:66-97 (4 fused MFMA launches), one vectorised env step (5 x RK4 mj_step, game
rules, rewards, done, auto-reset, observation write) and the reward curriculum -- for `--envs` environments per GPU
(BASELINE configs[1]: RoboSumo Ant-vs-Ant, 4096 envs, MLP(64,64), 1 MI355X).  Inputs are synthetic and resident in HBM:
random-init networks of the reference architecture (zero-init logstd => N(0,1)-scale actions, the initial policy's law)
and env states warmed up from the env's own reset distribution.  After the timed region one full PPO2
update (rollout of --ppo-nsteps + noptepochs x nminibatches optimiser steps) is timed for the "+ PPO2 iters/sec" half
of the metric and reported under config.  Multi-GPU: one process per GPU (torch.distributed / RCCL), envs sharded with
no data-path collective -> weak scaling; value = total env-steps of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from robosumo_selfplay_amd import hostcfg  # noqa: E402  (first: caps the host thread pools at the cgroup CPU quota)

hostcfg.apply()
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_env_step(m):
    """SURVEY.md §8(d) with float64 device state: read qpos,qvel,warm + actions; write qpos,qvel,warm + obs (2 agents)
    + reward components + done + counters."""
    nq, nv, nu = m.nq, m.nv, m.nu
    obs = sum(m.obs_dims)
    return 8 * 2 * (nq + 2 * nv) + 4 * nu + 4 * obs + 8 * 16 + 2 + 8 + 2 * 8 + 4


def cpu_baseline(model, states, actions, steps, threads):
    """Times the CPU oracle (the float64 restatement; the reference's MuJoCo path cannot run here) on a bounded
    sample of the same workload: the first len(states[0]) envs of the GPU batch, same states, same action law."""
    from oracle.oracle import OracleSim, build
    build()
    n = states[0].shape[0]
    sim = OracleSim(model, n)
    sim.set_state(*states)
    sim.set_seeds(np.arange(n, dtype=np.uint64))
    sim.step(actions[0], nthreads=threads)
    t0 = time.perf_counter()
    for k in range(steps):
        sim.step(actions[k % len(actions)], nthreads=threads)
    dt = time.perf_counter() - t0
    return n * steps / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--env-id", default="RoboSumo-Ant-vs-Ant-v0")
    ap.add_argument("--groups", type=int, default=2, help="env groups per GPU, each stepped on its own stream (1 = one launch for all envs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=512)
    ap.add_argument("--cpu-sample-steps", type=int, default=60)
    ap.add_argument("--ppo-nsteps", type=int, default=128, help="rollout length of the PPO2 update timed after the main region (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the cgroup CPU quota of this job (hostcfg.cpu_quota)")
    ap.add_argument("--state-warmup", type=int, default=100,
                    help="untimed rollout steps that bring the env states from the reset law to the steady workload (SURVEY.md 8(d): >= 100)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU through torch.distributed.run, started BEFORE this process touches a GPU (a process that
        # has initialised HIP must never exec; children are ordinary subprocesses).  Rank 0's JSON line passes through.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    from robosumo_selfplay_amd import mjcf
    from robosumo_selfplay_amd.vec_env import SumoVecEnv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")      # "nccl" is RCCL; "gloo" only to rehearse N>1 on a 1-GPU box
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(1, ndev)
        torch.cuda.set_device(local_rank)
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model = mjcf.load_model(args.env_id)
    N = args.envs
    env = SumoVecEnv(args.env_id, num_envs=N, seed=1000 + rank * N, device=local_rank, model=model, groups=args.groups)
    from robosumo_selfplay_amd import defaults
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.policies import build_policy
    from robosumo_selfplay_amd.runner import Runner, anneal_alpha
    np.random.seed(0)                                    # identical random-init weights on every rank
    hp = defaults.get_default_params(args.env_id)
    spec = build_policy(env, "mlp", value_network=hp["value_network"], num_hidden=hp["num_hidden"], activation=hp["activation"])
    group = dist.group.WORLD if dist is not None else None
    learner = PPOModel(policy=spec, ent_coef=hp["ent_coef"], vf_coef=0.5, max_grad_norm=0.5, model_scope="model_0",
                       device=local_rank, comm=group)
    opponent = PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False, model_scope="model_1",
                        device=local_rank)
    opponent.set_param_list(learner.get_param_list())
    learner.act_model.seed(1234 + rank)
    opponent.act_model.seed(4321 + rank)
    runner = Runner(env=env, models=[learner, opponent], nsteps=args.ppo_nsteps, nagent=2, gamma=hp["gamma"], lam=hp["lam"],
                    rho_bar=hp["rho_bar"], c_bar=hp["c_bar"], anneal_bound=hp["anneal_bound"])
    alpha = anneal_alpha(1, hp["anneal_bound"])

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ring = 8
    B = runner._alloc_device(ring)
    # workload priming (not a timing knob): the reset law drops every agent from z = 1.25 at the same instant; >= 100 rollout
    # steps under the initial policy's N(0,1)-scale actions spread the episodes out and make contacts active
    for k in range(args.state_warmup):
        runner._step_device(B, k % ring, alpha)
    torch.cuda.synchronize(dev)
    for k in range(args.warmup):                                   # timing warm-up
        runner._step_device(B, (args.state_warmup + k) % ring, alpha)
    torch.cuda.synchronize(dev)
    st0 = env.stats()
    states = None
    if rank == 0 and not args.no_cpu_baseline:
        q, v, w, c = env.engine.get_state()
        ns = min(args.cpu_sample_envs, env.group_size)
        states = (q[:ns].copy(), v[:ns].copy(), w[:ns].copy(), c[:ns].copy())

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    thr0 = hostcfg.throttle_stats()
    t0 = time.perf_counter()
    ev_every = int(os.environ.get("BENCH_EVENT_EVERY", "1"))       # HIP events around the env launch of every n-th step (0 = none)
    timed = [k for k in range(args.steps) if ev_every and k % ev_every == 0]
    k0 = args.state_warmup + args.warmup
    for k in range(args.steps):
        runner._step_device(B, (k0 + k) % ring, alpha, env_events=(ev0[k], ev1[k]) if ev_every and k % ev_every == 0 else None)
    barrier()
    dt = time.perf_counter() - t0
    thr1 = hostcfg.throttle_stats()
    kern_ms = float(np.mean([ev0[k].elapsed_time(ev1[k]) for k in timed])) if timed else float("nan")
    st1 = env.stats()
    sample_acts = [B["act"][:, k].permute(1, 0, 2).contiguous() for k in range(ring)]   # [N, 2, A] per step

    # ---- "+ PPO2 iters/sec": one full update, outside the timed region above
    ppo = None
    if args.ppo_nsteps > 0:
        T, nmb, nep = args.ppo_nsteps, hp["nminibatches"], hp["noptepochs"]
        for timed_update in (False, True):     # the first update pays one-time costs (graph capture, lazy initialisation): untimed
            barrier()
            tu = time.perf_counter()
            out = runner.run(1)
            torch.cuda.synchronize(dev)
            t_roll = time.perf_counter() - tu
            obs_b, ret_b, act_b, val_b, nlp_b = out[0][0].contiguous(), out[1][0], out[3][0], out[4][0], out[5][0]
            nb = N * T
            wts = torch.ones(nb, dtype=torch.float32, device=dev)
            learner.begin_update(obs_b, ret_b, act_b, val_b, nlp_b, wts)
            for ep in range(nep):
                inds = torch.randperm(nb, device=dev).to(torch.int32)      # device-side shuffle (np.random.shuffle in alg_ppo.py:375)
                for start in range(0, nb, nb // nmb):
                    mb = inds[start:start + nb // nmb]
                    learner.train_indexed(hp["lr"], hp["cliprange"], obs_b, ret_b, act_b, val_b, nlp_b, wts, mb, int(mb.numel()), sync=False)
            learner.end_update()
            barrier()
            t_upd = time.perf_counter() - tu
            tt = torch.tensor([t_upd, t_roll], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ppo = {"iters_per_sec": 1.0 / float(tt[0].item()), "nsteps": T, "nminibatches": nmb, "noptepochs": nep,
                   "rollout_s": float(tt[1].item()), "sgd_s": float(tt[0].item() - tt[1].item()),
                   "samples_per_iter": nb * world}

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / dt_max
        B = algorithmic_bytes_per_env_step(model)
        traffic = None
        try:   # PMC traffic is collected offline with rocprofv3 (profiles/README.md); reported only for the profiled config
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")))   # newest collection
            if ("<%d>" % model.nv) in pj["kernel"] and pj["envs"] == env.group_size:
                traffic = pj["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        achieved = B * env.group_size / (kern_ms * 1e-3) / 1e9      # per launch: one env group (HIP events on its stream)
        nfwd = max(1.0, st1["forward"] - st0["forward"])
        out = {
            "metric": "env-steps/sec (whole node), RoboSumoAnts-v0 4096 envs, + PPO2 iters/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d envs per GPU, MLP(64,64) policy+value: one self-play rollout step per bench step "
                                   "(5 policy/value evaluations + env step of frame_skip 5 x RK4 + reward mix), random-init "
                                   "networks, auto-reset on" % (args.env_id, N),
                       "ppo2": ppo,
                       "state_warmup_steps": args.state_warmup,
                       "envs_per_gpu": N, "env_groups_per_gpu": env.groups, "total_envs": N * world, "parallelism": "env-shard x%d" % world,
                       "mean_contacts_per_forward": (st1["contacts"] - st0["contacts"]) / nfwd,
                       "mean_newton_iters_per_forward": (st1["newton"] - st0["newton"]) / nfwd,
                       "lds_bytes_per_env": env.engine.lds_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "sumo_step_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": B,
                         "envs_per_launch": env.group_size, "concurrent_launches": env.groups,
                         "aggregate_achieved": B * value / max(1, world) / 1e9,
                         "note": "latency/ALU-bound physics: ~20 forward-dynamics solves per 2.4 KB of state traffic"},
        }
        out["host"] = {"cpu_model": hostcfg.cpu_model(), "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": hostcfg.cpu_quota(),
                       "pool_threads": hostcfg.apply(),
                       "throttled_periods_in_timed_region": None if thr0 is None or thr1 is None else thr1[0] - thr0[0]}
        if states is not None:
            threads = args.cpu_threads or hostcfg.cpu_quota()
            cpu_acts = [a[:states[0].shape[0]].cpu().numpy() for a in sample_acts]
            v = cpu_baseline(model, states, cpu_acts, args.cpu_sample_steps, threads)
            out["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": threads, "kind": "port",
                                   "sample": "%d envs x %d steps of the same warmed-up workload, OpenMP over envs"
                                             % (states[0].shape[0], args.cpu_sample_steps)}
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
