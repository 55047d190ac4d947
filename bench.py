#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: vectorised RoboSumo env steps per second.

A "step" is ONE pass of the hot path over one batch: one vectorised env step (5 x RK4 mj_step, game rules, rewards,
done, auto-reset, observation write) for `--envs` environments per GPU (BASELINE configs[1]: RoboSumo Ant-vs-Ant,
4096 envs, 1 MI355X), on synthetic inputs resident in HBM (states warmed up from the env's own reset distribution
under N(0,1) actions).  Multi-GPU: one process per GPU (torch.distributed / RCCL), envs sharded with
no data-path collective -> weak scaling; value = total env-steps of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=512)
    ap.add_argument("--cpu-sample-steps", type=int, default=60)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(os.cpu_count(), 16): a 1-GPU box's CPU share")
    args = ap.parse_args()

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
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model = mjcf.load_model(args.env_id)
    N = args.envs
    env = SumoVecEnv(args.env_id, num_envs=N, seed=1000 + rank * N, device=local_rank, model=model)
    env.reset_device()
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = 16
    acts = [torch.randn((N, 2, env.engine.act_stride), generator=gen, device=dev, dtype=torch.float32) for _ in range(n_act)]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        env.step_device(acts[k % n_act])
    torch.cuda.synchronize(dev)
    st0 = env.engine.stats()
    states = None
    if rank == 0 and not args.no_cpu_baseline:
        q, v, w, c = env.engine.get_state()
        ns = min(args.cpu_sample_envs, N)
        states = (q[:ns].copy(), v[:ns].copy(), w[:ns].copy(), c[:ns].copy())

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev0[k].record()
        env.step_device(acts[k % n_act])
        ev1[k].record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    st1 = env.engine.stats()

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / dt_max
        B = algorithmic_bytes_per_env_step(model)
        achieved = B * N / (kern_ms * 1e-3) / 1e9
        nfwd = max(1.0, st1["forward"] - st0["forward"])
        out = {
            "metric": "env-steps/sec (whole node), RoboSumo Ant-vs-Ant 4096 envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d envs per GPU, one vectorised env step (frame_skip 5 x RK4) per bench step, "
                                   "N(0,1) actions, auto-reset on" % (args.env_id, N),
                       "envs_per_gpu": N, "total_envs": N * world, "parallelism": "env-shard x%d" % world,
                       "mean_contacts_per_forward": (st1["contacts"] - st0["contacts"]) / nfwd,
                       "mean_newton_iters_per_forward": (st1["newton"] - st0["newton"]) / nfwd,
                       "lds_bytes_per_env": env.engine.lds_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "sumo_step_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": B,
                         "note": "latency/ALU-bound physics: ~20 forward-dynamics solves per 2.4 KB of state traffic"},
        }
        if states is not None:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            cpu_acts = [a[:states[0].shape[0]].cpu().numpy() for a in acts]
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
