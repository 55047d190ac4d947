#!/usr/bin/env python3
"""One profiled workload per process, for rocprofv3 passes (tools/profile_r03.sh): put `python3 tools/prof_workload.py <name>` directly
after `--`.  Each workload primes its envs (100 rollout steps), then issues `--launches` fused launches of `--steps` steps each, so
every PMC pass sees the same few dispatches of ONE kernel:

  ant        sumo_rollout_kernel<28, 0>   Ant-vs-Ant, 4096 envs, MLP(64,64)             (BASELINE config 2, the bench default)
  spider     sumo_rollout_kernel<44, 0>   Spider-vs-Spider, 4096 envs, MLP(64,64)       (config 4)
  rec1024    sumo_rollout_kernel<28, 1>   Ant-vs-Ant, 1024 envs, LSTM(128), pool of 16  (config 5's one-GPU shard)
  rec4096    sumo_rollout_kernel<28, 1>   the same kernel with every wave slot filled (4096 envs)
  mfma       ppo_grad_kernel / ppo_selfplay_kernel: `--launches` x 8 calls of ppo_grad on a 16 384-row minibatch and of
             ppo_selfplay_forward on 4096 envs (the update's and the rollout step's matrix-core kernels)

Prints one JSON line with the wall-clock rate (not the judged number: bench.py's line is)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=["ant", "spider", "rec1024", "rec4096", "mfma"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--state-warmup", type=int, default=100)
    args = ap.parse_args()
    import numpy as np
    import torch
    from robosumo_selfplay_amd import defaults, lstm_model, mjcf
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.policies import build_policy
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    dev = torch.device("cuda", 0)
    hp = defaults.get_default_params("RoboSumo-Ant-vs-Ant-v0", "ppo")
    K = args.steps
    out = {"workload": args.name, "steps_per_launch": K, "launches": args.launches}
    if args.name in ("ant", "spider", "mfma"):
        env_id = "RoboSumo-Spider-vs-Spider-v0" if args.name == "spider" else "RoboSumo-Ant-vs-Ant-v0"
        N = 4096
        env = SumoVecEnv(env_id, num_envs=N, seed=77, model=mjcf.load_model(env_id), groups=1)
        spec = build_policy(env, "mlp", value_network=hp["value_network"], num_hidden=hp["num_hidden"], activation=hp["activation"])
        np.random.seed(0)
        ms = [PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=(i == 0 and args.name == "mfma"),
                       model_scope="model_%d" % i) for i in range(2)]
        ms[1].set_param_list(ms[0].get_param_list())
        r = Runner(env=env, models=ms, nsteps=max(K, 8), nagent=2, gamma=hp["gamma"], lam=hp["lam"], rho_bar=hp["rho_bar"], c_bar=hp["c_bar"])
        assert r.fused_ok()
        B = r._alloc_device(max(K, 8))
        step = lambda n: r._steps_fused(B, 0, n, 1.0)
    else:
        from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
        env_id, N, P = "RoboSumo-Ant-vs-Ant-v0", (1024 if args.name == "rec1024" else 4096), 16
        env = SumoVecEnv(env_id, num_envs=N, seed=78, model=mjcf.load_model(env_id), groups=1)
        spec = lstm_model.LstmSpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], 128)
        np.random.seed(0)
        learner = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=K, trainable=False)
        pool = LstmOpponentPool(spec, P, N, dev)
        for k in range(P):
            pool.set_snapshot(k, learner.get_param_list(), label="v%d" % k)
        pool.assign(np.arange(N // 16) % P)
        learner.seed(1); pool.seed(2)
        r = Runner(env=env, models=[learner, pool], nsteps=K, nagent=2, gamma=hp["gamma"], lam=hp["lam"], rho_bar=hp["rho_bar"], c_bar=hp["c_bar"])
        assert r.fused_lstm_ok()
        B = r._alloc_device(K)
        step = lambda n: r._steps_fused_lstm(B, 0, n, 1.0)
    done = 0
    while done < args.state_warmup:                       # priming, in launches of K (profile summaries skip them)
        step(min(K, args.state_warmup - done)); done += K
    r.join_groups()
    torch.cuda.synchronize(dev)
    if args.name == "mfma":
        import bench
        out_r = r.run(1)
        torch.cuda.synchronize(dev)
        T = max(K, 8)
        obs_b, ret_b, act_b, val_b, nlp_b = out_r[0][0].contiguous(), out_r[1][0], out_r[3][0], out_r[4][0], out_r[5][0]
        res = bench.mfma_probe(torch, dev, ms[0], ms[1], env, obs_b, ret_b, act_b, val_b, nlp_b, 16384, reps=8 * args.launches, use_graph=False)
        out["mfma_probe"] = res
    else:
        t0 = time.perf_counter()
        for _ in range(args.launches):
            step(K)
        r.join_groups()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        for E in env.engines:
            E.rollout_status()
        st = env.stats()
        out.update(env_id=env_id, envs=N, env_steps_per_s=N * K * args.launches / dt, ms_per_step=dt / (K * args.launches) * 1e3,
                   lds_bytes_per_env=env.engine.lds_bytes, dropped=st["dropped"], diverged=st["diverged"])
    print(json.dumps(out), flush=True)
    env.close()


if __name__ == "__main__":
    main()
