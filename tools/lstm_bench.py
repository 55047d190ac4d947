"""Recurrent PPO2 on the config-5 shape of one GPU (Ant, 1024 envs x 128 steps, LSTM(128), 4 epochs x 8 whole-sequence
minibatches): rollout / update seconds per iteration.  usage: lstm_bench.py [groups] [pool] [updates]; SUMO_FUSED_ROLLOUT=0
selects the launch-per-evaluation rollout."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import alg_ppo
from robosumo_selfplay_amd.vec_env import SumoVecEnv
N, T = 1024, 128
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pool = int(sys.argv[2]) if len(sys.argv) > 2 else 1
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 3
env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=1, groups=groups)
m = alg_ppo.learn(network="lstm", env=env, seed=1, total_timesteps=N * T * updates, nagent=2, log_dir="/tmp/lstm_bench_log", verbose=False,
                  nsteps=T, nminibatches=8, noptepochs=4, lr=3e-4, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0, opponent_mode="latest",
                  nlstm=128, anneal_bound=1000, log_interval=1, opponent_pool=pool)
r, u = m.history["rollout_s"], m.history["update_s"]
print("groups %d pool %d fused %s: rollout_s %s update_s %s -> %.0f samples/s (last update)" % (
    groups, pool, os.environ.get("SUMO_FUSED_ROLLOUT", "1"), np.round(r, 4), np.round(u, 4), N * T / (r[-1] + u[-1])))
print("env stats", env.stats())
