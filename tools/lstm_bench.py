import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import alg_ppo
from robosumo_selfplay_amd.vec_env import SumoVecEnv
N, T = 1024, 128
env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=1)
m = alg_ppo.learn(network="lstm", env=env, seed=1, total_timesteps=N * T * 2, nagent=2, log_dir="/tmp/lstm_bench_log", verbose=True,
                  nsteps=T, nminibatches=8, noptepochs=4, lr=3e-4, gamma=0.995, lam=1.0, rho_bar=10.0, c_bar=1.0, opponent_mode="latest",
                  nlstm=128, anneal_bound=1000, log_interval=1)
print("history fps", m.history["fps"], "rollout_s", m.history["rollout_s"], "update_s", m.history["update_s"])
