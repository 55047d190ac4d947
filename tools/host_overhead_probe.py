"""Dev probe: host time to enqueue one rollout step (no synchronisation) vs groups."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import policies
from robosumo_selfplay_amd.model import PPOModel
from robosumo_selfplay_amd.runner import Runner
from robosumo_selfplay_amd.vec_env import SumoVecEnv
for G in [int(x) for x in (sys.argv[1:] or ["1", "2", "4"])]:
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=4096, seed=1, groups=G)
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    ms = [PPOModel(policy=spec, trainable=(i == 0)) for i in range(2)]
    r = Runner(env=env, models=ms, nsteps=8, nagent=2, gamma=0.99, lam=0.95, rho_bar=1.0, c_bar=1.0)
    B = r._alloc_device(8)
    for k in range(20): r._step_device(B, k % 8, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50): r._step_device(B, k % 8, 1.0)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("groups %d: host enqueue %.3f ms/step, total %.3f ms/step" % (G, th / 50 * 1e3, tt / 50 * 1e3), flush=True)
    env.close()
