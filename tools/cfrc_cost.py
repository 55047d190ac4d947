#!/usr/bin/env python3
"""Development: cost of cfrc_mode='rne_post' (second launch per env step) on the step-by-step env path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from robosumo_selfplay_amd.vec_env import SumoVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for mode in ("zero", "rne_post"):
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=1, groups=2, cfrc_mode=mode)
    env.reset_device()
    acts = [torch.randn((N, 2, env.engine.act_stride), device="cuda") for _ in range(8)]
    for k in range(100):
        env.step_device(acts[k % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(40):
        env.step_device(acts[k % 8])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("cfrc_mode %-8s: %.3f ms per env step of %d envs, %.0f env-steps/s" % (mode, dt / 40 * 1e3, N, N * 40 / dt))
    env.close()
