"""Dev probe: two half-size engines stepped on two streams (no global barrier between the halves) vs one full-size engine."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from robosumo_selfplay_amd.vec_env import SumoVecEnv

N, K = 4096, 60
def run(groups):
    envs = [SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N // groups, seed=1 + 7 * g) for g in range(groups)]
    streams = [torch.cuda.Stream() for _ in range(groups)]
    acts = [[torch.randn((N // groups, 2, 8), device="cuda") for _ in range(8)] for _ in range(groups)]
    for g, e in enumerate(envs):
        e.reset_device()
    def loop(steps):
        for k in range(steps):
            for g, e in enumerate(envs):
                with torch.cuda.stream(streams[g]):
                    e.step_device(acts[g][k % 8])
    loop(100); torch.cuda.synchronize()
    t0 = time.time(); loop(K); torch.cuda.synchronize(); dt = time.time() - t0
    for e in envs: e.close()
    return N * K / dt
for groups in (1, 2, 4, 1, 2, 4):
    print("groups %d: %.0f env-steps/s" % (groups, run(groups)), flush=True)
