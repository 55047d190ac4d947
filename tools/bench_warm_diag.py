#!/usr/bin/env python3
"""Development: why does a short bench window (--steps 20 --warmup 5) run slower than a long one?  Replays bench.py's rollout
loop with per-step host enqueue stamps and HIP events around every group's env launch, and prints one row per step.
    python tools/bench_warm_diag.py [--warmup 5] [--steps 30] [--groups 2]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import model as model_mod, policies  # noqa: E402
from robosumo_selfplay_amd.runner import Runner  # noqa: E402
from robosumo_selfplay_amd.vec_env import SumoVecEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--quiet", action="store_true", help="one summary line only")
    ap.add_argument("--stream-sync", action="store_true", help="end the window with per-group stream synchronize instead of a device synchronize")
    a = ap.parse_args()
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=a.envs, seed=1000, groups=a.groups)
    spec = policies.PolicySpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], value_network="copy", activation="relu")
    ms = [model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    r = Runner(env=env, models=ms, nsteps=128, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    ring = 8
    B = r._alloc_device(ring)
    G = a.groups
    K = a.steps
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(G)] for _ in range(K)]
    base = torch.cuda.Event(enable_timing=True)
    cur = {"k": None}
    orig = env.step_device_group

    def traced(g, act):
        k = cur["k"]
        if k is not None:
            ev[k][g][0].record()
        orig(g, act)
        if k is not None:
            ev[k][g][1].record()
    env.step_device_group = traced
    for k in range(a.warmup):
        r._step_device(B, k % ring, 1.0)
    torch.cuda.synchronize()
    st0 = env.stats()
    base.record()
    torch.cuda.synchronize()
    host = np.zeros((K, 2))
    t0 = time.perf_counter()
    stats = []
    for k in range(K):
        cur["k"] = k
        host[k, 0] = time.perf_counter() - t0
        r._step_device(B, (a.warmup + k) % ring, 1.0)
        host[k, 1] = time.perf_counter() - t0
    poll = os.environ.get("DIAG_POLL", "1") == "1"
    t_poll = None
    if poll:                                   # host-side completion time seen by polling the last events (no blocking wait)
        while not all(ev[K - 1][g][1].query() for g in range(G)):
            pass
        t_poll = time.perf_counter() - t0
    if a.stream_sync:
        for st in r._gstreams:
            st.synchronize()
    else:
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    torch.cuda.synchronize()
    gpu_end = max(base.elapsed_time(ev[K - 1][g][1]) for g in range(G))
    if a.quiet:
        print("enqueue done %7.3f ms | GPU clock: last env kernel ended %7.3f ms | host: wait returned %7.3f ms (%.3f ms/step)"
              % (host[K - 1, 1] * 1e3, gpu_end, wall * 1e3, wall / K * 1e3))
        env.close()
        return
    print("host: enqueue done %.3f ms, last env event seen complete by polling %s ms, synchronize returned %.3f ms"
          % (host[K - 1, 1] * 1e3, "%.3f" % (t_poll * 1e3) if t_poll else "-", wall * 1e3))
    st1 = env.stats()
    print("warmup %d, %d steps: %.3f ms/step wall; mean contacts/forward %.2f, newton/forward %.2f"
          % (a.warmup, K, wall / K * 1e3, (st1["contacts"] - st0["contacts"]) / (st1["forward"] - st0["forward"]),
             (st1["newton"] - st0["newton"]) / (st1["forward"] - st0["forward"])))
    print("step | host enqueue start..end (ms) | " + " | ".join("g%d env start..end (ms) dur" % g for g in range(G)))
    for k in range(K):
        row = "%3d | %8.3f %8.3f | " % (k, host[k, 0] * 1e3, host[k, 1] * 1e3)
        for g in range(G):
            s = base.elapsed_time(ev[k][g][0])
            e = base.elapsed_time(ev[k][g][1])
            row += "%8.3f %8.3f %6.3f | " % (s, e, e - s)
        print(row)
    env.close()


if __name__ == "__main__":
    main()
