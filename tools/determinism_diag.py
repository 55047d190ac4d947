#!/usr/bin/env python3
"""Development: run the same seeded 4096-env trajectory several times (fresh engine each) and report where repeats differ from the first
run: step, envs, which arrays, magnitudes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import mjcf
from robosumo_selfplay_amd.vec_env import SumoVecEnv
env_id = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Ant-vs-Ant-v0"
reps, T, N = int(sys.argv[2]) if len(sys.argv) > 2 else 6, 6, 4096
m = mjcf.load_model(env_id)
A = int(max(m.act_dims))          # the action tensor has the wider agent's width (act_stride)
g = torch.Generator(device="cpu").manual_seed(0)
acts = torch.randn((T, N, 2, A), generator=g).to("cuda")
ref = None
for rep in range(reps):
    env = SumoVecEnv(env_id, num_envs=N, seed=7, model=m)
    o0 = env.reset_device().cpu().numpy().copy()
    traj = [("reset", o0, *[x.copy() for x in env.engine.get_state()[:3]])]
    for t in range(T):
        obs, info, done, *_ = env.step_device(acts[t].contiguous())
        torch.cuda.synchronize()
        q, v, w, c = env.engine.get_state()
        traj.append((t, obs.cpu().numpy().copy(), q.copy(), v.copy(), w.copy(), info.cpu().numpy().copy()))
    st = env.stats()
    env.close()
    if ref is None:
        ref = traj
        print("rep 0: reference; stats", {k: st[k] for k in ("forward", "newton", "contacts", "dropped", "diverged")})
        continue
    nbad = 0
    for a, b in zip(ref, traj):
        for nm, x, y in zip(("obs", "qpos", "qvel", "warm", "info"), a[1:], b[1:]):
            if not np.array_equal(x, y):
                d = (x != y)
                envs = np.nonzero(d.reshape(N, -1).any(1))[0]
                print("rep %d step %s %s: %d envs differ %s; max |diff| %.3g; entries of env %d: %s" % (
                    rep, a[0], nm, len(envs), envs[:8], np.abs(x.astype(np.float64) - y.astype(np.float64)).max(), envs[0],
                    np.nonzero(d.reshape(N, -1)[envs[0]])[0][:12]))
                nbad += 1
    print("rep %d: %d arrays differ; stats" % (rep, nbad), {k: st[k] for k in ("forward", "newton", "contacts", "dropped", "diverged")})
