#!/usr/bin/env python3
"""Development soak: many PPO2 self-play updates end to end (learn() with the round's default paths) -- finite losses, no diverged env
steps beyond the guard's count, no aborted hand-over waits.  usage: soak_learn.py mlp|lstm [updates] [envs]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import alg_ppo, defaults
from robosumo_selfplay_amd.vec_env import SumoVecEnv
kind = sys.argv[1] if len(sys.argv) > 1 else "mlp"
updates = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = int(sys.argv[3]) if len(sys.argv) > 3 else (4096 if kind == "mlp" else 1024)
T = 128
env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=5)
kw = defaults.get_default_params("RoboSumo-Ant-vs-Ant-v0", "ppo")
kw.update(nsteps=T, log_interval=10 ** 9, verbose=False, save_interval=5)
if kind == "lstm":
    for k in ("value_network", "num_hidden", "num_layers", "activation"):
        kw.pop(k, None)
    kw.update(nminibatches=8, noptepochs=4, nlstm=128, opponent_pool=16)
else:
    kw.update(opponent_pool=4)
t0 = time.time()
with tempfile.TemporaryDirectory() as d:
    m = alg_ppo.learn(network=kind, env=env, seed=3, total_timesteps=N * T * updates, nagent=2, log_dir=d, **kw)
dt = time.time() - t0
torch.cuda.synchronize()
h = m.history
loss = np.asarray(h["lossvals"], np.float64)
st = env.stats()
print("%s: %d updates of %d x %d in %.1f s (%.0f samples/s incl. checkpoints and opponent loading); losses finite: %s; last loss row %s" % (
    kind, len(loss), N, T, dt, N * T * len(loss) / dt, bool(np.isfinite(loss).all()), np.round(loss[-1], 4)))
print("rollout %.3f s / update %.3f s (median); approxkl %.4f, clip fraction %.3f (last); env stats: diverged %d, rollout_aborts %d, dropped contacts %d, max contacts %d" % (
    float(np.median(h["rollout_s"])), float(np.median(h["update_s"])), float(h["approxkl"][-1]), float(h["ppo_clip_frac"][-1]),
    st["diverged"], st["rollout_aborts"], st["dropped"], st["max_ncon"]))
print("params finite:", bool(torch.isfinite(m.params).all()), "| opponent versions of the last update:", h["opponent_versions"][-1][:6])
env.close()
