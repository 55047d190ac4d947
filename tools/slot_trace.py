#!/usr/bin/env python3
"""Development: how full are the GPU's wave slots during a grouped rollout?  Every env wave stamps its start / end
(sumo_debug_trace, 100 MHz wall clock); the script replays the stamps of a window of steps into an occupancy curve.
    python tools/slot_trace.py [--envs 4096] [--groups 2] [--steps 40]
"""
import argparse
import os
import sys

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
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=100)
    a = ap.parse_args()
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=a.envs, seed=0, groups=a.groups)
    spec = policies.PolicySpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], value_network="copy", activation="relu")
    ms = [model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    r = Runner(env=env, models=ms, nsteps=a.steps, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    B = r._alloc_device(a.steps)
    for s in range(a.warmup):
        r._step_device(B, s % a.steps, 1.0)
    r.join_groups(); torch.cuda.synchronize()
    G = a.groups
    per = a.envs // G
    # one stamp buffer per (step, group): the engines are re-pointed before every step (host-side pointer swap only)
    stamps = torch.zeros((a.steps, a.envs, 4), dtype=torch.int64, device=env.device)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in range(a.steps):
        for g, E in enumerate(env.engines):
            E.debug_trace(stamps[s, g * per:(g + 1) * per].data_ptr())
        r._step_device(B, s, 1.0)
    r.join_groups(); t1.record(); torch.cuda.synchronize()
    for E in env.engines:
        E.debug_trace(None)
    wall_ms = t0.elapsed_time(t1)
    st = stamps.cpu().numpy().astype(np.int64)
    lo = st[2:, :, 0].min(); hi = st[2:, :, 1].max()               # skip the first steps (pipeline fill)
    busy = (st[2:, :, 1] - st[2:, :, 0]).sum()
    span = hi - lo
    slots = 2048
    print("envs %d groups %d: %.3f ms/step; wave time mean %.3f ms (min %.3f max %.3f); slot-time busy %.1f %% of %d slots"
          % (a.envs, G, wall_ms / a.steps, (st[2:, :, 1] - st[2:, :, 0]).mean() / 1e5, (st[2:, :, 1] - st[2:, :, 0]).min() / 1e5,
             (st[2:, :, 1] - st[2:, :, 0]).max() / 1e5, 100.0 * busy / (span * slots), slots))
    dur = (st[2:, :, 1] - st[2:, :, 0]) / 1e5
    print("wave time percentiles (ms): " + " ".join("p%d=%.3f" % (q, np.percentile(dur, q)) for q in (1, 10, 50, 90, 99, 99.9)))
    done = B["ep_done"].cpu().numpy().astype(bool)[2:]
    print("steps that ended an episode (reset inside the launch): %d of %d; wave time mean %.3f ms with reset, %.3f ms without; max without %.3f"
          % (done.sum(), done.size, dur[done].mean() if done.any() else 0.0, dur[~done].mean(), dur[~done].max()))
    newton = (st[2:, :, 2] & 0xFFFFFFFF); ncon = (st[2:, :, 2] >> 32); dense = (st[2:, :, 3] & 0xFFFF); cross = (st[2:, :, 3] >> 16) & 0xFFFF; nefc = st[2:, :, 3] >> 32
    slow = dur > np.percentile(dur, 99.5)
    print("all waves:   newton %.1f, contacts %.1f, rows %.1f, dense-path forwards %.2f of 20 per step" % (newton.mean(), ncon.mean(), nefc.mean(), dense.mean()))
    print("slowest 0.5%%: newton %.1f, contacts %.1f, rows %.1f, dense-path forwards %.2f; wave time %.3f ms"
          % (newton[slow].mean(), ncon[slow].mean(), nefc[slow].mean(), dense[slow].mean(), dur[slow].mean()))
    print("dense-path forwards: %d, of which with a contact between the two agents: %d (%.1f %%)" % (dense.sum(), cross.sum(), 100.0 * cross.sum() / max(1, dense.sum())))
    for lo_, hi_ in ((0, 0), (1, 10), (11, 19), (20, 20)):
        m = (dense >= lo_) & (dense <= hi_)
        if m.any():
            print("  dense forwards %2d..%2d: %6.2f %% of waves, wave time mean %.3f ms, newton %.1f, contacts %.1f" % (lo_, hi_, 100.0 * m.mean(), dur[m].mean(), newton[m].mean(), ncon[m].mean()))
    A_ = np.stack([np.ones(dur.size), newton.ravel(), ncon.ravel(), dense.ravel()], 1)
    coef = np.linalg.lstsq(A_, dur.ravel(), rcond=None)[0]
    print("least squares: wave ms = %.3f + %.5f newton + %.5f contacts + %.4f dense forwards" % tuple(coef))
    # the slowest wave of each launch: how far behind the launch's median
    for g in range(G):
        sl = slice(g * per, (g + 1) * per)
        d = dur[:, sl]
        print("group %d: per-launch max wave time mean %.3f ms, median wave %.3f ms" % (g, d.max(1).mean(), np.median(d, 1).mean()))
    # occupancy curve over two steps in the middle, 20 us bins
    mid = a.steps // 2
    w0 = st[mid, :, 0].min(); w1 = st[mid + 1, :, 1].max()
    ev = []
    for s in range(max(0, mid - 2), min(a.steps, mid + 4)):
        ev.append(np.stack([st[s, :, 0], np.ones(a.envs, np.int64)], 1)); ev.append(np.stack([st[s, :, 1], -np.ones(a.envs, np.int64)], 1))
    ev = np.concatenate(ev); ev = ev[np.argsort(ev[:, 0], kind="stable")]
    occ = np.cumsum(ev[:, 1])
    bins = np.arange(w0, w1, 2000)
    idx = np.searchsorted(ev[:, 0], bins, side="right") - 1
    print("occupancy (waves resident) every 20 us over steps %d..%d:" % (mid, mid + 1))
    print(" ".join("%d" % occ[max(i, 0)] for i in idx))
    for g in range(G):
        sl = slice(g * per, (g + 1) * per)
        print("group %d step %d: first start %.0f us, last start %.0f us, first end %.0f us, last end %.0f us (relative)"
              % (g, mid, (st[mid, sl, 0].min() - w0) / 100, (st[mid, sl, 0].max() - w0) / 100, (st[mid, sl, 1].min() - w0) / 100,
                 (st[mid, sl, 1].max() - w0) / 100))
    env.close()


if __name__ == "__main__":
    main()
