#!/usr/bin/env python3
"""Development: where does a fused rollout differ from the step-by-step one? (names, agents, envs, time steps)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_ppo as T
env_id, N, TT, groups = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Ant-vs-Ant-v0", int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 12, int(sys.argv[4]) if len(sys.argv) > 4 else 2
names = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opp_neglogpacs", "opp_obs", "opp_actions", "states",
         "epinfos", "off_policy_ratio", "off_env_ratio", "total_ratio"]
for rep in range(3):
    fo, fs, fstat = T._rollout_pair(env_id, N, TT, groups, True)
    so, ss, sstat = T._rollout_pair(env_id, N, TT, groups, False)
    bad = 0
    for ri, (f, s_) in enumerate(zip(fo, so)):
        for k, (x, y) in enumerate(zip(f, s_)):
            if torch.is_tensor(x):
                if not torch.equal(x, y):
                    d = (x != y)
                    while d.dim() > 2:
                        d = d.any(-1)
                    idx = torch.nonzero(d).cpu().numpy()
                    if d.dim() == 2:      # [agent, env*T + t]
                        print("rep %d run %d %s: %d entries differ; (agent, env, t) first: %s" % (rep, ri, names[k], len(idx), [(int(a), int(j) // TT, int(j) % TT) for a, j in idx[:6]]))
                    else:
                        print("rep %d run %d %s: %d entries differ; (env, t) first: %s" % (rep, ri, names[k], len(idx), [(int(j) // TT, int(j) % TT) for (j,) in idx[:6]]))
                    bad += 1
                    if names[k] == "obs":       # which entries of the first differing observation, and by how much
                        a_, j = idx[0]
                        xo, yo = x[int(a_), int(j)].cpu().numpy(), y[int(a_), int(j)].cpu().numpy()
                        w = np.nonzero(xo != yo)[0]
                        print("   first differing obs (agent %d env %d t %d): entries %s, |diff| max %.3g; fused %s stepwise %s" % (
                            a_, int(j) // TT, int(j) % TT, w[:12], np.abs(xo - yo).max(), xo[w[:4]], yo[w[:4]]))
                        t0_ = int(j) % TT
                        for ag in (0, 1):      # both agents' rows of that env and step: which blocks differ
                            xa, ya = x[ag, int(j)].cpu().numpy(), y[ag, int(j)].cpu().numpy()
                            wa = np.nonzero(xa != ya)[0]
                            print("   agent %d env %d t %d: %d entries differ: %s; max |diff| %.3g" % (ag, int(j) // TT, t0_, len(wa), wa[:40], np.abs(xa - ya).max() if len(wa) else 0.0))
                        ts = np.array([int(jj) % TT for _, jj in idx])
                        print("   differing rows per step:", np.bincount(ts, minlength=TT))
                    if names[k] == "rewards":
                        X, Y = x[0].reshape(N, TT).cpu().numpy(), y[0].reshape(N, TT).cpu().numpy()
                        for a_, j in idx[:4]:
                            e_, t_ = int(j) // TT, int(j) % TT
                            print("   env %d: fused    %s\n           stepwise %s" % (e_, np.array2string(X[e_], precision=3), np.array2string(Y[e_], precision=3)))
            elif x != y:
                print("rep %d run %d %s differ: %d vs %d items" % (rep, ri, names[k], len(x), len(y))); bad += 1
    for g, (a, b) in enumerate(zip(fs, ss)):
        for nm, x, y in zip(("qpos", "qvel", "warm", "counters"), a, b):
            if not np.array_equal(x, y):
                print("rep %d state group %d %s differs in envs %s" % (rep, g, nm, np.nonzero((x != y).any(1))[0][:8]))
    print("rep %d: %d arrays differ; aborts %s" % (rep, bad, fstat.get("rollout_aborts")))
