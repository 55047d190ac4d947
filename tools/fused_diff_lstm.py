#!/usr/bin/env python3
"""Development soak: fused recurrent rollout against the launch-per-evaluation path over long rollouts (bit-exact comparison of every
returned array, the recurrent states, the env states and the solver statistics).  usage: fused_diff_lstm.py [envs] [nsteps] [pool] [groups]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import test_gpu_lstm_rollout as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fused = T._run_pair(N, S, G, P, True)
step = T._run_pair(N, S, G, P, False)
T._assert_same(fused, step)
print("identical: %d envs x %d steps x 2 rollouts, pool %d, groups %d; episodes %d, diverged %d, aborts %d" % (
    N, S, P, G, len(fused[0][0][11]) + len(fused[0][1][11]), fused[3]["diverged"], fused[3]["rollout_aborts"]))
