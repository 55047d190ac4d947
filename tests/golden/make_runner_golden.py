#!/usr/bin/env python3
"""Generates tests/golden/runner_*.npz by running the REFERENCE's Runner (imported from /root/reference/runner.py, the
one reference module that imports here: numpy/tqdm/matplotlib only) on the deterministic fake env/models of
tests/fake_rollout.py.  Two consecutive run() calls per case (the second starts from carried-over obs/dones).

Only outputs are stored; the inputs are regenerated from tests/fake_rollout.py.  Run from the repo root:
    python tests/golden/make_runner_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import matplotlib
matplotlib.use("Agg")
if not hasattr(np, "bool"):
    np.bool = bool  # runner.py:159 uses the alias removed in numpy 1.24
sys.path.insert(0, "/root/reference")
import runner as ref_runner  # noqa: E402
from fake_rollout import CASES, make_case  # noqa: E402

NAMES = ["obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards", "opponent_neglogpacs", "opponent_obs",
         "opponent_actions", "states", "epinfos", "off_policy_ratio", "off_env_ratio", "ratio"]

for c in CASES:
    env, models, kw, update = make_case(c)
    r = ref_runner.Runner(env=env, models=models, **kw)
    out = {}
    for call in range(2):
        res = r.run(update + call)
        assert len(res) == 15
        for nm, v in zip(NAMES, res):
            if nm == "states":
                assert v is None
            elif nm == "epinfos":
                out["c%d_epinfo_r" % call] = np.array([e["r"] for e in v], np.float64)
                out["c%d_epinfo_l" % call] = np.array([e["l"] for e in v], np.int64)
            else:
                out["c%d_%s" % (call, nm)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "runner_%s.npz" % c[0]), **out)
    print(c[0], {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith("c0_")})
