#!/usr/bin/env python3
"""Development (engine built with -DSUMO_DBG_DUMP, SUMO_HIP_LIB pointing at it): intermediate vectors of env 0's first two
forward evaluations -- qpos, qvel, ctrl, qfrc_smooth, qacc_smooth, (ncon, nefc, nlim), qacc, warm start -- from the fused rollout
launch and from the step-by-step launch on the same state and policies; prints the first quantity that differs."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import capi, policies
from robosumo_selfplay_amd.model import PPOModel
from robosumo_selfplay_amd.runner import Runner
from robosumo_selfplay_amd.vec_env import SumoVecEnv
NAMES = ["qpos", "qvel", "ctrl", "qfrc_smooth", "qacc_smooth", "ncon/nefc/nlim", "qacc", "warm"]
L = capi.lib()
L.sumo_debug_dump.argtypes = [C.c_void_p, C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = {}
for fused in (True, False):
    env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=3)
    np.random.seed(5)
    spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
    ms = [PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    rng = np.random.RandomState(1)
    for m in ms:
        m.set_param_list([p + rng.normal(0, 0.1, p.shape).astype(np.float32) for p in m.get_param_list()])
    ms[0].act_model.seed(1); ms[1].act_model.seed(2)
    r = Runner(env=env, models=ms, nsteps=1, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    r.fused_rollout = fused
    buf = torch.zeros(20 * 8 * 64, dtype=torch.float64, device="cuda")
    assert L.sumo_debug_dump(env.engine.h, buf.data_ptr()) == 0
    o = r.run(1)
    torch.cuda.synchronize()
    out[fused] = (buf.cpu().numpy().reshape(20, 8, 64).copy(), o[3].cpu().numpy().copy())
    L.sumo_debug_dump(env.engine.h, None)
    env.close()
f, s = out[True][0], out[False][0]
print("recorded actions equal:", np.array_equal(out[True][1], out[False][1]))
np.set_printoptions(precision=6, linewidth=200)
for fw in range(20):
    for k in range(8):
        d = np.abs(f[fw, k] - s[fw, k])
        bad = np.nonzero(f[fw, k] != s[fw, k])[0]
        if len(bad):
            print("forward %d %-16s DIFFERS at lanes %s max %.3g" % (fw, NAMES[k], bad[:32], d.max()))
            print("    fused   ", f[fw, k][:30]); print("    stepwise", s[fw, k][:30])
            sys.exit(0)
print("all 20 forwards equal")
