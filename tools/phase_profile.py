"""Dev tool: per-phase shader-cycle breakdown of sumo_step_kernel (needs csrc/libsumo_hip_prof.so built with
-DSUMO_PROFILE; run with SUMO_HIP_LIB pointing at it)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import mjcf
from robosumo_selfplay_amd.vec_env import SumoVecEnv

NAMES = ["zeroM+init", "kinematics", "com+cinert+cdof", "comvel+rne+bias", "coll broad", "coll narrow", "row params",
         "Jb build", "aref", "mass matrix", "qacc_smooth", "newton warm", "newton grad", "newton H", "newton chol+solve",
         "linesearch", "newton tail", "load", "(mj_step rest)", "epilogue", "probe0", "probe1", "probe2", "probe3"]
env_id = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Ant-vs-Ant-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = SumoVecEnv(env_id, num_envs=N, seed=1)
env.reset_device()
acts = [torch.randn((N, 2, env.engine.act_stride), device="cuda") for _ in range(8)]
for k in range(100):
    env.step_device(acts[k % 8])
torch.cuda.synchronize()
p0 = env.engine.profile(); s0 = env.engine.stats()
t0 = time.time(); K = 20
for k in range(K):
    env.step_device(acts[k % 8])
torch.cuda.synchronize(); dt = time.time() - t0
p1 = env.engine.profile(); s1 = env.engine.stats()
d = p1 - p0
nfwd = s1["forward"] - s0["forward"]
print("%s N=%d: %.2f ms/step, %.0f env-steps/s; newton/fwd %.2f contacts/fwd %.2f" % (env_id, N, dt / K * 1e3, N * K / dt,
      (s1["newton"] - s0["newton"]) / nfwd, (s1["contacts"] - s0["contacts"]) / nfwd))
tot = d.sum()
for n, v in zip(NAMES, d):
    if n.startswith("probe") and v == 0:
        continue
    print("%-20s %12.0f cyc/forward  %5.1f%%" % (n, v / nfwd, 100 * v / tot))
print("total cycles/forward %.0f" % (tot / nfwd))
