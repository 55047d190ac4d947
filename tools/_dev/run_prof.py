import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from robosumo_selfplay_amd import model as model_mod, policies, runner as R
from robosumo_selfplay_amd.vec_env import SumoVecEnv
N, T = 4096, 128
env = SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=N, seed=1000)
spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
ms = [model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=(i == 0)) for i in range(2)]
r = R.Runner(env=env, models=ms, nsteps=T, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
for _ in range(2): r.run(1)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter(); out = r.run(1); torch.cuda.synchronize(); print("run() total %.1f ms, %d epinfos" % ((time.perf_counter() - t0) * 1e3, len(out[11])))
# sections
sync = torch.cuda.synchronize
t0 = time.perf_counter(); B = r._alloc_device(T); sync(); t1 = time.perf_counter()
r._steps_fused(B, 0, T, 1.0); sync(); t2 = time.perf_counter()
d = B["ep_done"].cpu().numpy().astype(bool); rr, ll = B["ep_r"].cpu().numpy(), B["ep_l"].cpu().numpy(); t3 = time.perf_counter()
ep = [{"r": round(float(rr[s, e]), 6), "l": int(ll[s, e]), "t": 0.0} for s, e in zip(*np.nonzero(d))]; t4 = time.perf_counter()
x = R.sf01(B["obs"]); y = R.sf01(B["act"]); sync(); t5 = time.perf_counter()
print("alloc %.1f ms | fused launch (incl. noise) %.1f ms | D2H episode records %.1f ms | epinfo dicts (%d) %.1f ms | sf01 views %.2f ms"
      % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(ep), (t4 - t3) * 1e3, (t5 - t4) * 1e3))
pr = cProfile.Profile(); pr.enable(); r.run(1); sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
