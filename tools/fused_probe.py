#!/usr/bin/env python3
"""Development: fused rollout launch (sumo_rollout_steps) vs the step-by-step path -- env-steps/s for K steps per launch, and where
a wave of the fused launch spends its time (policy phases / env steps, from the per-wave phase clock)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import model as model_mod, policies
from robosumo_selfplay_amd.runner import Runner
from robosumo_selfplay_amd.vec_env import SumoVecEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--env-id", default="RoboSumo-Ant-vs-Ant-v0")
ap.add_argument("--configs", default="2:0:0,1:1:0,1:1:0")
ap.add_argument("--probe", action="store_true", help="library built with -DSUMO_POLICY_PROBE (SUMO_HIP_LIB): split of the MLP policy phase")
ap.add_argument("--lstm", type=int, default=0, help="recurrent policies (LSTM(128)); value = opponent pool size (1 = a single opponent model)")
a = ap.parse_args()
for cfg in a.configs.split(","):
    groups, fused, chunk = (int(x) for x in cfg.split(":"))
    env = SumoVecEnv(a.env_id, num_envs=a.envs, seed=1000, groups=groups)
    spec = policies.PolicySpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], value_network="copy", activation="relu")
    if a.lstm:
        from robosumo_selfplay_amd import lstm_model
        from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
        lspec = lstm_model.LstmSpec(spec.ob_dim, spec.ac_dim, 128)
        ms = [lstm_model.LstmPPOModel(policy=lspec, nbatch_act=a.envs, nsteps=a.steps, trainable=False) for _ in range(2)]
        if a.lstm > 1:
            pool = LstmOpponentPool(lspec, a.lstm, a.envs, env.device)
            for k in range(a.lstm):
                pool.set_snapshot(k, ms[1].get_param_list())
            pool.assign_round_robin()
            ms[1] = pool
    else:
        ms = [model_mod.PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False) for _ in range(2)]
    r = Runner(env=env, models=ms, nsteps=a.steps, nagent=2, gamma=0.995, lam=0.95, rho_bar=1.0, c_bar=1.0)
    B = r._alloc_device(a.steps)
    def run():
        if fused:
            c = chunk or a.steps
            for s0 in range(0, a.steps, c):
                (r._steps_fused_lstm if a.lstm else r._steps_fused)(B, s0, min(c, a.steps - s0), 1.0)
        else:
            for s in range(a.steps):
                r._step_device(B, s, 1.0)
        r.join_groups()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("groups %d fused %d chunk %d: %.3f ms/step, %.0f env-steps/s" % (groups, fused, chunk or a.steps, dt / a.steps * 1e3, a.envs * a.steps / dt), flush=True)
    if fused and groups == 1:
        stamps = torch.zeros((a.envs, 12 if a.probe else 4), dtype=torch.int64, device=env.device)
        env.engine.debug_trace(stamps.data_ptr())
        run(); torch.cuda.synchronize()   # (one launch: chunk = steps)
        env.engine.debug_trace(None)
        st = stamps.cpu().numpy().astype(np.int64)
        pol = st[:, 2] / 1e5; envt = st[:, 3] / 1e5
        span = (st[:, 1].max() - st[:, 0].min()) / 1e5
        print("  per env step: policy phase %.1f us, env step %.1f us (max env total %.2f ms, mean %.2f ms) | launch span %.2f ms, slot-time busy %.1f %% "
              "of 2048 slots | aborts %d" % (pol.mean() / a.steps * 1e3, envt.mean() / a.steps * 1e3, (pol + envt).max(), (pol + envt).mean(), span,
                                             100.0 * (pol + envt).sum() / (span * 2048), env.stats()["rollout_aborts"]), flush=True)
        if a.probe:
            import ctypes as C
            from robosumo_selfplay_amd import capi
            L = capi.lib()
            buf = (C.c_double * 8)()
            L.sumo_debug_tprobe(buf, 1)
            run(); torch.cuda.synchronize()
            L.sumo_debug_tprobe(buf, 1)
            per = [buf[k] / 1e5 / (a.envs * a.steps * 3) * 1e3 for k in range(5)]      # us per trunk
            print("  inside a trunk (us, mean of the three): layer-1 loads + products %.2f | bias / relu / LDS write %.2f | layer-2 %.2f | "
                  "bias / relu / LDS write %.2f | head %.2f" % tuple(per), flush=True)
            print("  policy phase split (us per env step): obs staging %.1f | learner policy trunk %.1f | opponent policy trunk %.1f | learner value trunk %.1f | heads + records %.1f"
                  % tuple(st[:, 4 + q].mean() / 1e5 / a.steps * 1e3 for q in range(5)), flush=True)
    env.close()
