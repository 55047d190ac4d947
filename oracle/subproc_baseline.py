"""CPU-baseline variant B3: the oracle's env step under a harness shaped like the reference's
``SubprocVecEnv`` (subproc_vec_env.py:6-32 worker loop, :65-76 step_async / step_wait): one OS process per env, commands and
results pickled over a pipe, the parent stacks the results.  TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg).
"""
import multiprocessing as mp
import os
import time

import numpy as np


def _worker(remote, parent_remote, env_id, seed):
    parent_remote.close()
    os.environ["OMP_NUM_THREADS"] = "1"
    from robosumo_selfplay_amd import mjcf
    from oracle.oracle import OracleSim
    sim = OracleSim(mjcf.load_model(env_id), 1)
    sim.reset(seeds=[seed])
    try:
        while True:
            cmd, data = remote.recv()
            if cmd == "step":                                   # auto-reset happens inside the oracle's env step, like the worker's
                obs, info, done, ep_r, ep_dr, ep_l = sim.step(data[None], nthreads=1)
                rew = info[0, :, 3] + info[0, :, 6]
                infos = tuple({"shaping_reward": float(info[0, a, 6]), "main_reward": float(info[0, a, 3])} for a in range(2))
                remote.send((obs[0], rew, done[0].astype(bool), infos))
            elif cmd == "reset":
                remote.send(sim.reset()[0])
            elif cmd == "close":
                remote.close()
                break
    except (KeyboardInterrupt, EOFError):
        pass


def run(env_id, nenvs, steps, act_dim, seed=0, policy=True):
    """env-steps/s of ``nenvs`` single-env worker processes driven in lock step for ``steps`` steps.  ``policy``: the parent
    evaluates the rollout's five policy / value passes of every step in numpy (runner.py:66-97: learner.step on obs 0, the
    opponent's likelihood of that action, opponent.step on obs 1, learner.value and learner's likelihood on obs 1) with
    random-init MLP(64,64) nets (oracle/ppo_oracle.py), like the reference's parent process does through TensorFlow; False: the
    actions are pre-drawn N(0,1) (the env + IPC cost alone)."""
    ctx = mp.get_context("spawn")
    remotes, work_remotes = zip(*[ctx.Pipe() for _ in range(nenvs)])
    ps = [ctx.Process(target=_worker, args=(w, r, env_id, seed + i), daemon=True) for i, (w, r) in enumerate(zip(work_remotes, remotes))]
    for p in ps:
        p.start()
    for w in work_remotes:
        w.close()
    for r in remotes:
        r.send(("reset", None))
    first = np.stack([r.recv() for r in remotes])
    rng = np.random.default_rng(seed)
    acts = rng.standard_normal((16, nenvs, 2, act_dim)).astype(np.float32)
    state = {"obs": first}
    if policy:
        from oracle import ppo_oracle as po
        ob_dim = first.shape[-1]
        prng = np.random.RandomState(seed)
        nets = [po.init_params(prng, ob_dim, act_dim), po.init_params(prng, ob_dim, act_dim)]        # learner, opponent

        def act(obs, s):
            o0, o1 = obs[:, 0, :ob_dim].astype(np.float32), obs[:, 1, :ob_dim].astype(np.float32)
            m0, v0, _ = po.forward(nets[0], o0, np.float32)                                            # learner.step(obs 0)
            a0 = po.sample(m0, nets[0][10], acts[s % 16][:, 0]).astype(np.float32)
            po.neglogp(m0, nets[0][10], a0)
            mo, _, _ = po.forward(nets[1], o0, np.float32); po.neglogp(mo, nets[1][10], a0)           # opponent scores it
            m1, _, _ = po.forward(nets[1], o1, np.float32)                                             # opponent.step(obs 1)
            a1 = po.sample(m1, nets[1][10], acts[s % 16][:, 1]).astype(np.float32)
            ml, vl, _ = po.forward(nets[0], o1, np.float32); po.neglogp(ml, nets[0][10], a1)           # learner's value + likelihood there
            return np.stack([a0, a1], axis=1)

    def loop(k):
        for s in range(k):
            a_all = act(state["obs"], s) if policy else acts[s % 16]
            for r, a in zip(remotes, a_all):
                r.send(("step", a))
            res = [r.recv() for r in remotes]
            obs, rews, dones, infos = zip(*res)
            state["obs"] = np.stack(obs); np.stack(rews); np.stack(dones)
    loop(20)
    t0 = time.perf_counter()
    loop(steps)
    dt = time.perf_counter() - t0
    for r in remotes:
        r.send(("close", None))
    for p in ps:
        p.join(timeout=10)
    return nenvs * steps / dt
