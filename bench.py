#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: vectorised RoboSumo env steps per second.

A "step" is ONE pass of the hot path over one batch: one rollout step of the PPO2 self-play Runner -- the 5 policy /
value evaluations of reference runner.py:66-97 (4 fused MFMA launches), one vectorised env step (5 x RK4 mj_step, game
rules, rewards, done, auto-reset, observation write) and the reward curriculum -- for `--envs` environments per GPU
(BASELINE configs[1]: RoboSumo Ant-vs-Ant, 4096 envs, MLP(64,64), 1 MI355X).  Inputs are synthetic and resident in HBM:
random-init networks of the reference architecture (zero-init logstd => N(0,1)-scale actions, the initial policy's law)
and env states warmed up from the env's own reset distribution.  After the timed region one full PPO2
update (rollout of --ppo-nsteps + noptepochs x nminibatches optimiser steps) is timed for the "+ PPO2 iters/sec" half
of the metric and reported under config.  Multi-GPU: one process per GPU (torch.distributed / RCCL), envs sharded with
no data-path collective -> weak scaling; value = total env-steps of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from robosumo_selfplay_amd import hostcfg  # noqa: E402  (first: caps the host thread pools at the cgroup CPU quota)

hostcfg.apply()
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_env_step(m):
    """SURVEY.md §8(d) with float64 device state: read qpos,qvel,warm + actions; write qpos,qvel,warm + obs (2 agents)
    + reward components + done + counters."""
    nq, nv, nu = m.nq, m.nv, m.nu
    obs = sum(m.obs_dims)
    return 8 * 2 * (nq + 2 * nv) + 4 * nu + 4 * obs + 8 * 16 + 2 + 8 + 2 * 8 + 4


F64_VALU_PEAK_TF = 78.6   # MI355X vector f64 peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz), MI355X_MICROARCH.md
F32_MFMA_PEAK_TF = 157.3  # dense f32 matrix-core peak (v_mfma_f32_*_f32), same guide


def mfma_probe(torch, dev, learner, opponent, env, obs_b, ret_b, act_b, val_b, nlp_b, nmb_rows, reps=20, use_graph=True):
    """MFMA side of the roofline, measured live with HIP events on the launching stream: `reps` back-to-back calls of ppo_grad on one
    PPO2 minibatch (forward + backward + weight-gradient MFMAs of both nets; the call also runs the slab-reduction kernel,
    which are counted in the time but not in the flops) and of ppo_selfplay_forward on one env group (the 5 evaluations of a
    rollout step)."""
    import ctypes as C
    from robosumo_selfplay_amd import ppo_capi
    L = ppo_capi.lib()
    D, A = learner.spec.ob_dim, learner.spec.ac_dim
    S = {"st": torch.cuda.current_stream(dev).cuda_stream}
    n = int(nmb_rows)
    idx = torch.randperm(obs_b.shape[0], device=dev)[:n].to(torch.int32)
    adv = torch.randn(n, device=dev)
    w = torch.ones(obs_b.shape[0], device=dev)
    lr = torch.empty(n, device=dev)
    stats = torch.zeros(ppo_capi.NSTATS, dtype=torch.float64, device=dev)
    grads = torch.zeros_like(learner.grads)
    call_g = lambda: ppo_capi.chk(L.ppo_grad(learner.params.data_ptr(), obs_b.data_ptr(), obs_b.stride(0), D, A, act_b.data_ptr(), adv.data_ptr(),
                                             ret_b.data_ptr(), nlp_b.data_ptr(), w.data_ptr(), idx.data_ptr(), n, 1.0 / n, 0.2, 0.0, 0.5,
                                             grads.data_ptr(), stats.data_ptr(), lr.data_ptr(), learner.workspace.data_ptr(), S["st"]))
    ng = env.group_size
    f32 = torch.float32
    outs = [torch.empty((ng, D), dtype=f32, device=dev), torch.empty((ng, D), dtype=f32, device=dev)] + \
           [torch.empty((ng, A), dtype=f32, device=dev) for _ in range(2)] + [torch.empty(ng, dtype=f32, device=dev) for _ in range(6)]
    fp = (C.c_void_p * 10)(*[x.data_ptr() for x in outs])
    dn = [torch.empty(ng, dtype=torch.uint8, device=dev) for _ in range(2)]
    dp = (C.c_void_p * 2)(dn[0].data_ptr(), dn[1].data_ptr())
    noise = [torch.randn((ng, A), device=dev) for _ in range(2)]
    act_env = torch.empty((ng, 2, A), dtype=f32, device=dev)
    ob = env.obs_dev[:ng]
    call_s = lambda: ppo_capi.chk(L.ppo_selfplay_forward(learner.params.data_ptr(), opponent.params.data_ptr(), ob.data_ptr(), ng, ob.stride(0),
                                                         ob.stride(1), D, A, noise[0].data_ptr(), noise[1].data_ptr(), env.done_dev[:ng].data_ptr(),
                                                         act_env.data_ptr(), fp, dp, S["st"]))
    res = {}
    pi = D * 64 + 64 * 64 + 64 * A
    vf = D * 64 + 64 * 64 + 64
    for name, call, flops in (("ppo_grad_kernel", call_g, 3.0 * 2.0 * (pi + vf) * n),                    # fwd + dX + dW, SURVEY.md 8(d)
                              ("ppo_selfplay_kernel", call_s, 2.0 * (2 * (2 * pi + vf)) * ng)):         # both sides: 2 policy trunks + value
        for _ in range(3):
            call()
        torch.cuda.synchronize(dev)
        # the calls are replayed from a HIP graph so that the GPU, not the Python enqueue loop, sets the pace
        # (use_graph=False: plain launches, for counter collection under rocprofv3 -- tools/prof_workload.py)
        if use_graph:
            eager = S["st"]
            graph = torch.cuda.CUDAGraph()
            with hostcfg.gc_paused(), torch.cuda.graph(graph):
                S["st"] = torch.cuda.current_stream(dev).cuda_stream
                for _ in range(reps):
                    call()
            S["st"] = eager
            replay = graph.replay
        else:
            def replay():
                for _ in range(reps):
                    call()
        replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        replay()
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / reps
        tf = flops / (us * 1e-6) / 1e12
        res[name] = {"achieved": tf, "frac": tf / F32_MFMA_PEAK_TF, "us_per_call": us, "flops_per_call": flops,
                     "rows_per_call": n if name == "ppo_grad_kernel" else ng}
    res["ppo_grad_kernel"]["note"] = "time of the whole ppo_grad call (gradient kernel + the slab-reduction launch); flops of the gradient kernel"
    return res


def spider_segment(torch, dev, local_rank, args, hp):
    """Secondary driver-timed line (BASELINE config 4): RoboSumo-Spider-vs-Spider-v0, same number of envs, same rollout step."""
    from robosumo_selfplay_amd import mjcf
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.policies import build_policy
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    env_id = "RoboSumo-Spider-vs-Spider-v0"
    N = args.envs
    env = SumoVecEnv(env_id, num_envs=N, seed=77, device=local_rank, model=mjcf.load_model(env_id), groups=args.groups)
    spec = build_policy(env, "mlp", value_network=hp["value_network"], num_hidden=hp["num_hidden"], activation=hp["activation"])
    ms = [PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False, model_scope="model_%d" % i, device=local_rank)
          for i in range(2)]
    ms[1].set_param_list(ms[0].get_param_list())
    r = Runner(env=env, models=ms, nsteps=8, nagent=2, gamma=hp["gamma"], lam=hp["lam"], rho_bar=hp["rho_bar"], c_bar=hp["c_bar"])
    r.fused_rollout = not args.stepwise
    fused = r.fused_ok()
    K = args.spider_steps
    B = r._alloc_device(max(K, 8) if fused else 8)
    T = B["T"]

    def advance(n):
        if fused:
            for s0 in range(0, n, T):
                r._steps_fused(B, 0, min(T, n - s0), 1.0)
        else:
            for k in range(n):
                r._step_device(B, k % 8, 1.0)
    advance(args.state_warmup + 5)
    r.join_groups()
    torch.cuda.synchronize(dev)
    st0 = env.stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    advance(K)
    r.join_groups()
    e1.record()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    check_rollout(env, fused)
    st1 = env.stats()
    nf = max(1.0, st1["forward"] - st0["forward"])
    out = {"env_id": env_id, "envs": N, "steps": args.spider_steps, "env_steps_per_s": N * args.spider_steps / dt,
           "ms_per_step": dt / args.spider_steps * 1e3, "gpu_ms": e0.elapsed_time(e1), "rollout_path": "fused" if fused else "stepwise",
           "envs_per_launch": env.group_size, "lds_bytes_per_env": env.engine.lds_bytes,
           "waves_per_cu": int(160 * 1024 // env.engine.lds_bytes), "mean_contacts_per_forward": (st1["contacts"] - st0["contacts"]) / nf,
           "mean_newton_iters_per_forward": (st1["newton"] - st0["newton"]) / nf,
           "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(env.model)}
    if fused:
        out.update(profile_rooflines("r03_spider", "sumo_rollout_kernel<%d, 0" % env.model.nv, N, env.model, out["env_steps_per_s"],
                                     e0.elapsed_time(e1), N * K))
    env.close()
    return out


def recurrent_segment(torch, dev, local_rank, args, hp):
    """Secondary driver-timed line (BASELINE config 5's one-GPU shard): Ant-vs-Ant, 1024 envs, LSTM(128) policies against a pool of 16
    frozen snapshots (one per 16-env tile), the whole recurrent rollout step in the fused launch (sumo_rollout_steps_lstm)."""
    from robosumo_selfplay_amd import lstm_model, mjcf
    from robosumo_selfplay_amd.opponent_pool import LstmOpponentPool
    from robosumo_selfplay_amd.runner import Runner
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    env_id, N, K, P = "RoboSumo-Ant-vs-Ant-v0", 1024, args.recurrent_steps, 16
    env = SumoVecEnv(env_id, num_envs=N, seed=78, device=local_rank, model=mjcf.load_model(env_id), groups=1)
    spec = lstm_model.LstmSpec(env.observation_space[0].shape[0], env.action_space[0].shape[0], 128)
    learner = lstm_model.LstmPPOModel(policy=spec, nbatch_act=N, nsteps=K, trainable=False, device=local_rank)
    pool = LstmOpponentPool(spec, P, N, env.device)
    for k in range(P):
        pool.set_snapshot(k, learner.get_param_list())
    pool.assign_round_robin()
    r = Runner(env=env, models=[learner, pool], nsteps=K, nagent=2, gamma=hp["gamma"], lam=hp["lam"], rho_bar=hp["rho_bar"], c_bar=hp["c_bar"])
    r.fused_rollout = not args.stepwise
    fused = r.fused_lstm_ok()
    B = r._alloc_device(K)

    def advance(n):
        for s0 in range(0, n, K):
            if fused:
                r._steps_fused_lstm(B, 0, min(K, n - s0), 1.0)
            else:
                for k in range(min(K, n - s0)):
                    r._step_device(B, k, 1.0)
    advance(args.state_warmup)
    r.join_groups()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    advance(K)
    r.join_groups()
    e1.record()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    check_rollout(env, fused)
    out = {"env_id": env_id, "envs": N, "steps": K, "policy": "lstm(128), shared value head", "opponent_pool": P, "gpu_ms": e0.elapsed_time(e1),
           "env_steps_per_s": N * K / dt, "ms_per_step": dt / K * 1e3, "rollout_path": "fused" if fused else "stepwise",
           "note": "1024 envs leave half of the chip's 2048 wave slots empty: the launch lasts as long as its slowest env's chain of steps"}
    if fused:
        rf = profile_rooflines("r03_rec1024", "sumo_rollout_kernel<%d, 1" % env.model.nv, N, env.model, out["env_steps_per_s"], out["gpu_ms"], N * K)
        rf["roofline"]["algorithmic_bytes_per_env_step_note"] = "env record only; the recurrent states add 2 x 2 x 1 KB per env step"
        out.update(rf)
    env.close()
    return out


def profile_rooflines(prefix, kernel_sub, envs, model, env_steps_per_s, launch_ms, steps_per_launch):
    """HBM / ALU roofline objects of a secondary config from its rocprofv3 summaries profiles/<prefix>_pmc_{traffic,sq}.json
    (tools/profile_r03.sh; quoted only if the file is for this kernel variant and env count) and the live launch time."""
    B = algorithmic_bytes_per_env_step(model)
    out = {}

    def load(name):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", name)))
            return pj if kernel_sub in pj["kernel"] and pj["envs"] == envs else None
        except Exception:
            return None
    tr, sq = load(prefix + "_pmc_traffic.json"), load(prefix + "_pmc_sq.json")
    ach = B * steps_per_launch / (launch_ms * 1e-3) / 1e9
    out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "traffic": None if tr is None else tr["traffic_bytes_per_env_step"] * steps_per_launch,
                       "traffic_source": None if tr is None else "profiles/%s_pmc_traffic.json" % prefix, "kernel": kernel_sub,
                       "kernel_ms": launch_ms, "algorithmic_bytes_per_env_step": B, "env_steps_per_launch": steps_per_launch}
    if sq is not None:
        out["roofline_alu"] = alu_roofline(sq["derived"], env_steps_per_s, kernel_sub, "profiles/%s_pmc_sq.json" % prefix)
    return out


def alu_roofline(d, env_steps_per_s, kernel, source):
    """ALU side of a physics kernel from its SQ counter summary: f64 flops as ISSUED over 64 lanes and the part that lands on
    active lanes (x mean active lanes / 64), both scaled by the live env-step rate; VALU issue share of SIMD time with every
    instruction priced at 4 cycles and with f64 arithmetic at 4, everything else at 2 (MI355X_MICROARCH.md)."""
    fl = d["f64_flops_issued_per_env_step"]
    tf = fl * env_steps_per_s / 1e12
    o = {"bound": "valu_f64", "kernel": kernel, "achieved": tf, "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F64_VALU_PEAK_TF,
         "f64_flops_issued_per_env_step": fl, "valu_issue_frac": d["valu_issue_frac"], "valu_issue_frac_priced": d.get("valu_issue_frac_priced"),
         "valu_active_frac_of_wave_cycles": d.get("active_valu_frac"), "waiting_frac_of_wave_cycles": d.get("wait_any_frac"),
         "mean_active_lanes": d.get("mean_active_lanes"), "valu_insts_per_forward": d.get("valu_insts_per_forward_per_wave"),
         "wave_cycles_per_forward": d.get("wave_cycles_per_forward"), "source": source,
         "note": "flops = (ADD+MUL+TRANS + 2 FMA) f64 wave-instructions x 64 lanes per env step from the profile, times the live rate"}
    if d.get("f64_flops_on_active_lanes_per_env_step") is not None:
        tfa = d["f64_flops_on_active_lanes_per_env_step"] * env_steps_per_s / 1e12
        o["achieved_on_active_lanes"] = tfa
        o["frac_on_active_lanes"] = tfa / F64_VALU_PEAK_TF
    return o


def check_rollout(env, fused):
    """A fused launch that was cut short would make the timed number meaningless (and leave garbage rows): raise, outside the
    timed region (capi.Engine.rollout_status -> sumo_rollout_status)."""
    if fused:
        for E in env.engines:
            E.rollout_status()


def cpu_baseline(model, states, actions, steps, threads):
    """Times the CPU oracle (the float64 restatement; the reference's MuJoCo path cannot run here) on a bounded
    sample of the same workload: the first len(states[0]) envs of the GPU batch, same states, same action law."""
    from oracle.oracle import OracleSim, build
    build()
    n = states[0].shape[0]
    sim = OracleSim(model, n)
    sim.set_state(*states)
    sim.set_seeds(np.arange(n, dtype=np.uint64))
    sim.step(actions[0], nthreads=threads)
    t0 = time.perf_counter()
    for k in range(steps):
        sim.step(actions[k % len(actions)], nthreads=threads)
    dt = time.perf_counter() - t0
    return n * steps / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--env-id", default="RoboSumo-Ant-vs-Ant-v0")
    ap.add_argument("--groups", type=int, default=0, help="env groups per GPU (0 = 1 with the fused rollout launch, 2 with the step-by-step launches: "
                    "each group is stepped on its own stream there)")
    ap.add_argument("--stepwise", action="store_true", help="time the step-by-step launches (ppo_selfplay_forward + sumo_step + ppo_post_step per "
                    "rollout step) instead of the fused rollout launch (sumo_rollout_steps: all timed steps in one launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-envs", type=int, default=1024)
    ap.add_argument("--cpu-sample-steps", type=int, default=300)
    ap.add_argument("--ppo-nsteps", type=int, default=128, help="rollout length of the PPO2 update timed after the main region (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the cgroup CPU quota of this job (hostcfg.cpu_quota)")
    ap.add_argument("--no-mfma-probe", action="store_true")
    ap.add_argument("--spider-steps", type=int, default=30, help="timed rollout steps of the secondary Spider-vs-Spider 4096-env line (0 = skip)")
    ap.add_argument("--recurrent-steps", type=int, default=64, help="timed rollout steps of the secondary recurrent line (config-5 shard: 1024 envs, LSTM(128), pool of 16; 0 = skip)")
    ap.add_argument("--state-warmup", type=int, default=100,
                    help="untimed rollout steps that bring the env states from the reset law to the steady workload (SURVEY.md 8(d): >= 100)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU through torch.distributed.run, started BEFORE this process touches a GPU (a process that
        # has initialised HIP must never exec; children are ordinary subprocesses).  Rank 0's JSON line passes through.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    from robosumo_selfplay_amd import mjcf
    from robosumo_selfplay_amd.vec_env import SumoVecEnv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND") or os.environ.get("SUMO_DIST_BACKEND") or "nccl"   # "nccl" is RCCL; "gloo" only to rehearse N>1 on a 1-GPU box
        ndev = torch.cuda.device_count()
        local_rank = local_rank % max(1, ndev)
        torch.cuda.set_device(local_rank)
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model = mjcf.load_model(args.env_id)
    N = args.envs
    if args.groups <= 0:
        args.groups = 2 if args.stepwise else 1
    env = SumoVecEnv(args.env_id, num_envs=N, seed=1000 + rank * N, device=local_rank, model=model, groups=args.groups)
    from robosumo_selfplay_amd import defaults
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.policies import build_policy
    from robosumo_selfplay_amd.runner import Runner, anneal_alpha
    np.random.seed(0)                                    # identical random-init weights on every rank
    hp = defaults.get_default_params(args.env_id)
    spec = build_policy(env, "mlp", value_network=hp["value_network"], num_hidden=hp["num_hidden"], activation=hp["activation"])
    group = dist.group.WORLD if dist is not None else None
    learner = PPOModel(policy=spec, ent_coef=hp["ent_coef"], vf_coef=0.5, max_grad_norm=0.5, model_scope="model_0",
                       device=local_rank, comm=group)
    opponent = PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5, trainable=False, model_scope="model_1",
                        device=local_rank)
    opponent.set_param_list(learner.get_param_list())
    learner.act_model.seed(1234 + rank)
    opponent.act_model.seed(4321 + rank)
    runner = Runner(env=env, models=[learner, opponent], nsteps=args.ppo_nsteps, nagent=2, gamma=hp["gamma"], lam=hp["lam"],
                    rho_bar=hp["rho_bar"], c_bar=hp["c_bar"], anneal_bound=hp["anneal_bound"])
    alpha = anneal_alpha(1, hp["anneal_bound"])

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    runner.fused_rollout = not args.stepwise
    fused = runner.fused_ok()
    ring = max(8, args.steps, args.warmup) if fused else 8
    B = runner._alloc_device(ring)

    def advance(k0, n, events=None):
        """n rollout steps starting at running step index k0: fused = one launch per pass over the ring (the timed region of K steps
        is ONE launch), else one launch chain per step."""
        if fused:
            done = 0
            while done < n:
                s0 = (k0 + done) % ring
                c = min(n - done, ring - s0)
                if events is not None:
                    events[0].record()
                runner._steps_fused(B, s0, c, alpha)
                if events is not None:
                    events[1].record()
                done += c
        else:
            for k in range(n):
                runner._step_device(B, (k0 + k) % ring, alpha, env_events=(ev0[k], ev1[k]) if events is not None and ev_every and k % ev_every == 0 else None)
    # workload priming (not a timing knob): the reset law drops every agent from z = 1.25 at the same instant; >= 100 rollout
    # steps under the initial policy's N(0,1)-scale actions spread the episodes out and make contacts active
    advance(0, args.state_warmup)
    runner.join_groups()
    torch.cuda.synchronize(dev)
    k0 = ((args.state_warmup + ring - 1) // ring) * ring       # fused: warm-up and the timed region each start a pass over the ring
    advance(k0, args.warmup)                                   # timing warm-up
    runner.join_groups()
    torch.cuda.synchronize(dev)
    st0 = env.stats()
    states = None
    if rank == 0 and not args.no_cpu_baseline:
        q, v, w, c = env.engine.get_state()
        ns = min(args.cpu_sample_envs, env.group_size)
        states = (q[:ns].copy(), v[:ns].copy(), w[:ns].copy(), c[:ns].copy())

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_every = int(os.environ.get("BENCH_EVENT_EVERY", "1"))       # HIP events around the env launch of every n-th step (0 = none)
    timed = [k for k in range(args.steps) if ev_every and k % ev_every == 0]
    k0 += ((args.warmup + ring - 1) // ring) * ring
    barrier()
    thr0 = hostcfg.throttle_stats()
    t0 = time.perf_counter()
    advance(k0, args.steps, events=(ev0[0], ev1[0]))
    runner.join_groups()
    barrier()
    dt = time.perf_counter() - t0
    thr1 = hostcfg.throttle_stats()
    check_rollout(env, fused)
    if fused:
        timed = [0]
    kern_ms = float(np.mean([ev0[k].elapsed_time(ev1[k]) for k in timed])) if timed else float("nan")   # fused: the K-step launch
    st1 = env.stats()
    sample_acts = [B["act"][:, k].permute(1, 0, 2).contiguous() for k in range(8)]   # [N, 2, A] per step

    # ---- "+ PPO2 iters/sec": one full update, outside the timed region above
    ppo = None
    if args.ppo_nsteps > 0:
        T, nmb, nep = args.ppo_nsteps, hp["nminibatches"], hp["noptepochs"]
        for timed_update in (False, True):     # the first update pays one-time costs (graph capture, lazy initialisation): untimed
            barrier()
            tu = time.perf_counter()
            out = runner.run(1)
            torch.cuda.synchronize(dev)
            t_roll = time.perf_counter() - tu
            obs_b, ret_b, act_b, val_b, nlp_b = out[0][0].contiguous(), out[1][0], out[3][0], out[4][0], out[5][0]
            nb = N * T
            wts = torch.ones(nb, dtype=torch.float32, device=dev)
            learner.begin_update(obs_b, ret_b, act_b, val_b, nlp_b, wts)
            for ep in range(nep):
                inds = torch.randperm(nb, device=dev).to(torch.int32)      # device-side shuffle (np.random.shuffle in alg_ppo.py:375)
                learner.prepare_epoch(inds, nb // nmb)     # multi-GPU: the epoch's advantage moments in one all-reduce (no-op on one GPU)
                for k, start in enumerate(range(0, nb, nb // nmb)):
                    mb = inds[start:start + nb // nmb]
                    learner.train_indexed(hp["lr"], hp["cliprange"], obs_b, ret_b, act_b, val_b, nlp_b, wts, mb, int(mb.numel()), sync=False,
                                          mb_index=k)
            learner.end_update()
            barrier()
            t_upd = time.perf_counter() - tu
            tt = torch.tensor([t_upd, t_roll], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ppo = {"iters_per_sec": 1.0 / float(tt[0].item()), "nsteps": T, "nminibatches": nmb, "noptepochs": nep,
                   "rollout_s": float(tt[1].item()), "sgd_s": float(tt[0].item() - tt[1].item()),
                   "samples_per_iter": nb * world}

    mfma = None
    if args.ppo_nsteps > 0 and rank == 0 and not args.no_mfma_probe:
        mfma = mfma_probe(torch, dev, learner, opponent, env, obs_b, ret_b, act_b, val_b, nlp_b, nb // nmb)
    spider = None
    if args.spider_steps > 0 and rank == 0 and "Ant-vs-Ant" in args.env_id:
        spider = spider_segment(torch, dev, local_rank, args, hp)
    recurrent = None
    if args.recurrent_steps > 0 and rank == 0 and "Ant-vs-Ant" in args.env_id:
        recurrent = recurrent_segment(torch, dev, local_rank, args, hp)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    if rank == 0:
        total_steps = N * world * args.steps
        value = total_steps / dt_max
        B = algorithmic_bytes_per_env_step(model)
        # PMC figures are collected offline with rocprofv3 (profiles/README.md; separate --pmc passes as MI355X_MICROARCH.md
        # prescribes) and are only quoted for the configuration they were profiled on; every one names its file
        kernel_name = "sumo_rollout_kernel" if fused else "sumo_step_kernel"
        steps_per_launch = env.group_size * (args.steps if fused else 1)          # env steps one launch advances

        def profile_json(name):
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", name)))
                kn = pj["kernel"]          # "sumo_step_kernel<28>" / "sumo_rollout_kernel<28, 0>" (nv, policy variant: 0 = the MLP one timed here)
                ok = (("<%d>" % model.nv) in kn or ("<%d, 0" % model.nv) in kn) and kernel_name in kn and pj["envs"] == env.group_size
                return pj if ok else None
            except Exception:
                return None
        traffic, traffic_src = None, None
        for cand in ("r03_ant_pmc_traffic.json", "r02c_pmc_traffic.json", "r02b_pmc_traffic.json", "r02_pmc_traffic.json", "r01f_pmc_traffic.json"):
            pj = profile_json(cand)
            if pj is not None:          # per env step in the profile, scaled to this launch's env steps
                traffic, traffic_src = pj["traffic_bytes_per_env_step"] * steps_per_launch, "profiles/" + cand
                break
        sq, sq_src = None, None
        for cand in ("r03_ant_pmc_sq.json", "r02c_pmc_sq.json", "r02b_pmc_sq.json", "r02_pmc_sq.json"):
            sq = profile_json(cand)
            if sq is not None:
                sq_src = "profiles/" + cand
                break
        achieved = B * steps_per_launch / (kern_ms * 1e-3) / 1e9      # per launch (HIP events on its stream)
        nfwd = max(1.0, st1["forward"] - st0["forward"])
        out = {
            "metric": "env-steps/sec (whole node), RoboSumoAnts-v0 4096 envs, + PPO2 iters/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d envs per GPU, MLP(64,64) policy+value: one self-play rollout step per bench step "
                                   "(5 policy/value evaluations + env step of frame_skip 5 x RK4 + reward mix), random-init "
                                   "networks, auto-reset on" % (args.env_id, N),
                       "rollout_path": "fused: the timed steps are one sumo_rollout_steps launch per env group" if fused else
                                       "stepwise: ppo_selfplay_forward + sumo_step + ppo_post_step launches per step",
                       "ppo2": ppo,
                       "state_warmup_steps": args.state_warmup,
                       "envs_per_gpu": N, "env_groups_per_gpu": env.groups, "total_envs": N * world, "parallelism": "env-shard x%d" % world,
                       "mean_contacts_per_forward": (st1["contacts"] - st0["contacts"]) / nfwd,
                       "mean_newton_iters_per_forward": (st1["newton"] - st0["newton"]) / nfwd,
                       "lds_bytes_per_env": env.engine.lds_bytes,
                       # contact-generation fidelity accounting (DESIGN.md deviations 1-2), sampled in the forward evaluation that opens
                       # each env step: capsule-box calls with 3 active contacts (MuJoCo: <= 2) and contacts on a border rod beyond the
                       # cylinder's flat end (rods collide as capsules), per million sampled forwards; contacts dropped for lack of room
                       "contact_fidelity": {"sampled_forwards": total_steps / max(1, world),
                                            "capsule_box_3_per_million": 1e6 * (st1.get("capsule_box_3", 0) - st0.get("capsule_box_3", 0)) / max(1.0, total_steps / max(1, world)),
                                            "rod_endcap_per_million": 1e6 * (st1.get("rod_endcap", 0) - st0.get("rod_endcap", 0)) / max(1.0, total_steps / max(1, world)),
                                            "dropped_contacts": st1["dropped"] - st0["dropped"]}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": B,
                         "envs_per_launch": env.group_size, "env_steps_per_launch": steps_per_launch, "concurrent_launches": env.groups,
                         "aggregate_achieved": B * value / max(1, world) / 1e9,
                         "note": "latency/ALU-bound physics: ~20 forward-dynamics solves per 2.4 KB of state traffic"},
        }
        if sq is not None:
            out["roofline_alu"] = alu_roofline(sq["derived"], value / max(1, world), kernel_name, sq_src)
        if mfma is not None:
            out["roofline_mfma"] = dict(mfma, bound="mfma", peak=F32_MFMA_PEAK_TF, unit="TFLOP/s", dtype="f32 (v_mfma_f32_16x16x4_f32)")
            # the same kernels by rocprofv3 counters (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32; tools/profile_r03.sh mfma)
            for kn, fn in (("ppo_grad_kernel", "r03_pmc_mfma_grad.json"), ("ppo_selfplay_kernel", "r03_pmc_mfma_selfplay.json")):
                try:
                    pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
                    dd = pj["derived"]
                    out["roofline_mfma"][kn]["counters"] = {
                        "mfma_busy_frac_of_simd_time": dd.get("mfma_busy_frac_of_simd_time"), "mfma_tflops": dd.get("mfma_tflops_at_2p4ghz"),
                        "frac_of_peak": dd.get("frac_of_f32_mfma_peak_157p3"), "kernel_us": dd["kernel_cycles"] / 2.4e3,
                        "counted_over_algorithmic_flops": dd.get("counted_over_algorithmic"), "source": "profiles/" + fn}
                except Exception:
                    pass
        if spider is not None:
            out["config"]["spider"] = spider
        if recurrent is not None:
            out["config"]["recurrent"] = recurrent
        # static scan of the code objects this run loaded for the register-allocator defect of DESIGN.md section 4 ("code-generation
        # hazard"): suspicious copies per library (0 = clean), None if the tools are missing
        try:
            from robosumo_selfplay_amd import build as _build, codegen_check as _cc
            out["config"]["codegen_check"] = {name: sum(len(v) for v in _cc.scan_library(_build.lib_path(name)).values())
                                              for name in ("libsumo_hip.so", "libsumo_ppo.so")}
        except Exception:
            out["config"]["codegen_check"] = None
        out["host"] = {"cpu_model": hostcfg.cpu_model(), "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": hostcfg.cpu_quota(),
                       "pool_threads": hostcfg.apply(),
                       "throttled_periods_in_timed_region": None if thr0 is None or thr1 is None else thr1[0] - thr0[0]}
        if states is not None:
            # the reference's CPU path (mujoco-py + SubprocVecEnv) cannot run offline; the C restatement is timed in the three shapes
            # BASELINE.md 3 lists: B2 OpenMP over envs on every core of this job's quota (the headline figure), B1 one thread,
            # B3 eight single-env worker processes behind pipes like subproc_vec_env.py:6-32,65-76 (config 1's shape)
            threads = args.cpu_threads or hostcfg.cpu_quota()
            cpu_acts = [a[:states[0].shape[0]].cpu().numpy() for a in sample_acts]
            v = cpu_baseline(model, states, cpu_acts, args.cpu_sample_steps, threads)
            out["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": threads, "kind": "port",
                                   "sample": "%d envs x %d steps of the same warmed-up workload, OpenMP over envs"
                                             % (states[0].shape[0], args.cpu_sample_steps)}
            n1 = min(128, states[0].shape[0])
            s1 = tuple(x[:n1] for x in states)
            v1 = cpu_baseline(model, s1, [a[:n1] for a in cpu_acts], max(10, args.cpu_sample_steps // 3), 1)
            out["cpu_baseline"]["single_thread"] = {"value": v1, "cores": 1, "sample": "%d envs x %d steps, one thread"
                                                    % (n1, max(10, args.cpu_sample_steps // 3))}
            try:
                from oracle import subproc_baseline
                v3 = subproc_baseline.run(args.env_id, 8, 1500, model.act_dims[0])
                out["cpu_baseline"]["subproc8"] = {"value": v3, "cores": 8, "sample": "8 single-env worker processes x 1500 lock-step steps over pipes, the five policy / "
                                                   "value passes of every step evaluated in numpy on the parent (random-init MLP(64,64) nets, from the reset law)"}
            except Exception as e:                                    # the harness is a report, never a reason to lose the bench line
                out["cpu_baseline"]["subproc8"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
