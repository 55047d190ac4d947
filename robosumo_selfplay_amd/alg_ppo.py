"""PPO2 self-play driver: counterpart of the reference's ``alg_ppo.learn`` (alg_ppo.py:25-513), non-recurrent branch.

Same keyword surface and per-update sequence -- opponent selection from the checkpoint directory (the checkpoint dir IS
the opponent pool, alg_ppo.py:217-244), ``runner.run(update)``, IS-ratio hygiene (:258-280), optional opponent-data reuse
(:325-344), ``noptepochs x nminibatches`` shuffled minibatch steps with KL early stop (:355-398), checkpoint every update
(:459-464) -- but rollout buffers stay in HBM and each minibatch is a row-index gather inside the gradient kernel.
Not reproduced: matplotlib histograms per update (:292-318), TF summaries.
Multi-GPU: every rank runs this with its own env shard; ``comm`` (torch.distributed group) makes PPOModel all-reduce the
advantage moments and the fused gradient buffer.
"""
import os
import os.path as osp
import time
from collections import deque

import numpy as np

from . import dist as sdist
from .model import PPOModel
from .policies import build_policy
from .runner import Runner


def constfn(val):
    def f(_):
        return val
    return f


def safemean(xs):
    return np.nan if len(xs) == 0 else np.mean(xs)


def explained_variance(ypred, y):
    """baselines/baselines/common/math_util.py:25-38"""
    vary = np.var(y)
    return np.nan if vary == 0 else 1 - np.var(y - ypred) / vary


def clean_ratio(r, clip_ratio):
    """alg_ppo.py:258-280 (one of its three identical blocks) on a device tensor: NaN -> clip_ratio, the mean and the
    fraction above the clip taken BEFORE clipping, then clamp to [0, clip_ratio].  Returns (cleaned, mean, clip_frac)."""
    import torch
    r = torch.where(torch.isnan(r), torch.full_like(r, clip_ratio), r)
    return r.clamp(0.0, clip_ratio), float(r.mean().item()), float((r > clip_ratio).float().mean().item())


def assemble_update_batch(obs, returns, masks, actions, values, neglogpacs, rewards, off_policy_ratio, total_ratio, *, nbatch,
                          neglogp_threshold, use_opponent_data, vgap=None, version_gap=None):
    """alg_ppo.py:286-344 on device tensors: the opponent samples whose learner-neglogp is below the threshold
    (``usable_index``), agent 0's rollout alone or followed by agent 1's usable rows, and the importance weights of
    ``direct`` / ``off_policy`` / ``both``.  The ratios are the CLEANED ones.  Returns a dict of the minibatch
    source arrays plus ``weights``, ``usable_index`` and ``useful_ratio`` (tests/test_update_glue.py checks it against a numpy
    restatement of the cited lines)."""
    import torch
    dev = returns.device
    usable = torch.nonzero(neglogpacs[1] < neglogp_threshold).flatten()
    names = ("obs", "returns", "masks", "actions", "values", "neglogpacs", "rewards")
    arrs = (obs, returns, masks, actions, values, neglogpacs, rewards)
    use_opp = use_opponent_data is not None and not (vgap is not None and version_gap is not None and version_gap > vgap)
    if not use_opp:                                                      # alg_ppo.py:325-330
        out = {k: x[0] for k, x in zip(names, arrs)}
    else:                                                                # :331-335
        out = {k: torch.cat([x[0], x[1][usable]], dim=0) for k, x in zip(names, arrs)}
    ones = torch.ones(nbatch, dtype=torch.float32, device=dev)
    if use_opponent_data is None:                                        # :337-344, keyed on the mode alone like the reference
        weights = ones
    elif use_opponent_data == "direct":
        weights = torch.ones(out["obs"].shape[0], dtype=torch.float32, device=dev)
    elif use_opponent_data == "off_policy":
        weights = torch.cat([ones, off_policy_ratio[usable]])
    elif use_opponent_data == "both":
        weights = torch.cat([ones, total_ratio[usable]])
    else:
        raise ValueError("use_opponent_data %r" % (use_opponent_data,))
    out.update(weights=weights, usable_index=usable, useful_ratio=float(usable.numel()) / float(neglogpacs[1].numel()))
    return out


def selection_probs(action_prob, new_action_probs):
    """alg_ppo.py:237-242: mean |new/old - 1| of the candidates' ``action_probability`` outputs on the last rollout's opponent
    samples, normalised to sampling probabilities (uniform when every candidate equals the current opponent)."""
    import torch

    def score(nap):
        r = nap / action_prob - 1.0
        r = r[torch.isfinite(r)]           # a probability that underflowed to 0 in float32 (sharp late-training policies) gives inf / NaN here;
        return float(r.abs().mean().item()) if r.numel() else 0.0      # the reference would pass NaN to np.random.choice and stop -- those samples are left out
    rd = np.array([score(nap) for nap in new_action_probs])
    tot = rd.sum()
    return rd / tot if np.isfinite(tot) and tot > 0 else np.full(len(rd), 1.0 / len(rd))


def learn(*, network, env, total_timesteps, opponent_mode="ours", use_opponent_data=None, eval_env=None, seed=None, nsteps=2048,
          ent_coef=0.0, lr=3e-4, vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95, rho_bar=1.0, c_bar=1.0, log_interval=10,
          nminibatches=4, noptepochs=4, cliprange=0.2, save_interval=1, load_path=None, model_fn=None, update_fn=None, init_fn=None,
          nagent=1, anneal_bound=500, vgap=None, kl_threshold=None, neglogp_threshold=10000.0, log_dir=None, comm=None,
          verbose=True, fix_opponent_path=None, opponent_pool=1, **network_kwargs):
    import torch
    if seed is not None:                                                # set_global_seeds (misc_util.py:48-62)
        np.random.seed(seed)
        torch.manual_seed(seed)
    lr = constfn(lr) if isinstance(lr, float) else lr
    cliprange = constfn(cliprange) if isinstance(cliprange, float) else cliprange
    total_timesteps = int(total_timesteps)
    rank = 0 if comm is None else torch.distributed.get_rank(comm)
    world = 1 if comm is None else torch.distributed.get_world_size(comm)
    policy = build_policy(env, network, **network_kwargs)
    nenvs = env.num_envs
    ob_space, ac_space = env.observation_space[0], env.action_space[0]
    nbatch = nenvs * nsteps
    nbatch_train = nbatch // nminibatches
    recurrent = network == "lstm"
    if recurrent:
        from .lstm_model import LstmPPOModel
        if use_opponent_data is not None:
            raise NotImplementedError("recurrent policies: no opponent-data reuse")
        assert nenvs % nminibatches == 0, "recurrent minibatches are whole env sequences: nenvs %% nminibatches must be 0"
    model_fn = model_fn or (LstmPPOModel if recurrent else PPOModel)
    dev = getattr(env, "device", torch.device("cuda", 0))
    mk = lambda scope, trainable: model_fn(policy=policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=nenvs,
                                           nbatch_train=nbatch_train, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef,
                                           max_grad_norm=max_grad_norm, trainable=trainable, model_scope=scope, device=dev.index or 0,
                                           comm=comm if trainable else None)
    model = mk("model_0", True)
    model.equal_counts = use_opponent_data is None      # opponent-data reuse makes per-rank minibatch sizes differ
    sdist.broadcast_params(model.params, comm)                          # sync_from_root (ppo2/model.py:129-131)
    models = [model] + [mk("model_%d" % i, False) for i in range(1, nagent)]
    model_util = mk("model_util", False)
    log_dir = log_dir or os.environ.get("OPENAI_LOGDIR") or "/tmp/robosumo_selfplay_amd"
    checkdir = osp.join(log_dir, "checkpoints") if world == 1 else osp.join(log_dir, "checkpoints")
    if rank == 0:
        model.save(osp.join(checkdir, "00000"))                          # alg_ppo.py:122-123
    if comm is not None:
        torch.distributed.barrier(comm)
    if load_path is not None:
        for m in models:
            m.load(load_path)
    for i, m in enumerate(models):
        m.act_model.seed((seed or 0) * 1000 + 17 * i + rank)
    runner = Runner(env=env, models=models, nsteps=nsteps, nagent=nagent, gamma=gamma, lam=lam, rho_bar=rho_bar, c_bar=c_bar,
                    anneal_bound=anneal_bound)
    # opponent_pool = K > 1 (extension, BASELINE config 5): K frozen snapshots stay resident in HBM and every env plays against its own
    # one (opponent_pool.py); each update draws K snapshots by the selection law of ``opponent_mode`` instead of one.  K = 1 is the
    # reference: one snapshot for all parallel envs (alg_ppo.py:213-214).
    pool = None
    opp_ref = models[1] if nagent > 1 else None      # the single-snapshot opponent model (reference of the 'ours' selector under a pool)
    if int(opponent_pool) > 1:
        if opponent_mode == "fix":
            raise ValueError("opponent_pool > 1 makes no sense with a fixed opponent")
        from .opponent_pool import LstmOpponentPool, OpponentPool
        pool = (LstmOpponentPool if recurrent else OpponentPool)(policy, int(opponent_pool), nenvs, dev)
        if recurrent:
            pool.seed((seed or 0) * 1000 + 17 + rank)
            runner.models[1] = pool           # acts for agent 1 and scores agent 0's actions, each env tile with its own snapshot
        else:
            if not runner.fused_ok():
                raise NotImplementedError("opponent_pool > 1 with MLP policies runs inside the fused rollout launch (SUMO_FUSED_ROLLOUT != 0)")
            runner.opponent_pool = pool
    epinfobuf = deque(maxlen=100)
    shuffle_gen = torch.Generator(device=dev)
    shuffle_gen.manual_seed((seed or 0) * 7919 + 13)          # same minibatch order on every rank (equal shards)
    if init_fn is not None:
        init_fn()
    tfirststart = time.perf_counter()
    history = dict(version_gap=[], off_policy_ratio_mean=[], off_env_ratio_mean=[], total_ratio_mean=[], off_policy_ratio_clip_frac=[],
                   off_env_ratio_clip_frac=[], total_ratio_clip_frac=[], useful_ratio=[], opponent_versions=[], ppo_clip_frac=[],
                   approxkl=[], early_stop_info=[], lossvals=[], fps=[], rollout_s=[], update_s=[],
                   env_diverged=[], env_dropped_contacts=[], env_rollout_aborts=[])     # per update, from the engine's counters
    env_stats_prev = env.stats() if hasattr(env, "stats") else None
    nupdates = total_timesteps // nbatch
    idx_choice = 0
    opponent_obs = opponent_actions = None
    for update in range(1, nupdates + 1):
        assert nbatch % nminibatches == 0
        tstart = time.perf_counter()
        frac = 1.0 - (update - 1.0) / nupdates
        lrnow, cliprangenow = lr(frac), cliprange(frac)
        # ---- opponent selection (alg_ppo.py:191-247); rank 0 decides, everyone loads the same file
        if opponent_mode == "fix":                                       # alg_ppo.py:194-206: a policy-zoo MLP net
            if update == 1:
                from .policy_zoo import FixedOpponentModel, load_zoo_policy
                if fix_opponent_path is None:
                    raise ValueError("opponent_mode='fix' needs fix_opponent_path=<policy_zoo .npy> (reference default: "
                                     "robosumo/robosumo/policy_zoo/assets/ant/mlp/agent-params-v3.npy)")
                zoo = load_zoo_policy(fix_opponent_path, ac_space.shape[0], device=dev)
                zoo.seed((seed or 0) * 1000 + 17 + rank)
                runner.models[1] = FixedOpponentModel(zoo)
        elif update == 1:
            if not (recurrent and pool is not None):
                runner.models[1].load(osp.join(checkdir, "00000"))
            else:        # alg_ppo.py:208 loads 00000 into models[1]: the 'ours' selector of update 2 scores the opponent the rollout faced
                opp_ref.load(osp.join(checkdir, "00000"))
            if pool is not None:
                pool.set_snapshot(0, osp.join(checkdir, "00000"))
                pool.assign_round_robin([0])
            history["version_gap"].append(0)
            history["opponent_versions"].append([0])
        else:
            paths = sorted(osp.join(checkdir, f) for f in os.listdir(checkdir))
            K = pool.capacity if pool is not None else 1
            if opponent_mode == "random":
                choices = [int(x) for x in np.random.choice(update, K)]
                history["version_gap"].append(update - 1 - choices[0])
            elif opponent_mode == "latest":
                choices = list(range(len(paths) - 1, max(-1, len(paths) - 1 - K), -1))
            elif opponent_mode == "ours":                                # ratio-divergence sampling (:227-244)
                ref = opp_ref if (recurrent and pool is not None) else runner.models[1]
                ap = ref.act_model.action_probability(opponent_obs, given_action=opponent_actions)
                sub = np.sort(np.random.choice(len(paths), 30, replace=False)) if len(paths) > 30 else np.arange(len(paths))
                naps = []
                for i in sub:
                    model_util.load(paths[i])
                    naps.append(model_util.act_model.action_probability(opponent_obs, given_action=opponent_actions))
                rd = selection_probs(ap, naps)
                choices = [int(sub[j]) for j in np.random.choice(len(rd), K, p=rd)]
            else:
                raise ValueError("opponent_mode %r" % (opponent_mode,))
            if comm is not None:                                         # rank 0 decides, everyone loads the same files
                c = torch.tensor(choices + [-1] * (K - len(choices)), device=dev)
                torch.distributed.broadcast(c, 0, group=comm)
                choices = [int(x) for x in c.tolist() if x >= 0]
            idx_choice = choices[0]
            history["opponent_versions"].append(list(choices))
            if recurrent and pool is not None:
                opp_ref.load(paths[idx_choice])                          # reference opponent of the 'ours' selector
            else:
                runner.models[1].load(paths[idx_choice])
            if pool is not None:
                for k, ci in enumerate(choices):
                    pool.set_snapshot(k, paths[ci])
                pool.assign_round_robin(range(len(choices)))
        # ---- rollout
        obs, returns, masks, actions, values, neglogpacs, rewards, opponent_neglogpacs, opp_obs_s, opp_act_s, states, epinfos, \
            off_policy_ratio, off_env_ratio, total_ratio = runner.run(update)
        torch.cuda.synchronize(dev)
        t_roll = time.perf_counter() - tstart
        if isinstance(obs, np.ndarray):       # host-mode Runner (recurrent models): continue on the device like the MLP path
            up = lambda x: torch.as_tensor(np.ascontiguousarray(x)).to(dev)
            obs, returns, masks, actions, values, neglogpacs, rewards, opponent_neglogpacs = map(
                up, (obs, returns, masks, actions, values, neglogpacs, rewards, opponent_neglogpacs))
            off_policy_ratio, off_env_ratio, total_ratio = map(up, (off_policy_ratio, off_env_ratio, total_ratio))
        # un-scrambled opponent data for the 'ours' selector: rows = agent 1's (obs, action), env-major
        opponent_obs, opponent_actions = obs[1], actions[1]
        # ---- ratio hygiene (alg_ppo.py:258-280)
        off_policy_ratio, m, f = clean_ratio(off_policy_ratio, rho_bar)
        history["off_policy_ratio_mean"].append(m); history["off_policy_ratio_clip_frac"].append(f)
        off_env_ratio, m, f = clean_ratio(off_env_ratio, rho_bar)
        history["off_env_ratio_mean"].append(m); history["off_env_ratio_clip_frac"].append(f)
        total_ratio, m, f = clean_ratio(total_ratio, rho_bar)
        history["total_ratio_mean"].append(m); history["total_ratio_clip_frac"].append(f)
        # ---- usable opponent samples, batch assembly, weights (alg_ppo.py:286-344)
        ub = assemble_update_batch(obs, returns, masks, actions, values, neglogpacs, rewards, off_policy_ratio, total_ratio,
                                   nbatch=nbatch, neglogp_threshold=neglogp_threshold, use_opponent_data=use_opponent_data, vgap=vgap,
                                   version_gap=history["version_gap"][-1] if history["version_gap"] else None)
        b_obs, b_ret, b_act, b_val, b_nlp, weights = ub["obs"], ub["returns"], ub["actions"], ub["values"], ub["neglogpacs"], ub["weights"]
        history["useful_ratio"].append(ub["useful_ratio"])
        b_obs = b_obs.contiguous()
        epinfobuf.extend(epinfos[-epinfobuf.maxlen:])                    # the deque keeps the last 100 anyway (alg_ppo.py:160,347)
        # ---- minibatch SGD (alg_ppo.py:355-398)
        nsamp = b_obs.shape[0]
        mblossvals, early_stop, stop_info = [], False, None
        if recurrent:                          # baselines ppo2 recurrent convention: minibatches of whole env sequences
            envsperbatch = nenvs // nminibatches
            envinds = np.arange(nenvs)
            flatinds = np.arange(nenvs * nsteps).reshape(nenvs, nsteps)
            st0 = states if torch.is_tensor(states) else torch.as_tensor(np.asarray(states, np.float32)).to(dev)
            b_masks = masks[0]
            for epoch in range(noptepochs):
                np.random.shuffle(envinds)
                for start in range(0, nenvs, envsperbatch):
                    mbenv = envinds[start:start + envsperbatch]
                    mbflat = torch.from_numpy(flatinds[mbenv].ravel()).to(dev)
                    out = model.train(lrnow, cliprangenow, b_obs[mbflat], b_ret[mbflat], b_masks[mbflat], b_act[mbflat], b_val[mbflat],
                                      b_nlp[mbflat], None, weights[mbflat], st0[torch.from_numpy(mbenv).to(dev)], nsteps=nsteps)
                    mblossvals.append(torch.tensor([float(x) for x in out[:5]], dtype=torch.float64))
        if not recurrent and hasattr(model, "begin_update"):
            model.begin_update(b_obs, b_ret, b_act, b_val, b_nlp, weights)   # the update's batch, handed over once (model.py)
        # minibatch steps per epoch (alg_ppo.py:378): with opponent-data reuse the ranks hold different numbers of rows, so they
        # agree on the largest count and short ranks finish the epoch with empty minibatches (same collectives on every rank)
        nmb_steps = -(-nsamp // nbatch_train)
        if comm is not None and not model.equal_counts:
            nmb_steps = sdist.agree_max(nmb_steps, comm, device=dev)
        for epoch in range(noptepochs if not recurrent else 0):
            inds = torch.randperm(nsamp, device=dev, generator=shuffle_gen).to(torch.int32)   # np.random.shuffle (:375), on the device
            if comm is not None and hasattr(model, "prepare_epoch"):
                # equal shards: the advantage moments of all the epoch's minibatches in ONE all-reduce, so that every optimiser step
                # issues exactly one collective; a no-op with opponent-data reuse (unequal shards)
                model.prepare_epoch(inds, nbatch_train)
            for ii in range(nmb_steps):
                mb = inds[ii * nbatch_train:(ii + 1) * nbatch_train]       # empty once this rank's rows are used up
                # statistics stay on the device unless the KL early stop needs them now (alg_ppo.py:389-398)
                out = model.train_indexed(lrnow, cliprangenow, b_obs, b_ret, b_act, b_val, b_nlp, weights, mb, int(mb.numel()),
                                          sync=kl_threshold is not None, mb_index=ii)
                mblossvals.append(out[:5] if kl_threshold is not None else out)
                if kl_threshold is not None and out[3] > kl_threshold * 1.5:
                    early_stop, stop_info = True, [epoch, ii]
                    break
            if early_stop:
                break
        if not recurrent and hasattr(model, "end_update"):
            model.end_update()
        history["early_stop_info"].append(stop_info)
        if kl_threshold is None or recurrent:
            lossvals = torch.stack(mblossvals).mean(dim=0).cpu().numpy().astype(np.float64)
        else:
            lossvals = np.mean(np.array(mblossvals, dtype=np.float64), axis=0)
        history["lossvals"].append(lossvals)
        history["ppo_clip_frac"].append(lossvals[-1])
        history["approxkl"].append(lossvals[-2])
        torch.cuda.synchronize(dev)
        tnow = time.perf_counter()
        history["rollout_s"].append(t_roll)
        history["update_s"].append(tnow - tstart - t_roll)
        history["fps"].append(nbatch * world / (tnow - tstart))
        # the engine's fault counters of this update (the reference is loud here: a MuJoCo warning raises MujocoException,
        # mujoco-py builder.py:351-369; the fused launch's abort already raised in Runner.run): diverged env steps (episodes ended
        # by the bad-value guard), contacts dropped for lack of room, aborted hand-over waits
        env_note = ""
        if env_stats_prev is not None:
            st_now = env.stats()
            dv, dc = st_now["diverged"] - env_stats_prev["diverged"], st_now["dropped"] - env_stats_prev["dropped"]
            ab = (st_now["rollout_aborts"] + st_now.get("handover_mismatches", 0)
                  - env_stats_prev["rollout_aborts"] - env_stats_prev.get("handover_mismatches", 0))
            env_stats_prev = st_now
            history["env_diverged"].append(int(dv)); history["env_dropped_contacts"].append(int(dc)); history["env_rollout_aborts"].append(int(ab))
            if dv or dc or ab:
                env_note = "  [env: %d diverged steps, %d dropped contacts, %d rollout aborts]" % (dv, dc, ab)
        if update_fn is not None:
            update_fn(update)
        if verbose and rank == 0 and (update % log_interval == 0 or update == 1):
            ev = explained_variance(b_val.cpu().numpy(), b_ret.cpu().numpy())
            print("update %d/%d  fps %.0f  rollout %.2fs  sgd %.2fs  ev %.3f  eprewmean %.2f  eplenmean %.1f  %s" % (
                update, nupdates, history["fps"][-1], t_roll, tnow - tstart - t_roll, ev, safemean([e["r"] for e in epinfobuf]),
                safemean([e["l"] for e in epinfobuf]), " ".join("%s %.4g" % (n, v) for n, v in zip(model.loss_names, lossvals))) + env_note,
                flush=True)
        elif env_note and rank == 0:
            print("update %d/%d%s" % (update, nupdates, env_note), flush=True)
        if save_interval and (update % save_interval == 0 or update == 1) and rank == 0:
            model.save(osp.join(checkdir, "%.5i" % update))               # alg_ppo.py:459-464
        if comm is not None:
            sdist.assert_synced(model.params, comm)                       # MpiAdamOptimizer.check_synced (mpi_adam_optimizer.py:54-67)
            torch.distributed.barrier(comm)
    model.history = history
    model.time_elapsed = time.perf_counter() - tfirststart
    return model
