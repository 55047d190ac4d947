#!/usr/bin/env python3
"""CLI counterpart of the reference's run.py (run.py:211-251) for the hot path:
    python run.py --env RoboSumo-Ant-vs-Ant-v0 --num_env 4096 --num_timesteps 2097152 --nsteps=128
Unknown ``--key=value`` flags are forwarded to ``learn`` like the reference does (run.py:29-63), but parsed with
``ast.literal_eval`` instead of ``eval``.  Under ``torchrun`` every rank takes an equal shard of ``--num_env``.
"""
import argparse
import ast
import os
import pickle
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_unknown(args):
    out = {}
    for a in args:
        if not a.startswith("--") or "=" not in a:
            raise SystemExit("cannot parse extra argument %r (expected --key=value)" % a)
        k, v = a[2:].split("=", 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="RoboSumo-Ant-vs-Ant-v0")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--num_timesteps", type=float, default=1e8)
    ap.add_argument("--network", default="mlp")
    ap.add_argument("--num_env", type=int, default=1)
    ap.add_argument("--algo", default="ppo")
    ap.add_argument("--log_path", default="results")
    ap.add_argument("--suffix", default="0")
    ap.add_argument("--env_groups", type=int, default=0, help="env groups per GPU (vec_env.SumoVecEnv): 0 = 1 when the whole rollout is one fused "
                    "launch that balances the envs itself (MLP policies, LSTM policies with nlstm 128), 2 for step-by-step launches on two streams")
    ap.add_argument("--cfrc_mode", default="zero", choices=["zero", "rne_post"], help="contact-force observation entries: zero = the reference's "
                    "behaviour (MuJoCo 2.1 without force sensors), rne_post = as mj_rnePostConstraint would fill them (second launch per step)")
    ap.add_argument("--adjust_z", type=float, default=0.0, help="Agent._adjust_z (agents.py:33): offset of the torso height the agents report "
                    "(observations, lose test).  0 = the reference's training setting (its run.py:76-77 leaves the -0.5 commented out); its "
                    "evaluation / play scripts use -0.5, which is what the policy-zoo nets expect")
    args, unknown = ap.parse_known_args(argv)
    extra = parse_unknown(unknown)
    from robosumo_selfplay_amd import alg_ppo, defaults, dist as sdist
    from robosumo_selfplay_amd.vec_env import make_vec_env
    comm = sdist.init_process_group()
    rank, local_rank, world = sdist.env_rank_world()
    log_path = os.path.join(args.log_path, "%s-%s" % (args.env, args.suffix))
    if rank == 0:
        shutil.rmtree(log_path, ignore_errors=True)                        # run.py:233-234
        os.makedirs(log_path, exist_ok=True)
    start, per = sdist.shard_envs(args.num_env, rank, world)
    if args.env_groups <= 0:
        stepwise = (os.environ.get("SUMO_FUSED_ROLLOUT", "1") == "0" or args.cfrc_mode != "zero"
                    or (args.network == "lstm" and int(extra.get("nlstm", 128)) != 128))
        args.env_groups = 2 if stepwise else 1
    groups = args.env_groups if per % max(1, args.env_groups) == 0 else 1
    import torch
    local_rank = local_rank % max(1, torch.cuda.device_count())        # gloo rehearsal of N ranks on fewer GPUs
    env = make_vec_env(args.env, per, args.seed + start, device=local_rank, groups=groups, cfrc_mode=args.cfrc_mode, adjust_z=args.adjust_z)  # run.py:144: env i gets seed + i
    kw = defaults.get_default_params(args.env, args.algo)
    kw.update(extra)
    if args.network == "lstm":       # the RoboSumo defaults describe the MLP (defaults.py:8-26); recurrent nets share the latent
        for k in ("value_network", "num_hidden", "num_layers", "activation"):
            kw.pop(k, None)
    if rank == 0:
        with open(os.path.join(log_path, "config.pkl"), "wb") as f:        # run.py:176-177
            pickle.dump(dict(vars(args), **kw), f)
    model = alg_ppo.learn(network=args.network, env=env, seed=args.seed, total_timesteps=int(args.num_timesteps) // world,
                          nagent=len(env.agents), log_dir=log_path, comm=comm, **kw)
    env.close()
    return model


if __name__ == "__main__":
    main(sys.argv[1:])
