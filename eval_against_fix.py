#!/usr/bin/env python3
"""CLI counterpart of the reference's eval_robosumo_against_fix.py (:121-262): play saved checkpoints of a run against a
fixed policy-zoo MLP opponent on the GPU and print / save win, draw and lose rates per checkpoint.

    python eval_against_fix.py --path results/RoboSumo-Ant-vs-Ant-v0-0 --opponent_path <zoo>/ant/mlp/agent-params-v3.npy \\
        --num_env 256 --rounds 512 --interval 10
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", required=True, help="run directory holding checkpoints/NNNNN (run.py --log_path/<env>-<suffix>)")
    ap.add_argument("--opponent_path", required=True, help="policy-zoo .npy (robosumo/robosumo/policy_zoo/assets/<agent>/mlp/...)")
    ap.add_argument("--env", default="RoboSumo-Ant-vs-Ant-v0")
    ap.add_argument("--num_env", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=500)
    ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--interval", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stochastic", action="store_true")
    ap.add_argument("--adjust_z", type=float, default=-0.5, help="Agent._adjust_z of every agent; the reference's evaluator sets -0.5 "
                    "(eval_robosumo_against_fix.py:108-115): the zoo nets were trained with the tatami surface at z = 0")
    ap.add_argument("--cfrc_mode", default="zero", choices=["zero", "rne_post"])
    args = ap.parse_args(argv)
    import numpy as np
    from robosumo_selfplay_amd import policy_zoo
    from robosumo_selfplay_amd.model import PPOModel
    from robosumo_selfplay_amd.policies import build_policy
    from robosumo_selfplay_amd.vec_env import make_vec_env
    env = make_vec_env(args.env, args.num_env, args.seed, adjust_z=args.adjust_z, cfrc_mode=args.cfrc_mode)   # eval_robosumo_against_fix.py:108-115
    policy = build_policy(env, "mlp", value_network="copy", num_hidden=64, activation="relu")
    model = PPOModel(policy=policy, ob_space=env.observation_space[0], ac_space=env.action_space[0], trainable=False,
                     model_scope="model_0")
    opp = policy_zoo.load_zoo_policy(args.opponent_path, env.action_space[1].shape[0])
    ckdir = os.path.join(args.path, "checkpoints")
    ids = sorted(int(f) for f in os.listdir(ckdir) if f.isdigit())
    table = []
    for cid in ids:
        if cid < args.start or (cid - args.start) % args.interval:
            continue
        model.load(os.path.join(ckdir, "%.5i" % cid))
        r = policy_zoo.evaluate_against(model, opp, env, args.rounds, deterministic=not args.stochastic)
        table.append([cid, r["win"], r["draw"], r["lose"]])
        print("-----Episode %d win: %.2f, draw: %.2f, lose: %.2f (%d rounds, %d steps)-----" % (cid, r["win"], r["draw"], r["lose"],
                                                                                           r["rounds"], r["steps"]))
    with open(os.path.join(args.path, "eval_against_fix.json"), "w") as f:
        json.dump(table, f)
    env.close()
    return np.array(table)


if __name__ == "__main__":
    main(sys.argv[1:])
