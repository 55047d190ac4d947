"""Regenerates the policy-zoo fixtures from the parameter files the reference ships
(/root/reference/robosumo/robosumo/policy_zoo/assets/{ant,bug,spider}/{mlp,lstm}/agent-params-v{1,2,3}.npy):

  zoo_obsfilter_stats.json  for all 18 files: the observation filter's count, mean and std (utils.py:9-32: sum / count,
                            sqrt(max(sumsq / count - mean^2, 1e-2))), the raw variance before the 1e-2 floor, the return
                            filter's mean / std and the policy's logstd.  These running statistics were accumulated by the
                            authors' training runs in real MuJoCo over ~5e8 observations: they are the only MuJoCo-produced
                            numbers in the reference tree, and tests/test_zoo_validation.py / test_gpu_zoo_validation.py
                            compare simulated zoo-vs-zoo play against them.
  zoo_v3_params.npz         the six v3 parameter vectors themselves (float32, 1.4 MB), so that those tests can roll the nets
                            out where /root/reference does not exist (the GPU box).  Data, not source: flat weight vectors.

  ant_density10_model.json  Ant-vs-Ant compiled with agent_densities = [10, 10] (see below).

The files are read with numpy.load(allow_pickle=False); nothing of the reference is imported or executed."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from robosumo_selfplay_amd import policy_zoo  # noqa: E402

ROOT = "/root/reference/robosumo/robosumo/policy_zoo/assets"
AC = {"ant": 8, "bug": 12, "spider": 16}


def raw_stats(p, prefix):
    cnt = np.float64(p[prefix + "/count"])
    mean = np.asarray(p[prefix + "/sum"], np.float64) / cnt
    var = np.asarray(p[prefix + "/sumsq"], np.float64) / cnt - mean * mean
    return cnt, mean, var


def main():
    stats, params = {}, {}
    for kind, ac in AC.items():
        for net in ("mlp", "lstm"):
            for v in (1, 2, 3):
                flat = np.load(os.path.join(ROOT, kind, net, "agent-params-v%d.npy" % v), allow_pickle=False)
                assert flat.dtype == np.float32 and flat.ndim == 1
                ob_dim, p = (policy_zoo.split_zoo_mlp if net == "mlp" else policy_zoo.split_zoo_lstm)(flat, ac)
                cnt, mean, var = raw_stats(p, "obsfilter")
                rcnt, rmean, rvar = raw_stats(p, "retfilter")
                stats["%s-%s-v%d" % (kind, net, v)] = dict(
                    ob_dim=int(ob_dim), ac_dim=ac, nparams=int(flat.size), obs_count=float(cnt),
                    obs_mean=[float(x) for x in mean], obs_var_raw=[float(x) for x in var],
                    obs_std=[float(x) for x in np.sqrt(np.maximum(var, 1e-2))],
                    ret_count=float(rcnt), ret_mean=float(rmean), ret_std=float(np.sqrt(max(float(rvar), 1e-2))),
                    logstd=[float(x) for x in p["logstd"].ravel()])
                if v == 3:
                    params["%s-%s-v3" % (kind, net)] = flat
    with open(os.path.join(HERE, "zoo_obsfilter_stats.json"), "w") as f:
        json.dump(stats, f)
    np.savez(os.path.join(HERE, "zoo_v3_params.npz"), **params)
    # the Ant-vs-Ant scene at construct_scene's DEFAULT density 10 (utils.py:97-99) instead of the registry's 13: the zoo's force
    # statistics match this mass (tests/test_zoo_validation.py).  Derived constant tables only, like robosumo_selfplay_amd/assets.
    from robosumo_selfplay_amd import mjcf
    ad = "/root/reference/robosumo/robosumo/envs/assets"
    m10 = mjcf.compile_scene(os.path.join(ad, "tatami.xml"), [os.path.join(ad, "ant.xml")] * 2, ["ant", "ant"], [10.0, 10.0], 2.0, 500,
                             name="RoboSumo-Ant-vs-Ant-v0@density10")
    with open(os.path.join(HERE, "ant_density10_model.json"), "w") as f:
        f.write(m10.to_json())
    for k, s in stats.items():
        print(k, s["ob_dim"], "count %.3g" % s["obs_count"], "z mean %.3f std %.3f" % (s["obs_mean"][2], s["obs_std"][2]))


if __name__ == "__main__":
    main()
