"""Regenerates tests/golden/zoo_mlp_layout.json from the policy-zoo parameter files the reference ships
(robosumo/robosumo/policy_zoo/assets/*/mlp/agent-params-v*.npy).  Data only: vector lengths and a few slices that pin the
flat layout (filter counts, logstd tail); the files are read with numpy.load(allow_pickle=False)."""
import json, os, sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from robosumo_selfplay_amd import policy_zoo

ROOT = "/root/reference/robosumo/robosumo/policy_zoo/assets"
AC = {"ant": 8, "bug": 12, "spider": 16}
out = {}
for kind, ac in AC.items():
    for v in (1, 2, 3):
        flat = np.load(os.path.join(ROOT, kind, "mlp", "agent-params-v%d.npy" % v), allow_pickle=False)
        ob_dim, p = policy_zoo.split_zoo_mlp(flat, ac)
        mean, std = policy_zoo.filter_stats(p, "obsfilter")
        out["%s-v%d" % (kind, v)] = dict(
            nparams=int(flat.size), dtype=str(flat.dtype), ac_dim=ac, ob_dim=ob_dim,
            obs_count=float(p["obsfilter/count"]), ret_count=float(p["retfilter/count"]),
            logstd=[float(x) for x in p["logstd"].ravel()],
            head=[float(x) for x in flat[:3]], tail=[float(x) for x in flat[-4:]],
            obs_mean_first4=[float(x) for x in mean[:4]], obs_std_first4=[float(x) for x in std[:4]],
            polfinal_b=[float(x) for x in p["polfinal/b"].ravel()[:4]])
json.dump(out, open(os.path.join(os.path.dirname(__file__), "zoo_mlp_layout.json"), "w"), indent=1)
lout = {}
for kind, ac in AC.items():
    for v in (1, 2, 3):
        flat = np.load(os.path.join(ROOT, kind, "lstm", "agent-params-v%d.npy" % v), allow_pickle=False)
        ob_dim, p = policy_zoo.split_zoo_lstm(flat, ac)
        lout["%s-v%d" % (kind, v)] = dict(
            nparams=int(flat.size), dtype=str(flat.dtype), ac_dim=ac, ob_dim=ob_dim,
            obs_count=float(p["obsfilter/count"]), ret_count=float(p["retfilter/count"]),
            logstd=[float(x) for x in p["logstd"].ravel()], tail=[float(x) for x in flat[-4:]],
            lstmp_bias_absmax=float(np.abs(p["lstmp/bias"]).max()), p_out_b=[float(x) for x in p["p/out/b"].ravel()[:4]])
json.dump(lout, open(os.path.join(os.path.dirname(__file__), "zoo_lstm_layout.json"), "w"), indent=1)
print(json.dumps({k: (v["nparams"], v["ob_dim"], v["obs_count"], v["logstd"][:2]) for k, v in lout.items()}))
print(json.dumps({k: (v["nparams"], v["ob_dim"], v["obs_count"], v["logstd"][:3]) for k, v in out.items()}, indent=0))
