"""Dev probe: agents placed in contact; compare one forward (debug_forward vs oracle) env by env."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from test_gpu_env_parity import Pair

env_id = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Ant-vs-Ant-v0"
p = Pair(env_id, 48)
p.reset()
q, v, w, c = p.ora.get_state()
m = p.m
a0, a1 = int(m.agent_qposadr[0]), int(m.agent_qposadr[1])
rng = np.random.default_rng(5)
for e in range(p.N):
    ang = rng.uniform(0, 2 * np.pi); d = rng.uniform(0.55, 0.95); ctr = rng.uniform(-0.3, 0.3, 2)
    q[e, a0:a0 + 2] = ctr + 0.5 * d * np.array([np.cos(ang), np.sin(ang)])
    q[e, a1:a1 + 2] = ctr - 0.5 * d * np.array([np.cos(ang), np.sin(ang)])
v[:] *= 0.5
p.ora.set_state(q, v, w, c); p.eng.set_state(q, v, w, c)
ctrl = np.clip(rng.standard_normal((p.N, p.eng.nu)) * 0.7, -1, 1)
qacc, counts = p.eng.debug_forward(ctrl)
gb = m.tables["geom_bodyid"]
for e in range(p.N):
    p.ora.forward(e, ctrl[e])
    oq = p.ora.array("qacc", e); oc = p.ora.array("counts", e)
    con = p.ora.array("contacts", e).reshape(-1, 9)
    two = sum(1 for cc in con if gb[int(cc[7])] != 0 and gb[int(cc[8])] != 0)
    err = np.abs(qacc[e] - oq).max() / (1 + np.abs(oq).max())
    flag = "" if err < 1e-8 else "  <<<<<"
    print("env %2d gpu counts %s ora counts %s two-body %d mindist %.4f qacc relerr %.3e |qacc| %.1f%s" % (
        e, counts[e][:2], oc[:2].astype(int), two, con[:, 0].min() if len(con) else 0, err, np.abs(oq).max(), flag))
