"""Dev probe: replay the step-parity test sequence; on the first mismatching env print contact counts / qacc differences
at every forward of that env step (engine debug_forward vs oracle forward, both driven through the oracle's RK4 states)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from test_gpu_env_parity import Pair

env_id = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Spider-vs-Spider-v0"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
p = Pair(env_id, 64)
p.reset()
rng = np.random.default_rng(1)
for t in range(steps):
    a = (rng.standard_normal((p.N, 2, p.eng.act_stride)) * (1.0 if t % 3 else 2.5)).astype(np.float32)
    pre = p.ora.get_state()
    (gobs, ginfo, gdone, gr, gdr, gl), (oobs, oinfo, odone, orr, odr, ol) = p.step(a)
    err = np.abs(gobs - oobs).reshape(p.N, -1).max(axis=1)
    badenvs = np.nonzero(err > 1e-5)[0]
    print("step", t, "max obs err", err.max(), "bad envs", badenvs, flush=True)
    if len(badenvs):
        e = int(badenvs[0])
        # replay env e sub-step by sub-step on the oracle; at each mj_step start compare one forward
        p.ora.set_state(*pre); p.eng.set_state(*pre)
        ctrl = np.clip(a.reshape(p.N, -1).astype(np.float64), -1, 1)
        for sub in range(5):
            q, v, w, c = p.ora.get_state()
            p.eng.set_state(q, v, w, c)
            qacc, counts = p.eng.debug_forward(ctrl)
            p.ora.forward(e, ctrl[e])
            oq = p.ora.array("qacc", e); oc = p.ora.array("counts", e)
            print(" sub", sub, "gpu counts", counts[e], "ora counts", oc, "qacc err", np.abs(qacc[e] - oq).max(), flush=True)
            if counts[e][0] != oc[0] or np.abs(qacc[e] - oq).max() > 1e-6:
                con = p.ora.array("contacts", e).reshape(-1, 9)
                gt = p.m.tables["geom_type"]
                for cc in con:
                    print("   ora contact dist %.6g g1 %d (t%d) g2 %d (t%d) pos %s" % (cc[0], cc[7], gt[int(cc[7])], cc[8], gt[int(cc[8])], cc[1:4]))
            p.ora.mj_step(e, ctrl[e], 1)
        break
    oq, ov, ow, oc = p.ora.get_state()
    p.eng.set_state(oq, ov, ow, oc)
