"""Dev probe: compare HIP engine against the oracle on the GPU box (forward + multi-step)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import mjcf, capi
from oracle.oracle import OracleSim

env_id = sys.argv[1] if len(sys.argv) > 1 else "RoboSumo-Ant-vs-Ant-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = mjcf.load_model(env_id)
eng = capi.Engine(m, N)
print("dims", eng.nq, eng.nv, "maxcon", eng.maxcon, "lds", eng.lds_bytes, flush=True)
ora = OracleSim(m, N, maxcon=eng.maxcon, jbcap=eng.jbcap)
dev = torch.device("cuda:0")
obs = torch.zeros((N, 2, eng.obs_stride), dtype=torch.float32, device=dev)
seeds = np.arange(N, dtype=np.uint64) + 1000
eng.reset(obs.data_ptr(), seeds=seeds)
torch.cuda.synchronize()
oobs = ora.reset(seeds=seeds)
print("reset obs maxdiff", np.abs(obs.cpu().numpy() - oobs).max(), flush=True)
q, v, w, c = eng.get_state(); oq, ov, ow, oc = ora.get_state()
print("reset state diff", np.abs(q - oq).max(), np.abs(v - ov).max(), (c != oc).sum(), flush=True)
# forward parity at reset state
ctrl = np.random.default_rng(0).uniform(-1, 1, (N, eng.nu))
qacc, counts = eng.debug_forward(ctrl)
oqacc = np.zeros_like(qacc); ocounts = np.zeros((N, 2))
for e in range(N):
    ora.set_state(q, v, w, c) if e == 0 else None
    ora.forward(e, ctrl[e]); oqacc[e] = ora.array("qacc", e); ocounts[e] = ora.array("counts", e)[:2]
err = np.abs(qacc - oqacc).max(axis=1) / (1e-9 + np.abs(oqacc).max(axis=1))
print("forward qacc rel err max", err.max(), "counts equal", (counts[:, :2] == ocounts).all(), flush=True)
# multi-step parity with resync each step
rng = np.random.default_rng(1)
act = torch.zeros((N, 2, eng.act_stride), dtype=torch.float32, device=dev)
info = torch.zeros((N, 2, 8), dtype=torch.float64, device=dev)
done = torch.zeros((N, 2), dtype=torch.uint8, device=dev)
ep_r = torch.zeros(N, dtype=torch.float64, device=dev); ep_dr = torch.zeros_like(ep_r)
ep_l = torch.zeros(N, dtype=torch.int32, device=dev)
worst = 0; flagdiff = 0
for t in range(int(sys.argv[3]) if len(sys.argv) > 3 else 60):
    a = rng.standard_normal((N, 2, eng.act_stride)).astype(np.float32)
    act.copy_(torch.from_numpy(a))
    eng.step(act.data_ptr(), obs.data_ptr(), info.data_ptr(), done.data_ptr(), ep_r.data_ptr(), ep_dr.data_ptr(), ep_l.data_ptr())
    torch.cuda.synchronize()
    oobs, oinfo, odone, oepr, oepdr, oepl = ora.step(a, nthreads=8)
    gobs = obs.cpu().numpy(); ginfo = info.cpu().numpy(); gdone = done.cpu().numpy()
    e_obs = np.abs(gobs - oobs).max(); e_info = (np.abs(ginfo - oinfo) / (1 + np.abs(oinfo))).max()
    flagdiff += int((gdone != odone).sum()) + int((ep_l.cpu().numpy() != oepl).sum())
    worst = max(worst, e_obs, e_info)
    if t % 10 == 0:
        print(t, "obs err", e_obs, "info err", e_info, "dones", int(gdone[:, 0].sum()), flush=True)
    # resync the device to the oracle state so errors don't compound chaotically
    oq, ov, ow, oc = ora.get_state()
    eng.set_state(oq, ov, ow, oc)
print("worst err", worst, "flag mismatches", flagdiff)
print("gpu stats", eng.stats()); print("ora stats", ora.stats())
# throughput
Nb = 4096
eng2 = capi.Engine(m, Nb)
obs2 = torch.zeros((Nb, 2, eng2.obs_stride), dtype=torch.float32, device=dev)
act2 = torch.randn((Nb, 2, eng2.act_stride), dtype=torch.float32, device=dev)
info2 = torch.zeros((Nb, 2, 8), dtype=torch.float64, device=dev); done2 = torch.zeros((Nb, 2), dtype=torch.uint8, device=dev)
epr2 = torch.zeros(Nb, dtype=torch.float64, device=dev); epdr2 = torch.zeros_like(epr2); epl2 = torch.zeros(Nb, dtype=torch.int32, device=dev)
eng2.reset(obs2.data_ptr(), seeds=np.arange(Nb, dtype=np.uint64))
for i in range(20):
    eng2.step(act2.data_ptr(), obs2.data_ptr(), info2.data_ptr(), done2.data_ptr(), epr2.data_ptr(), epdr2.data_ptr(), epl2.data_ptr())
torch.cuda.synchronize(); t0 = time.time()
K = 50
for i in range(K):
    eng2.step(act2.data_ptr(), obs2.data_ptr(), info2.data_ptr(), done2.data_ptr(), epr2.data_ptr(), epdr2.data_ptr(), epl2.data_ptr())
torch.cuda.synchronize(); dt = time.time() - t0
print("N=%d: %.3f ms/step, %.0f env-steps/s" % (Nb, dt / K * 1e3, Nb * K / dt))
print("stats", eng2.stats())
