#!/usr/bin/env python3
"""Development: does the step kernel read hardware state it did not write, and WHERE does that state live?  Before every env launch a scrub
kernel (tools/dev/scrub.hip -> libscrub.so: hipcc -O3 --offload-arch=gfx950 -shared -fPIC) fills VGPRs / LDS / a scratch frame of every wave slot
with a pattern; identical seeded 4096-env runs are then compared with the first one.  How the round-3 nondeterminism was located: any full
scrub (zeros, NaNs, large numbers alike) removed it, scrubbing the VGPRs alone did, then 16-register groups (vmask), then single registers
(fgrp / fmask) -> v140, which the disassembly showed to be a save slot filled under a partial EXEC mask (robosumo_selfplay_amd/codegen_check.py).
usage: scrub_bisect.py <none | zero | nan | big | once | tiny | sleep> [reps] [what: 1 LDS | 2 scratch | 4 VGPRs] [vmask] [fgrp] [fmask]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robosumo_selfplay_amd import mjcf
from robosumo_selfplay_amd.vec_env import SumoVecEnv
mode = sys.argv[1]
pat = {"none": None, "zero": 0, "nan": 0x7ff80000, "big": 0x7f700000, "once": 0, "tiny": 0, "sleep": None, "noscratch": 0}[mode]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = C.CDLL(os.path.join(ROOT, "tools", "dev", "libscrub.so"))
S.scrub.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint, C.c_int, C.c_uint]
FGRP = int(sys.argv[5]) if len(sys.argv) > 5 else -1
FMASK = int(sys.argv[6], 0) if len(sys.argv) > 6 else 0   # single registers of group FGRP
VMASK = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xFFFF   # groups of 16 VGPRs
WHAT = int(sys.argv[3]) if len(sys.argv) > 3 else 15      # 1 LDS | 2 scratch frame | 4 VGPRs | 8 SGPRs
once_done = [False]
def scrub():
    if mode == "sleep":
        torch.cuda._sleep(2000000)      # ~1 ms of spinning on the stream: a delay without scratch, LDS or register traffic
        return
    if mode == "once":
        if once_done[0]:
            return
        once_done[0] = True
    if pat is not None:
        rc = S.scrub(pat, 1 if mode == "tiny" else 2048, 20480, torch.cuda.current_stream().cuda_stream, WHAT, VMASK, FGRP, FMASK)
        assert rc == 0, rc
env_id, N, T = "RoboSumo-Ant-vs-Ant-v0", 4096, 4
m = mjcf.load_model(env_id)
A = int(m.act_dims[0])
g = torch.Generator(device="cpu").manual_seed(0)
acts = torch.randn((6, N, 2, A), generator=g).to("cuda")
ref, nbad_runs, nbad_envs, ndiv = None, 0, 0, 0
for rep in range(reps):
    env = SumoVecEnv(env_id, num_envs=N, seed=7, model=m)
    scrub()
    env.reset_device()
    tr = []
    for t in range(T):
        scrub()
        env.step_device(acts[t].contiguous())
        torch.cuda.synchronize()
        q, v, w, c = env.engine.get_state()
        tr.append((q.copy(), v.copy()))
    ndiv += env.stats()["diverged"]
    env.close()
    if ref is None:
        ref = tr
        continue
    bad = set()
    for t in range(T):
        bad.update(np.nonzero((tr[t][0] != ref[t][0]).any(1) | (tr[t][1] != ref[t][1]).any(1))[0].tolist())
    nbad_runs += bool(bad); nbad_envs += len(bad)
print("what %d vmask %#x fine %d:%#x " % (WHAT, VMASK, FGRP, FMASK) + "pattern %s: %d of %d repeats differ from the first run, %d envs in total; diverged (all runs) %d" % (sys.argv[1], nbad_runs, reps - 1, nbad_envs, ndiv))
