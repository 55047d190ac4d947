#!/usr/bin/env python3
"""Development: shader-clock cycles per phase of ppo_grad_kernel (workgroup 0 of the policy net, wave 0) from the probe build
(hipcc ... -DPPO_GRAD_PROBE -o libsumo_ppo_probe.so; SUMO_PPO_LIB points at it).  16 384-row Ant minibatch, the bench's shapes."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import ppo_capi, policies
from robosumo_selfplay_amd.model import PPOModel
dev = torch.device("cuda", 0)
spec = policies.PolicySpec(121, 8, value_network="copy", activation="relu")
np.random.seed(0)
m = PPOModel(policy=spec, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5)
N, n, D, A = 4096 * 128, 16384, 121, 8
obs = torch.randn(N, D, device=dev)
ret, nlp, adv = torch.randn(N, device=dev), torch.rand(N, device=dev) + 8.0, torch.randn(n, device=dev)
act = torch.randn(N, A, device=dev)
idx = torch.randperm(N, device=dev)[:n].to(torch.int32)
w = torch.ones(N, device=dev)
lr = torch.empty(n, device=dev)
stats = torch.zeros(ppo_capi.NSTATS, dtype=torch.float64, device=dev)
grads = torch.zeros_like(m.grads)
L = ppo_capi.lib()
st = torch.cuda.current_stream(dev).cuda_stream
call = lambda: ppo_capi.chk(L.ppo_grad(m.params.data_ptr(), obs.data_ptr(), obs.stride(0), D, A, act.data_ptr(), adv.data_ptr(), ret.data_ptr(),
                                        nlp.data_ptr(), w.data_ptr(), idx.data_ptr(), n, 1.0 / n, 0.2, 0.0, 0.5, grads.data_ptr(), stats.data_ptr(),
                                        lr.data_ptr(), m.workspace.data_ptr(), st))
names = ["resident operands + first tile", "first layer + barrier", "second layer + barrier", "head + loss deltas", "gW2 + dh2 + barrier",
         "gW1 + dh1", "gW0 + commit + barrier", "slab"]
L.ppo_debug_gprobe.argtypes = [C.c_void_p]
for rep in range(3):
    for _ in range(4):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(8):
        call()
    e1.record()
    torch.cuda.synchronize(dev)
    out = (C.c_ulonglong * 8)()
    assert L.ppo_debug_gprobe(out) == 0
    c = np.array(list(out), dtype=np.float64)
    print("rep %d: total %d cycles (%.1f us at 2.4 GHz); call %.1f us" % (rep, c.sum(), c.sum() / 2400.0, e0.elapsed_time(e1) * 1e3 / 8))
    for k, nm in enumerate(names):
        print("   %-32s %8d  %5.1f %%" % (nm, c[k], 100 * c[k] / c.sum()))
