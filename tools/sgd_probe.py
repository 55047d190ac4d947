"""Dev probe: host enqueue time vs total time of the PPO minibatch loop (graph and eager paths)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from robosumo_selfplay_amd import policies
from robosumo_selfplay_amd.model import PPOModel
dev = torch.device("cuda:0")
nb, n, D, A = 524288, 16384, 121, 8
obs = torch.randn((nb, D), device=dev); act = torch.randn((nb, A), device=dev)
ret = torch.randn(nb, device=dev); val = torch.randn(nb, device=dev); w = torch.ones(nb, device=dev)
for graph in (True, False):
    PPOModel.use_graph = graph
    m = PPOModel(policy=policies.PolicySpec(D, A, value_network="copy", activation="relu"))
    nlp = m.act_model.action_probability(obs, given_action=act)
    if graph:
        m.begin_update(obs, ret, act, val, nlp, w)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train_indexed(3e-4, 0.2, obs, ret, act, val, nlp, w, torch.arange(n, device=dev, dtype=torch.int32), n, sync=False)
    torch.cuda.synchronize(); print('graph %s: first step (capture) %.1f ms' % (graph, (time.perf_counter() - t0) * 1e3), flush=True)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for ep in range(6):
            inds = torch.randperm(nb, device=dev).to(torch.int32)
            for start in range(0, nb, n):
                m.train_indexed(3e-4, 0.2, obs, ret, act, val, nlp, w, inds[start:start + n], n, sync=False)
        th = time.perf_counter() - t0
        torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print("graph %s: host %.1f ms, total %.1f ms for 192 steps (%.0f us/step)" % (graph, th * 1e3, tt * 1e3, tt / 192 * 1e6), flush=True)
