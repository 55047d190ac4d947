"""Hyper-parameter defaults of the reference (defaults.py:5-84), RoboSumo / ppo branch only."""


def robosumo_ppo():
    return dict(nsteps=8192, nminibatches=32, lam=1.0, gamma=0.995, rho_bar=10.0, c_bar=1.0, noptepochs=6, lr=1e-3, cliprange=0.2,
                ent_coef=0.0, value_network="copy", anneal_bound=1000, num_hidden=64, activation="relu")   # defaults.py:8-26


def get_default_params(env_id, algo="ppo"):
    if algo != "ppo":
        raise NotImplementedError("only the ppo branch of defaults.py is on the hot path (SURVEY.md §2.1)")
    if not env_id.startswith("RoboSumo"):
        raise NotImplementedError("only RoboSumo envs")
    return robosumo_ppo()
