"""The numpy BPTT restatement (oracle/ppo_oracle.py::lstm_ppo_loss_and_grads) against central finite differences, so that the
GPU recurrent-training kernels are checked against something independently verified."""
import numpy as np

from oracle import ppo_oracle as po


def _case(seed=0, T=5, n=4, D=6, A=3, H=8):
    rng = np.random.default_rng(seed)
    params = [rng.normal(0, 0.4, s) for s in [(D, 4 * H), (H, 4 * H), (4 * H,), (H, A), (A,), (1, A), (H, 1), (1,)]]
    params[5] = rng.normal(-0.3, 0.1, (1, A))
    obs = rng.normal(0, 1, (T, n, D))
    masks = (rng.random((T, n)) < 0.25).astype(np.float64)
    actions = rng.normal(0, 0.7, (T, n, A))
    advs = rng.normal(0, 1, (T, n))
    returns = rng.normal(0, 1, (T, n))
    S0 = rng.normal(0, 0.5, (n, 2 * H))
    # old neglogp near the current one so ratios straddle the clip range
    loss0, _, _, _ = po.lstm_ppo_loss_and_grads(params, obs, masks, actions, advs, returns, np.zeros((T, n)), np.ones((T, n)), S0, 0.2, 0.01, 0.5)
    return params, obs, masks, actions, advs, returns, S0, rng


def test_bptt_gradients_match_finite_differences():
    params, obs, masks, actions, advs, returns, S0, rng = _case()
    T, n = masks.shape
    H = params[1].shape[0]
    # current neglogp, then perturb so some ratios are clipped and some are not
    lat_loss = lambda p, old: po.lstm_ppo_loss_and_grads(p, obs, masks, actions, advs, returns, old, w, S0, 0.2, 0.01, 0.5)
    w = rng.uniform(0.5, 1.5, (T, n))
    # neglogp under the current parameters (forward only): reuse the oracle's pieces
    c, h = S0[:, :H].copy(), S0[:, H:].copy()
    old = np.zeros((T, n))
    state = S0.copy()
    for t in range(T):
        hh, state = po.lstm_step_baselines(params[0].astype(np.float32), params[1].astype(np.float32), params[2].astype(np.float32),
                                           obs[t].astype(np.float32), state.astype(np.float32), masks[t].astype(np.float32))
        mean = hh @ params[3] + params[4]
        old[t] = po.neglogp(mean, params[5], actions[t])
    old = old + rng.normal(0, 0.25, (T, n))
    loss, stats, grads, _ = lat_loss(params, old)
    assert np.isfinite(loss) and 0.0 < stats[4] < 1.0          # both clipped and unclipped rows present
    eps = 1e-6
    for k, (p, g) in enumerate(zip(params, grads)):
        flat = p.ravel()
        for j in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            orig = flat[j]
            flat[j] = orig + eps; lp = lat_loss(params, old)[0]
            flat[j] = orig - eps; lm = lat_loss(params, old)[0]
            flat[j] = orig
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - np.ravel(g)[j]) < 1e-6 * (1 + abs(fd)), (k, j, fd, np.ravel(g)[j])


def test_forward_pieces_agree_with_step_function():
    """The unrolled forward inside the loss equals repeated lstm_step_baselines calls (final state)."""
    params, obs, masks, actions, advs, returns, S0, rng = _case(seed=3)
    T, n = masks.shape
    _, _, _, Sf = po.lstm_ppo_loss_and_grads(params, obs, masks, actions, advs, returns, np.zeros((T, n)), np.ones((T, n)), S0, 0.2, 0.0, 0.5)
    state = S0.astype(np.float32)
    for t in range(T):
        _, state = po.lstm_step_baselines(*(q.astype(np.float32) for q in params[:3]), obs[t].astype(np.float32), state, masks[t].astype(np.float32))
    assert np.abs(Sf - state).max() < 1e-5
