/* sumo_ppo.h -- C ABI of the MI355X-native PPO2 self-play arithmetic (libsumo_ppo.so).
 *
 * Stands in for the TF1 graph + numpy glue of the reference's learner/rollout path.  Each entry point names the
 * reference code it replaces (paths relative to the reference checkout):
 *
 *   ppo_param_count / layout    PPOModel.save/load variable order (model.py:153-177):
 *                               pi/mlp_fc0/{w,b} pi/mlp_fc1/{w,b} vf/mlp_fc0/{w,b} vf/mlp_fc1/{w,b} pi/{w,b,logstd} vf/{w,b}
 *                               flattened row-major into ONE float32 vector, w stored [in][out] as TF does.
 *   ppo_forward                 PolicyWithValue.step / .value / .action_probability (policies.py:84-128) over the
 *                               mlp(64,64,relu) trunks (baselines/baselines/common/models.py:74-103) and the diagonal
 *                               Gaussian head (baselines/baselines/common/distributions.py:96-113,227-251)
 *   ppo_reward_mix              Runner.run reward curriculum (runner.py:127-143)
 *   ppo_vtrace                  Runner.run IS ratios + V-trace targets (runner.py:166-196)
 *   ppo_adv_moments/_normalize  PPOModel.train advantage normalisation (model.py:180-185); split in two so a
 *                               multi-GPU run can all-reduce the three moments in between
 *   ppo_grad                    loss + gradients of model.py:65-132 (sums, not means: divide by the global count)
 *   ppo_clip_adam               tf.clip_by_global_norm + tf.train.AdamOptimizer(epsilon=1e-5).apply_gradients
 *                               (model.py:121-139)
 *
 * All pointers are DEVICE pointers owned by the caller; `stream` is a hipStream_t as void*.  Return 0 / negative
 * + ppo_last_error().  float32 arithmetic on MFMA (v_mfma_f32_16x16x4_f32, exact f32) for the dense layers.
 */
#ifndef SUMO_PPO_H
#define SUMO_PPO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPO_HIDDEN 64
#define PPO_NSTATS 8 /* sum pg, sum vf, entropy, sum approxkl, sum clipfrac, sum w (unused), count, grad norm */

const char* ppo_last_error(void);
int ppo_param_count(int ob_dim, int ac_dim);

/* flags for ppo_forward */
#define PPO_FWD_PI 1        /* evaluate the policy trunk: mean / action / neglogp */
#define PPO_FWD_VF 2        /* evaluate the value trunk: value */
/* obs [n][obs_stride]; noise, given_action, action_out, mean_out [n][ac_dim] (any may be NULL);
 * action = given_action if given, else mean + exp(logstd)*noise if noise given, else mean (deterministic);
 * neglogp_out [n] = -log pi(action | obs); value_out [n]. */
int ppo_forward(const float* params, const float* obs, int n, int obs_stride, int ob_dim, int ac_dim, int flags,
                const float* noise, const float* given_action, float* action_out, float* neglogp_out,
                float* value_out, float* mean_out, void* stream);
/* Same, for the fixed opponents of the policy zoo (robosumo/robosumo/policy_zoo/policy.py:23-91, utils.py:9-33): tanh
 * hidden units (flag PPO_FWD_TANH) and the running-mean observation filter clip((obs - mean) * invstd, -clip, clip)
 * applied to the ob_dim columns read (obs_mean / obs_invstd float32 [ob_dim]; both NULL = no filter). */
#define PPO_FWD_TANH 4
int ppo_forward_filtered(const float* params, const float* obs, int n, int obs_stride, int ob_dim, int ac_dim, int flags,
                         const float* obs_mean, const float* obs_invstd, float obs_clip, const float* noise,
                         const float* given_action, float* action_out, float* neglogp_out, float* value_out,
                         float* mean_out, void* stream);

/* One time step of an LSTM policy for n rows (envs), state carried by the caller.  Covers the two recurrent nets of the
 * reference tree:
 *   - baselines `lstm(nlstm)` (baselines/baselines/common/models.py:131-183 + a2c/utils.py:82-103): x = obs, gates
 *     z = x*wx + h*wh + b split as (i, f, o, u), c,h first multiplied by (1 - mask), shared latent for both heads
 *     (policies.py:160-181);  gate_order = PPO_LSTM_GATES_IFOU, forget_bias = 0
 *   - policy-zoo `LSTMPolicy` (robosumo/robosumo/policy_zoo/policy.py:94-199): observation filter, one relu embedding
 *     layer, tf BasicLSTMCell (kernel [x|h] -> (i, j, f, o), forget bias 1);  gate_order = PPO_LSTM_GATES_IJFO
 * All weights are [in][out] row-major float32 device pointers.  emb_w == NULL: no embedding (x = filtered obs).
 * wh points at the recurrent rows (for a TF kernel: kernel + x_dim * 4 * hidden).  c / h are read and overwritten
 * ([n] rows of `state_stride` floats each); mask float32 [n] or NULL.  Heads (each optional): head_w [hidden][ac_dim],
 * head_b, logstd -> mean / action / neglogp exactly as ppo_forward; vf_w [hidden], vf_b -> value_out. */
#define PPO_LSTM_GATES_IFOU 0
#define PPO_LSTM_GATES_IJFO 1
typedef struct ppo_lstm_net {
  int ob_dim, emb_dim, hidden, ac_dim, gate_order;
  float forget_bias;
  const float *obs_mean, *obs_invstd; float obs_clip;     /* optional observation filter (both NULL = none) */
  const float *emb_w, *emb_b;                             /* optional embedding [ob_dim][emb_dim] + relu */
  const float *wx, *wh, *b;                               /* [x_dim][4*hidden], [hidden][4*hidden], [4*hidden] */
  const float *head_w, *head_b, *logstd;                  /* optional Gaussian head */
  const float *vf_w, *vf_b;                               /* optional value head */
} ppo_lstm_net;
int ppo_lstm_step(const ppo_lstm_net* net, const float* obs, int n, int obs_stride, const float* mask, float* c, float* h,
                  int state_stride, const float* noise, const float* given_action, float* action_out, float* neglogp_out,
                  float* value_out, float* mean_out, void* stream);

/* ppo_lstm_step against a POOL of frozen nets (BASELINE config 5: 16 opponent snapshots, one per env tile; the reference loads one
 * snapshot for all envs per update, alg_ppo.py:213-214): rows 16 t .. 16 t + 15 are evaluated with nets_dev[tile_net_dev[t]].
 * `proto` (host) gives the dimensions / gate order every net of the pool shares and is checked like ppo_lstm_step's net;
 * nets_dev is a DEVICE array of ppo_lstm_net whose pointers address each snapshot's weights, tile_net_dev DEVICE int32
 * [ceil(n / 16)].  Everything else as ppo_lstm_step (one launch for all tiles). */
int ppo_lstm_step_pool(const ppo_lstm_net* proto, const ppo_lstm_net* nets_dev, const int32_t* tile_net_dev, const float* obs, int n,
                       int obs_stride, const float* mask, float* c, float* h, int state_stride, const float* noise,
                       const float* given_action, float* action_out, float* neglogp_out, float* value_out, float* mean_out,
                       void* stream);

/* --- recurrent training (back-propagation through time over the baselines LSTM; the reference builds the unrolled graph in
 * a2c/utils.py:82-103 + model.py:65-139, its own recurrent minibatch loop alg_ppo.py:408-421 is dead code) ---------------
 * ppo_lstm_step_save: ppo_lstm_step that also records what the backward pass needs for this time step:
 *   save_gates [n][4*hidden] activated gates in the net's column order, save_cprev / save_hprev [n][hidden] the masked
 *   previous state, save_tanhc [n][hidden] tanh of the new cell state (the new h is the latent: read it from `h`). */
int ppo_lstm_step_save(const ppo_lstm_net* net, const float* obs, int n, int obs_stride, const float* mask, float* c, float* h,
                       int state_stride, float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc, void* stream);
/* The unrolled forward with the input block hoisted out of the recurrence: ppo_lstm_xproj computes z_out [rows][4*hidden] = x * wx
 * for all rows = (time, env) pairs of a minibatch in one launch (same tiles and k order as the step kernel's input block);
 * ppo_lstm_step_save_z is ppo_lstm_step_save for one time step whose gate sums start from z_t [n][4*hidden] (that step's rows of
 * z_out) instead of from the observations -- bit-identical to ppo_lstm_step_save on the same inputs, with half the dependent
 * chain per step -- and also writes the new latent to latent_out [n][hidden] (NULL = skip).  Nets without embedding /
 * observation filter only (what learn(network='lstm') trains). */
int ppo_lstm_xproj(const ppo_lstm_net* net, const float* obs, int rows, int obs_stride, float* z_out, void* stream);
int ppo_lstm_step_save_z(const ppo_lstm_net* net, const float* z_t, int n, const float* mask, float* c, float* h, int state_stride,
                         float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc, float* latent_out, void* stream);
/* All T steps in one launch each (hidden 128, gate order i,f,o,u): a workgroup of eight waves owns a 16-row tile of env sequences for
 * the whole sequence and keeps its slice of wh in registers; time-major buffers [T][n][...], mask [T][n] (NULL = none).
 *   ppo_lstm_seq_forward   z0 = ppo_lstm_xproj of all rows; c / h [n] (state_stride) hold the state at the start and receive the state
 *                          after step T-1; records as ppo_lstm_step_save per step + latent [T][n][128].  Same numbers as T calls of
 *                          ppo_lstm_step_save_z, bit for bit.
 *   ppo_lstm_seq_backward  BPTT from zero carries: dz_out [T][n][512] as T calls of ppo_lstm_bwd_step from t = T-1 down (the
 *                          contraction dz * wh^T is summed in one chain instead of four partial tiles: equal up to rounding). */
int ppo_lstm_seq_forward(const ppo_lstm_net* net, const float* z0, int T, int n, const float* mask, float* c, float* h, int state_stride,
                         float* save_gates, float* save_cprev, float* save_hprev, float* save_tanhc, float* latent, void* stream);
int ppo_lstm_seq_backward(const ppo_lstm_net* net, int T, int n, const float* dlatent, const float* mask, const float* gates,
                          const float* cprev, const float* tanhc, float* dz_out, void* stream);
/* PPO loss heads on stored latents (rows = all (time, env) pairs of the minibatch): forward of the Gaussian / value heads,
 * loss terms of model.py:65-111 and their gradients.  Outputs: dlatent [rows][hidden], dmean [rows][ac_dim], dvalue [rows],
 * dlogstd_rows [rows][ac_dim] (sum over rows - ent_coef = d loss / d logstd), stats double[PPO_NSTATS] (+= un-normalised sums
 * as ppo_grad; stats[2] is not touched).  inv_count = 1 / (global number of rows). */
int ppo_lstm_head_grad(const ppo_lstm_net* net, const float* latent, int rows, const float* actions, const float* adv,
                       const float* returns, const float* old_neglogp, const float* is_weight, double inv_count, float cliprange,
                       float vf_coef, float* dlatent, float* dmean, float* dvalue, float* dlogstd_rows, double* stats, void* stream);
/* One step of BPTT (gate order i,f,o,u): dh_carry / dc_carry [n][hidden] hold d loss / d (h_t, c_t) flowing from later steps and
 * are replaced by the values for step t-1 (already multiplied by 1 - mask_t); dz_out [n][4*hidden] receives d loss / d
 * (pre-activation gates), whose products with the inputs give the weight gradients (plain GEMMs, done by the caller). */
int ppo_lstm_bwd_step(const ppo_lstm_net* net, int n, const float* dlatent_t, const float* mask_t, const float* gates_t,
                      const float* cprev_t, const float* tanhc_t, float* dh_carry, float* dc_carry, float* dz_out, void* stream);

/* Weight gradients of the recurrent PPO step from the stored forward records and the deltas of the backward sweep (the matmuls
 * TF's autodiff emits for a2c/utils.py:82-103 + the heads of policies.py:50,70): over all `rows` = (time, env) pairs,
 *   [x | h_prev | 1]^T dz -> wx, wh, b      and      [latent | 1]^T [dmean | dvalue | dlogstd_rows] -> pi/w, pi/b, logstd, vf/w, vf/b
 * written into `grads` in checkpoint order (wx | wh | b | pi/w | pi/b | logstd | vf/w | vf/b); logstd_shift is added to the
 * logstd gradient (the entropy term - ent_coef / world).  x [rows][ob_dim], hprev / latent [rows][hidden], dz [rows][4 hidden],
 * dmean / dlogstd_rows [rows][ac_dim], dvalue [rows].  Split-K MFMA kernel + fixed-order slab reduction (deterministic).
 * workspace: ppo_lstm_wgrad_workspace_bytes(ob_dim, hidden, ac_dim) bytes. */
size_t ppo_lstm_wgrad_workspace_bytes(int ob_dim, int hidden, int ac_dim);
int ppo_lstm_wgrad(const ppo_lstm_net* net, int rows, const float* x, const float* hprev, const float* dz, const float* latent,
                   const float* dmean, const float* dvalue, const float* dlogstd_rows, float logstd_shift, float* grads, void* workspace,
                   void* stream);

/* info float64 [n][2][8] as written by sumo_step (slot 6 shaping, slot 3 main); reward_out float32 [2][n]
 * (agent-major, one time slice of the rollout buffer) = alpha*shaping + (1-alpha)*main evaluated in float64. */
int ppo_reward_mix(const double* info, int n, double alpha, float* reward_out, int agent_stride, void* stream);

/* ppo_reward_mix plus the step's episode records (agent 0's done flag, episode return and length as the monitor wrapper
 * reports them, bench/monitor.py:63-78) copied into one time slice of the rollout buffers: one launch per rollout step.
 * done uint8 [n][2]; ep_r float64 [n]; ep_l int32 [n]. */
int ppo_post_step(const double* info, int n, double alpha, float* reward_out, int agent_stride, const uint8_t* done,
                  const double* ep_r, const int32_t* ep_l, uint8_t* ep_done_out, double* ep_r_out, int32_t* ep_l_out, void* stream);

/* The five policy/value evaluations of one self-play rollout step (runner.py:62-96) in one launch, relu MLP nets.
 * obs: observation of env e, agent g at obs + e*env_stride + g*agent_stride (float32, ob_dim values); the learner acts on
 * agent 0 and the opponent on agent 1 (action = mean + exp(logstd) * noise_g[e]), each action is scored by the other net,
 * and the learner's value net is evaluated on both agents' observations.
 * act_env float32 [n][2][ac_dim] receives both actions (the env's action buffer).
 * out_f32[10] = { obs_out0, obs_out1 (float32 [n][ob_dim] copies of the inputs, may be NULL), act0, act1 ([n][ac_dim]),
 *                 nlp0, nlp1 (learner's neglogp of the action on side g), onlp0, onlp1 (opponent's), val0, val1 ([n]) }.
 * out_done (may be NULL) = { done_out0, done_out1 } uint8 [n] copies of done_in[e][g] (done_in uint8 [n][2], may be NULL).
 * Outputs equal those of the corresponding four ppo_forward calls bit for bit. */
int ppo_selfplay_forward(const float* learner_params, const float* opponent_params, const float* obs, int n, int env_stride,
                         int agent_stride, int ob_dim, int ac_dim, const float* noise0, const float* noise1,
                         const uint8_t* done_in, float* act_env, float* const* out_f32, uint8_t* const* out_done, void* stream);

/* Buffers [2][T][N] float32 (agent, time, env); dones uint8 [2][T][N] = done flags BEFORE each step; last_dones uint8
 * [N][2]; last_values float32 [2][N].  Outputs returns float32 [2][T][N], ratios float32 [T][N]. */
int ppo_vtrace(const float* rewards, const float* values, const float* neglogp, const float* opp_neglogp,
               const uint8_t* dones, const uint8_t* last_dones, const float* last_values, int T, int N, double gamma,
               double lam, double rho_bar, double c_bar, float* returns, float* off_policy_ratio, float* off_env_ratio,
               float* ratio, void* stream);

/* minibatch rows are data rows idx[0..n) (idx may be NULL = identity).  moments double[3] = {sum adv, sum adv^2, n}
 * with adv = returns - values.  Deterministic (fixed summation order), any n.  ppo_adv_moments_ws keeps its per-block partial sums
 * and arrival counter in the caller's workspace (ppo_adv_moments_workspace_bytes() bytes, zero-initialised once; one call in flight per
 * workspace): models stepping concurrently on different streams do not share state.  ppo_adv_moments is the convenience form on
 * library-owned workspaces handed out round robin (16; one device per process). */
size_t ppo_adv_moments_workspace_bytes(void);
int ppo_adv_moments_ws(const float* returns, const float* values, const int32_t* idx, int n, double* moments, void* workspace, void* stream);
int ppo_adv_moments(const float* returns, const float* values, const int32_t* idx, int n, double* moments, void* stream);
int ppo_adv_normalize(const float* returns, const float* values, const int32_t* idx, int n, const double* moments,
                      float* adv_out, void* stream);

/* Gradient of the PPO loss over minibatch rows idx[0..n): grads float32 [P] receives d(sum-loss)/d(theta) with
 * sum-loss = sum_i w_i*pg_i + vf_coef * sum_i 0.5 (v_i-R_i)^2 - n_local*ent_coef*entropy, where every per-row term
 * is pre-divided by inv_count = 1/global_count (so summing grads over ranks gives the gradient of the global mean loss).
 * stats double[PPO_NSTATS] accumulate the un-normalised sums.  log_ratio_out [n] (minibatch order) may be NULL.
 * workspace: ppo_grad_workspace_bytes(ob_dim, ac_dim) bytes of scratch, ZERO-INITIALISED once by the caller (its tail holds the arrival
 * counters of the in-launch slab reduction, which every call leaves at zero); one ppo_grad in flight per workspace. */
size_t ppo_grad_workspace_bytes(int ob_dim, int ac_dim);
int ppo_grad(const float* params, const float* obs, int obs_stride, int ob_dim, int ac_dim, const float* actions,
             const float* adv_mb, const float* returns, const float* old_neglogp, const float* is_weight,
             const int32_t* idx, int n, double inv_count, float cliprange, float ent_coef, float vf_coef, float* grads,
             double* stats, float* log_ratio_out, void* workspace, void* stream);

/* The statistics PPOModel.train returns (model.py:138,205-213) from the sums ppo_grad accumulated: out5 = {policy loss, value loss,
 * entropy, approxkl, clipfrac}; entropy = sum(logstd + 0.5 log(2 pi e)) of the policy the loss was evaluated with
 * (baselines distributions.py:246-247).  logstd float32 [ac_dim] (inside the flat parameter vector), stats double[PPO_NSTATS]. */
int ppo_loss_stats(const double* stats, const float* logstd, int ac_dim, double* out5, void* stream);

/* params -= Adam(clip_by_global_norm(grads, max_grad_norm)); m, v float32 [P]; step t >= 1 (TF1 bias correction).
 * max_grad_norm <= 0 disables clipping.  stats[7] receives the global gradient norm. */
int ppo_clip_adam(float* params, const float* grads, float* m, float* v, int P, int t, double lr, double beta1,
                  double beta2, double eps, double max_grad_norm, double* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif
