/* sumo_hip.h -- C ABI of the MI355X-native batched RoboSumo environment engine (libsumo_hip.so).
 *
 * This is the drop-in boundary for the vectorised env step.  The reference has no FFI for this path -- its
 * seam is the Python VecEnv interface -- so each entry point names the reference interface it stands in for:
 *
 *   sumo_create   load_model_from_path + MjSim construction for every worker process
 *                 (reference robosumo/robosumo/envs/mujoco_env.py:55-57, subproc_vec_env.py:49-56)
 *   sumo_reset    SubprocVecEnv.reset -> env.reset() in each worker
 *                 (reference subproc_vec_env.py:78-82, mujoco_env.py:104-108, sumo.py:232-253)
 *   sumo_step     SubprocVecEnv.step_async/step_wait -> worker 'step' incl. auto-reset
 *                 (reference subproc_vec_env.py:10-16,65-76; sumo_env.py:40-72; sumo.py:120-192;
 *                  mujoco_env.py:125-129 -> mj_step x frame_skip, mujoco-py/mujoco_py/mjsim.pyx:115-129)
 *   sumo_get_state / sumo_set_state   MjSim.get_state / set_state (mujoco-py/mujoco_py/mjsim.pyx:247-276)
 *
 * All array arguments of sumo_reset / sumo_step are DEVICE pointers owned by the caller (obs, rewards and
 * dones never leave HBM); sumo_get_state / sumo_set_state take HOST pointers and synchronise.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Every function returns 0 on success and a negative
 * code on error; sumo_last_error() then describes it.  A handle is bound to one GPU; calls on one handle are
 * not re-entrant.
 *
 * Shapes (E = num_envs, A = 2 agents, row-major):
 *   actions  float32 [E][A][act_stride]     raw policy outputs (clamped to ctrlrange inside, as MuJoCo does)
 *   obs      float32 [E][A][obs_stride]     agents.py:190-214 layout + time feature (sumo_env.py:68-70)
 *   info     float64 [E][A][8]              ctrl, lose, win, main, move, push, shaping, flags(bit0 winner, bit1 timeout,
 *                                           bit2 diverged: the state failed MuJoCo's bad-value test -- NaN or |x| > 1e10 in
 *                                           qpos/qvel/qacc, mujoco-py/mujoco_py/builder.py:351-369 raises there -- the step
 *                                           then reports zero rewards, done, and the env auto-resets)
 *   done     uint8   [E][A]
 *   ep_r, ep_dr float64 [E]; ep_l int32 [E] episode return / dense return / length of agent 0, valid where done
 */
#ifndef SUMO_HIP_H
#define SUMO_HIP_H
#include <stddef.h>
#include <stdint.h>
#include "sumo_ppo.h"   /* ppo_lstm_net (sumo_rollout_steps_lstm) */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sumo_engine* sumo_handle_t;

#define SUMO_INFO_STRIDE 8
#define SUMO_NDIMS 16 /* nq nv nu nbody njnt ngeom npair nagent obs_stride act_stride maxcon maxefc lds_bytes state_stride jbcap 0 */

const char* sumo_last_error(void);
int sumo_create(const void* model_blob, size_t nbytes, int num_envs, int device, sumo_handle_t* out);
int sumo_destroy(sumo_handle_t h);
int sumo_dims(sumo_handle_t h, int32_t* out /* [SUMO_NDIMS] */);
int sumo_reset(sumo_handle_t h, const uint64_t* seeds_host /* [E] or NULL */, const uint8_t* mask_dev /* [E] or NULL */,
               float* obs_dev, void* stream);
int sumo_step(sumo_handle_t h, const float* actions_dev, float* obs_dev, double* info_dev, uint8_t* done_dev,
              double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream);
/* K consecutive self-play rollout steps of every env of the engine in ONE launch: replaces the body of Runner.run's step loop
 * (runner.py:62-151: the five policy / value evaluations, env.step, the reward curriculum, the buffer appends) together with
 * the SubprocVecEnv round trip (subproc_vec_env.py:65-76) for MLP(64,64) policies (policies.py:14-128, baselines models.py:74-103,
 * distributions.py:227-251).  The wavefront that owns an env evaluates the learner's policy and value nets and the opponent's
 * policy net on the env's two observations on the matrix cores, samples both actions (action = mean + exp(logstd) * noise),
 * scores each with the other net, steps the env and appends to the rollout buffers; nothing is launched and no env waits for
 * another between two steps.  Results equal sumo_step + ppo_selfplay_forward + ppo_post_step called step by step, bit for bit.
 *   parameters: flat float32 vectors in the sumo_ppo.h layout; opponent_params holds npool frozen snapshots [npool][P] and
 *     opponent_index[e] (NULL = all 0) selects the snapshot env e plays against (the reference loads ONE snapshot for all envs
 *     per update, alg_ppo.py:213-214: npool = 1)
 *   noise0 / noise1 float32 [T][E][ac_dim]: standard-normal draws for the learner's (agent 0) / the opponent's (agent 1) actions
 *   rollout buffers, agent-major like Runner's mb_* lists: obs [2][T][Ntot][ob_dim], act [2][T][Ntot][ac_dim], rew / val / nlp
 *     (learner's neglogp) / onlp (opponent's neglogp) float32 [2][T][Ntot], done uint8 [2][T][Ntot] (flags BEFORE each step),
 *     ep_done uint8 / ep_r float64 / ep_l int32 [T][Ntot] (agent 0's episode records, monitor.py:63-78); this engine's envs are
 *     columns env_offset .. env_offset + E - 1; steps s0 .. s0 + K - 1 are written
 *   alpha: weight of the shaping reward (runner.py:130-134)
 * The env-side buffers are those of sumo_step (actions is written: it receives the sampled actions).  Launches of one engine must
 * be ordered on one stream (the engine owns the launch's ticket / progress counters); different engines may run concurrently.
 * After the launch obs / done hold the state after the last step as after sumo_step; info / ep_* hold the last step's values. */
typedef struct sumo_rollout {
  const float* learner_params;
  const float* opponent_params;
  const int32_t* opponent_index;
  int npool, ob_dim, ac_dim;
  int T, Ntot, env_offset, s0, K;
  double alpha;
  const float *noise0, *noise1;
  float *obs, *act, *rew, *val, *nlp, *onlp;
  uint8_t *done, *ep_done;
  double* ep_r;
  int32_t* ep_l;
} sumo_rollout;
int sumo_rollout_steps(sumo_handle_t h, const sumo_rollout* r, float* actions_dev, float* obs_dev, double* info_dev, uint8_t* done_dev,
                       double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream);

/* The same launch for RECURRENT policies (learn(network='lstm'): baselines lstm(128), value head on the same latent;
 * Runner's recurrent branch = reference runner.py:62-96 with the S / M feeds of models.py:163-170).  Per step and env the wave
 * evaluates: learner(obs 0, state0) -> action 0 / neglogp / value / new state0; opponent(obs 0, zero state) -> its likelihood of
 * action 0; opponent(obs 1, state1) -> action 1 / its neglogp / new state1; learner(obs 1, NEW state1) -> the value recorded for
 * agent 1; learner(obs 1, zero state) -> its likelihood of action 1.  State rows are zeroed where the previous step's done flag
 * is set before a cell runs.  Numbers equal the ppo_lstm_step launches of the step-by-step path bit for bit.
 *   learner        HOST struct (device pointers inside): hidden 128, gate order i,f,o,u, no embedding / observation filter
 *   opponents_dev  DEVICE array of npool structs of the same shape; tile_net_dev DEVICE int32 [Ntot / 16]: the snapshot every
 *                  16-env tile of the WHOLE env set faces (tile of env e of this engine: (env_offset + e) / 16), NULL = snapshot 0
 *   state0/state1  DEVICE float32 [E][256] (c | h): recurrent state of agent 0's / agent 1's acting net, rows of THIS engine's
 *                  envs; read at step s0, left at the state after step s0 + K - 1
 * Everything else as sumo_rollout.  Ordering contract as sumo_rollout_steps. */
typedef struct sumo_rollout_lstm {
  const ppo_lstm_net* learner;
  const ppo_lstm_net* opponents_dev;
  const int32_t* tile_net_dev;
  int npool;
  float *state0, *state1;
  int T, Ntot, env_offset, s0, K;
  double alpha;
  const float *noise0, *noise1;
  float *obs, *act, *rew, *val, *nlp, *onlp;
  uint8_t *done, *ep_done;
  double* ep_r;
  int32_t* ep_l;
} sumo_rollout_lstm;
int sumo_rollout_steps_lstm(sumo_handle_t h, const sumo_rollout_lstm* r, float* actions_dev, float* obs_dev, double* info_dev,
                            uint8_t* done_dev, double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream);
/* cfrc_mode (reference agents.py:190-214 reads sim.data.cfrc_ext into 84 of the 121 observation entries):
 *   0 = zero (default): what the reference produces -- its MuJoCo 2.1 scenes declare no force / torque / accelerometer sensor, so
 *       mj_rnePostConstraint never runs and cfrc_ext stays at its reset value 0;
 *   1 = rne_post: the entries as a MuJoCo 2.1 with such a sensor would fill them: |clip(cfrc_ext, +-100)| with cfrc_ext = per-body sum
 *       of the contact wrenches ([torque about the subtree CoM of the body's root ; force], world axes; -wrench on the contact's
 *       first body, + on its second) of the forward evaluation that OPENS the last mj_step of the env step (sensors are evaluated
 *       once per mj_step at its start state; the RK4 sub-stages skip them).  sumo_step then issues a second launch that re-derives
 *       that state from the pre-step state (frame_skip - 1 sub-steps), so a step costs ~1.85x; sumo_rollout_steps* refuse the mode
 *       (the policies read the observations inside their launch).  Envs whose episode ended in the step show the zeros of the reset
 *       observation.  Parity: against the oracle's restatement of mj_rnePostConstraint (unpinned: no MuJoCo here).
 * sumo_get_cfrc_ext: HOST float64 [E][nbody][6] of the last step (mode 1). */
int sumo_set_cfrc_mode(sumo_handle_t h, int mode);
/* Agent._adjust_z (reference robosumo/robosumo/envs/agents.py:33,155-161): a constant added to the torso height an agent REPORTS --
 * get_qpos() returns a copy with qpos[2] += _adjust_z, so it shifts the own-z entry (index 2) and the opponent-z entry (index
 * nq + nv + 6 nbody + 2) of every observation (agents.py:190-214) and both lose tests (sumo.py:147-160: z + adjust_z < 0.29); the
 * physics state, the xy used by the rewards and sumo_get_state are untouched.  0 (default) = training (run.py:76-77 leaves it
 * commented out); the reference's evaluation / play scripts set -0.5 on every agent (eval_robosumo_against_fix.py:108-115,
 * play_fixed.py:23, compare_history_version.py:74): the policy-zoo nets were trained with the tatami surface at z = 0, this fork's
 * is at z = 0.5.  Applies to sumo_reset / sumo_step / sumo_rollout_steps* alike from the next launch on (synchronises). */
int sumo_set_adjust_z(sumo_handle_t h, double adjust_z);
int sumo_get_cfrc_ext(sumo_handle_t h, double* out);
int sumo_get_state(sumo_handle_t h, double* qpos, double* qvel, double* warm, int32_t* counters /* [E][2] */);
int sumo_set_state(sumo_handle_t h, const double* qpos, const double* qvel, const double* warm,
                   const int32_t* counters);
/* debug / parity hook: run mj_forward once per env at its current state with ctrl (HOST float64 [E][nu]) and return
 * qacc (HOST float64 [E][nv]) plus per-env {ncon, nefc, newton iterations, dropped contacts} (HOST int32 [E][4]). */
int sumo_debug_forward(sumo_handle_t h, const double* ctrl, double* qacc, int32_t* counts);
/* Outcome of the engine's most recent sumo_rollout_steps* launch; waits for that launch (its stream) to finish.
 * A MuJoCo fault is loud in the reference (mujoco-py/mujoco_py/builder.py:351-369 raises MujocoException out of env.step); so is a
 * fused launch that did not complete: returns -20 (sumo_last_error() says why) when the launch's abort flag is set -- a wave's bounded
 * wait for its env's previous step expired, or the record a wave took over did not carry the sequence tag / checksum its writer
 * published (every hand-over is checked) -- and 0 otherwise.  After -20 the rollout buffers hold unwritten rows and the env states
 * are partly advanced: reset before continuing.  out4 (HOST, may be NULL): {abort flag, tickets drawn, tickets of the launch
 * (E x K), hand-over mismatches since creation}.  Runner.run (device mode) and bench.py call this after every rollout. */
int sumo_rollout_status(sumo_handle_t h, int64_t* out4);
/* 1 if this engine runs the static-Layout kernel variants: for the flagship scene (RoboSumo-Ant-vs-Ant-v0, default settings) the LDS
 * layout, the model's dimensions / table offsets and the derived-table offsets are compile-time constants of the per-step and the fused
 * rollout kernels (csrc/layout_static.h, generated by tools/gen_static_layout.py from the engine's own host code; sumo_create compares
 * the scene's runtime values with the tables word for word).  Results agree with the runtime-Layout variants every other scene uses
 * (SUMO_STATIC_LAYOUT=0 forces those) to float64 rounding -- a different compilation of the same source: literals change which multiply-add
 * pairs are contracted -- and the oracle parity tests run on the static variants; 0 otherwise. */
int sumo_static_layout(sumo_handle_t h);
/* development, host only (no device): the Layout / the model's and the derived tables' integer members the engine computes for a scene
 * (tools/gen_static_layout.py) */
int sumo_debug_layout(const void* model_blob, size_t nbytes, int32_t* out, int cap);
int sumo_debug_model_ints(const void* model_blob, size_t nbytes, int32_t* out, int cap, int32_t* aux_out, int aux_cap);
/* development (-DSUMO_DBG_DUMP builds): device buffer [20][8][64] float64 that receives intermediate vectors of env 0's forward
 * evaluations (tools/dump_diff.py); NULL = off */
int sumo_debug_dump(sumo_handle_t h, double* dev_buf);
/* development / tests: the first hand-over of env `env` in the following fused launches carries a wrong checksum (-1 = off) */
int sumo_debug_fault(sumo_handle_t h, int env);
/* device-side statistics accumulated since creation: forward calls, newton iterations, contacts, efc rows,
 * max ncon, max nefc, max newton iterations, dropped contacts, diverged env steps, aborted waits of the fused rollout's step
 * hand-over, hand-over tag / checksum mismatches (both always 0 unless a launch was cut short); contact-generation fidelity
 * accounting, SAMPLED in the forward evaluation that opens each env step (1 in 20 forwards): capsule-box calls that yielded 3 active contacts (MuJoCo's mjc_CapsuleBox yields at most 2), active contacts on a
 * border rod beyond the cylinder's flat end (the rods collide as capsules of the same radius / half length: the only place where
 * the shape differs from MuJoCo's cylinder) (HOST float64 [SUMO_NSTATS]). */
#define SUMO_NSTATS 13
int sumo_stats(sumo_handle_t h, double* out);
/* per-phase shader-cycle totals (20 phases + 4 ad-hoc probe slots); all zero unless the library was built with
 * -DSUMO_PROFILE (HOST float64 [24]). */
int sumo_profile(sumo_handle_t h, double* out24);

/* development: from the next sumo_step on, every env wave stores {start stamp, end stamp (100 MHz s_memrealtime),
 * Newton iterations | contacts << 32, dense-solver forwards | constraint rows << 32} of its step in stamps_dev (DEVICE
 * uint64 [E][4]); NULL switches the trace off.  Used by tools/slot_trace.py. */
int sumo_debug_trace(sumo_handle_t h, uint64_t* stamps_dev);

#ifdef __cplusplus
}
#endif
#endif
