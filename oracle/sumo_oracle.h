/* sumo_oracle.h -- CPU float64 restatement of the RoboSumo env step (TEST INFRASTRUCTURE, not product).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED for the physics: the reference's arithmetic lives in the closed MuJoCo 2.1.0 binary
 * (libmujoco210.so, pinned by reference Dockerfile:21-30 / requirements.txt:89), which is absent from
 * /root/reference and from this image; the reference holds no trajectory fixtures.  This file restates
 * MuJoCo 2.1's published forward-dynamics pipeline for the RoboSumo scenes and
 * the reference's own game logic (which IS in the tree and is cited line by line).
 */
#ifndef SUMO_ORACLE_H
#define SUMO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct so_sim so_sim; /* model + N env states */

#define SO_INFO_STRIDE 8 /* ctrl, lose, win, main, move, push, shaping, flags(bit0 winner, bit1 timeout) */

so_sim* so_create(const void* model_blob, size_t nbytes, int num_envs);
void so_destroy(so_sim* s);
const char* so_last_error(void);
int so_dims(const so_sim* s, int* out10); /* nq nv nu nbody njnt ngeom npair nagent obs_stride act_stride */

/* reset every env (or those with mask[e]!=0 when mask!=NULL) from the counter RNG keyed by seeds[e] */
int so_reset(so_sim* s, const uint64_t* seeds, const uint8_t* mask, float* obs);
/* one vectorised env step with SubprocVecEnv auto-reset semantics (subproc_vec_env.py:10-16) */
int so_step(so_sim* s, const float* actions, float* obs, double* info, uint8_t* done, double* ep_r,
            double* ep_dr, int32_t* ep_l, int nthreads);
int so_get_state(const so_sim* s, double* qpos, double* qvel, double* warm, int32_t* counters);
int so_set_state(so_sim* s, const double* qpos, const double* qvel, const double* warm, const int32_t* counters);

/* --- white-box hooks for tests (single env `e`) --- */
int so_forward(so_sim* s, int e, const double* ctrl);       /* mj_forward at the env's current state */
int so_mj_step(so_sim* s, int e, const double* ctrl, int n); /* n x mj_step (RK4) */
int so_get_array(so_sim* s, int e, const char* name, double* out, int cap); /* returns count */
int so_stats(const so_sim* s, double* out11); /* forward calls, newton iters, contacts, efc rows (totals); max ncon, max nefc, max newton iters, dropped contacts, diverged env steps, capsule-box calls with 3 active contacts, rod contacts beyond the cylinder's flat end */
int so_set_maxcon(so_sim* s, int maxcon);
int so_set_jbcap(so_sim* s, int jbcap); /* capacity of the engine's Jacobian pool in halves (0 = unlimited) */
int so_set_seeds(so_sim* s, const uint64_t* seeds);
int so_set_cfrc_mode(so_sim* s, int mode);       /* 0 zero (reference), 1 rne_post */
int so_set_adjust_z(so_sim* s, double adjust_z); /* Agent._adjust_z, agents.py:33,155-161 (observed z and lose test) */

#ifdef __cplusplus
}
#endif
#endif
