// sumo_engine.hip -- MI355X (gfx950) batched RoboSumo environment engine: kernels + C ABI (include/sumo_hip.h).
//
// One environment per 64-lane wavefront (one wave per workgroup).  The whole env step -- frame_skip x RK4 x
// forward dynamics (kinematics, CoM/CRB mass matrix, collision, pyramidal contact + joint-limit rows, bias forces,
// primal Newton solve), game rules, rewards, done, auto-reset with a counter RNG, observation write -- runs in ONE
// launch; per-env state (qpos, qvel, warm start) is read from / written to HBM once per env step as a contiguous
// record, everything else lives in LDS.  float64 throughout (the reference's mjtNum is double,
// mujoco-py/mujoco_py/pxd/mjmodel.pxd:81).
//
// Reference behaviour being reproduced (file:line are in the reference checkout):
//   env step + auto-reset     subproc_vec_env.py:10-16, sumo_env.py:40-72, robosumo/robosumo/envs/sumo.py:120-202
//   observation               robosumo/robosumo/envs/agents.py:190-214
//   reset                     robosumo/robosumo/envs/sumo.py:232-253, mujoco_env.py:104-119
//   physics                   mujoco_env.py:125-129 -> mjsim.pyx:115-129 -> mj_step (MuJoCo 2.1)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "../../include/sumo_hip.h"
#include "../../include/sumo_model.h"
#include "../../include/sumo_ppo.h"   /* ppo_lstm_net: the recurrent nets of the fused rollout */
#ifdef SUMO_POLICY_PROBE   /* development build: time inside the policy trunks, summed over all waves (sumo_debug_tprobe) */
__device__ unsigned long long g_tprobe[8];
#define PT_PROBE_INIT() unsigned long long tpl_ = __builtin_amdgcn_s_memrealtime()
#define PT_PROBE(k) do { if (lane == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_tprobe[k], t_ - tpl_); tpl_ = t_; } } while (0)
#endif
#include "ppo_tile.h"   /* MLP(64,64) trunk / Gaussian head on one MFMA tile: the policy phase of the fused rollout kernel */

#define WAVE 64
#ifndef SUMO_WPE
#define SUMO_WPE 2  /* waves per SIMD the register allocator targets: 2 -> at most 256 registers per lane */
#endif
/* Scenes whose LDS footprint leaves (almost) one wave per SIMD anyway -- nv >= 36: 29-38 KB per env, 4-5 waves per CU -- use
 * the whole 512-entry register file of the lane instead of spilling at 256: scratch 648-792 -> 8 B per lane, Spider-vs-Spider
 * 598 -> 858 k env-steps/s, Ant-vs-Spider 727 -> 946 k, Bug-vs-Bug 512 -> 554 k (measured; nv = 32 with 6 waves per CU is
 * faster at two waves per SIMD: 1.21 M against 1.02 M). */
#define SUMO_WPE_OF(nv) ((nv) >= 36 ? 1 : SUMO_WPE)
#define MINVAL 1e-15
#ifndef SUMO_STAT_LDS
#define SUMO_STAT_LDS 0
#endif
#define MAXCHAIN 8   /* dofs on the path root -> body (free joint 6 + hip + ankle) */
#define MAXBCHAIN 4  /* bodies on the path root -> body */
#define PI_D 3.14159265358979323846
#define RNG_NORMAL_BASE 64

// ---------------------------------------------------------------------------------------------------------
// device-side model view
// ---------------------------------------------------------------------------------------------------------
struct Aux {  // derived integer tables (built on the host in build_aux)
  const int* ai;
  const double* af;
  const signed char* pic;  // pos_in_chain[nbody][nv]: slot of dof in the body's chain, -1 if absent
  int ndepth;              // levels 0..ndepth-1 (world = level 0)
  int ntri;                // nv*(nv+1)/2
  int o_lvl_adr, o_lvl_body, o_child_adr, o_child, o_chain_len, o_chain, o_bchain_len, o_bchain, o_body_agent,
      o_tri_i, o_tri_j, o_ent;
  int o_wgmat;  // double table: 9 per geom (valid for world geoms)
  int o_stat_d, n_stat_d, o_stat_i, n_stat_i;  // tables copied into LDS at kernel start (see build_aux)
  int o_wpa;                                   // double table: world geom positions [3*nworld] | axes [3*nworld]
  int nc, nworld;  // collision centres: bodies 0..nbody-1, then one per world geom
  int maxchild;               // largest number of children of any body
  int nstub;                  // jointless leaf bodies whose inertia is merged into their (effective) parent
  int nwp, wrounds, arounds;  // pairs with a static world geom (listed first), rounds of 64 for them / for the rest
};

#define AUX_NINTS ((int)((offsetof(Aux, arounds) + sizeof(int) - offsetof(Aux, ndepth)) / sizeof(int)))   /* the int members: ndepth .. arounds */

struct Layout {  // LDS offsets in doubles unless noted
  int ld;        // leading dimension of H (odd)
  int mld, msize, d1;  // mass matrix: one dense block per agent tree, leading dim mld (odd); d1 = first dof of agent 1
  int maxcon, maxefc, maxcand;
  int jbcap;     // capacity of the contact-Jacobian pool in 24-double halves (3 rows x 8 slots); a contact between two
                 // moving bodies takes two halves, a contact with the world one
  int tree_ok;    // the dof tree has the shape tree_factor_solve assumes (checked in build_layout)
  int force_dense;  // development (SUMO_FORCE_DENSE_H=1): Newton Hessians always through the general dense path
  int warm_mode;  // 0: MuJoCo semantics (solver starts from qacc_warmstart of the previous mj_step); 1: RK stages 2-4 start
                  // from the previous stage's solution (same optimum within the solver tolerance, fewer Newton iterations)
  int qpos, qvel, warm, ctrl, x0, accv, acca, tmpv;
  int xpos, xquat, xipos, gaxis, xanchor, xaxis, com, cinert, cdof, abuf, cfrc;  // "kin scratch"
  int H;                                                                           // aliases kin scratch
  int M, bias, qsm, asmo, Ma, grad, search, Mv, x, dlim;
  int stash;     // 4 doubles parked across the step loop (+ 8 for the fused rollout: ticket, reward inputs, episode record, development clock)
  int cmask;     // per dof: 64-bit mask of the contacts whose Jacobian touches the dof
  int cond, Jb, cpar, cW, cp, jar, D, aref;
  int limD, limA;  // joint-limit slots, hinge-indexed: [nhinge] lower side | [nhinge] upper side (D = 0: side not active)
  int nhinge;
  int stat_d;    // static double tables: csize[2*nc] | cinvw[nc] | wbox[12*nworld] | wlim[6*nworld]
  int i_base;  // start of int region (in doubles)
  // int region offsets (in ints, relative to int base)
  int con_b;
  int b_dofidx;  // byte offset (relative to int base): signed char [16 * maxcon], dof of each Jacobian slot or -1
  int stat_i;    // static int tables: ctype[nc] | cbody[nc] | chain[2*nbody] | chlen_agent[nbody]
  int total_bytes;
};

// Compile-time copy of the Layout of the flagship scene (Ant-vs-Ant, default settings), generated by tools/gen_static_layout.py from
// build_layout() itself (sumo_debug_layout).  With it every `c.L.field` of the hot kernels is an immediate: LDS addresses become
// `lane-dependent base + constant offset` (one shift per lane index instead of an add per array), and the ~60 wave-uniform words of
// the runtime Layout no longer compete for the 102 SGPRs.  sumo_create compares the scene's runtime Layout with this table word
// for word and uses the static kernel variants only on an exact match (any other scene / setting: the runtime-Layout variants).
#include "layout_static.h"

struct StepArgs {
  double* state;       // [N][state_stride]
  int* counters;       // [N][4] num_steps, reset_count, -, -
  uint64_t* seeds;     // [N]
  int state_stride;
  const float* actions;
  float* obs;
  double* info;
  uint8_t* done;
  double* ep_r;
  double* ep_dr;
  int* ep_l;
  const uint8_t* mask;  // reset only
  const int* perm;      // step: workgroup b advances env perm[b] (NULL = identity); see sumo_step
  int* cost;            // step: per-env work estimate written for the next launch's schedule
  const int* rank_cost; // step: the first rank_blocks workgroups rank these estimates (of the previous launch) into
  int* rank_perm;       //       the schedule of the NEXT launch instead of advancing an env (sched_rank)
  int rank_blocks;
  unsigned long long* trace;   // development: [N][4] wall-clock (100 MHz) stamps of each env wave's start / end, work counters (sumo_debug_trace)
  unsigned long long* stats;
  int obs_stride, act_stride;
  int N;
  // fused rollout (sumo_rollout_kernel): hand-over tag of this launch (tag of an env after its k-th step = seq_base + k), the
  // launch's abort flag, development fault injection (env whose first hand-over carries a wrong checksum; -1 = off)
  int seq_base;
  int* abort_flag;
  int dbg_fault_env;
  // debug forward
  const double* dbg_ctrl;
  double* dbg_qacc;
  int* dbg_counts;
};

#define MI(name) (pt_global(mdl.ibase) + mdl.o_##name)
#define MF(name) (pt_global(mdl.fbase) + mdl.o_##name)
#define AI(name) (pt_global(aux.ai) + aux.o_##name)

/* One wavefront per workgroup: LDS operations of a wave are issued and completed in order, so phases only need a
 * compiler barrier between them (no s_waitcnt drain of unrelated loads, no s_barrier). */
#define SYNC() asm volatile("" ::: "memory")

// ---------------------------------------------------------------------------------------------------------
// small math
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fast_rcp(double x) {  // v_rcp_f64 + two Newton steps (~1 ulp)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {  // v_rsq_f64 + two Newton steps (~1 ulp)
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ double normalize3(double* v) {
  double s = dot3(v, v);
  if (s < MINVAL * MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; return sqrt(s); }
  double r = fast_rsqrt(s);
  v[0] *= r; v[1] *= r; v[2] *= r;
  return s * r;
}
__device__ __forceinline__ void normalize4(double* q) {
  double s = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s < MINVAL * MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double r = fast_rsqrt(s);
  q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}
__device__ __forceinline__ void mulquat(double* r, const double* a, const double* b) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
__device__ __forceinline__ void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
__device__ __forceinline__ void mulmatvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
__device__ __forceinline__ void mulmatTvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
// Library transcendentals used once per env step (reset draws, push reward) are called out of line: inlined, their
// coefficient tables are materialised in the kernel prologue and then live in registers / scratch through all twenty
// forward-dynamics evaluations.
__device__ __noinline__ double slow_sin(double x) { return sin(x); }
__device__ __noinline__ double slow_cos(double x) { return cos(x); }
__device__ __noinline__ double slow_log(double x) { return log(x); }
__device__ __noinline__ double slow_exp(double x) { return exp(x); }
// a floating-point constant the compiler cannot hoist (same reason)
#define OPAQUE_F64(name, val)                                                                                        \
  double name;                                                                                                       \
  {                                                                                                                  \
    int lo_, hi_;                                                                                                    \
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"                                                              \
                 : "=v"(lo_), "=v"(hi_)                                                                              \
                 : "i"((int)(__builtin_bit_cast(unsigned long long, (double)(val)) & 0xFFFFFFFFull)),                \
                   "i"((int)(__builtin_bit_cast(unsigned long long, (double)(val)) >> 32)));                         \
    name = __hiloint2double(hi_, lo_);                                                                               \
  }
// sin / cos to ~1 ulp for |x| < 1e6 (Cody-Waite reduction by pi/2 + the fdlibm kernel polynomials); the library routine,
// whose argument reduction dominates its cost, only for larger arguments
__device__ __forceinline__ void fast_sincos(double x, double* sn, double* cs) {
  if (fabs(x) > 1e6) { *sn = slow_sin(x); *cs = slow_cos(x); return; }   // (two-term reduction is exact below ~1.6e6)
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632673412561417e+00, x);
  r = fma(-fn, 6.07710050650619224932e-11, r);
  const double z = r * r;
  // (coefficients are rematerialised at each call: hoisted out of the step loop they would sit in registers / scratch for
  // the whole launch)
#define SC_K(name, val) OPAQUE_F64(name, val)
  SC_K(S1, -1.66666666666666324348e-01); SC_K(S2, 8.33333333332248946124e-03); SC_K(S3, -1.98412698298579493134e-04);
  SC_K(S4, 2.75573137070700676789e-06); SC_K(S5, -2.50507602534068634195e-08); SC_K(S6, 1.58969099521155010221e-10);
  SC_K(C1, 4.16666666666666019037e-02); SC_K(C2, -1.38888888888741095749e-03); SC_K(C3, 2.48015872894767294178e-05);
  SC_K(C4, -2.75573143513906633035e-07); SC_K(C5, 2.08757232129817482790e-09); SC_K(C6, -1.13596475577881948265e-11);
#undef SC_K
  const double ps = r + r * z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
  const double pc = 1.0 - 0.5 * z + z * z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const int q = (int)fn & 3;
  const double s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void axisangle2quat(double* q, const double* axis, double angle) {
  double s, cs;
  fast_sincos(angle * 0.5, &s, &cs);
  q[0] = cs; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
__device__ __forceinline__ double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
__device__ __forceinline__ void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ void cross_motion(double* r, const double* vel, const double* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
__device__ __forceinline__ void cross_force(double* r, const double* vel, const double* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}
// DPP move of a double (two 32-bit halves); lanes whose source is out of range / masked read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// wave-wide sum, identical bits in every lane (in-row scan with row_shr, then row_bcast 15 / 31, total in lane 63)
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov_f64<0x111, 0xF>(v);  // row_shr:1
  v += dpp_mov_f64<0x112, 0xF>(v);  // row_shr:2
  v += dpp_mov_f64<0x114, 0xF>(v);  // row_shr:4
  v += dpp_mov_f64<0x118, 0xF>(v);  // row_shr:8
  v += dpp_mov_f64<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v += dpp_mov_f64<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int* total) {
  int inc = v;
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xA, 0xF, false);
  inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xC, 0xF, false);
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
}

// Contact-generation fidelity accounting, once per forward evaluation and OUT OF LINE (its own register allocation; the hot loops
// of the collision phase are not touched): among the `ncon` contacts just stored (cond [14 per contact]: dist, pos[3], frame[9], -;
// cb [4 per contact]: body1, body2, pair, -) counts
//   * capsule-box calls that produced three active contacts -- three consecutive contacts of one pair (only capsule-box yields
//     three; MuJoCo's mjc_CapsuleBox at most two) -- low 16 bits;
//   * contacts on a border rod (a world CYLINDER, collided as a capsule) that lie beyond the cylinder's flat end, i.e. on the
//     capsule's end cap, the only place where the two shapes differ -- high 16 bits.
__device__ __noinline__ int fidelity_count(const double* cond, const int* cb, int ncon, int lane, const int32_t PT_GAS* pair_geom2,
                                           const int32_t PT_GAS* geom_type, const double PT_GAS* geom_pos, const double PT_GAS* geom_quat,
                                           const double PT_GAS* geom_size) {
  int cb3 = 0, rod = 0;
  if (lane < ncon) {
    const int p = cb[4 * lane + 2];
    if (lane + 2 < ncon && cb[4 * (lane + 1) + 2] == p && cb[4 * (lane + 2) + 2] == p) cb3 = 1;
    if (cb[4 * lane + 1] == 0) {   // the second body is the world
      const int g2 = pair_geom2[p];
      if (geom_type[g2] == SUMO_GEOM_CYLINDER) {
        const double w = geom_quat[4 * g2], x = geom_quat[4 * g2 + 1], y = geom_quat[4 * g2 + 2], z = geom_quat[4 * g2 + 3];
        const double ax[3] = {2 * (x * z + w * y), 2 * (y * z - w * x), w * w - x * x - y * y + z * z};   // the rod's axis: z column of its frame
        const double d = (cond[14 * lane + 1] - geom_pos[3 * g2]) * ax[0] + (cond[14 * lane + 2] - geom_pos[3 * g2 + 1]) * ax[1] +
                         (cond[14 * lane + 3] - geom_pos[3 * g2 + 2]) * ax[2];
        rod = fabs(d) > geom_size[3 * g2 + 1];
      }
    }
  }
  return __popcll(__ballot(cb3)) | (__popcll(__ballot(rod)) << 16);
}

// ---------------------------------------------------------------------------------------------------------
// narrow phase primitives.  A lane produces up to 3 contacts {dist, pos, normal} in registers.
// ---------------------------------------------------------------------------------------------------------
struct Con1 { double dist, pos[3], nrm[3]; int ok; };

__device__ __forceinline__ void sphere_sphere(Con1& c, double margin, const double* p1, double r1, const double* p2, double r2) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double cd2 = dot3(dif, dif), mind = margin + r1 + r2;
  c.ok = 0;
  if (cd2 > mind * mind) return;
  double len;
  if (cd2 < MINVAL * MINVAL) { len = sqrt(cd2); dif[0] = 0; dif[1] = 0; dif[2] = 1; }
  else { double ri = fast_rsqrt(cd2); len = cd2 * ri; dif[0] *= ri; dif[1] *= ri; dif[2] *= ri; }
  c.dist = len - r1 - r2;
  for (int k = 0; k < 3; k++) { c.nrm[k] = dif[k]; c.pos[k] = p1[k] + dif[k] * (r1 + 0.5 * c.dist); }
  c.ok = 1;
}
__device__ __forceinline__ void plane_sphere(Con1& c, double margin, const double* pp, const double* n, const double* sp, double r) {
  double t[3] = {sp[0] - pp[0], sp[1] - pp[1], sp[2] - pp[2]};
  double cd = dot3(t, n);
  c.ok = 0;
  if (cd > margin + r) return;
  c.dist = cd - r;
  for (int k = 0; k < 3; k++) { c.nrm[k] = n[k]; c.pos[k] = sp[k] - n[k] * (r + 0.5 * c.dist); }
  c.ok = 1;
}
__device__ __forceinline__ void sphere_box(Con1& c, double margin, const double* sp, double r, const double* bp, const double* bm,
                                           const double* bs) {
  double t[3] = {sp[0] - bp[0], sp[1] - bp[1], sp[2] - bp[2]}, ctr[3], cl[3], dl[3];
  mulmatTvec3(ctr, bm, t);
  for (int k = 0; k < 3; k++) { cl[k] = ctr[k] > bs[k] ? bs[k] : (ctr[k] < -bs[k] ? -bs[k] : ctr[k]); dl[k] = cl[k] - ctr[k]; }
  const double dd2 = dot3(dl, dl);
  const double idist = dd2 > 0 ? fast_rsqrt(dd2) : 0.0, dist = dd2 * idist;
  c.ok = 0;
  if (dist - r > margin) return;
  double nl[3], pl[3];
  if (dist <= MINVAL) {
    double best = 1e300; int kb = 0, sb = 1;
    for (int k = 0; k < 3; k++)
      for (int s = -1; s <= 1; s += 2) {
        double fd = fabs(s * bs[k] - ctr[k]);
        if (fd < best) { best = fd; kb = k; sb = s; }
      }
    nl[0] = nl[1] = nl[2] = 0;
    if (kb == 0) nl[0] = -sb; else if (kb == 1) nl[1] = -sb; else nl[2] = -sb;
    c.dist = -best - r;
  } else {
    for (int k = 0; k < 3; k++) nl[k] = dl[k] * idist;
    c.dist = dist - r;
  }
  for (int k = 0; k < 3; k++) pl[k] = ctr[k] + nl[k] * (r + 0.5 * c.dist);
  double pw[3];
  mulmatvec3(c.nrm, bm, nl);
  mulmatvec3(pw, bm, pl);
  for (int k = 0; k < 3; k++) c.pos[k] = pw[k] + bp[k];
  c.ok = 1;
}
// `skip`: coordinate that sits exactly on a box face at t (its own breakpoint) and contributes exactly 0
__device__ __forceinline__ double seg_box_dgrad(const double* cc, const double* a, const double* bs, double t, int skip = -1) {
  double g = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k == skip) continue;
    double p = cc[k] + t * a[k];
    if (p > bs[k]) g += a[k] * (p - bs[k]);
    else if (p < -bs[k]) g += a[k] * (p + bs[k]);
  }
  return g;
}

// ---------------------------------------------------------------------------------------------------------
// the per-wave environment context
// ---------------------------------------------------------------------------------------------------------
// Per-lane constants, loaded into registers once per launch (a lane keeps the same roles -- body `lane`, dof `lane`,
// joint `lane`, actuator `lane`, pair `lane + 64 r` -- for all 20 forward-dynamics evaluations of an env step).
struct LaneRec {
  // body role
  double b_pos[3], b_quat[4], b_ipos[3], b_iquat[4], b_inertia[3], b_mass;
  double j_pos[3], j_axis[3], j_qpos0;      // the body's hinge (bodies carry at most one joint)
  int b_parent, b_level, b_agent, b_isfree, b_jnt, b_qadr, b_nchild, b_bchain_len, b_chain_len;
  unsigned long long b_child;               // child body ids, one per byte, descending
  unsigned b_bchain;                        // body ids root -> self, one per byte
  unsigned long long b_chain;               // dof ids root -> body, one per byte
  unsigned b_chain_own, b_chain_free;       // bit p: chain[p] belongs to this body / starts a free joint
  // inertial constants used for cinert: the body's own, or -- when jointless leaf bodies are rigidly attached to it -- those of
  // the merged rigid body; body frame, inertia about the (merged) centre of mass as xx yy zz xy xz yz; zero for the leaves
  double c_I[6], c_ipos[3], c_mass;
  // dof role
  double d_arm, d_damp;
  int d_body, d_pos;                        // d_pos: position of the dof in d_chain
  unsigned long long d_chain;               // chain of the dof's body
  // joint role
  double jt_lo, jt_hi, jt_margin, jt_invw;
  int jt_type, jt_qadr, jt_dadr, jt_limited, jt_body, jt_agent;
  int jt_hid, d_hid;                        // running index of the hinge (joint role / dof role), -1 for free-joint dofs
  // actuator role
  double a_gear, a_lo, a_hi;
  int a_dof, pad_;
};

struct Params {  // lives in device memory; read through scalar / per-lane loads
  sumo_model_t mdl; Aux aux; Layout L;
  const LaneRec* lanes;     // [64]
  const int* pair_rec;      // [(wrounds + arounds) * 64] packed pair records (see build_aux)
  const float* pair_bound;  // conservative (rounded-up) reach of each pair
  double adjust_z;          // Agent._adjust_z (agents.py:33,155-161): added to the z an agent REPORTS -- observations and lose test; 0 in training
};

template <int NV_, class LT_ = Layout>
struct Ctx {
  typedef LT_ LT;
  static constexpr int NV = NV_;                                  // compile-time nv (register-resident factorisation)
  static constexpr int EPL = (NV_ * (NV_ + 1) / 2 + WAVE - 1) / WAVE;
  LT L;                 // LDS layout: the runtime Layout copied from Params once per launch (wave-uniform -> SGPRs), or the static table
  const LaneRec* kp;    // this lane's constant record (device memory, L1-resident); phases copy the fields they need
  const int* prp;       // this lane's packed pair records / bounds: element r*64
  const float* pbp;
  const Params* P;
  double* sm;    // LDS base (doubles)
  int* si;       // LDS int region
  unsigned char* sb;  // LDS byte region (slotof)
  int lane;
  int hid, dpos;   // this lane's dof: hinge index (-1: none) and position in its body's dof chain
  // per-forward scalars (wave-uniform)
  int ncon, nlim, nefc, ndropped, use_prev, htree;  // htree: every contact has one moving body -> H is tree-sparse like M
#ifdef SUMO_PROFILE
  long long tprev;
  unsigned long long prof[24];   // 20 phases + 4 ad-hoc probe slots (PROBE(k))
#endif
  // statistics accumulated over the launch
  int st_forward, st_newton, st_ncon, st_nefc, st_maxcon, st_maxefc, st_maxnewton, st_dropped, st_dense, st_cross;
  int st_diverged;   // env steps of this launch whose state failed MuJoCo's bad-value test (state_is_bad)
  // contact-generation fidelity accounting (deviations 1-2): capsule-box calls that produced 3 active contacts (MuJoCo's
  // mjc_CapsuleBox gives at most 2) and active contacts on a border rod that lie beyond the cylinder's flat end (the rods collide
  // as capsules: only there does the shape differ from MuJoCo's cylinder)
  int st_cb3, st_rodcap;
  int hcross;
#ifdef SUMO_DBG_DUMP
  double* dbg;   // development: intermediate vectors of env 0's first two forward evaluations (sumo_debug_dump)
#endif
};

// The model / aux views of a context: the runtime structs in device memory, or -- static-Layout variants -- objects whose integer
// members are compile-time constants of the flagship scene (layout_static.h) and whose three pointers are read from Params.
template <class C>
__device__ __forceinline__ decltype(auto) model_view(const C& c) {
  if constexpr (std::is_same<typename C::LT, Layout>::value) return (const sumo_model_t&)c.P->mdl;
  else { typename C::LT::Model m; m.ibase = c.P->mdl.ibase; m.fbase = c.P->mdl.fbase; m.tatami_size = c.P->mdl.tatami_size; return m; }
}
template <class C>
__device__ __forceinline__ decltype(auto) aux_view(const C& c) {
  if constexpr (std::is_same<typename C::LT, Layout>::value) return (const Aux&)c.P->aux;
  else { typename C::LT::AuxT x; x.ai = c.P->aux.ai; x.af = c.P->aux.af; x.pic = c.P->aux.pic; return x; }
}
#define S(off) (c.sm + c.L.off)
// Per-lane constants are re-read at the start of each phase instead of being pinned in registers for the whole launch
// (the optimisation barrier stops the compiler from hoisting the loads back to kernel entry): this keeps the kernel
// within 256 registers, i.e. two waves per SIMD.
template <class T>
__device__ __forceinline__ const T* launder_ptr(const T* p) { asm volatile("" : "+v"(p)); return p; }
template <class PTR>
__device__ __forceinline__ PTR launder_sptr(PTR p) { asm volatile("" : "+s"(p)); return p; }   // wave-uniform pointer (SGPR pair), any address space
// a record of a constant table, read through a global-address-space pointer dword by dword (only the fields that are used
// survive; a laundered pointer is a generic one, whose flat loads would occupy the LDS counter too -- see PT_GAS in ppo_tile.h)
template <class T>
__device__ __forceinline__ T load_rec(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  T out;
  const uint32_t PT_GAS* s_ = (const uint32_t PT_GAS*)p;
  uint32_t* d_ = (uint32_t*)&out;
#pragma unroll
  for (unsigned i_ = 0; i_ < sizeof(T) / 4; i_++) d_[i_] = s_[i_];
  return out;
}
#define KCONSTS() const LaneRec K = load_rec(launder_ptr(c.kp))
// mass matrix element (i, j) of the block-diagonal storage; valid when i and j belong to the same agent tree
#define MIDX(i, j) ((i) * c.L.mld + ((j) >= c.L.d1 ? (j) - c.L.d1 : (j)))
#define SAME_TREE(i, j) (((i) >= c.L.d1) == ((j) >= c.L.d1))
#define HP(i, k) ((i) * ((i) + 1) / 2 + (k))  /* packed lower-triangular index, i >= k */
#ifdef SUMO_PROFILE
#define PROF(k) do { long long _t = clock64(); c.prof[k] += (unsigned long long)(_t - c.tprev); c.tprev = _t; } while (0)
#else
#define PROF(k) do { } while (0)
#endif
#define PROBE(k) PROF(20 + (k))   /* development: split a phase; the time up to the probe goes to slot 20+k */

// ---- position / velocity stage --------------------------------------------------------------------------
#define BYTE_OF(word64, p) ((int)(((word64) >> (8 * (p))) & 0xFFull))

// One body's frame from its parent's (mj_kinematics, one tree level per call; lane == body id).  Only what the CHILDREN wait for is
// computed here: the joint's own rotation `ql` (a sin / cos pair that depends on qpos alone) comes from a pass over all bodies before
// the level loop, and the inertial-frame outputs (xipos, gaxis) are left to kin_inertial_frames after it -- the same operations on the
// same operands, but once on all body lanes instead of once per level on 2 - 8 of them.
template <class C>
__device__ __forceinline__ void kin_own_body(C& c, const double (&ql)[4]) {
  KCONSTS();
  const int b = c.lane, pid = K.b_parent;
  double* qpos = S(qpos);
  double xp[3], xq[4], R[9], v[3];
  if (K.b_isfree) {
    double* q = qpos + K.b_qadr;
    double qq[4] = {q[3], q[4], q[5], q[6]};
    normalize4(qq);
    q[3] = qq[0]; q[4] = qq[1]; q[5] = qq[2]; q[6] = qq[3];  // in-place normalisation, as mj_kinematics
    xp[0] = q[0]; xp[1] = q[1]; xp[2] = q[2];
    xq[0] = qq[0]; xq[1] = qq[1]; xq[2] = qq[2]; xq[3] = qq[3];
    for (int k = 0; k < 3; k++) S(xanchor)[3 * K.b_jnt + k] = xp[k];
  } else {
    quat2mat(R, S(xquat) + 4 * pid);
    mulmatvec3(v, R, K.b_pos);
    for (int k = 0; k < 3; k++) xp[k] = S(xpos)[3 * pid + k] + v[k];
    mulquat(xq, S(xquat) + 4 * pid, K.b_quat);
    if (K.b_jnt >= 0) {
      const int j = K.b_jnt;
      double anchor[3];
      quat2mat(R, xq);
      mulmatvec3(v, R, K.j_pos);
      for (int k = 0; k < 3; k++) { anchor[k] = xp[k] + v[k]; S(xanchor)[3 * j + k] = anchor[k]; }
      mulmatvec3(v, R, K.j_axis);
      for (int k = 0; k < 3; k++) S(xaxis)[3 * j + k] = v[k];
      mulquat(xq, xq, ql);
      quat2mat(R, xq);
      mulmatvec3(v, R, K.j_pos);
      for (int k = 0; k < 3; k++) xp[k] = anchor[k] - v[k];
    }
    normalize4(xq);
  }
  for (int k = 0; k < 3; k++) S(xpos)[3 * b + k] = xp[k];
  for (int k = 0; k < 4; k++) S(xquat)[4 * b + k] = xq[k];
}
template <class C>
__device__ __forceinline__ void kin_joint_rotation(C& c, double (&ql)[4]) {   // every body lane at once, before the level loop
  KCONSTS();
  ql[0] = 1.0; ql[1] = ql[2] = ql[3] = 0.0;
  if (c.lane >= 1 && c.lane < c.P->mdl.nbody && !K.b_isfree && K.b_jnt >= 0) axisangle2quat(ql, K.j_axis, S(qpos)[K.b_qadr] - K.j_qpos0);
}
template <class C>
__device__ __forceinline__ void kin_inertial_frames(C& c) {   // every body lane at once, after the level loop (reads its own frame back)
  KCONSTS();
  const int b = c.lane;
  double xp[3], xq[4], R[9], v[3], qi[4];
  for (int k = 0; k < 3; k++) xp[k] = S(xpos)[3 * b + k];
  for (int k = 0; k < 4; k++) xq[k] = S(xquat)[4 * b + k];
  quat2mat(R, xq);
  mulmatvec3(v, R, K.b_ipos);
  for (int k = 0; k < 3; k++) S(xipos)[3 * b + k] = xp[k] + v[k];
  mulquat(qi, xq, K.b_iquat);
  quat2mat(R, qi);
  S(gaxis)[3 * b] = R[2]; S(gaxis)[3 * b + 1] = R[5]; S(gaxis)[3 * b + 2] = R[8];
}

// gather children into parents, level by level, for an array of W doubles per body (lane == body id).  The sums are
// accumulated in registers with the loads of all (up to 8) children issued together: one LDS round trip per level instead
// of a read-modify-write chain per child.
template <int W, int NCH, class C>
__device__ __forceinline__ void gather_up_n(C& c, double* arr) {
  KCONSTS();
  const int b = c.lane;
  for (int lvl = c.P->aux.ndepth - 2; lvl >= 1; lvl--) {
    if (K.b_level == lvl && K.b_nchild > 0) {
      double acc[W], ch[NCH][W];
#pragma unroll
      for (int k = 0; k < W; k++) acc[k] = arr[W * b + k];
#pragma unroll
      for (int ci = 0; ci < NCH; ci++) {
        const int cb = BYTE_OF(K.b_child, ci < K.b_nchild ? ci : 0);
#pragma unroll
        for (int k = 0; k < W; k++) ch[ci][k] = arr[W * cb + k];
      }
#pragma unroll
      for (int ci = 0; ci < NCH; ci++)
        if (ci < K.b_nchild) {
#pragma unroll
          for (int k = 0; k < W; k++) acc[k] += ch[ci][k];
        }
#pragma unroll
      for (int k = 0; k < W; k++) arr[W * b + k] = acc[k];
    }
    SYNC();
  }
}

template <int W, class C>
__device__ __forceinline__ void gather_up(C& c, double* arr) {
  if (c.P->aux.maxchild <= 4) gather_up_n<W, 4>(c, arr);   // Ant torsos carry 4 legs, Bug 6, Spider 8
  else gather_up_n<W, 8>(c, arr);
}

template <class C>
__device__ __forceinline__ void position_velocity(C& c) {
  const auto& mdl = model_view(c);
  const auto& aux = aux_view(c);
  const int lane = c.lane;
  const int nb = mdl.nbody, nv = mdl.nv;
  if (lane == 0) {
    S(xpos)[0] = S(xpos)[1] = S(xpos)[2] = 0;
    S(xquat)[0] = 1; S(xquat)[1] = S(xquat)[2] = S(xquat)[3] = 0;
    for (int k = 0; k < 3; k++) { S(xipos)[k] = 0; S(gaxis)[k] = 0; }
  }
  SYNC();
  PROF(0);
  {
    const int my_level = pt_global(launder_ptr(c.kp))->b_level;
    double ql[4];
    kin_joint_rotation(c, ql);
    for (int lvl = 1; lvl < aux.ndepth; lvl++) {
      if (my_level == lvl) kin_own_body(c, ql);
      SYNC();
    }
    if (my_level >= 1 && my_level < aux.ndepth) kin_inertial_frames(c);
    SYNC();
  }
  PROF(1);
  // subtree CoM of each agent's root
  {
    KCONSTS();
    const bool isb = lane >= 1 && lane < nb;
    double px = 0, py = 0, pz = 0;
    if (isb) { px = K.b_mass * S(xipos)[3 * lane]; py = K.b_mass * S(xipos)[3 * lane + 1]; pz = K.b_mass * S(xipos)[3 * lane + 2]; }
    const int ag = isb ? K.b_agent : -1;
    // six independent wave sums (x, y, z of both agents) issued together so their DPP chains overlap
    const double s0x = wave_sum(ag == 0 ? px : 0.0), s0y = wave_sum(ag == 0 ? py : 0.0), s0z = wave_sum(ag == 0 ? pz : 0.0);
    const double s1x = wave_sum(ag == 1 ? px : 0.0), s1y = wave_sum(ag == 1 ? py : 0.0), s1z = wave_sum(ag == 1 ? pz : 0.0);
    if (lane < 2) {
      const double istm = fast_rcp(MF(body_subtreemass)[MI(agent_torso)[lane]]);
      S(com)[3 * lane] = (lane ? s1x : s0x) * istm; S(com)[3 * lane + 1] = (lane ? s1y : s0y) * istm; S(com)[3 * lane + 2] = (lane ? s1z : s0z) * istm;
    }
  }
  SYNC();
  // cinert per body, cdof per joint
  if (lane >= 1 && lane < nb) {
    KCONSTS();
    const int b = lane, ag = K.b_agent;
    double R[9], cp[3], dif[3], RI[9], T[6];
    quat2mat(R, S(xquat) + 4 * b);
    mulmatvec3(cp, R, K.c_ipos);
    for (int k = 0; k < 3; k++) dif[k] = S(xpos)[3 * b + k] + cp[k] - S(com)[3 * ag + k];
    // T = R I R^T with I = [xx xy xz; xy yy yz; xz yz zz] in the body frame
    const double Ixx = K.c_I[0], Iyy = K.c_I[1], Izz = K.c_I[2], Ixy = K.c_I[3], Ixz = K.c_I[4], Iyz = K.c_I[5];
    for (int r = 0; r < 3; r++) {
      RI[3 * r] = R[3 * r] * Ixx + R[3 * r + 1] * Ixy + R[3 * r + 2] * Ixz;
      RI[3 * r + 1] = R[3 * r] * Ixy + R[3 * r + 1] * Iyy + R[3 * r + 2] * Iyz;
      RI[3 * r + 2] = R[3 * r] * Ixz + R[3 * r + 1] * Iyz + R[3 * r + 2] * Izz;
    }
    T[0] = RI[0] * R[0] + RI[1] * R[1] + RI[2] * R[2];   // xx
    T[1] = RI[3] * R[3] + RI[4] * R[4] + RI[5] * R[5];   // yy
    T[2] = RI[6] * R[6] + RI[7] * R[7] + RI[8] * R[8];   // zz
    T[3] = RI[0] * R[3] + RI[1] * R[4] + RI[2] * R[5];   // xy
    T[4] = RI[0] * R[6] + RI[1] * R[7] + RI[2] * R[8];   // xz
    T[5] = RI[3] * R[6] + RI[4] * R[7] + RI[5] * R[8];   // yz
    double ms = K.c_mass;
    double* res = S(cinert) + 10 * b;
    res[0] = T[0] + ms * (dif[1] * dif[1] + dif[2] * dif[2]);
    res[1] = T[1] + ms * (dif[0] * dif[0] + dif[2] * dif[2]);
    res[2] = T[2] + ms * (dif[0] * dif[0] + dif[1] * dif[1]);
    res[3] = T[3] - ms * dif[0] * dif[1];
    res[4] = T[4] - ms * dif[0] * dif[2];
    res[5] = T[5] - ms * dif[1] * dif[2];
    res[6] = ms * dif[0]; res[7] = ms * dif[1]; res[8] = ms * dif[2];
    res[9] = ms;
  }
  if (lane < mdl.njnt) {
    KCONSTS();
    const int j = lane, b = K.jt_body, da = K.jt_dadr, ag = K.jt_agent;
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = S(com)[3 * ag + k] - S(xanchor)[3 * j + k];
    if (K.jt_type == SUMO_JNT_FREE) {
      double R[9];
      quat2mat(R, S(xquat) + 4 * b);
      for (int k = 0; k < 3; k++) {
        double* cd = S(cdof) + 6 * (da + k);
        cd[0] = cd[1] = cd[2] = cd[3] = cd[4] = cd[5] = 0;
        cd[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) {
        double ax[3] = {R[k], R[3 + k], R[6 + k]};
        double* cd = S(cdof) + 6 * (da + 3 + k);
        cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
        cross3(cd + 3, ax, off);
      }
    } else {
      double* cd = S(cdof) + 6 * da;
      const double* ax = S(xaxis) + 3 * j;
      cd[0] = ax[0]; cd[1] = ax[1]; cd[2] = ax[2];
      cross3(cd + 3, ax, off);
    }
  }
  SYNC();
  PROF(2);
  // body velocities along each body's dof chain; a_b = sum over the body's own dofs of cdof_dot * qvel
  double cvel[6] = {0, 0, 0, 0, 0, 0};
  if (c.L.tree_ok && lane >= 1 && lane < nb) {
    // every body hangs off a free joint (6 dofs) plus at most two hinges (Layout::tree_ok): fixed shape, all loads of the
    // chain's motion axes issued together instead of one round trip per chain element
    KCONSTS();
    const int b = lane, len = K.b_chain_len;
    const int B = BYTE_OF(K.b_chain, 0), j6 = BYTE_OF(K.b_chain, len > 6 ? 6 : 0), j7 = BYTE_OF(K.b_chain, len > 7 ? 7 : 0);
    const double* qvel = S(qvel);
    double cd[8][6], qv[8], a[6] = {0, 0, 0, 0, 0, 0}, t[6];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int j = k < 6 ? B + k : (k == 6 ? j6 : j7);
      qv[k] = qvel[j];
#pragma unroll
      for (int q = 0; q < 6; q++) cd[k][q] = S(cdof)[6 * j + q];
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
      for (int q = 0; q < 6; q++) cvel[q] += cd[k][q] * qv[k];
    if (K.b_chain_own & 1u) {
#pragma unroll
      for (int k = 3; k < 6; k++) {
        cross_motion(t, cvel, cd[k]);
#pragma unroll
        for (int q = 0; q < 6; q++) a[q] += t[q] * qv[k];
      }
    }
#pragma unroll
    for (int k = 3; k < 6; k++)
#pragma unroll
      for (int q = 0; q < 6; q++) cvel[q] += cd[k][q] * qv[k];
#pragma unroll
    for (int k = 6; k < 8; k++) {
      if (k < len) {
        if ((K.b_chain_own >> k) & 1u) {
          cross_motion(t, cvel, cd[k]);
#pragma unroll
          for (int q = 0; q < 6; q++) a[q] += t[q] * qv[k];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) cvel[q] += cd[k][q] * qv[k];
      }
    }
#pragma unroll
    for (int q = 0; q < 6; q++) S(abuf)[6 * b + q] = a[q];
  } else if (lane >= 1 && lane < nb) {
    KCONSTS();
    const int b = lane;
    double a[6] = {0, 0, 0, 0, 0, 0}, cd[6];
    const double* qvel = S(qvel);
    const int len = K.b_chain_len;
    int p = 0;
    while (p < len) {
      const int j = BYTE_OF(K.b_chain, p);
      const bool own = (K.b_chain_own >> p) & 1u;
      if ((K.b_chain_free >> p) & 1u) {
        for (int k = 0; k < 3; k++)
          for (int q = 0; q < 6; q++) cvel[q] += S(cdof)[6 * (j + k) + q] * qvel[j + k];
        if (own)
          for (int k = 3; k < 6; k++) {
            cross_motion(cd, cvel, S(cdof) + 6 * (j + k));
            for (int q = 0; q < 6; q++) a[q] += cd[q] * qvel[j + k];
          }
        for (int k = 3; k < 6; k++)
          for (int q = 0; q < 6; q++) cvel[q] += S(cdof)[6 * (j + k) + q] * qvel[j + k];
        p += 6;
      } else {
        if (own) {
          cross_motion(cd, cvel, S(cdof) + 6 * j);
          for (int q = 0; q < 6; q++) a[q] += cd[q] * qvel[j];
        }
        for (int q = 0; q < 6; q++) cvel[q] += S(cdof)[6 * j + q] * qvel[j];
        p += 1;
      }
    }
    for (int q = 0; q < 6; q++) S(abuf)[6 * b + q] = a[q];
  }
  SYNC();
  // RNE forward: cacc along the body chain, cfrc = I*cacc + cvel x* (I*cvel)
  if (lane < nb) {
    const int b = lane;
    double* f = S(cfrc) + 6 * b;
    if (b == 0) { for (int q = 0; q < 6; q++) f[q] = 0; }
    else {
      KCONSTS();
      const double* g = MF(opt) + SUMO_OPT_GRAVITY;
      double cacc[6] = {0, 0, 0, -g[0], -g[1], -g[2]}, t[6], t1[6];
      double ab[MAXBCHAIN][6];
#pragma unroll
      for (int p = 0; p < MAXBCHAIN; p++) {   // all loads in flight; entries past the chain re-read its first body
        const int kb = (K.b_bchain >> (8 * (p < K.b_bchain_len ? p : 0))) & 0xFFu;
#pragma unroll
        for (int q = 0; q < 6; q++) ab[p][q] = S(abuf)[6 * kb + q];
      }
#pragma unroll
      for (int p = 0; p < MAXBCHAIN; p++)
        if (p < K.b_bchain_len) {
#pragma unroll
          for (int q = 0; q < 6; q++) cacc[q] += ab[p][q];
        }
      mul_inert_vec(f, S(cinert) + 10 * b, cacc);
      mul_inert_vec(t, S(cinert) + 10 * b, cvel);
      cross_force(t1, cvel, t);
      for (int q = 0; q < 6; q++) f[q] += t1[q];
    }
  }
  SYNC();
  gather_up<6>(c, S(cfrc));
  if (lane < nv) S(bias)[lane] = dot6(S(cdof) + 6 * lane, S(cfrc) + 6 * pt_global(launder_ptr(c.kp))->d_body);
  PROF(3);
}

template <class C>
__device__ __forceinline__ void mass_matrix(C& c) {
  KCONSTS();
  const int lane = c.lane, nv = c.P->mdl.nv;
  // the contact records (dead by now) share the mass matrix's storage: clear it only here
  for (int i = lane; i < c.L.msize; i += WAVE) S(M)[i] = 0.0;
  gather_up<10>(c, S(cinert));  // cinert -> composite rigid body inertia, in place (ends with a SYNC)
  if (c.L.tree_ok && lane < nv) {
    // ancestors of dof i in the fixed shape: the root dofs below it, the leg's hip (for an ankle), itself
    const int i = lane, B = i >= c.L.d1 ? c.L.d1 : 0, il = i - B, nr = il < 6 ? il : 6;
    double ci[6], buf[6], r[7][6];
#pragma unroll
    for (int q = 0; q < 6; q++) ci[q] = S(cdof)[6 * i + q];
#pragma unroll
    for (int k = 0; k < 7; k++) {
      const int j = k < 6 ? B + k : i - 1;            // k == 6: the hip of an ankle dof (unused otherwise)
#pragma unroll
      for (int q = 0; q < 6; q++) r[k][q] = S(cdof)[6 * (j < 0 ? 0 : j) + q];
    }
    mul_inert_vec(buf, S(cinert) + 10 * K.d_body, ci);
    double* Mi = S(M) + i * c.L.mld;
    Mi[il] = dot6(ci, buf) + K.d_arm;
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < nr) { const double v = dot6(r[k], buf); Mi[k] = v; S(M)[(B + k) * c.L.mld + il] = v; }
    if (il >= 6 && (il & 1)) { const double v = dot6(r[6], buf); Mi[il - 1] = v; S(M)[(i - 1) * c.L.mld + il] = v; }
  } else if (lane < nv) {
    const int i = lane;
    double buf[6];
    mul_inert_vec(buf, S(cinert) + 10 * K.d_body, S(cdof) + 6 * i);
    for (int p = K.d_pos; p >= 0; p--) {  // ancestors of dof i = its body's chain up to the dof itself
      const int j = BYTE_OF(K.d_chain, p);
      double v = dot6(S(cdof) + 6 * j, buf);
      if (j == i) S(M)[MIDX(i, i)] = v + K.d_arm;
      else { S(M)[MIDX(i, j)] = v; S(M)[MIDX(j, i)] = v; }
    }
  }
  SYNC();
}

// ---- collision ---------------------------------------------------------------------------------------------
// "centre" index of a geom: its body id for agent geoms (one geom per moving body), nbody + w for world geom w.
// Positions / axes of all centres live in the xipos / gaxis arrays (world entries are written once per launch).
// The static collision tables (centre types / bodies / sizes, body invweights, world-geom frames and extents, dof chains)
// stay in device memory: a few hundred bytes that every wave of the CU reads (L1-resident).  SUMO_STAT_LDS=1 keeps them in
// LDS instead (2.2 KB per env for the Ant scene, which costs the seventh wave per CU).
#if SUMO_STAT_LDS
#define STAT_I(k) (c.si[c.L.stat_i + (k)])
#define STAT_D(k) (c.sm + c.L.stat_d + (k))
#else
#define STAT_I(k) (pt_global(c.P->aux.ai)[c.P->aux.o_stat_i + (k)])
#define STAT_D(k) (pt_global(c.P->aux.af) + c.P->aux.o_stat_d + (k))
#endif
#define CTYPE(ci) STAT_I(ci)
#define CBODY(ci) STAT_I(c.P->aux.nc + (ci))
#define CSIZE(ci) STAT_D(2 * (ci))
#define CINVW(ci) (STAT_D(2 * c.P->aux.nc + (ci))[0])
#define WBOX(w) STAT_D(3 * c.P->aux.nc + 12 * (w))
#define WLIM(w) STAT_D(3 * c.P->aux.nc + 12 * c.P->aux.nworld + 6 * (w))
#define CHAINW(b) (&STAT_I(2 * c.P->aux.nc + 2 * (b)))
#define CHLEN_AGENT(b) STAT_I(2 * c.P->aux.nc + 2 * c.P->mdl.nbody + (b))

__device__ __forceinline__ void make_frame(double* f) {
  if (f[3] * f[3] + f[4] * f[4] + f[5] * f[5] < 0.25) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double dd = dot3(f, f + 3);
  f[3] -= f[0] * dd; f[4] -= f[1] * dd; f[5] -= f[2] * dd;
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

template <class C>
__device__ __forceinline__ void collision(C& c) {
  const int lane = c.lane, nb = c.P->mdl.nbody;
  // survivors of the broad phase (pair order preserved): pair ids and packed centre records (c1 | c2 << 8).  The queue
  // borrows the constraint-row arrays (jar / aref), which are dead until make_constraint.
  int* plist = (int*)S(jar);
  int* prlist = (int*)S(aref);
  int ncon = 0, dropped = 0;
  const int WR = c.P->aux.wrounds, AR = c.P->aux.arounds, nwp = c.P->aux.nwp, maxcand = c.L.maxcand;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // The pair table lists the pairs with a static world geom first (WR rounds of 64, padded), then the pairs between
  // moving geoms (AR rounds).  Rounds are tested four at a time, loads first, so their latencies overlap; a candidate
  // with queue position in [base, base + maxcand) is stored, and the (rare) overflow is handled by running the tests
  // again for the next window.
  int base = 0, total = 0;
  do {
    int run = 0;
    const int PT_GAS* prp = pt_global(launder_ptr(c.prp));
    const float PT_GAS* pbp = pt_global(launder_ptr(c.pbp));
    // ---- world pairs: distance from the moving geom's centre to the static geom's extent (box / segment / half space)
    for (int r0 = 0; r0 < WR; r0 += 4) {
      int rec[4], pass[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + q;
        rec[q] = 0; pass[q] = 0;
        if (r < WR) {
          rec[q] = prp[WAVE * r];
          const double bound = (double)pbp[WAVE * r];
          const int ca = rec[q] & 0xFF, cw = (rec[q] >> 8) & 0xFF, w = cw >= nb ? cw - nb : 0;
          const double* pa = S(xipos) + 3 * ca;
          const double* pw = S(xipos) + 3 * cw;
          const double* bm = WBOX(w);
          const double* wl = WLIM(w);
          const double t[3] = {pa[0] - pw[0], pa[1] - pw[1], pa[2] - pw[2]};
          double ctr[3], d2 = 0;
          mulmatTvec3(ctr, bm, t);
#pragma unroll
          for (int k = 0; k < 3; k++) { double e = fmax(fmax(ctr[k] - wl[k], wl[3 + k] - ctr[k]), 0.0); d2 += e * e; }
          pass[q] = ((rec[q] >> 17) & 1) && !(d2 > bound * bound);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + q;
        if (r < WR) {
          const unsigned long long bal = __ballot(pass[q]);
          const int pos = run + __popcll(bal & lt_mask) - base;
          if (pass[q] && pos >= 0 && pos < maxcand) {
            const int ca = rec[q] & 0xFF, cw = (rec[q] >> 8) & 0xFF;
            plist[pos] = lane + WAVE * r;
            prlist[pos] = (rec[q] & (1 << 16)) ? (cw | (ca << 8)) : (ca | (cw << 8));   // bit 16: the world geom is geom1
          }
          run += __popcll(bal);
        }
      }
    }
    // ---- pairs of moving geoms: bounding spheres
    prp += WAVE * WR; pbp += WAVE * WR;
    for (int r0 = 0; r0 < AR; r0 += 4) {
      int rec[4], pass[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + q;
        rec[q] = 0; pass[q] = 0;
        if (r < AR) {
          rec[q] = prp[WAVE * r];
          const double bound = (double)pbp[WAVE * r];
          const double* p1 = S(xipos) + 3 * (rec[q] & 0xFF);
          const double* p2 = S(xipos) + 3 * ((rec[q] >> 8) & 0xFF);
          const double t[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
          pass[q] = ((rec[q] >> 17) & 1) && !(dot3(t, t) > bound * bound);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = r0 + q;
        if (r < AR) {
          const unsigned long long bal = __ballot(pass[q]);
          const int pos = run + __popcll(bal & lt_mask) - base;
          if (pass[q] && pos >= 0 && pos < maxcand) { plist[pos] = nwp + lane + WAVE * r; prlist[pos] = rec[q] & 0xFFFF; }
          run += __popcll(bal);
        }
      }
    }
    total = run;
    const int ncand = total - base < maxcand ? total - base : maxcand;
    SYNC();
    PROF(4);
    for (int k0 = 0; k0 < ncand; k0 += WAVE) {
      int k = k0 + lane;
      Con1 cs[3];
      cs[0].ok = cs[1].ok = cs[2].ok = 0;
      int p = 0, b1 = 0, b2 = 0;
      double margin = 0;
      if (k < ncand) {
        p = plist[k];
        const int rc = prlist[k];
        const int c1 = rc & 0xFF, c2 = (rc >> 8) & 0xFF;
        const int t1 = CTYPE(c1), t2 = CTYPE(c2);  // cylinders already mapped to capsules (border rods)
        b1 = CBODY(c1); b2 = CBODY(c2);
        margin = c.P->mdl.fbase[c.P->mdl.o_pair_margin + p];
        const double* s1 = CSIZE(c1);
        const double* s2 = CSIZE(c2);
        const double* p1 = S(xipos) + 3 * c1;
        const double* p2 = S(xipos) + 3 * c2;
        const double* g1 = S(gaxis) + 3 * c1;
        const double* g2 = S(gaxis) + 3 * c2;
        double a1[3], a2[3];
        if (t1 == SUMO_GEOM_PLANE) {
          if (t2 == SUMO_GEOM_SPHERE) plane_sphere(cs[0], margin, p1, g1, p2, s2[0]);
          else if (t2 == SUMO_GEOM_CAPSULE) {
            double e[3];
            for (int q = 0; q < 3; q++) e[q] = p2[q] + g2[q] * s2[1];
            plane_sphere(cs[0], margin, p1, g1, e, s2[0]);
            for (int q = 0; q < 3; q++) e[q] = p2[q] - g2[q] * s2[1];
            plane_sphere(cs[1], margin, p1, g1, e, s2[0]);
          }
        } else if (t1 == SUMO_GEOM_SPHERE && t2 == SUMO_GEOM_SPHERE) {
          sphere_sphere(cs[0], margin, p1, s1[0], p2, s2[0]);
        } else if (t1 == SUMO_GEOM_SPHERE && t2 == SUMO_GEOM_CAPSULE) {
          double v[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          double x = dot3(g2, v);
          if (x > s2[1]) x = s2[1];
          if (x < -s2[1]) x = -s2[1];
          double e[3] = {p2[0] + g2[0] * x, p2[1] + g2[1] * x, p2[2] + g2[2] * x};
          sphere_sphere(cs[0], margin, p1, s1[0], e, s2[0]);
        } else if (t1 == SUMO_GEOM_CAPSULE && t2 == SUMO_GEOM_CAPSULE) {
          for (int q = 0; q < 3; q++) { a1[q] = g1[q] * s1[1]; a2[q] = g2[q] * s2[1]; }
          double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
          double det = ma * mc - mb * mb, v1[3], v2[3];
          if (fabs(det) >= MINVAL) {
            const double idet = fast_rcp(det), imc = fast_rcp(mc), ima = fast_rcp(ma);
            double x1 = (mc * u - mb * v) * idet, x2 = (ma * v - mb * u) * idet;
            if (x1 > 1) { x1 = 1; x2 = (v - mb) * imc; }
            else if (x1 < -1) { x1 = -1; x2 = (v + mb) * imc; }
            if (x2 > 1) { x2 = 1; x1 = (u - mb) * ima; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
            else if (x2 < -1) { x2 = -1; x1 = (u + mb) * ima; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
            for (int q = 0; q < 3; q++) { v1[q] = p1[q] + a1[q] * x1; v2[q] = p2[q] + a2[q] * x2; }
            sphere_sphere(cs[0], margin, v1, s1[0], v2, s2[0]);
          } else {
            for (int si = 0; si < 2; si++) {
              double sg = si == 0 ? 1.0 : -1.0;
              double x2 = (v - sg * mb) / mc;
              if (x2 > 1) x2 = 1; else if (x2 < -1) x2 = -1;
              for (int q = 0; q < 3; q++) { v1[q] = p1[q] + sg * a1[q]; v2[q] = p2[q] + a2[q] * x2; }
              if (si == 0) sphere_sphere(cs[0], margin, v1, s1[0], v2, s2[0]);
              else sphere_sphere(cs[1], margin, v1, s1[0], v2, s2[0]);
            }
          }
        } else if (t2 == SUMO_GEOM_BOX) {
          const double* bm = WBOX(c2 - nb);
          const double* bs = bm + 9;
          if (t1 == SUMO_GEOM_SPHERE) sphere_box(cs[0], margin, p1, s1[0], p2, bm, bs);
          else if (t1 == SUMO_GEOM_CAPSULE) {
            double axw[3] = {g1[0] * s1[1], g1[1] * s1[1], g1[2] * s1[1]}, e[3];
            for (int q = 0; q < 3; q++) e[q] = p1[q] + axw[q];
            sphere_box(cs[0], margin, e, s1[0], p2, bm, bs);
            for (int q = 0; q < 3; q++) e[q] = p1[q] - axw[q];
            sphere_box(cs[1], margin, e, s1[0], p2, bm, bs);
            double t[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]}, cc[3], a[3];
            mulmatTvec3(cc, bm, t);
            mulmatTvec3(a, bm, axw);
            double glo = seg_box_dgrad(cc, a, bs, -1.0), ghi = seg_box_dgrad(cc, a, bs, 1.0);
            if (glo < 0 && ghi > 0) {
              // the derivative is piecewise linear and non-decreasing in t: bracket the root between consecutive
              // breakpoints (where a coordinate crosses a box face), then interpolate exactly
              double lo = -1, hi = 1;
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const double ra = fast_rcp(a[k]);
#pragma unroll
                for (int sg = 0; sg < 2; sg++) {
                  const double tb = ((sg ? -bs[k] : bs[k]) - cc[k]) * ra;
                  if (tb > lo && tb < hi) {
                    const double gb = seg_box_dgrad(cc, a, bs, tb, k);
                    if (gb > 0) { hi = tb; ghi = gb; } else { lo = tb; glo = gb; }
                  }
                }
              }
              double ts = lo - glo * (hi - lo) * fast_rcp(ghi - glo);
              for (int q = 0; q < 3; q++) e[q] = p1[q] + ts * axw[q];
              sphere_box(cs[2], margin, e, s1[0], p2, bm, bs);
            }
          }
        }
      }
      int act[3], n = 0;
#pragma unroll
      for (int q = 0; q < 3; q++) { act[q] = cs[q].ok && (cs[q].dist < margin); n += act[q]; }
      int ctot, cbase = wave_excl_scan(n, lane, &ctot);
      int slot = ncon + cbase;
#pragma unroll
      for (int q = 0; q < 3; q++) {
        if (act[q]) {
          if (slot < c.L.maxcon) {
            double* cd = S(cond) + 14 * slot;
            double fr[9] = {cs[q].nrm[0], cs[q].nrm[1], cs[q].nrm[2], 0, 0, 0, 0, 0, 0};
            make_frame(fr);
            cd[0] = cs[q].dist; cd[1] = cs[q].pos[0]; cd[2] = cs[q].pos[1]; cd[3] = cs[q].pos[2];
            for (int w = 0; w < 9; w++) cd[4 + w] = fr[w];
            cd[13] = 0;
            int* cb = c.si + c.L.con_b + 4 * slot;
            cb[0] = b1; cb[1] = b2; cb[2] = p; cb[3] = 0;
          }
          slot++;
        }
      }
      ncon += ctot;
    }
    SYNC();
    PROF(5);
    base += maxcand;
  } while (base < total);
  if (ncon > c.L.maxcon) { dropped = ncon - c.L.maxcon; ncon = c.L.maxcon; }
  c.ncon = ncon;
  c.ndropped = dropped;
  SYNC();
#ifndef SUMO_NO_FIDELITY
  // contact-generation fidelity accounting (Ctx::st_cb3), out of line (the narrow-phase loop above stays as it was) and SAMPLED: the
  // forward evaluation that opens an env step, 1 in frame_skip x 4 RK stages = 20 (every forward: +0.9 % on the bench; sampled: nil)
  if ((c.st_forward - 1) % 20 == 0) {
    const auto& mdl = model_view(c);
    const int r = fidelity_count(S(cond), c.si + c.L.con_b, ncon, lane, MI(pair_geom2), MI(geom_type), MF(geom_pos), MF(geom_quat), MF(geom_size));
    c.st_cb3 += r & 0xFFFF;
    c.st_rodcap += r >> 16;
  }
#endif
}

// ---- constraint rows -----------------------------------------------------------------------------------------
__device__ __noinline__ double impedance_general_pow(double xx, double mid, double power) {  // rare: power not in {1, 2}
  if (xx <= mid) return pow(xx, power) / pow(mid, power - 1);
  return 1 - pow(1 - xx, power) / pow(1 - mid, power - 1);
}
__device__ __forceinline__ double impedance(const double* solimp, double x) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = fmin(fmax(dmin, 0.0001), 0.9999);
  dmax = fmin(fmax(dmax, 0.0001), 0.9999);
  if (width < MINVAL) width = MINVAL;
  mid = fmin(fmax(mid, 0.0001), 0.9999);
  if (power < 1) power = 1;
  double xx = fabs(x) * fast_rcp(width), y;
  if (xx >= 1) return dmax;
  if (xx <= 0) return dmin;
  if (power == 1) y = xx;
  else if (power == 2) y = xx <= mid ? xx * xx * fast_rcp(mid) : 1 - (1 - xx) * (1 - xx) * fast_rcp(1 - mid);  // MuJoCo default solimp
  else y = impedance_general_pow(xx, mid, power);
  return dmin + y * (dmax - dmin);
}
// returns R; writes B and K*imp*(pos-margin)
__device__ __forceinline__ double row_params(double timestep, const double* solref, const double* solimp, double pos, double margin,
                                             double diag, double* B, double* kterm) {
  double tc = solref[0], dr = solref[1], dmax = fmin(fmax(solimp[1], 0.0001), 0.9999);
  if (tc < 2 * timestep) tc = 2 * timestep;
  double imp = impedance(solimp, pos - margin);
  double kk = dmax * dmax * tc * tc * dr * dr, bb = dmax * tc;
  double K = fast_rcp(kk < MINVAL ? MINVAL : kk);
  *B = 2.0 * fast_rcp(bb < MINVAL ? MINVAL : bb);
  *kterm = K * imp * (pos - margin);
  double R = (1 - imp) * diag * fast_rcp(imp);
  return R < MINVAL ? MINVAL : R;
}

template <class C>
__device__ __forceinline__ void make_constraint(C& c) {
  const auto& mdl = model_view(c);
  KCONSTS();
  const int lane = c.lane, nv = mdl.nv;
  // ---- Jacobian pool allocation (contact order): one 3x8 half per moving body of the contact
  int ncon = c.ncon;
  {
    int nh = 0;
    int* cb = c.si + c.L.con_b + 4 * (lane < ncon ? lane : 0);
    if (lane < ncon) nh = (cb[0] != 0 && cb[1] != 0) ? 2 : 1;
    int tot, pre = wave_excl_scan(nh, lane, &tot);
    unsigned long long nofit = __ballot(lane < ncon && pre + nh > c.L.jbcap);
    if (nofit) { int nfit = __builtin_ctzll(nofit); c.ndropped += ncon - nfit; ncon = nfit; c.ncon = nfit; }
    if (lane < ncon) cb[3] = (24 * pre) | ((nh == 2) << 20);
    SYNC();
  }
  const double timestep = MF(opt)[SUMO_OPT_TIMESTEP];
  OPAQUE_F64(sr0_, 0.02); OPAQUE_F64(si0_, 0.9); OPAQUE_F64(si1_, 0.95); OPAQUE_F64(si2_, 0.001);   // MuJoCo's default solref / solimp
  const double def_solref[2] = {sr0_, 1.0};
  const double def_solimp[5] = {si0_, si1_, si2_, 0.5, 2.0};
  // joint limits; lane == joint id.  A hinge owns two fixed slots indexed by its dof (lower side, upper side): D = 1/R and
  // aref, with D = 0 when the side is not active (every term of the solver carries D, so absent rows cost nothing).
  // The Jacobian of a limit row is +-e_dof: the dof's own lane evaluates its rows without touching LDS row arrays.
  int act_lo = 0, act_hi = 0;
  if (lane < nv) ((unsigned long long*)S(cmask))[lane] = 0ull;
  if (lane < mdl.njnt && K.jt_type == SUMO_JNT_HINGE) {
    const int dof = K.jt_dadr;
    const double jm = K.jt_margin, value = S(qpos)[K.jt_qadr], vel = S(qvel)[dof];
    const double dlo = value - K.jt_lo, dhi = K.jt_hi - value;
    act_lo = K.jt_limited && dlo < jm; act_hi = K.jt_limited && dhi < jm;
    double D0 = 0, A0 = 0, D1 = 0, A1 = 0;
    if (act_lo) {
      double B, kt, R = row_params(timestep, def_solref, def_solimp, dlo, jm, K.jt_invw, &B, &kt);
      D0 = fast_rcp(R); A0 = -B * vel - kt;
    }
    if (act_hi) {
      double B, kt, R = row_params(timestep, def_solref, def_solimp, dhi, jm, K.jt_invw, &B, &kt);
      D1 = fast_rcp(R); A1 = B * vel - kt;     // row = -e_dof: efc_vel = -qvel
    }
    const int hid = K.jt_hid, nh = c.L.nhinge;
    S(limD)[hid] = D0; S(limD)[nh + hid] = D1; S(limA)[hid] = A0; S(limA)[nh + hid] = A1;
  }
  const int nlim = __popcll(__ballot(act_lo)) + __popcll(__ballot(act_hi));
  SYNC();
  c.nlim = nlim;
  c.nefc = nlim + 4 * ncon;
  {
    int two = 0, cross = 0;
    const int b1 = MI(agent_bodyadr)[1];
    for (int ci = lane; ci < ncon; ci += WAVE) {
      const int* cb = c.si + c.L.con_b + 4 * ci;
      if (cb[0] != 0 && cb[1] != 0) { two = 1; if ((cb[0] >= b1) != (cb[1] >= b1)) cross = 1; }
    }
    c.htree = c.L.tree_ok && !c.L.force_dense && __ballot(two) == 0ull;
    c.hcross = __ballot(cross) != 0ull;
  }
  // contact row parameters (same for the 4 pyramid edges of a contact)
  for (int ci = lane; ci < ncon; ci += WAVE) {
    const double* cd = S(cond) + 14 * ci;
    const int* cb = c.si + c.L.con_b + 4 * ci;
    int p = cb[2];
    double mu = MF(pair_friction)[3 * p];
    double pm = MF(pair_margin)[p], pg = MF(pair_gap)[p];
    double sr[2] = {MF(pair_solref)[2 * p], MF(pair_solref)[2 * p + 1]};
    double si_[5];
    for (int q = 0; q < 5; q++) si_[q] = MF(pair_solimp)[5 * p + q];
    double tran = CINVW(cb[0]) + CINVW(cb[1]);   // body ids double as centre ids (world body 0 -> 0)
    double diag = tran + mu * mu * tran, B, kt;
    double R = row_params(timestep, sr, si_, cd[0], pm - pg, diag, &B, &kt);
    double Rpy = 2 * mu * mu * R;
    if (Rpy < MINVAL) Rpy = MINVAL;
    S(cpar)[ci] = mu;
    S(D)[ci] = fast_rcp(Rpy);   // one D per contact (its four pyramid rows share it)
    for (int k = 0; k < 4; k++) { int r = 4 * ci + k; S(aref)[r] = B; S(jar)[r] = kt; }
  }
  PROF(6);
  // compact contact Jacobians: 3 base rows (normal, t1, t2) x ns slots.  Two moving bodies: ns = 16 (chain of body1,
  // sign -, then chain of body2, sign +); one moving body (contact with the world): ns = 8, that body's chain only.
  for (int idx = lane; idx < ncon * 16; idx += WAVE) {
    int ci = idx >> 4, s = idx & 15;
    const int* cb = c.si + c.L.con_b + 4 * ci;
    const int two = cb[3] >> 20, jb = cb[3] & 0xFFFFF, ns = two ? 16 : 8;
    int side, pos = s & 7;
    bool slot_ok = true;
    if (two) side = s >> 3;
    else { side = cb[0] != 0 ? 0 : 1; slot_ok = s < 8; }
    int body = side ? cb[1] : cb[0], other = side ? cb[0] : cb[1];
    int dof = -1, ag = 0;
    double j0 = 0, j1 = 0, j2 = 0;
    if (slot_ok && body != 0) {
      int la = CHLEN_AGENT(body);
      ag = la >> 8;
      if (pos < (la & 0xFF)) {
        const int* cw = CHAINW(body);
        dof = ((unsigned)cw[pos >> 2] >> (8 * (pos & 3))) & 0xFF;
        // a dof shared by both bodies' root paths sits at the same chain position: its contributions cancel
        if (other != 0 && pos < (CHLEN_AGENT(other) & 0xFF)) {
          const int* ow = CHAINW(other);
          if ((int)(((unsigned)ow[pos >> 2] >> (8 * (pos & 3))) & 0xFF) == dof) dof = -1;
        }
      }
    }
    if (dof >= 0) {
      const double* cd = S(cond) + 14 * ci;
      const double* cdf = S(cdof) + 6 * dof;
      double off[3] = {cd[1] - S(com)[3 * ag], cd[2] - S(com)[3 * ag + 1], cd[3] - S(com)[3 * ag + 2]}, t[3];
      cross3(t, cdf, off);
      t[0] += cdf[3]; t[1] += cdf[4]; t[2] += cdf[5];
      double sg = side ? 1.0 : -1.0;
      j0 = sg * dot3(cd + 4, t); j1 = sg * dot3(cd + 7, t); j2 = sg * dot3(cd + 10, t);
      atomicOr((unsigned long long*)S(cmask) + dof, 1ull << ci);
    }
    ((signed char*)c.sb)[c.L.b_dofidx + idx] = (signed char)dof;
    if (slot_ok) {
      double* Jb = S(Jb) + jb;
      Jb[s] = j0; Jb[ns + s] = j1; Jb[2 * ns + s] = j2;
    }
  }
  SYNC();
  PROF(7);
}

// Jacobian slot of dof `d` in contact `ci`: slots are positions in the dof chain of the contact's body (second body: + 8);
// `p` is the dof's position in its own body's chain, which is also its position in the chain of any body below it
#define SLOT_OF(ci, d, p) ((((const signed char*)c.sb)[c.L.b_dofidx + 16 * (ci) + (p)] == (signed char)(d)) ? (p) : (p) + 8)
// cp[3*c + a] = sum_s Jb[c][a][s] * x[dof(c,s)].  Branch-free and fully unrolled so all 16 slot loads are in flight at
// once: Jb is exactly 0 in unused slots, so those may read any x.
template <class C>
__device__ __forceinline__ void contact_Jx(C& c, const double* x) {
  for (int idx = c.lane; idx < 3 * c.ncon; idx += WAVE) {
    int ci = idx / 3, a = idx - 3 * ci;
    const int info = (c.si + c.L.con_b)[4 * ci + 3], ns = (info >> 20) ? 16 : 8;
    const double* Jb = S(Jb) + (info & 0xFFFFF) + ns * a;
    const signed char* di = (const signed char*)c.sb + c.L.b_dofidx + 16 * ci;
    double xv[16], jv[16];
#pragma unroll
    for (int s = 0; s < 16; s++) { int d = di[s]; xv[s] = x[d < 0 ? 0 : d]; double jq = Jb[s]; jv[s] = s < ns ? jq : 0.0; }
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int s = 0; s < 16; s += 4) { a0 += jv[s] * xv[s]; a1 += jv[s + 1] * xv[s + 1]; a2 += jv[s + 2] * xv[s + 2]; a3 += jv[s + 3] * xv[s + 3]; }
    S(cp)[idx] = (a0 + a1) + (a2 + a3);
  }
  SYNC();
}
// value of contact row r (= 4 * contact + pyramid edge) of J*x given cp (after contact_Jx)
template <class C>
__device__ __forceinline__ double row_Jx(const C& c, int r) {
  const int ci = r >> 2, k = r & 3;
  double mu = c.sm[c.L.cpar + ci];
  const double* cp = c.sm + c.L.cp + 3 * ci;
  return cp[0] + ((k & 1) ? -mu : mu) * cp[1 + (k >> 1)];
}
// this lane's two joint-limit slots (lower, upper): D (0 = absent) and aref
struct LimRows { double D0, D1, A0, A1; };
template <class C>
__device__ __forceinline__ LimRows lim_rows(const C& c) {
  const int nh = c.L.nhinge, h = c.hid < 0 ? 0 : c.hid;
  LimRows q;
  q.D0 = c.sm[c.L.limD + h]; q.D1 = c.sm[c.L.limD + nh + h]; q.A0 = c.sm[c.L.limA + h]; q.A1 = c.sm[c.L.limA + nh + h];
  if (c.hid < 0) q.D0 = q.D1 = 0.0;     // free-joint dofs and lanes past nv carry no limit
  return q;
}

// y_i = sum_k M[i][k] x[k]   (x in LDS); M is block diagonal, so only the lane's own tree contributes
template <class C>
__device__ __forceinline__ double dense_Mx(const C& c, const double* x) {
  constexpr int NV = C::NV;
  double acc = 0;
  if (c.L.d1 * 2 == NV) {  // two equal trees: fixed trip count, all loads issued up front
    const int li = c.lane < NV ? c.lane : 0;
    const int k0 = li >= NV / 2 ? NV / 2 : 0;
    const double* row = c.sm + c.L.M + li * c.L.mld;
    double a0 = 0, a1 = 0;
#pragma unroll
    for (int k = 0; k < NV / 2; k += 2) { a0 += row[k] * x[k0 + k]; a1 += row[k + 1] * x[k0 + k + 1]; }
    acc = c.lane < NV ? a0 + a1 : 0.0;
  } else if (c.lane < c.P->mdl.nv) {
    const int d1 = c.L.d1, nv = c.P->mdl.nv;
    const int k0 = c.lane >= d1 ? d1 : 0, k1 = c.lane >= d1 ? nv : d1;
    const double* row = c.sm + c.L.M + c.lane * c.L.mld;
    for (int k = k0; k < k1; k++) acc += row[k - k0] * x[k];
  }
  return acc;
}

// Solve A x = b for a symmetric positive definite A held in LDS (packed lower triangle): lane i loads row i
// into registers, the wave runs a right-looking LDL^T entirely with v_readlane broadcasts (no LDS traffic, no barriers;
// fully unrolled so every register index is static), forward-substitutes, transposes L through LDS once (`T`, packed)
// and back-substitutes.  Lane i holds b_i on entry and returns x_i.  *fail is wave-uniform.
// This is the general path (a contact between two moving bodies fills H outside the tree pattern).
template <class C>
__device__ __forceinline__ double ldl_solve_rows(C& c, const double* A, double* T, double b, int* fail) {
  constexpr int NV = C::NV;
  const int lane = c.lane;
  const int li = lane < NV ? lane : NV - 1;
  double a[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) a[k] = A[HP(li, (k <= li ? k : li))];
  double dinv = 0;
  int bad = 0;
#pragma unroll
  for (int j = 0; j < NV; j++) {
    double ajj = readlane_f64(a[j], j);
    if (ajj < MINVAL) bad = 1;
    double r = fast_rcp(ajj);
    double lij = a[j] * r;
#pragma unroll
    for (int k = j + 1; k < NV; k++) a[k] -= lij * readlane_f64(a[j], k);  // lanes < k only touch unused entries
    if (lane == j) dinv = r;
    a[j] = lij;
  }
  double y = b;
#pragma unroll
  for (int k = 0; k < NV - 1; k++) {
    double yk = readlane_f64(y, k);
    if (lane > k) y -= a[k] * yk;
  }
  y *= dinv;
  // write L back into the packed triangle (row `lane`), then every lane reads its COLUMN from there (back substitution)
  {
    const int rbase = HP(li, 0);
#pragma unroll
    for (int k = 0; k < NV - 1; k++)
      if (lane > k && lane < NV) T[rbase + k] = a[k];
  }
  SYNC();
  double t[NV];
#pragma unroll
  for (int k = 1; k < NV; k++) t[k] = T[HP(k, li)];   // L[k][lane]; entries with k <= lane are never used
#pragma unroll
  for (int k = NV - 1; k > 0; k--) {
    double xk = readlane_f64(y, k);
    if (lane < k) y -= t[k] * xk;
  }
  SYNC();
  *fail = bad;
  return y;
}

// ---- tree-sparse factorisation ----------------------------------------------------------------------------------
// Every agent of the nine scenes is a free body (6 dofs) carrying legs of two hinges (hip, ankle), with the dofs ordered
// root 0..5, then (hip, ankle) per leg (checked on the host, Layout::tree_ok).  The mass matrix -- and the Newton
// Hessian as long as every contact touches a single moving body -- is then nonzero only between a dof and its
// ancestors: lane i keeps row i as row[p] = A[i][ancestor at chain position p] (root dof m: p <= m; hip: 0..5 root,
// 6 diag; ankle: 0..5 root, 6 hip, 7 diag).  A x = b is solved by block elimination (MuJoCo's mj_factorM order, leaves
// first): the 2x2 leg blocks are inverted in their own lanes, the 6x6 Schur complement of each root is accumulated by
// the six root lanes and factorised redundantly by every lane of the agent.  No fill-in, five short phases.
#define DPP_SWAP_PAIR 0xB1  /* quad_perm [1,0,3,2]: exchange with the neighbouring lane (hip <-> ankle; hips sit on even lanes) */
__device__ __forceinline__ double pair_swap_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, DPP_SWAP_PAIR, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, DPP_SWAP_PAIR, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// row <- tree row `lane` of M + diag_add on the diagonal (+ J^T W J of the contacts touching this dof)
template <class C>
__device__ __forceinline__ void tree_rows(C& c, double (&row)[8], double diag_add, bool with_contacts) {
  const int lane = c.lane, nv = c.P->mdl.nv, d1 = c.L.d1;
  const int li = lane < nv ? lane : 0;
  const int il = li >= d1 ? li - d1 : li;           // dof index inside the agent
  const int pos = il < 6 ? il : 6 + (il & 1);       // chain position of the dof itself
  const double* Mrow = S(M) + li * c.L.mld;
#pragma unroll
  for (int p = 0; p < 6; p++) row[p] = Mrow[p] + (p == il ? diag_add : 0.0);
  const double mh = Mrow[il < 6 ? 0 : il - (il & 1)], md = Mrow[il];   // column of the leg's hip, own diagonal
  row[6] = il < 6 ? 0.0 : ((il & 1) ? mh : md + diag_add);
  row[7] = (il >= 6 && (il & 1)) ? md + diag_add : 0.0;
  if (with_contacts && lane < nv) {
    for (unsigned long long mk = ((const unsigned long long*)S(cmask))[lane]; mk; mk &= mk - 1) {
      const int ci = __builtin_ctzll(mk);
      const double* Jb = S(Jb) + ((c.si + c.L.con_b)[4 * ci + 3] & 0xFFFFF);   // one moving body: 3 rows x 8 chain slots
      const double* W = S(cW) + 5 * ci;
      const double a0 = Jb[pos], a1 = Jb[8 + pos], a2 = Jb[16 + pos];
      const double u0 = W[0] * a0 + W[1] * a1 + W[2] * a2, u1 = W[1] * a0 + W[3] * a1, u2 = W[2] * a0 + W[4] * a2;
#pragma unroll
      for (int p = 0; p < 8; p++) row[p] += u0 * Jb[p] + u1 * Jb[8 + p] + u2 * Jb[16 + p];   // slots past `pos` are unused entries
    }
  }
}

template <class C>
__device__ __forceinline__ double tree_factor_solve(C& c, const double (&row)[8], double b, int* fail) {
  const int lane = c.lane, nv = c.P->mdl.nv, d1 = c.L.d1;
  const bool act = lane < nv;
  const int B = lane >= d1 ? d1 : 0, il = lane - B;
  const int nvg = lane >= d1 ? nv - d1 : d1;        // dofs of this lane's agent
  const bool isleg = act && il >= 6, iship = !(il & 1);
  double* ROW = S(H);           // [nv][8] rows (root rows are overwritten by the Schur complement: [0..5] S, [6] rhs)
  double* WW = S(H) + 8 * nv;   // [nv][6] leg dofs: row of A_ll^-1 A_lr
  double* Y = S(tmpv);          // [nv]    leg dofs: A_ll^-1 b_l
  int bad = 0;
  // ---- leg blocks: [[dh, o], [o, da]] (hip, ankle); each lane computes its own row of the inverse applied to (A_lr | b_l)
  double pr[8];
#pragma unroll
  for (int p = 0; p < 8; p++) pr[p] = pair_swap_f64(row[p]);
  const double pb = pair_swap_f64(b);
  const double d_own = iship ? row[6] : row[7], d_par = iship ? pr[7] : pr[6], off = iship ? pr[6] : row[6];
  const double det = d_own * d_par - off * off;
  if (isleg && (d_par < MINVAL || det < MINVAL * d_par)) bad = 1;   // pivots of the 2x2 LDL: d_par, det / d_par
  const double idet = fast_rcp(det);
  double w[6];
#pragma unroll
  for (int p = 0; p < 6; p++) w[p] = (d_par * row[p] - off * pr[p]) * idet;
  const double y = (d_par * b - off * pb) * idet;
  if (act) {
#pragma unroll
    for (int p = 0; p < 8; p++) ROW[8 * lane + p] = row[p];
    if (isleg) {
#pragma unroll
      for (int p = 0; p < 6; p++) WW[6 * lane + p] = w[p];
      Y[lane] = y;
    }
  }
  SYNC();
  // ---- Schur complement of the root block: S = A_rr - A_rl A_ll^-1 A_lr, rhs' = b_r - A_rl A_ll^-1 b_l (root lane m: row m)
  if (act && il < 6) {
    double s[6], br = b;
#pragma unroll
    for (int p = 0; p < 6; p++) s[p] = row[p];
    // leg dofs of this agent: at most MAXLD; unrolled over the bound in batches of four with the 32 loads of a batch issued
    // before its 28 FMAs (a leg dof past the agent's own re-reads the first one and is masked out of the sums)
    constexpr int MAXLD = C::NV == 28 ? 8 : (C::NV == 32 ? 12 : 16);
    const int nld = nvg - 6;
#pragma unroll
    for (int k0 = 0; k0 < MAXLD; k0 += 4) {
      double am[4], yv[4], wv[4][6];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int d = B + 6 + (k0 + k < nld ? k0 + k : 0);
        am[k] = ROW[8 * d + il];
        yv[k] = Y[d];
#pragma unroll
        for (int p = 0; p < 6; p++) wv[k][p] = WW[6 * d + p];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double a_ = k0 + k < nld ? am[k] : 0.0;
#pragma unroll
        for (int p = 0; p < 6; p++) s[p] -= a_ * wv[k][p];
        br -= a_ * yv[k];
      }
    }
#pragma unroll
    for (int p = 0; p < 6; p++) ROW[8 * lane + p] = s[p];
    ROW[8 * lane + 6] = br;
  }
  SYNC();
  // ---- 6x6 LDL^T + solve, redundantly in every lane of the agent
  double L[6][6], xr[6];
  {
    const double* Sg = ROW + 8 * B;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
      for (int j = 0; j <= i; j++) L[i][j] = Sg[8 * i + j];
      xr[i] = Sg[8 * i + 6];
    }
    double rinv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (act && L[j][j] < MINVAL) bad = 1;
      rinv[j] = fast_rcp(L[j][j]);
      double u[6];
#pragma unroll
      for (int i = j + 1; i < 6; i++) u[i] = L[i][j];
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        const double lij = u[i] * rinv[j];
#pragma unroll
        for (int k = j + 1; k <= i; k++) L[i][k] -= lij * u[k];
        L[i][j] = lij;
      }
    }
#pragma unroll
    for (int i = 1; i < 6; i++)
#pragma unroll
      for (int j = 0; j < i; j++) xr[i] -= L[i][j] * xr[j];
#pragma unroll
    for (int i = 0; i < 6; i++) xr[i] *= rinv[i];
#pragma unroll
    for (int i = 4; i >= 0; i--)
#pragma unroll
      for (int k = i + 1; k < 6; k++) xr[i] -= L[k][i] * xr[k];
  }
  // ---- back substitution into the legs: x_l = A_ll^-1 b_l - (A_ll^-1 A_lr) x_r
  double x = y;
#pragma unroll
  for (int p = 0; p < 6; p++) x -= w[p] * xr[p];
#pragma unroll
  for (int p = 0; p < 6; p++) x = il == p ? xr[p] : x;
  SYNC();
  *fail = __ballot(bad) != 0ull;
  return act ? x : 0.0;
}

// cost(x) = 1/2 (Ma - qfrc_smooth).(x - qacc_smooth) + sum_active 1/2 D jar^2   (jar of the contact rows in LDS; the
// limit rows of a dof are evaluated by its lane: jar = +-x - aref)
template <class C>
__device__ __forceinline__ double solver_cost(C& c, double Ma_i, double x_i) {
  const int lane = c.lane, nv = c.P->mdl.nv;
  double v = 0;
  if (lane < nv) v = 0.5 * (Ma_i - S(qsm)[lane]) * (x_i - S(asmo)[lane]);
  const LimRows q = lim_rows(c);
  const double j0 = x_i - q.A0, j1 = -x_i - q.A1;
  if (j0 < 0) v += 0.5 * q.D0 * j0 * j0;
  if (j1 < 0) v += 0.5 * q.D1 * j1 * j1;
  for (int r = lane; r < 4 * c.ncon; r += WAVE) { double j = S(jar)[r]; if (j < 0) v += 0.5 * S(D)[r >> 2] * j * j; }
  return wave_sum(v);
}

// TREE: every contact touches one moving body (c.htree) -- two instantiations so that the register-hungry general
// factorisation does not share a loop (and its spills) with the common tree-sparse one
template <bool TREE, class C>
__device__ __forceinline__ void newton_solve(C& c) {
  const auto& mdl = model_view(c);
  const auto& aux = aux_view(c);
  const int lane = c.lane, nv = mdl.nv, nefc = c.nefc, ncon = c.ncon, nrow = 4 * c.ncon;
  const double tol = MF(opt)[SUMO_OPT_TOLERANCE];
  const int maxiter = (int)MF(opt)[SUMO_OPT_ITERATIONS];
  const double scale = 1.0 / (MF(opt)[SUMO_OPT_MEANINERTIA] * (nv > 1 ? nv : 1));
  double* x = S(x);
  if (nefc == 0) {
    if (lane < nv) x[lane] = S(asmo)[lane];
    SYNC();
    return;
  }
  constexpr int RPL = C::NV <= 36 ? 2 : 3;   // contact rows per lane: 4 * maxcon <= 64 * RPL (build_layout)
  // ---- warm start: better of qacc_warmstart (or, in warm_mode 1, the previous RK stage's qacc still held in x) and
  // qacc_smooth
  if (!c.use_prev && lane < nv) x[lane] = S(warm)[lane];
  SYNC();
  double Ma = dense_Mx(c, x);
  contact_Jx(c, x);
  for (int r = lane; r < nrow; r += WAVE) S(jar)[r] = row_Jx(c, r) - S(aref)[r];
  SYNC();
  double xi = lane < nv ? x[lane] : 0.0;
  double cost_ws = solver_cost(c, Ma, xi);
  contact_Jx(c, S(asmo));
  double jsm[RPL];
  double csm = 0;
  {
    const LimRows q = lim_rows(c);
    const double xs = lane < nv ? S(asmo)[lane] : 0.0, j0 = xs - q.A0, j1 = -xs - q.A1;
    if (j0 < 0) csm += 0.5 * q.D0 * j0 * j0;
    if (j1 < 0) csm += 0.5 * q.D1 * j1 * j1;
  }
#pragma unroll
  for (int q = 0; q < RPL; q++) {
    int r = lane + WAVE * q;
    double jv = r < nrow ? row_Jx(c, r) - S(aref)[r] : 1.0;
    jsm[q] = jv;
    if (jv < 0) csm += 0.5 * S(D)[r >> 2] * jv * jv;
  }
  double cost_sm = wave_sum(csm);
  double cost = cost_ws;
  if (cost_ws > cost_sm) {
    if (lane < nv) { xi = S(asmo)[lane]; x[lane] = xi; }
#pragma unroll
    for (int q = 0; q < RPL; q++) { int r = lane + WAVE * q; if (r < nrow) S(jar)[r] = jsm[q]; }
    SYNC();
    Ma = dense_Mx(c, x);
    cost = solver_cost(c, Ma, xi);
  }
  int iters = 0;
  // loop invariants of this lane's dof, kept in registers: its limit rows, smooth force and unconstrained acceleration
  const LimRows lq = lim_rows(c);
  // (unconditional loads at a clamped index + selects: no predicated region right in front of the iteration loop -- the register
  // allocator parked loop-invariant registers of the step loop in exactly such a region once, see tools/exec_copy_check.py)
  double qsm_i = S(qsm)[lane < nv ? lane : 0], asmo_i = S(asmo)[lane < nv ? lane : 0];
  asm volatile("" : "+v"(qsm_i), "+v"(asmo_i));
  qsm_i = lane < nv ? qsm_i : 0.0; asmo_i = lane < nv ? asmo_i : 0.0;
  PROF(11);
  for (int iter = 0; iter < maxiter; iter++) {
    iters++;
    // ---- per-contact force / weight summaries of the active set
    for (int ci = lane; ci < ncon; ci += WAVE) {
      double mu = S(cpar)[ci], f[4], dact[4];
      const double Dr = S(D)[ci];
      for (int k = 0; k < 4; k++) {
        double j = S(jar)[4 * ci + k];
        dact[k] = j < 0 ? Dr : 0.0;
        f[k] = j < 0 ? -Dr * j : 0.0;
      }
      double* cp = S(cp) + 3 * ci;
      cp[0] = ((f[0] + f[1]) + f[2]) + f[3];
      cp[1] = mu * (f[0] - f[1]);
      cp[2] = mu * (f[2] - f[3]);
      double* W = S(cW) + 5 * ci;
      W[0] = ((dact[0] + dact[1]) + dact[2]) + dact[3];
      W[1] = mu * (dact[0] - dact[1]);
      W[2] = mu * (dact[2] - dact[3]);
      W[3] = mu * mu * (dact[0] + dact[1]);
      W[4] = mu * mu * (dact[2] + dact[3]);
    }
    SYNC();
    // ---- gradient: own limit rows (lane-local) + dof-major gather of J^T f over the contacts touching the dof
    double g = 0, dl = 0;
    if (lane < nv) {
      double qc = 0;
      const double j0 = xi - lq.A0, j1 = -xi - lq.A1;
      if (j0 < 0) { qc -= lq.D0 * j0; dl += lq.D0; }   // force -D jar through the row +e_dof
      if (j1 < 0) { qc += lq.D1 * j1; dl += lq.D1; }   // row -e_dof
      for (unsigned long long mk = ((const unsigned long long*)S(cmask))[lane]; mk; mk &= mk - 1) {
        int ci = __builtin_ctzll(mk);
        const int s = SLOT_OF(ci, lane, c.dpos);
        const int info = (c.si + c.L.con_b)[4 * ci + 3], ns = (info >> 20) ? 16 : 8;
        const double* Jb = S(Jb) + (info & 0xFFFFF);
        const double* cp = S(cp) + 3 * ci;
        qc += Jb[s] * cp[0] + Jb[ns + s] * cp[1] + Jb[2 * ns + s] * cp[2];
      }
      g = Ma - qsm_i - qc;
      if (!TREE) S(dlim)[lane] = dl;
    }
    double gn = wave_sum(g * g);
    PROF(12);
    if (scale * sqrt(gn) < tol) break;
    SYNC();
    // ---- Hessian H = M + J^T diag(D_active) J, lower triangle: entry-major gather (every lane a few entries).
    // All loads of the mass-matrix / mask words are issued before any store so they pipeline.
    if (!TREE) {
      const unsigned long long* cm = (const unsigned long long*)S(cmask);
      double hreg[C::EPL];
      unsigned long long mreg[C::EPL];
      // (the entry words are made opaque here so that the addresses derived from them are computed in this rarely taken
      // block instead of being hoisted to kernel entry and kept in registers / scratch for the whole launch)
      unsigned ent[C::EPL];
      int pi[C::EPL], pj[C::EPL];   // chain positions of the entry's two dofs (slot lookup, SLOT_OF)
#pragma unroll
      for (int m = 0; m < C::EPL; m++) {
        unsigned w = (unsigned)pt_global(launder_ptr(c.P->aux.ai + c.P->aux.o_ent))[c.lane + WAVE * m];   // loaded here, in the rarely taken block
        ent[m] = w & 0xFFFFu;
        pi[m] = (w >> 16) & 0xF; pj[m] = (w >> 20) & 0xF;
      }
#pragma unroll
      for (int m = 0; m < C::EPL; m++) {
        unsigned e = ent[m];
        int i = e == 0xFFFFu ? 0 : (int)(e >> 8), jj = e == 0xFFFFu ? 0 : (int)(e & 0xFF);
        double h = SAME_TREE(i, jj) ? S(M)[MIDX(i, jj)] : 0.0;
        if (i == jj) h += S(dlim)[i];
        hreg[m] = h;
        mreg[m] = e == 0xFFFFu ? 0ull : (cm[i] & cm[jj]);
      }
#pragma unroll
      for (int m = 0; m < C::EPL; m++) {
        unsigned e = ent[m];
        int i = e >> 8, jj = e & 0xFF;
        double h = hreg[m];
        for (unsigned long long mk = mreg[m]; mk; mk &= mk - 1) {
          int ci = __builtin_ctzll(mk);
          const int si = SLOT_OF(ci, i, pi[m]), sj = SLOT_OF(ci, jj, pj[m]);
          const int info = (c.si + c.L.con_b)[4 * ci + 3], ns = (info >> 20) ? 16 : 8;
          const double* Jb = S(Jb) + (info & 0xFFFFF);
          const double* W = S(cW) + 5 * ci;
          double a0 = Jb[si], a1 = Jb[ns + si], a2 = Jb[2 * ns + si], b0 = Jb[sj], b1 = Jb[ns + sj], b2 = Jb[2 * ns + sj];
          h += a0 * (W[0] * b0 + W[1] * b1 + W[2] * b2) + a1 * (W[1] * b0 + W[3] * b1) + a2 * (W[2] * b0 + W[4] * b2);
        }
        hreg[m] = h;
      }
#pragma unroll
      for (int m = 0; m < C::EPL; m++) {
        unsigned e = ent[m];
        if (e != 0xFFFFu) S(H)[HP((int)(e >> 8), (int)(e & 0xFF))] = hreg[m];
      }
    }
    SYNC();
    PROF(13);
    int hfail;
    double sr;
    if (TREE) {  // every contact touches one moving body: H keeps the tree sparsity of M (see tree_factor_solve)
      double row[8];
      tree_rows(c, row, dl, ncon > 0);
      sr = -tree_factor_solve(c, row, g, &hfail);
    } else {
      sr = -ldl_solve_rows(c, S(H), S(H), g, &hfail);
    }
    if (hfail) break;
    if (lane < nv) S(search)[lane] = sr;
    SYNC();
    PROF(14);
    // ---- exact line search
    double Mv = dense_Mx(c, S(search));
    contact_Jx(c, S(search));
    // each lane keeps its contact rows (r = lane + 64 q) and its dof's two limit rows (jar, Jv, D) in registers
    double rj[RPL + 2], rjv[RPL + 2], rD[RPL + 2];
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      int r = lane + WAVE * q;
      bool ok = r < nrow;
      rjv[q] = ok ? row_Jx(c, r) : 0.0;
      rj[q] = ok ? S(jar)[r] : 1.0;      // absent rows: jar > 0 and Jv = 0 never contribute
      rD[q] = ok ? S(D)[r >> 2] : 0.0;
    }
    rj[RPL] = xi - lq.A0; rjv[RPL] = sr; rD[RPL] = lq.D0;
    rj[RPL + 1] = -xi - lq.A1; rjv[RPL + 1] = -sr; rD[RPL + 1] = lq.D1;
    double g1 = wave_sum(lane < nv ? sr * (Ma - qsm_i) : 0.0);
    double g2 = wave_sum(lane < nv ? sr * Mv : 0.0);
    double alpha = 0, lo = 0, hi = -1, d0 = 0;
    for (int ls = 0; ls < 50; ls++) {
      double f1 = 0, f2 = 0;
#pragma unroll
      for (int q = 0; q < RPL + 2; q++) {
        double j = rj[q] + alpha * rjv[q];
        if (j < 0) { f1 += rD[q] * j * rjv[q]; f2 += rD[q] * rjv[q] * rjv[q]; }
      }
      f1 = wave_sum(f1) + (g1 + alpha * g2);
      f2 = wave_sum(f2) + g2;
      if (ls == 0) d0 = fabs(f1);
      if (fabs(f1) <= 1e-10 * d0) break;
      if (f1 < 0) lo = alpha; else hi = alpha;
      double an = alpha - f1 * fast_rcp(f2);
      if (an <= lo || (hi >= 0 && an >= hi)) an = hi >= 0 ? 0.5 * (lo + hi) : 2 * (alpha > 0 ? alpha : 1.0);
      alpha = an;
    }
    PROF(15);
    if (alpha == 0) break;
    if (lane < nv) { xi += alpha * sr; x[lane] = xi; Ma += alpha * Mv; }
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      int r = lane + WAVE * q;
      if (r < nrow) S(jar)[r] = rj[q] + alpha * rjv[q];
    }
    SYNC();
    double oldcost = cost;
    {  // cost at the new point from the rows this lane already holds (same terms as solver_cost, no LDS round trip)
      double v = lane < nv ? 0.5 * (Ma - qsm_i) * (xi - asmo_i) : 0.0;
      const double j0 = xi - lq.A0, j1 = -xi - lq.A1;
      if (j0 < 0) v += 0.5 * lq.D0 * j0 * j0;
      if (j1 < 0) v += 0.5 * lq.D1 * j1 * j1;
#pragma unroll
      for (int q = 0; q < RPL; q++) { const double jn = rj[q] + alpha * rjv[q]; if (jn < 0) v += 0.5 * rD[q] * jn * jn; }
      cost = wave_sum(v);
    }
    if (scale * (oldcost - cost) < tol) break;
  }
  c.st_newton += iters;
  if (iters > c.st_maxnewton) c.st_maxnewton = iters;
  SYNC();
}

#ifdef SUMO_DBG_DUMP
#define DUMP(slot, ptr, n) do { if (c.dbg && c.st_forward <= 20 && c.lane < (n)) c.dbg[((c.st_forward - 1) * 8 + (slot)) * 64 + c.lane] = (ptr)[c.lane]; } while (0)
#else
#define DUMP(slot, ptr, n) do { } while (0)
#endif
// ---- mj_forward ----------------------------------------------------------------------------------------------
template <class C>
__device__ __forceinline__ void forward(C& c) {
  const auto& mdl = model_view(c);
  const int lane = c.lane, nv = mdl.nv;
  c.st_forward++;
  DUMP(0, S(qpos), mdl.nq); DUMP(1, S(qvel), nv); DUMP(2, S(ctrl), mdl.nu); DUMP(7, S(warm), nv);
  position_velocity(c);
  collision(c);
  make_constraint(c);
#ifdef SUMO_DBG_DUMP
  if (c.dbg && c.st_forward <= 20 && lane < 3) c.dbg[((c.st_forward - 1) * 8 + 5) * 64 + lane] = lane == 0 ? c.ncon : (lane == 1 ? c.nefc : c.nlim);
#endif
  // efc_vel and aref (B and K*imp*(pos-margin) were parked in Jv / jar)
  contact_Jx(c, S(qvel));
  for (int r = lane; r < 4 * c.ncon; r += WAVE) S(aref)[r] = -S(aref)[r] * row_Jx(c, r) - S(jar)[r];  // B was parked in aref
  SYNC();
  PROF(8);
  mass_matrix(c);  // last user of the kinematic scratch; H may overwrite it from here on
  PROF(9);
  // smooth forces: passive (damping) - bias + actuation
  KCONSTS();
  if (lane < nv) S(qsm)[lane] = -K.d_damp * S(qvel)[lane] - S(bias)[lane];
  SYNC();
  if (lane < mdl.nu) {
    double u = S(ctrl)[lane];
    if (u < K.a_lo) u = K.a_lo;
    if (u > K.a_hi) u = K.a_hi;
    S(qsm)[K.a_dof] += K.a_gear * u;  // one motor per dof in these scenes
  }
  SYNC();
  DUMP(3, S(qsm), nv);
  // qacc_smooth = M^-1 qfrc_smooth
  int mfail;
  double as;
  if (c.L.tree_ok) {
    double row[8];
    tree_rows(c, row, 0.0, false);
    as = tree_factor_solve(c, row, lane < nv ? S(qsm)[lane] : 0.0, &mfail);
  } else {  // unknown tree shape: pack M into the Hessian buffer and use the dense factorisation
#pragma unroll
    for (int m = 0; m < C::EPL; m++) {
      unsigned e = (unsigned)pt_global(launder_ptr(c.P->aux.ai + c.P->aux.o_ent))[c.lane + WAVE * m];   // (address arithmetic stays in this rarely taken block)
      e &= 0xFFFFu;
      if (e != 0xFFFFu) { int i = e >> 8, jj = e & 0xFF; S(H)[HP(i, jj)] = SAME_TREE(i, jj) ? S(M)[MIDX(i, jj)] : 0.0; }
    }
    SYNC();
    as = ldl_solve_rows(c, S(H), S(H), lane < nv ? S(qsm)[lane] : 0.0, &mfail);
  }
  if (mfail) as = 0.0;
  if (lane < nv) S(asmo)[lane] = as;
  SYNC();
  PROF(10);
  DUMP(4, S(asmo), nv);
  if (c.htree) newton_solve<true>(c);
  else {
    // The general (dense-factorisation) path costs ~1.65x a tree-path forward and the envs on it -- agents in contact with each
    // other, a few per launch -- are the stragglers every launch waits for (tools/slot_trace.py).  From its first dense
    // forward on, such a wave issues ahead of the wave it shares the SIMD with (which has slack).
    __builtin_amdgcn_s_setprio(3);
    newton_solve<false>(c);
    c.st_dense++;
    if (c.hcross) c.st_cross++;
  }
  PROF(16);
  DUMP(6, S(x), nv);
  c.st_ncon += c.ncon;
  c.st_nefc += c.nefc;
  if (c.ncon > c.st_maxcon) c.st_maxcon = c.ncon;
  if (c.nefc > c.st_maxefc) c.st_maxefc = c.nefc;
  c.st_dropped += c.ndropped;
}

// qpos '+'= h * vel  (vel in LDS), one lane per joint
template <class C>
__device__ __forceinline__ void integrate_pos(C& c, double* qpos, const double* vel, double h) {
  KCONSTS();
  if (c.lane < c.P->mdl.njnt) {
    const int qa = K.jt_qadr, da = K.jt_dadr;
    if (K.jt_type == SUMO_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[da + k];
      double ax[3] = {vel[da + 3], vel[da + 4], vel[da + 5]}, qr[4], q[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]};
      double ang = h * normalize3(ax);
      axisangle2quat(qr, ax, ang);
      normalize4(q);
      mulquat(q, q, qr);
      qpos[qa + 3] = q[0]; qpos[qa + 4] = q[1]; qpos[qa + 5] = q[2]; qpos[qa + 6] = q[3];
    } else {
      qpos[qa] += h * vel[da];
    }
  }
}

// frame_skip x mj_step with the RK4 stages flattened into one loop, so forward() has a single (inlined) call site.
// MuJoCo's bad-value test (mj_checkPos / mj_checkVel / mj_checkAcc: NaN or |x| > mjMAXVAL = 1e10 in qpos, qvel, qacc; the
// reference turns the resulting warning into a MujocoException, mujoco-py/mujoco_py/builder.py:351-369).  Wave-uniform result.
template <class C>
__device__ __forceinline__ int state_is_bad(C& c) {
  const auto& mdl = model_view(c);
  const int lane = c.lane;
  bool bad = false;
  for (int i = lane; i < mdl.nq; i += WAVE) bad |= !(fabs(S(qpos)[i]) <= 1e10);   // NaN fails the comparison too
  for (int i = lane; i < mdl.nv; i += WAVE) bad |= !(fabs(S(qvel)[i]) <= 1e10) || !(fabs(S(warm)[i]) <= 1e10);
  return __builtin_amdgcn_ballot_w64(bad) != 0ull;
}

// RK4 tableau (MuJoCo): A = diag(1/2, 1/2, 1), B = (1/6, 1/3, 1/3, 1/6); the warm start saved at the end of a step is
// the last stage's qacc.
template <class C>
__device__ __forceinline__ void mj_steps(C& c, int nsteps) {
  const auto& mdl = model_view(c);
  const int lane = c.lane, nq = mdl.nq, nv = mdl.nv;
  const double h = MF(opt)[SUMO_OPT_TIMESTEP];
  double q0 = 0, v0 = 0, accv = 0, acca = 0;   // this lane's entry of the step's start state and of the RK4 accumulators
  for (int sub = 0; sub < 4 * nsteps; sub++) {
    const int stage = sub & 3;
    c.use_prev = (c.L.warm_mode == 1 && stage != 0) ? 1 : 0;
    // make the lane id opaque per iteration: stops the compiler from hoisting hundreds of lane-dependent address
    // computations out of this loop (they would be spilled to scratch under the 256-register budget)
    asm volatile("" : "+v"(c.lane));
    forward(c);
    if (stage == 0) {
      if (lane < nq) q0 = S(qpos)[lane];
      if (lane < nv) {
        v0 = S(qvel)[lane];
        accv = 0.0 + (1.0 / 6.0) * S(qvel)[lane];
        acca = 0.0 + (1.0 / 6.0) * S(x)[lane];
      }
    } else {
      const double Bi = stage == 3 ? 1.0 / 6.0 : 1.0 / 3.0;
      if (lane < nv) { accv += Bi * S(qvel)[lane]; acca += Bi * S(x)[lane]; }
    }
    SYNC();
    if (stage < 3) {
      const double Ai = stage == 2 ? 1.0 : 0.5;  // A[stage]
      double dv = 0;
      if (lane < nv) { S(tmpv)[lane] = Ai * S(qvel)[lane]; dv = Ai * S(x)[lane]; }
      if (lane < nq) S(qpos)[lane] = q0;
      SYNC();
      integrate_pos(c, S(qpos), S(tmpv), h);
      if (lane < nv) S(qvel)[lane] = v0 + h * dv;
      SYNC();
    } else {
      if (lane < nq) S(qpos)[lane] = q0;
      if (lane < nv) { S(qvel)[lane] = v0 + h * acca; S(warm)[lane] = S(x)[lane]; S(tmpv)[lane] = accv; }
      SYNC();
      integrate_pos(c, S(qpos), S(tmpv), h);
      SYNC();
    }
  }
}

// ---- RNG / reset / observation -----------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t* out, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double rng_uniform(uint64_t seed, uint32_t reset_count, uint32_t k) {
  uint32_t o[4];
  philox4x32(o, reset_count, k >> 1, 0u, 0x53554D4Fu, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t a = ((k & 1) ? o[2] : o[0]) >> 5, b = ((k & 1) ? o[3] : o[1]) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

// sumo.py:232-253 with a counter RNG; state ends up in LDS (qpos, qvel, warm)
template <class C>
__device__ __forceinline__ void reset_state(C& c, uint64_t seed, uint32_t rc) {
  const auto& mdl = model_view(c);
  const int lane = c.lane, nq = mdl.nq, nv = mdl.nv;
  if (lane < nq) S(qpos)[lane] = MF(qpos0)[lane];
  SYNC();
  if (lane < mdl.nagent) {
    double phi = 2.0 * PI_D * rng_uniform(seed, rc, 0);
    double ang = phi + lane * (2.0 * PI_D / 2.0);
    int qa = MI(agent_qposadr)[lane];
    S(qpos)[qa] = 1.15 * slow_cos(ang);
    S(qpos)[qa + 1] = 1.15 * slow_sin(ang);
    S(qpos)[qa + 2] = 1.25;
  }
  SYNC();
  if (lane < nq) S(qpos)[lane] += -0.1 + 0.2 * rng_uniform(seed, rc, 1 + lane);
  if (lane < nv) {
    double u1 = rng_uniform(seed, rc, RNG_NORMAL_BASE + 2 * (lane >> 1));
    double u2 = rng_uniform(seed, rc, RNG_NORMAL_BASE + 2 * (lane >> 1) + 1);
    double rr = sqrt(-2.0 * slow_log(1.0 - u1));
    double z = (lane & 1) ? rr * slow_sin(2.0 * PI_D * u2) : rr * slow_cos(2.0 * PI_D * u2);
    S(qvel)[lane] = 0.1 * z;
    S(warm)[lane] = 0.0;
  }
  SYNC();
  if (lane < mdl.njnt && MI(jnt_type)[lane] == SUMO_JNT_FREE) {
    double* q = S(qpos) + MI(jnt_qposadr)[lane] + 3;
    double qq[4] = {q[0], q[1], q[2], q[3]};
    normalize4(qq);
    q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
  }
  SYNC();
}

// agents.py:190-214 (cfrc_ext == 0) + time feature (sumo_env.py:68-70)
// ---- coherent hand-over accessors -----------------------------------------------------------------------------------
// In the fused rollout launch an env moves between waves (possibly on another XCD, whose L2 is not coherent with ours) from one
// step to the next.  Only a few hundred bytes cross: the state record, the counters, the observations and the done flags.
// They are written and read with agent-scope relaxed atomics -- `sc1` stores that write through to memory and `sc1` loads
// that bypass the non-coherent cache levels (MI355X_MICROARCH.md, inter-workgroup visibility, the all-sc1 form) -- so no
// L2 write-back / invalidate fence is needed per step: a release fence there flushes every dirty line of the XCD's L2,
// scratch frames included, 2 million times a second (measured: 67 KB of HBM writes per env step against 2.5 KB algorithmic).
// COH = false (per-step launch): plain accesses, the kernel boundary orders them.
// (through global-address-space pointers: in the rollout kernel the buffers' addresses come out of the laundered launch arguments,
// i.e. generic pointers, whose flat accesses would tie up the LDS counter as well -- see PT_GAS in ppo_tile.h)
template <bool COH, class T>
__device__ __forceinline__ T hand_load(const T* p) {
  if (COH) return __hip_atomic_load(pt_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *pt_global(p);
}
template <bool COH, class T>
__device__ __forceinline__ void hand_store(T* p, T v) {
  if (COH) __hip_atomic_store(pt_global(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *pt_global(p) = v;
}

template <bool COH = false, class C>
__device__ __forceinline__ void write_obs(C& c, float* obs, int obs_stride, int num_steps) {
  const auto& mdl = model_view(c);
  const double adjz = c.P->adjust_z;
  for (int idx = c.lane; idx < 2 * obs_stride; idx += WAVE) {
    int a = idx >= obs_stride, k = idx - a * obs_stride, o = 1 - a;
    int nqa = MI(agent_nq)[a], nva = MI(agent_nv)[a], nba = MI(agent_nbody)[a];
    int dim = nqa + nva + 6 * nba + 14;
    float v = 0.0f;
    // get_qpos() adds _adjust_z to the z entry of the copy it returns (agents.py:155-161): own z and the opponent's z
    if (k < nqa) { double q = S(qpos)[MI(agent_qposadr)[a] + k]; if (k == 2) q += adjz; v = (float)q; }
    else if (k < nqa + nva) v = (float)S(qvel)[MI(agent_dofadr)[a] + (k - nqa)];
    else if (k < nqa + nva + 6 * nba) v = 0.0f;
    else if (k < nqa + nva + 6 * nba + 7) { const int j = k - nqa - nva - 6 * nba; double q = S(qpos)[MI(agent_qposadr)[o] + j]; if (j == 2) q += adjz; v = (float)q; }
    else if (k < dim - 1) v = 0.0f;
    else if (k == dim - 1) v = (float)(-1.0 + 2.0 * num_steps / 500.0);
    hand_store<COH>(obs + idx, v);
  }
}

__device__ __forceinline__ float sumsq_f32(const float* a, int n) {  // numpy float32 pairwise sum of squares (n < 128)
  if (n < 8) { float r = 0.0f; for (int i = 0; i < n; i++) r += a[i] * a[i]; return r; }
  float r[8];
  for (int j = 0; j < 8; j++) r[j] = a[j] * a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) r[j] += a[i + j] * a[i + j];
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += a[i] * a[i];
  return res;
}

// the same sum over the step's control vector in LDS (S(ctrl)[i] == (double)action[i] exactly: float -> double -> float)
__device__ __forceinline__ float sumsq_f32_ctrl(const double* u, int n) {
  if (n < 8) { float r = 0.0f; for (int i = 0; i < n; i++) { const float a = (float)u[i]; r += a * a; } return r; }
  float r[8];
  for (int j = 0; j < 8; j++) { const float a = (float)u[j]; r[j] = a * a; }
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) { const float a = (float)u[i + j]; r[j] += a * a; }
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) { const float a = (float)u[i]; res += a * a; }
  return res;
}

template <class C>
__device__ __forceinline__ void ctx_init(C& c, const Params* P, double* smem) {
  c.P = P;
  if constexpr (std::is_same<typename C::LT, Layout>::value) c.L = P->L;
  c.sm = smem;
  c.si = (int*)(smem + c.L.i_base);
  c.sb = (unsigned char*)(smem + c.L.i_base);
  c.lane = threadIdx.x;
  c.kp = P->lanes + c.lane;
  c.prp = P->pair_rec + c.lane;
  c.pbp = P->pair_bound + c.lane;
  // static tables -> LDS (once per launch); world centres into the tail of xipos / gaxis
#if SUMO_STAT_LDS
  for (int i = c.lane; i < P->aux.n_stat_d; i += WAVE) smem[P->L.stat_d + i] = P->aux.af[P->aux.o_stat_d + i];
  for (int i = c.lane; i < P->aux.n_stat_i; i += WAVE) c.si[P->L.stat_i + i] = P->aux.ai[P->aux.o_stat_i + i];
#endif
  {
    const int nb = P->mdl.nbody, nw = P->aux.nworld, nc = P->aux.nc;
    const double* wpos = P->aux.af + P->aux.o_wpa;
    for (int i = c.lane; i < 3 * nw; i += WAVE) { smem[P->L.xipos + 3 * nb + i] = wpos[i]; smem[P->L.gaxis + 3 * nb + i] = wpos[3 * nw + i]; }
  }
  c.hid = c.lane < P->mdl.nv ? P->lanes[c.lane].d_hid : -1;
  c.dpos = P->lanes[c.lane].d_pos;
  __syncthreads();
  c.ncon = c.nlim = c.nefc = c.ndropped = c.use_prev = 0;
  c.st_forward = c.st_newton = c.st_ncon = c.st_nefc = c.st_maxcon = c.st_maxefc = c.st_maxnewton = c.st_dropped = c.st_dense = c.st_cross = 0;
  c.st_diverged = 0;
  c.st_cb3 = c.st_rodcap = 0;
#ifdef SUMO_PROFILE
  for (int k = 0; k < 24; k++) c.prof[k] = 0;
  c.tprev = clock64();
#endif
}
// Checksum of a state record as it crosses between waves in the fused rollout launch (COH): every word is mixed with its index,
// the lanes' contributions are summed over the wave.  The writer stores it next to the record's sequence tag (counters[2..3]),
// the wave that takes the env over recomputes it from what it LOADED: a stale or torn read of the record -- the failure the
// hand-rolled sc1 protocol would produce silently -- raises the launch's abort flag instead (sumo_rollout_status).
__device__ __forceinline__ unsigned rec_hash(double v, int i) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  return (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA77u + (unsigned)(i + 1) * 0xC2B2AE3Du;
}
__device__ __forceinline__ unsigned wave_usum(unsigned v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
template <bool COH = false, class C, class SA>   // SA: StepArgs by value (per-step kernel) or in the kernel-argument segment (rollout kernel)
__device__ __forceinline__ unsigned load_state(C& c, const SA& a, int e) {   // returns the record's checksum (COH) or 0
  const int nq = c.P->mdl.nq, nv = c.P->mdl.nv, lane = c.lane;
  const double* st = a.state + (size_t)e * a.state_stride;
  unsigned h = 0;
  for (int i = lane; i < nq + 2 * nv; i += WAVE) {
    double v = hand_load<COH>(st + i);
    if (COH) h += rec_hash(v, i);
    if (i < nq) S(qpos)[i] = v; else if (i < nq + nv) S(qvel)[i - nq] = v; else S(warm)[i - nq - nv] = v;
  }
  return COH ? wave_usum(h) : 0u;
}
template <bool COH = false, class C, class SA>
__device__ __forceinline__ unsigned store_state(C& c, const SA& a, int e) {
  const int nq = c.P->mdl.nq, nv = c.P->mdl.nv, lane = c.lane;
  double* st = a.state + (size_t)e * a.state_stride;
  unsigned h = 0;
  for (int i = lane; i < nq + 2 * nv; i += WAVE) {
    const double v = i < nq ? S(qpos)[i] : (i < nq + nv ? S(qvel)[i - nq] : S(warm)[i - nq - nv]);
    if (COH) h += rec_hash(v, i);
    hand_store<COH>(st + i, v);
  }
  return COH ? wave_usum(h) : 0u;
}
template <class C>
__device__ __forceinline__ void flush_stats(C& c, unsigned long long* stats) {
  if (c.lane == 0 && stats) {
    atomicAdd(stats + 0, (unsigned long long)c.st_forward);
    atomicAdd(stats + 1, (unsigned long long)c.st_newton);
    atomicAdd(stats + 2, (unsigned long long)c.st_ncon);
    atomicAdd(stats + 3, (unsigned long long)c.st_nefc);
    atomicMax(stats + 4, (unsigned long long)c.st_maxcon);
    atomicMax(stats + 5, (unsigned long long)c.st_maxefc);
    atomicMax(stats + 6, (unsigned long long)c.st_maxnewton);
    atomicAdd(stats + 7, (unsigned long long)c.st_dropped);
    if (c.st_diverged) atomicAdd(stats + 8, (unsigned long long)c.st_diverged);
    if (c.st_cb3) atomicAdd(stats + 11, (unsigned long long)c.st_cb3);
    if (c.st_rodcap) atomicAdd(stats + 12, (unsigned long long)c.st_rodcap);
#ifdef SUMO_PROFILE
    for (int k = 0; k < 24; k++) atomicAdd(stats + 16 + k, c.prof[k]);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------
// Longest-first schedule for small launches: perm[rank] = env, rank = number of envs with a larger (cost, -index) key --
// the same order as a stable descending sort.  Run by the first workgroups of the step launch itself (on the estimates
// of the PREVIOUS launch, for the NEXT one), so no kernel sits between two env steps of a group for it: a library sort
// there cost 136 us on average (its large workgroups wait until the other group's whole launch has been placed), a
// separate rank kernel still 35-40 us.
#define SCHED_RANK_MAX 4096
__device__ __forceinline__ void sched_rank(const int* __restrict__ cost, int n, int* __restrict__ perm, int block, int lane) {
  const int e = block * WAVE + lane;
  const int mine = e < n ? ((cost[e] << 12) | (SCHED_RANK_MAX - 1 - e)) : 0x7FFFFFFF;
  int rank = 0;
  for (int j0 = 0; j0 < n; j0 += 8 * WAVE) {   // 512 keys per batch: eight coalesced loads in flight, then lane broadcasts
    int key[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = j0 + WAVE * u + lane;
      key[u] = j < n ? ((cost[j] << 12) | (SCHED_RANK_MAX - 1 - j)) : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll 8
      for (int l = 0; l < WAVE; l++) rank += (__builtin_amdgcn_readlane(key[u], l) > mine) ? 1 : 0;
  }
  if (e < n) perm[rank] = e;
}

extern __shared__ double smem_dyn[];
#ifdef SUMO_DBG_POISON_LDS
// development: every LDS word holds a signalling pattern before an env step starts, so that a phase that reads a word nobody
// wrote shows up as NaN in the parity tests instead of depending on what the previous step / workgroup left there
__device__ __forceinline__ void poison_lds(const Params* P, int lane) {
  const int n = P->L.total_bytes / 8;
#ifndef SUMO_DBG_POISON_LO
#define SUMO_DBG_POISON_LO 0
#define SUMO_DBG_POISON_HI 0x7fffffff
#endif
  // kinds 4 / 5: PLAUSIBLE leftovers -- what another env's step would leave in the slot: doubles of order one (kind 4) or pairs of
  // small integers (kind 5), different in every word and workgroup (uniform patterns hide reads that are compared or used as masks);
  // LO / HI restrict the fill to a range of 8-byte words (bisection)
  for (int i = lane; i < n; i += WAVE) {
    if (i < SUMO_DBG_POISON_LO || i >= SUMO_DBG_POISON_HI) continue;
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    double v = 0.0;
    if (SUMO_DBG_POISON_LDS == 1) v = 1e300;
    else if (SUMO_DBG_POISON_LDS == 2) v = __longlong_as_double(0x7FF8DEADBEEF0000ll);
    else if (SUMO_DBG_POISON_LDS == 4) v = ((double)(h & 0xFFFFFu) / 524288.0 - 1.0) * 1.5;
    else if (SUMO_DBG_POISON_LDS == 5) v = __longlong_as_double((long long)(h % 24u) | ((long long)((h >> 8) % 24u) << 32));
    smem_dyn[i] = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {   // what ctx_init keeps in LDS for the whole launch: the world geoms' centres / axes
    const int nb = P->mdl.nbody, nw = P->aux.nworld;
    const double* wpos = P->aux.af + P->aux.o_wpa;
    for (int i = lane; i < 3 * nw; i += WAVE) { smem_dyn[P->L.xipos + 3 * nb + i] = wpos[i]; smem_dyn[P->L.gaxis + 3 * nb + i] = wpos[3 * nw + i]; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#endif

// One env step of env `e` by this wave: state record -> LDS, frame_skip x RK4 mj_step, game rules, rewards, done, auto-reset,
// observation write, state record back (the whole of SumoEnv._step + the wrappers + the worker's auto-reset:
// sumo.py:120-202, sumo_env.py:40-72, monitor.py:51-78, subproc_vec_env.py:10-19).  Shared by the per-step launch
// (sumo_step_kernel) and the fused multi-step rollout launch (sumo_rollout_kernel).
template <bool COH = false, class C, class SA>
__device__ __forceinline__ void env_step_body(C& c, const SA& a, int e) {
  const auto& mdl = model_view(c);
  const int lane = c.lane;
  if (a.trace && lane == 0) a.trace[4 * e] = wall_clock64();
  const unsigned rec_sum = load_state<COH>(c, a, e);
#ifndef SUMO_NO_HANDCHECK
  if (COH) {   // fused rollout: the record must be the one the env's previous step of THIS launch published (tag and checksum)
    const int k0 = ((const int*)(S(stash) + 4))[1];   // parked by the ticket loop
    if (k0 > 0) {
      const int* cn = a.counters + 4 * e;
      const int tag = hand_load<true>(cn + 2);
      const unsigned chk = (unsigned)hand_load<true>(cn + 3);
      if ((tag != a.seq_base + k0 || chk != rec_sum) && lane == 0) {
        __hip_atomic_store(pt_global(a.abort_flag), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.stats) atomicAdd(a.stats + 10, 1ull);
      }
    }
  }
#endif
#ifdef SUMO_DBG_RELOAD_CTRL
  if (COH && lane < mdl.nu) {
    const float* act0 = a.actions + (size_t)e * 2 * a.act_stride;
    int ag = lane >= MI(agent_uadr)[1] ? 1 : 0;
    S(ctrl)[lane] = (double)hand_load<true>(act0 + ag * a.act_stride + (lane - MI(agent_uadr)[ag]));
  }
#endif
  if (!COH && lane < mdl.nu) {   // fused rollout: the policy phase has put the step's actions into S(ctrl) itself
    const float PT_GAS* act0 = pt_global(a.actions) + (size_t)e * 2 * a.act_stride;
    int ag = lane >= MI(agent_uadr)[1] ? 1 : 0;
    S(ctrl)[lane] = (double)act0[ag * a.act_stride + (lane - MI(agent_uadr)[ag])];
  }
  SYNC();
  // torso xy before the step (agents.py:216-217) is parked in LDS so nothing but the context stays live across the
  // twenty forward-dynamics evaluations (keeps the kernel within the two-waves-per-SIMD register budget)
  if (lane < 2) { int qa = MI(agent_qposadr)[lane]; S(stash)[2 * lane] = S(qpos)[qa]; S(stash)[2 * lane + 1] = S(qpos)[qa + 1]; }
  SYNC();
  PROF(17);
  // bad-value guard, entry half: a state that already fails the test (host-imposed, or left by a launch that was cut short) is
  // not integrated at all; the exit half sits in the epilogue.  Checked per env step, not per mj_step: a flag that lives across
  // the twenty forward-dynamics evaluations costs the register budget more than stepping a lost state to the end (all loops
  // of the pipeline are bounded; NaN comparisons fall through)
  // (nothing stays live for it: a skipped state is still bad when the exit half looks)
  if (!state_is_bad(c)) mj_steps(c, mdl.frame_skip);
  PROF(18);
  int* cnt = a.counters + 4 * e;
  int num_steps = hand_load<COH>(cnt), reset_count = hand_load<COH>(cnt + 1);
  double* st = a.state + (size_t)e * a.state_stride;
  double ep_ret = hand_load<COH>(st + mdl.nq + 2 * mdl.nv), ep_dense = hand_load<COH>(st + mdl.nq + 2 * mdl.nv + 1);
  double before[2][2] = {{S(stash)[0], S(stash)[1]}, {S(stash)[2], S(stash)[3]}};
  // ---- game rules (sumo.py:120-202), evaluated redundantly by every lane (wave-uniform result)
  double after[2][2], z[2];
  for (int g = 0; g < 2; g++) {
    int qa = MI(agent_qposadr)[g];
    after[g][0] = S(qpos)[qa]; after[g][1] = S(qpos)[qa + 1]; z[g] = S(qpos)[qa + 2];
  }
  num_steps++;
  const double lim = mdl.tatami_size + 0.1;
  const double adjz = c.P->adjust_z;   // sumo.py:147,157 read get_qpos(): the lose test sees the adjusted z
  int lost[2];
  for (int g = 0; g < 2; g++) {
    double mx = fabs(after[g][0]) > fabs(after[g][1]) ? fabs(after[g][0]) : fabs(after[g][1]);
    lost[g] = (z[g] + adjz < 0.29) || (mx >= lim);
  }
  const double dt = MF(opt)[SUMO_OPT_TIMESTEP] * mdl.frame_skip;
  int dn = 0;
  double inf[2][SUMO_INFO_STRIDE];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    int o = 1 - g, flags = 0;
    double ctrl_r = -0.1 * (double)sumsq_f32_ctrl(S(ctrl) + MI(agent_uadr)[g], MI(agent_nu)[g]);   // the actions as the step received them
    double lose = lost[g] ? -2000.0 : 0.0, win = lost[o] ? 2000.0 : 0.0;
    if (lost[g] || lost[o]) dn = 1;
    if (lost[o]) flags |= 1;
    double main_r = win + lose;
    if (num_steps > mdl.timestep_limit) { main_r += -1000.0; dn = 1; }
    double mv[2] = {(after[g][0] - before[g][0]) / dt, (after[g][1] - before[g][1]) / dt};
    double dir[2] = {after[o][0] - before[g][0], after[o][1] - before[g][1]};
    double nrm = sqrt(dir[0] * dir[0] + dir[1] * dir[1]);
    dir[0] /= nrm; dir[1] /= nrm;
    double proj = mv[0] * dir[0] + mv[1] * dir[1];
    double move = (proj > 0 ? proj : 0.0) * 0.1;
    double push = -10.0 * slow_exp(-sqrt(after[o][0] * after[o][0] + after[o][1] * after[o][1]));
    double shaping = ctrl_r + push + move;
    inf[g][0] = ctrl_r; inf[g][1] = lose; inf[g][2] = win; inf[g][3] = main_r; inf[g][4] = move; inf[g][5] = push;
    inf[g][6] = shaping; inf[g][7] = (double)flags;
  }
  // per-env divergence guard: a state that failed the bad-value test (during the stepping or after its last mj_step) ends the
  // episode with zero rewards and info flag 4; the auto-reset below replaces it, the engine counts it (sumo_stats[8])
  const int diverged = state_is_bad(c);
  c.st_diverged += diverged;
  if (diverged) {
#pragma unroll
    for (int g = 0; g < 2; g++) {
#pragma unroll
      for (int k = 0; k < SUMO_INFO_STRIDE; k++) inf[g][k] = 0.0;
      inf[g][7] = 4.0;
    }
    dn = 1;
  }
  ep_ret += inf[0][3] + inf[0][6];
  ep_dense += inf[0][6];
  if (!diverged && dn && inf[0][3] == -1000.0) { inf[0][7] += 2.0; inf[1][7] += 2.0; }
  if (lane == 0) {
    double* io = a.info + (size_t)e * 2 * SUMO_INFO_STRIDE;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int k = 0; k < SUMO_INFO_STRIDE; k++) hand_store<COH>(io + g * SUMO_INFO_STRIDE + k, inf[g][k]);
    if (COH) {   // what the rollout's post phase needs of this step stays in LDS (nothing written in a launch is read back from HBM)
      double* sh = S(stash) + 5;
      sh[0] = inf[0][6]; sh[1] = inf[0][3]; sh[2] = inf[1][6]; sh[3] = inf[1][3]; sh[4] = dn ? ep_ret : 0.0;
      ((int*)(sh + 5))[0] = dn ? num_steps : 0; ((int*)(sh + 5))[1] = dn;
    }
  }
  // both agents' flags as one 16-bit store (one coherent store on the hand-over path)
  if (lane == 0) hand_store<COH>((uint16_t*)(a.done + 2 * e), (uint16_t)(dn ? 0x0101 : 0));
  if (lane == 0) { hand_store<COH>(a.ep_r + e, dn ? ep_ret : 0.0); hand_store<COH>(a.ep_dr + e, dn ? ep_dense : 0.0); hand_store<COH>(a.ep_l + e, dn ? num_steps : 0); }
  if (dn) {  // subproc_vec_env.py:13-16: auto-reset, reset observation replaces the terminal one
    reset_state(c, pt_global(a.seeds)[e], (uint32_t)reset_count);
    reset_count++;
    num_steps = 0; ep_ret = 0; ep_dense = 0;
  }
  write_obs<COH>(c, a.obs + (size_t)e * 2 * a.obs_stride, a.obs_stride, num_steps);
  const unsigned rec_out = store_state<COH>(c, a, e);
  if (COH && lane == 0) {   // sequence tag + checksum of the record just written (the next wave checks them first)
    const int k1 = ((const int*)(S(stash) + 4))[1] + 1;
    hand_store<true>(cnt + 3, (int)(rec_out + (unsigned)(k1 == 1 && a.dbg_fault_env == e)));   // dbg_fault_env: injected fault (tests)
    hand_store<true>(cnt + 2, a.seq_base + k1);
  }
  if (lane == 0) {
    hand_store<COH>(cnt, num_steps); hand_store<COH>(cnt + 1, reset_count);
    hand_store<COH>(st + mdl.nq + 2 * mdl.nv, ep_ret); hand_store<COH>(st + mdl.nq + 2 * mdl.nv + 1, ep_dense);
  }
  PROF(19);
  // work estimate for the next launch's longest-first schedule (sumo_step): Newton iterations dominate the variation
  // (a contention-independent proxy; sorting by the measured cycle count of the previous step schedules no better)
  // (least-squares fit of measured wave times, tools/slot_trace.py: Newton iterations, contacts, dense-path forwards)
  if (lane == 0 && a.cost) { const int w = 1000 + 12 * c.st_newton + 10 * c.st_ncon + 60 * c.st_dense; pt_global(a.cost)[e] = w < 65535 ? w : 65535; }
  if (a.trace && lane == 0) {
    a.trace[4 * e + 1] = wall_clock64();
    a.trace[4 * e + 2] = (unsigned long long)c.st_newton | ((unsigned long long)c.st_ncon << 32);
    a.trace[4 * e + 3] = (unsigned long long)c.st_dense | ((unsigned long long)c.st_cross << 16) | ((unsigned long long)c.st_nefc << 32);
  }
}

template <int SL> struct LayoutSel { typedef Layout type; };
template <> struct LayoutSel<1> { typedef LayoutAntAnt type; };
template <> struct LayoutSel<2> { typedef LayoutSpiderSpider type; };

template <int NV, int SL = 0>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(SUMO_WPE_OF(NV), SUMO_WPE_OF(NV)))) sumo_step_kernel(const Params* P, StepArgs a) {
  Ctx<NV, typename LayoutSel<SL>::type> c;
  ctx_init(c, P, smem_dyn);
  const int lane = c.lane;
  if ((int)blockIdx.x < a.rank_blocks) { sched_rank(a.rank_cost, a.N, a.rank_perm, blockIdx.x, lane); return; }
  const int bid = (int)blockIdx.x - a.rank_blocks;
  if (bid >= a.N) return;
  const int e = a.perm ? a.perm[bid] : bid;
  if (a.perm && bid < (a.N >> 3)) __builtin_amdgcn_s_setprio(1);   // predicted-longest eighth of the launch: issue ahead of the SIMD mate
#ifdef SUMO_DBG_POISON_LDS
  poison_lds(P, lane);
#endif
#ifdef SUMO_DBG_DUMP
#ifdef SUMO_DBG_DUMP_ALL   /* every env dumps: the buffer holds [N][20][8][64] doubles */
  c.dbg = a.dbg_qacc ? a.dbg_qacc + (size_t)e * (20 * 8 * 64) : nullptr;
#else
  c.dbg = e == 0 ? a.dbg_qacc : nullptr;
#endif
#endif
  env_step_body(c, a, e);
  flush_stats(c, a.stats);
}

// ---------------------------------------------------------------------------------------------------------
// cfrc_mode = rne_post: the contact-force entries of the observations as a MuJoCo 2.1 WITH force sensors
// would show them.  The reference's scenes have no such sensor, so its cfrc_ext is zero and `zero` is the default (and what
// sumo_step writes); this optional second launch fills the entries in afterwards and costs most of another env step: it
// repeats the first frame_skip - 1 sub-steps from the pre-step state (saved by the host before sumo_step) to reach the state
// the LAST mj_step started from -- mj_forward evaluates sensors, hence mj_rnePostConstraint, once per mj_step at its start
// state; the RK4 sub-stages skip them -- evaluates forward() there, turns the solution into the pyramid-row forces
// f = -D min(J qacc - aref, 0), recomputes the contact geometry (the records shared the mass matrix's storage) and sums,
// per body, -wrench (body 1) / +wrench (body 2) about the subtree CoM of the body's agent: [torque ; force], world axes.
// The hot kernels are untouched by this mode.  Envs whose episode ended in the step keep the zeros of their reset observation.
// ---------------------------------------------------------------------------------------------------------
struct CfrcArgs {
  const double* prev_state;   // [N][state_stride]: the state before the step
  double* fbuf;               // [N][5 * maxcon] scratch: row forces | friction coefficients
  double* cfrc_out;           // [N][nbody][6] or NULL (tests)
};

template <int NV>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(SUMO_WPE_OF(NV), SUMO_WPE_OF(NV))))
sumo_cfrc_kernel(const Params* P, StepArgs a, CfrcArgs q) {
  Ctx<NV> c;
  ctx_init(c, P, smem_dyn);
  const sumo_model_t& mdl = P->mdl;
  const int e = blockIdx.x, lane = c.lane, nb = mdl.nbody;
  if (e >= a.N) return;
  if (q.cfrc_out) for (int i = lane; i < 6 * nb; i += WAVE) q.cfrc_out[(size_t)e * 6 * nb + i] = 0.0;
  if (a.done[2 * e]) return;
  StepArgs ap = a;
  ap.state = const_cast<double*>(q.prev_state);
  load_state(c, ap, e);
  if (lane < mdl.nu) {
    const float PT_GAS* act0 = pt_global(a.actions) + (size_t)e * 2 * a.act_stride;
    const int ag = lane >= MI(agent_uadr)[1] ? 1 : 0;
    S(ctrl)[lane] = (double)act0[ag * a.act_stride + (lane - MI(agent_uadr)[ag])];
  }
  SYNC();
  if (state_is_bad(c)) return;
  if (mdl.frame_skip > 1) mj_steps(c, mdl.frame_skip - 1);
  c.use_prev = 0;
  forward(c);
  // pyramid-row forces of the solution (x = qacc): rows 4 ci + k of contact ci; limit rows carry no external force
  const int ncon = c.ncon, maxcon = c.L.maxcon;
  double* fb = q.fbuf + (size_t)e * 5 * maxcon;
  contact_Jx(c, S(x));
  for (int r = lane; r < 4 * ncon; r += WAVE) {
    const double jar = row_Jx(c, r) - S(aref)[r];
    fb[r] = jar < 0 ? -S(D)[r >> 2] * jar : 0.0;
  }
  for (int ci = lane; ci < ncon; ci += WAVE) fb[4 * maxcon + ci] = S(cpar)[ci];
  __threadfence();
  SYNC();
  // contact points / frames / bodies again (same state, same order), subtree CoMs
  position_velocity(c);
  collision(c);
  const int nc2 = c.ncon < ncon ? c.ncon : ncon;   // (make_constraint may have cut the list to the Jacobian pool)
  KCONSTS();
  double w[6] = {0, 0, 0, 0, 0, 0};
  const int b = lane, ag = K.b_agent;   // one body per lane: build_aux refuses scenes with nbody > WAVE at sumo_create ("scene too large for one wavefront per env")
  if (b > 0 && b < nb) {
    const double* com = S(com) + 3 * ag;
    for (int ci = 0; ci < nc2; ci++) {
      const int* cb = c.si + c.L.con_b + 4 * ci;
      const int side = cb[1] == b ? 1 : (cb[0] == b ? 0 : -1);
      if (side < 0) continue;
      const double* cd = S(cond) + 14 * ci;
      const double f0 = fb[4 * ci], f1 = fb[4 * ci + 1], f2 = fb[4 * ci + 2], f3 = fb[4 * ci + 3], mu = fb[4 * maxcon + ci];
      const double fc[3] = {(f0 + f1) + (f2 + f3), mu * (f0 - f1), mu * (f2 - f3)};
      double F[3], arm[3], tq[3];
#pragma unroll
      for (int k = 0; k < 3; k++) { F[k] = cd[4 + k] * fc[0] + cd[7 + k] * fc[1] + cd[10 + k] * fc[2]; arm[k] = cd[1 + k] - com[k]; }
      cross3(tq, arm, F);
      const double sg = side ? 1.0 : -1.0;
#pragma unroll
      for (int k = 0; k < 3; k++) { w[k] += sg * tq[k]; w[3 + k] += sg * F[k]; }
    }
    if (q.cfrc_out)
      for (int k = 0; k < 6; k++) q.cfrc_out[((size_t)e * nb + b) * 6 + k] = w[k];
    // agents.py:192-208: |clip(cfrc_ext, +-100)| of the own bodies, then (for the other side) of this agent's torso
    const int i0 = MI(agent_bodyadr)[ag], o = 1 - ag;
    float* ob = a.obs + (size_t)e * 2 * a.obs_stride;
    float* own = ob + ag * a.obs_stride + MI(agent_nq)[ag] + MI(agent_nv)[ag] + 6 * (b - i0);
#pragma unroll
    for (int k = 0; k < 6; k++) own[k] = (float)fabs(fmin(fmax(w[k], -100.0), 100.0));
    if (b == i0) {
      float* opp = ob + o * a.obs_stride + MI(agent_nq)[o] + MI(agent_nv)[o] + 6 * MI(agent_nbody)[o] + 7;
#pragma unroll
      for (int k = 0; k < 6; k++) opp[k] = (float)fabs(fmin(fmax(w[k], -100.0), 100.0));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Fused rollout: K consecutive self-play rollout steps of an env in ONE launch (reference runner.py:62-151 for MLP(64,64)
// policies).  The wave that owns env e evaluates the five policy / value passes of a step itself -- the env's two
// observations are rows 0 and 1 of an MFMA tile, the three trunks (learner policy, opponent policy, learner value) run through
// the same device functions as the batched PPO kernels, so every number equals the stepwise path bit for bit -- records
// obs / actions / values / neglogps / done flags straight into the [agent][time][env] rollout buffers, steps its env and
// records the mixed reward and the episode statistics.  No launch boundary, no policy launch and no per-step barrier over
// the envs remain between two env steps; each env may face its own frozen opponent snapshot (opp_idx).
// ---------------------------------------------------------------------------------------------------------
#ifdef SUMO_POLICY_PROBE   /* development build: the per-env phase clock also splits the MLP policy phase (tools/fused_probe.py --probe) */
#define PROF_STRIDE 12
#define PPROBE(slot) do { if (r.prof && lane == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(r.prof + PROF_STRIDE * e + (slot), t_ - tp_); tp_ = t_; } } while (0)
#else
#define PROF_STRIDE 4
#define PPROBE(slot) do { } while (0)
#endif
struct RolloutArgs {
  const float *learner, *opponent;   // flat parameter vectors; opponent: [npool][P]
  const int32_t* opp_idx;            // [N] snapshot per env or NULL
  const float *noise0, *noise1;      // [T][N][A]
  float *obs, *act, *rew, *val, *nlp, *onlp;   // [2][T][Ntot][...]
  uint8_t *done, *ep_done;           // [2][T][Ntot], [T][Ntot]
  double* ep_r;                      // [T][Ntot]
  int32_t* ep_l;                     // [T][Ntot]
  double alpha;
  int T, Ntot, env_offset, s0, K, XS, lds_off;   // lds_off: LDS offset (doubles) of the policy scratch (the mass-matrix region)
  int* sched;                        // ticket counter | abort flag | finished steps per env [N] (zeroed by the host per launch)
  unsigned long long* prof;          // development (sumo_debug_trace): [N][4] wave start, end, ticks in the policy phases, ticks in the env steps (100 MHz)
  ParamLayout L;
  // recurrent policies (sumo_rollout_steps_lstm): the learner's net, the opponent snapshots (device array) with the snapshot of
  // every 16-env tile of the whole env set (NULL: snapshot 0), and the acting nets' states [N][2 hidden] (c | h) per agent
  ppo_lstm_net lnet;
  const ppo_lstm_net* onets;
  const int32_t* tile_net;
  float *st0, *st1;
};

template <class C, class SA, class RA>
__device__ __forceinline__ void rollout_policy_phase(C& c, const SA& a, const RA& r, int e, int s) {
  const int lane = c.lane, i = lane & 15, kq = lane >> 4;
  const int D = r.L.D, A = r.L.A, XS = r.XS;
  float* xbuf = (float*)(c.sm + r.lds_off);        // [2][XS] | h1 [2][PT_HS] | h2 [2][PT_HS]
  float* h1 = xbuf + 2 * XS;
  float* h2 = h1 + 2 * PT_HS;
  const size_t col = (size_t)r.env_offset + e;
  const size_t slot0 = ((size_t)0 * r.T + s) * r.Ntot + col, slot1 = ((size_t)1 * r.T + s) * r.Ntot + col;
  // the env's two observations: into the tile and into the rollout record (runner.py:98-101)
  const float* ob = a.obs + (size_t)e * 2 * a.obs_stride;
#ifdef SUMO_POLICY_PROBE
  unsigned long long tp_ = wall_clock64();
#endif
  for (int k = lane; k < D; k += WAVE) {
    const float o0 = hand_load<true>(ob + k), o1 = hand_load<true>(ob + a.obs_stride + k);
    xbuf[k] = o0; xbuf[XS + k] = o1;
    pt_global(r.obs)[slot0 * D + k] = o0; pt_global(r.obs)[slot1 * D + k] = o1;
  }
  if (lane < 2) pt_global(r.done)[lane == 0 ? slot0 : slot1] = (uint8_t)(hand_load<true>((const uint16_t*)(a.done + 2 * e)) >> (8 * lane));
  wave_sync();
  const float PT_GAS* lp = pt_global(r.learner);
  const float PT_GAS* op = pt_global(r.opponent) + (size_t)(r.opp_idx ? pt_global(r.opp_idx)[e] : 0) * r.L.P;
  PPROBE(4);
  const f32x4 mL = trunk_forward<false, 2>(pi_net((const float*)lp, r.L), xbuf, XS, D, h1, h2, lane);
  wave_sync();
  PPROBE(5);
  const f32x4 mO = trunk_forward<false, 2>(pi_net((const float*)op, r.L), xbuf, XS, D, h1, h2, lane);
  wave_sync();
  PPROBE(6);
  const f32x4 vL = trunk_forward<false, 2>(vf_net((const float*)lp, r.L), xbuf, XS, D, h1, h2, lane);
  PPROBE(7);
  // heads: row 0 = agent 0 (learner acts, opponent scores), row 1 = agent 1 (opponent acts, learner scores and values)
  const bool colk = i < A;
  const float lsL = colk ? lp[r.L.logstd + i] : 0.0f, lsO = colk ? op[r.L.logstd + i] : 0.0f;
  const float stdL = expf(lsL), stdO = expf(lsO);
  const float sumL = row16_sum(lsL), sumO = row16_sum(lsO);
  const bool ok = colk && kq == 0;                  // rows 0 and 1 live in the first 16 lanes (D layout: rows 4 kq + r)
  const size_t nz = ((size_t)s * a.N + e) * A + i;
  const float n0 = ok ? pt_global(r.noise0)[nz] : 0.0f, n1 = ok ? pt_global(r.noise1)[nz] : 0.0f;
  float act0 = 0.0f, act1 = 0.0f;
  const float nlp0 = gauss_row(mL[0], stdL, sumL, ok, true, n0, act0, A);      // learner samples for agent 0 ...
  const float onlp0 = gauss_row(mO[0], stdO, sumO, ok, false, 0.0f, act0, A);  // ... the opponent net scores that action
  const float onlp1 = gauss_row(mO[1], stdO, sumO, ok, true, n1, act1, A);     // opponent samples for agent 1 ...
  const float nlp1 = gauss_row(mL[1], stdL, sumL, ok, false, 0.0f, act1, A);   // ... the learner scores it
  if (ok) {
    pt_global(r.act)[slot0 * A + i] = act0; pt_global(r.act)[slot1 * A + i] = act1;
    float* ae = const_cast<float*>(a.actions) + (size_t)e * 2 * a.act_stride;
    hand_store<true>(ae + i, act0); hand_store<true>(ae + a.act_stride + i, act1);     // the env's action buffer (output only)
    const auto& mdl = model_view(c);
    S(ctrl)[MI(agent_uadr)[0] + i] = (double)act0; S(ctrl)[MI(agent_uadr)[1] + i] = (double)act1;   // the step's control vector
  }
  if (lane == 0) {
    pt_global(r.nlp)[slot0] = nlp0; pt_global(r.nlp)[slot1] = nlp1; pt_global(r.onlp)[slot0] = onlp0; pt_global(r.onlp)[slot1] = onlp1;
    pt_global(r.val)[slot0] = vL[0]; pt_global(r.val)[slot1] = vL[1];
  }
  PPROBE(8);
  wave_sync();   // the action buffer is read back by the env step (other lanes), the scratch region becomes the mass matrix again
}

// The same phase for recurrent policies (baselines lstm(128) with shared value head; Runner._step_group's recurrent branch,
// reference runner.py:62-96 with the S / M feeds of models.py:163-170).  Five evaluations per step, two passes over weights:
//   opponent net: row A = (obs 1, agent 1's state)  -> action 1, its neglogp (opponent_neglogp), agent 1's NEW state
//                 row B = (obs 0, zero state)        -> the opponent's likelihood of action 0 (runner.py:85 feeds no state)
//   learner net:  row C = (obs 0, agent 0's state)  -> action 0, neglogp, value, agent 0's new state
//                 row D = (obs 1, agent 1's new state, masked again) -> the value recorded for agent 1 (runner.py:93)
//                 row E = (obs 1, zero state)        -> the learner's likelihood of action 1
// Gate pre-activations run on the vector ALU in the MFMA tiles' accumulation order (lstm_gates_valu), cell update and heads
// through the functions ppo_lstm_step_kernel uses: every number equals the launch-per-evaluation path bit for bit.
// LDS (floats, from lds_off): x [2][XS] | zero row [NH] | previous latents [3][NH] | new latents [3][NH].
template <int NH, class C, class SA, class RA>
__device__ __forceinline__ void rollout_policy_phase_lstm(C& c, const SA& a, const RA& r, int e, int s) {
  const int lane = c.lane;
  const auto& NL = r.lnet;
  const int D = NL.ob_dim, A = NL.ac_dim, XS = r.XS;
  float* xo = (float*)(c.sm + r.lds_off);
  float* hz = xo + 2 * XS;
  float* hp = hz + NH;
  float* hn = hp + 3 * NH;
  const size_t col = (size_t)r.env_offset + e;
  const size_t slot0 = ((size_t)0 * r.T + s) * r.Ntot + col, slot1 = ((size_t)1 * r.T + s) * r.Ntot + col;
  const float* ob = a.obs + (size_t)e * 2 * a.obs_stride;
  for (int k = lane; k < XS; k += WAVE) {
    float o0 = 0.0f, o1 = 0.0f;
    if (k < D) {
      o0 = hand_load<true>(ob + k); o1 = hand_load<true>(ob + a.obs_stride + k);
      pt_global(r.obs)[slot0 * D + k] = o0; pt_global(r.obs)[slot1 * D + k] = o1;
    }
    xo[k] = o0; xo[XS + k] = o1;
  }
  const unsigned dn = hand_load<true>((const uint16_t*)(a.done + 2 * e));   // done flags of the previous step = the masks M
  if (lane < 2) pt_global(r.done)[lane == 0 ? slot0 : slot1] = (uint8_t)(dn >> (8 * lane));
  const float keep0 = 1.0f - (float)(dn & 0xff), keep1 = 1.0f - (float)((dn >> 8) & 0xff);
  // the acting nets' states: lane owns the units 2 lane, 2 lane + 1
  float* s0p = r.st0 + (size_t)e * 2 * NH;
  float* s1p = r.st1 + (size_t)e * 2 * NH;
  const int j0 = 2 * lane;
  float c0[2], c1[2], cD[2];
  {
    union { unsigned long long u; float f[2]; } q;
    q.u = hand_load<true>((const unsigned long long*)(s0p + j0)); c0[0] = q.f[0] * keep0; c0[1] = q.f[1] * keep0;
    q.u = hand_load<true>((const unsigned long long*)(s1p + j0)); c1[0] = q.f[0] * keep1; c1[1] = q.f[1] * keep1;
    q.u = hand_load<true>((const unsigned long long*)(s0p + NH + j0)); hp[NH + j0] = q.f[0] * keep0; hp[NH + j0 + 1] = q.f[1] * keep0;
    q.u = hand_load<true>((const unsigned long long*)(s1p + NH + j0)); hp[j0] = q.f[0] * keep1; hp[j0 + 1] = q.f[1] * keep1;
    hz[j0] = 0.0f; hz[j0 + 1] = 0.0f;
  }
  wave_sync();
  const ppo_lstm_net PT_GAS* NOp = pt_global(r.onets) + (r.tile_net ? pt_global(r.tile_net)[col >> 4] : 0);
  const bool ok = lane < A;
  const size_t nz = ((size_t)s * a.N + e) * A + lane;
  float act0 = 0.0f, act1 = 0.0f, onlp1, mOB, stdO, sumO;
  {  // ---- opponent net: rows A, B
    const float *owx = NOp->wx, *owh = NOp->wh;
    const float PT_GAS* ob_ = pt_global(NOp->b);
    const float fb = NOp->forget_bias;
    float z[4][2][2], bz[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) { bz[g][0] = ob_[g * NH + j0]; bz[g][1] = ob_[g * NH + j0 + 1]; }   // (in flight during the gate sums)
    const float* const xr[2] = {xo + XS, xo};
    const float* const hr[2] = {hp, hz};
    lstm_gates_valu<NH, 2>(owx, owh, D, xr, hr, lane, z);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = j0 + u;
      const float bi = bz[0][u], bf = bz[1][u] + fb, bo = bz[2][u], bu = bz[3][u];   // gate order i, f, o, u
      const LstmCell ca = lstm_cell(z[0][u][0], z[1][u][0], z[2][u][0], z[3][u][0], bi, bf, bo, bu, c1[u]);
      const LstmCell cb = lstm_cell(z[0][u][1], z[1][u][1], z[2][u][1], z[3][u][1], bi, bf, bo, bu, hz[j]);
      hand_store<true>(s1p + j, ca.cn); hand_store<true>(s1p + NH + j, ca.hn);      // agent 1's state after the opponent's step
      hn[j] = ca.hn; hn[NH + j] = cb.hn;
      cD[u] = ca.cn * keep1; hp[2 * NH + j] = ca.hn * keep1;                         // row D: masked once more by the same flag
    }
    wave_sync();
    float m[2];
    lstm_heads_valu<NH, 2>(NOp->head_w, NOp->vf_w, A, hn, lane, m);
    const float hb = ok ? pt_global(NOp->head_b)[lane] : 0.0f, ls = ok ? pt_global(NOp->logstd)[lane] : 0.0f;
    stdO = expf(ls); sumO = row16_sum(ls);
    mOB = m[1] + hb;
    const float n1 = ok ? pt_global(r.noise1)[nz] : 0.0f;
    onlp1 = gauss_row(m[0] + hb, stdO, sumO, ok, true, n1, act1, A);     // the opponent samples for agent 1
    wave_sync();   // the latent rows are rewritten by the learner's pass
  }
  float nlp0, nlp1, onlp0, v0, v1;
  {  // ---- learner net: rows C, D, E
    const float fb = NL.forget_bias;
    float z[4][2][3], bz[4][2];
#pragma unroll
    for (int g = 0; g < 4; g++) { bz[g][0] = pt_global(NL.b)[g * NH + j0]; bz[g][1] = pt_global(NL.b)[g * NH + j0 + 1]; }
    const float* const xr[3] = {xo, xo + XS, xo + XS};
    const float* const hr[3] = {hp + NH, hp + 2 * NH, hz};
    lstm_gates_valu<NH, 3>(NL.wx, NL.wh, D, xr, hr, lane, z);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = j0 + u;
      const float bi = bz[0][u], bf = bz[1][u] + fb, bo = bz[2][u], bu = bz[3][u];
      const LstmCell cc = lstm_cell(z[0][u][0], z[1][u][0], z[2][u][0], z[3][u][0], bi, bf, bo, bu, c0[u]);
      const LstmCell cd = lstm_cell(z[0][u][1], z[1][u][1], z[2][u][1], z[3][u][1], bi, bf, bo, bu, cD[u]);
      const LstmCell ce = lstm_cell(z[0][u][2], z[1][u][2], z[2][u][2], z[3][u][2], bi, bf, bo, bu, hz[j]);
      hand_store<true>(s0p + j, cc.cn); hand_store<true>(s0p + NH + j, cc.hn);      // agent 0's state after the learner's step
      hn[j] = cc.hn; hn[NH + j] = cd.hn; hn[2 * NH + j] = ce.hn;
    }
    wave_sync();
    float m[3];
    lstm_heads_valu<NH, 3>(NL.head_w, NL.vf_w, A, hn, lane, m);
    const float hb = ok ? pt_global(NL.head_b)[lane] : 0.0f, ls = ok ? pt_global(NL.logstd)[lane] : 0.0f;
    const float stdL = expf(ls), sumL = row16_sum(ls);
    const float vb = pt_global(NL.vf_b)[0];
    v0 = __shfl(m[0], 16) + vb; v1 = __shfl(m[1], 16) + vb;
    const float n0 = ok ? pt_global(r.noise0)[nz] : 0.0f;
    nlp0 = gauss_row(m[0] + hb, stdL, sumL, ok, true, n0, act0, A);      // the learner samples for agent 0 ...
    onlp0 = gauss_row(mOB, stdO, sumO, ok, false, 0.0f, act0, A);        // ... the opponent net (zero state) scores that action
    nlp1 = gauss_row(m[2] + hb, stdL, sumL, ok, false, 0.0f, act1, A);   // the learner (zero state) scores the opponent's action
  }
  if (ok) {
    pt_global(r.act)[slot0 * A + lane] = act0; pt_global(r.act)[slot1 * A + lane] = act1;
    float* ae = const_cast<float*>(a.actions) + (size_t)e * 2 * a.act_stride;
    hand_store<true>(ae + lane, act0); hand_store<true>(ae + a.act_stride + lane, act1);
    const auto& mdl = model_view(c);
    S(ctrl)[MI(agent_uadr)[0] + lane] = (double)act0; S(ctrl)[MI(agent_uadr)[1] + lane] = (double)act1;
  }
  if (lane == 0) {
    pt_global(r.nlp)[slot0] = nlp0; pt_global(r.nlp)[slot1] = nlp1; pt_global(r.onlp)[slot0] = onlp0; pt_global(r.onlp)[slot1] = onlp1;
    pt_global(r.val)[slot0] = v0; pt_global(r.val)[slot1] = v1;
  }
  wave_sync();
}

template <class C, class SA, class RA>
__device__ __forceinline__ void rollout_post_phase(C& c, const SA& a, const RA& r, int e, int s) {
  SYNC();   // the step's reward inputs and episode record were parked in LDS by lane 0 of the epilogue
  const int lane = c.lane;
  const size_t col = (size_t)r.env_offset + e;
  if (lane < 2) {   // runner.py:134 + monitor.py:63-78 harvest, as ppo_post_step_kernel
    const double* sh = S(stash) + 5;
    pt_global(r.rew)[((size_t)lane * r.T + s) * r.Ntot + col] = reward_mix(r.alpha, sh[2 * lane], sh[2 * lane + 1]);
    if (lane == 0) {
      const size_t t = (size_t)s * r.Ntot + col;
      const int* rec = (const int*)(sh + 5);
      pt_global(r.ep_done)[t] = (uint8_t)rec[1]; pt_global(r.ep_r)[t] = sh[4]; pt_global(r.ep_l)[t] = rec[0];
    }
  }
}

// The launch arguments are re-read through a laundered pointer in every iteration: accessed as ordinary by-value kernel
// arguments their loads are loop-invariant, get hoisted out of the step loop and stay live across the twenty forward-dynamics
// evaluations of every step (the kernel sits exactly at its 256-register budget).  The pointer addresses the kernel-argument
// segment itself, where the runtime has placed the struct at launch (no separate copy to keep alive).
struct RolloutLaunch { StepArgs a; RolloutArgs r; };

// Scheduling: the launch is a set of persistent waves (one per wave slot of the chip) that draw TICKETS from a global counter;
// ticket t is step t / N of env t % N.  Env steps differ in cost by 3x (contacts, Newton iterations, agents wrestling), so
// binding a wave to one env for the whole launch leaves a fifth of the slots idle behind the slowest envs (measured: 40-step
// wave lifetimes of 34 ms mean, 50 ms max); with tickets a wave that drew a long step simply draws fewer of them.  A step needs
// its env's previous step: prog[e] counts the finished steps of env e, the wave that draws (e, k) waits for prog[e] == k
// (rarely: that ticket was handed out N tickets earlier).  Dependencies only point to earlier tickets, which are held by waves
// that are already running, so the scheme cannot deadlock whatever the number of resident waves; every wave leaves when the
// tickets run out.  Hand-over of an env between waves: the few hundred bytes that cross (hand_load / hand_store above) go
// through sc1 accesses, the writer drains them (s_waitcnt vmcnt(0)) before its relaxed agent-scope store of prog[e], the
// next wave polls prog[e] with one lane and then reads the record with sc1 loads.  The wait is bounded: a wave that saw no progress for ~2 s raises the
// launch's abort flag (counted in sumo_stats[9]) and every wave drains.
#define ROLLOUT_SPIN_LIMIT (1u << 22)   /* polls of ~0.5 us each */

template <int NV, int POLICY, int SL = 0>   // POLICY 0: MLP(64,64) policy / value nets; 1: LSTM(128) with shared value head; SL 1: static Layout
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(SUMO_WPE_OF(NV), SUMO_WPE_OF(NV))))
sumo_rollout_kernel(const Params* P, RolloutLaunch launch_args) {
  // `launch_args` is read in place from the kernel-argument segment (second argument, 8-byte aligned right behind P) through a
  // pointer the loop launders per ticket -- see the note above RolloutLaunch
  (void)launch_args;
  typedef const RolloutLaunch PT_CAS* LaunchPtr;   // constant address space: every field read is a scalar load
  const LaunchPtr LP = (LaunchPtr)((const char PT_CAS*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(const Params*));
  Ctx<NV, typename LayoutSel<SL>::type> c;
  ctx_init(c, P, smem_dyn);
  for (;;) {
    LaunchPtr lp = launder_sptr(LP);
    int* sched = lp->r.sched;                      // [0] ticket counter, [1] abort flag, [2 + e] finished steps of env e
    const int N = lp->a.N;
    int t = 0;
    if (c.lane == 0) {
      t = __hip_atomic_fetch_add(sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(sched + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) t = 0x7FFFFFFF;   // launch aborted: drain
    }
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= N * lp->r.K) break;
    int e = t % N, k = t / N;
    if (k > 0) {
      int ok = 1;
      if (c.lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(sched + 2 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > ROLLOUT_SPIN_LIMIT || __hip_atomic_load(sched + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
        }
        if (!ok) __hip_atomic_store(sched + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ok = __builtin_amdgcn_readfirstlane(ok);
      if (!ok) { if (c.lane == 0 && lp->a.stats) atomicAdd(lp->a.stats + 9, 1ull); break; }
      asm volatile("" ::: "memory");   // the env's record is read after the poll, through sc1 loads (hand_load): nothing to invalidate
    }
    __builtin_amdgcn_s_setprio(0);   // a wave that raised its issue priority for a dense-solver step (forward()) starts the next ticket level
    // (e, k) wait in LDS during the phases (nothing but the context stays live across the forward-dynamics evaluations)
#ifdef SUMO_DBG_POISON_LDS
    poison_lds(P, c.lane);
#endif
    if (c.lane == 0) { int* tk = (int*)(S(stash) + 4); tk[0] = e; tk[1] = k; }
    lp = launder_sptr(LP);
    int s = lp->r.s0 + k;
    // development (sumo_debug_trace): per-env phase clock, accumulated with atomics (an env's steps run on many waves);
    // the caller zeroes the buffer: [4e] first start, [4e+1] last end, [4e+2] ticks in policy phases, [4e+3] ticks in env steps
    unsigned long long* prof = lp->r.prof;
    unsigned long long t0 = 0;
    if (prof && c.lane == 0) { t0 = wall_clock64(); if (k == 0) atomicExch(prof + PROF_STRIDE * e, t0); }
#ifdef SUMO_DBG_DUMP
    c.dbg = (e == 0 && k == 0) ? lp->a.dbg_qacc : nullptr;
    if (c.dbg) c.st_forward = 0;
#endif
    if constexpr (POLICY == 1) rollout_policy_phase_lstm<128>(c, lp->a, lp->r, e, s);
    else rollout_policy_phase(c, lp->a, lp->r, e, s);
#ifdef SUMO_DBG_HARD_BARRIER
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    prof = launder_sptr(LP)->r.prof;
    if (prof && c.lane == 0) { const unsigned long long t1 = wall_clock64(); atomicAdd(prof + PROF_STRIDE * e + 2, t1 - t0); S(stash)[11] = __longlong_as_double((long long)t1); }
    env_step_body<true>(c, launder_sptr(LP)->a, e);
    SYNC();
    { const int* tk = (const int*)(S(stash) + 4); e = __builtin_amdgcn_readfirstlane(tk[0]); k = __builtin_amdgcn_readfirstlane(tk[1]); }
    asm volatile("" : "+s"(e), "+s"(k));
    lp = launder_sptr(LP);
    s = lp->r.s0 + k;
    rollout_post_phase(c, lp->a, lp->r, e, s);
    prof = lp->r.prof;
    if (prof && c.lane == 0) {
      const unsigned long long t2 = wall_clock64(), t1 = (unsigned long long)__double_as_longlong(S(stash)[11]);
      atomicAdd(prof + PROF_STRIDE * e + 3, t2 - t1); atomicMax(prof + PROF_STRIDE * e + 1, t2);
    }
    // hand the env over: every sc1 store of the step has reached memory (vmcnt = 0), then the progress counter
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (c.lane == 0) __hip_atomic_store(launder_sptr(LP)->r.sched + 2 + e, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  flush_stats(c, launder_sptr(LP)->a.stats);
}

template <int NV>
__global__ void __launch_bounds__(WAVE) sumo_reset_kernel(const Params* P, StepArgs a) {
  Ctx<NV> c;
  ctx_init(c, P, smem_dyn);
  const sumo_model_t& mdl = P->mdl;
  const int e = blockIdx.x, lane = c.lane;
  if (e >= a.N) return;
  if (a.mask && !a.mask[e]) return;
  int* cnt = a.counters + 4 * e;
  int reset_count = cnt[1];
  reset_state(c, a.seeds[e], (uint32_t)reset_count);
  if (a.obs) write_obs(c, a.obs + (size_t)e * 2 * a.obs_stride, a.obs_stride, 0);
  store_state(c, a, e);
  if (lane == 0) {
    double* st = a.state + (size_t)e * a.state_stride;
    cnt[0] = 0; cnt[1] = reset_count + 1;
    st[mdl.nq + 2 * mdl.nv] = 0; st[mdl.nq + 2 * mdl.nv + 1] = 0;
  }
}

template <int NV>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(SUMO_WPE_OF(NV), SUMO_WPE_OF(NV)))) sumo_forward_kernel(const Params* P, StepArgs a) {
  Ctx<NV> c;
  ctx_init(c, P, smem_dyn);
  const sumo_model_t& mdl = P->mdl;
  const int e = blockIdx.x, lane = c.lane;
  if (e >= a.N) return;
  load_state(c, a, e);
  if (lane < mdl.nu) S(ctrl)[lane] = a.dbg_ctrl[(size_t)e * mdl.nu + lane];
  SYNC();
  forward(c);
  if (lane < mdl.nv) a.dbg_qacc[(size_t)e * mdl.nv + lane] = S(x)[lane];
  if (lane == 0) {
    int* o = a.dbg_counts + 4 * e;
    o[0] = c.ncon; o[1] = c.nefc; o[2] = c.st_newton; o[3] = c.ndropped;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
#ifdef SUMO_DEV_NV  /* development builds: compile a single kernel variant */
#define SUMO_FOR_NV(X) X(SUMO_DEV_NV)
#else
#define SUMO_FOR_NV(X) X(28) X(32) X(36) X(40) X(44)
#endif
static thread_local char g_err[512];
extern "C" const char* sumo_last_error(void) { return g_err; }
#define FAIL(code, ...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return code; } while (0)
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) FAIL(-100, "%s failed: %s", #expr, hipGetErrorString(_e)); } while (0)

struct sumo_engine {
  int device, N;
  std::vector<char> blob;
  sumo_model_t hm;  // host view
  sumo_model_t dm;  // device view
  Aux aux;
  Layout L;
  Params* d_params = nullptr;
  LaneRec* d_lanes = nullptr;
  int* d_pair_rec = nullptr;
  float* d_pair_bound = nullptr;
  std::vector<LaneRec> lanes;
  std::vector<int> pair_rec;
  std::vector<float> pair_bound;
  void* d_blob = nullptr;
  int* d_ai = nullptr;
  double* d_af = nullptr;
  signed char* d_pic = nullptr;
  double* d_state = nullptr;
  int cfrc_mode = 0;                       // 0 zero (reference behaviour), 1 rne_post (sumo_set_cfrc_mode)
  double adjust_z = 0.0;                   // Agent._adjust_z (sumo_set_adjust_z)
  double *d_state_prev = nullptr, *d_fbuf = nullptr, *d_cfrc = nullptr;   // rne_post: pre-step state, row-force scratch, cfrc_ext [N][nbody][6]
  int* d_counters = nullptr;
  uint64_t* d_seeds = nullptr;
  unsigned long long* d_stats = nullptr;
  // longest-first scheduling of the env steps (see sumo_step)
  int *d_cost = nullptr, *d_cost_sorted = nullptr, *d_iota = nullptr, *d_perm = nullptr;   // d_cost / d_perm: two buffers of N each
  unsigned long long* d_trace = nullptr;   // sumo_debug_trace
  int num_cus = 0;                         // cached device property (sumo_rollout_steps sizes its persistent grid with it)
  int* d_rsched = nullptr;                 // sumo_rollout_steps: ticket counter, abort flag, per-env progress [2 + N]
  unsigned rollout_seq = 0;                // fused launches so far (hand-over tags: seq_base = (rollout_seq & 0x7FFF) << 16)
  hipStream_t rollout_stream = nullptr;    // stream of the most recent fused launch (sumo_rollout_status waits on it)
  long long rollout_tickets = -1;          // N * K of the most recent fused launch, -1: none yet
  int dbg_fault_env = -1;                  // sumo_debug_fault
  double* dbg_dump = nullptr;              // sumo_debug_dump (SUMO_DBG_DUMP builds)
  int static_layout = 0;                   // 1 / 2: the scene's Layout equals a compile-time table (Ant-vs-Ant / Spider-vs-Spider): static kernel variants
  unsigned long long acked_faults = 0;     // stats[9] + stats[10] already reported by sumo_rollout_status
  long long sched_t = 0;                                                                     // step launches so far (in-kernel ranking)
  void* d_sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  bool sched = true, perm_valid = false;
  int state_stride, obs_stride, act_stride;
};

static void quat2mat_h(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}

static int build_aux(sumo_engine* E, std::vector<int>& ai, std::vector<double>& af, std::vector<signed char>& pic) {
  const sumo_model_t* m = &E->hm;
  int nb = m->nbody, nv = m->nv;
  if (nb > WAVE || nv > WAVE || m->njnt > WAVE || m->nq > WAVE) FAIL(-10, "scene too large for one wavefront per env (nbody %d nv %d)", nb, nv);
  if (m->nagent != 2) FAIL(-11, "exactly two agents expected");
  if (nv != 28 && nv != 32 && nv != 36 && nv != 40 && nv != 44) FAIL(-18, "no kernel variant for nv=%d", nv);
  const int* tparent = SUMO_I(m, body_parentid);
  // "Effective" tree of the kernel's level-by-level passes: a body whose parent carries no joint is re-attached to the nearest
  // ancestor that does, with the static transforms composed (the jointless body keeps its own frame as a leaf: its frame
  // still feeds its geom and inertia, and subtree sums do not care which rigidly connected ancestor collects it).  For the
  // RoboSumo agents (torso -> jointless leg stub -> hip body -> ankle body) this removes one level from every pass.
  std::vector<int> eparent(nb, 0);
  std::vector<double> epos(3 * (size_t)nb, 0.0), equat(4 * (size_t)nb, 0.0);
  for (int b = 0; b < nb; b++) {
    double pos[3] = {SUMO_F(m, body_pos)[3 * b], SUMO_F(m, body_pos)[3 * b + 1], SUMO_F(m, body_pos)[3 * b + 2]};
    double q[4] = {SUMO_F(m, body_quat)[4 * b], SUMO_F(m, body_quat)[4 * b + 1], SUMO_F(m, body_quat)[4 * b + 2], SUMO_F(m, body_quat)[4 * b + 3]};
    int pp = b ? tparent[b] : 0;
    while (pp != 0 && SUMO_I(m, body_jntnum)[pp] == 0) {
      const double* pq = SUMO_F(m, body_quat) + 4 * pp;
      const double* ppos = SUMO_F(m, body_pos) + 3 * pp;
      double R[9];
      quat2mat_h(R, pq);
      double np_[3] = {ppos[0] + R[0] * pos[0] + R[1] * pos[1] + R[2] * pos[2], ppos[1] + R[3] * pos[0] + R[4] * pos[1] + R[5] * pos[2],
                       ppos[2] + R[6] * pos[0] + R[7] * pos[1] + R[8] * pos[2]};
      double nq[4] = {pq[0] * q[0] - pq[1] * q[1] - pq[2] * q[2] - pq[3] * q[3], pq[0] * q[1] + pq[1] * q[0] + pq[2] * q[3] - pq[3] * q[2],
                      pq[0] * q[2] - pq[1] * q[3] + pq[2] * q[0] + pq[3] * q[1], pq[0] * q[3] + pq[1] * q[2] - pq[2] * q[1] + pq[3] * q[0]};
      double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
      for (int k = 0; k < 3; k++) pos[k] = np_[k];
      for (int k = 0; k < 4; k++) q[k] = nq[k] / nn;
      pp = tparent[pp];
    }
    eparent[b] = pp;
    for (int k = 0; k < 3; k++) epos[3 * b + k] = pos[k];
    for (int k = 0; k < 4; k++) equat[4 * b + k] = q[k];
  }
  const int* parent = getenv("SUMO_TRUE_TREE") && atoi(getenv("SUMO_TRUE_TREE")) ? tparent : eparent.data();
  const bool folded = parent != tparent;
  std::vector<int> depth(nb, 0);
  int ndepth = 1;
  for (int b = 1; b < nb; b++) { depth[b] = depth[parent[b]] + 1; if (depth[b] + 1 > ndepth) ndepth = depth[b] + 1; }
  Aux& A = E->aux;
  A.ndepth = ndepth;
  auto push_tbl = [&](const std::vector<int>& v) { int o = (int)ai.size(); ai.insert(ai.end(), v.begin(), v.end()); return o; };
  std::vector<int> lvl_adr(ndepth + 1, 0), lvl_body;
  for (int d = 0; d < ndepth; d++) { lvl_adr[d] = (int)lvl_body.size(); for (int b = 0; b < nb; b++) if (depth[b] == d) lvl_body.push_back(b); }
  lvl_adr[ndepth] = (int)lvl_body.size();
  // jointless leaves of the effective tree ("stubs", e.g. the fixed upper leg segments): rigidly attached to their parent, so
  // their inertia is added to the parent's once per evaluation and they take no part in the subtree gathers
  std::vector<int> is_stub(nb, 0);
  if (folded)
    for (int b = 1; b < nb; b++) {
      if (SUMO_I(m, body_jntnum)[b] != 0 || parent[b] == 0) continue;
      bool leaf = true;
      for (int ch = 1; ch < nb; ch++) if (ch != b && parent[ch] == b) leaf = false;
      is_stub[b] = leaf;
    }
  std::vector<int> child_adr(nb + 1, 0), child;
  for (int b = 0; b < nb; b++) {
    child_adr[b] = (int)child.size();
    for (int ch = nb - 1; ch > b; ch--) if (parent[ch] == b && b != 0 && !is_stub[ch]) child.push_back(ch);
  }
  child_adr[nb] = (int)child.size();
  std::vector<int> chain_len(nb, 0), chain(nb * MAXCHAIN, -1), bchain_len(nb, 0), bchain(nb * MAXBCHAIN, -1), body_agent(nb, -1);
  pic.assign((size_t)nb * nv, -1);
  const int* dofnum = SUMO_I(m, body_dofnum);
  const int* dofadr = SUMO_I(m, body_dofadr);
  const int* dpar = SUMO_I(m, dof_parentid);
  for (int b = 1; b < nb; b++) {
    std::vector<int> bc;
    for (int k = b; k != 0; k = parent[k]) bc.insert(bc.begin(), k);
    if ((int)bc.size() > MAXBCHAIN) FAIL(-12, "kinematic tree deeper than %d bodies", MAXBCHAIN);
    bchain_len[b] = (int)bc.size();
    for (size_t i = 0; i < bc.size(); i++) bchain[b * MAXBCHAIN + i] = bc[i];
    int bb = b;
    while (bb && dofnum[bb] == 0) bb = tparent[bb];
    std::vector<int> dc;
    if (bb) for (int d = dofadr[bb] + dofnum[bb] - 1; d >= 0; d = dpar[d]) dc.insert(dc.begin(), d);
    if ((int)dc.size() > MAXCHAIN) FAIL(-13, "dof chain longer than %d", MAXCHAIN);
    chain_len[b] = (int)dc.size();
    for (size_t i = 0; i < dc.size(); i++) { chain[b * MAXCHAIN + i] = dc[i]; pic[(size_t)b * nv + dc[i]] = (signed char)i; }
    for (int a = 0; a < m->nagent; a++) {
      int adr = SUMO_I(m, agent_bodyadr)[a];
      if (b >= adr && b < adr + SUMO_I(m, agent_nbody)[a]) body_agent[b] = a;
    }
    if (body_agent[b] < 0) FAIL(-14, "body %d belongs to no agent", b);
    // one geom per moving body, and its frame is the inertial frame (what the kernel assumes)
    int g0 = SUMO_I(m, body_geomadr)[b], g1 = SUMO_I(m, body_geomadr)[b + 1];
    if (g1 - g0 != 1) FAIL(-15, "body %d has %d geoms; the engine needs exactly one per moving body", b, g1 - g0);
    for (int k = 0; k < 3; k++) if (SUMO_F(m, geom_pos)[3 * g0 + k] != SUMO_F(m, body_ipos)[3 * b + k]) FAIL(-16, "geom/inertial frame mismatch");
    for (int k = 0; k < 4; k++) if (SUMO_F(m, geom_quat)[4 * g0 + k] != SUMO_F(m, body_iquat)[4 * b + k]) FAIL(-16, "geom/inertial frame mismatch");
  }
  // one actuator per dof at most (kernel adds actuator forces without atomics)
  {
    std::vector<int> seen(nv, 0);
    for (int u = 0; u < m->nu; u++) { int d = SUMO_I(m, actuator_dofid)[u]; if (seen[d]++) FAIL(-17, "two actuators on one dof"); }
  }
  std::vector<int> tri_i, tri_j;
  for (int i = 0; i < nv; i++) for (int j = 0; j <= i; j++) { tri_i.push_back(i); tri_j.push_back(j); }
  A.ntri = (int)tri_i.size();
  A.o_lvl_adr = push_tbl(lvl_adr); A.o_lvl_body = push_tbl(lvl_body); A.o_child_adr = push_tbl(child_adr);
  A.o_child = push_tbl(child.empty() ? std::vector<int>(1, 0) : child);
  A.o_chain_len = push_tbl(chain_len); A.o_chain = push_tbl(chain); A.o_bchain_len = push_tbl(bchain_len);
  A.o_bchain = push_tbl(bchain); A.o_body_agent = push_tbl(body_agent); A.o_tri_i = push_tbl(tri_i); A.o_tri_j = push_tbl(tri_j);
  A.o_wgmat = 0;
  af.assign((size_t)9 * m->ngeom, 0.0);
  for (int g = 0; g < m->ngeom; g++) quat2mat_h(&af[9 * g], SUMO_F(m, geom_quat) + 4 * g);

  // ---- collision centres: body ids for agent geoms, nbody + w for world geom w
  const int* gbody = SUMO_I(m, geom_bodyid);
  const int* gtype = SUMO_I(m, geom_type);
  int nworld = SUMO_I(m, body_geomadr)[1] - SUMO_I(m, body_geomadr)[0];
  int nc = nb + nworld;
  if (nc > 255) FAIL(-20, "too many collision centres");
  A.nc = nc; A.nworld = nworld;
  std::vector<int> cen_of_geom(m->ngeom, 0), geom_of_cen(nc, -1);
  for (int g = 0; g < m->ngeom; g++) {
    int ci = gbody[g] == 0 ? nb + (g - SUMO_I(m, body_geomadr)[0]) : gbody[g];
    cen_of_geom[g] = ci; geom_of_cen[ci] = g;
  }
  // static double tables: csize[2*nc] | cinvw[nc] | wbox[12*nworld] | wlim[6*nworld]
  A.o_stat_d = (int)af.size();
  {
    std::vector<double> sd((size_t)3 * nc + 18 * nworld, 0.0), wpa((size_t)6 * nworld, 0.0);
    for (int ci = 0; ci < nc; ci++) {
      int g = geom_of_cen[ci];
      if (g >= 0) { sd[2 * ci] = SUMO_F(m, geom_size)[3 * g]; sd[2 * ci + 1] = SUMO_F(m, geom_size)[3 * g + 1]; }
      int body = ci < nb ? ci : 0;
      sd[2 * nc + ci] = SUMO_F(m, body_invweight0)[2 * body];
    }
    const double BIG = 1e300;
    for (int w = 0; w < nworld; w++) {
      int g = SUMO_I(m, body_geomadr)[0] + w;
      double R[9];
      quat2mat_h(R, SUMO_F(m, geom_quat) + 4 * g);
      double* wb = &sd[3 * nc + 12 * w];
      for (int k = 0; k < 9; k++) wb[k] = R[k];
      for (int k = 0; k < 3; k++) wb[9 + k] = SUMO_F(m, geom_size)[3 * g + k];
      // broad-phase extent of the geom in its own frame, hi[3] | lo[3] (the other geom's bounding radius goes into the
      // pair bound): box = its half sizes; capsule / border rod = its axis segment; plane = the half space z <= 0;
      // anything else = its centre with the bounding radius
      double* wl = &sd[3 * nc + 12 * nworld + 6 * w];
      const double* gs = SUMO_F(m, geom_size) + 3 * g;
      if (gtype[g] == SUMO_GEOM_BOX) { for (int k = 0; k < 3; k++) { wl[k] = gs[k]; wl[3 + k] = -gs[k]; } }
      else if (gtype[g] == SUMO_GEOM_CAPSULE || gtype[g] == SUMO_GEOM_CYLINDER) { wl[2] = gs[1]; wl[5] = -gs[1]; }
      else if (gtype[g] == SUMO_GEOM_PLANE) { wl[0] = wl[1] = BIG; wl[2] = 0; wl[3] = wl[4] = wl[5] = -BIG; }
      for (int k = 0; k < 3; k++) wpa[3 * w + k] = SUMO_F(m, geom_pos)[3 * g + k];
      wpa[3 * nworld + 3 * w + 0] = R[2]; wpa[3 * nworld + 3 * w + 1] = R[5]; wpa[3 * nworld + 3 * w + 2] = R[8];
    }
    A.n_stat_d = (int)sd.size();
    af.insert(af.end(), sd.begin(), sd.end());
    A.o_wpa = (int)af.size();   // world geom positions / axes: read once per launch, not kept in LDS
    af.insert(af.end(), wpa.begin(), wpa.end());
  }
  // static int tables: ctype[nc] | cbody[nc] | chain words [2*nb] | chlen_agent[nb]
  {
    std::vector<int> si((size_t)2 * nc + 3 * nb, 0);
    for (int ci = 0; ci < nc; ci++) {
      int g = geom_of_cen[ci];
      int t = g >= 0 ? gtype[g] : -1;
      if (t == SUMO_GEOM_CYLINDER) t = SUMO_GEOM_CAPSULE;  // border rods collide as capsules
      si[ci] = t; si[nc + ci] = ci < nb ? ci : 0;
    }
    for (int b = 0; b < nb; b++) {
      unsigned w0 = 0, w1 = 0;
      for (int p = 0; p < chain_len[b]; p++) {
        unsigned d = (unsigned)chain[b * MAXCHAIN + p];
        if (p < 4) w0 |= d << (8 * p); else w1 |= d << (8 * (p - 4));
      }
      si[2 * nc + 2 * b] = (int)w0; si[2 * nc + 2 * b + 1] = (int)w1;
      si[2 * nc + 2 * nb + b] = chain_len[b] | ((body_agent[b] < 0 ? 0 : body_agent[b]) << 8);
    }
    A.o_stat_i = push_tbl(si);
    A.n_stat_i = (int)si.size();
  }
  // ---- per-lane constants
  E->lanes.assign(WAVE, LaneRec());
  const int* jbody = SUMO_I(m, jnt_bodyid);
  for (int l = 0; l < WAVE; l++) {
    LaneRec& K = E->lanes[l];
    memset(&K, 0, sizeof K);
    K.b_jnt = -1; K.b_level = -1; K.b_agent = 0;
    if (l < nb) {
      int b = l;
      for (int k = 0; k < 3; k++) { K.b_pos[k] = folded ? epos[3 * b + k] : SUMO_F(m, body_pos)[3 * b + k]; K.b_ipos[k] = SUMO_F(m, body_ipos)[3 * b + k]; K.b_inertia[k] = SUMO_F(m, body_inertia)[3 * b + k]; }
      for (int k = 0; k < 4; k++) { K.b_quat[k] = folded ? equat[4 * b + k] : SUMO_F(m, body_quat)[4 * b + k]; K.b_iquat[k] = SUMO_F(m, body_iquat)[4 * b + k]; }
      K.b_mass = SUMO_F(m, body_mass)[b];
      K.b_parent = parent[b]; K.b_level = depth[b]; K.b_agent = body_agent[b] < 0 ? 0 : body_agent[b];
      int jn = SUMO_I(m, body_jntnum)[b], ja = SUMO_I(m, body_jntadr)[b];
      if (jn > 1) FAIL(-21, "body %d has %d joints; the engine supports at most one per body", b, jn);
      if (jn == 1) {
        K.b_jnt = ja;
        K.b_isfree = SUMO_I(m, jnt_type)[ja] == SUMO_JNT_FREE;
        K.b_qadr = SUMO_I(m, jnt_qposadr)[ja];
        for (int k = 0; k < 3; k++) { K.j_pos[k] = SUMO_F(m, jnt_pos)[3 * ja + k]; K.j_axis[k] = SUMO_F(m, jnt_axis)[3 * ja + k]; }
        K.j_qpos0 = SUMO_F(m, qpos0)[K.b_qadr];
      }
      int c0 = child_adr[b], c1 = child_adr[b + 1];
      K.b_nchild = c1 - c0;
      if (K.b_nchild > 8) FAIL(-22, "body %d has more than 8 children", b);
      if (b > 0 && K.b_nchild > A.maxchild) A.maxchild = K.b_nchild;
      {  // inertial constants for cinert: own, merged with the rigidly attached leaves, or zero for such a leaf
        auto add_part = [&](double* I6, double* mc, double& M, double mass, const double* cpos, const double* Rb, const double* diag) {
          // accumulates mass, mass * centre and the inertia about the ORIGIN of the body frame (shifted to the centre below)
          double Ip[9];
          for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
            Ip[3 * r + cc] = Rb[3 * r] * diag[0] * Rb[3 * cc] + Rb[3 * r + 1] * diag[1] * Rb[3 * cc + 1] + Rb[3 * r + 2] * diag[2] * Rb[3 * cc + 2];
          double d2 = cpos[0] * cpos[0] + cpos[1] * cpos[1] + cpos[2] * cpos[2];
          I6[0] += Ip[0] + mass * (d2 - cpos[0] * cpos[0]); I6[1] += Ip[4] + mass * (d2 - cpos[1] * cpos[1]); I6[2] += Ip[8] + mass * (d2 - cpos[2] * cpos[2]);
          I6[3] += Ip[1] - mass * cpos[0] * cpos[1]; I6[4] += Ip[2] - mass * cpos[0] * cpos[2]; I6[5] += Ip[5] - mass * cpos[1] * cpos[2];
          for (int k = 0; k < 3; k++) mc[k] += mass * cpos[k];
          M += mass;
        };
        double I6[6] = {0, 0, 0, 0, 0, 0}, mc[3] = {0, 0, 0}, M = 0, Rb[9];
        if (!is_stub[b]) {
          quat2mat_h(Rb, SUMO_F(m, body_iquat) + 4 * b);
          add_part(I6, mc, M, SUMO_F(m, body_mass)[b], SUMO_F(m, body_ipos) + 3 * b, Rb, SUMO_F(m, body_inertia) + 3 * b);
          for (int sb = 1; sb < nb; sb++)
            if (is_stub[sb] && parent[sb] == b) {
              double Rs[9], Ri[9], Rsi[9], cs[3];
              quat2mat_h(Rs, &equat[4 * sb]);
              quat2mat_h(Ri, SUMO_F(m, body_iquat) + 4 * sb);
              for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
                Rsi[3 * r + cc] = Rs[3 * r] * Ri[cc] + Rs[3 * r + 1] * Ri[3 + cc] + Rs[3 * r + 2] * Ri[6 + cc];
              const double* ip = SUMO_F(m, body_ipos) + 3 * sb;
              for (int k = 0; k < 3; k++) cs[k] = epos[3 * sb + k] + Rs[3 * k] * ip[0] + Rs[3 * k + 1] * ip[1] + Rs[3 * k + 2] * ip[2];
              add_part(I6, mc, M, SUMO_F(m, body_mass)[sb], cs, Rsi, SUMO_F(m, body_inertia) + 3 * sb);
              A.nstub++;
            }
          double cm[3] = {0, 0, 0};
          if (M > 0) for (int k = 0; k < 3; k++) cm[k] = mc[k] / M;
          // shift from the frame origin to the (merged) centre of mass
          double d2 = cm[0] * cm[0] + cm[1] * cm[1] + cm[2] * cm[2];
          I6[0] -= M * (d2 - cm[0] * cm[0]); I6[1] -= M * (d2 - cm[1] * cm[1]); I6[2] -= M * (d2 - cm[2] * cm[2]);
          I6[3] += M * cm[0] * cm[1]; I6[4] += M * cm[0] * cm[2]; I6[5] += M * cm[1] * cm[2];
          for (int k = 0; k < 6; k++) K.c_I[k] = I6[k];
          for (int k = 0; k < 3; k++) K.c_ipos[k] = cm[k];
          K.c_mass = M;
        }
      }
      for (int q = 0; q < K.b_nchild; q++) K.b_child |= (unsigned long long)child[c0 + q] << (8 * q);
      K.b_bchain_len = bchain_len[b];
      for (int q = 0; q < bchain_len[b]; q++) K.b_bchain |= (unsigned)bchain[b * MAXBCHAIN + q] << (8 * q);
      K.b_chain_len = chain_len[b];
      for (int q = 0; q < chain_len[b]; q++) {
        int d = chain[b * MAXCHAIN + q];
        K.b_chain |= (unsigned long long)d << (8 * q);
        if (SUMO_I(m, dof_bodyid)[d] == b) K.b_chain_own |= 1u << q;
        int jt = SUMO_I(m, dof_jntid)[d];
        if (SUMO_I(m, jnt_type)[jt] == SUMO_JNT_FREE && SUMO_I(m, jnt_dofadr)[jt] == d) K.b_chain_free |= 1u << q;
      }
    }
    if (l < nv) {
      int d = l, b = SUMO_I(m, dof_bodyid)[d];
      K.d_arm = SUMO_F(m, dof_armature)[d]; K.d_damp = SUMO_F(m, dof_damping)[d]; K.d_body = b;
      K.d_pos = pic[(size_t)b * nv + d];
      K.d_hid = -1;
      { int jt = SUMO_I(m, dof_jntid)[d];
        if (SUMO_I(m, jnt_type)[jt] == SUMO_JNT_HINGE) { int h = 0; for (int jj = 0; jj < jt; jj++) if (SUMO_I(m, jnt_type)[jj] == SUMO_JNT_HINGE) h++; K.d_hid = h; } }
      for (int q = 0; q < chain_len[b]; q++) K.d_chain |= (unsigned long long)chain[b * MAXCHAIN + q] << (8 * q);
    }
    if (l < m->njnt) {
      int j = l;
      K.jt_type = SUMO_I(m, jnt_type)[j]; K.jt_qadr = SUMO_I(m, jnt_qposadr)[j]; K.jt_dadr = SUMO_I(m, jnt_dofadr)[j];
      K.jt_limited = SUMO_I(m, jnt_limited)[j]; K.jt_body = jbody[j]; K.jt_agent = body_agent[jbody[j]] < 0 ? 0 : body_agent[jbody[j]];
      K.jt_lo = SUMO_F(m, jnt_range)[2 * j]; K.jt_hi = SUMO_F(m, jnt_range)[2 * j + 1]; K.jt_margin = SUMO_F(m, jnt_margin)[j];
      K.jt_invw = SUMO_F(m, dof_invweight0)[K.jt_dadr];
      K.jt_hid = -1;
      if (K.jt_type == SUMO_JNT_HINGE) { int h = 0; for (int jj = 0; jj < j; jj++) if (SUMO_I(m, jnt_type)[jj] == SUMO_JNT_HINGE) h++; K.jt_hid = h; }
    }
    if (l < m->nu) {
      K.a_dof = SUMO_I(m, actuator_dofid)[l]; K.a_gear = SUMO_F(m, actuator_gear)[l];
      K.a_lo = SUMO_F(m, actuator_ctrlrange)[2 * l]; K.a_hi = SUMO_F(m, actuator_ctrlrange)[2 * l + 1];
    }
  }
  // ---- lower-triangle entry words of the dense Hessian assembly, entry t = lane + 64 m: (i << 8 | j) | chain position of dof i
  // << 16 | of dof j << 20; 0xFFFF = none.  Read where the (rare) dense path assembles H instead of living in 7 registers of
  // every lane for the whole launch.
  {
    const int epl = (A.ntri + WAVE - 1) / WAVE;
    std::vector<int> ent((size_t)epl * WAVE, 0xFFFF);
    for (int t = 0; t < A.ntri; t++) {
      const int i = tri_i[t], j = tri_j[t];
      ent[t] = (int)((unsigned)((i << 8) | j) | ((unsigned)E->lanes[i].d_pos << 16) | ((unsigned)E->lanes[j].d_pos << 20));
    }
    A.o_ent = push_tbl(ent);
  }
  // ---- packed pair records for the broad phase: pairs with a static world geom first (the model lists them first:
  // body-pair order, world body = 0), padded to whole rounds of 64, then the pairs between moving geoms.
  //   world pair:  moving centre | world centre << 8 | (world geom is geom1) << 16 | valid << 17
  //                bound = margin + rbound(moving geom) [+ radius of a static capsule; + rbound of other static shapes]
  //   moving pair: c1 | c2 << 8 | valid << 17;  bound = margin + rbound1 + rbound2
  // bounds are rounded up in float: the broad phase may only over-include
  {
    int nwp = 0;
    while (nwp < m->npair && (gbody[SUMO_I(m, pair_geom1)[nwp]] == 0 || gbody[SUMO_I(m, pair_geom2)[nwp]] == 0)) nwp++;
    for (int p = nwp; p < m->npair; p++)
      if (gbody[SUMO_I(m, pair_geom1)[p]] == 0 || gbody[SUMO_I(m, pair_geom2)[p]] == 0)
        FAIL(-23, "collision pair %d: pairs with a world geom must precede the others", p);
    const int WR = (nwp + WAVE - 1) / WAVE, AR = (m->npair - nwp + WAVE - 1) / WAVE;
    A.nwp = nwp; A.wrounds = WR; A.arounds = AR;
    E->pair_rec.assign((size_t)(WR + AR) * WAVE, 0);
    E->pair_bound.assign((size_t)(WR + AR) * WAVE, 0.0f);
    auto up = [](double bound) {
      float bf = (float)bound;
      while ((double)bf < bound) bf = nextafterf(bf, INFINITY);
      return nextafterf(bf, INFINITY);
    };
    for (int p = 0; p < m->npair; p++) {
      int g1 = SUMO_I(m, pair_geom1)[p], g2 = SUMO_I(m, pair_geom2)[p];
      const double margin = SUMO_F(m, pair_margin)[p];
      if (p < nwp) {
        if (gbody[g1] == 0 && gbody[g2] == 0) continue;   // static-static: never collides (slot stays invalid)
        const int w1 = gbody[g1] == 0, gw = w1 ? g1 : g2, ga = w1 ? g2 : g1, tw = gtype[gw];
        double bound = margin + SUMO_F(m, geom_rbound)[ga];
        if (tw == SUMO_GEOM_CAPSULE || tw == SUMO_GEOM_CYLINDER) bound += SUMO_F(m, geom_size)[3 * gw];
        else if (tw != SUMO_GEOM_BOX && tw != SUMO_GEOM_PLANE) bound += SUMO_F(m, geom_rbound)[gw];
        E->pair_rec[p] = cen_of_geom[ga] | (cen_of_geom[gw] << 8) | (w1 << 16) | (1 << 17);
        E->pair_bound[p] = up(bound);
      } else {
        const int sl = WR * WAVE + (p - nwp);
        E->pair_rec[sl] = cen_of_geom[g1] | (cen_of_geom[g2] << 8) | (1 << 17);
        E->pair_bound[sl] = up(margin + SUMO_F(m, geom_rbound)[g1] + SUMO_F(m, geom_rbound)[g2]);
      }
    }
  }
  return 0;
}

static void build_layout_with(sumo_engine* E, int jb_extra) {
  const sumo_model_t* m = &E->hm;
  Layout& L = E->L;
  int nq = m->nq, nv = m->nv, nb = m->nbody, nj = m->njnt, nu = m->nu;
  L.ld = nv | 1;
  int nhinge = 0;
  for (int j = 0; j < nj; j++) if (SUMO_I(m, jnt_type)[j] == SUMO_JNT_HINGE) nhinge++;
  // Ant-vs-Ant: 18 contact records (the soaks and the zoo play never saw more than 14) instead of 24: the 1 KB goes to the contact-Jacobian pool
  // (build_layout: 29 halves instead of 24 -- 14 contacts between two moving bodies fit), which is what overflowed a few times per 10^8 forwards
  L.maxcon = nv <= 28 ? 18 : (nv <= 36 ? 32 : 40);
  const char* mc = getenv("SUMO_MAXCON");
  if (mc && atoi(mc) > 0) L.maxcon = atoi(mc);
  { int cap = 16 * (nv <= 36 ? 2 : 3); if (L.maxcon > cap) L.maxcon = cap; }  // contact rows per lane held in registers by the line search (newton_solve RPL)
  L.maxefc = 4 * L.maxcon + 2 * nhinge;
  { const char* wm = getenv("SUMO_WARM_MODE"); L.warm_mode = wm ? atoi(wm) : 0; }
  { const char* fd = getenv("SUMO_FORCE_DENSE_H"); L.force_dense = fd ? atoi(fd) : 0; }
  int o = 0;
  auto take = [&](int n) { int r = o; o += n; return r; };
  L.qpos = take(nq); L.qvel = take(nv); L.warm = take(nv); L.ctrl = take(nu);
  L.x0 = 0; L.accv = 0; L.acca = 0; L.tmpv = take(nv);
  // kinematic data needed until the mass matrix is built (com, cinert -> crb, cdof); the packed Hessian aliases it
  const int nc = E->aux.nc;
  const int ntri = nv * (nv + 1) / 2;
  int kin0 = o;
  L.com = take(3 * m->nagent); L.cinert = take(10 * nb); L.cdof = take(6 * nv);
  L.H = kin0;
  if (o - kin0 < ntri) o = kin0 + ntri;
  const int hsize = o - kin0;
  // centre positions / axes: bodies (rewritten every forward) then world geoms (static)
  L.xipos = take(3 * nc); L.gaxis = take(3 * nc);
  L.stat_d = take(SUMO_STAT_LDS ? E->aux.n_stat_d : 0);
  {
    int nv0 = SUMO_I(m, agent_nv)[0], nv1 = SUMO_I(m, agent_nv)[1];
    int mx = nv0 > nv1 ? nv0 : nv1;
    L.mld = mx | 1;
    L.d1 = SUMO_I(m, agent_dofadr)[1];
    L.msize = (nv0 + nv1) * L.mld;
  }
  {  // shape tree_factor_solve relies on: two agents, each a free joint (6 dofs) + legs of (hip, ankle), bases even
    const int* dpar = SUMO_I(m, dof_parentid);
    int ok = m->nagent == 2 && SUMO_I(m, agent_dofadr)[0] == 0 && SUMO_I(m, agent_nv)[0] + SUMO_I(m, agent_nv)[1] == nv;
    for (int g = 0; ok && g < 2; g++) {
      int B = SUMO_I(m, agent_dofadr)[g], n = SUMO_I(m, agent_nv)[g];
      if ((B & 1) || n < 6 || ((n - 6) & 1)) { ok = 0; break; }
      if (dpar[B] != -1) ok = 0;
      for (int k = 1; ok && k < 6; k++) if (dpar[B + k] != B + k - 1) ok = 0;
      for (int d = B + 6; ok && d < B + n; d += 2) if (dpar[d] != B + 5 || dpar[d + 1] != d) ok = 0;
    }
    if (14 * nv > hsize) ok = 0;   // exchange buffers of the tree solver live in the Hessian region
    L.tree_ok = ok;
    const char* nt = getenv("SUMO_NO_TREE");
    if (nt && atoi(nt)) L.tree_ok = 0;
  }
  L.qsm = take(nv); L.asmo = take(nv); L.Ma = 0; L.grad = 0;
  L.search = take(nv); L.bias = L.search;   // the bias force is consumed (into qsm) before the solver writes its search direction
  L.Mv = 0; L.x = take(nv); L.dlim = take(nv); L.cmask = take(nv); L.stash = take(12);   // + the fused rollout's ticket (env, step) and the step's reward inputs / episode record for its post phase
  // contact records live from the narrow phase to the Jacobian build only: they borrow the mass matrix's storage
  if (L.msize < 14 * L.maxcon) L.msize = 14 * L.maxcon;
  if (L.msize < 12 * nb) L.msize = 12 * nb;      // ... and so do the velocity-pass temporaries (abuf, cfrc), see below
  L.M = take(L.msize);
  L.cond = L.M;
  // body frames / joint anchors / RNE temporaries are dead before the contact Jacobians are written: same storage
  {
    // pool of Jacobian halves (3 rows x 8 slots, one half per moving body of a contact): room for maxcon contacts with the
    // world, or half as many between two moving bodies -- more where the frames' storage leaves space anyway
    int need = 7 * nb + 6 * nj;
    L.jbcap = L.maxcon;
    if ((need - 16) / 24 > L.jbcap) L.jbcap = (need - 16) / 24;
    L.jbcap += jb_extra;
    { const char* jc = getenv("SUMO_JBCAP"); if (jc && atoi(jc) > 0) L.jbcap = atoi(jc); }
    int jb0 = o, jbsz = 24 * L.jbcap + 16;
    L.Jb = take(jbsz > need ? jbsz : need);
    int q = jb0;
    L.xpos = q; q += 3 * nb; L.xquat = q; q += 4 * nb; L.xanchor = q; q += 3 * nj; L.xaxis = q; q += 3 * nj;
    // velocity-pass temporaries (dead before the narrow phase writes contact records there): the mass matrix's storage
    L.abuf = L.M; L.cfrc = L.M + 6 * nb;
  }
  L.cpar = take(L.maxcon); L.cW = take(5 * L.maxcon);
  L.cp = take(3 * L.maxcon);
  // row arrays hold the contact rows only (4 per contact); the limit rows live in the dof-indexed slots limD / limA
  L.jar = take(4 * L.maxcon); L.D = take(L.maxcon); L.aref = take(4 * L.maxcon);
  L.nhinge = nhinge; L.limD = take(2 * nhinge); L.limA = take(2 * nhinge);
  // queue of broad-phase survivors (ints, inside jar / aref); drained by the narrow phase before it can overflow
  L.maxcand = 8 * L.maxcon >= 192 ? 192 : (8 * L.maxcon) / WAVE * WAVE;
  L.i_base = o;
  int io = 0;
  auto itake = [&](int n) { int r = io; io += n; return r; };
  L.con_b = itake(4 * L.maxcon);
  L.stat_i = itake(SUMO_STAT_LDS ? E->aux.n_stat_i : 0);
  L.b_dofidx = io * 4;
  L.total_bytes = o * 8 + io * 4 + 16 * L.maxcon;
}
// The LDS a wave slot leaves unused at the scene's residency (160 KB / waves per CU - the env record) goes to the contact-Jacobian
// pool: a contact between two moving bodies takes two halves, and wrestling Spiders overflowed a pool of `maxcon` halves about twice
// per million env steps (counted in `dropped`).  Ant-vs-Ant has 32 B to spare (8 x 20 448 B = 163 584 of 163 840): unchanged.
static void build_layout(sumo_engine* E) {
  build_layout_with(E, 0);
  const int lds_cu = 160 * 1024, slots = lds_cu / E->L.total_bytes;
  if (slots < 1 || getenv("SUMO_JBCAP")) return;
  const int budget = lds_cu / slots / 1280 * 1280;   // LDS is allocated in granules of 1280 B on gfx950
  int extra = (budget - E->L.total_bytes) / (24 * 8);
  if (extra > E->L.maxcon) extra = E->L.maxcon;      // (a pool of 2 x maxcon halves can never overflow)
  if (extra <= 0) return;
  build_layout_with(E, extra);
  if (lds_cu / E->L.total_bytes != slots) build_layout_with(E, 0);   // (cannot happen: the growth was sized to fit)
}

static int model_ints(const sumo_model_t* m, int32_t* out);
extern "C" int sumo_create(const void* model_blob, size_t nbytes, int num_envs, int device, sumo_handle_t* out) {
  if (!model_blob || !out || num_envs <= 0) FAIL(-1, "bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) FAIL(-2, "no HIP device available: the engine has no CPU fallback");
  if (device < 0 || device >= ndev) FAIL(-3, "device %d out of range (%d devices)", device, ndev);
  HIPCHK(hipSetDevice(device));
  sumo_engine* E = new sumo_engine();
  E->device = device; E->N = num_envs;
  E->blob.assign((const char*)model_blob, (const char*)model_blob + nbytes);
  int rc = sumo_model_parse(&E->hm, E->blob.data(), nbytes);
  if (rc != 0) { delete E; FAIL(-4, "malformed model blob (%d)", rc); }
  std::vector<int> ai; std::vector<double> af; std::vector<signed char> pic;
  rc = build_aux(E, ai, af, pic);
  if (rc != 0) { delete E; return rc; }
  build_layout(E);
  {   // static kernel variants only when the runtime Layout IS the compile-time table (SUMO_STATIC_LAYOUT=0 switches them off)
    static_assert(sizeof(Layout) % sizeof(int) == 0, "Layout is all ints");
    const char* sl = getenv("SUMO_STATIC_LAYOUT");
    int32_t mi[160];
    const int nmi = model_ints(&E->hm, mi);
    auto same = [&](const int* kl, size_t nl, const int* km, size_t nm_, const int* ka, size_t na) {
      return nl == sizeof(Layout) && memcmp(&E->L, kl, nl) == 0 && nm_ == nmi * sizeof(int) && memcmp(mi, km, nm_) == 0 &&
             na == AUX_NINTS * sizeof(int) && memcmp(&E->aux.ndepth, ka, na) == 0;
    };
    E->static_layout = 0;      // 1: Ant-vs-Ant, 2: Spider-vs-Spider (the scenes of csrc/layout_static.h), 0: runtime-Layout variants
    if (!(sl && atoi(sl) == 0)) {
      if (same(kLayoutAntAnt, sizeof(kLayoutAntAnt), kModelAntAnt, sizeof(kModelAntAnt), kAuxAntAnt, sizeof(kAuxAntAnt))) E->static_layout = 1;
      else if (same(kLayoutSpiderSpider, sizeof(kLayoutSpiderSpider), kModelSpiderSpider, sizeof(kModelSpiderSpider), kAuxSpiderSpider, sizeof(kAuxSpiderSpider)))
        E->static_layout = 2;
    }
  }
  if (E->L.maxefc > WAVE * (E->hm.nv <= 28 ? 2 : 4)) { int me = E->L.maxefc; delete E; FAIL(-24, "maxefc %d exceeds the rows a lane keeps in registers", me); }
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if ((size_t)E->L.total_bytes > prop.sharedMemPerBlock && (size_t)E->L.total_bytes > 160 * 1024) {
    delete E; FAIL(-5, "LDS need %d bytes exceeds the device limit", E->L.total_bytes);
  }
  const sumo_model_t* m = &E->hm;
  int od = 0, ad = 0;
  for (int a = 0; a < m->nagent; a++) {
    int o = SUMO_I(m, agent_nq)[a] + SUMO_I(m, agent_nv)[a] + 6 * SUMO_I(m, agent_nbody)[a] + 14;
    if (o > od) od = o;
    if (SUMO_I(m, agent_nu)[a] > ad) ad = SUMO_I(m, agent_nu)[a];
  }
  E->obs_stride = od; E->act_stride = ad;
  E->state_stride = (m->nq + 2 * m->nv + 2 + 15) & ~15;
  HIPCHK(hipMalloc(&E->d_blob, nbytes));
  HIPCHK(hipMemcpy(E->d_blob, E->blob.data(), nbytes, hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_ai, ai.size() * sizeof(int)));
  HIPCHK(hipMemcpy(E->d_ai, ai.data(), ai.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_af, af.size() * sizeof(double)));
  HIPCHK(hipMemcpy(E->d_af, af.data(), af.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_pic, pic.size()));
  HIPCHK(hipMemcpy(E->d_pic, pic.data(), pic.size(), hipMemcpyHostToDevice));
  E->aux.ai = E->d_ai; E->aux.af = E->d_af; E->aux.pic = E->d_pic;
  E->dm = E->hm;
  E->dm.ibase = (const int32_t*)((const char*)E->d_blob + ((const char*)E->hm.ibase - E->blob.data()));
  E->dm.fbase = (const double*)((const char*)E->d_blob + ((const char*)E->hm.fbase - E->blob.data()));
  HIPCHK(hipMalloc((void**)&E->d_lanes, E->lanes.size() * sizeof(LaneRec)));
  HIPCHK(hipMemcpy(E->d_lanes, E->lanes.data(), E->lanes.size() * sizeof(LaneRec), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_pair_rec, E->pair_rec.size() * sizeof(int)));
  HIPCHK(hipMemcpy(E->d_pair_rec, E->pair_rec.data(), E->pair_rec.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_pair_bound, E->pair_bound.size() * sizeof(float)));
  HIPCHK(hipMemcpy(E->d_pair_bound, E->pair_bound.data(), E->pair_bound.size() * sizeof(float), hipMemcpyHostToDevice));
  {
    Params hp;
    hp.mdl = E->dm; hp.aux = E->aux; hp.L = E->L;
    hp.lanes = E->d_lanes; hp.pair_rec = E->d_pair_rec; hp.pair_bound = E->d_pair_bound;
    hp.adjust_z = 0.0;
    HIPCHK(hipMalloc((void**)&E->d_params, sizeof(Params)));
    HIPCHK(hipMemcpy(E->d_params, &hp, sizeof(Params), hipMemcpyHostToDevice));
  }
  size_t N = (size_t)num_envs;
  HIPCHK(hipMalloc((void**)&E->d_state, N * E->state_stride * sizeof(double)));
  HIPCHK(hipMemset(E->d_state, 0, N * E->state_stride * sizeof(double)));
  HIPCHK(hipMalloc((void**)&E->d_counters, N * 4 * sizeof(int)));
  HIPCHK(hipMemset(E->d_counters, 0, N * 4 * sizeof(int)));
  HIPCHK(hipMalloc((void**)&E->d_seeds, N * sizeof(uint64_t)));
  std::vector<uint64_t> seeds(N);
  for (size_t i = 0; i < N; i++) seeds[i] = i;
  HIPCHK(hipMemcpy(E->d_seeds, seeds.data(), N * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&E->d_stats, 48 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(E->d_stats, 0, 48 * sizeof(unsigned long long)));
  {
    const char* sc = getenv("SUMO_SCHED");
    E->sched = !(sc && atoi(sc) == 0);
    HIPCHK(hipMalloc((void**)&E->d_cost, 2 * N * sizeof(int)));
    HIPCHK(hipMalloc((void**)&E->d_cost_sorted, N * sizeof(int)));
    HIPCHK(hipMalloc((void**)&E->d_iota, N * sizeof(int)));
    HIPCHK(hipMalloc((void**)&E->d_perm, 2 * N * sizeof(int)));
    std::vector<int> iota(N);
    for (size_t i = 0; i < N; i++) iota[i] = (int)i;
    HIPCHK(hipMemcpy(E->d_iota, iota.data(), N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(rocprim::radix_sort_pairs_desc(nullptr, E->sort_tmp_bytes, E->d_cost, E->d_cost_sorted, E->d_iota, E->d_perm, N, 0, 16,
                                          (hipStream_t)0));
    HIPCHK(hipMalloc(&E->d_sort_tmp, E->sort_tmp_bytes ? E->sort_tmp_bytes : 16));
  }
  // qpos = qpos0 for every env until the first reset / set_state
  {
    std::vector<double> st(N * E->state_stride, 0.0);
    for (size_t e = 0; e < N; e++) memcpy(&st[e * E->state_stride], SUMO_F(m, qpos0), m->nq * sizeof(double));
    HIPCHK(hipMemcpy(E->d_state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  if (E->L.total_bytes > 64 * 1024) {
#define X(NVV)                                                                                                         \
    if (E->hm.nv == NVV) {                                                                                            \
      HIPCHK(hipFuncSetAttribute((const void*)sumo_step_kernel<NVV>, hipFuncAttributeMaxDynamicSharedMemorySize, E->L.total_bytes));    \
      HIPCHK(hipFuncSetAttribute((const void*)sumo_reset_kernel<NVV>, hipFuncAttributeMaxDynamicSharedMemorySize, E->L.total_bytes));   \
      HIPCHK(hipFuncSetAttribute((const void*)sumo_forward_kernel<NVV>, hipFuncAttributeMaxDynamicSharedMemorySize, E->L.total_bytes)); \
    }
    SUMO_FOR_NV(X)
#undef X
  }
  *out = E;
  return 0;
}

extern "C" int sumo_destroy(sumo_handle_t E) {
  if (!E) return 0;
  (void)hipSetDevice(E->device);
  (void)hipFree(E->d_params); (void)hipFree(E->d_lanes); (void)hipFree(E->d_pair_rec); (void)hipFree(E->d_pair_bound); (void)hipFree(E->d_blob); (void)hipFree(E->d_ai); (void)hipFree(E->d_af); (void)hipFree(E->d_pic); (void)hipFree(E->d_state); (void)hipFree(E->d_state_prev); (void)hipFree(E->d_fbuf); (void)hipFree(E->d_cfrc);
  (void)hipFree(E->d_counters); (void)hipFree(E->d_seeds); (void)hipFree(E->d_stats); (void)hipFree(E->d_rsched);
  (void)hipFree(E->d_cost); (void)hipFree(E->d_cost_sorted); (void)hipFree(E->d_iota); (void)hipFree(E->d_perm); (void)hipFree(E->d_sort_tmp);
  delete E;
  return 0;
}

extern "C" int sumo_dims(sumo_handle_t E, int32_t* o) {
  if (!E || !o) FAIL(-1, "bad arguments");
  const sumo_model_t* m = &E->hm;
  int v[SUMO_NDIMS] = {m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->npair, m->nagent, E->obs_stride, E->act_stride,
                       E->L.maxcon, E->L.maxefc, E->L.total_bytes, E->state_stride, E->L.jbcap, 0};
  memcpy(o, v, sizeof v);
  return 0;
}

// kernel variants by nv (the factorisation is unrolled over a compile-time nv); the nine registered scenes have
// nv in {28, 32, 36, 40, 44}
template <class F>
static bool for_kernel_variant(int nv, F&& f) {
#define X(NVV) if (nv == NVV) { f(std::integral_constant<int, NVV>()); return true; }
  SUMO_FOR_NV(X)
#undef X
  return false;
}
#define SUMO_DISPATCH(KERNEL, E, stream, args) SUMO_DISPATCH_N(KERNEL, E, stream, args, (E)->N)
#define SUMO_DISPATCH_N(KERNEL, E, stream, args, nblocks)                                                   \
  do {                                                                                                       \
    dim3 g_(nblocks), b_(WAVE);                                                                              \
    size_t lds_ = (size_t)(E)->L.total_bytes;                                                                \
    if (!for_kernel_variant((E)->hm.nv, [&](auto nvc_) {                                                     \
          hipLaunchKernelGGL(KERNEL<decltype(nvc_)::value>, g_, b_, lds_, stream, (E)->d_params, args);      \
        }))                                                                                                  \
      FAIL(-19, "no kernel variant for nv=%d", (E)->hm.nv);                                                  \
  } while (0)

static StepArgs base_args(sumo_engine* E) {
  StepArgs a;
  memset(&a, 0, sizeof a);
  a.state = E->d_state; a.counters = E->d_counters; a.seeds = E->d_seeds; a.state_stride = E->state_stride;
  a.stats = E->d_stats; a.obs_stride = E->obs_stride; a.act_stride = E->act_stride; a.N = E->N;
  a.dbg_fault_env = -1;
  a.dbg_qacc = E->dbg_dump;
  return a;
}

extern "C" int sumo_reset(sumo_handle_t E, const uint64_t* seeds_host, const uint8_t* mask_dev, float* obs_dev, void* stream) {
  if (!E) FAIL(-1, "bad handle");
  HIPCHK(hipSetDevice(E->device));
  hipStream_t s = (hipStream_t)stream;
  if (seeds_host) {
    HIPCHK(hipMemcpyAsync(E->d_seeds, seeds_host, (size_t)E->N * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(E->d_counters, 0, (size_t)E->N * 4 * sizeof(int), s));
  }
  StepArgs a = base_args(E);
  a.mask = mask_dev; a.obs = obs_dev;
  SUMO_DISPATCH(sumo_reset_kernel, E, s, a);
  HIPCHK(hipGetLastError());
  if (seeds_host) HIPCHK(hipStreamSynchronize(s));  // seeds_host may be freed by the caller after return
  return 0;
}

extern "C" int sumo_step(sumo_handle_t E, const float* actions_dev, float* obs_dev, double* info_dev, uint8_t* done_dev,
                         double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream) {
  if (!E || !actions_dev || !obs_dev || !info_dev || !done_dev || !ep_r_dev || !ep_dr_dev || !ep_l_dev) FAIL(-1, "bad arguments");
  HIPCHK(hipSetDevice(E->device));
  StepArgs a = base_args(E);
  a.actions = actions_dev; a.obs = obs_dev; a.info = info_dev; a.done = done_dev; a.ep_r = ep_r_dev; a.ep_dr = ep_dr_dev;
  a.ep_l = ep_l_dev;
  // Env steps differ in cost (Newton iterations, contacts) by up to ~1.6x and a launch is only N / (6 * 256) rounds deep, so
  // the slowest workgroups of the last round set the launch time.  Workgroups are dispatched in index order: hand the
  // envs out longest-first, using the work each env reported in its previous step (results do not depend on the order).
  a.trace = E->d_trace;
  int nblocks = E->N;
  if (E->sched && E->N <= SCHED_RANK_MAX) {
    // in-kernel ranking (sched_rank): launch t writes its estimates to cost[t & 1]; its first workgroups rank cost[(t-1) & 1]
    // into perm[(t+1) & 1]; it is itself scheduled by perm[t & 1], ranked during launch t-1 from the estimates of launch t-2
    const long long t = E->sched_t++;
    a.cost = E->d_cost + (size_t)(t & 1) * E->N;
    a.perm = t >= 2 ? E->d_perm + (size_t)(t & 1) * E->N : nullptr;
    if (t >= 1) {
      a.rank_cost = E->d_cost + (size_t)((t - 1) & 1) * E->N;
      a.rank_perm = E->d_perm + (size_t)((t + 1) & 1) * E->N;
      a.rank_blocks = (E->N + WAVE - 1) / WAVE;
      nblocks += a.rank_blocks;
    }
  } else if (E->sched) {
    a.cost = E->d_cost; a.perm = E->perm_valid ? E->d_perm : nullptr;
  }
  if (E->cfrc_mode)   // rne_post: the state this step starts from, for the second launch below
    HIPCHK(hipMemcpyAsync(E->d_state_prev, E->d_state, (size_t)E->N * E->state_stride * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (E->static_layout == 1) {
    dim3 g_(nblocks), b_(WAVE);
    hipLaunchKernelGGL((sumo_step_kernel<28, 1>), g_, b_, (size_t)E->L.total_bytes, (hipStream_t)stream, E->d_params, a);
  } else if (E->static_layout == 2) {
    dim3 g_(nblocks), b_(WAVE);
    hipLaunchKernelGGL((sumo_step_kernel<44, 2>), g_, b_, (size_t)E->L.total_bytes, (hipStream_t)stream, E->d_params, a);
  } else
    SUMO_DISPATCH_N(sumo_step_kernel, E, (hipStream_t)stream, a, nblocks);
  HIPCHK(hipGetLastError());
  if (E->cfrc_mode) {
    CfrcArgs q;
    q.prev_state = E->d_state_prev; q.fbuf = E->d_fbuf; q.cfrc_out = E->d_cfrc;
    StepArgs ac = base_args(E);
    ac.actions = actions_dev; ac.obs = obs_dev; ac.done = done_dev;
    dim3 g_(E->N), b_(WAVE);
    size_t lds_ = (size_t)E->L.total_bytes;
    hipStream_t st_ = (hipStream_t)stream;
    if (!for_kernel_variant(E->hm.nv, [&](auto nvc_) {
          hipLaunchKernelGGL(sumo_cfrc_kernel<decltype(nvc_)::value>, g_, b_, lds_, st_, E->d_params, ac, q);
        }))
      FAIL(-19, "no kernel variant for nv=%d", E->hm.nv);
    HIPCHK(hipGetLastError());
  }
  if (E->sched && E->N > SCHED_RANK_MAX) {   // costs stay below 2^16 (see the step kernel's epilogue)
    HIPCHK(rocprim::radix_sort_pairs_desc(E->d_sort_tmp, E->sort_tmp_bytes, E->d_cost, E->d_cost_sorted, E->d_iota, E->d_perm,
                                          (size_t)E->N, 0, 16, (hipStream_t)stream));
    E->perm_valid = true;
  }
  return 0;
}

// common tail of the two fused-rollout entry points: scheduler state, persistent grid, launch
static int rollout_launch(sumo_engine* E, const RolloutArgs& r, int policy, float* actions_dev, float* obs_dev, double* info_dev,
                          uint8_t* done_dev, double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream) {
  RolloutLaunch rl;
  rl.a = base_args(E);
  StepArgs& a = rl.a;
  a.actions = actions_dev; a.obs = obs_dev; a.info = info_dev; a.done = done_dev; a.ep_r = ep_r_dev; a.ep_dr = ep_dr_dev; a.ep_l = ep_l_dev;
  rl.r = r;
  rl.r.prof = E->d_trace;   // development: sumo_debug_trace(stamps) switches the per-wave phase clock on
  hipStream_t st_ = (hipStream_t)stream;
  if (!E->d_rsched) HIPCHK(hipMalloc((void**)&E->d_rsched, (size_t)(2 + E->N) * sizeof(int)));
  HIPCHK(hipMemsetAsync(E->d_rsched, 0, (size_t)(2 + E->N) * sizeof(int), st_));
  rl.r.sched = E->d_rsched;
  if (r.K >= 65536) FAIL(-5, "K = %d: at most 65535 steps per fused launch", r.K);
  E->rollout_seq++;
  a.seq_base = (int)((E->rollout_seq & 0x7FFFu) << 16);
  a.abort_flag = E->d_rsched + 1;
  a.dbg_fault_env = E->dbg_fault_env;
  E->rollout_stream = st_;
  E->rollout_tickets = (long long)E->N * r.K;
  // persistent waves: as many as the chip holds at this kernel's LDS footprint (8 per CU at most: two per SIMD)
  int slots = (int)((size_t)160 * 1024 / (size_t)E->L.total_bytes);
  if (slots > 4 * SUMO_WPE_OF(E->hm.nv)) slots = 4 * SUMO_WPE_OF(E->hm.nv);
  if (slots < 1) slots = 1;
  if (E->num_cus <= 0) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, E->device));
    E->num_cus = prop.multiProcessorCount;
  }
  long long nw = (long long)slots * E->num_cus;
  if (nw > (long long)E->N) nw = E->N;   // more waves than envs would only wait on each other's steps
  dim3 g_((unsigned)nw), b_(WAVE);
  size_t lds_ = (size_t)E->L.total_bytes;
  if (E->static_layout == 1) {
    if (policy == 1) hipLaunchKernelGGL((sumo_rollout_kernel<28, 1, 1>), g_, b_, lds_, st_, E->d_params, rl);
    else hipLaunchKernelGGL((sumo_rollout_kernel<28, 0, 1>), g_, b_, lds_, st_, E->d_params, rl);
  } else if (E->static_layout == 2) {
    if (policy == 1) hipLaunchKernelGGL((sumo_rollout_kernel<44, 1, 2>), g_, b_, lds_, st_, E->d_params, rl);
    else hipLaunchKernelGGL((sumo_rollout_kernel<44, 0, 2>), g_, b_, lds_, st_, E->d_params, rl);
  } else if (!for_kernel_variant(E->hm.nv, [&](auto nvc_) {
        if (policy == 1) hipLaunchKernelGGL((sumo_rollout_kernel<decltype(nvc_)::value, 1>), g_, b_, lds_, st_, E->d_params, rl);
        else hipLaunchKernelGGL((sumo_rollout_kernel<decltype(nvc_)::value, 0>), g_, b_, lds_, st_, E->d_params, rl);
      }))
    FAIL(-19, "no kernel variant for nv=%d", E->hm.nv);
  HIPCHK(hipGetLastError());
  return 0;
}

// scene checks shared by the two entry points; returns the observation / action width through od / ad
static int rollout_scene(sumo_engine* E, int T, int Ntot, int env_offset, int s0, int K, int* od, int* ad) {
  const sumo_model_t* m = &E->hm;
  if (E->cfrc_mode) FAIL(-12, "cfrc_mode rne_post fills the observations in a second launch per step: use sumo_step (the fused rollout evaluates the policies inside its launch)");
  const int* anq = SUMO_I(m, agent_nq); const int* anv = SUMO_I(m, agent_nv); const int* anb = SUMO_I(m, agent_nbody);
  const int* anu = SUMO_I(m, agent_nu);
  const int od0 = anq[0] + anv[0] + 6 * anb[0] + 14, od1 = anq[1] + anv[1] + 6 * anb[1] + 14;
  if (od0 != od1 || anu[0] != anu[1]) FAIL(-3, "fused rollout needs a homogeneous match-up (one observation / action space for both sides, runner.py:14-16)");
  if (T < 1 || K < 1 || s0 < 0 || s0 + K > T) FAIL(-5, "steps [%d, %d) outside the rollout buffers (T = %d)", s0, s0 + K, T);
  if (env_offset < 0 || env_offset + E->N > Ntot) FAIL(-6, "envs [%d, %d) outside the rollout buffers (N = %d)", env_offset, env_offset + E->N, Ntot);
  *od = od0; *ad = anu[0];
  return 0;
}

extern "C" int sumo_rollout_steps(sumo_handle_t E, const sumo_rollout* ro, float* actions_dev, float* obs_dev, double* info_dev,
                                  uint8_t* done_dev, double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream) {
  if (!E || !ro || !actions_dev || !obs_dev || !info_dev || !done_dev || !ep_r_dev || !ep_dr_dev || !ep_l_dev) FAIL(-1, "bad arguments");
  if (!ro->learner_params || !ro->opponent_params || !ro->noise0 || !ro->noise1 || !ro->obs || !ro->act || !ro->rew || !ro->val ||
      !ro->nlp || !ro->onlp || !ro->done || !ro->ep_done || !ro->ep_r || !ro->ep_l)
    FAIL(-2, "sumo_rollout: missing buffer");
  int od = 0, ad = 0;
  if (int rc = rollout_scene(E, ro->T, ro->Ntot, ro->env_offset, ro->s0, ro->K, &od, &ad)) return rc;
  if (ro->ob_dim != od || ro->ac_dim != ad || ro->ac_dim > PT_MAXA) FAIL(-4, "ob_dim %d / ac_dim %d do not match the scene (%d / %d)", ro->ob_dim, ro->ac_dim, od, ad);
  if (ro->npool < 1) FAIL(-7, "npool %d", ro->npool);
  HIPCHK(hipSetDevice(E->device));
  RolloutArgs r;
  memset(&r, 0, sizeof r);
  r.learner = ro->learner_params; r.opponent = ro->opponent_params; r.opp_idx = ro->opponent_index; r.noise0 = ro->noise0; r.noise1 = ro->noise1;
  r.obs = ro->obs; r.act = ro->act; r.rew = ro->rew; r.val = ro->val; r.nlp = ro->nlp; r.onlp = ro->onlp; r.done = ro->done;
  r.ep_done = ro->ep_done; r.ep_r = ro->ep_r; r.ep_l = ro->ep_l; r.alpha = ro->alpha;
  r.T = ro->T; r.Ntot = ro->Ntot; r.env_offset = ro->env_offset; r.s0 = ro->s0; r.K = ro->K;
  r.XS = x_stride(ro->ob_dim); r.L = make_layout(ro->ob_dim, ro->ac_dim); r.lds_off = E->L.M;
  if ((size_t)(2 * r.XS + 4 * PT_HS) * sizeof(float) > (size_t)E->L.msize * sizeof(double))
    FAIL(-8, "policy scratch (%zu B) does not fit the mass-matrix region (%zu B)", (size_t)(2 * r.XS + 4 * PT_HS) * sizeof(float), (size_t)E->L.msize * sizeof(double));
  return rollout_launch(E, r, 0, actions_dev, obs_dev, info_dev, done_dev, ep_r_dev, ep_dr_dev, ep_l_dev, stream);
}

extern "C" int sumo_rollout_steps_lstm(sumo_handle_t E, const sumo_rollout_lstm* ro, float* actions_dev, float* obs_dev, double* info_dev,
                                       uint8_t* done_dev, double* ep_r_dev, double* ep_dr_dev, int32_t* ep_l_dev, void* stream) {
  if (!E || !ro || !actions_dev || !obs_dev || !info_dev || !done_dev || !ep_r_dev || !ep_dr_dev || !ep_l_dev) FAIL(-1, "bad arguments");
  if (!ro->learner || !ro->opponents_dev || !ro->state0 || !ro->state1 || !ro->noise0 || !ro->noise1 || !ro->obs || !ro->act || !ro->rew ||
      !ro->val || !ro->nlp || !ro->onlp || !ro->done || !ro->ep_done || !ro->ep_r || !ro->ep_l)
    FAIL(-2, "sumo_rollout_lstm: missing buffer");
  int od = 0, ad = 0;
  if (int rc = rollout_scene(E, ro->T, ro->Ntot, ro->env_offset, ro->s0, ro->K, &od, &ad)) return rc;
  const ppo_lstm_net& n = *ro->learner;
  if (n.ob_dim != od || n.ac_dim != ad || n.ac_dim > PT_MAXA) FAIL(-4, "ob_dim %d / ac_dim %d do not match the scene (%d / %d)", n.ob_dim, n.ac_dim, od, ad);
  // what the in-wave evaluation is built for: the nets `learn(network='lstm')` trains (policy-zoo LSTM nets carry an observation
  // filter, an embedding and the other gate order: they go through ppo_lstm_step)
  if (n.hidden != 128 || n.gate_order != PPO_LSTM_GATES_IFOU || n.emb_w || n.emb_dim != 0 || n.obs_mean || n.obs_invstd)
    FAIL(-9, "fused recurrent rollout: hidden 128, gate order i,f,o,u, no embedding, no observation filter (got hidden %d, order %d, emb %d)", n.hidden, n.gate_order, n.emb_dim);
  if (!n.wx || !n.wh || !n.b || !n.head_w || !n.head_b || !n.logstd || !n.vf_w || !n.vf_b) FAIL(-10, "learner net: missing weights");
  if (ro->npool < 1) FAIL(-7, "npool %d", ro->npool);
  if (ro->tile_net_dev && ((ro->env_offset & 15) || (E->N & 15))) FAIL(-11, "a snapshot per 16-env tile needs env_offset (%d) and the env count (%d) to be multiples of 16", ro->env_offset, E->N);
  HIPCHK(hipSetDevice(E->device));
  RolloutArgs r;
  memset(&r, 0, sizeof r);
  r.lnet = n; r.onets = ro->opponents_dev; r.tile_net = ro->tile_net_dev; r.st0 = ro->state0; r.st1 = ro->state1;
  r.noise0 = ro->noise0; r.noise1 = ro->noise1;
  r.obs = ro->obs; r.act = ro->act; r.rew = ro->rew; r.val = ro->val; r.nlp = ro->nlp; r.onlp = ro->onlp; r.done = ro->done;
  r.ep_done = ro->ep_done; r.ep_r = ro->ep_r; r.ep_l = ro->ep_l; r.alpha = ro->alpha;
  r.T = ro->T; r.Ntot = ro->Ntot; r.env_offset = ro->env_offset; r.s0 = ro->s0; r.K = ro->K;
  r.XS = (od + 3) & ~3;
  // between two steps of an env nothing from the mass matrix to the end of the float64 area is live (contact records, Jacobian
  // pool and row arrays are rebuilt by every forward): the phase's rows go there, 16-byte aligned for the 128-bit LDS reads
  r.lds_off = (E->L.M + 1) & ~1;
  const size_t need = (size_t)(2 * r.XS + 7 * 128) * sizeof(float), have = (size_t)(E->L.i_base - r.lds_off) * sizeof(double);
  if (E->L.ctrl >= E->L.M || E->L.stash >= E->L.M || need > have) FAIL(-8, "policy scratch (%zu B) does not fit the per-step LDS area (%zu B)", need, have);
  return rollout_launch(E, r, 1, actions_dev, obs_dev, info_dev, done_dev, ep_r_dev, ep_dr_dev, ep_l_dev, stream);
}

#ifdef SUMO_POLICY_PROBE
extern "C" int sumo_debug_tprobe(double* out8, int reset) {   // development: ticks (100 MHz) inside the five sections of trunk_forward
  unsigned long long h[8];
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tprobe), sizeof h));
  for (int k = 0; k < 8; k++) out8[k] = (double)h[k];
  if (reset) { memset(h, 0, sizeof h); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_tprobe), h, sizeof h)); }
  return 0;
}
#endif
extern "C" int sumo_set_cfrc_mode(sumo_handle_t E, int mode) {
  if (!E) FAIL(-1, "bad handle");
  if (mode != 0 && mode != 1) FAIL(-2, "cfrc_mode %d: 0 (zero) or 1 (rne_post)", mode);
  HIPCHK(hipSetDevice(E->device));
  if (mode && !E->d_state_prev) {
    HIPCHK(hipMalloc((void**)&E->d_state_prev, (size_t)E->N * E->state_stride * sizeof(double)));
    HIPCHK(hipMalloc((void**)&E->d_fbuf, (size_t)E->N * 5 * E->L.maxcon * sizeof(double)));
    HIPCHK(hipMalloc((void**)&E->d_cfrc, (size_t)E->N * 6 * E->hm.nbody * sizeof(double)));
    HIPCHK(hipMemset(E->d_cfrc, 0, (size_t)E->N * 6 * E->hm.nbody * sizeof(double)));
  }
  E->cfrc_mode = mode;
  return 0;
}
extern "C" int sumo_set_adjust_z(sumo_handle_t E, double adjust_z) {
  if (!E) FAIL(-1, "bad handle");
  if (!(fabs(adjust_z) <= 1e10)) FAIL(-2, "adjust_z must be finite");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());   // no launch of this engine may be reading the parameter block
  HIPCHK(hipMemcpy((char*)E->d_params + offsetof(Params, adjust_z), &adjust_z, sizeof(double), hipMemcpyHostToDevice));
  E->adjust_z = adjust_z;
  return 0;
}
extern "C" int sumo_get_cfrc_ext(sumo_handle_t E, double* out) {   // HOST float64 [E][nbody][6] of the last step (rne_post mode)
  if (!E || !out) FAIL(-1, "bad arguments");
  if (!E->d_cfrc) FAIL(-2, "cfrc_mode is zero: nothing was computed");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out, E->d_cfrc, (size_t)E->N * 6 * E->hm.nbody * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}
extern "C" int sumo_rollout_status(sumo_handle_t E, int64_t* out4) {
  if (!E) FAIL(-1, "bad handle");
  long long o[4] = {0, 0, E->rollout_tickets < 0 ? 0 : E->rollout_tickets, 0};
  if (E->rollout_tickets >= 0) {
    HIPCHK(hipSetDevice(E->device));
    HIPCHK(hipStreamSynchronize(E->rollout_stream));
    int sc[2];
    unsigned long long st[2];
    HIPCHK(hipMemcpy(sc, E->d_rsched, sizeof sc, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(st, E->d_stats + 9, sizeof st, hipMemcpyDeviceToHost));
    // the counters accumulate over launches: an earlier chunk's abort (whose flag the next launch's memset has cleared) still shows
    const bool cut = sc[1] != 0 || st[0] + st[1] != E->acked_faults;
    E->acked_faults = st[0] + st[1];
    o[0] = cut; o[1] = sc[0]; o[3] = (long long)st[1];
    if (out4) for (int i = 0; i < 4; i++) out4[i] = o[i];
    if (cut)
      FAIL(-20, "fused rollout launch was cut short (abort flag set: %llu expired hand-over waits, %llu hand-over tag / checksum mismatches "
                "since creation; %d of %lld tickets were drawn): the rollout buffers hold unwritten rows and the env states are "
                "partly advanced -- reset the envs before continuing", st[0], st[1], sc[0], E->rollout_tickets);
  } else if (out4) for (int i = 0; i < 4; i++) out4[i] = o[i];
  return 0;
}
// host only (no device is touched): the Layout build_layout() computes for a scene, as ints in declaration order; returns the count.
// tools/gen_static_layout.py turns it into csrc/layout_static.h.
extern "C" int sumo_debug_layout(const void* model_blob, size_t nbytes, int32_t* out, int cap) {
  if (!model_blob || !out) FAIL(-1, "bad arguments");
  sumo_engine E;
  E.blob.assign((const char*)model_blob, (const char*)model_blob + nbytes);
  if (int rc = sumo_model_parse(&E.hm, E.blob.data(), nbytes)) FAIL(-4, "malformed model blob (%d)", rc);
  std::vector<int> ai; std::vector<double> af; std::vector<signed char> pic;
  if (int rc = build_aux(&E, ai, af, pic)) return rc;
  build_layout(&E);
  const int n = (int)(sizeof(Layout) / sizeof(int));
  if (cap < n) FAIL(-2, "need room for %d ints", n);
  memcpy(out, &E.L, sizeof(Layout));
  return n;
}
// the integer members of the compiled model view (header dims, then the int / float table offsets in SUMO_*_TABLES order) and of Aux
static int model_ints(const sumo_model_t* m, int32_t* out) {
  int n = 0;
  out[n++] = m->nq; out[n++] = m->nv; out[n++] = m->nu; out[n++] = m->nbody; out[n++] = m->njnt; out[n++] = m->ngeom; out[n++] = m->npair;
  out[n++] = m->nagent; out[n++] = m->frame_skip; out[n++] = m->timestep_limit;
#define X(name, len) out[n++] = m->o_##name;
  SUMO_INT_TABLES(X)
  SUMO_FLT_TABLES(X)
#undef X
  return n;
}
extern "C" int sumo_debug_model_ints(const void* model_blob, size_t nbytes, int32_t* out, int cap, int32_t* aux_out, int aux_cap) {
  if (!model_blob || !out || !aux_out) FAIL(-1, "bad arguments");
  sumo_engine E;
  E.blob.assign((const char*)model_blob, (const char*)model_blob + nbytes);
  if (int rc = sumo_model_parse(&E.hm, E.blob.data(), nbytes)) FAIL(-4, "malformed model blob (%d)", rc);
  std::vector<int> ai; std::vector<double> af; std::vector<signed char> pic;
  if (int rc = build_aux(&E, ai, af, pic)) return rc;
  if (cap < 128 || aux_cap < AUX_NINTS) FAIL(-2, "buffers too small");
  const int n = model_ints(&E.hm, out);
  memcpy(aux_out, &E.aux.ndepth, AUX_NINTS * sizeof(int));
  return n | (AUX_NINTS << 16);
}
extern "C" int sumo_static_layout(sumo_handle_t E) { return E ? E->static_layout : -1; }   // 1: this engine runs the static-Layout kernel variants
extern "C" int sumo_debug_dump(sumo_handle_t E, double* dev_buf /* [2][8][64] or NULL */) {   // development (-DSUMO_DBG_DUMP builds)
  if (!E) FAIL(-1, "bad handle");
  E->dbg_dump = dev_buf;
  return 0;
}
extern "C" int sumo_debug_fault(sumo_handle_t E, int env) {
  if (!E) FAIL(-1, "bad handle");
  E->dbg_fault_env = env;
  return 0;
}
extern "C" int sumo_debug_trace(sumo_handle_t E, uint64_t* stamps_dev) {
  if (!E) FAIL(-1, "bad handle");
  E->d_trace = (unsigned long long*)stamps_dev;
  return 0;
}

extern "C" int sumo_get_state(sumo_handle_t E, double* qpos, double* qvel, double* warm, int32_t* counters) {
  if (!E) FAIL(-1, "bad handle");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());
  const sumo_model_t* m = &E->hm;
  size_t N = (size_t)E->N;
  std::vector<double> st(N * E->state_stride);
  std::vector<int> cn(N * 4);
  HIPCHK(hipMemcpy(st.data(), E->d_state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cn.data(), E->d_counters, cn.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < N; e++) {
    const double* s = &st[e * E->state_stride];
    if (qpos) memcpy(qpos + e * m->nq, s, m->nq * sizeof(double));
    if (qvel) memcpy(qvel + e * m->nv, s + m->nq, m->nv * sizeof(double));
    if (warm) memcpy(warm + e * m->nv, s + m->nq + m->nv, m->nv * sizeof(double));
    if (counters) { counters[2 * e] = cn[4 * e]; counters[2 * e + 1] = cn[4 * e + 1]; }
  }
  return 0;
}

extern "C" int sumo_set_state(sumo_handle_t E, const double* qpos, const double* qvel, const double* warm, const int32_t* counters) {
  if (!E) FAIL(-1, "bad handle");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());
  const sumo_model_t* m = &E->hm;
  size_t N = (size_t)E->N;
  std::vector<double> st(N * E->state_stride);
  std::vector<int> cn(N * 4);
  HIPCHK(hipMemcpy(st.data(), E->d_state, st.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cn.data(), E->d_counters, cn.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < N; e++) {
    double* s = &st[e * E->state_stride];
    if (qpos) memcpy(s, qpos + e * m->nq, m->nq * sizeof(double));
    if (qvel) memcpy(s + m->nq, qvel + e * m->nv, m->nv * sizeof(double));
    if (warm) memcpy(s + m->nq + m->nv, warm + e * m->nv, m->nv * sizeof(double));
    if (counters) { cn[4 * e] = counters[2 * e]; cn[4 * e + 1] = counters[2 * e + 1]; }
  }
  HIPCHK(hipMemcpy(E->d_state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(E->d_counters, cn.data(), cn.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int sumo_debug_forward(sumo_handle_t E, const double* ctrl, double* qacc, int32_t* counts) {
  if (!E || !ctrl || !qacc || !counts) FAIL(-1, "bad arguments");
  HIPCHK(hipSetDevice(E->device));
  const sumo_model_t* m = &E->hm;
  size_t N = (size_t)E->N;
  double *d_ctrl, *d_qacc;
  int* d_cnt;
  HIPCHK(hipMalloc((void**)&d_ctrl, N * m->nu * sizeof(double)));
  HIPCHK(hipMalloc((void**)&d_qacc, N * m->nv * sizeof(double)));
  HIPCHK(hipMalloc((void**)&d_cnt, N * 4 * sizeof(int)));
  HIPCHK(hipMemcpy(d_ctrl, ctrl, N * m->nu * sizeof(double), hipMemcpyHostToDevice));
  StepArgs a = base_args(E);
  a.stats = nullptr; a.dbg_ctrl = d_ctrl; a.dbg_qacc = d_qacc; a.dbg_counts = d_cnt;
  SUMO_DISPATCH(sumo_forward_kernel, E, (hipStream_t)0, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(qacc, d_qacc, N * m->nv * sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(counts, d_cnt, N * 4 * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d_ctrl); (void)hipFree(d_qacc); (void)hipFree(d_cnt);
  return 0;
}

extern "C" int sumo_profile(sumo_handle_t E, double* out24) {  // cycle totals per phase (SUMO_PROFILE builds only)
  if (!E || !out24) FAIL(-1, "bad arguments");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());
  unsigned long long h[48];
  HIPCHK(hipMemcpy(h, E->d_stats, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < 24; i++) out24[i] = (double)h[16 + i];
  return 0;
}

extern "C" int sumo_stats(sumo_handle_t E, double* out) {
  if (!E || !out) FAIL(-1, "bad arguments");
  HIPCHK(hipSetDevice(E->device));
  HIPCHK(hipDeviceSynchronize());
  unsigned long long h[SUMO_NSTATS];
  HIPCHK(hipMemcpy(h, E->d_stats, sizeof h, hipMemcpyDeviceToHost));
  for (int i = 0; i < SUMO_NSTATS; i++) out[i] = (double)h[i];
  return 0;
}
