/* sumo_oracle.c -- CPU float64 restatement of the RoboSumo env step.  TEST INFRASTRUCTURE ONLY.
 *
 * See sumo_oracle.h for the parity statement ("parity unpinned" for the MuJoCo part).
 *
 * What each block follows:
 *   game step / reward / done  reference robosumo/robosumo/envs/sumo.py:120-202, agents.py:216-223
 *   observation                reference robosumo/robosumo/envs/agents.py:190-214
 *   time feature / epinfo      reference sumo_env.py:40-72, baselines/baselines/bench/monitor.py:51-78
 *   reset                      reference sumo.py:232-253, mujoco_env.py:104-119, agents.py:117-125
 *   auto-reset                 reference subproc_vec_env.py:10-16
 *   frame skip / ctrl write    reference mujoco_env.py:121-129
 *   mj_step (RK4) and the forward-dynamics pipeline: MuJoCo 2.1 semantics as listed in SURVEY.md
 *     Appendix A (the call boundary is reference mujoco-py/mujoco_py/mjsim.pyx:115-129); stage/field names
 *     follow mujoco-py/mujoco_py/pxd/mjdata.pxd:131-285.
 * Plain serial C, dense matrices, no cleverness: this is the checker, not the product.
 */
#include "sumo_oracle.h"
#include "../include/sumo_model.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MINVAL 1e-15
#define MAXCON_CAP 128
#define PI 3.14159265358979323846

static char g_err[256];
const char* so_last_error(void) { return g_err; }

typedef struct {
  double dist, pos[3], frame[9], includemargin, mu, solref[2], solimp[5];
  int g1, g2;
} contact_t;

typedef struct {
  /* persistent state */
  double *qpos, *qvel, *warm;
  int num_steps, reset_count;
  uint64_t seed;
  double ep_ret, ep_dense;
  /* scratch (mjData) */
  double *ctrl, *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *gxpos, *gxmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *M, *L, *qfrc_bias, *qfrc_passive, *qfrc_act, *qfrc_smooth, *qacc_smooth, *qacc, *qfrc_constraint;
  contact_t* con;
  int ncon, nefc, ncon_dropped;
  double *J, *epos, *emargin, *ediag, *eR, *eD, *eK, *eB, *eimp, *evel, *earef, *ejar, *eforce;
  int* etype; /* 0 limit, 1 contact */
  double *Ma, *grad, *Mgrad, *search, *Mv, *Jv, *H, *tmpv;
  double *rkX[4], *rkF[4], *rkdX;
  long n_forward, n_newton, n_contacts, n_efc, max_ncon, max_nefc, max_newton, n_diverged;
  long n_cb3, n_rodcap; /* fidelity accounting: capsule-box calls with 3 active contacts; active contacts on a border rod beyond the cylinder's flat end */
  double* cfrc_ext;   /* [nbody][6] (cfrc_mode = rne_post): contact wrenches about the root's subtree CoM, world axes, [torque ; force] */
} env_t;

struct so_sim {
  void* blob;
  sumo_model_t m;
  int N, maxcon, jbcap, maxefc, obs_stride, act_stride;
  int cfrc_mode;      /* 0: cfrc_ext == 0 (the reference's MuJoCo 2.1 without force sensors); 1: as mj_rnePostConstraint fills it */
  double adjust_z;    /* Agent._adjust_z (agents.py:33): 0 in training (run.py:76-77), -0.5 in the evaluation scripts (eval_robosumo_against_fix.py:108-115) */
  env_t* env;
};

/* ------------------------------------------------------------------------------------------------
 * small math
 * ---------------------------------------------------------------------------------------------- */
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double normalize3(double* v) {
  double n = sqrt(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; } else { v[0] /= n; v[1] /= n; v[2] /= n; }
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void mulquat(double* r, const double* a, const double* b) {
  double t[4];
  t[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  t[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  t[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  t[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  memcpy(r, t, sizeof t);
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mulmatvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmatTvec3(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2],
         z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
/* spatial inertia (10 numbers: Ixx Iyy Izz Ixy Ixz Iyz, m*c, m) times motion vector [rot; lin] */
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static void cross_motion(double* r, const double* vel, const double* v) {
  r[0] = -vel[2] * v[1] + vel[1] * v[2];
  r[1] = vel[2] * v[0] - vel[0] * v[2];
  r[2] = -vel[1] * v[0] + vel[0] * v[1];
  r[3] = -vel[2] * v[4] + vel[1] * v[5] - vel[5] * v[1] + vel[4] * v[2];
  r[4] = vel[2] * v[3] - vel[0] * v[5] + vel[5] * v[0] - vel[3] * v[2];
  r[5] = -vel[1] * v[3] + vel[0] * v[4] - vel[4] * v[0] + vel[3] * v[1];
}
static void cross_force(double* r, const double* vel, const double* f) {
  r[0] = -vel[2] * f[1] + vel[1] * f[2] - vel[5] * f[4] + vel[4] * f[5];
  r[1] = vel[2] * f[0] - vel[0] * f[2] + vel[5] * f[3] - vel[3] * f[5];
  r[2] = -vel[1] * f[0] + vel[0] * f[1] - vel[4] * f[3] + vel[3] * f[4];
  r[3] = -vel[2] * f[4] + vel[1] * f[5];
  r[4] = vel[2] * f[3] - vel[0] * f[5];
  r[5] = -vel[1] * f[3] + vel[0] * f[4];
}

/* ------------------------------------------------------------------------------------------------
 * position stage: kinematics, CoM quantities, mass matrix (mjdata.pxd:160-236)
 * ---------------------------------------------------------------------------------------------- */
static void kinematics(const sumo_model_t* m, env_t* d) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* jntadr = SUMO_I(m, body_jntadr);
  const int* jntnum = SUMO_I(m, body_jntnum);
  const int* jtype = SUMO_I(m, jnt_type);
  const int* jqadr = SUMO_I(m, jnt_qposadr);
  const double* qpos0 = SUMO_F(m, qpos0);
  for (int k = 0; k < 3; k++) d->xpos[k] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    double* xpos = d->xpos + 3 * b;
    double* xquat = d->xquat + 4 * b;
    int pid = parent[b], ja = jntadr[b], jn = jntnum[b];
    if (jn == 1 && jtype[ja] == SUMO_JNT_FREE) {
      double* q = d->qpos + jqadr[ja];
      normalize4(q + 3); /* mj_kinematics normalises the quaternion in qpos in place */
      memcpy(xpos, q, 3 * sizeof(double));
      memcpy(xquat, q + 3, 4 * sizeof(double));
      memcpy(d->xanchor + 3 * ja, xpos, 3 * sizeof(double));
      quat2mat(d->xmat + 9 * b, xquat);
      mulmatvec3(d->xaxis + 3 * ja, d->xmat + 9 * b, SUMO_F(m, jnt_axis) + 3 * ja);
    } else {
      double v[3];
      mulmatvec3(v, d->xmat + 9 * pid, SUMO_F(m, body_pos) + 3 * b);
      for (int k = 0; k < 3; k++) xpos[k] = d->xpos[3 * pid + k] + v[k];
      mulquat(xquat, d->xquat + 4 * pid, SUMO_F(m, body_quat) + 4 * b);
      for (int j = ja; j < ja + jn; j++) {
        double R[9], ql[4];
        quat2mat(R, xquat);
        mulmatvec3(v, R, SUMO_F(m, jnt_pos) + 3 * j);
        for (int k = 0; k < 3; k++) d->xanchor[3 * j + k] = xpos[k] + v[k];
        mulmatvec3(d->xaxis + 3 * j, R, SUMO_F(m, jnt_axis) + 3 * j);
        axisangle2quat(ql, SUMO_F(m, jnt_axis) + 3 * j, d->qpos[jqadr[j]] - qpos0[jqadr[j]]);
        mulquat(xquat, xquat, ql);
        quat2mat(R, xquat);
        mulmatvec3(v, R, SUMO_F(m, jnt_pos) + 3 * j);
        for (int k = 0; k < 3; k++) xpos[k] = d->xanchor[3 * j + k] - v[k];
      }
      normalize4(xquat);
      quat2mat(d->xmat + 9 * b, xquat);
    }
    /* inertial frame */
    double v[3], q[4];
    mulmatvec3(v, d->xmat + 9 * b, SUMO_F(m, body_ipos) + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = xpos[k] + v[k];
    mulquat(q, xquat, SUMO_F(m, body_iquat) + 4 * b);
    quat2mat(d->ximat + 9 * b, q);
  }
  const int* gbody = SUMO_I(m, geom_bodyid);
  for (int g = 0; g < m->ngeom; g++) {
    int b = gbody[g];
    double v[3], q[4];
    mulmatvec3(v, d->xmat + 9 * b, SUMO_F(m, geom_pos) + 3 * g);
    for (int k = 0; k < 3; k++) d->gxpos[3 * g + k] = d->xpos[3 * b + k] + v[k];
    mulquat(q, d->xquat + 4 * b, SUMO_F(m, geom_quat) + 4 * g);
    quat2mat(d->gxmat + 9 * g, q);
  }
}

static void com_pos(const sumo_model_t* m, env_t* d) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* rootid = SUMO_I(m, body_rootid);
  const double* mass = SUMO_F(m, body_mass);
  const double* stm = SUMO_F(m, body_subtreemass);
  int nb = m->nbody;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = mass[b] * d->xipos[3 * b + k];
  for (int b = nb - 1; b > 0; b--)
    for (int k = 0; k < 3; k++) d->subtree_com[3 * parent[b] + k] += d->subtree_com[3 * b + k];
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 3; k++)
      d->subtree_com[3 * b + k] = stm[b] < MINVAL ? d->xipos[3 * b + k] : d->subtree_com[3 * b + k] / stm[b];
  /* body inertias about the root's subtree CoM, world orientation */
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    const double* inert = SUMO_F(m, body_inertia) + 3 * b;
    const double* R = d->ximat + 9 * b;
    double dif[3], T[9], *res = d->cinert + 10 * b;
    for (int k = 0; k < 3; k++) dif[k] = d->xipos[3 * b + k] - d->subtree_com[3 * rootid[b] + k];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++)
        T[3 * r + c] = R[3 * r] * inert[0] * R[3 * c] + R[3 * r + 1] * inert[1] * R[3 * c + 1] +
                       R[3 * r + 2] * inert[2] * R[3 * c + 2];
    double ms = mass[b];
    res[0] = T[0] + ms * (dif[1] * dif[1] + dif[2] * dif[2]);
    res[1] = T[4] + ms * (dif[0] * dif[0] + dif[2] * dif[2]);
    res[2] = T[8] + ms * (dif[0] * dif[0] + dif[1] * dif[1]);
    res[3] = T[1] - ms * dif[0] * dif[1];
    res[4] = T[2] - ms * dif[0] * dif[2];
    res[5] = T[5] - ms * dif[1] * dif[2];
    res[6] = ms * dif[0]; res[7] = ms * dif[1]; res[8] = ms * dif[2];
    res[9] = ms;
  }
  /* motion axes of every dof about the root's subtree CoM */
  const int* jtype = SUMO_I(m, jnt_type);
  const int* jdof = SUMO_I(m, jnt_dofadr);
  const int* jbody = SUMO_I(m, jnt_bodyid);
  for (int j = 0; j < m->njnt; j++) {
    int b = jbody[j], da = jdof[j];
    double off[3];
    for (int k = 0; k < 3; k++) off[k] = d->subtree_com[3 * rootid[b] + k] - d->xanchor[3 * j + k];
    if (jtype[j] == SUMO_JNT_FREE) {
      memset(d->cdof + 6 * da, 0, 18 * sizeof(double));
      for (int k = 0; k < 3; k++) d->cdof[6 * (da + k) + 3 + k] = 1;
      for (int k = 0; k < 3; k++) {
        double ax[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        double* c = d->cdof + 6 * (da + 3 + k);
        c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
        cross3(c + 3, ax, off);
      }
    } else {
      double* c = d->cdof + 6 * da;
      const double* ax = d->xaxis + 3 * j;
      c[0] = ax[0]; c[1] = ax[1]; c[2] = ax[2];
      cross3(c + 3, ax, off);
    }
  }
}

static void crb_and_factor(const sumo_model_t* m, env_t* d) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* dbody = SUMO_I(m, dof_bodyid);
  const int* dpar = SUMO_I(m, dof_parentid);
  const double* arm = SUMO_F(m, dof_armature);
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, 10 * nb * sizeof(double));
  for (int b = nb - 1; b > 0; b--)
    if (parent[b] > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * parent[b] + k] += d->crb[10 * b + k];
  memset(d->M, 0, nv * nv * sizeof(double));
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10 * dbody[i], d->cdof + 6 * i);
    d->M[i * nv + i] = arm[i];
    for (int j = i; j >= 0; j = dpar[j]) {
      double v = dot6(d->cdof + 6 * j, buf);
      d->M[i * nv + j] += v;
      if (j != i) d->M[j * nv + i] += v;
    }
  }
}

/* dense Cholesky A = L L^T (lower), in place into L; returns 0 ok */
static int cholesky(double* L, const double* A, int n) {
  memcpy(L, A, n * n * sizeof(double));
  for (int j = 0; j < n; j++) {
    double s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (s < MINVAL) return -1;
    s = sqrt(s);
    L[j * n + j] = s;
    for (int i = j + 1; i < n; i++) {
      double t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / s;
    }
  }
  return 0;
}
static void chol_solve(double* x, const double* L, const double* b, int n) {
  for (int i = 0; i < n; i++) {
    double t = b[i];
    for (int k = 0; k < i; k++) t -= L[i * n + k] * x[k];
    x[i] = t / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= L[k * n + i] * x[k];
    x[i] = t / L[i * n + i];
  }
}

/* ------------------------------------------------------------------------------------------------
 * collision (mjdata.pxd:43-69 contact record)
 * ---------------------------------------------------------------------------------------------- */
static void make_frame(double* f) {
  /* MuJoCo mju_makeFrame: f[0..2] given (unit normal); f[3..5] optional */
  double tmp[3];
  if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double dd = dot3(f, f + 3);
  for (int k = 0; k < 3; k++) tmp[k] = f[k] * dd;
  for (int k = 0; k < 3; k++) f[3 + k] -= tmp[k];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

static int sphere_sphere(contact_t* c, double margin, const double* p1, double r1, const double* p2, double r2) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double cd2 = dot3(dif, dif), mind = margin + r1 + r2;
  if (cd2 > mind * mind) return 0;
  double len = sqrt(cd2);
  if (len < MINVAL) { dif[0] = 0; dif[1] = 0; dif[2] = 1; } else { dif[0] /= len; dif[1] /= len; dif[2] /= len; }
  c->dist = len - r1 - r2;
  for (int k = 0; k < 3; k++) { c->frame[k] = dif[k]; c->frame[3 + k] = 0; c->pos[k] = p1[k] + dif[k] * (r1 + 0.5 * c->dist); }
  return 1;
}
static int plane_sphere(contact_t* c, double margin, const double* pp, const double* pm, const double* sp, double r) {
  double n[3] = {pm[2], pm[5], pm[8]};
  double t[3] = {sp[0] - pp[0], sp[1] - pp[1], sp[2] - pp[2]};
  double cd = dot3(t, n);
  if (cd > margin + r) return 0;
  c->dist = cd - r;
  for (int k = 0; k < 3; k++) { c->frame[k] = n[k]; c->frame[3 + k] = 0; c->pos[k] = sp[k] - n[k] * (r + 0.5 * c->dist); }
  return 1;
}
static int plane_capsule(contact_t* c, double margin, const double* pp, const double* pm, const double* cp,
                         const double* cm, const double* cs) {
  double ax[3] = {cm[2] * cs[1], cm[5] * cs[1], cm[8] * cs[1]};
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) {
    double e[3] = {cp[0] + s * ax[0], cp[1] + s * ax[1], cp[2] + s * ax[2]};
    n += plane_sphere(c + n, margin, pp, pm, e, cs[0]);
  }
  return n;
}
static int sphere_capsule(contact_t* c, double margin, const double* sp, double r, const double* cp,
                          const double* cm, const double* cs) {
  double ax[3] = {cm[2], cm[5], cm[8]};
  double v[3] = {sp[0] - cp[0], sp[1] - cp[1], sp[2] - cp[2]};
  double x = dot3(ax, v);
  if (x > cs[1]) x = cs[1];
  if (x < -cs[1]) x = -cs[1];
  double e[3] = {cp[0] + ax[0] * x, cp[1] + ax[1] * x, cp[2] + ax[2] * x};
  return sphere_sphere(c, margin, sp, r, e, cs[0]);
}
static int capsule_capsule(contact_t* c, double margin, const double* p1, const double* m1, const double* s1,
                           const double* p2, const double* m2, const double* s2) {
  double a1[3] = {m1[2] * s1[1], m1[5] * s1[1], m1[8] * s1[1]};
  double a2[3] = {m2[2] * s2[1], m2[5] * s2[1], m2[8] * s2[1]};
  double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = ma * mc - mb * mb;
  double v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > 1) { x1 = 1; x2 = (v - mb) / mc; }
    else if (x1 < -1) { x1 = -1; x2 = (v + mb) / mc; }
    if (x2 > 1) { x2 = 1; x1 = (u - mb) / ma; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
    else if (x2 < -1) { x2 = -1; x1 = (u + mb) / ma; if (x1 > 1) x1 = 1; else if (x1 < -1) x1 = -1; }
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k] * x1; v2[k] = p2[k] + a2[k] * x2; }
    return sphere_sphere(c, margin, v1, s1[0], v2, s2[0]);
  }
  /* parallel axes: test both ends of capsule 1 against the clamped projection on capsule 2 */
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) {
    double x2 = (v - s * mb) / mc;
    if (x2 > 1) x2 = 1; else if (x2 < -1) x2 = -1;
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + s * a1[k]; v2[k] = p2[k] + a2[k] * x2; }
    n += sphere_sphere(c + n, margin, v1, s1[0], v2, s2[0]);
  }
  return n;
}
static int sphere_box(contact_t* c, double margin, const double* sp, double r, const double* bp, const double* bm,
                      const double* bs) {
  double t[3] = {sp[0] - bp[0], sp[1] - bp[1], sp[2] - bp[2]}, ctr[3], cl[3], dl[3];
  mulmatTvec3(ctr, bm, t);
  for (int k = 0; k < 3; k++) { cl[k] = ctr[k] > bs[k] ? bs[k] : (ctr[k] < -bs[k] ? -bs[k] : ctr[k]); dl[k] = cl[k] - ctr[k]; }
  double dist = sqrt(dot3(dl, dl));
  if (dist - r > margin) return 0;
  double nl[3], pl[3];
  if (dist <= MINVAL) {
    /* centre inside the box: push out through the nearest face */
    double best = 1e300; int kb = 0, sb = 1;
    for (int k = 0; k < 3; k++)
      for (int s = -1; s <= 1; s += 2) {
        double fd = fabs(s * bs[k] - ctr[k]);
        if (fd < best) { best = fd; kb = k; sb = s; }
      }
    nl[0] = nl[1] = nl[2] = 0; nl[kb] = -sb;
    c->dist = -best - r;
  } else {
    for (int k = 0; k < 3; k++) nl[k] = dl[k] / dist;
    c->dist = dist - r;
  }
  for (int k = 0; k < 3; k++) pl[k] = ctr[k] + nl[k] * (r + 0.5 * c->dist); /* midpoint of the two surfaces */
  double pw[3];
  mulmatvec3(c->frame, bm, nl);
  mulmatvec3(pw, bm, pl);
  for (int k = 0; k < 3; k++) { c->pos[k] = pw[k] + bp[k]; c->frame[3 + k] = 0; }
  return 1;
}
/* derivative (up to a factor 2) of the squared distance from segment point c + t*a to the box, box frame */
static double seg_box_dgrad(const double* cc, const double* a, const double* bs, double t, int skip) {
  double g = 0; /* skip: the coordinate that sits exactly on a box face at t (its own breakpoint) contributes exactly 0 */
  for (int k = 0; k < 3; k++) {
    if (k == skip) continue;
    double p = cc[k] + t * a[k];
    if (p > bs[k]) g += a[k] * (p - bs[k]);
    else if (p < -bs[k]) g += a[k] * (p + bs[k]);
  }
  return g;
}
/* capsule-box: both end spheres, plus the sphere at the interior closest point of the segment when the
 * squared-distance derivative changes sign strictly inside the segment (documented deviation from MuJoCo's
 * mjc_CapsuleBox feature search, whose source is not available offline). */
static int capsule_box(contact_t* c, double margin, const double* cp, const double* cm, const double* cs,
                       const double* bp, const double* bm, const double* bs) {
  double axw[3] = {cm[2] * cs[1], cm[5] * cs[1], cm[8] * cs[1]};
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) {
    double e[3] = {cp[0] + s * axw[0], cp[1] + s * axw[1], cp[2] + s * axw[2]};
    n += sphere_box(c + n, margin, e, cs[0], bp, bm, bs);
  }
  double t[3] = {cp[0] - bp[0], cp[1] - bp[1], cp[2] - bp[2]}, cc[3], a[3];
  mulmatTvec3(cc, bm, t);
  mulmatTvec3(a, bm, axw);
  double glo = seg_box_dgrad(cc, a, bs, -1.0, -1), ghi = seg_box_dgrad(cc, a, bs, 1.0, -1);
  if (glo < 0 && ghi > 0) {
    /* the derivative is piecewise linear and non-decreasing in t: bracket the root between consecutive breakpoints
     * (where a coordinate crosses a box face), then interpolate exactly */
    double lo = -1, hi = 1;
    for (int k = 0; k < 3; k++) {
      double ra = 1.0 / a[k]; /* a[k] == 0: the breakpoints are inf / nan and fail the comparisons below */
      for (int sg = 0; sg < 2; sg++) {
        double tb = ((sg ? -bs[k] : bs[k]) - cc[k]) * ra;
        if (tb > lo && tb < hi) {
          double gb = seg_box_dgrad(cc, a, bs, tb, k);
          if (gb > 0) { hi = tb; ghi = gb; } else { lo = tb; glo = gb; }
        }
      }
    }
    double ts = lo - glo * (hi - lo) / (ghi - glo);
    double e[3] = {cp[0] + ts * axw[0], cp[1] + ts * axw[1], cp[2] + ts * axw[2]};
    n += sphere_box(c + n, margin, e, cs[0], bp, bm, bs);
  }
  return n;
}

static void collision(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  const int* g1a = SUMO_I(m, pair_geom1);
  const int* g2a = SUMO_I(m, pair_geom2);
  const int* gtype = SUMO_I(m, geom_type);
  const double* gsize = SUMO_F(m, geom_size);
  const double* rb = SUMO_F(m, geom_rbound);
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = g1a[p], g2 = g2a[p], t1 = gtype[g1], t2 = gtype[g2];
    double margin = SUMO_F(m, pair_margin)[p];
    const double *p1 = d->gxpos + 3 * g1, *p2 = d->gxpos + 3 * g2, *m1 = d->gxmat + 9 * g1, *m2 = d->gxmat + 9 * g2;
    const double *s1 = gsize + 3 * g1, *s2 = gsize + 3 * g2;
    /* broad phase: bounding spheres (plane: signed distance of the centre) */
    if (t1 == SUMO_GEOM_PLANE) {
      double n[3] = {m1[2], m1[5], m1[8]}, t[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      if (dot3(t, n) > margin + rb[g2]) continue;
    } else {
      double t[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, bound = margin + rb[g1] + rb[g2];
      if (dot3(t, t) > bound * bound) continue;
    }
    contact_t tmp[4];
    int n = 0;
    /* cylinders (the four border rods) are treated as capsules of the same radius / half length */
    int u1 = t1 == SUMO_GEOM_CYLINDER ? SUMO_GEOM_CAPSULE : t1;
    int u2 = t2 == SUMO_GEOM_CYLINDER ? SUMO_GEOM_CAPSULE : t2;
    /* the compiler orders every pair by geom type, so u1 <= u2 here */
    {
      if (u1 == SUMO_GEOM_PLANE && u2 == SUMO_GEOM_SPHERE) n = plane_sphere(tmp, margin, p1, m1, p2, s2[0]);
      else if (u1 == SUMO_GEOM_PLANE && u2 == SUMO_GEOM_CAPSULE) n = plane_capsule(tmp, margin, p1, m1, p2, m2, s2);
      else if (u1 == SUMO_GEOM_SPHERE && u2 == SUMO_GEOM_SPHERE) n = sphere_sphere(tmp, margin, p1, s1[0], p2, s2[0]);
      else if (u1 == SUMO_GEOM_SPHERE && u2 == SUMO_GEOM_CAPSULE) n = sphere_capsule(tmp, margin, p1, s1[0], p2, m2, s2);
      else if (u1 == SUMO_GEOM_CAPSULE && u2 == SUMO_GEOM_CAPSULE) n = capsule_capsule(tmp, margin, p1, m1, s1, p2, m2, s2);
      else if (u1 == SUMO_GEOM_SPHERE && u2 == SUMO_GEOM_BOX) n = sphere_box(tmp, margin, p1, s1[0], p2, m2, s2);
      else if (u1 == SUMO_GEOM_CAPSULE && u2 == SUMO_GEOM_BOX) n = capsule_box(tmp, margin, p1, m1, s1, p2, m2, s2);
      else n = 0; /* plane-box etc.: static-static, filtered at compile time */
    }
    if ((d->n_forward - 1) % 20 == 0) { /* sampled like the engine: the forward evaluation that opens an env step (1 in 20) */
      int nact = 0;
      for (int i = 0; i < n; i++) nact += tmp[i].dist < margin;
      if (nact == 3) d->n_cb3++;
      if (t2 == SUMO_GEOM_CYLINDER) { /* the rod's axis is the z axis of its frame, half length size[1] */
        for (int i = 0; i < n; i++) {
          if (!(tmp[i].dist < margin)) continue;
          double d3[3] = {tmp[i].pos[0] - p2[0], tmp[i].pos[1] - p2[1], tmp[i].pos[2] - p2[2]}, ax[3] = {m2[2], m2[5], m2[8]};
          if (fabs(dot3(d3, ax)) > s2[1]) d->n_rodcap++;
        }
      }
    }
    for (int i = 0; i < n; i++) {
      if (!(tmp[i].dist < margin)) continue; /* active iff dist < includemargin (gap = 0) */
      if (d->ncon >= s->maxcon) { d->ncon_dropped++; continue; }
      contact_t* c = d->con + d->ncon++;
      *c = tmp[i];
      make_frame(c->frame);
      c->includemargin = margin - SUMO_F(m, pair_gap)[p];
      c->mu = SUMO_F(m, pair_friction)[3 * p];
      memcpy(c->solref, SUMO_F(m, pair_solref) + 2 * p, 2 * sizeof(double));
      memcpy(c->solimp, SUMO_F(m, pair_solimp) + 5 * p, 5 * sizeof(double));
      c->g1 = g1; c->g2 = g2;
    }
  }
  /* The HIP engine keeps contact Jacobians in a pool of `jbcap` "halves" (one per moving body of a contact; sumo_dims):
   * the first contact, in pair order, that does not fit and all later ones are dropped and counted.  Mirrored here so the
   * two stay comparable when the pool overflows (many simultaneous contacts between moving bodies).  jbcap 0 = no pool limit. */
  {
    const int* gbody = SUMO_I(m, geom_bodyid);
    int cap = s->jbcap > 0 ? s->jbcap : 2 * s->maxcon, used = 0;
    for (int i = 0; i < d->ncon; i++) {
      int nh = (gbody[d->con[i].g1] != 0 && gbody[d->con[i].g2] != 0) ? 2 : 1;
      if (used + nh > cap) { d->ncon_dropped += d->ncon - i; d->ncon = i; break; }
      used += nh;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * constraints (mjdata.pxd:210-230)
 * ---------------------------------------------------------------------------------------------- */
static void jac_point(const sumo_model_t* m, const env_t* d, int body, const double* point, double* jacp /*3 x nv*/) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* rootid = SUMO_I(m, body_rootid);
  const int* dofnum = SUMO_I(m, body_dofnum);
  const int* dofadr = SUMO_I(m, body_dofadr);
  const int* dpar = SUMO_I(m, dof_parentid);
  int nv = m->nv;
  memset(jacp, 0, 3 * nv * sizeof(double));
  while (body && dofnum[body] == 0) body = parent[body];
  if (!body) return;
  double off[3];
  for (int k = 0; k < 3; k++) off[k] = point[k] - d->subtree_com[3 * rootid[body] + k];
  for (int i = dofadr[body] + dofnum[body] - 1; i >= 0; i = dpar[i]) {
    const double* c = d->cdof + 6 * i;
    double t[3];
    cross3(t, c, off);
    for (int k = 0; k < 3; k++) jacp[k * nv + i] = c[3 + k] + t[k];
  }
}

static void impedance(const double* solimp, double x, double* imp) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin < 0.0001) dmin = 0.0001; if (dmin > 0.9999) dmin = 0.9999;
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  if (width < MINVAL) width = MINVAL;
  if (mid < 0.0001) mid = 0.0001; if (mid > 0.9999) mid = 0.9999;
  if (power < 1) power = 1;
  double xx = fabs(x) / width, y;
  if (xx >= 1) { *imp = dmax; return; }
  if (xx <= 0) { *imp = dmin; return; }
  if (power == 1) y = xx;
  else if (xx <= mid) y = pow(xx, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - xx, power) / pow(1 - mid, power - 1);
  *imp = dmin + y * (dmax - dmin);
}

static void add_row_params(const sumo_model_t* m, env_t* d, int r, const double* solref, const double* solimp,
                           double pos, double margin, double diag) {
  double timestep = SUMO_F(m, opt)[SUMO_OPT_TIMESTEP];
  double tc = solref[0], dr = solref[1], dmax = solimp[1], imp;
  if (dmax < 0.0001) dmax = 0.0001; if (dmax > 0.9999) dmax = 0.9999;
  if (tc < 2 * timestep) tc = 2 * timestep; /* refsafe */
  impedance(solimp, pos - margin, &imp);
  double kk = dmax * dmax * tc * tc * dr * dr, bb = dmax * tc;
  d->eK[r] = 1.0 / (kk < MINVAL ? MINVAL : kk);
  d->eB[r] = 2.0 / (bb < MINVAL ? MINVAL : bb);
  d->eimp[r] = imp;
  d->epos[r] = pos;
  d->emargin[r] = margin;
  d->ediag[r] = diag;
  double R = (1 - imp) * diag / imp;
  d->eR[r] = R < MINVAL ? MINVAL : R;
}

static void make_constraint(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  int nv = m->nv;
  const int* jtype = SUMO_I(m, jnt_type);
  const int* jlim = SUMO_I(m, jnt_limited);
  const int* jq = SUMO_I(m, jnt_qposadr);
  const int* jd = SUMO_I(m, jnt_dofadr);
  const double* range = SUMO_F(m, jnt_range);
  const double* jmargin = SUMO_F(m, jnt_margin);
  const double* dinv = SUMO_F(m, dof_invweight0);
  const double* binv = SUMO_F(m, body_invweight0);
  const int* gbody = SUMO_I(m, geom_bodyid);
  static const double def_solref[2] = {0.02, 1.0};
  static const double def_solimp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
  int r = 0;
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (jtype[j] != SUMO_JNT_HINGE || !jlim[j]) continue;
    double value = d->qpos[jq[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (range[2 * j + (side + 1) / 2] - value);
      if (dist < jmargin[j] && r < s->maxefc) {
        memset(d->J + r * nv, 0, nv * sizeof(double));
        d->J[r * nv + jd[j]] = -side;
        d->etype[r] = 0;
        add_row_params(m, d, r, def_solref, def_solimp, dist, jmargin[j], dinv[jd[j]]);
        r++;
      }
    }
  }
  /* pyramidal contacts, condim 3: rows n+mu*t1, n-mu*t1, n+mu*t2, n-mu*t2 */
  double* j1 = d->tmpv;            /* 3 x nv */
  double* j2 = d->tmpv + 3 * nv;   /* 3 x nv */
  double* jc = d->tmpv + 6 * nv;   /* 3 x nv in contact frame */
  for (int ci = 0; ci < d->ncon; ci++) {
    contact_t* c = d->con + ci;
    int b1 = gbody[c->g1], b2 = gbody[c->g2];
    jac_point(m, d, b1, c->pos, j1);
    jac_point(m, d, b2, c->pos, j2);
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < nv; i++) {
        double v = 0;
        for (int k = 0; k < 3; k++) v += c->frame[3 * a + k] * (j2[k * nv + i] - j1[k * nv + i]);
        jc[a * nv + i] = v;
      }
    double tran = binv[2 * b1] + binv[2 * b2];
    double mu = c->mu;
    double diag = tran + mu * mu * tran;
    int r0 = r;
    for (int k = 1; k <= 2; k++)
      for (int sgn = 1; sgn >= -1; sgn -= 2) {
        if (r >= s->maxefc) break;
        for (int i = 0; i < nv; i++) d->J[r * nv + i] = jc[i] + sgn * mu * jc[k * nv + i];
        d->etype[r] = 1;
        add_row_params(m, d, r, c->solref, c->solimp, c->dist, c->includemargin, diag);
        r++;
      }
    /* pyramid regularisation: every edge gets 2*mu^2*R(first row) */
    double Rpy = 2 * mu * mu * d->eR[r0];
    if (Rpy < MINVAL) Rpy = MINVAL;
    for (int q = r0; q < r; q++) d->eR[q] = Rpy;
  }
  d->nefc = r;
  for (int i = 0; i < r; i++) d->eD[i] = 1.0 / d->eR[i];
}

/* ------------------------------------------------------------------------------------------------
 * velocity stage (mjdata.pxd:238-262)
 * ---------------------------------------------------------------------------------------------- */
static void com_vel(const sumo_model_t* m, env_t* d) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* dofnum = SUMO_I(m, body_dofnum);
  const int* dofadr = SUMO_I(m, body_dofadr);
  const int* djnt = SUMO_I(m, dof_jntid);
  const int* jtype = SUMO_I(m, jnt_type);
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6];
    memcpy(cvel, d->cvel + 6 * parent[b], sizeof cvel);
    int bda = dofadr[b], n = dofnum[b];
    for (int j = 0; j < n; j++) {
      int i = bda + j;
      if (jtype[djnt[i]] == SUMO_JNT_FREE) {
        memset(d->cdof_dot + 6 * i, 0, 18 * sizeof(double));
        for (int k = 0; k < 3; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (i + k) + c] * d->qvel[i + k];
        for (int k = 3; k < 6; k++) cross_motion(d->cdof_dot + 6 * (i + k), cvel, d->cdof + 6 * (i + k));
        for (int k = 3; k < 6; k++)
          for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (i + k) + c] * d->qvel[i + k];
        j += 5;
      } else {
        cross_motion(d->cdof_dot + 6 * i, cvel, d->cdof + 6 * i);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * i + c] * d->qvel[i];
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof cvel);
  }
}

static void rne_bias(const sumo_model_t* m, env_t* d) {
  const int* parent = SUMO_I(m, body_parentid);
  const int* dofnum = SUMO_I(m, body_dofnum);
  const int* dofadr = SUMO_I(m, body_dofadr);
  const int* dbody = SUMO_I(m, dof_bodyid);
  const double* g = SUMO_F(m, opt) + SUMO_OPT_GRAVITY;
  int nb = m->nbody;
  memset(d->cacc, 0, 6 * sizeof(double));
  for (int k = 0; k < 3; k++) d->cacc[3 + k] = -g[k];
  memset(d->cfrc, 0, 6 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    double* cacc = d->cacc + 6 * b;
    memcpy(cacc, d->cacc + 6 * parent[b], 6 * sizeof(double));
    for (int j = 0; j < dofnum[b]; j++) {
      int i = dofadr[b] + j;
      for (int c = 0; c < 6; c++) cacc[c] += d->cdof_dot[6 * i + c] * d->qvel[i];
    }
    double t[6], t1[6];
    mul_inert_vec(d->cfrc + 6 * b, d->cinert + 10 * b, cacc);
    mul_inert_vec(t, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(t1, d->cvel + 6 * b, t);
    for (int c = 0; c < 6; c++) d->cfrc[6 * b + c] += t1[c];
  }
  for (int b = nb - 1; b > 0; b--)
    if (parent[b])
      for (int c = 0; c < 6; c++) d->cfrc[6 * parent[b] + c] += d->cfrc[6 * b + c];
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof + 6 * i, d->cfrc + 6 * dbody[i]);
}

/* ------------------------------------------------------------------------------------------------
 * constraint solver: primal Newton with exact line search (mjmodel.pxd:192-195 selects Newton;
 * SURVEY.md App. A.8).  The optimum is unique (strictly convex), so any converged solver agrees.
 * ---------------------------------------------------------------------------------------------- */
static double solver_cost(const env_t* d, int nv, int nefc, const double* x, const double* Ma, const double* jar) {
  double c = 0;
  for (int i = 0; i < nv; i++) c += 0.5 * (Ma[i] - d->qfrc_smooth[i]) * (x[i] - d->qacc_smooth[i]);
  for (int r = 0; r < nefc; r++)
    if (jar[r] < 0) c += 0.5 * d->eD[r] * jar[r] * jar[r];
  return c;
}
static void matvec(double* y, const double* A, const double* x, int rows, int cols) {
  for (int r = 0; r < rows; r++) {
    double t = 0;
    for (int c = 0; c < cols; c++) t += A[r * cols + c] * x[c];
    y[r] = t;
  }
}

static void newton_solve(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  int nv = m->nv, nefc = d->nefc;
  const double* opt = SUMO_F(m, opt);
  double tol = opt[SUMO_OPT_TOLERANCE];
  int maxiter = (int)opt[SUMO_OPT_ITERATIONS];
  double scale = 1.0 / (opt[SUMO_OPT_MEANINERTIA] * (nv > 1 ? nv : 1));
  double* x = d->qacc;
  if (nefc == 0) {
    memcpy(x, d->qacc_smooth, nv * sizeof(double));
    memset(d->qfrc_constraint, 0, nv * sizeof(double));
    return;
  }
  /* warm start: better of qacc_warmstart and qacc_smooth */
  memcpy(x, d->warm, nv * sizeof(double));
  matvec(d->Ma, d->M, x, nv, nv);
  matvec(d->ejar, d->J, x, nefc, nv);
  for (int r = 0; r < nefc; r++) d->ejar[r] -= d->earef[r];
  double cost_ws = solver_cost(d, nv, nefc, x, d->Ma, d->ejar);
  /* cost at qacc_smooth: gauss term is zero */
  matvec(d->Jv, d->J, d->qacc_smooth, nefc, nv);
  double cost_sm = 0;
  for (int r = 0; r < nefc; r++) {
    double j = d->Jv[r] - d->earef[r];
    if (j < 0) cost_sm += 0.5 * d->eD[r] * j * j;
  }
  if (cost_ws > cost_sm) {
    memcpy(x, d->qacc_smooth, nv * sizeof(double));
    matvec(d->Ma, d->M, x, nv, nv);
    for (int r = 0; r < nefc; r++) d->ejar[r] = d->Jv[r] - d->earef[r];
  }
  double cost = solver_cost(d, nv, nefc, x, d->Ma, d->ejar);
  int iter;
  for (iter = 0; iter < maxiter; iter++) {
    d->n_newton++;
    if (iter + 1 > d->max_newton) d->max_newton = iter + 1;
    /* gradient and Hessian on the current active set */
    for (int i = 0; i < nv; i++) d->grad[i] = d->Ma[i] - d->qfrc_smooth[i];
    memcpy(d->H, d->M, nv * nv * sizeof(double));
    for (int r = 0; r < nefc; r++) {
      if (!(d->ejar[r] < 0)) continue;
      const double* Jr = d->J + r * nv;
      double f = -d->eD[r] * d->ejar[r];
      for (int i = 0; i < nv; i++) {
        d->grad[i] -= Jr[i] * f;
        if (Jr[i] == 0) continue;
        double di = d->eD[r] * Jr[i];
        for (int j = 0; j < nv; j++) d->H[i * nv + j] += di * Jr[j];
      }
    }
    double gnorm = 0;
    for (int i = 0; i < nv; i++) gnorm += d->grad[i] * d->grad[i];
    if (scale * sqrt(gnorm) < tol) break;
    if (cholesky(d->L, d->H, nv) != 0) break;
    chol_solve(d->Mgrad, d->L, d->grad, nv);
    for (int i = 0; i < nv; i++) d->search[i] = -d->Mgrad[i];
    /* exact line search on the piecewise-quadratic 1-D cost: safeguarded Newton on its derivative */
    matvec(d->Mv, d->M, d->search, nv, nv);
    matvec(d->Jv, d->J, d->search, nefc, nv);
    double g1 = 0, g2 = 0;
    for (int i = 0; i < nv; i++) { g1 += d->search[i] * (d->Ma[i] - d->qfrc_smooth[i]); g2 += d->search[i] * d->Mv[i]; }
    double alpha = 0, lo = 0, hi = -1, d0 = 0;
    for (int ls = 0; ls < 50; ls++) {
      double f1 = g1 + alpha * g2, f2 = g2;
      for (int r = 0; r < nefc; r++) {
        double j = d->ejar[r] + alpha * d->Jv[r];
        if (j < 0) { f1 += d->eD[r] * j * d->Jv[r]; f2 += d->eD[r] * d->Jv[r] * d->Jv[r]; }
      }
      if (ls == 0) d0 = fabs(f1);
      if (fabs(f1) <= 1e-10 * d0) break;
      if (f1 < 0) lo = alpha; else hi = alpha;
      double an = alpha - f1 / f2;
      if (an <= lo || (hi >= 0 && an >= hi)) an = hi >= 0 ? 0.5 * (lo + hi) : 2 * (alpha > 0 ? alpha : 1.0);
      alpha = an;
    }
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { x[i] += alpha * d->search[i]; d->Ma[i] += alpha * d->Mv[i]; }
    for (int r = 0; r < nefc; r++) d->ejar[r] += alpha * d->Jv[r];
    double oldcost = cost;
    cost = solver_cost(d, nv, nefc, x, d->Ma, d->ejar);
    if (scale * (oldcost - cost) < tol) break;
  }
  for (int r = 0; r < nefc; r++) d->eforce[r] = d->ejar[r] < 0 ? -d->eD[r] * d->ejar[r] : 0;
  for (int i = 0; i < nv; i++) {
    double t = 0;
    for (int r = 0; r < nefc; r++) t += d->J[r * nv + i] * d->eforce[r];
    d->qfrc_constraint[i] = t;
  }
}

/* ------------------------------------------------------------------------------------------------
 * mj_forward / mj_step
 * ---------------------------------------------------------------------------------------------- */
static void forward(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  int nv = m->nv;
  d->n_forward++;
  /* position */
  kinematics(m, d);
  com_pos(m, d);
  crb_and_factor(m, d);
  collision(s, d);
  make_constraint(s, d);
  /* velocity */
  com_vel(m, d);
  rne_bias(m, d);
  const double* damp = SUMO_F(m, dof_damping);
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -damp[i] * d->qvel[i];
  matvec(d->evel, d->J, d->qvel, d->nefc, nv);
  for (int r = 0; r < d->nefc; r++)
    d->earef[r] = -d->eB[r] * d->evel[r] - d->eK[r] * d->eimp[r] * (d->epos[r] - d->emargin[r]);
  /* actuation: ctrl clamped to ctrlrange, plain motors with gear */
  memset(d->qfrc_act, 0, nv * sizeof(double));
  const int* adof = SUMO_I(m, actuator_dofid);
  const double* gear = SUMO_F(m, actuator_gear);
  const double* cr = SUMO_F(m, actuator_ctrlrange);
  for (int u = 0; u < m->nu; u++) {
    double c = d->ctrl[u];
    if (c < cr[2 * u]) c = cr[2 * u];
    if (c > cr[2 * u + 1]) c = cr[2 * u + 1];
    d->qfrc_act[adof[u]] += gear[u] * c;
  }
  /* acceleration */
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_act[i];
  if (cholesky(d->L, d->M, nv) != 0) { memset(d->qacc_smooth, 0, nv * sizeof(double)); }
  else chol_solve(d->qacc_smooth, d->L, d->qfrc_smooth, nv);
  /* constraint */
  newton_solve(s, d);
  d->n_contacts += d->ncon;
  d->n_efc += d->nefc;
  if (d->ncon > d->max_ncon) d->max_ncon = d->ncon;
  if (d->nefc > d->max_nefc) d->max_nefc = d->nefc;
}

/* mj_integratePos: qpos '+'= h * vel */
static void integrate_pos(const sumo_model_t* m, double* qpos, const double* vel, double h) {
  const int* jtype = SUMO_I(m, jnt_type);
  const int* jq = SUMO_I(m, jnt_qposadr);
  const int* jd = SUMO_I(m, jnt_dofadr);
  for (int j = 0; j < m->njnt; j++) {
    int qa = jq[j], da = jd[j];
    if (jtype[j] == SUMO_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * vel[da + k];
      double ax[3] = {vel[da + 3], vel[da + 4], vel[da + 5]}, qr[4];
      double ang = h * normalize3(ax);
      axisangle2quat(qr, ax, ang);
      normalize4(qpos + qa + 3);
      mulquat(qpos + qa + 3, qpos + qa + 3, qr);
    } else {
      qpos[qa] += h * vel[da];
    }
  }
}

/* cfrc_ext as mj_rnePostConstraint leaves it (the contact part: no xfrc_applied here), SURVEY.md App. A.9/A.10: for every contact
 * the force in the contact frame is decoded from its four pyramid rows (mju_decodePyramid: normal = sum of the edge forces,
 * tangent k = mu (f_2k - f_2k+1)), rotated to world axes and applied at the contact point: -wrench to body1, +wrench to body2,
 * each as [torque about the subtree CoM of the body's root ; force].  With force sensors in the model MuJoCo 2.1 evaluates this in
 * mj_forward -- once per mj_step, at the step's START state (the RK4 sub-stages skip sensors) -- so after frame_skip steps the data
 * hold the forces of the evaluation that opened the last step. */
static void rne_post(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  const int* gbody = SUMO_I(m, geom_bodyid);
  const int* rootid = SUMO_I(m, body_rootid);
  memset(d->cfrc_ext, 0, (size_t)6 * m->nbody * sizeof(double));
  int r0 = 0;
  while (r0 < d->nefc && d->etype[r0] == 0) r0++;             /* limit rows come first */
  for (int ci = 0; ci < d->ncon; ci++, r0 += 4) {
    if (r0 + 4 > d->nefc) break;                              /* rows cut off by maxefc carry no force */
    const contact_t* c = d->con + ci;
    const double* f = d->eforce + r0;
    const double fc[3] = {f[0] + f[1] + f[2] + f[3], c->mu * (f[0] - f[1]), c->mu * (f[2] - f[3])};
    double F[3];
    for (int k = 0; k < 3; k++) F[k] = c->frame[k] * fc[0] + c->frame[3 + k] * fc[1] + c->frame[6 + k] * fc[2];
    for (int side = 0; side < 2; side++) {
      const int b = gbody[side ? c->g2 : c->g1];
      if (b == 0) continue;                                   /* the world body's entry is observed by nobody: not computed */
      const double sg = side ? 1.0 : -1.0;
      double arm[3], tq[3];
      for (int k = 0; k < 3; k++) arm[k] = c->pos[k] - d->subtree_com[3 * rootid[b] + k];
      cross3(tq, arm, F);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6 * b + k] += sg * tq[k]; d->cfrc_ext[6 * b + 3 + k] += sg * F[k]; }
    }
  }
}

static void mj_step(const so_sim* s, env_t* d) {
  static const double A[3] = {0.5, 0.5, 1.0};
  static const double B[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
  const sumo_model_t* m = &s->m;
  int nq = m->nq, nv = m->nv;
  double h = SUMO_F(m, opt)[SUMO_OPT_TIMESTEP];
  forward(s, d);
  if (s->cfrc_mode) rne_post(s, d);
  memcpy(d->rkX[0], d->qpos, nq * sizeof(double));
  memcpy(d->rkX[0] + nq, d->qvel, nv * sizeof(double));
  memcpy(d->rkF[0], d->qacc, nv * sizeof(double));
  for (int i = 1; i < 4; i++) {
    /* dX = A[i-1] * (velocity of stage i-1, acceleration of stage i-1) */
    for (int k = 0; k < nv; k++) { d->rkdX[k] = A[i - 1] * d->rkX[i - 1][nq + k]; d->rkdX[nv + k] = A[i - 1] * d->rkF[i - 1][k]; }
    memcpy(d->rkX[i], d->rkX[0], (nq + nv) * sizeof(double));
    integrate_pos(m, d->rkX[i], d->rkdX, h);
    for (int k = 0; k < nv; k++) d->rkX[i][nq + k] += h * d->rkdX[nv + k];
    memcpy(d->qpos, d->rkX[i], nq * sizeof(double));
    memcpy(d->qvel, d->rkX[i] + nq, nv * sizeof(double));
    forward(s, d);
    memcpy(d->rkF[i], d->qacc, nv * sizeof(double));
  }
  memset(d->rkdX, 0, 2 * nv * sizeof(double));
  for (int j = 0; j < 4; j++)
    for (int k = 0; k < nv; k++) { d->rkdX[k] += B[j] * d->rkX[j][nq + k]; d->rkdX[nv + k] += B[j] * d->rkF[j][k]; }
  memcpy(d->qpos, d->rkX[0], nq * sizeof(double));
  memcpy(d->qvel, d->rkX[0] + nq, nv * sizeof(double));
  for (int k = 0; k < nv; k++) d->qvel[k] += h * d->rkdX[nv + k];
  integrate_pos(m, d->qpos, d->rkdX, h);
  memcpy(d->warm, d->qacc, nv * sizeof(double)); /* warm start for the next step = last stage's qacc */
}

/* ------------------------------------------------------------------------------------------------
 * counter RNG for resets (Philox4x32-10).  Distribution follows sumo.py:232-253; the reference's
 * MT19937 stream itself (gym seeding) is not reproduced -- see DESIGN.md.
 * ---------------------------------------------------------------------------------------------- */
static void philox4x32(uint32_t out[4], uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static double rng_uniform(uint64_t seed, uint32_t reset_count, uint32_t k) {
  uint32_t o[4];
  philox4x32(o, reset_count, k >> 1, 0u, 0x53554D4Fu, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t a = o[2 * (k & 1)] >> 5, b = o[2 * (k & 1) + 1] >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
#define RNG_NORMAL_BASE 64

static void write_obs(const so_sim* s, const env_t* d, float* obs) {
  /* agents.py:190-214 with cfrc_ext == 0 (MuJoCo >= 2.0 without force sensors),
   * then the time feature of sumo_env.py:68-70. */
  const sumo_model_t* m = &s->m;
  const int *aq = SUMO_I(m, agent_qposadr), *anq = SUMO_I(m, agent_nq), *ad = SUMO_I(m, agent_dofadr),
            *anv = SUMO_I(m, agent_nv), *anb = SUMO_I(m, agent_nbody);
  for (int a = 0; a < 2; a++) {
    float* o = obs + a * s->obs_stride;
    int o_ = 1 - a, k = 0;
    /* get_qpos(): qpos[2] += self._adjust_z on the returned copy (agents.py:155-161) */
    for (int i = 0; i < anq[a]; i++) o[k++] = (float)(i == 2 ? d->qpos[aq[a] + i] + s->adjust_z : d->qpos[aq[a] + i]);
    for (int i = 0; i < anv[a]; i++) o[k++] = (float)d->qvel[ad[a] + i];
    /* |clip(cfrc_ext, +-CFRC_CLIP)| of the own bodies and of the opponent's torso (agents.py:13,192-208); zeros in the default mode */
    const int* ab = SUMO_I(m, agent_bodyadr);
    for (int i = 0; i < 6 * anb[a]; i++) {
      double v = s->cfrc_mode ? d->cfrc_ext[6 * ab[a] + i] : 0.0;
      o[k++] = (float)fabs(v > 100.0 ? 100.0 : (v < -100.0 ? -100.0 : v));
    }
    for (int i = 0; i < 7; i++) o[k++] = (float)(i == 2 ? d->qpos[aq[o_] + i] + s->adjust_z : d->qpos[aq[o_] + i]);   /* opp.get_qpos()[:7] */
    for (int i = 0; i < 6; i++) {
      double v = s->cfrc_mode ? d->cfrc_ext[6 * ab[o_] + i] : 0.0;
      o[k++] = (float)fabs(v > 100.0 ? 100.0 : (v < -100.0 ? -100.0 : v));
    }
    o[k++] = (float)(-1.0 + 2.0 * d->num_steps / 500.0);
    for (; k < s->obs_stride; k++) o[k] = 0.0f;
  }
}

static void reset_env(const so_sim* s, env_t* d) {
  const sumo_model_t* m = &s->m;
  int nq = m->nq, nv = m->nv;
  const int* aq = SUMO_I(m, agent_qposadr);
  uint32_t rc = (uint32_t)d->reset_count;
  memcpy(d->qpos, SUMO_F(m, qpos0), nq * sizeof(double)); /* mj_resetData */
  memset(d->qvel, 0, nv * sizeof(double));
  memset(d->warm, 0, nv * sizeof(double));
  memset(d->cfrc_ext, 0, (size_t)6 * m->nbody * sizeof(double));   /* mj_resetData; the reset observation shows zeros */
  double phi = 2.0 * PI * rng_uniform(d->seed, rc, 0);
  for (int a = 0; a < 2; a++) {
    double ang = phi + a * (2.0 * PI / 2.0);
    d->qpos[aq[a] + 0] = 1.15 * cos(ang);
    d->qpos[aq[a] + 1] = 1.15 * sin(ang);
    d->qpos[aq[a] + 2] = 1.25;
  }
  for (int i = 0; i < nq; i++) d->qpos[i] += -0.1 + 0.2 * rng_uniform(d->seed, rc, 1 + i);
  for (int i = 0; i < nv; i++) {
    double u1 = rng_uniform(d->seed, rc, RNG_NORMAL_BASE + 2 * (i >> 1));
    double u2 = rng_uniform(d->seed, rc, RNG_NORMAL_BASE + 2 * (i >> 1) + 1);
    double rr = sqrt(-2.0 * log(1.0 - u1));
    double z = (i & 1) ? rr * sin(2.0 * PI * u2) : rr * cos(2.0 * PI * u2);
    d->qvel[i] = 0.1 * z;
  }
  /* set_state + forward: the only lasting effect on (qpos, qvel) is the in-place quaternion normalisation */
  const int* jtype = SUMO_I(m, jnt_type);
  const int* jq = SUMO_I(m, jnt_qposadr);
  for (int j = 0; j < m->njnt; j++)
    if (jtype[j] == SUMO_JNT_FREE) normalize4(d->qpos + jq[j] + 3);
  d->num_steps = 0;
  d->ep_ret = 0;
  d->ep_dense = 0;
  d->reset_count++;
}

/* numpy float32 pairwise sum of squares, as np.square(action).sum() evaluates it for n < 128
 * (agents.py:221-222 with a float32 action array). */
static float sumsq_f32(const float* a, int n) {
  if (n < 8) {
    float r = 0.0f; /* numpy starts from -0.0f for floats; irrelevant for squares */
    for (int i = 0; i < n; i++) r += a[i] * a[i];
    return r;
  }
  float r[8];
  for (int j = 0; j < 8; j++) r[j] = a[j] * a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; j++) r[j] += a[i + j] * a[i + j];
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; i++) res += a[i] * a[i];
  return res;
}

/* MuJoCo's bad-value test (mj_checkPos / mj_checkVel / mj_checkAcc [EXT]: NaN or |x| > mjMAXVAL = 1e10 in qpos, qvel, qacc;
 * the warning it raises becomes a MujocoException in the reference, mujoco-py/mujoco_py/builder.py:351-369). */
static int state_is_bad(const so_sim* s, const env_t* d) {
  const sumo_model_t* m = &s->m;
  for (int i = 0; i < m->nq; i++) if (!(fabs(d->qpos[i]) <= 1e10)) return 1;
  for (int i = 0; i < m->nv; i++) if (!(fabs(d->qvel[i]) <= 1e10) || !(fabs(d->warm[i]) <= 1e10)) return 1;
  return 0;
}

static void env_step(const so_sim* s, env_t* d, const float* act, float* obs, double* info, uint8_t* done,
                     double* ep_r, double* ep_dr, int32_t* ep_l) {
  const sumo_model_t* m = &s->m;
  const int *aq = SUMO_I(m, agent_qposadr), *au = SUMO_I(m, agent_uadr), *anu = SUMO_I(m, agent_nu);
  double before[2][2], after[2][2];
  for (int a = 0; a < 2; a++) { before[a][0] = d->qpos[aq[a]]; before[a][1] = d->qpos[aq[a] + 1]; }
  /* do_simulation: ctrl = concat(actions); frame_skip x mj_step */
  for (int a = 0; a < 2; a++)
    for (int i = 0; i < anu[a]; i++) d->ctrl[au[a] + i] = (double)act[a * s->act_stride + i];
  /* a state that fails the bad-value test before or after the frame_skip mj_steps ends the episode: zero rewards, info flag 4,
   * auto-reset (the reference would have raised; MuJoCo itself tests in every mj_step, resets the data and warns) */
  int diverged = state_is_bad(s, d);
  if (!diverged) {
    for (int f = 0; f < m->frame_skip; f++) mj_step(s, d);
    diverged = state_is_bad(s, d);
  }
  double ctrl_r[2];
  for (int a = 0; a < 2; a++) {
    after[a][0] = d->qpos[aq[a]]; after[a][1] = d->qpos[aq[a] + 1];
    ctrl_r[a] = -0.1 * (double)sumsq_f32(act + a * s->act_stride, anu[a]);
  }
  d->num_steps++;
  double lim = m->tatami_size + 0.1;
  int lost[2];
  for (int a = 0; a < 2; a++) {
    const double* p = d->qpos + aq[a];
    double mx = fabs(p[0]) > fabs(p[1]) ? fabs(p[0]) : fabs(p[1]);
    lost[a] = (p[2] + s->adjust_z < 0.29) || (mx >= lim);   /* sumo.py:147-160: get_qpos()[:3], i.e. the adjusted z */
  }
  double dt = SUMO_F(m, opt)[SUMO_OPT_TIMESTEP] * m->frame_skip;
  int dn = 0;
  for (int a = 0; a < 2; a++) {
    int o = 1 - a, flags = 0;
    double* I = info + a * SO_INFO_STRIDE;
    double lose = lost[a] ? -2000.0 : 0.0, win = lost[o] ? 2000.0 : 0.0;
    if (lost[a] || lost[o]) dn = 1;
    if (lost[o]) flags |= 1;
    double main_r = win + lose;
    if (d->num_steps > m->timestep_limit) { main_r += -1000.0; dn = 1; }
    double mv[2] = {(after[a][0] - before[a][0]) / dt, (after[a][1] - before[a][1]) / dt};
    double dir[2] = {after[o][0] - before[a][0], after[o][1] - before[a][1]};
    double nrm = sqrt(dir[0] * dir[0] + dir[1] * dir[1]);
    dir[0] /= nrm; dir[1] /= nrm;
    double proj = mv[0] * dir[0] + mv[1] * dir[1];
    double move = (proj > 0 ? proj : 0.0) * 0.1;
    double push = -10.0 * exp(-sqrt(after[o][0] * after[o][0] + after[o][1] * after[o][1]));
    double shaping = ctrl_r[a] + push + move;
    I[0] = ctrl_r[a]; I[1] = lose; I[2] = win; I[3] = main_r; I[4] = move; I[5] = push; I[6] = shaping;
    I[7] = (double)flags;
  }
  if (diverged) {
    for (int k = 0; k < 2 * SO_INFO_STRIDE; k++) info[k] = 0.0;
    info[7] = info[SO_INFO_STRIDE + 7] = 4.0;
    dn = 1;
    d->n_diverged++;
  }
  /* wrapper bookkeeping (sumo_env.py:44-65, monitor.py:60-78): agent 0 only */
  d->ep_ret += info[3] + info[6];
  d->ep_dense += info[6];
  if (!diverged && dn && info[3] == -1000.0) { info[7] += 2.0; info[SO_INFO_STRIDE + 7] += 2.0; }
  done[0] = done[1] = (uint8_t)dn;
  if (dn) {
    *ep_r = d->ep_ret; *ep_dr = d->ep_dense; *ep_l = d->num_steps;
    reset_env(s, d); /* subproc_vec_env.py:13-16: the reset observation replaces the terminal one */
  } else {
    *ep_r = 0; *ep_dr = 0; *ep_l = 0;
  }
  write_obs(s, d, obs);
}

/* ------------------------------------------------------------------------------------------------
 * public API
 * ---------------------------------------------------------------------------------------------- */
static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

so_sim* so_create(const void* blob, size_t nbytes, int num_envs) {
  so_sim* s = (so_sim*)calloc(1, sizeof(so_sim));
  s->blob = malloc(nbytes);
  memcpy(s->blob, blob, nbytes);
  int rc = sumo_model_parse(&s->m, s->blob, nbytes);
  if (rc != 0) { snprintf(g_err, sizeof g_err, "bad model blob (%d)", rc); free(s->blob); free(s); return NULL; }
  const sumo_model_t* m = &s->m;
  s->N = num_envs;
  s->maxcon = MAXCON_CAP;
  s->maxefc = 4 * MAXCON_CAP + 2 * m->njnt;
  int od = 0, adim = 0;
  for (int a = 0; a < m->nagent; a++) {
    int o = SUMO_I(m, agent_nq)[a] + SUMO_I(m, agent_nv)[a] + 6 * SUMO_I(m, agent_nbody)[a] + 14;
    if (o > od) od = o;
    if (SUMO_I(m, agent_nu)[a] > adim) adim = SUMO_I(m, agent_nu)[a];
  }
  s->obs_stride = od; s->act_stride = adim;
  s->env = (env_t*)calloc(num_envs, sizeof(env_t));
  int nq = m->nq, nv = m->nv, nb = m->nbody, nj = m->njnt, ng = m->ngeom, ne = s->maxefc;
  for (int e = 0; e < num_envs; e++) {
    env_t* d = s->env + e;
    d->qpos = dalloc(nq); d->qvel = dalloc(nv); d->warm = dalloc(nv); d->ctrl = dalloc(m->nu);
    d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb);
    d->ximat = dalloc(9 * nb); d->xanchor = dalloc(3 * nj); d->xaxis = dalloc(3 * nj); d->gxpos = dalloc(3 * ng);
    d->gxmat = dalloc(9 * ng); d->subtree_com = dalloc(3 * nb); d->cinert = dalloc(10 * nb); d->crb = dalloc(10 * nb);
    d->cdof = dalloc(6 * nv); d->cdof_dot = dalloc(6 * nv); d->cvel = dalloc(6 * nb); d->cacc = dalloc(6 * nb);
    d->cfrc = dalloc(6 * nb); d->cfrc_ext = dalloc(6 * nb); d->M = dalloc(nv * nv); d->L = dalloc(nv * nv); d->qfrc_bias = dalloc(nv);
    d->qfrc_passive = dalloc(nv); d->qfrc_act = dalloc(nv); d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv);
    d->qacc = dalloc(nv); d->qfrc_constraint = dalloc(nv);
    d->con = (contact_t*)calloc(MAXCON_CAP, sizeof(contact_t));
    d->J = dalloc((size_t)ne * nv); d->epos = dalloc(ne); d->emargin = dalloc(ne); d->ediag = dalloc(ne);
    d->eR = dalloc(ne); d->eD = dalloc(ne); d->eK = dalloc(ne); d->eB = dalloc(ne); d->eimp = dalloc(ne);
    d->evel = dalloc(ne); d->earef = dalloc(ne); d->ejar = dalloc(ne); d->eforce = dalloc(ne);
    d->etype = (int*)calloc(ne, sizeof(int));
    d->Ma = dalloc(nv); d->grad = dalloc(nv); d->Mgrad = dalloc(nv); d->search = dalloc(nv); d->Mv = dalloc(nv);
    d->Jv = dalloc(ne); d->H = dalloc(nv * nv); d->tmpv = dalloc(9 * nv);
    for (int i = 0; i < 4; i++) { d->rkX[i] = dalloc(nq + nv); d->rkF[i] = dalloc(nv); }
    d->rkdX = dalloc(2 * nv);
    memcpy(d->qpos, SUMO_F(m, qpos0), nq * sizeof(double));
    d->seed = (uint64_t)e;
  }
  return s;
}

void so_destroy(so_sim* s) {
  if (!s) return;
  for (int e = 0; e < s->N; e++) {
    env_t* d = s->env + e;
    double* ptrs[] = {d->qpos, d->qvel, d->warm, d->ctrl, d->xpos, d->xquat, d->xmat, d->xipos, d->ximat, d->xanchor,
                      d->xaxis, d->gxpos, d->gxmat, d->subtree_com, d->cinert, d->crb, d->cdof, d->cdof_dot, d->cvel,
                      d->cacc, d->cfrc, d->M, d->L, d->qfrc_bias, d->qfrc_passive, d->qfrc_act, d->qfrc_smooth,
                      d->qacc_smooth, d->qacc, d->qfrc_constraint, d->J, d->epos, d->emargin, d->ediag, d->eR, d->eD,
                      d->eK, d->eB, d->eimp, d->evel, d->earef, d->ejar, d->eforce, d->Ma, d->grad, d->Mgrad,
                      d->search, d->Mv, d->Jv, d->H, d->tmpv, d->cfrc_ext, d->rkX[0], d->rkX[1], d->rkX[2], d->rkX[3], d->rkF[0],
                      d->rkF[1], d->rkF[2], d->rkF[3], d->rkdX};
    for (size_t i = 0; i < sizeof ptrs / sizeof ptrs[0]; i++) free(ptrs[i]);
    free(d->con); free(d->etype);
  }
  free(s->env); free(s->blob); free(s);
}

int so_dims(const so_sim* s, int* o) {
  const sumo_model_t* m = &s->m;
  o[0] = m->nq; o[1] = m->nv; o[2] = m->nu; o[3] = m->nbody; o[4] = m->njnt; o[5] = m->ngeom; o[6] = m->npair;
  o[7] = m->nagent; o[8] = s->obs_stride; o[9] = s->act_stride;
  return 0;
}

int so_set_jbcap(so_sim* s, int jbcap) {
  if (jbcap < 0) return -1;
  s->jbcap = jbcap;
  return 0;
}
int so_set_cfrc_mode(so_sim* s, int mode) {
  if (mode != 0 && mode != 1) return -1;
  s->cfrc_mode = mode;
  return 0;
}
int so_set_adjust_z(so_sim* s, double adjust_z) {
  s->adjust_z = adjust_z;
  return 0;
}
int so_set_maxcon(so_sim* s, int maxcon) {
  if (maxcon < 1 || maxcon > MAXCON_CAP) return -1;
  s->maxcon = maxcon;
  return 0;
}

int so_reset(so_sim* s, const uint64_t* seeds, const uint8_t* mask, float* obs) {
  for (int e = 0; e < s->N; e++) {
    if (mask && !mask[e]) continue;
    env_t* d = s->env + e;
    if (seeds) { d->seed = seeds[e]; d->reset_count = 0; }
    reset_env(s, d);
    if (obs) write_obs(s, d, obs + (size_t)e * 2 * s->obs_stride);
  }
  return 0;
}

int so_step(so_sim* s, const float* actions, float* obs, double* info, uint8_t* done, double* ep_r, double* ep_dr,
            int32_t* ep_l, int nthreads) {
  int N = s->N;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int e = 0; e < N; e++)
    env_step(s, s->env + e, actions + (size_t)e * 2 * s->act_stride, obs + (size_t)e * 2 * s->obs_stride,
             info + (size_t)e * 2 * SO_INFO_STRIDE, done + 2 * e, ep_r + e, ep_dr + e, ep_l + e);
  return 0;
}

int so_get_state(const so_sim* s, double* qpos, double* qvel, double* warm, int32_t* counters) {
  const sumo_model_t* m = &s->m;
  for (int e = 0; e < s->N; e++) {
    const env_t* d = s->env + e;
    if (qpos) memcpy(qpos + (size_t)e * m->nq, d->qpos, m->nq * sizeof(double));
    if (qvel) memcpy(qvel + (size_t)e * m->nv, d->qvel, m->nv * sizeof(double));
    if (warm) memcpy(warm + (size_t)e * m->nv, d->warm, m->nv * sizeof(double));
    if (counters) { counters[2 * e] = d->num_steps; counters[2 * e + 1] = d->reset_count; }
  }
  return 0;
}
int so_set_state(so_sim* s, const double* qpos, const double* qvel, const double* warm, const int32_t* counters) {
  const sumo_model_t* m = &s->m;
  for (int e = 0; e < s->N; e++) {
    env_t* d = s->env + e;
    if (qpos) memcpy(d->qpos, qpos + (size_t)e * m->nq, m->nq * sizeof(double));
    if (qvel) memcpy(d->qvel, qvel + (size_t)e * m->nv, m->nv * sizeof(double));
    if (warm) memcpy(d->warm, warm + (size_t)e * m->nv, m->nv * sizeof(double));
    if (counters) { d->num_steps = counters[2 * e]; d->reset_count = counters[2 * e + 1]; }
  }
  return 0;
}
int so_set_seeds(so_sim* s, const uint64_t* seeds) {
  for (int e = 0; e < s->N; e++) s->env[e].seed = seeds[e];
  return 0;
}

int so_forward(so_sim* s, int e, const double* ctrl) {
  env_t* d = s->env + e;
  if (ctrl) memcpy(d->ctrl, ctrl, s->m.nu * sizeof(double));
  forward(s, d);
  if (s->cfrc_mode) rne_post(s, d);
  return 0;
}
int so_mj_step(so_sim* s, int e, const double* ctrl, int n) {
  env_t* d = s->env + e;
  if (ctrl) memcpy(d->ctrl, ctrl, s->m.nu * sizeof(double));
  for (int i = 0; i < n; i++) mj_step(s, d);
  return 0;
}

int so_get_array(so_sim* s, int e, const char* name, double* out, int cap) {
  env_t* d = s->env + e;
  const sumo_model_t* m = &s->m;
  int nv = m->nv, nb = m->nbody, n = 0;
  const double* src = NULL;
#define ARR(nm, ptr, cnt) if (!strcmp(name, nm)) { src = ptr; n = cnt; }
  ARR("qpos", d->qpos, m->nq) ARR("qvel", d->qvel, nv) ARR("qacc", d->qacc, nv) ARR("qacc_smooth", d->qacc_smooth, nv)
  ARR("warm", d->warm, nv) ARR("M", d->M, nv * nv) ARR("qfrc_bias", d->qfrc_bias, nv) ARR("xpos", d->xpos, 3 * nb)
  ARR("xquat", d->xquat, 4 * nb) ARR("xipos", d->xipos, 3 * nb) ARR("gxpos", d->gxpos, 3 * m->ngeom)
  ARR("subtree_com", d->subtree_com, 3 * nb) ARR("qfrc_constraint", d->qfrc_constraint, nv)
  ARR("qfrc_smooth", d->qfrc_smooth, nv) ARR("cvel", d->cvel, 6 * nb) ARR("efc_J", d->J, d->nefc * nv)
  ARR("cfrc_ext", d->cfrc_ext, 6 * s->m.nbody) ARR("efc_force", d->eforce, d->nefc) ARR("efc_aref", d->earef, d->nefc) ARR("efc_R", d->eR, d->nefc)
  ARR("efc_pos", d->epos, d->nefc) ARR("efc_jar", d->ejar, d->nefc)
#undef ARR
  if (!strcmp(name, "contacts")) { /* per contact: dist, pos3, normal3, g1, g2 */
    n = 9 * d->ncon;
    if (n > cap) return -n;
    for (int i = 0; i < d->ncon; i++) {
      double* o = out + 9 * i;
      o[0] = d->con[i].dist; memcpy(o + 1, d->con[i].pos, 24); memcpy(o + 4, d->con[i].frame, 24);
      o[7] = d->con[i].g1; o[8] = d->con[i].g2;
    }
    return n;
  }
  if (!strcmp(name, "counts")) { if (cap < 3) return -3; out[0] = d->ncon; out[1] = d->nefc; out[2] = d->ncon_dropped; return 3; }
  if (!src) return 0;
  if (n > cap) return -n;
  memcpy(out, src, n * sizeof(double));
  return n;
}

int so_stats(const so_sim* s, double* o) {
  for (int k = 0; k < 11; k++) o[k] = 0;
  for (int e = 0; e < s->N; e++) {
    const env_t* d = s->env + e;
    o[0] += d->n_forward; o[1] += d->n_newton; o[2] += d->n_contacts; o[3] += d->n_efc;
    if (d->max_ncon > o[4]) o[4] = d->max_ncon;
    if (d->max_nefc > o[5]) o[5] = d->max_nefc;
    if (d->max_newton > o[6]) o[6] = d->max_newton;
    o[7] += d->ncon_dropped;
    o[8] += d->n_diverged;
    o[9] += d->n_cb3; o[10] += d->n_rodcap;
  }
  return 0;
}
